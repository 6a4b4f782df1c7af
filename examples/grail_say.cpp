// grail_say — the shape of the reference's examples/cli.rs (text in, WAV out, timing line),
// with the synthesis on an MI355X through include/grail.hpp.   usage: grail_say [-o out.wav] text...
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "grail.hpp"

int main(int argc, char **argv)
{
    std::string out_path, text;
    for (int i = 1; i < argc; ++i) {
        if ((!std::strcmp(argv[i], "-o") || !std::strcmp(argv[i], "--output")) && i + 1 < argc) out_path = argv[++i];
        else text += (text.empty() ? "" : " ") + std::string(argv[i]);
    }
    if (text.empty()) {
        std::fprintf(stderr, "usage: grail_say [-o out.wav] text...\n");
        return 2;
    }
    try {
        const grail::Voice voice = grail::voices::generic();       // 44.1 kHz, as the CLI
        grail::Gpu gpu(0, {voice});
        const auto t0 = std::chrono::steady_clock::now();
        const auto pcm = gpu.say({text});                           // cli.rs:175-184
        const auto us = std::chrono::duration_cast<std::chrono::microseconds>(
                            std::chrono::steady_clock::now() - t0).count();
        std::printf("\"%s\"\n%.2f seconds of audio, generated in %lld microseconds\n", text.c_str(),
                    pcm[0].size() / voice.sample_rate, (long long)us);   // cli.rs:189-193
        if (!out_path.empty()) {
            std::printf("Writing generated sound to %s\n", out_path.c_str());
            gpu.save_wav(out_path, pcm[0], (uint32_t)voice.sample_rate);
        }
    } catch (const grail::Error &e) {
        std::fprintf(stderr, "grail_say: %s (status %d)\n", e.what(), e.status);
        return 1;
    }
    return 0;
}
