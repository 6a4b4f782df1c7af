// grail_interactive — the shape of the reference's examples/interactive.rs without the sound card:
// every line read from stdin is spoken, pulled from the GPU a buffer at a time (10 ms by default, the
// role of the cpal callback at interactive.rs:42-48), and the f32 samples go to stdout as raw
// little-endian PCM.   usage: grail_interactive [chunk_samples] < lines.txt > out.f32
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "grail.hpp"

int main(int argc, char **argv)
{
    const uint32_t chunk = argc > 1 ? (uint32_t)std::strtoul(argv[1], nullptr, 10) : 441;
    try {
        const grail::Voice voice = grail::voices::generic();       // interactive.rs:28
        grail::Gpu gpu(0, {voice});
        std::string line;
        while (std::getline(std::cin, line)) {                      // interactive.rs:55-63
            if (line.empty()) continue;
            grail::Stream stream(gpu, {grail::Utterance{grail::phoneme_elems(voice, line), 0, 0}}, chunk);
            std::vector<std::vector<float>> part;
            size_t total = 0, calls = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (stream.next(part)) {
                std::fwrite(part[0].data(), sizeof(float), part[0].size(), stdout);
                total += part[0].size();
                ++calls;
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::fprintf(stderr, "\"%s\": %zu samples in %zu buffers of %u, %.2f ms per buffer (%.1f ms of audio each)\n",
                         line.c_str(), total, calls, chunk, calls ? ms / calls : 0.0, 1e3 * chunk / voice.sample_rate);
        }
    } catch (const grail::Error &e) {
        std::fprintf(stderr, "grail_interactive: %s (status %d)\n", e.what(), e.status);
        return 1;
    }
    return 0;
}
