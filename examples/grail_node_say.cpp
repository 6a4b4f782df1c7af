// grail_node_say — the reference's one call for a whole job (examples/cli.rs:175-184) over every GPU of a node, through
// grail::Node (include/grail.hpp): each argument is one utterance, the batch is sharded over the devices, and the rows
// are checked against the same call on the first device alone (they must be the same bits).
//   usage: grail_node_say [--devices 0,1,...] [--without-rccl] text...
// --without-rccl: the voice table is installed per context instead of by ncclBroadcast — for a --devices list that names
// one GPU several times (RCCL refuses such a communicator); never the default.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "grail.hpp"

int main(int argc, char **argv)
{
    std::vector<int> devices;
    std::vector<std::string> texts;
    bool without_rccl = false;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {
            for (const char *p = argv[++i]; *p;) {
                devices.push_back((int)std::strtol(p, const_cast<char **>(&p), 10));
                if (*p == ',') ++p;
            }
        } else if (!std::strcmp(argv[i], "--without-rccl")) {
            without_rccl = true;
        } else {
            texts.push_back(argv[i]);
        }
    }
    if (devices.empty()) devices.push_back(0);
    if (texts.empty()) {
        std::fprintf(stderr, "usage: grail_node_say [--devices 0,1,...] [--without-rccl] text...\n");
        return 2;
    }
    try {
        const grail::Voice voice = grail::voices::generic();       // 44.1 kHz, as the CLI
        grail::Node node(devices, {voice}, without_rccl);
        const auto t0 = std::chrono::steady_clock::now();
        const auto pcm = node.say(texts);
        const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        size_t samples = 0;
        for (const auto &row : pcm) samples += row.size();
        std::printf("%zu utterances over %u device slots (RCCL ranks %u): %.2f seconds of audio, generated in %lld microseconds\n",
                    pcm.size(), node.size(), node.rccl_ranks(), samples / voice.sample_rate, (long long)us);
        grail::Gpu gpu(devices[0], {voice});
        const auto one = gpu.say(texts);
        size_t differing = 0;
        for (size_t u = 0; u < pcm.size(); ++u)
            if (pcm[u].size() != one[u].size() ||
                std::memcmp(pcm[u].data(), one[u].data(), pcm[u].size() * sizeof(float)) != 0)
                ++differing;
        std::printf("rows that differ from one device's: %zu\n", differing);
        return differing ? 1 : 0;
    } catch (const grail::Error &e) {
        std::fprintf(stderr, "grail_node_say: %s (status %d)\n", e.what(), e.status);
        return 1;
    }
}
