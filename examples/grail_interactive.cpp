// grail_interactive — the reference's examples/interactive.rs without the sound card.
//
// There ONE iterator chain runs for the whole session (interactive.rs:31-38): its source is
// `repeat_with(|| receiver.try_recv().unwrap_or(' '))`, so text arrives while the audio callback (:42-48) is pulling
// samples, a ' ' — which no rule matches: a Silence phoneme (src/lib.rs:1158-1163) — is handed over whenever the chain
// asks and nothing is waiting, and carrier phase, noise seed, jitter and filter state carry from one line into the next.
// Here the chain is a grail::LiveStream on the GPU: every line is transcribed as `line.trim().chars() + ' '`
// (interactive.rs:78-80; the leading Silence of `.transcribe()`, src/lib.rs:1201, once per session), its phonemes wait in
// a queue, and whenever the Sequencer asks for its next segment (LiveStream::next() comes back short) it is fed ONE
// phoneme: the next of the queue, or a Silence.  Audio is pulled `chunk` samples at a time, the role of the cpal
// callback, and written to stdout as raw little-endian f32.
//
// Without a sound card time is the audio itself: a line "@1.5 text" arrives once 1.5 s of audio have been rendered, a line
// without a stamp arrives at once.  After the last line has been spoken the session runs `tail` seconds longer (the
// reference never ends) and the source is closed.   usage: grail_interactive [chunk_samples] [tail_seconds] < script > out.f32
// Every phoneme fed is reported on stderr ("fed <phoneme> at <sample>"), so that a test can render the same list with
// the oracle in one piece and compare bit for bit.
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <iostream>
#include <string>
#include <vector>

#include "grail.hpp"

namespace {

struct Line {
    double at;          // seconds of audio after which the line arrives
    std::string text;
};

// line.trim().chars().chain(Some(' '))  interactive.rs:78-80, through `.transcribe(generic()).intonate(generic(), voice)`
std::vector<grail::PhonemeElem> speak(const grail::Voice &voice, const std::string &line, bool first)
{
    const grail_rule *rules = nullptr;
    int case_sensitive = 0;
    const uint32_t n_rules = grail_language_generic(&rules, &case_sensitive);
    std::string t = line;
    while (!t.empty() && (t.back() == ' ' || t.back() == '\t' || t.back() == '\r')) t.pop_back();
    size_t lead = 0;
    while (lead < t.size() && (t[lead] == ' ' || t[lead] == '\t')) ++lead;
    std::vector<uint32_t> cps;
    for (size_t i = lead; i < t.size(); ++i) cps.push_back((unsigned char)t[i]);      // (the generic language is ASCII)
    cps.push_back(' ');
    std::vector<int32_t> ph(4 * cps.size() + 8);
    uint32_t n = 0;
    grail::check(grail_transcribe(cps.data(), (uint32_t)cps.size(), rules, n_rules, case_sensitive, first ? 1 : 0, ph.data(),
                                  (uint32_t)ph.size(), &n));
    std::vector<grail::PhonemeElem> out(n);
    if (n) grail::check(grail_intonate(&voice, ph.data(), n, out.data()));
    return out;
}

}  // namespace

int main(int argc, char **argv)
{
    const uint32_t chunk = argc > 1 ? (uint32_t)std::strtoul(argv[1], nullptr, 10) : 441;
    const double tail = argc > 2 ? std::strtod(argv[2], nullptr) : 1.0;
    if (chunk == 0 || !(tail >= 0.0)) {        // (a chunk of 0 samples — or text strtoul makes 0 of — would never feed the stream)
        std::fprintf(stderr, "usage: grail_interactive [samples per pull >= 1 (default 441)] [seconds of tail >= 0 (default 1)]\n");
        return 2;
    }
    try {
        const grail::Voice voice = grail::voices::generic();       // interactive.rs:33-36
        grail::Gpu gpu(0, {voice});
        std::vector<Line> script;
        std::string raw;
        while (std::getline(std::cin, raw)) {                        // interactive.rs:77-81
            Line l{0.0, raw};
            if (!raw.empty() && raw[0] == '@') {
                char *end = nullptr;
                l.at = std::strtod(raw.c_str() + 1, &end);
                l.text = end ? std::string(end) : std::string();
            }
            script.push_back(l);
        }
        grail::LiveStream chain(gpu, chunk, 0, 0);                   // .jitter(0, ...) interactive.rs:37
        std::deque<grail::PhonemeElem> waiting;                      // what the transcriber has, the Sequencer has not
        const grail::PhonemeElem silence = speak(voice, "", false).at(0);   // ' ': no rule matches
        size_t next_line = 0, total = 0, fed = 0;
        bool first = true, closed = false;
        double close_at = -1.0;
        const char *names[] = {"Silence", "Stop", "Glide", "A", "E"};
        for (;;) {
            const double now = (double)total / voice.sample_rate;
            while (next_line < script.size() && script[next_line].at <= now) {      // the channel receives a line
                for (const grail::PhonemeElem &p : speak(voice, script[next_line].text, first)) waiting.push_back(p);
                first = false;
                ++next_line;
            }
            const std::vector<float> part = chain.next();
            std::fwrite(part.data(), sizeof(float), part.size(), stdout);
            total += part.size();
            if (part.size() == chunk) continue;
            if (closed) {
                if (part.empty()) break;                              // the chain has returned None
                continue;
            }
            // the Sequencer asks its source for the next segment
            if (next_line == script.size() && waiting.empty()) {
                if (close_at < 0.0) close_at = (double)total / voice.sample_rate + tail;
                if ((double)total / voice.sample_rate >= close_at) {
                    chain.finish();
                    closed = true;
                    continue;
                }
            }
            grail::PhonemeElem p = silence;
            if (!waiting.empty()) {
                p = waiting.front();
                waiting.pop_front();
            } else if (first) {
                // nothing has been said yet: the chain's very first pull is the leading Silence of .transcribe()
                first = false;
            }
            chain.append({p});
            std::fprintf(stderr, "fed %s at %zu\n", names[p.phoneme], total);
            ++fed;
        }
        std::fprintf(stderr, "session: %zu samples (%.2f s), %zu phonemes fed, %zu lines, buffers of %u\n", total,
                     total / voice.sample_rate, fed, script.size(), chunk);
    } catch (const grail::Error &e) {
        std::fprintf(stderr, "grail_interactive: %s (status %d)\n", e.what(), e.status);
        return 1;
    }
    return 0;
}
