// text_front.cpp — the host front half of the chain, once per
// character / phoneme, exactly as in the reference where it also runs on the host:
//   Transcriber::next  src/lib.rs:1116-1191   (.transcribe(): :1197-1204, SILENCE :1114)
//   Intonator::next    src/lib.rs:1057-1075
//   languages::generic()  src/languages/mod.rs:4-34
// plus the RIFF writer of examples/cli.rs:28-67.  No synthesis arithmetic lives here.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/grail_hip.h"

namespace {

// One TranscriptionRule (src/lib.rs:1030-1036) as UTF-32.
struct RuleView {
    const uint32_t *str;
    uint32_t len;
    const int32_t *ph;
    uint32_t n_ph;
};

inline uint32_t to_ascii_lowercase(uint32_t c) { return (c >= 'A' && c <= 'Z') ? (c | 0x20u) : c; }

// The iterator adapter, kept as a class with the reference's fields: a Peekable<chars>
// (text + cursor), the ruleset, case flag and the pending phoneme buffer.
class Transcriber {
public:
    Transcriber(const uint32_t *text, uint32_t n, const grail_rule *rules, uint32_t n_rules,
                bool case_sensitive, bool leading_silence)
        : text_(text), n_(n), rules_(rules), n_rules_(n_rules), case_sensitive_(case_sensitive)
    {
        static const int32_t kSilence[1] = {GRAIL_PH_SILENCE};
        silence_ = kSilence;
        buf_ = leading_silence ? silence_ : nullptr;   // :1201
        buf_len_ = leading_silence ? 1 : 0;
    }

    // Option<Phoneme>: false == None
    bool next(int32_t *out)
    {
        uint32_t lo = 0, hi = n_rules_, index = 0;      // :1120-1122
        while (buf_len_ == 0) {                          // :1125
            if (cur_ >= n_) return false;                // peek()? :1127-1133
            const uint32_t ch = case_sensitive_ ? text_[cur_] : to_ascii_lowercase(text_[cur_]);
            // :1140-1150: first rule (within [lo,hi)) whose index-th char is >= ch / > ch;
            // a rule shorter than index+1 sorts before everything for the lower bound
            const uint32_t new_lo = partition(lo, hi, [&](const grail_rule &r) {
                return index >= r.string_len || r.string[index] < ch;
            });
            const uint32_t new_hi = partition(lo, hi, [&](const grail_rule &r) {
                return index < r.string_len && r.string[index] <= ch;
            });
            const bool exhausted_rule = rules_[lo].string_len == index;
            if (new_lo >= new_hi && exhausted_rule) {    // :1153-1155 emit the matched rule
                set_buffer(rules_[lo].phonemes, rules_[lo].n_phonemes);
            } else if (new_lo >= new_hi) {               // :1156-1161 garbled: silence, skip char
                set_buffer(silence_, 1);
                ++cur_;
            } else {                                     // :1162-1180 keep narrowing
                lo = new_lo;
                hi = new_hi;
                ++index;
                ++cur_;
                if (cur_ >= n_) {
                    if (rules_[lo].string_len == index) set_buffer(rules_[lo].phonemes, rules_[lo].n_phonemes);
                    else set_buffer(silence_, 1);
                }
            }
        }
        if (buf_len_ == 0) return false;                 // buffer.get(0) == None :1183
        *out = buf_[0];
        ++buf_;
        --buf_len_;
        return true;
    }

private:
    template <typename Pred>
    uint32_t partition(uint32_t lo, uint32_t hi, Pred pred) const
    {   // slice::partition_point on rules_[lo..hi), returned as an absolute index
        uint32_t a = lo, b = hi;
        while (a < b) {
            const uint32_t mid = a + (b - a) / 2;
            if (pred(rules_[mid])) a = mid + 1;
            else b = mid;
        }
        return a;
    }
    void set_buffer(const int32_t *p, uint32_t n)
    {
        buf_ = p;
        buf_len_ = n;
        // a rule without phonemes would spin forever on an empty buffer in the reference's
        // `while self.buffer.is_empty()` only if the text also ended; mirror: loop continues
    }

    const uint32_t *text_;
    uint32_t n_;
    uint32_t cur_ = 0;
    const grail_rule *rules_;
    uint32_t n_rules_;
    bool case_sensitive_;
    const int32_t *buf_ = nullptr;
    uint32_t buf_len_ = 0;
    const int32_t *silence_;
};

// str::chars(): UTF-8 -> Unicode scalar values (invalid bytes become U+FFFD)
std::vector<uint32_t> decode_utf8(const char *s)
{
    std::vector<uint32_t> out;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(s);
    while (*p) {
        uint32_t c = *p;
        int extra = c < 0x80 ? 0 : (c >> 5) == 0x6 ? 1 : (c >> 4) == 0xE ? 2 : (c >> 3) == 0x1E ? 3 : -1;
        if (extra < 0) { out.push_back(0xFFFD); ++p; continue; }
        if (extra) c &= (0x3Fu >> extra);
        ++p;
        bool ok = true;
        for (int i = 0; i < extra; ++i) {
            if ((*p & 0xC0) != 0x80) { ok = false; break; }
            c = (c << 6) | (*p & 0x3F);
            ++p;
        }
        out.push_back(ok ? c : 0xFFFD);
    }
    return out;
}

// languages::generic(), src/languages/mod.rs:4-34
const uint32_t kA[] = {'a'}, kE[] = {'e'}, kI[] = {'i'}, kII[] = {'i', 'i'}, kOUI[] = {'o', 'u', 'i'},
               kP[] = {'p'};
const int32_t pA[] = {GRAIL_PH_A}, pE[] = {GRAIL_PH_E}, pEA[] = {GRAIL_PH_E, GRAIL_PH_A},
              pAEA[] = {GRAIL_PH_A, GRAIL_PH_E, GRAIL_PH_A}, pS[] = {GRAIL_PH_SILENCE};
const grail_rule kGenericRules[] = {
    {kA, 1, pA, 1}, {kE, 1, pE, 1}, {kI, 1, pA, 1}, {kII, 2, pEA, 2}, {kOUI, 3, pAEA, 3}, {kP, 1, pS, 1},
};

}  // namespace

extern "C" {

uint32_t grail_language_generic(const grail_rule **rules, int *case_sensitive)
{
    if (rules) *rules = kGenericRules;
    if (case_sensitive) *case_sensitive = 0;   // mod.rs:6
    return sizeof(kGenericRules) / sizeof(kGenericRules[0]);
}

int grail_transcribe(const uint32_t *text, uint32_t text_len, const grail_rule *rules,
                     uint32_t n_rules, int case_sensitive, int leading_silence,
                     int32_t *out_phonemes, uint32_t cap, uint32_t *n_out)
{
    if ((!text && text_len) || !rules || n_rules == 0 || !n_out) return GRAIL_ERR_INVALID_ARG;
    Transcriber t(text, text_len, rules, n_rules, case_sensitive != 0, leading_silence != 0);
    uint32_t n = 0;
    int32_t ph;
    while (t.next(&ph)) {
        if (out_phonemes && n < cap) out_phonemes[n] = ph;
        ++n;
    }
    *n_out = n;
    return (out_phonemes && n > cap) ? GRAIL_ERR_BUFFER_TOO_SMALL : GRAIL_OK;
}

int grail_intonate(const grail_voice *voice, const int32_t *phonemes, uint32_t n,
                   grail_phoneme_elem *out)
{
    if (!voice || (!phonemes && n) || (!out && n)) return GRAIL_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) {   // src/lib.rs:1068-1073: constants, no intonation yet
        out[i].phoneme = phonemes[i];
        out[i].length = 0.5f;
        out[i].blend_length = 0.5f;
        out[i].frequency = voice->center_frequency;
    }
    return GRAIL_OK;
}

int grail_text_to_phoneme_elems(const grail_voice *voice, const char *text_utf8,
                                grail_phoneme_elem *out, uint32_t cap, uint32_t *n_out)
{
    if (!voice || !text_utf8 || !n_out) return GRAIL_ERR_INVALID_ARG;
    const std::vector<uint32_t> cps = decode_utf8(text_utf8);
    const grail_rule *rules;
    int cs;
    const uint32_t n_rules = grail_language_generic(&rules, &cs);
    Transcriber t(cps.data(), (uint32_t)cps.size(), rules, n_rules, cs != 0, true);
    std::vector<int32_t> ph;
    int32_t p;
    while (t.next(&p)) ph.push_back(p);
    *n_out = (uint32_t)ph.size();
    if (!out) return GRAIL_OK;
    if (ph.size() > cap) return GRAIL_ERR_BUFFER_TOO_SMALL;
    return grail_intonate(voice, ph.data(), (uint32_t)ph.size(), out);
}

// examples/cli.rs:28-67: 44-byte RIFF/WAVE header, PCM 16 bit mono, then the samples.
int grail_wav_write_i16(const char *path, const int16_t *pcm, uint32_t n, uint32_t sample_rate)
{
    if (!path || (!pcm && n)) return GRAIL_ERR_INVALID_ARG;
    FILE *f = std::fopen(path, "wb");
    if (!f) return GRAIL_ERR_INVALID_ARG;
    unsigned char h[44];
    auto le32 = [](unsigned char *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; };
    auto le16 = [](unsigned char *p, uint16_t v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); };
    std::memcpy(h, "RIFF", 4);
    le32(h + 4, 36u + n * 2u);           // file size :35
    std::memcpy(h + 8, "WAVEfmt ", 8);
    le32(h + 16, 16);                    // sub chunk size
    le16(h + 20, 1);                     // PCM
    le16(h + 22, 1);                     // mono
    le32(h + 24, sample_rate);
    le32(h + 28, sample_rate * 2u);      // byte rate :42
    le16(h + 32, 2);                     // block align
    le16(h + 34, 16);                    // bits per sample
    std::memcpy(h + 36, "data", 4);
    le32(h + 40, n * 2u);                // section size :46
    bool ok = std::fwrite(h, 1, 44, f) == 44;
    std::vector<unsigned char> body((size_t)n * 2);
    for (uint32_t i = 0; i < n; ++i) le16(&body[(size_t)i * 2], (uint16_t)pcm[i]);
    ok = ok && std::fwrite(body.data(), 1, body.size(), f) == body.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? GRAIL_OK : GRAIL_ERR_INVALID_ARG;
}

}  // extern "C"
