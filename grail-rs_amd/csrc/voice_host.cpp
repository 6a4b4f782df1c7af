// voice_host.cpp — host-side parameter algebra of the C ABI: the once-per-voice /
// once-per-phoneme table preparation that the reference also does on the host
// (SURVEY.md §3.4, §8a5/a9).  Plain IEEE binary32, no FMA (x86-64 baseline has
// none; -ffp-contract=off is passed anyway).
//
// Reference: SynthesisElem::{new, silent, new_phoneme, blend, resample}
// src/lib.rs:343-440, VoiceStorage::get :664-671, voices::generic()
// src/voices/generic.rs:5-40, MKPHON src/voices/mod.rs:7-14.
#include <cmath>
#include <cstring>

#include "../../include/grail_hip.h"

namespace {

constexpr int NF = GRAIL_NUM_FORMANTS;

// f32::min of Rust core: the non-NaN operand wins.
inline float rust_min(float a, float b) { return std::fmin(a, b); }

// Array::blend, src/lib.rs:135-137
inline float lerp(float a, float b, float alpha) { return a * (1.0f - alpha) + b * alpha; }

struct PhonemeSpec {
    float freq[NF], bw[NF], smooth[NF], turb[NF], breath[NF], amp[NF];
};

// src/voices/generic.rs:9-32 — MKPHON argument order: freq, bw, smooth, turb, breath, amp
const PhonemeSpec kGenericA = {
    {910.0f, 1271.0f, 2851.0f, 3213.0f, 1200.0f, 2000.0f, 3000.0f, 4000.0f},
    {60.0f, 160.0f, 180.0f, 200.0f, 100.0f, 100.0f, 100.0f, 100.0f},
    {1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f},
    {0.2f, 0.2f, 0.1f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f},
    {0.5f, 0.2f, 0.05f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f},
    {0.3f, 0.3f, 0.2f, 0.1f, 0.0f, 0.0f, 0.0f, 0.0f},
};
const PhonemeSpec kGenericE = {
    {910.0f, 1871.0f, 2851.0f, 3213.0f, 1200.0f, 2000.0f, 3000.0f, 4000.0f},
    {80.0f, 180.0f, 180.0f, 200.0f, 100.0f, 100.0f, 100.0f, 100.0f},
    {1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f, 1600.0f},
    {0.2f, 0.4f, 0.4f, 0.4f, 0.4f, 0.4f, 0.4f, 0.4f},
    {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 0.1f, 0.1f},
    {0.5f, 0.4f, 0.3f, 0.2f, 0.0f, 0.0f, 0.0f, 0.0f},
};

}  // namespace

extern "C" {

void grail_elem_silent(grail_synthesis_elem *out)
{
    out->frequency = 0.25f;
    for (int i = 0; i < NF; ++i) {
        out->formant_freq[i] = 0.25f;
        out->formant_bw[i] = 0.25f;
        out->formant_smooth[i] = 0.25f;
        out->formant_breath[i] = 0.0f;
        out->formant_turb[i] = 0.0f;
        out->formant_amp[i] = 0.0f;
    }
}

void grail_elem_resample(grail_synthesis_elem *e, float old_sample_rate, float new_sample_rate)
{
    const float scale = old_sample_rate / new_sample_rate;  // src/lib.rs:420
    e->frequency = rust_min(e->frequency * scale, 0.5f);    // :427
    for (int i = 0; i < NF; ++i) {
        const float scaled = e->formant_freq[i] * scale;    // :423
        if (scaled > 0.5f) e->formant_amp[i] = 0.0f;        // :433-435 (tested before the clamp)
        e->formant_freq[i] = rust_min(scaled, 0.5f);        // :428
        e->formant_bw[i] *= scale;                          // :429
        e->formant_smooth[i] *= scale;                      // :430
    }
}

void grail_elem_new_phoneme(grail_synthesis_elem *out, const float *formant_freq,
                            const float *formant_bw, const float *formant_smooth,
                            const float *formant_turb, const float *formant_breath,
                            const float *formant_amp)
{
    float total = 0.0f;  // Array::sum: left fold, src/lib.rs:123-125
    for (int i = 0; i < NF; ++i) total += formant_amp[i];
    out->frequency = 0.0f;
    for (int i = 0; i < NF; ++i) {
        out->formant_freq[i] = formant_freq[i];
        out->formant_bw[i] = formant_bw[i];
        out->formant_smooth[i] = formant_smooth[i];
        out->formant_breath[i] = formant_breath[i];
        out->formant_turb[i] = formant_turb[i];
        out->formant_amp[i] = formant_amp[i] / total;  // unit gain, :398
    }
    grail_elem_resample(out, 1.0f, GRAIL_DEFAULT_SAMPLE_RATE);  // :400
}

void grail_elem_new(grail_synthesis_elem *out, float sample_rate, float frequency,
                    const float *formant_freq, const float *formant_smooth,
                    const float *formant_bw, const float *formant_breath,
                    const float *formant_turb, const float *formant_amp)
{
    out->frequency = frequency;
    std::memcpy(out->formant_freq, formant_freq, sizeof out->formant_freq);
    std::memcpy(out->formant_bw, formant_bw, sizeof out->formant_bw);
    std::memcpy(out->formant_smooth, formant_smooth, sizeof out->formant_smooth);
    std::memcpy(out->formant_breath, formant_breath, sizeof out->formant_breath);
    std::memcpy(out->formant_turb, formant_turb, sizeof out->formant_turb);
    std::memcpy(out->formant_amp, formant_amp, sizeof out->formant_amp);
    grail_elem_resample(out, 1.0f, sample_rate);  // src/lib.rs:363
}

void grail_elem_blend(grail_synthesis_elem *out, const grail_synthesis_elem *self,
                      const grail_synthesis_elem *other, float alpha)
{
    grail_synthesis_elem r;
    r.frequency = lerp(self->frequency, other->frequency, alpha);
    for (int i = 0; i < NF; ++i) {
        r.formant_freq[i] = lerp(self->formant_freq[i], other->formant_freq[i], alpha);
        r.formant_bw[i] = lerp(self->formant_bw[i], other->formant_bw[i], alpha);
        r.formant_smooth[i] = lerp(self->formant_smooth[i], other->formant_smooth[i], alpha);
        r.formant_breath[i] = lerp(self->formant_breath[i], other->formant_breath[i], alpha);
        r.formant_turb[i] = lerp(self->formant_turb[i], other->formant_turb[i], alpha);
        r.formant_amp[i] = lerp(self->formant_amp[i], other->formant_amp[i], alpha);
    }
    *out = r;
}

void grail_voice_generic(grail_voice *out)
{
    out->sample_rate = GRAIL_DEFAULT_SAMPLE_RATE;
    const PhonemeSpec *specs[GRAIL_NUM_VOICED] = {&kGenericA, &kGenericE};
    for (int p = 0; p < GRAIL_NUM_VOICED; ++p)
        grail_elem_new_phoneme(&out->phonemes[p], specs[p]->freq, specs[p]->bw, specs[p]->smooth,
                               specs[p]->turb, specs[p]->breath, specs[p]->amp);
    out->center_frequency = 120.0f / GRAIL_DEFAULT_SAMPLE_RATE;
    out->jitter_frequency = 16.0f / GRAIL_DEFAULT_SAMPLE_RATE;
    out->jitter_delta_frequency = 6.0f / GRAIL_DEFAULT_SAMPLE_RATE;
    out->jitter_delta_formant_frequency = 6.0f / GRAIL_DEFAULT_SAMPLE_RATE;
    out->jitter_delta_amplitude = 0.2f;
}

void grail_voice_generic_at(grail_voice *out, float sample_rate)
{
    grail_voice_generic(out);
    if (sample_rate == GRAIL_DEFAULT_SAMPLE_RATE) return;
    for (int p = 0; p < GRAIL_NUM_VOICED; ++p)
        grail_elem_resample(&out->phonemes[p], GRAIL_DEFAULT_SAMPLE_RATE, sample_rate);
    out->sample_rate = sample_rate;
    out->center_frequency = 120.0f / sample_rate;
    out->jitter_frequency = 16.0f / sample_rate;
    out->jitter_delta_frequency = 6.0f / sample_rate;
    out->jitter_delta_formant_frequency = 6.0f / sample_rate;
    out->jitter_delta_amplitude = 0.2f;
}

int grail_voice_get(const grail_voice *voice, int32_t phoneme, grail_synthesis_elem *out)
{
    if (phoneme < GRAIL_PH_FIRST_VOICED || phoneme >= GRAIL_PH_FIRST_VOICED + GRAIL_NUM_VOICED)
        return 0;  // Silence | Stop | Glide => None, src/lib.rs:666
    *out = voice->phonemes[phoneme - GRAIL_PH_FIRST_VOICED];
    return 1;
}

void grail_shard_range(uint64_t n_utt, uint32_t rank, uint32_t world, uint64_t *begin,
                       uint64_t *end)
{
    if (world == 0) world = 1;
    if (rank >= world) rank = world - 1;
    // 128-bit products so n_utt * rank cannot wrap
    *begin = (uint64_t)(((unsigned __int128)n_utt * rank) / world);
    *end = (uint64_t)(((unsigned __int128)n_utt * (rank + 1)) / world);
}

}  // extern "C"
