// kernels.h — device-side data layout and launcher prototypes (internal).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace grail {

constexpr int NF = 8;            // NUM_FORMANTS, reference src/lib.rs:24
constexpr int ELEM_FLOATS = 49;  // SynthesisElem, reference src/lib.rs:316-337
constexpr int NUM_VOICED = 2;    // VoiceStorage {a, e}, reference src/lib.rs:653-659
constexpr int PH_FIRST_VOICED = 3;
constexpr int SPLIT_MAX_CHUNKS = 64;   // time-split fast kernels: chunks per utterance

// field offsets inside a 49-float SynthesisElem (declared order)
constexpr int F_FREQ = 1, F_BW = 9, F_SMOOTH = 17, F_BREATH = 25, F_TURB = 33, F_AMP = 41;

// One segment as the kernel reads it: 16 B, one dwordx4 load.
// Phoneme mode: bit-identical to grail_phoneme_elem (PhonemeElem, src/lib.rs:961-973),
//   `elem` holds the Phoneme discriminant and the Selector runs on the device.
// Elem mode: `elem` is the row of the batch's elem table, or -1 for None
//   (SequenceElem.elem, src/lib.rs:817), `frequency` repeats the elem's own.
struct DevSeg {
    int32_t elem;
    float length;
    float blend_length;
    float frequency;
};
static_assert(sizeof(DevSeg) == 16, "DevSeg must be one dwordx4");

// Per-voice scalars (Voice, src/lib.rs:696-717) + where its phoneme elems sit
// in the elem table. 32 B.
struct DevVoice {
    float sample_rate;
    float jitter_frequency;
    float jitter_delta_frequency;
    float jitter_delta_formant_frequency;
    float jitter_delta_amplitude;
    uint32_t elem_base;  // row of phonemes.a in the voice elem table
    uint32_t warmup;     // time-split fast kernels: samples after which a filter state started from zero has
                         // decayed below half an ulp of full scale (a multiple of 64; 0: the voice does not qualify)
    uint32_t pad;
};
static_assert(sizeof(DevVoice) == 32, "DevVoice layout");

struct SynthArgs {
    const DevSeg *segs;
    const uint32_t *seg_offsets;  // [n_utt + 1]
    const uint32_t *voice_ids;    // [n_utt] or nullptr (voice 0)
    const uint32_t *seeds;        // [n_utt] or nullptr (seed 0)
    const uint32_t *perm;         // [n_utt] or nullptr: launch slot -> utterance (ragged batches: sorted by
                                  // length on upload, so the lanes of a wave end together)
    const float *elems;           // phoneme mode: voice elem table; elem mode: batch elem table
    const DevVoice *voices;
    float *out;
    int16_t *out_pcm16;           // not nullptr: rows of i16 PCM instead (examples/cli.rs:49 fused
                                  // into the tile flush); `out` is then unused
    uint32_t *out_len;            // may be nullptr
    uint32_t *truncated;          // [0]: set to 1 when an utterance is cut at out_stride;
                                  // [1]: += wave-steps that ran the IEEE-division body
                                  // [2]: += wave-tiles rendered in fast arithmetic, [3]: += general wave-steps
    uint64_t out_stride;          // samples between rows
    uint64_t cap;                 // samples a row may receive in this launch (<= out_stride)
    uint32_t n_utt;
    uint32_t n_voices;
    uint32_t phoneme_mode;        // 1: run the Selector on the device
    uint32_t skip_silent;         // 1: skip the band-pass of formant vectors proven silent (same bits)
    uint32_t half_capable;        // host hint: every voice has amplitude 0 in formants 5-8 of every
                                  // phoneme (phoneme batches) / every elem of the batch has (caller-built elems,
                                  // with parameters that keep them at +0: live4_elems_ok), so the half-live loops can be used
    uint32_t live4;               // host-verified: formants 5-8 contribute exactly +0.0 for the whole
                                  // batch (see voice_analysis.cpp live4_ok); selects the NFA = 4 kernels
    uint32_t pipe;                // live4 batches small enough to leave SIMDs idle: the four-wave
                                  // pipelined workgroups (synth_kernel<..., PIPE>)
    uint32_t fast;                // tolerance-mode arithmetic in calm tiles (option "arithmetic" = 1): 1 = coefficients
                                  // interpolated, 2 = the reference's own coefficients at every sample (MID)
    uint32_t any_blend;           // host hint: some segment has a blend length that is not +-2^k
    uint32_t cohabit;             // lane kernels on 2 / 4 / 8 lanes per utterance: the instantiation built for two waves per
                                  // SIMD (launches of more waves than the device has SIMDs; launch_plan.cpp family_cohabits)
    const uint32_t *len_bound;    // time-split kernels: per utterance an upper bound of its length in samples, or nullptr — a
                                  // chunk's lane whose utterance ends before the chunk begins renders nothing
    uint32_t pipe_fill;           // pipelined workgroups, one-shot: utterances per workgroup (0: all 16 / 8 slots) — a small batch
                                  // of rows that differ in length is spread thinly, so that a workgroup's tiles hold few events
    uint32_t fold_from;           // two waves per SIMD, at most two rounds of the device: workgroups from this index on take the
                                  // launch slots in reverse order (0: none) — see synth_kernel.h
    uint32_t *state;              // resumable synthesis: state[word][lane] or nullptr (one-shot)
    uint64_t state_stride;        // lanes of the launch (= state_lanes())
    uint32_t resume;              // 1: load the state first (not the first call of a stream)
    // live streams (grail_stream_open_live: the lazy source of examples/interactive.rs:31-38): the segments of
    // utterance u sit in a ring, segs[u * ring_cap + (i & (ring_cap - 1))] for its i-th segment ever appended;
    // seg_counts[u] of them have been appended so far, and while seg_open[u] != 0 more may follow: a Sequencer that
    // needs a segment which is not there yet PAUSES (src/lib.rs:866-888 pulls iter.next() on demand) instead of
    // ending the utterance.  seg_offsets is unused then.
    uint32_t ring_cap;            // 0: not a live stream
    const uint32_t *seg_counts;   // [n_utt]
    const uint32_t *seg_open;     // [n_utt]
    uint32_t *seg_consumed;       // [n_utt] out: segments the Sequencer has pulled so far
    // time-split fast kernels (synth_kernel<..., SPLIT>): chunk k of every utterance is the samples
    // [split_bounds[k], split_bounds[k + 1]) (multiples of 64; the last bound is `cap`), one lane each
    uint32_t split_chunks;        // K, 0: not a time-split launch
    uint32_t split_warmup;        // caller-built elems: the warm-up length of the batch (0: the lane's voice has it, DevVoice::warmup)
    uint32_t split_bounds[SPLIT_MAX_CHUNKS + 1];
};

struct LenArgs {
    const DevSeg *segs;
    const uint32_t *seg_offsets;
    const uint32_t *voice_ids;
    const DevVoice *voices;
    uint32_t *out_len;
    uint32_t n_utt;
    uint32_t n_voices;
    uint32_t max_len;
};

// the instantiation the calling thread's last launch_synth started, e.g. "synth_kernel<L=1,T=32,...>"
const char *last_kernel_name();
// lanes_per_utt in {1, 2, 4, 8}; returns hipSuccess or the launch error.
hipError_t launch_synth(const SynthArgs &args, int lanes_per_utt, hipStream_t stream);
hipError_t launch_lengths(const LenArgs &args, hipStream_t stream);
// live streams: utterance u's new segments new_segs[new_offsets[u] .. new_offsets[u + 1]) go to the next slots of its
// ring and counts[u] grows by their number; elem mode (new_elems != nullptr): new_elems[i] is new_segs[i]'s elem (49
// floats) and the segment's `elem` field is rewritten to the ring slot's row of ring_elems (or stays -1 for None)
hipError_t launch_ring_append(DevSeg *ring, float *ring_elems, uint32_t *counts, uint32_t ring_cap, const DevSeg *new_segs,
                              const float *new_elems, const uint32_t *new_offsets, uint32_t n_utt, hipStream_t stream);
// small batches, fast arithmetic: one workgroup per utterance, lanes = time, recurrences by parallel scan
// (scan_kernels.hip).  args.live4 selects two formant-pair waves instead of four.
hipError_t launch_scan(const SynthArgs &args, hipStream_t stream);
// f32 rows -> i16 PCM rows (examples/cli.rs:49); only the first len[u] (<= max_len) samples of row u
hipError_t launch_pcm16(const float *in, uint64_t in_stride, const uint32_t *len, uint32_t n_utt,
                        uint32_t max_len, int16_t *out, uint64_t out_stride, hipStream_t stream);
// per-row digest (bit-pattern sum mod 2^64, max |x|, count of NaN/Inf) of rendered rows
hipError_t launch_digest(const float *in, uint64_t in_stride, const uint32_t *len, uint32_t n_utt,
                         unsigned long long *sums, float *maxabs, uint32_t *nonfinite,
                         hipStream_t stream);
// per-row distance of two renderings: max |a-b|, sum of squared differences, structural mismatches
hipError_t launch_compare(const float *a, const float *b, uint64_t stride, const uint32_t *len_a,
                          const uint32_t *len_b, uint32_t n_utt, float *maxdiff, double *sumsq, uint32_t *bad,
                          hipStream_t stream);
// resumable synthesis: words per lane and lanes per launch of the state buffer
uint32_t state_words(int lanes_per_utt);
uint64_t state_lanes(uint32_t n_utt, int lanes_per_utt);

}  // namespace grail
