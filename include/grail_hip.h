/*
 * grail_hip.h — C ABI of the MI355X-native (gfx950) batched implementation of
 * the grail-rs per-sample synthesis hot path.
 *
 * What it replaces.  grail-rs has no FFI layer; its boundary for this path is
 * the iterator-adapter API (reference = /root/reference):
 *
 *     phoneme_elems.into_iter()
 *         .select(voice)          // src/lib.rs:1013  Selector::next   :990
 *         .sequence(voice)        // src/lib.rs:941   Sequencer::next  :859
 *         .jitter(seed, voice)    // src/lib.rs:786   Jitter::next     :753
 *         .synthesize()           // src/lib.rs:587   Synthesize::next :497
 *         .collect::<Vec<f32>>()
 *
 * grail_synthesize_batch() is that expression evaluated for N independent
 * utterances at once on one GPU; every entry point below cites the reference
 * item it stands for.  INTEGRATION.md shows the Rust `extern "C"` block and
 * the IntoSynthesizeBatch trait shim a maintainer would add.
 *
 * Types are plain-old-data with the reference's declared field order; no
 * torch / HIP types appear in any signature (device memory is `void *`).
 *
 * Determinism contract: the samples of utterance u are a pure
 * function of (its segments, its voice, its jitter seed) — independent of the
 * batch size, its position in the batch, the lane mapping and the GPU count —
 * and are bit-identical to the reference's IEEE-754 binary32 arithmetic
 * (no FMA contraction, correctly rounded division, denormals kept).  That is the
 * default, "arithmetic" = 0.  With grail_set_option(ctx, "arithmetic", 1) the
 * samples are within GRAIL_FAST_TOLERANCE x max(1, the utterance's largest
 * |sample|) of those bits instead (lengths, segment boundaries, noise wraps and
 * saw edges still exactly the reference's).  Two tolerance tiers serve it, chosen from
 * the sharpness of the resonances of the voices a batch uses (grail_fast_sharpness):
 * up to GRAIL_FAST_SHARPNESS_LIMIT the filter coefficients are interpolated across
 * sub-tiles (2.6x the exact mode on the headline batch); sharper voices get the
 * reference's own band-pass coefficients at every sample and fast arithmetic for
 * the rest (1.25x - 1.55x; at most 17.3 * 2^-23 off on 1000 random voice tables of
 * any sharpness, profiles/r04_middle_tier.txt); beyond
 * GRAIL_FAST_SHARPNESS_LIMIT_EXACT_COEFFICIENTS, and wherever an exact kernel is the
 * faster way to render a block, the exact bits (they satisfy any tolerance).
 * Fast-mode samples are a pure function of (the utterance, the kernel family):
 * every lane decides from its own state, so they do not depend on the batch size,
 * the position in the batch or the other utterances of the batch AS LONG AS THE
 * KERNEL FAMILY IS THE SAME.  The family is what the host picks for the BLOCK of rows
 * an utterance is rendered in: a batch is cut into blocks by size (grail_plan_blocks
 * predicts the cut: whole rounds of the one-lane kernels, then the rest on whatever
 * suits its size) and each block takes the time-parallel scan kernel (few rows),
 * the time-split kernels (their chunk grid follows the block's size and the batch's
 * longest utterance; "time_split_chunks" / "time_split_span_samples" pin it) or the
 * lane kernels ("lanes_per_utterance" pins the mapping; a pinned option also keeps
 * the batch in ONE block).  The host-output calls render a batch in blocks of up to
 * 4096 rows (2 GB) and choose the family for that block size, the short last block
 * included: there the "batch size" is min(n_utt, 4096).
 * A caller that needs the reference's contract in fast mode too — an utterance's
 * samples a pure function of (segments, voice, seed), whatever batch it is part of —
 * pins ONE family: "lanes_per_utterance" = 1 renders every batch size with the fast
 * one-lane kernels (batch-invariant by construction; it gives up the time-split and
 * scan kernels' speed on batches that do not fill the machine).
 *
 * There is no CPU fallback: every compute entry point fails with
 * GRAIL_ERR_NO_DEVICE when no HIP device is usable, and grail_create() refuses a
 * device whose architecture is not gfx950 (the library holds gfx950 code objects
 * only).  The launch policy is derived from the device: every capacity is a multiple
 * of hipDeviceProp_t::multiProcessorCount (256 on a whole MI355X, 32 on a CPX
 * partition), every crossover follows the utterances' length
 * (profiles/r04_duration_sweep.txt).
 */
#ifndef GRAIL_HIP_H
#define GRAIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point, an option name or the meaning of an option changes; a binding compares it with
 * grail_abi_version() of the library it loaded BEFORE its first call (the Python and Rust bindings do).
 *   1: rounds 1-2.   2: round 3-4 — grail_device_pci_bus_id, grail_time_split_warmup / _grid, grail_fast_sharpness,
 *   grail_plan_blocks, grail_stream_open_live / _append / _append_elems / _finish / _pending; option "kernel_variant" removed, "scan_debug" in
 *   development builds only; "arithmetic" = 1 is served up to a sharpness of the voice table.
 *   3: round 5 — grail_length_bound; options "two_waves_per_simd", "pipeline_spread", "pipeline_round32" = 2; "ragged_plan" also weighs the scan and
 *   time-split kernels by the rows.
 *   4: round 6 — grail_node_* (one call, every GPU of the node); option "packed_launch_order". */
#define GRAIL_ABI_VERSION 4
/* fast mode ("arithmetic" = 1): bound on |fast - exact| per sample, full scale = 1.0; k * 2^-23 */
#define GRAIL_FAST_TOLERANCE_ULPS 64
#define GRAIL_FAST_TOLERANCE (GRAIL_FAST_TOLERANCE_ULPS * 1.1920928955078125e-07f)
/* fast mode is served for voices whose grail_fast_sharpness() is at most this (predicted deviation, units of 2^-23) */
#define GRAIL_FAST_SHARPNESS_LIMIT 28.0
/* sharper voices get the second tolerance tier — the reference's own band-pass coefficients at every sample, fast
 * arithmetic elsewhere: at most 17.3 * 2^-23 from the reference on 1000 random voice tables of any sharpness on the
 * device, 16.3 in the CPU experiment over 3000 (profiles/r04_middle_tier.txt) — up to this sharpness; beyond it the exact
 * kernels */
#define GRAIL_FAST_SHARPNESS_LIMIT_EXACT_COEFFICIENTS 1024.0

/* src/lib.rs:24  NUM_FORMANTS, src/lib.rs:21 DEFAULT_SAMPLE_RATE */
#define GRAIL_NUM_FORMANTS 8
#define GRAIL_DEFAULT_SAMPLE_RATE 44100.0f

/* Phoneme: src/lib.rs:632-649 with make_phonemes!(A a test, E e test) :686-689.
 * Discriminants follow the declaration order of the Rust enum. */
typedef enum grail_phoneme {
    GRAIL_PH_SILENCE = 0, /* src/lib.rs:635 */
    GRAIL_PH_STOP    = 1, /* src/lib.rs:640 */
    GRAIL_PH_GLIDE   = 2, /* src/lib.rs:643 */
    GRAIL_PH_A       = 3,
    GRAIL_PH_E       = 4,
    GRAIL_PH_COUNT   = 5
} grail_phoneme;
/* number of VoiceStorage fields (src/lib.rs:653-659): a, e */
#define GRAIL_NUM_VOICED 2
#define GRAIL_PH_FIRST_VOICED GRAIL_PH_A

typedef enum grail_status {
    GRAIL_OK                   = 0,
    GRAIL_ERR_INVALID_ARG      = -1,
    GRAIL_ERR_NO_DEVICE        = -2, /* no usable HIP device: there is NO CPU fallback */
    GRAIL_ERR_HIP              = -3, /* a hip* call failed; see grail_last_error() */
    GRAIL_ERR_BUFFER_TOO_SMALL = -4, /* >=1 utterance did not end within out_stride samples */
    GRAIL_ERR_OUT_OF_MEMORY    = -5,
    GRAIL_ERR_RCCL             = -6,
    GRAIL_ERR_NO_VOICES        = -7  /* grail_set_voices() has not been called */
} grail_status;

/* SynthesisElem: src/lib.rs:316-337, 49 x f32 in declared order.
 * `Array` (src/lib.rs:88) is float[8]. */
typedef struct grail_synthesis_elem {
    float frequency;
    float formant_freq[GRAIL_NUM_FORMANTS];
    float formant_bw[GRAIL_NUM_FORMANTS];
    float formant_smooth[GRAIL_NUM_FORMANTS];
    float formant_breath[GRAIL_NUM_FORMANTS];
    float formant_turb[GRAIL_NUM_FORMANTS];
    float formant_amp[GRAIL_NUM_FORMANTS];
} grail_synthesis_elem;

/* Voice: src/lib.rs:696-717; `phonemes` is VoiceStorage {a, e} (src/lib.rs:653-659). */
typedef struct grail_voice {
    float                sample_rate;
    grail_synthesis_elem phonemes[GRAIL_NUM_VOICED];
    float                center_frequency;
    float                jitter_frequency;
    float                jitter_delta_frequency;
    float                jitter_delta_formant_frequency;
    float                jitter_delta_amplitude;
} grail_voice;

/* PhonemeElem: src/lib.rs:961-973 (the Selector's input item). */
typedef struct grail_phoneme_elem {
    int32_t phoneme;      /* grail_phoneme */
    float   length;       /* seconds */
    float   blend_length; /* seconds */
    float   frequency;    /* normalised to the sample rate */
} grail_phoneme_elem;

/* SequenceElem: src/lib.rs:814-824 (the Sequencer's input item);
 * Option<SynthesisElem> is (has_elem, elem). */
typedef struct grail_sequence_elem {
    int32_t              has_elem;
    grail_synthesis_elem elem;
    float                length;
    float                blend_length;
} grail_sequence_elem;

typedef struct grail_ctx   grail_ctx;   /* one per (process, GPU); owns a HIP stream */
typedef struct grail_batch grail_batch; /* inputs of one batch, resident in HBM */
typedef struct grail_stream grail_stream; /* resumable synthesis of one batch */
typedef struct grail_node  grail_node;  /* one process, several GPUs: a grail_ctx and a host thread per GPU */

/* flags of grail_synthesize_batch*() */
#define GRAIL_OUT_HOST   0u /* `out` is host memory (pageable or pinned) */
#define GRAIL_OUT_DEVICE 1u /* `out` is device memory of ctx's GPU; no copy is made */

/* ---- library ----------------------------------------------------------- */
int         grail_abi_version(void);
const char *grail_status_string(int status);
/* text of the last failure on this thread ("" if none) */
const char *grail_last_error(void);

/* ---- host-side parameter algebra (per voice / per phoneme, once) ------- */
/* SynthesisElem::silent()  src/lib.rs:367-377 */
void grail_elem_silent(grail_synthesis_elem *out);
/* SynthesisElem::new_phoneme(freq, bw, smooth, turb, breath, amp)  src/lib.rs:381-401
 * (== voices::MKPHON, src/voices/mod.rs:7-14); each argument is float[8]. */
void grail_elem_new_phoneme(grail_synthesis_elem *out, const float *formant_freq,
                            const float *formant_bw, const float *formant_smooth,
                            const float *formant_turb, const float *formant_breath,
                            const float *formant_amp);
/* SynthesisElem::new(sample_rate, frequency, freq, smooth, bw, breath, turb, amp)
 * src/lib.rs:343-364 — note the argument order differs from new_phoneme. */
void grail_elem_new(grail_synthesis_elem *out, float sample_rate, float frequency,
                    const float *formant_freq, const float *formant_smooth,
                    const float *formant_bw, const float *formant_breath,
                    const float *formant_turb, const float *formant_amp);
/* SynthesisElem::resample(old, new) in place  src/lib.rs:418-440 */
void grail_elem_resample(grail_synthesis_elem *elem, float old_sample_rate,
                         float new_sample_rate);
/* SynthesisElem::blend(self, other, alpha)  src/lib.rs:404-414 */
void grail_elem_blend(grail_synthesis_elem *out, const grail_synthesis_elem *self,
                      const grail_synthesis_elem *other, float alpha);
/* voices::generic()  src/voices/generic.rs:5-40 */
void grail_voice_generic(grail_voice *out);
/* generic() carried to another sample rate (the reference ships 44.1 kHz only):
 * every phoneme .resample(44100, rate) (src/lib.rs:418 through for_all :674),
 * sample_rate = rate, scalars 120/rate, 16/rate, 6/rate, 6/rate, 0.2
 * (cf. src/voices/generic.rs:34-38).  SURVEY.md §8d "48 kHz voice". */
void grail_voice_generic_at(grail_voice *out, float sample_rate);
/* VoiceStorage::get(phoneme)  src/lib.rs:664-671; returns 1 and fills *out for a
 * voiced phoneme, 0 for Silence/Stop/Glide (None). */
int grail_voice_get(const grail_voice *voice, int32_t phoneme, grail_synthesis_elem *out);

/* ---- context ----------------------------------------------------------- */
/* Binds to HIP device `device` (>= 0) and creates a stream.
 * GRAIL_ERR_NO_DEVICE when HIP reports no such device or its architecture is not gfx950. */
int grail_create(int device, grail_ctx **out);
int grail_destroy(grail_ctx *ctx);
int grail_device_count(int *count);
/* hipDeviceGetPCIBusId of ctx's GPU ("0000:05:00.0", NUL-terminated; cap >= 16).  Lets a launcher prove
 * that its ranks sit on distinct GPUs. */
int grail_device_pci_bus_id(grail_ctx *ctx, char *out, size_t cap);
/* Uploads the voice table (the `voice` argument of .select/.sequence/.jitter,
 * src/lib.rs:1013, 941, 786) to HBM.  Utterances refer to it by voice id. */
int grail_set_voices(grail_ctx *ctx, const grail_voice *voices, uint32_t n_voices);
int grail_get_voices(grail_ctx *ctx, grail_voice *voices, uint32_t cap, uint32_t *n_voices);
/* Options (grail_set_option / grail_get_option; int64 values).  What each one may change: "bits: fast" = the samples of a
 * tolerance-mode rendering, within GRAIL_FAST_TOLERANCE; every other option changes time only.  With "arithmetic" = 0 no
 * option ever changes a result bit.  What each is worth on an MI355X is measured in DESIGN.md section 4, not here.
 *   "arithmetic"            0 (default) exact: every sample the reference's binary32 bits.  1 fast: the tolerance mode —
 *                           |fast - reference| <= GRAIL_FAST_TOLERANCE per sample, identical lengths, the discontinuous
 *                           state (Sequencer clock, jitter phase, carrier phase, both LCGs) exact; served by the tier the
 *                           voices' sharpness allows (below).  2: the second tier whatever the voices.  THE option that
 *                           changes result bits.
 *   "fast_sharpness_limit"  (default GRAIL_FAST_SHARPNESS_LIMIT) first tier (filter coefficients interpolated) for batches
 *                           whose voices / elems have a grail_fast_sharpness() of at most this.  bits: fast.
 *   "fast_exact_coefficients" 1 (default) / 0: sharper batches, up to "fast_sharpness_limit_exact_coefficients" (default
 *                           GRAIL_FAST_SHARPNESS_LIMIT_EXACT_COEFFICIENTS), get the second tier (the reference's own band-pass
 *                           coefficients at every sample); 0 or beyond: the exact kernels.  bits: fast.
 *   "lanes_per_utterance"   0 (default) auto, or 1 / 2 / 4 / 8: wavefront lanes that share an utterance's formants; pins the
 *                           lane kernels (no pipelined workgroups, no composite cut).  bits: fast (the kernel family).
 *   "skip_silent_formants"  1 (default) / 0: formants that provably contribute exactly +0.0 (amplitude 0 in both elems of a
 *                           segment pair, band-pass state 0) are skipped, and not laid out at all where the voice table and
 *                           the batch guarantee it for formants 5-8; 0: all eight evaluated literally.
 *   "small_batch_pipeline"  1 (default) / 0: small exact blocks run four-wave pipelined workgroups; streams of that size too.
 *   "pipeline_round32"      1 (default) / 0 / 2: ... in rounds of 32 samples while one workgroup per compute unit suffices and
 *                           the rows are of one length / never / whatever the rows (rounds of 16 otherwise).
 *   "pipeline_spread"       1 (default) / 0: ... of rows that differ in length hold as few utterances each as give every compute
 *                           unit two workgroups (a tile with an event of one utterance costs the whole workgroup).
 *   "pipeline4_max_groups", "pipeline8_max_groups"  (default -1: two per compute unit) workgroups a block may need to take them.
 *   "time_parallel_scan"    1 (default) / 0: fast, first tier: few utterances run one workgroup each, lanes = time (parallel
 *                           scans).  "time_parallel_scan_max_utterances" (-1 auto): hard upper limit;
 *                           "time_parallel_scan_split_max_utterances" (-1 auto): up to here the three-stage flavour.  bits: fast.
 *   "time_split"            1 (default) / 0: fast, both tiers: up to half a device's lanes' worth of utterances are cut along
 *                           their time axis, one lane per chunk (filters warmed up from zero: grail_time_split_warmup).
 *                           "time_split_min_utterances" (-1 auto), "time_split_chunks" (0 auto, 2..64) and
 *                           "time_split_span_samples" (0: the longest utterance) pin the grid (grail_time_split_grid);
 *                           "time_split_ff_cost_permille" (default 165): a fast-forwarded sample against a rendered one.  bits: fast.
 *   "composite_launches"    1 (default) / 0: a batch is cut into blocks with a kernel family each (grail_plan_blocks); 0: one
 *                           launch per call.  bits: fast (a row follows its block's family).
 *   "row_groups"            1 (default) / 0 / 2: rows the lean kernel families cannot take (a segment shorter than two samples,
 *                           a non-finite length, blend length or pitch) are planned apart where cheaper / never / always.
 *   "ragged_plan"           1 (default) / 0: batches whose utterances differ in length are planned by the rows' lengths and
 *                           events (grail_plan_ragged_blocks): one launch of each lane mapping in several rounds, and in fast
 *                           arithmetic the scan and time-split kernels, are weighed against the cut by size; a fast request may
 *                           be served by an exact mapping where that is cheaper ("last_launch_fast").  bits: fast.
 *   "two_waves_per_simd"    1 (default) / 0: launches of the 2 / 4 / 8-lane kernels with more wavefronts than the device has
 *                           SIMDs take instantiations built for two wavefronts per SIMD (same operations, same bits).
 *   "packed_launch_order"   1 (default) / 0: a launch of more one-wave-per-SIMD workgroups than the device holds at once, of rows
 *                           that differ in length, takes its workgroups in a packed order (the SIMDs end together) instead of
 *                           longest first.
 *   "sort_by_length"        1 (default) / 0: batches uploaded afterwards fill the launch slots longest first (rows stay put).
 *   "assume_compute_units"  0 (default: the device's own) or a count to plan for: tests, callers that share a device.
 *   "scan_debug"            development builds only.
 * Read-only (grail_get_option): "compute_units"; "fast_arithmetic_served" (what "arithmetic" = 1 gets for the voice table as
 *   a whole: 1 / 2 / 0 exact kernels); of the last launch (its largest block): "last_launch_fast" (tier that ran, 0 exact),
 *   "last_launch_blocks", "last_launch_formants" (4 / 8), "last_launch_lanes", "last_launch_pipelined", "last_launch_chunks",
 *   "last_launch_packed" (blocks launched in packed order);
 *   statistics: "slow_division_wave_steps", "fast_wave_tiles" (tiles rendered without a slow sample), "general_wave_steps"
 *   (tolerance mode: slow samples; exact: general steps). */
int grail_set_option(grail_ctx *ctx, const char *name, int64_t value);
int grail_get_option(grail_ctx *ctx, const char *name, int64_t *value);
/* The planning behind "time_split", as pure host functions (no GPU, no context): what a caller needs to
 * predict or pin a fast-mode kernel family, and what the CPU tests check.
 * grail_time_split_warmup: the warm-up length of `voice` in samples, a multiple of 64 — after so many samples a
 *   filter state started from zero is within 2^-21 of the one Synthesize::next (src/lib.rs:530-575) would hold,
 *   from the slowest one-pole / band-pass decay over the voice's phonemes with a 5 % margin.  0: the voice does
 *   not qualify (a formant parameter outside (0, 0.5) / (0, 1), or more than 16384 samples).
 * grail_time_split_grid: bounds[0 .. chunks) of `chunks` chunks (2..64) over span_samples, multiples of 64 with
 *   bounds[0] = 0, spaced so that fast-forwarding (ff_cost_permille per sample, a rendered sample = 1000),
 *   warming up and rendering take every chunk's lane the same time.  GRAIL_ERR_INVALID_ARG when so many chunks
 *   do not fit (a chunk would render fewer than 64 samples). */
uint32_t grail_time_split_warmup(const grail_voice *voice);
/* An upper bound of an utterance's length in samples (pure host arithmetic): the lengths of its segments in seconds and the
 * sample rate of its voice.  Sequencer::next (src/lib.rs:859-888) adds every segment's length to an f32 clock and takes
 * 1 / sample_rate off it per sample, so a segment lasts length * sample_rate samples only up to the clock's rounding: a step
 * lowers it by at least dt - ulp(length) / 2, and the bound is the sum of length / (dt - ulp(length) / 2) + 2 over the
 * segments (a segment of 16 s at 192 kHz may last 22 % longer than its nominal length — in the reference too).  For sizing
 * out_stride without the device pre-pass (grail_batch_lengths gives the exact lengths); the time-split kernels use it to
 * skip the chunks an utterance does not reach.  UINT64_MAX: no bound (a length that is not finite, a clock that may not
 * move: dt <= ulp(length)). */
uint64_t grail_length_bound(const float *segment_lengths, uint32_t n_segments, float sample_rate);
/* The sharpness of a voice's resonances as the fast kernels see it: the predicted |fast - reference| in units of
 * 2^-23 of max(1, peak).  Per formant E_i = share_i * (0.0709 / bw_i) * (1 + (f_i / 0.075)^2) — share = the
 * formant's part of the phoneme's amplitudes, f and bw in cycles per sample, each the worst of the voice's
 * phonemes — and S = sqrt(sum_i E_i^2) (voices::generic(): 24, measured 13 - 20).  A rounding-level difference of a
 * filter coefficient (src/lib.rs:555-562) is amplified by the quality and the ring time of the band-pass, in the
 * reference's own arithmetic as well; the formula is a fit to measurements (profiles/r03_sharpness.txt).
 * The interpolating tier of fast arithmetic is served up to GRAIL_FAST_SHARPNESS_LIMIT; sharper voices (and
 * caller-built elems, judged the same way at upload: every two consecutive elems of an utterance like a voice of two
 * phonemes, the worst pair of the batch counts) get the tier that evaluates the reference's own coefficients, or the
 * exact kernels — read-only options "fast_arithmetic_served" / "last_launch_fast" tell.  A batch is judged by the
 * voices it names, not by the whole table.  +inf: a formant outside (0, 0.5) or a bandwidth <= 0. */
float grail_fast_sharpness(const grail_voice *voice);
int grail_time_split_grid(uint32_t span_samples, uint32_t warmup, uint32_t chunks, uint32_t ff_cost_permille,
                          uint32_t *bounds);
/* The launch plan, as a pure host function (no GPU, no context).  A kernel family fills the machine with a fixed
 * number of utterances (one wavefront per SIMD: 16 / 32 per compute unit for the pipelined workgroups, 256 / L for L
 * lanes per utterance), and one utterance more costs it a whole further round.  A batch is therefore cut into BLOCKS,
 * each rendered by the family that suits the block's size: whole rounds of the one-lane kernels first, the rest with
 * wider mappings (65537 utterances: 65536 on one lane each + 1 on a pipelined workgroup).  The
 * cut minimises a cost model calibrated on the device (profiles/r04_duration_sweep.txt) that follows the compute-unit
 * count and the utterances' length.  Exact arithmetic is mapping-invariant: the cut never changes a bit.  In fast
 * arithmetic a row's samples follow the family of ITS block, which this function predicts: rows keep batch order
 * (length-sorted batches: slot order), block i covers the next blocks[i].rows of them.
 *   compute_units: hipDeviceProp_t::multiProcessorCount (256 for a whole MI355X; option "compute_units" tells)
 *   arithmetic: 0 exact / 1 fast (voices the interpolating tier is served for) / 2 fast with the reference's own
 *     coefficients (sharper voices);  live_formants: 4 (formants 5-8 silent in every phoneme, as voices::generic()) or 8
 *   warmup: grail_time_split_warmup() of the voice table's slowest voice (0: no time-split kernels)
 *   rows, span_samples: the batch size and its longest utterance
 * *n_blocks receives the number of blocks even when it exceeds cap. */
typedef struct grail_plan_block {
    uint32_t rows;
    uint32_t lanes_per_utterance; /* lane kernels and pipelined workgroups: 1 / 2 / 4 / 8; scan kernel: 0 */
    uint32_t pipelined;           /* exact pipelined workgroups: 1 rounds of 16 samples, 2 rounds of 32; else 0 */
    uint32_t chunks;              /* time-split kernels: chunks per utterance; else 0 */
    uint32_t scan;                /* scan kernel: 1 two-stage, 2 three-stage workgroups; else 0 */
    uint32_t fast;                /* the block runs tolerance arithmetic */
    uint32_t formants;            /* formants laid out: 4 or 8 */
    float    model_ms;            /* the cost model's estimate for the block */
} grail_plan_block;
int grail_plan_blocks(uint32_t compute_units, int arithmetic, int live_formants, uint32_t warmup, uint32_t rows,
                      uint32_t span_samples, grail_plan_block *blocks, uint32_t cap, uint32_t *n_blocks);
/* ... of a batch whose utterances differ in length (option "ragged_plan"), rows in launch order = longest first:
 *   row_samples[rows]: the utterances' lengths in samples, descending
 *   row_segments[rows], row_kinks[rows]: their segments, and those among them with blend_length < length (the kink of
 *     alpha = min(clk / blend_length, 1), an event of its own in fast arithmetic); NULL: none
 * model_ms is then the estimate for the block's own rows. */
int grail_plan_ragged_blocks(uint32_t compute_units, int arithmetic, int live_formants, uint32_t warmup, uint32_t rows,
                             const uint32_t *row_samples, const uint32_t *row_segments, const uint32_t *row_kinks,
                             grail_plan_block *blocks, uint32_t cap, uint32_t *n_blocks);
/* The workgroup dispatcher as the library models it, and the packed launch order, as pure host functions (no GPU).  A launch
 * of workgroups that each hold their SIMDs alone (the one-wave-per-SIMD kernel families), n of them, more than the device
 * holds at once: workgroup b runs on XCC b mod 8; the k-th workgroup of an XCC goes to shader engine k mod 4 of it, whatever
 * the engines' load; it starts when that engine has room AND every earlier workgroup of the XCC has started (measured:
 * tools/dispatch_order.hip, profiles/r06_dispatch_order.txt — the model gives the makespan of recorded launches to the
 * microsecond).  waves_per_workgroup: 1 (a SIMD each: 32 per engine) or 4 (a compute unit each: 8 per engine);
 * compute_units that are not whole XCCs of 32: one pool.
 * grail_dispatch_model: the makespan of workgroups that take workgroup_ms[b], launched in `order` (order[position] =
 *   workgroup; NULL: 0, 1, 2 ...).
 * grail_packed_launch_order: the order option "packed_launch_order" launches them in — dealt to the pools by cost, each pool
 *   packed into its SIMDs (best-fit decreasing under the smallest capacity that fits), launched by planned start time — so
 *   that the SIMDs end together where "longest first" leaves them uneven (two or three workgroups per SIMD: up to 15 %). */
int grail_dispatch_model(uint32_t compute_units, uint32_t waves_per_workgroup, const double *workgroup_ms,
                         const uint32_t *order, uint32_t n, double *makespan_ms);
int grail_packed_launch_order(uint32_t compute_units, uint32_t waves_per_workgroup, const double *workgroup_ms, uint32_t n,
                              uint32_t *order);

/* ---- batches ----------------------------------------------------------- */
/* Uploads the inputs of n_utt utterances: utterance u is
 *   segs[seg_offsets[u] .. seg_offsets[u+1]).into_iter()
 *       .select(v).sequence(v).jitter(jitter_seeds[u], v).synthesize()
 * with v = voices[voice_ids[u]].  voice_ids == NULL means voice 0 for all,
 * jitter_seeds == NULL means seed 0 for all (examples/cli.rs:182). */
int grail_batch_upload(grail_ctx *ctx, const grail_phoneme_elem *segs,
                       const uint32_t *seg_offsets, const uint32_t *voice_ids,
                       const uint32_t *jitter_seeds, uint32_t n_utt, grail_batch **out);
/* Same, for callers that build SequenceElems themselves (skips the Selector):
 *   segs[..].into_iter().sequence(v).jitter(seed, v).synthesize() */
int grail_batch_upload_elems(grail_ctx *ctx, const grail_sequence_elem *segs,
                             const uint32_t *seg_offsets, const uint32_t *voice_ids,
                             const uint32_t *jitter_seeds, uint32_t n_utt,
                             grail_batch **out);
int grail_batch_free(grail_ctx *ctx, grail_batch *batch);
uint32_t grail_batch_size(const grail_batch *batch);

/* Sequencer clock pre-pass (src/lib.rs:861-888 only): the number of samples
 * each utterance yields, capped at max_len.  out_len is host memory [n_utt]. */
int grail_batch_lengths(grail_ctx *ctx, const grail_batch *batch, uint32_t max_len,
                        uint32_t *out_len);

/* Enqueue the fused Sequencer->Jitter->Synthesize kernel on ctx's stream.
 * out_dev: device memory, utterance u written at out_dev + u*out_stride,
 * samples past its end are left untouched.  out_len_dev: device memory [n_utt]
 * (samples written, <= out_stride) or NULL.  Returns without waiting.
 * Any out_stride >= the longest utterance works (grail_batch_lengths tells).  Rows that start 16-byte aligned — out_dev
 * from grail_device_alloc and out_stride a multiple of 4 samples — are written with 16-byte stores, and a multiple of 64
 * makes every 64-sample tile one aligned 256-byte run; an odd stride takes 4-byte stores in runs of 16 samples: + 2 - 4 %
 * (exact) / + 7 - 8 % (fast) on the headline batch (96006 instead of 96064 samples per row; with all eight formants
 * live on one lane the general flush: + 7 % / + 14 %). */
int grail_batch_synthesize_async(grail_ctx *ctx, const grail_batch *batch, float *out_dev,
                                 uint64_t out_stride, uint32_t *out_len_dev);
/* Wait for ctx's stream.  Returns GRAIL_ERR_BUFFER_TOO_SMALL if any utterance
 * of a kernel enqueued since the last sync was cut at out_stride. */
int grail_sync(grail_ctx *ctx);
/* HIP-event time (ms) of the most recent synthesis kernel on ctx's stream
 * (events recorded on the stream the kernel is launched on).  Syncs. */
int grail_last_kernel_ms(grail_ctx *ctx, float *ms);
/* Which kernel instantiation the most recent synthesis launch of ctx started, e.g.
 * "synth_kernel<L=1,T=32,W=1,1,NFA=4>" (profiling bookkeeping: counters measured on one
 * instantiation are never reported for another).  Owned by ctx; valid until its next launch. */
const char *grail_last_kernel_name(grail_ctx *ctx);

/* Resumable synthesis (the lazy-iterator use of the chain, examples/interactive.rs:31-48):
 * the per-utterance iterator state (Sequencer :839-854, Jitter :724-748, Synthesize :470-488)
 * lives in HBM between launches.  Each call renders the NEXT max_samples (<= out_stride)
 * samples of every utterance to out_dev + u*out_stride (from index 0) and the count to
 * out_len_dev[u] (0 once the utterance has ended).  Concatenating the chunks gives exactly the
 * one-shot result, whatever the chunk sizes ("arithmetic" = 0; in fast mode the chunks follow the
 * exact state to the bit — same lengths, same boundaries — and the samples agree with the one-shot
 * rendering within the fast-mode tolerance).  The batch must outlive the stream. */
int grail_stream_open(grail_ctx *ctx, const grail_batch *batch, grail_stream **out);
int grail_stream_next_async(grail_ctx *ctx, grail_stream *stream, uint32_t max_samples,
                            float *out_dev, uint64_t out_stride, uint32_t *out_len_dev);
/* The same chunk as i16 PCM (the WAV sink's conversion, examples/cli.rs:49, fused into the store):
 * what a sound-card callback wants.  f32 and i16 calls may be mixed on one stream. */
int grail_stream_next_pcm16_async(grail_ctx *ctx, grail_stream *stream, uint32_t max_samples,
                                  int16_t *out_dev, uint64_t out_stride, uint32_t *out_len_dev);
int grail_stream_close(grail_ctx *ctx, grail_stream *stream);
/* LIVE streams — the lazy source of examples/interactive.rs:31-48.  There ONE chain runs for the whole session and its
 * source never ends: text arrives while the audio callback is pulling samples, Sequencer::next fetches the next
 * SequenceElem only when a segment runs out (src/lib.rs:866-888), and carrier phase, noise seed, jitter and filter
 * state carry across everything that is ever said.  A live stream is that: it is opened EMPTY for n_utt utterances
 * (each one chain: voice_ids / jitter_seeds as in grail_batch_upload, NULL = voice 0 / seed 0), segments are appended
 * while it runs, and samples are pulled with grail_stream_next_async / _pcm16_async as from any stream.
 *   grail_stream_append: utterance u receives segs[seg_offsets[u] .. seg_offsets[u + 1]) behind what it already has
 *     (seg_offsets as in grail_batch_upload: n_utt + 1 non-decreasing entries; an empty range appends nothing).
 *   grail_stream_append_elems: the same for streams opened with caller_built_elems != 0 (SequenceElems, no Selector).
 *     Both return when the segments are queued on the context's stream, behind the launches before the call and ahead of
 *     those after it (the caller's arrays have been copied and may be reused at once); a failure of the queued work
 *     surfaces at the next grail_sync, like a kernel's.
 *   A Sequencer that needs a segment which has not been appended yet PAUSES: the call returns fewer than max_samples
 *     for that row (possibly 0) and the next call after an append carries on from exactly the same state — so the
 *     samples are those of the one-shot rendering of everything appended, bit for bit ("arithmetic" = 0), whatever
 *     the interleaving of appends and pulls.  (The reference starts by pulling TWO segments, :877-878: a fresh stream
 *     renders nothing until two are there or it is finished.)
 *   grail_stream_finish: the source of the utterances marked in `which` (n_utt bytes; NULL = all) has ended: what is
 *     pending is rendered, the last segment fades out (:906-912) and the row ends, as with a closed batch.
 *   grail_stream_pending: segments appended but not yet pulled by the Sequencer, per utterance (host memory [n_utt]);
 *     synchronises.  What an interactive front end needs to feed its chain just in time — the reference's source
 *     hands over ' ' (a Silence phoneme, src/lib.rs:1201 and the transcriber's no-rule case :1158-1163) whenever the
 *     Sequencer asks and no text is waiting.
 * Every ring holds ring_segments segments per utterance (a power of two, 4 .. 65536; 0 = 64): two the Sequencer is
 * working on and ring_segments - 2 pending; an append that does not fit fails with GRAIL_ERR_BUFFER_TOO_SMALL and
 * changes nothing.  Live streams run the general kernel instantiations (nothing is known about segments to come). */
int grail_stream_open_live(grail_ctx *ctx, uint32_t n_utt, const uint32_t *voice_ids, const uint32_t *jitter_seeds,
                           uint32_t ring_segments, int caller_built_elems, grail_stream **out);
int grail_stream_append(grail_ctx *ctx, grail_stream *stream, const grail_phoneme_elem *segs,
                        const uint32_t *seg_offsets);
int grail_stream_append_elems(grail_ctx *ctx, grail_stream *stream, const grail_sequence_elem *segs,
                              const uint32_t *seg_offsets);
int grail_stream_finish(grail_ctx *ctx, grail_stream *stream, const uint8_t *which);
int grail_stream_pending(grail_ctx *ctx, grail_stream *stream, uint32_t *pending);

/* One-call forms: upload, synthesize, copy back (GRAIL_OUT_HOST) or leave in
 * place (GRAIL_OUT_DEVICE), wait.  With GRAIL_OUT_HOST the rows are rendered in blocks of up to
 * 4096 utterances while the previous block travels over PCIe on a second stream, so the call
 * costs about max(kernel, copy) instead of their sum and needs two blocks of HBM, not the batch.  out_len is host memory [n_utt] or NULL.
 * GRAIL_OUT_HOST overwrites all n_utt*out_stride floats: each row is its
 * samples followed by zeros.  GRAIL_OUT_DEVICE leaves the tail untouched. */
int grail_synthesize_batch(grail_ctx *ctx, const grail_phoneme_elem *segs,
                           const uint32_t *seg_offsets, const uint32_t *voice_ids,
                           const uint32_t *jitter_seeds, uint32_t n_utt, float *out,
                           uint64_t out_stride, uint32_t *out_len, uint32_t flags);
int grail_synthesize_batch_elems(grail_ctx *ctx, const grail_sequence_elem *segs,
                                 const uint32_t *seg_offsets, const uint32_t *voice_ids,
                                 const uint32_t *jitter_seeds, uint32_t n_utt, float *out,
                                 uint64_t out_stride, uint32_t *out_len, uint32_t flags);

/* ---- text front half + PCM sink ------------------ */
/* TranscriptionRule  src/lib.rs:1030-1036; strings are Unicode scalar values (str::chars). */
typedef struct grail_rule {
    const uint32_t *string;
    uint32_t        string_len;
    const int32_t  *phonemes;   /* grail_phoneme */
    uint32_t        n_phonemes;
} grail_rule;
/* languages::generic()  src/languages/mod.rs:4-34; returns the rule count. */
uint32_t grail_language_generic(const grail_rule **rules, int *case_sensitive);
/* Transcriber::next until None  src/lib.rs:1116-1191.  leading_silence != 0 is
 * `.transcribe(language)` (buffer seeded with Silence, :1201); 0 starts with an empty buffer as
 * the reference's own unit tests do (:1212-1225).  *n_out is the full count even when > cap. */
int grail_transcribe(const uint32_t *text, uint32_t text_len, const grail_rule *rules,
                     uint32_t n_rules, int case_sensitive, int leading_silence,
                     int32_t *out_phonemes, uint32_t cap, uint32_t *n_out);
/* Intonator::next  src/lib.rs:1057-1075 (`.intonate(language, voice)` :1081). */
int grail_intonate(const grail_voice *voice, const int32_t *phonemes, uint32_t n,
                   grail_phoneme_elem *out);
/* text.chars().transcribe(languages::generic()).intonate(languages::generic(), voice)
 * — the front of examples/cli.rs:176-179.  out == NULL only counts. */
int grail_text_to_phoneme_elems(const grail_voice *voice, const char *text_utf8,
                                grail_phoneme_elem *out, uint32_t cap, uint32_t *n_out);
/* The whole chain of examples/cli.rs:175-184 for n texts: text i is spoken with
 * voices[voice_ids[i]] (NULL: voice 0) and jitter seed seeds[i] (NULL: 0, as the CLI). */
int grail_say_batch(grail_ctx *ctx, const char *const *texts_utf8, uint32_t n_texts,
                    const uint32_t *voice_ids, const uint32_t *jitter_seeds, float *out,
                    uint64_t out_stride, uint32_t *out_len, uint32_t flags);
/* `(x * i16::MAX as f32) as i16` of examples/cli.rs:49 on the device: rows of f32 -> rows of
 * i16, first len_dev[u] samples of each row; all pointers are device memory. Asynchronous. */
int grail_pcm16_async(grail_ctx *ctx, const float *in_dev, uint64_t in_stride,
                      const uint32_t *len_dev, uint32_t n_utt, uint32_t max_len,
                      int16_t *out_dev, uint64_t out_stride);
/* grail_batch_synthesize_async() with the examples/cli.rs:49 conversion fused into the kernel's
 * store: rows of i16 PCM in device memory, out_stride in samples; 2 B instead of 4 B of HBM
 * written per sample and no f32 copy anywhere.  out_dev 8-byte aligned and out_stride % 4 == 0
 * give vector stores (any other stride: 2-byte stores in runs of 16 samples, + 2 % on the headline batch). */
int grail_batch_synthesize_pcm16_async(grail_ctx *ctx, const grail_batch *batch, int16_t *out_dev,
                                       uint64_t out_stride, uint32_t *out_len_dev);
/* grail_synthesize_batch() with the conversion fused the same way: rows of i16 PCM (half the
 * PCIe bytes of the f32 form).  Same flags and row semantics. */
int grail_synthesize_batch_pcm16(grail_ctx *ctx, const grail_phoneme_elem *segs,
                                 const uint32_t *seg_offsets, const uint32_t *voice_ids,
                                 const uint32_t *jitter_seeds, uint32_t n_utt, int16_t *out,
                                 uint64_t out_stride, uint32_t *out_len, uint32_t flags);
/* Per-row digest of rendered rows, computed on the device (comparing a 25 GB batch over PCIe is
 * pointless): sums[u] = sum of the samples' IEEE bit patterns mod 2^64, maxabs[u] = largest
 * finite |x|, nonfinite[u] = count of NaN/Inf, over the first len_dev[u] samples of row u.
 * in_dev/len_dev are device memory, the three results host memory [n_utt].  Synchronous. */
int grail_batch_digest(grail_ctx *ctx, const float *in_dev, uint64_t in_stride,
                       const uint32_t *len_dev, uint32_t n_utt, uint64_t *sums, float *maxabs,
                       uint32_t *nonfinite);
/* Per-row distance between two renderings of one batch, computed on the device (fast mode against
 * exact mode at full size): maxdiff[u] = max |a - b|, sumsq[u] = sum (a - b)^2 over the first
 * len_a_dev[u] samples, mismatches[u] = samples where exactly one side is non-finite, plus 1 if
 * the two lengths differ.  a/b/len_* are device memory, the three results host memory [n_utt]. */
int grail_batch_compare(grail_ctx *ctx, const float *a_dev, const float *b_dev, uint64_t stride,
                        const uint32_t *len_a_dev, const uint32_t *len_b_dev, uint32_t n_utt,
                        float *maxdiff, double *sumsq, uint32_t *mismatches);
/* save_wav  examples/cli.rs:28-67: 44-byte RIFF header (PCM, mono, 16 bit) + samples. */
int grail_wav_write_i16(const char *path, const int16_t *pcm, uint32_t n, uint32_t sample_rate);

/* ---- device memory plumbing ------------------------------------------- */
int grail_device_alloc(grail_ctx *ctx, size_t bytes, void **out);
int grail_device_free(grail_ctx *ctx, void *ptr);
/* Pinned (page-locked) host memory.  A GRAIL_OUT_HOST destination that lives in it receives the
 * device-to-host copies directly, at PCIe rate; a pageable destination is fed through pinned
 * staging buffers and copier threads (still overlapped with the kernels, a little slower). */
int grail_host_alloc(grail_ctx *ctx, size_t bytes, void **out);
int grail_host_free(grail_ctx *ctx, void *ptr);
int grail_memcpy_d2h(grail_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int grail_memcpy_h2d(grail_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int grail_memset_d(grail_ctx *ctx, void *dst_dev, int value, size_t bytes);

/* ---- multi-GPU: utterance sharding + voice-table broadcast ------------- */
/* Contiguous utterance range of `rank` out of `world`:
 * [rank*n/world, (rank+1)*n/world) in 64-bit arithmetic. */
void grail_shard_range(uint64_t n_utt, uint32_t rank, uint32_t world, uint64_t *begin,
                       uint64_t *end);
/* RCCL path (one process per GPU).  grail_comm_unique_id fills a 128-byte id
 * on rank 0 that the launcher hands to every rank out of band;
 * grail_comm_init joins the communicator; grail_broadcast_voices sends rank
 * `root`'s voice table (grail_set_voices) to every rank's HBM with one
 * ncclBroadcast on ctx's stream and installs it there. */
#define GRAIL_UNIQUE_ID_BYTES 128
int grail_comm_unique_id(uint8_t id[GRAIL_UNIQUE_ID_BYTES]);
int grail_comm_init(grail_ctx *ctx, const uint8_t id[GRAIL_UNIQUE_ID_BYTES], uint32_t rank,
                    uint32_t world);
int grail_broadcast_voices(grail_ctx *ctx, uint32_t n_voices, uint32_t root);
/* What RCCL itself reports for ctx's communicator: *ranks = ncclCommCount (0 when no communicator
 * has been formed), *rank = ncclCommUserRank.  Lets a launcher prove that N ranks really met. */
int grail_comm_info(grail_ctx *ctx, uint32_t *ranks, uint32_t *rank);
int grail_comm_destroy(grail_ctx *ctx);

/* ---- one call, the whole node (SURVEY.md §8e "one process, 8 devices") ---------------------------------------------
 * The reference host makes ONE call for its whole job (examples/cli.rs:175-184: one chain, collected) and the chain's
 * state is per utterance (src/lib.rs:470-488, 724-748, 839-854), so a batch shards over the GPUs of a node with no
 * exchange step.  A grail_node is that for a host that stays one process (a Rust binary, say): one grail_ctx and one
 * host thread per device; a call cuts the batch into the contiguous ranges of grail_shard_range, renders the shards
 * concurrently and lands every shard in its slice of ONE host buffer.  Utterance u's samples are the same bits as from
 * grail_synthesize_batch on a single context ("arithmetic" = 0; in fast arithmetic they follow the kernel family of the
 * shard's size, as with any batch size — pin "lanes_per_utterance" for batch-invariant fast bits).
 * A node is used from one thread at a time.  Failures: the first failing device's status and message
 * ("device[i] = d: ..."); GRAIL_ERR_BUFFER_TOO_SMALL when some shard reported it and none failed harder. */
/* devices[n_devices]: HIP device ordinals, one shard each (NULL: 0 .. n_devices - 1).  A device may be named more than
 * once (several contexts on it — what the tests on a one-GPU box do); RCCL then cannot form the communicator, see
 * grail_node_set_voices. */
int grail_node_create(const int *devices, uint32_t n_devices, grail_node **out);
int grail_node_destroy(grail_node *node);
uint32_t grail_node_size(const grail_node *node);
/* The context of device slot `index` (borrowed; owned by the node): per-device queries — grail_get_option,
 * grail_device_pci_bus_id, grail_comm_info, grail_get_voices — between node calls. */
int grail_node_context(grail_node *node, uint32_t index, grail_ctx **ctx);
/* Installs the voice table on every device: device slot 0 receives it (grail_set_voices) and ONE ncclBroadcast over
 * xGMI carries it to the HBM of the others, over a communicator formed inside the process (ncclCommInitAll over the
 * node's devices, on the first call) — the collective of §8e.  RCCL refuses a communicator that names a GPU twice: a
 * node created with duplicate devices fails here with GRAIL_ERR_RCCL unless option "node_voices_without_rccl" = 1 was
 * set, which installs the table with one grail_set_voices per context instead (for tests on a one-GPU box; never
 * chosen silently). */
int grail_node_set_voices(grail_node *node, const grail_voice *voices, uint32_t n_voices);
/* "node_voices_without_rccl" (0 default / 1) belongs to the node; every other name is grail_set_option on each of
 * its contexts.  grail_node_get_option: "node_devices", "node_voices_without_rccl", "node_rccl_ranks" (the smallest
 * ncclCommCount over the contexts, 0: no communicator); any other name is read from device slot 0. */
int grail_node_set_option(grail_node *node, const char *name, int64_t value);
int grail_node_get_option(grail_node *node, const char *name, int64_t *value);
/* The shard of device slot `index` out of n_devices, as a pure host function (no GPU): rows
 * [first_row, first_row + rows) = grail_shard_range(n_utt, index, n_devices), their segments
 * segs[first_seg .. first_seg + n_segs), and — rebased_offsets != NULL, cap >= rows + 1 — the rows' seg_offsets
 * relative to first_seg.  The node calls below hand exactly this view to grail_synthesize_batch*(): segs + first_seg,
 * rebased_offsets, voice_ids + first_row, jitter_seeds + first_row, out + first_row * out_stride, out_len + first_row. */
typedef struct grail_node_shard {
    uint64_t first_row;
    uint64_t rows;
    uint32_t first_seg;
    uint32_t n_segs;
} grail_node_shard;
int grail_node_shard_of(const uint32_t *seg_offsets, uint64_t n_utt, uint32_t index, uint32_t n_devices,
                        grail_node_shard *shard, uint32_t *rebased_offsets, uint64_t cap);
/* grail_synthesize_batch / _elems / _pcm16 / grail_say_batch over the node: same arguments and row semantics; `out`
 * and `out_len` are HOST memory (flags must not hold GRAIL_OUT_DEVICE: there is no one device to leave the rows on).
 * Every shard goes through its context's overlapped device-to-host pipeline into its slice of `out`; pinned memory
 * from grail_node_host_alloc receives the copies directly. */
int grail_node_synthesize_batch(grail_node *node, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                                const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt, float *out,
                                uint64_t out_stride, uint32_t *out_len, uint32_t flags);
int grail_node_synthesize_batch_elems(grail_node *node, const grail_sequence_elem *segs, const uint32_t *seg_offsets,
                                      const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt,
                                      float *out, uint64_t out_stride, uint32_t *out_len, uint32_t flags);
int grail_node_synthesize_batch_pcm16(grail_node *node, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                                      const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt,
                                      int16_t *out, uint64_t out_stride, uint32_t *out_len, uint32_t flags);
/* ... with the rows LEFT IN HBM (what a downstream GPU consumer wants, and what the headline metric measures): slot i's shard is
 * rendered into out_dev[i] — device memory of slot i's GPU holding its rows x out_stride floats (grail_device_alloc on
 * grail_node_context(node, i); rows from grail_node_shard_of), row r of the shard at out_dev[i] + r * out_stride; NULL for a slot
 * whose shard is empty.  out_len: host memory [n_utt] or NULL.  No copy is made; the call returns when every slot has finished. */
int grail_node_synthesize_batch_device(grail_node *node, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                                       const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt,
                                       float *const *out_dev, uint64_t out_stride, uint32_t *out_len);
int grail_node_say_batch(grail_node *node, const char *const *texts_utf8, uint32_t n_texts, const uint32_t *voice_ids,
                         const uint32_t *jitter_seeds, float *out, uint64_t out_stride, uint32_t *out_len,
                         uint32_t flags);
/* grail_batch_lengths over the node (the Sequencer clock pre-pass, src/lib.rs:861-888): what a caller sizes out_stride
 * with.  out_len is host memory [n_utt]. */
int grail_node_lengths(grail_node *node, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                       const uint32_t *voice_ids, uint32_t n_utt, uint32_t max_len, uint32_t *out_len);
/* Wall-clock milliseconds each device slot spent in the last node call (host memory [grail_node_size]; 0 for a slot
 * whose shard was empty): how evenly the shards loaded the node. */
int grail_node_last_shard_ms(grail_node *node, float *ms, uint32_t cap);
/* Pinned host memory every device of the node can copy into (hipHostMallocPortable). */
int grail_node_host_alloc(grail_node *node, size_t bytes, void **out);
int grail_node_host_free(grail_node *node, void *ptr);

#ifdef __cplusplus
}
#endif
#endif /* GRAIL_HIP_H */
