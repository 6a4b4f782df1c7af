// launch_plan.cpp — which kernel family renders a block of rows, what a block costs, how a batch is cut into blocks.
// Pure host arithmetic over the context's options and the batch's summary (no HIP call): grail_plan_blocks runs it
// without a device.
#include "api_internal.hpp"

#include <functional>
#include <queue>

using namespace grail;
using namespace grail::host;

namespace grail {
namespace host {

int auto_lanes_per_utt(uint32_t n_utt, uint64_t simds)
{
    // Measured (profiles/r01_lanes_sweep.txt): a wave alone on its SIMD renders 2 s of audio
    // in 92 / 60 / 42 / 38 ms for L = 1 / 2 / 4 / 8, and a second wave on the same SIMD costs
    // more than it brings (the packed-f32 stream of one wave already keeps the VALU ~80 %
    // busy).  So: the widest mapping that still fits one wave per SIMD (4 per compute unit: 1024 on a
    // whole MI355X).
    for (int L = 8; L > 1; L /= 2)
        if (((uint64_t)n_utt * L + 63) / 64 <= simds) return L;
    return 1;
}

// the longest utterance of the batch in samples, as far as the host knows it (the f32 clock adds a few per segment)
double batch_span(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride)
{
    double span = std::ceil((double)batch->max_seconds * ctx->max_rate) + 64.0;
    if (!(span >= 64.0)) span = 64.0;                         // NaN / negative lengths
    return std::fmin(span, (double)(out_stride ? out_stride : 1));
}

// Cost model of the planner, in milliseconds per SAMPLE OF THE LONGEST UTTERANCE for one round of a family (a round:
// as many rows as give every SIMD one wave).  Calibrated on 2 s utterances at 48 kHz, one MI355X
// (profiles/r03_small_batch.txt, profiles/r04_duration_sweep.txt); only the ratios matter.  Indexed [L = 1, 2, 4, 8].
constexpr double MID_MS_4 = 32.6, MID_MS_8 = 57.2;     // 65 536 x 2 s, the MID kernels (profiles/r04_middle_tier.txt)
static double lane_ms_per_sample(bool fast, bool live4, int L)
{
    static const double exact4[4] = {40.6, 26.9, 16.3, 18.2}, exact8[4] = {77.1, 43.5, 25.8, 15.7};
    static const double fast4[4] = {15.5, 13.6, 12.8, 12.1}, fast8[4] = {23.6, 16.4, 13.3, 12.0};
    const int i = L == 1 ? 0 : L == 2 ? 1 : L == 4 ? 2 : 3;
    return (fast ? (live4 ? fast4 : fast8) : (live4 ? exact4 : exact8))[i] / 96006.0;
}
// ... of the second tolerance tier (MID, one lane per utterance)
static double mid_ms_per_sample(bool live4) { return (live4 ? MID_MS_4 : MID_MS_8) / 96006.0; }

// Lane kernels that keep their state in 256 registers — tolerance mode on two (four live formants), four and eight lanes per
// utterance, exact on two (four live formants) and four — can share a SIMD with a second wave, and a launch of more waves
// than the device has SIMDs takes the instantiations built for that (tolerance mode: the lone wave leaves the VALU idle a
// quarter of the time; 65 536 aligned utterances on two lanes each 21.5 ms instead of 26.9, on four 38.6 instead of 51.0.
// Exact: 49.0 instead of 53.7 and 55.1 instead of 64.6; a speech-like corpus 58.1 instead of 67.0: profiles/r05_two_waves.txt).
// Launches that fit one wave per SIMD keep the one-wave instantiations — which cannot share a SIMD, so where the dispatcher
// puts their waves cannot matter.
bool family_cohabits(const grail_ctx *ctx, const Family &f, uint32_t rows)
{
    if (!ctx->two_waves_option || f.scan || f.pipe || f.split_k || f.fast > 1u) return false;
    // (tolerance mode: 2 lanes with four formants laid out, 4 and 8 lanes; exact: 2 lanes with four formants, 4 lanes —
    // the instantiations that hold their state in 256 registers without a scratch segment)
    const bool built = f.fast ? (f.L == 4 || f.L == 8 || (f.L == 2 && f.live4)) : (f.L == 4 || (f.L == 2 && f.live4));
    return built && ((uint64_t)rows * (uint64_t)f.L + 63u) / 64u > ctx_simds(ctx);
}
// What two waves on a SIMD take together, over twice the lone wave's time.  Tolerance mode: 0.76 - 0.80 on aligned batches
// (the plain pairs of one wave fill three quarters of the issue slots), 0.50 - 0.68 where events are dense — a slow sample is
// latency (the elems' loads behind a segment advance, branches, end-point evaluations under one lane's predicate), which the
// other wave fills.  Exact: the lone wave issues at 94 % of the best rate this SIMD has shown, yet two
// resident waves take 0.85 - 0.91 of two in turn on aligned batches — tile heads, flushes and the general steps of the segment
// boundaries are latency too — and 0.70 on phonemes of 4 - 16 ms (profiles/r05_two_waves.txt).
// `density`: events per lane and sample, as in ragged_wave_ms.  Fitted with tools/ragged_fit.py (profiles/r05_ragged_fit.txt).
static double cohabit_gain(const Family &f, double density = 0.0)
{
    if (!f.fast) {
        // (two lanes: 0.905 aligned, 0.873 on the speech-like corpus, 0.85 with phonemes of 4 - 16 ms; four: 0.85 / 0.83 / 0.70)
        const double aligned = f.L == 2 ? 0.91 : 0.85, dense = f.L == 2 ? 0.86 : 0.70;
        return aligned - (aligned - dense) * std::fmin(1.0, density / (f.L == 2 ? 6.0e-4 : 3.0e-3));
    }
    const double aligned = f.L == 2 ? 0.80 : 0.76, dense = f.L == 2 ? 0.52 : f.L == 4 ? 0.65 : 0.68;
    return aligned - (aligned - dense) * std::fmin(1.0, density / 3.0e-4);
}

// Pipelined workgroups on rows that differ in length: a tile with an event of ONE of a workgroup's utterances costs the
// workgroup several calm tiles, so a small batch is spread thinly — as few utterances per workgroup as give every compute
// unit two workgroups — instead of filling 16 (8) slots of a few workgroups and leaving the other units idle: 256 speech-like
// utterances, one per workgroup, hold no event but their own (profiles/r05_mixed_runs.txt).
uint32_t pipe_fill_for(const grail_ctx *ctx, const grail_batch *batch, const Family &f, uint32_t rows)
{
    if (!f.pipe || !ctx->pipe_spread || batch == nullptr || batch->granule_samples.size() < 2 ||
        batch->granule_samples.front() == batch->granule_samples.back())
        return 0u;
    const uint32_t slots = f.live4 ? 16u : 8u;
    // (two workgroups to a compute unit — what the rounds of 16 samples leave room for: they fill each other's stalls.
    // 4 096 speech-like utterances, eight to a workgroup: 16.8 ms; sixteen to a workgroup, one per unit: 19.4)
    const uint32_t units = 2u * (uint32_t)ctx->cus;
    const uint32_t fill = (rows + units - 1u) / units;
    return fill >= slots ? 0u : (fill < 1u ? 1u : fill);
}

// what launching `rows` rows with family f costs (model milliseconds)
double family_cost(const grail_ctx *ctx, const Family &f, uint32_t rows, double span)
{
    const double cus = (double)ctx->cus, lanes = (double)ctx_lanes(ctx);
    if (f.scan) {
        // one workgroup per utterance; g = workgroups per compute unit.  Three-stage flavour: the latency of one
        // utterance's chain up to ~2 per CU, then ~0.53 ms per workgroup and CU (2 s); two-stage: 0.41 (four live
        // formants) / 0.65 (eight)
        const double g = std::ceil((double)rows / cus);
        const double ms2s = f.scan_pipe ? (f.live4 ? std::fmax(1.14, 0.53 * g) : std::fmax(1.67, 0.75 * g))
                                        : (f.live4 ? 0.41 * g + 0.1 : 0.65 * g + 0.2);
        return ms2s * span / 96006.0;
    }
    if (f.split_k) {
        // every lane takes as long as the first chunk's, which renders split_bounds[1] samples and nothing else
        // (a wave holds 64 utterances at ONE chunk index: ceil(rows / 64) waves per chunk, whatever rows modulo 64 is)
        const double rounds = f.split_active ? std::ceil((double)f.split_active / (double)ctx_simds(ctx))
                                             : std::ceil(std::ceil((double)rows / 64.0) * f.split_k / (double)ctx_simds(ctx));
        // (+ 0.12 ms: what a launch of chunk lanes costs before any of them renders — short utterances see it)
        if (f.fast == 2u) return rounds * ((double)f.split_bounds[1] * mid_ms_per_sample(f.live4 != 0) + 0.12);
        return rounds * ((double)f.split_bounds[1] * (f.live4 ? 15.7 : 23.3) / 96006.0 + 0.12);
    }
    if (f.pipe) {
        const double groups = std::ceil((double)rows / (f.live4 ? 16.0 : 8.0));
        const double per_cu = std::ceil(groups / cus);
        // rounds of 32: one workgroup per CU; rounds of 16: two per CU are resident together, further ones queue
        // (rounds of 16 with one workgroup per CU: 7.9 ms for config 2 where rounds of 32 take 6.5)
        const double ms2s = f.pipe == 2 ? (f.live4 ? 6.5 : 7.3) * per_cu
                                        : per_cu <= 1.0 ? (f.live4 ? 7.9 : 8.8) : 11.2 * std::ceil(per_cu / 2.0);
        return ms2s * span / 96006.0;
    }
    const double rounds = std::ceil((double)rows * f.L / lanes);
    if (f.fast == 2u) return rounds * span * mid_ms_per_sample(f.live4 != 0);
    // (eight formants laid out, the upper four silent: the half-live loops of the one-lane kernel, 45.7 ms where all
    // eight live take 77.1)
    if (f.half) return rounds * span * 45.7 / 96006.0;
    if (family_cohabits(ctx, f, rows)) {
        // (pairs of waves: an odd wave-round at the end runs alone, at the lone wave's rate)
        const double pairs = std::floor(rounds / 2.0), rest = rounds - 2.0 * pairs;
        return (2.0 * pairs * cohabit_gain(f) + rest) * span * lane_ms_per_sample(f.fast != 0, f.live4 != 0, f.L);
    }
    return rounds * span * lane_ms_per_sample(f.fast != 0, f.live4 != 0, f.L);
}

// A phoneme batch is judged by the voices IT names (batch->used_voices), not by the whole table: a preset with eight
// live formants somewhere in the table does not take the four-formant kernels away from batches that never use it.
// (A context without per-voice records — grail_plan_blocks — goes by the table-wide flag.)
template <typename Pred>
static bool used_voices_all(const grail_ctx *ctx, const grail_batch *batch, bool table_wide, Pred pred)
{
    if (ctx->voice_info.empty() || batch->used_voices.empty()) return table_wide;
    for (const uint32_t v : batch->used_voices)
        if (v >= ctx->voice_info.size() || !pred(ctx->voice_info[v])) return false;
    return true;
}
bool batch_half_capable(const grail_ctx *ctx, const grail_batch *batch)
{
    // (caller-built elems: judged at upload over the batch's distinct elems, against the voice table of that moment)
    if (!batch->phoneme_mode)
        return ctx->skip_silent_option && batch->elems_live4_ok && batch->elems_warmup_epoch == ctx->voices_epoch;
    return ctx->skip_silent_option &&
           used_voices_all(ctx, batch, ctx->voices_upper_silent, [](const grail_ctx::VoiceInfo &v) { return v.upper_silent; });
}

// formants 5-8 left out altogether: the table qualifies (live4_ok); every segment is at least
// two samples long, so the Sequencer clock never goes negative and alpha stays in [0,1]; and
// every pitch stays >= 2^-20 under the pitch jitter, so the polyBLEP quotient and with it the
// saw every formant is fed from stay finite (a dead formant fed +-inf would emit NaN)
bool batch_live4_any_blend(const grail_ctx *ctx, const grail_batch *batch)
{
    return batch_half_capable(ctx, batch) &&
           (!batch->phoneme_mode ||
            used_voices_all(ctx, batch, ctx->voices_live4_ok, [](const grail_ctx::VoiceInfo &v) { return v.live4_ok; })) &&
           batch->plain &&
           batch->min_length >= 2.0f * ctx->max_dt &&
           batch->min_pitch * 0.999f - 1.002f * ctx->max_pitch_jitter >= 9.5367431640625e-07f;
}
// ... and (the lane kernels' four-formant instantiations) every blend length a power of two
bool batch_live4(const grail_ctx *ctx, const grail_batch *batch)
{
    return batch_live4_any_blend(ctx, batch) && !batch->any_blend;
}

// The family a block of `fam` rows of this batch takes.  Exact arithmetic: the widest mapping that still gives every
// SIMD at most one wave (pipelined workgroups, then 8 / 4 / 2 / 1 lanes per utterance).  Fast arithmetic: the cheapest
// of the scan kernel, the time-split kernels and the fast lane kernels by the cost model above (which follows the
// utterances' length: a time-split pays a warm-up per chunk, the scan kernel the latency of one utterance's chain),
// unless an option pins the choice.
void choose_family(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride, uint32_t fam, Family &f,
                   bool exact_only, int pin_lanes)
{
    const uint64_t simds = ctx_simds(ctx), cus = (uint64_t)ctx->cus;
    const int lanes_option = pin_lanes ? pin_lanes : ctx->lanes_option;   // (pin_lanes: as if the option named it)
    f = Family();
    // (the lane kernels and the pipelined workgroups have four-formant instantiations for every blend length; the lean
    // stream kernels for power-of-two blend lengths only: batch_live4)
    f.live4 = batch_live4_any_blend(ctx, batch) ? 1u : 0u;
    // fast arithmetic is served up to a sharpness of the resonances (elems_sharpness); beyond it the exact kernels run
    // ... in the tier the sharpness allows: 1 = coefficients interpolated, 2 = the reference's own coefficients (MID)
    f.fast = exact_only ? 0u : (uint32_t)fast_tier(ctx, batch);
    // (MID kernels exist one-shot with one lane per utterance, and time-split: a pinned wider mapping gets the exact kernels)
    if (f.fast == 2u && lanes_option > 1) f.fast = 0u;
    int L = lanes_option ? lanes_option : auto_lanes_per_utt(fam, simds);
    if (f.fast == 2u) L = 1;          // (before the four-formant layout is decided: eight lanes would give it up)
    // small batches leave SIMDs idle: four-wave workgroups (one wave renders 16 utterances, one carries
    // the per-utterance chain, two prepare the filter coefficients), up to two per CU (tools/pipe4_range.py:
    // 11.5 ms up to 4 096 utterances, 15.7 up to 8 192 where the lane kernels take 18.0; three per CU lose)
    const bool want_pipe4 = batch_live4_any_blend(ctx, batch) && !lanes_option && ctx->pipeline_option &&
                            (int64_t)(((uint64_t)fam + 15) / 16) <= pipe4_groups(ctx);
    const bool want_pipe8 = !batch_live4_any_blend(ctx, batch) && !lanes_option && ctx->pipeline_option &&
                            (int64_t)(((uint64_t)fam + 7) / 8) <= pipe8_groups(ctx);
    // one workgroup per CU suffices: rounds of 32 samples instead of 16 (pipe = 2) — for batches whose rows are aligned.
    // Rows that differ in length (the upload kept their summary) have their events at times of their own, and a tile with an
    // event is rendered in whole rounds where nobody has one (synth_kernel.h pipe_rounds): rounds of 16 fit between two events
    // far more often than rounds of 32 (256 speech-like utterances 18.8 ms against 20.0, with phonemes of 16 - 64 ms 10.0 / 11.2,
    // 4 - 16 ms 4.6 / 5.3; profiles/r05_mixed_runs.txt)
    const bool rows_differ = batch != nullptr && !batch->granule_samples.empty() &&
                             batch->granule_samples.front() != batch->granule_samples.back();
    const bool round32 = ctx->pipe_round32 == 2 || (ctx->pipe_round32 == 1 && !rows_differ);      // (2: tests, A/B)
    const uint32_t pipe4_kind = round32 && ((uint64_t)fam + 15) / 16 <= cus ? 2u : 1u;
    const uint32_t pipe8_kind = round32 && ((uint64_t)fam + 7) / 8 <= cus ? 2u : 1u;
    if (want_pipe4 && !f.fast) {
        f.pipe = pipe4_kind;
        L = 4;
    } else if (want_pipe8 && !f.fast) {
        f.pipe = pipe8_kind;                              // eight formants: 8 utterances per workgroup
        L = 8;
    }
    // eight lanes per utterance need eight formants to lay out; for batches that small the
    // 8-lane kernel is also the fastest (18.2 against 18.8 ms: half the rows to flush per wave)
    if (f.live4 && !f.pipe && L == 8) f.live4 = 0u;
    if (f.live4 && !f.pipe && !lanes_option) {
        // same rule as auto_lanes_per_utt — the widest mapping with one wave per SIMD — over 4 formants
        L = ((uint64_t)fam * 4 + 63) / 64 <= simds ? 4 : ((uint64_t)fam * 2 + 63) / 64 <= simds ? 2 : 1;
    }
    // voices whose upper formants are never audible but that do not qualify for the 4-formant
    // kernels: one lane per utterance runs the half-live loop and ties two lanes per utterance,
    // whose second lane would only hold silent formants
    if (!lanes_option && !f.live4 && L == 2 && batch_half_capable(ctx, batch)) L = 1;
    f.L = L;
    f.half = !f.fast && !f.live4 && !f.pipe && L == 1 && batch_half_capable(ctx, batch);
    if (!f.fast) return;

    const double span = batch_span(ctx, batch, out_stride);
    const bool l4ab = batch_live4_any_blend(ctx, batch);
    // fast arithmetic, mid-size batches: one lane per utterance would leave most of the machine idle, so the time
    // axis of every utterance is cut into chunks with a lane each (synth_kernel<..., SPLIT>): as many chunks as
    // fill the machine, laid out over the batch's longest utterance so that all lanes finish together
    Family split = f;
    // (which voices / elems qualify: voice_warmup.  A phoneme batch is covered by its voice table; caller-built elems
    // by the warm-up computed over them at upload, against the voice table of that moment)
    const bool split_ok = batch->phoneme_mode ? used_voices_all(ctx, batch, ctx->voices_split_ok,
                                                                [](const grail_ctx::VoiceInfo &v) { return v.split_ok; })
                                              : (batch->elems_warmup != 0u && batch->elems_warmup_epoch == ctx->voices_epoch);
    // (the grid is laid out for the longest warm-up of the TABLE, not of the voices the batch names: for a pinned grid an
    // utterance's samples may not depend on what else is in the batch; each lane still warms up for its own voice's length)
    const uint32_t warmup = batch->phoneme_mode ? ctx->max_warmup : batch->elems_warmup;
    if (ctx->split_option && !lanes_option && split_ok && batch->plain &&
        out_stride <= 0xFFFFFFFFull && (ctx->split_chunks >= 2 || ctx->split_chunks == 0)) {
        const double sp = ctx->split_span ? std::fmin((double)ctx->split_span, (double)out_stride) : span;
        // as many chunks as give every SIMD one wave: ceil(fam / 64) waves per chunk index (5 000 utterances are 79 waves
        // per chunk: 12 chunks, not 65 536 / 5 000 = 13, which would be 1 027 waves and a second round for three of them)
        int K = ctx->split_chunks ? (int)ctx->split_chunks
                                  : (int)std::min<uint64_t>(simds / (((uint64_t)fam + 63u) / 64u), SPLIT_MAX_CHUNKS);
        // Rows that differ in length (whole batch, launched longest first, the upload's length bounds on the device): a
        // chunk's wave whose utterances all end before the chunk begins is gone at once (synth_kernel.h, SPLIT), so the
        // grid may hold more (wave, chunk) pairs than the device has SIMDs (a speech-like corpus keeps a third of them: the
        // grid's chunks are shortest at the far end, where only the longest rows still are)
        auto active_pairs = [&](const uint32_t *bounds, const int k) {
            uint64_t pairs = 0;
            for (size_t g = 0; g < batch->granule_samples.size(); g += 8) {      // 64 launch slots: the first is the longest
                const double len = (double)batch->granule_samples[g] * 1.02 + 64.0;
                int c = 1;
                while (c < k && (double)bounds[c] < len) ++c;
                pairs += (uint64_t)c;
            }
            return pairs;
        };
        const bool by_length = !ctx->split_chunks && !ctx->split_span && rows_differ && fam == batch->n_utt &&
                               batch->len_bound_known && batch->len_bound_epoch == ctx->voices_epoch;
        K = (int)std::fmin((double)K, sp / 512.0);
        // (a fast-forwarded sample costs the same whatever is rendered afterwards; a rendered sample of eight live
        // formants costs 1.5 x one of four; 0.8 from a sweep, profiles/r03_small_batch.txt)
        const double ff_cost = 1e-3 * (double)ctx->split_ff_permille * (l4ab ? 1.0 : 0.8) * (f.fast == 2u ? 0.6 : 1.0);
        // the largest K <= K whose chunks fit (a chunk must render at least a tile): fitting is monotone in K
        if (K >= 2 && !split_grid((uint32_t)sp, warmup, K, ff_cost, split.split_bounds)) {
            int lo = 1, hi = K;                  // lo fits (or is 1), hi does not
            while (hi - lo > 1) {
                const int mid = (lo + hi) / 2;
                if (split_grid((uint32_t)sp, warmup, mid, ff_cost, split.split_bounds)) lo = mid;
                else hi = mid;
            }
            K = lo;
            if (K >= 2) (void)split_grid((uint32_t)sp, warmup, K, ff_cost, split.split_bounds);
        }
        if (K >= 2 && K <= 4 && by_length) {
            // (Batches of 16 384 utterances and more, whose grids are coarse — 4 chunks, 2 — and leave SIMDs without a wave:
            // up to 2 K - 1 chunks while the active pairs fit the device.  The count is an estimate from the upload's summary
            // — measured 6 % above what the kernel finds — and finer grids of smaller batches gain nothing: a chunk's lane
            // fast-forwards through everything before it, and on a speech-like corpus that costs more per sample than the
            // grid is laid out for: 4 096 utterances 15.4 ms on 16 chunks, 14.9 on 24, 19.6 on 34.)
            const int most = 2 * K - 1;
            for (int k = K + 1; k <= most && (double)k <= sp / 512.0; ++k) {
                uint32_t wider[SPLIT_MAX_CHUNKS + 1] = {};
                if (!split_grid((uint32_t)sp, warmup, k, ff_cost, wider) || active_pairs(wider, k) * 16 > simds * 17) break;
                K = k;
                std::memcpy(split.split_bounds, wider, sizeof wider);
            }
        }
        if (K >= 2) {
            split.split_k = K;
            split.split_active = by_length ? (uint32_t)active_pairs(split.split_bounds, K) : 0u;
            split.split_bounds[K] = (uint32_t)out_stride;
            split.live4 = l4ab ? 1u : 0u;
            split.pipe = 0u;
            split.L = 1;
        }
    }
    // fast arithmetic, few utterances: one workgroup per utterance with the time axis across the lanes and the
    // filter recurrences solved by parallel scans (scan_kernels.hip).  Needs every parameter inside the safe window
    // (no IEEE fallback).
    Family scan = f;
    // (rows that differ in length, option "ragged_plan": up to 64 workgroups per compute unit — there the cost model decides,
    // by the rows: 10 000 speech-like utterances 15.3 ms against 19.1 time-split, with phonemes of 16 - 64 ms 7.0 / 13.5)
    const int64_t scan_limit = ctx->scan_max_utts < 0 && ctx->ragged_option && rows_differ && fam == batch->n_utt
                                   ? 64 * (int64_t)ctx->cus : scan_max_utts(ctx);
    if (f.fast == 1u && ctx->scan_option && !lanes_option && (int64_t)fam * (l4ab ? 4 : 7) <= 4 * scan_limit &&
        used_voices_all(ctx, batch, ctx->voices_scan_ok, [](const grail_ctx::VoiceInfo &v) { return v.scan_ok; }) &&
        (batch->phoneme_mode || (batch->elems_scan_ok && batch->elems_warmup_epoch == ctx->voices_epoch)) &&
        batch->plain && batch->min_length >= 2.0f * ctx->max_dt &&
        batch->min_pitch * 0.999f - 1.002f * ctx->max_pitch_jitter >= 9.5367431640625e-07f) {
        scan.scan = true;
        scan.live4 = l4ab ? 1u : 0u;                          // (the scan kernel takes any blend length)
        // three-stage workgroups for few utterances (tools/scan_split_crossover.py: up to ~1500 with four
        // live formants, half that with eight, where the filter wave is the slower stage either way)
        scan.scan_pipe = (int64_t)fam * (scan.live4 ? 1 : 2) <= scan_split_max(ctx) ? 1u : 0u;
        scan.pipe = 0u;
    }
    // which of them: a pinned grid or an explicit "time_split_min_utterances" decide as they always did; otherwise
    // the cost model does (2 s utterances: the scan kernel up to ~1 500 of them, the time-split kernels up to half the
    // machine's lanes, the lane kernels beyond; shorter utterances move the first crossover up — a chunk's warm-up
    // does not shrink with the utterance)
    bool take_split = false, take_scan = false;
    if (split.split_k && ctx->split_chunks >= 2) {
        take_split = true;
    } else if (ctx->split_min_utts >= 0) {
        take_split = split.split_k && (int64_t)fam * 6 >= ctx->split_min_utts * (l4ab ? 6 : 5);
        take_scan = !take_split && scan.scan;
    } else {
        double c_lane = family_cost(ctx, f, fam, span);
        double c_split = split.split_k ? family_cost(ctx, split, fam, span) : INFINITY;
        double c_scan = scan.scan ? family_cost(ctx, scan, fam, span) : INFINITY;
        if (ctx->ragged_option && rows_differ && fam == batch->n_utt) {
            // Rows that differ in length (the whole batch, its summary from the upload; part of option "ragged_plan"): the three
            // families part ways.  The
            // scan kernel gives every utterance a workgroup of its own — what a compute unit works off is the SUM of its
            // utterances' lengths, whatever their spread, and an event costs a workgroup next to nothing (lanes are time) —
            // while a time-split lane fast-forwards through the events of 64 utterances and waits for the longest of them:
            // 4 096 speech-like utterances 6.7 ms on the scan kernel, 15.5 time-split (aligned: 6.4 against 3.3); with
            // phonemes of 16 - 64 ms 3.2 against 11.2.  Priced by the rows: mean length for the scan kernel (at least one
            // workgroup's way through the longest row), lengths and events (ragged_cost) for the other two.
            double mean = 0.0;
            for (const float g : batch->granule_samples) mean += (double)g;
            mean /= (double)batch->granule_samples.size();
            if (scan.scan) c_scan = std::fmax(family_cost(ctx, scan, fam, std::fmin(mean + 64.0, span)), family_cost(ctx, scan, 1u, span));
            if (split.split_k) c_split = ragged_cost(ctx, batch, split, 0u, fam, span);
            c_lane = ragged_cost(ctx, batch, f, 0u, fam, span);
        }
        take_split = c_split <= c_scan && c_split < c_lane;
        take_scan = !take_split && c_scan < c_lane;
    }
    if (take_split) f = split;
    else if (take_scan) f = scan;
    if (f.fast == 2u && !lanes_option && ctx->split_chunks < 2) {
        // The second tier costs 0.8 of the exact one-lane kernel (0.64 - 0.8 time-split): where the exact kernels have a
        // wider mapping to fill the machine with — mid-size batches of voices that do not qualify for time-splitting —
        // they are the faster way to the same tolerance (their bits satisfy it trivially).
        Family exact;
        choose_family(ctx, batch, out_stride, fam, exact, true);
        if (family_cost(ctx, exact, fam, span) <= family_cost(ctx, f, fam, span)) f = exact;
        return;
    }
    if (take_split || take_scan) return;
    // fast arithmetic asked for, but the batch takes neither the scan kernel nor the time-split kernels (caller-built
    // elems, a voice outside their windows, an option switched off) and is small enough for the pipelined exact
    // workgroups: those are faster than the fast lane kernels there (8.1 - 11.5 against 12.4 ms), and exact bits
    // satisfy the tolerance trivially
    if (want_pipe4 || want_pipe8) {
        f.fast = 0u;
        f.live4 = batch_live4_any_blend(ctx, batch) ? 1u : 0u;
        f.pipe = want_pipe4 ? pipe4_kind : pipe8_kind;
        f.L = want_pipe4 ? 4 : 8;
    }
}

// Cut `rows` rows into blocks, each rendered by the family that suits ITS size, so that the time of a batch is not a
// step function of its size: a family fills the machine with a fixed number of rows (one wave per SIMD), one row more
// costs a whole further round of it — 65 537 utterances took two rounds of the one-lane kernel (81 ms) where one
// round and a pipelined workgroup launch (40.6 + 6.5 ms) do.  Candidates: the whole of it in one launch; or a full
// block of one of the families' capacities (as many rounds as fit for the one-lane kernels) followed by the best
// plan for the rest.  Exact arithmetic is mapping-invariant, so the cut never changes a bit; in fast arithmetic a row's
// samples follow the family of ITS block (include/grail_hip.h, "Determinism contract").
struct Planner {
    const grail_ctx *ctx;
    const grail_batch *batch;
    uint64_t out_stride;
    double span;
    bool exact_only;                           // (the plan a fast request is weighed against: ragged_plan)
    static constexpr double LAUNCH_MS = 0.05;  // what a further launch costs by itself (measured: 0.02 - 0.06 ms)
    // (choose_family lays out time-split grids by bisection: every size is looked at once)
    std::map<uint32_t, std::pair<Family, double>> families;
    std::map<uint32_t, std::pair<double, std::vector<Block>>> plans;

    const std::pair<Family, double> &family(uint32_t rows)
    {
        auto it = families.find(rows);
        if (it != families.end()) return it->second;
        std::pair<Family, double> e;
        choose_family(ctx, batch, out_stride, rows, e.first, exact_only);
        e.second = family_cost(ctx, e.first, rows, span);
        return families.emplace(rows, e).first->second;
    }
    const std::pair<double, std::vector<Block>> &plan(uint32_t rows, int depth)
    {
        auto it = plans.find(rows);
        if (it != plans.end()) return it->second;
        const std::pair<Family, double> &whole = family(rows);
        double best = whole.second;
        std::vector<Block> best_plan{Block{rows, whole.first}};
        if (depth < 4) {
            const uint64_t lanes = ctx_lanes(ctx), cus = (uint64_t)ctx->cus;
            // the capacities at which some family is exactly full (largest first: of two plans of equal cost the
            // one with the larger head wins)
            const uint64_t caps[] = {lanes, lanes / 2, lanes / 4, lanes / 8, 32 * cus, 16 * cus, 8 * cus};
            uint64_t seen = 0;
            for (const uint64_t c : caps) {
                if (c == 0 || c >= rows || c == seen) continue;
                seen = c;
                const uint32_t m = c == lanes ? (uint32_t)(rows / c) : 1u;
                const uint32_t head = (uint32_t)(m * c);
                const std::pair<Family, double> &fc = family(head);
                if (fc.second + LAUNCH_MS >= best) continue;
                const std::pair<double, std::vector<Block>> &rest = plan(rows - head, depth + 1);
                if (fc.second + LAUNCH_MS + rest.first < best) {
                    best = fc.second + LAUNCH_MS + rest.first;
                    best_plan.assign(1, Block{head, fc.first});
                    best_plan.insert(best_plan.end(), rest.second.begin(), rest.second.end());
                }
            }
        }
        return plans.emplace(rows, std::make_pair(best, best_plan)).first->second;
    }
};

double plan_blocks(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride, uint32_t rows, double span,
                   std::vector<Block> &out, bool exact_only)
{
    Planner p{ctx, batch, out_stride, span, exact_only, {}, {}};
    const std::pair<double, std::vector<Block>> &best = p.plan(rows, 0);
    out = best.second;
    if (out.size() > 1) {
        // launch order: the block that is cheapest PER ROW first (in practice: the largest).  Two reasons.  Length-
        // sorted (ragged) batches hand out their slots longest first and a block lasts as long as its longest
        // utterance: with lengths falling by g per slot, moving a block of r rows and per-sample cost c behind one of
        // r', c' saves g (c r' - c' r) — the long utterances belong where a ROW costs least.  And a small block leaves
        // most of the machine idle for milliseconds: the large kernel behind it then starts on lowered clocks and
        // loses 2.5 - 3 ms (65 537 utterances: 50.3 ms with the single utterance first, profiles/r04_tail.txt).
        std::stable_sort(out.begin(), out.end(), [&](const Block &x, const Block &y) {
            return family_cost(ctx, x.f, x.rows, span) * (double)y.rows < family_cost(ctx, y.f, y.rows, span) * (double)x.rows;
        });
    }
    return best.first;
}


// ---- Ragged batches ------------------------------------------------------------------------------------------------------
// The families above are laid out for ONE round: the widest lane mapping that gives every SIMD one wave, priced by the
// batch's longest utterance.  On a length-sorted batch whose utterances differ much in length that leaves most SIMDs idle
// most of the time — the one wave of a SIMD lasts as long as ITS longest row, the launch as long as the longest of all —
// where a wider mapping in several rounds keeps them busy: its waves are shorter, they start longest first, and a SIMD that
// finishes a short one takes the next.  And a wave holds fewer utterances, so fewer of its tiles hold some lane's event.
// (Speech-like corpus, 65 536 utterances of 0.5 - 3.8 s: exact 90.0 ms one lane per utterance, 71.7 two; eight formants
// 171.6 / 113.6; fast 87.7 / 73.6 and 145.7 / 88.8; 100 000: 125 -> 94 in ONE launch of the one-lane kernel.  profiles/r04_ragged_plan.txt.)
// Model: a wave costs its longest row's samples at the mapping's rate plus what its rows' events cost.  Exact: the tiles
// that hold some lane's segment boundary, tiles x (1 - exp(-boundaries per tile)), at 7 / 8 us per 32 samples (four / eight
// formants), and 1.5 / 3.7 us per boundary on one lane per utterance.  Fast: the aligned rate x m plus c per event (below).
// Waves are handed to the SIMDs in launch order as they fall free.  The exact constants were fitted on pinned-mapping
// measurements of the speech-like corpus at 65 536 utterances with phonemes of 40 - 160, 16 - 64 and 4 - 16 ms, L = 1 / 2 / 4,
// four and eight formants (within 10 %, three L = 1 cells of the densest corpus 20 - 28 % under) and checked at 16 384 ...
// 131 072 utterances: profiles/r04_ragged_plan.txt.
static double ragged_wave_ms(const Family &f, double samples, double segs, double kinks)
{
    const bool nfa4 = f.live4 != 0;
    const int li = f.L == 1 ? 0 : f.L == 2 ? 1 : f.L == 4 ? 2 : 3;
    if (!f.fast) {
        // (the half-live loops of the one-lane kernel, 45.7 ms per 2 s, showed on aligned batches only: not priced in)
        const double T = (f.L == 1 || (!nfa4 && f.L == 4)) ? 32.0 : 64.0;
        const double tiles = std::fmax(samples / T, 1.0);
        const double event_tiles = tiles * (1.0 - std::exp(-segs / tiles));
        // (round 5: the 2 / 4 / 8-lane kernels render the samples between the events of such a tile by the calm tile's loops —
        // synth_kernel.h MIXED_RUNS — and a tile with an event costs them 0.5 - 0.75 of what it did; the four-lane kernel with
        // eight formants, tiles of 32 in runs of 8, gains nothing: profiles/r05_mixed_runs.txt)
        const double runs = f.L == 1 ? 1.0 : f.L == 2 ? (nfa4 ? 0.74 : 0.60) : f.L == 4 ? (nfa4 ? 0.51 : 1.0) : 0.58;
        return samples * lane_ms_per_sample(false, nfa4, f.L) + event_tiles * (nfa4 ? 0.007 : 0.008) * (T / 32.0) * runs +
               (f.L == 1 ? segs * (nfa4 ? 0.0015 : 0.0037) : 0.0);
    }
    // Fast (round 5: sub-tiles that never span an event, one slow sample per event — synth_kernel.h fast_render_tile): a wave
    // costs its longest row at the mapping's aligned rate x m — while a lane's parameters move (blends of 30 - 80 ms) its
    // sub-tiles are 16 samples instead of 32, and the wave takes new slopes at the pace of its busiest lane — plus c
    // per event of its rows (boundaries and kinks of alpha: the slow sample, the lane's two end points, the shorter runs
    // around it).  Fitted on pinned-mapping measurements of the speech-like corpus with phonemes of 40 - 160, 16 - 64 and
    // 4 - 16 ms at 65 536 utterances, one and eight voices, L = 1 / 2 / 4 / 8: all 24 cells within 3.3 % (tools/ragged_fit.py,
    // profiles/r05_ragged_fit.txt).  m fades to 1 where events are rare (long segments have long blends: the bench corpora).
    static const double m4[4] = {1.16, 1.10, 1.06, 1.02}, c4[4] = {0.00475, 0.0050, 0.00675, 0.00925};
    static const double m8[4] = {1.08, 1.06, 1.04, 1.02}, c8[4] = {0.00775, 0.00675, 0.0070, 0.0095};
    const double events = segs + kinks;
    const double density = events / ((64.0 / (double)f.L) * std::fmax(samples, 1.0));        // per lane and sample
    const double m = 1.0 + ((nfa4 ? m4 : m8)[li] - 1.0) * std::fmin(1.0, density / 3.0e-4);
    const double rate = f.fast == 2u ? mid_ms_per_sample(nfa4) : lane_ms_per_sample(true, nfa4, f.L);
    return samples * rate * m + events * (nfa4 ? c4 : c8)[li];
}

double ragged_cost(const grail_ctx *ctx, const grail_batch *batch, const Family &f, uint32_t slot0, uint32_t rows, double span)
{
    const size_t n_gran = batch->granule_samples.size();
    if (n_gran == 0 || rows == 0) return family_cost(ctx, f, rows, span);
    const size_t g0 = std::min<size_t>(slot0 / 8, n_gran - 1), g1 = std::min<size_t>(((size_t)slot0 + rows + 7) / 8, n_gran);
    // (families that render an utterance with many lanes: by the block's own longest row.  A pipelined workgroup renders its
    // 16 / 8 utterances in rounds of 32 samples; a round that holds a segment boundary of one of them costs it ~11 us more (14
    // before the runs between events):
    // 256 utterances with phonemes of 4 - 16 ms take 5.7 ms where their 0.39 s alone would take 1.3)
    if (f.scan) {
        // one workgroup per utterance: a compute unit works off the sum of its utterances' lengths (choose_family)
        double mean = 0.0;
        for (size_t g = g0; g < g1; ++g) mean += (double)batch->granule_samples[g];
        mean /= (double)(g1 - g0);
        return std::fmax(family_cost(ctx, f, rows, std::fmin(mean + 64.0, span)),
                         family_cost(ctx, f, 1u, std::fmin(span, (double)batch->granule_samples[g0] + 64.0)));
    }
    if (f.pipe) {
        double longest = std::fmin(span, std::fmax((double)batch->granule_samples[g0], 0.0) + 64.0);
        // (two workgroups per compute unit: the one with the shorter rows ends early and leaves the unit to the other —
        // 8 192 speech-like utterances take 19.6 ms where their longest row at the rate of two resident workgroups would
        // take 21.0; with phonemes of 16 - 64 ms 10.3, of 4 - 16 ms 4.6: profiles/r05_mixed_runs.txt)
        if (std::ceil((double)rows / (f.live4 ? 16.0 : 8.0)) > (double)ctx->cus) {
            double mean = 0.0;
            for (size_t g = g0; g < g1; ++g) mean += (double)batch->granule_samples[g];
            longest = 0.55 * longest + 0.45 * std::fmin(longest, mean / (double)(g1 - g0) + 64.0);
        }
        double c = family_cost(ctx, f, rows, longest);
        if (f.pipe) {
            c *= 1.15;      // (ragged corpora measure 15 - 25 % over the aligned rate before any event: pitch contours, stops)
            double segs = batch->granule_segs[g0];
            if (f.live4 && g0 + 1 < g1) segs += batch->granule_segs[g0 + 1];
            // (spread thinly: a workgroup's tiles hold the events of `fill` utterances, not of 16 / 8)
            const uint32_t fill = pipe_fill_for(ctx, batch, f, rows);
            if (fill) segs *= (double)fill / (f.live4 ? 16.0 : 8.0);
            const double rounds = std::fmax(longest / 32.0, 1.0);
            const double groups = std::ceil((double)rows / (f.live4 ? 16.0 : 8.0));
            c += 0.011 * rounds * (1.0 - std::exp(-segs / rounds)) * std::ceil(groups / ((f.pipe == 2 ? 1.0 : 2.0) * (double)ctx->cus));
        }
        return c;
    }
    const uint64_t simds = ctx_simds(ctx);
    if (f.split_k) {
        // every wave holds 64 utterances at one chunk index: the events of a one-lane wave; the first chunk of the
        // longest rows sets the time (x 1.2: fast-forward and restarts, from the same corpus)
        double segs = 0.0, kinks = 0.0;
        for (size_t g = g0; g < std::min(g0 + 8, g1); ++g) {
            segs += batch->granule_segs[g];
            kinks += batch->granule_kinks[g];
        }
        const double len = std::fmax((double)batch->granule_samples[g0], 64.0), part = (double)f.split_bounds[1] / len;
        Family lane = f;
        lane.split_k = 0;
        lane.L = 1;
        const double rounds = f.split_active ? std::ceil((double)f.split_active / (double)simds)
                                             : std::ceil(std::ceil((double)rows / 64.0) * f.split_k / (double)simds);
        return rounds * (1.2 * ragged_wave_ms(lane, (double)f.split_bounds[1], segs * part, kinks * part) + 0.12);
    }
    // waves of 64 / L consecutive slots, handed out in launch order to the SIMD that falls free first
    const size_t per_wave = (size_t)(8 / f.L > 0 ? 8 / f.L : 1);
    auto wave_ms = [&](size_t g) {
        double samples = 0.0, segs = 0.0, kinks = 0.0;
        for (size_t k = g; k < std::min(g + per_wave, g1); ++k) {
            samples = std::fmax(samples, (double)batch->granule_samples[k]);
            segs += batch->granule_segs[k];
            kinks += batch->granule_kinks[k];
        }
        samples = std::fmin(samples + 64.0, span);
        return ragged_wave_ms(f, samples, segs, kinks);
    };
    if (family_cohabits(ctx, f, rows)) {
        // Two waves per SIMD: while two are resident each advances at 1 / (2 gain) of the lone wave's pace (together
        // 1 / gain: the 20 - 25 % the second wave adds); the one left behind by a shorter neighbour runs on alone at full
        // pace.  Waves are handed out in launch order, two per SIMD at first, then to the SIMD that has a slot free first.
        struct Simd { double t, a, b; };              // time reached; work left (in lone-wave ms) in the two slots, a <= b; < 0: empty
        double ev_sum = 0.0, lane_samples = 0.0;
        for (size_t g = g0; g < g1; ++g) {
            ev_sum += (double)batch->granule_segs[g] + (double)batch->granule_kinks[g];
            lane_samples += 8.0 * std::fmax((double)batch->granule_samples[g], 1.0);
        }
        const double pace2 = 1.0 / (2.0 * cohabit_gain(f, ev_sum / std::fmax(lane_samples, 1.0)));
        std::vector<Simd> sm(simds, Simd{0.0, -1.0, -1.0});
        auto next_done = [&](const Simd &m) {         // when the SIMD's next wave ends (it has at least one)
            return m.a < 0.0 ? m.t + m.b : m.t + m.a / pace2;
        };
        typedef std::pair<double, size_t> Ev;
        std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> done_at;
        // (a launch of at most two rounds: the waves of the second take their slots in reverse order — synth_kernel.h FOLD —
        // so that the SIMD with the longest rows of the first round gets the shortest of the second)
        const size_t n_waves = (g1 - g0 + per_wave - 1) / per_wave;
        const bool fold = n_waves > simds && n_waves <= 2 * simds;
        size_t g = g0, filled = 0;
        for (; g < g1 && filled < 2 * simds; g += per_wave, ++filled) {
            Simd &m = sm[filled % simds];
            const double w = wave_ms(fold && filled >= simds ? g0 + (n_waves - 1 - (filled - simds)) * per_wave : g);
            if (m.b < 0.0) m.b = w;
            else if (w <= m.b) m.a = w;
            else { m.a = m.b; m.b = w; }
        }
        for (size_t i = 0; i < simds; ++i)
            if (sm[i].b >= 0.0) done_at.push(Ev(next_done(sm[i]), i));
        double makespan = 0.0;
        while (!done_at.empty()) {
            const Ev ev = done_at.top();
            done_at.pop();
            Simd &m = sm[ev.second];
            if (m.a < 0.0) {                          // the lone wave ends
                m.t += m.b;
                m.b = -1.0;
            } else {                                  // the shorter of two ends; the other has advanced as far
                m.t += m.a / pace2;
                m.b -= m.a;
                m.a = -1.0;
            }
            makespan = std::fmax(makespan, m.t);
            if (g < g1) {                             // the slot takes the next wave of the launch
                const double w = wave_ms(g);
                g += per_wave;
                if (m.b < 0.0) m.b = w;
                else if (w <= m.b) m.a = w;
                else { m.a = m.b; m.b = w; }
            }
            if (m.b >= 0.0) done_at.push(Ev(next_done(m), ev.second));
        }
        // Against measurements of every pinned mapping on 40 000 ... 200 000 speech-like rows, one and eight voices
        // (profiles/r06_plan_debug.txt): the exact two-wave kernels run ahead of this simulation — two lanes 0.77 - 0.86 of
        // it (they got their runs between events after the wave prices were fitted), four lanes 0.86 - 0.95 — at every
        // size, and the tolerance-mode two-lane kernel falls behind it (1.13 - 1.20) once a launch holds more waves than
        // are resident at once.  With the one-wave-per-SIMD launches priced by the dispatcher's model (within 0.93 - 1.0
        // of their measurements), the two-wave prices are brought to the same scale here.
        const double waves = std::ceil((double)(g1 - g0) / (double)per_wave);
        const double scale = !f.fast ? (f.L == 2 ? 0.84 : 0.91) : (f.L == 2 && waves > 2.0 * (double)simds ? 1.15 : 1.0);
        return makespan * scale;
    }
    // one wave per SIMD: the workgroups of the launch through the dispatcher's model, in the plain (longest rows first) or
    // in the packed order, whichever the launch will take (packed_launch_order: the same decision)
    double plain_ms = 0.0, packed_ms = 0.0;
    const bool packed = packed_launch_order(ctx, batch, f, slot0, rows, span, nullptr, nullptr, &plain_ms, &packed_ms);
    // (x 0.96: the wave prices of ragged_wave_ms through the dispatcher's model come out 0 - 7 % over the measurements,
    // uniformly over mappings, sizes and both arithmetics — profiles/r06_plan_debug.txt)
    return 0.96 * (packed ? packed_ms : plain_ms);
}

// ---- The workgroup dispatcher, as measured (tools/dispatch_order.hip: workgroups that spin for given times and record
// where and when they ran; profiles/r06_dispatch_order.txt).  A launch of one-wave workgroups that each hold a SIMD alone:
//   * workgroup b runs on XCC b mod 8 (8 XCCs of 32 compute units);
//   * the k-th workgroup of an XCC goes to shader engine pattern[k mod 4] — a STATIC round robin over the XCC's four
//     engines (8 compute units = 32 SIMDs each), whatever their load;
//   * it starts when that engine has a SIMD free AND every earlier workgroup of the XCC has started (in order: a full
//     engine holds up the workgroups behind it that are bound for the others).
// This model reproduces the makespan of recorded launches of 2 048 - 3 125 workgroups to the microsecond (ten launches,
// longest first and packed orders).  So workgroup b belongs to pool b mod 32, each pool 32 SIMDs, and "the next wave goes
// to the SIMD that falls free first" holds within a pool only.  Workgroups of four waves (L >= 4) take a compute unit: 8 per
// engine.  A device that is not whole XCCs (a test's "assume_compute_units"): one pool.
struct Dispatcher {
    uint32_t xcc = 1, se = 1, slots = 1;
    uint32_t pools() const { return xcc * se; }
};

static Dispatcher dispatcher_of(const grail_ctx *ctx, const uint32_t waves_per_block)
{
    Dispatcher d;
    const uint32_t cus = (uint32_t)ctx->cus;
    if (cus >= 32u && cus % 32u == 0u) {
        d.xcc = cus / 32u;
        d.se = 4u;
        d.slots = waves_per_block == 1u ? 32u : 8u;
    } else {
        d.slots = std::max(1u, waves_per_block == 1u ? 4u * cus : cus);
    }
    return d;
}

// (costs that are not finite, or negative, count as 0 / 1e30: the model's callers hand over sane prices, the C entry points anything)
static double sane_cost(const double c) { return c > 0.0 ? (c < 1e30 ? c : 1e30) : 0.0; }

// cost[b]: what workgroup b takes; order (or nullptr: 0, 1, 2 ...): the workgroup at each launch position
static double dispatch_makespan(const Dispatcher &d, const std::vector<double> &cost, const std::vector<uint32_t> *order)
{
    typedef std::priority_queue<double, std::vector<double>, std::greater<double>> Free;
    const size_t n = order ? order->size() : cost.size();
    double worst = 0.0;
    for (uint32_t x = 0; x < d.xcc; ++x) {
        std::vector<Free> engine(d.se);
        std::vector<uint32_t> used(d.se, 0u);
        double prev = 0.0;
        uint32_t k = 0;
        for (size_t b = x; b < n; b += d.xcc, ++k) {
            const uint32_t e = k % d.se;
            double t = prev;                                     // every earlier workgroup of the XCC has started
            if (used[e] < d.slots) ++used[e];                    // (a SIMD of the engine that has not run anything yet)
            else {
                t = std::fmax(t, engine[e].top());
                engine[e].pop();
            }
            prev = t;
            const double end = t + sane_cost(cost[order ? (*order)[b] : b]);
            engine[e].push(end);
            if (end > worst) worst = end;
        }
    }
    return worst;
}

// Best-fit decreasing of `jobs` (indices into cost, cost descending) into at most `bins` bins under capacity `cap`:
// bin_of[i] for jobs[i], or false.
static bool fit_decreasing(const std::vector<double> &cost, const std::vector<uint32_t> &jobs, const uint32_t bins, const double cap,
                           std::vector<uint32_t> &bin_of)
{
    std::vector<double> room;                                    // remaining capacity of the bins opened so far (a few dozen: scanned)
    room.reserve(bins);
    bin_of.resize(jobs.size());
    for (size_t i = 0; i < jobs.size(); ++i) {
        const double c = cost[jobs[i]];
        size_t best = room.size();
        for (size_t b = 0; b < room.size(); ++b)
            if (room[b] >= c && (best == room.size() || room[b] < room[best])) best = b;
        if (best == room.size()) {
            if (room.size() == bins || c > cap) return false;
            room.push_back(cap);
        }
        bin_of[i] = (uint32_t)best;
        room[best] -= c;
    }
    return true;
}

// The packed order.  Longest first leaves the SIMDs of a pool uneven at the end when a SIMD gets two or three workgroups
// (131 072 speech-like rows: the slowest SIMD 1.15 x the mean); packing evens them out, and launching the workgroups in the
// order of their planned start times makes the dispatcher reproduce the packing.  Workgroups are dealt to the pools in turn
// (by cost: the pools get alike sets), each pool is packed into its SIMDs under the smallest capacity that fits
// (bisection), and position p + pools * k of the launch takes the k-th workgroup of pool p by planned start.
static void pack_order(const Dispatcher &d, const std::vector<double> &cost_in, const size_t n_jobs, std::vector<uint32_t> &order)
{
    std::vector<double> cost(cost_in.size());
    for (size_t i = 0; i < cost.size(); ++i) cost[i] = sane_cost(cost_in[i]);
    const uint32_t P = d.pools();
    std::vector<uint32_t> by_cost(n_jobs);
    for (size_t j = 0; j < n_jobs; ++j) by_cost[j] = (uint32_t)j;
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    order.resize(n_jobs);
    std::vector<uint32_t> jobs, bin_of, best_bin;
    for (uint32_t p = 0; p < P; ++p) {
        jobs.clear();
        for (size_t j = p; j < n_jobs; j += P) jobs.push_back(by_cost[j]);
        if (jobs.empty()) continue;
        double sum = 0.0;
        for (uint32_t j : jobs) sum += cost[j];
        double lo = std::fmax(sum / (double)d.slots, cost[jobs[0]]), hi = lo;
        hi = std::fmax(hi, 1e-300);
        while (!fit_decreasing(cost, jobs, d.slots, hi, best_bin)) hi *= 1.25;       // (terminates: at hi >= sum everything fits one bin)
        for (int it = 0; it < 14 && hi - lo > 5e-4 * hi; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (fit_decreasing(cost, jobs, d.slots, mid, bin_of)) {
                hi = mid;
                best_bin.swap(bin_of);
            } else {
                lo = mid;
            }
        }
        // planned start of every workgroup: behind the ones before it in its bin (cost descending within a bin)
        std::vector<double> bin_load(d.slots, 0.0);
        std::vector<std::pair<double, uint32_t>> seq(jobs.size());                   // (start, rank by cost): stable
        for (size_t i = 0; i < jobs.size(); ++i) {
            seq[i] = std::make_pair(bin_load[best_bin[i]], (uint32_t)i);
            bin_load[best_bin[i]] += cost[jobs[i]];
        }
        std::sort(seq.begin(), seq.end());
        for (size_t k = 0; k < seq.size(); ++k) order[p + (size_t)P * k] = jobs[seq[k].second];
    }
}

bool packed_launch_order(const grail_ctx *ctx, const grail_batch *batch, const Family &f, uint32_t slot0, uint32_t rows, double span,
                         std::vector<uint32_t> *order, uint32_t *rows_per_block, double *plain_ms, double *packed_ms)
{
    if (order) order->clear();
    const size_t n_gran = batch->granule_samples.size();
    const uint32_t waves_per_block = f.L >= 4 ? 4u : 1u;
    const uint32_t per_wave_rows = 64u / (uint32_t)f.L, per_block = per_wave_rows * waves_per_block;
    if (rows_per_block) *rows_per_block = per_block;
    if (n_gran == 0 || rows == 0 || f.scan || f.pipe || f.split_k || slot0 % 8u != 0u) return false;
    const size_t g0 = std::min<size_t>(slot0 / 8, n_gran - 1), g1 = std::min<size_t>(((size_t)slot0 + rows + 7) / 8, n_gran);
    const size_t per_wave = (size_t)(8 / f.L > 0 ? 8 / f.L : 1), per_block_g = per_wave * waves_per_block;
    // what every workgroup costs: its slowest wave (longest row, the rows' events)
    std::vector<double> cost;
    for (size_t g = g0; g < g1; g += per_block_g) {
        double worst = 0.0;
        for (size_t w = g; w < std::min(g + per_block_g, g1); w += per_wave) {
            double samples = 0.0, segs = 0.0, kinks = 0.0;
            for (size_t k = w; k < std::min(w + per_wave, g1); ++k) {
                samples = std::fmax(samples, (double)batch->granule_samples[k]);
                segs += batch->granule_segs[k];
                kinks += batch->granule_kinks[k];
            }
            worst = std::fmax(worst, ragged_wave_ms(f, std::fmin(samples + 64.0, span), segs, kinks));
        }
        cost.push_back(worst);
    }
    const Dispatcher d = dispatcher_of(ctx, waves_per_block);
    const double plain = dispatch_makespan(d, cost, nullptr);
    if (plain_ms) *plain_ms = plain;
    if (packed_ms) *packed_ms = plain;
    // more workgroups than the device holds at once, each alone on its SIMDs (two waves per SIMD: the fold of synth_kernel.h),
    // and only whole workgroups are moved: a last one with fewer rows keeps the last position
    const size_t n_full = rows / per_block;
    if (!ctx->packed_option || family_cohabits(ctx, f, rows) || cost.size() <= (size_t)d.pools() * d.slots || n_full < 2) return false;
    std::vector<uint32_t> o;
    pack_order(d, cost, n_full, o);
    for (size_t b = n_full; b < cost.size(); ++b) o.push_back((uint32_t)b);
    const double packed = dispatch_makespan(d, cost, &o);
    if (!(packed < 0.985 * plain)) return false;                 // (not worth a table of its own)
    if (packed_ms) *packed_ms = packed;
    if (order) order->swap(o);
    return true;
}

void ragged_plan(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride, uint32_t rows, std::vector<Block> &plan)
{
    if (!ctx->ragged_option || batch->granule_samples.empty() || rows != batch->n_utt || plan.empty()) return;
    if (plan.size() == 1 && plan[0].f.scan) return;   // (a few utterances in fast arithmetic: the scan kernel's)
    const double span = batch_span(ctx, batch, out_stride);
    auto cost_of = [&](const std::vector<Block> &blocks) {
        double c = 0.0;
        uint32_t slot0 = 0;
        for (const Block &b : blocks) {
            c += ragged_cost(ctx, batch, b.f, slot0, b.rows, span) + Planner::LAUNCH_MS;
            slot0 += b.rows;
        }
        return c;
    };
    // GRAIL_PLAN_DEBUG=1: the candidates and their prices on stderr (development aid; read once)
    static const bool debug = [] { const char *e = getenv("GRAIL_PLAN_DEBUG"); return e && *e && *e != '0'; }();
    // (a candidate has to be worth the change: in tolerance arithmetic a row's bits follow its family, and grail_plan_blocks
    // predicts the cut by size; exact bits follow nothing, and one launch in packed order beats the same mapping cut in two)
    const double worth = ctx->fast_option ? 0.95 : 0.98;
    double best = worth * cost_of(plan);
    if (debug) std::fprintf(stderr, "[ragged_plan] %u rows: the cut by size (%zu block(s)) %.2f ms\n", rows, plan.size(), best / worth);
    if (ctx->fast_option) {
        // fast arithmetic asked for: the cut exact arithmetic would get stands too (small batches: the pipelined
        // workgroups) — events this dense cost the fast kernels more than they save, and exact bits satisfy the tolerance
        // trivially.  Phonemes of 16 - 64 ms, 65 536 utterances: 32 ms exact against 57 fast.
        std::vector<Block> exact;
        plan_blocks(ctx, batch, out_stride, rows, span, exact, true);
        const double c = cost_of(exact);
        if (c < best) {
            best = c;
            plan = exact;
        }
    }
    // ... the ONE launch the batch as a whole would get (choose_family weighs scan, time-split and lane kernels by the rows
    // when it is asked about the whole batch; the cut above was made block by block, by the aligned model: 6 000 speech-like
    // utterances as 4 096 + 1 904 time-split rows took 26.9 ms, the scan kernel takes 10.9)
    if (ctx->fast_option) {       // (exact arithmetic: that launch is one of the lane mappings below)
        Family whole;
        choose_family(ctx, batch, out_stride, rows, whole, false);
        const double c = ragged_cost(ctx, batch, whole, 0, rows, span) + Planner::LAUNCH_MS;
        if (c < best) {
            best = c;
            plan.assign(1, Block{rows, whole});
        }
    }
    // ... and ONE launch of each lane mapping in as many rounds as it takes
    for (int exact_only = 0; exact_only <= (ctx->fast_option ? 1 : 0); ++exact_only)
        for (int L = 1; L <= 8; L *= 2) {
            Family f;
            choose_family(ctx, batch, out_stride, rows, f, exact_only != 0, L);
            if (f.scan || f.pipe || f.split_k) continue;
            const double c = ragged_cost(ctx, batch, f, 0, rows, span) + Planner::LAUNCH_MS;
            if (debug)
                std::fprintf(stderr, "[ragged_plan]   one launch on %d lane(s)%s%s: %.2f ms\n", f.L, f.fast ? " fast" : " exact",
                             family_cohabits(ctx, f, rows) ? ", two waves per SIMD" : "", c);
            if (c < best) {
                best = c;
                plan.assign(1, Block{rows, f});
            }
        }
}

}  // namespace host
}  // namespace grail

extern "C" {

int grail_dispatch_model(uint32_t compute_units, uint32_t waves_per_workgroup, const double *workgroup_ms,
                         const uint32_t *order, uint32_t n, double *makespan_ms)
{
    if (!makespan_ms || (n && !workgroup_ms)) return fail(GRAIL_ERR_INVALID_ARG, "NULL argument");
    if (compute_units == 0 || compute_units > 4096 || (waves_per_workgroup != 1 && waves_per_workgroup != 4))
        return fail(GRAIL_ERR_INVALID_ARG, "compute_units must be 1 .. 4096, waves_per_workgroup 1 or 4");
    grail_ctx ctx;
    ctx.cus = ctx.device_cus = (int)compute_units;
    std::vector<double> cost(workgroup_ms, workgroup_ms + n);
    std::vector<uint32_t> o;
    if (order) {
        o.assign(order, order + n);
        for (uint32_t b : o)
            if (b >= n) return fail(GRAIL_ERR_INVALID_ARG, "order names a workgroup beyond n");
    }
    *makespan_ms = dispatch_makespan(dispatcher_of(&ctx, waves_per_workgroup), cost, order ? &o : nullptr);
    return GRAIL_OK;
}

int grail_packed_launch_order(uint32_t compute_units, uint32_t waves_per_workgroup, const double *workgroup_ms, uint32_t n,
                              uint32_t *order)
{
    if (n && (!workgroup_ms || !order)) return fail(GRAIL_ERR_INVALID_ARG, "NULL argument");
    if (compute_units == 0 || compute_units > 4096 || (waves_per_workgroup != 1 && waves_per_workgroup != 4))
        return fail(GRAIL_ERR_INVALID_ARG, "compute_units must be 1 .. 4096, waves_per_workgroup 1 or 4");
    grail_ctx ctx;
    ctx.cus = ctx.device_cus = (int)compute_units;
    std::vector<double> cost(workgroup_ms, workgroup_ms + n);
    std::vector<uint32_t> o;
    pack_order(dispatcher_of(&ctx, waves_per_workgroup), cost, n, o);
    std::copy(o.begin(), o.end(), order);
    return GRAIL_OK;
}

// grail_plan_blocks / grail_plan_ragged_blocks: row_samples == nullptr is the aligned batch
static int plan_preview(uint32_t compute_units, int arithmetic, int live_formants, uint32_t warmup, uint32_t rows,
                        uint32_t span_samples, const uint32_t *row_samples, const uint32_t *row_segments,
                        const uint32_t *row_kinks, grail_plan_block *blocks, uint32_t cap, uint32_t *n_blocks)
{
    if (!n_blocks) return fail(GRAIL_ERR_INVALID_ARG, "n_blocks is NULL");
    *n_blocks = 0;
    if (compute_units == 0 || compute_units > 4096) return fail(GRAIL_ERR_INVALID_ARG, "compute_units must be 1 .. 4096");
    if (live_formants != 4 && live_formants != 8) return fail(GRAIL_ERR_INVALID_ARG, "live_formants must be 4 or 8");
    if (arithmetic != 0 && arithmetic != 1 && arithmetic != 2) return fail(GRAIL_ERR_INVALID_ARG, "arithmetic must be 0, 1 or 2");
    if (rows == 0) return GRAIL_OK;
    // a context and a batch as choose_family sees them: default options, a voice table that qualifies for every
    // family (four or eight live formants), a plain phoneme batch with power-of-two blend lengths
    grail_ctx ctx;
    ctx.cus = ctx.device_cus = (int)compute_units;
    ctx.fast_option = arithmetic;
    ctx.voices_sharpness = 0.0;
    ctx.voices_upper_silent = ctx.voices_live4_ok = live_formants == 4;
    ctx.voices_scan_ok = true;
    ctx.voices_split_ok = warmup != 0u;
    ctx.max_warmup = warmup;
    ctx.max_rate = 1.0f;                  // max_seconds below is in samples
    ctx.max_dt = 1.0f;
    grail_batch batch;
    batch.n_utt = rows;
    batch.phoneme_mode = true;
    batch.plain = true;
    batch.max_seconds = (float)span_samples;
    batch.min_length = 1e9f;
    batch.min_pitch = 0.25f;
    if (row_samples) {
        batch.len_bound_known = true;        // (what upload_len_bound keeps)
        // (what upload_length_order keeps of a length-sorted batch)
        const size_t n_gran = ((size_t)rows + 7) / 8;
        batch.granule_samples.assign(n_gran, 0.0f);
        batch.granule_segs.assign(n_gran, 0u);
        batch.granule_kinks.assign(n_gran, 0u);
        for (uint32_t s = 0; s < rows; ++s) {
            batch.granule_samples[s / 8] = std::fmax(batch.granule_samples[s / 8], (float)row_samples[s]);
            batch.granule_segs[s / 8] += row_segments ? row_segments[s] : 0u;
            batch.granule_kinks[s / 8] += row_kinks ? row_kinks[s] : 0u;
        }
    }
    const uint64_t stride = ((uint64_t)span_samples + 64u + 63u) / 64u * 64u;
    std::vector<Block> plan;
    plan_blocks(&ctx, &batch, stride, rows, batch_span(&ctx, &batch, stride), plan);
    ragged_plan(&ctx, &batch, stride, rows, plan);
    *n_blocks = (uint32_t)plan.size();
    uint32_t slot0 = 0;
    for (uint32_t i = 0; i < plan.size(); ++i) {
        const Family &f = plan[i].f;
        if (i < cap && blocks) {
            blocks[i].rows = plan[i].rows;
            blocks[i].lanes_per_utterance = f.scan ? 0u : (uint32_t)f.L;
            blocks[i].pipelined = f.scan ? 0u : f.pipe;
            blocks[i].chunks = (uint32_t)f.split_k;
            blocks[i].scan = f.scan ? (f.scan_pipe ? 2u : 1u) : 0u;
            blocks[i].fast = f.fast;
            blocks[i].formants = f.live4 ? 4u : 8u;
            blocks[i].model_ms = (float)ragged_cost(&ctx, &batch, f, slot0, plan[i].rows, batch_span(&ctx, &batch, stride));
        }
        slot0 += plan[i].rows;
    }
    return GRAIL_OK;
}

int grail_plan_blocks(uint32_t compute_units, int arithmetic, int live_formants, uint32_t warmup, uint32_t rows,
                      uint32_t span_samples, grail_plan_block *blocks, uint32_t cap, uint32_t *n_blocks)
{
    return plan_preview(compute_units, arithmetic, live_formants, warmup, rows, span_samples, nullptr, nullptr, nullptr,
                        blocks, cap, n_blocks);
}

int grail_plan_ragged_blocks(uint32_t compute_units, int arithmetic, int live_formants, uint32_t warmup, uint32_t rows,
                             const uint32_t *row_samples, const uint32_t *row_segments, const uint32_t *row_kinks,
                             grail_plan_block *blocks, uint32_t cap, uint32_t *n_blocks)
{
    if (n_blocks) *n_blocks = 0;
    if (rows && !row_samples) return fail(GRAIL_ERR_INVALID_ARG, "row_samples is NULL");
    uint32_t span = 0;
    for (uint32_t s = 0; s < rows; ++s) span = std::max(span, row_samples[s]);
    return plan_preview(compute_units, arithmetic, live_formants, warmup, rows, span, row_samples, row_segments, row_kinks,
                        blocks, cap, n_blocks);
}

}  // extern "C"
