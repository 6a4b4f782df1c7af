// synth_kernel.h — the fused Selector -> Sequencer -> Jitter -> Synthesize kernel
// for gfx950 (MI355X, CDNA4, wave64).  Hand-written HIP; no MFMA (the path is a
// per-sample IIR recurrence, VALU-issue bound, ~4 B of HBM traffic per sample).
//
// Reference behaviour (file:line in the grail-rs tree):
//   Selector::next    src/lib.rs:990-1005     Sequencer::next  src/lib.rs:859-932
//   Jitter::next      src/lib.rs:753-777      Synthesize::next src/lib.rs:497-578
//   ValueNoise        src/lib.rs:227-255      ArrayValueNoise  src/lib.rs:270-306
//   random_f32 :36    tan_approx :63          exp_approx :75   Array::sum :123
//
// Mapping.  One wavefront renders S = 64/L utterances; the 8 formants of an
// utterance are spread over L adjacent lanes (L in {1,2,4,8}, FPL = 8/L formants
// per lane).  Time is serial (phase, clocks, RNG and filter states all carry
// sample to sample, exactly as in the reference); the per-utterance scalar
// state is recomputed identically in each of its L lanes so lanes never wait on
// each other.  The 8-term `Array::sum` is a left fold and must stay one: for
// L = 2 it runs as a chain down the lanes with DPP row_shr:1 hand-offs, for
// L >= 4 the lanes park their band-pass outputs in LDS and the fold runs at
// flush time.  Samples are staged through LDS for T steps and flushed as
// 16-B-per-lane row stores, so every utterance row is written in contiguous
// 4*T-byte runs (8-B stores for i16 PCM rows).
//
// Steps.  general_step: the literal control flow of the reference with IEEE
// divisions, taken whenever some lane has an event (segment boundary, jitter
// wrap, full row) or its segment pair is outside the proven operand window.
// quiet_step: the same arithmetic straight-line, short exact divisions, behind
// one ballot per step.  Calm tiles: T quiet steps without that ballot, when no
// lane can have an event before the tile ends (see the tile loop).
//
// Exactness.  Built with -ffp-contract=off: the compiler never fuses a*b+c, so every
// multiply and add of the reference is an individually rounded IEEE operation.  The
// few explicit fma calls are places where a fused form is PROVEN to round the same
// real number once (the division sequences, 5 - 4*p with an exact 4*p, 2*x - 1 with an
// exact 2*x); divisions are correctly rounded (hipcc's IEEE sequence, or the
// proven-equal short sequence div_exact<true>); f32 denormals are kept (the kernel
// descriptor's default).  The result is bit-identical to the reference arithmetic,
// whatever L is.
//
// Packed math.  A lone wave issues at most one instruction every ~5 cycles, whatever the
// instruction (measured, tools/valu_microbench.hip), and the headline batch is exactly one
// wave per SIMD, so the scarce resource is issue slots.  The per-formant arithmetic is
// therefore written on float2 values, which hipcc lowers to v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32: two formants per issue slot, each component still an individually
// rounded IEEE operation.
//
// Files.  The kernel is ONE function template (its pieces are lambdas over the lane's state, so that every piece
// sees the registers of the others), cut textually into fragments that this file includes in order — the
// instantiation units preprocess to one token stream, and moving a line between fragments moves nothing else:
//   synth_kernel_parts.h          namespace level: DPP hand-offs, Part, formant_filters (:531-571), coefficients,
//                                 operand windows of the short division, the stream state mover
//   synth_kernel_state.h          prologue: the lane's utterance, the chain's state in registers, setup_pair
//   synth_kernel_general_step.h   general_step: the reference's control flow for one sample
//   synth_kernel_calm_steps.h     quiet_step; one formant per lane, two samples per packed slot
//   synth_kernel_pipe.h           PIPE: the calm steps as a three-stage pipeline over four waves
//   synth_kernel_scalar_packed.h  two calm samples per trip for L < 8
//   synth_kernel_fast.h           FAST: tolerance-mode arithmetic (sub-tiles, error guard, packed chain)
//   synth_kernel_flush.h          flush_rows: staged tile -> memory
//   synth_kernel_fast_tile.h      FAST: one tile, every lane deciding for itself
//   synth_kernel_split.h          SPLIT: fast-forward to a chunk's start
//   synth_kernel_tile_loop.h      the tile loop (calm tiles, tiles with events, the end of the launch)
#pragma once
#include <cstdio>
#include <type_traits>

#include "device_common.h"
#include "kernels.h"
#include "pcm16.h"

#ifndef GRAIL_MIXED_RUNS
#define GRAIL_MIXED_RUNS 1
#endif
#ifndef GRAIL_MIXED_RUNS_L1
#define GRAIL_MIXED_RUNS_L1 0
#endif
#ifndef GRAIL_MIXED_RUNS_PIPE
#define GRAIL_MIXED_RUNS_PIPE 1
#endif
#ifndef GRAIL_SPLIT_SKIP
#define GRAIL_SPLIT_SKIP 1
#endif
#ifndef GRAIL_PIPE_PARTIAL
#define GRAIL_PIPE_PARTIAL 1
#endif
// PIPE, rounds of 32 samples: how many of a round's 16 coefficient pairs the chain wave takes (8 / 4 lanes per utterance)
#ifndef PIPE_CP8_L8
#define PIPE_CP8_L8 4
#endif
#ifndef PIPE_CP8_L4
#define PIPE_CP8_L4 2
#endif
#ifndef GRAIL_SCALAR_PACK
#define GRAIL_SCALAR_PACK 1
#endif
#ifndef GRAIL_FAST_G_SCALE
// 2^20 / 4: interpolation error of G, H <= 2^-20 (fast_level).  2^-22 until round 5: the products of the amplitude with the
// amplitude jitter and with the turbulence kept a wave's busiest lane at sub-tiles of 4 - 8 samples through every blend of a
// speech-like corpus; four times the bound takes 13 % off those batches and moves the largest deviation measured anywhere
// from 13 to 18 * 2^-23 on such a corpus (25 on random voice tables at the served sharpness, where the resonances decide,
// unchanged; sixteen times: 57 — too close to the contract's 64).  profiles/r05_guard_scale.txt
#define GRAIL_FAST_G_SCALE 262144.0f
#endif
#ifndef GRAIL_FAST_A_SCALE
#define GRAIL_FAST_A_SCALE 724.0773439350247f   // 2^9.5: relative change of a1, a2 / a1, 1 - k per sub-tile <= 2^-9.5
#endif
#ifndef PIPE_MAX_TILES
#define PIPE_MAX_TILES 64         // PIPE kernels: consecutive calm tiles rendered without draining the pipeline (8: 7.85 ms for
                                  // config 2, 24: 7.49, 64: 7.41; a run ends at the next event anyway, ~40 tiles)
#endif

// GRAIL_FAST_PROF (debug builds only, `make EXTRA=-DGRAIL_FAST_PROF`): cycle and event counters of the tolerance-mode
// tile loop, summed over the waves of a launch into 32 u64 words behind A.truncated[8] (tools/fast_prof.py reads them)
#ifdef GRAIL_FAST_PROF
#define PROF_ADD(k) do { const unsigned long long n_ = clock64(); prof_c[k] += n_ - prof_t0; prof_t0 = n_; } while (0)
#define PROF_CNT(k, v) do { prof_c[k] += (unsigned long long)(v); } while (0)
#else
#define PROF_ADD(k) do { } while (0)
#define PROF_CNT(k, v) do { } while (0)
#endif

namespace grail {

// what the last launch_synth call of this thread started (synth_kernels.hip)
extern thread_local char g_kernel_name[96];

namespace {

#include "synth_kernel_parts.h"

// HALF: instantiate the quiet loops that skip a silent upper half of the lane's formants.  The
// host only asks for it when the voice table can make use of it (or for resumable streams), so
// batches whose formants are all audible run a kernel that does not carry those loops.
// ANYBL: blend lengths that are not powers of two also take the quiet step (clk / blend_length by
// the short exact division).  The host asks for it only when the batch holds such a segment, so the
// usual case (the Intonator always emits 0.5, src/lib.rs:1071) runs a kernel without that code.
// NFA: formants laid out over the lanes, 8 or 4.  NFA = 4 (phoneme batches only) renders
// formants 1-4 and nothing else: the host has verified (voice_analysis.cpp, live4_ok) that formants 5-8
// of every phoneme of every voice have amplitude +0 and parameters for which the reference's own
// arithmetic keeps their band-pass state and output at exactly +0 for the whole batch, so the fold
// only gains literal +0.0 terms.
// PIPE: the four waves of a workgroup share ONE set of 16 utterances (small batches, idle SIMDs).
// In calm tiles wave 0 runs the filter recurrences, wave 1 the per-utterance chain (its four lanes per
// utterance sharing the sample pairs of a round) and a quarter of the filter coefficients, waves 2 and 3
// the other coefficients, each stage handing its results on through LDS one round of 16 samples behind
// the previous one (pipe_chain / pipe_coeffs / pipe_render below); runs of up to PIPE_MAX_TILES calm
// tiles go through without draining the pipeline.  Outside calm tiles every wave runs the whole step
// redundantly (only wave 0 stores), so all four carry the same per-utterance state, take the same
// decisions and meet at the same barriers.
// FAST: calm tiles run the tolerance-mode arithmetic (fast_tile below): the discontinuous per-utterance
// state (Sequencer clock, jitter phase, carrier phase, the LCGs) is advanced exactly as in the exact
// kernels, so no segment boundary, noise wrap or saw edge ever moves; the continuous per-formant
// arithmetic uses fused multiply-adds, one uncorrected reciprocal per formant, and filter
// coefficients interpolated linearly across the tile.  Tiles with an event run the exact steps.
// SPLIT (FAST, one lane per utterance): the time axis of every utterance is cut into chunks
// [split_bounds[k], split_bounds[k + 1]) and each chunk gets a lane of its own, so that a few thousand
// utterances fill the machine.  A chunk's lane first FAST-FORWARDS the exact per-utterance chain (clock,
// segment advances, jitter phase and redraws, pitch track, carrier phase: the reference's operations, no
// filters, nothing stored) from sample 0 to `warmup` samples before its chunk, starts the filters there from
// zero state — by the chunk's first sample the difference to the true state has decayed below half an ulp of
// full scale (DevVoice::warmup, from the narrowest bandwidth of the voice) — and then renders like any fast
// kernel, storing from the chunk's first sample on.  All 64 lanes of a wave work on the same chunk index, so
// they sit at the same sample position and share the carrier noise of a tile as everywhere else.
// MID (FAST kernels): the second tolerance tier, for voices whose resonances are too sharp for interpolated
// coefficients.  What makes a sharp band-pass drift away from the reference is not the size of a coefficient error but
// its PERSISTENCE: a1, a2 = g a1, a3 = g a2 (:560-562) that differ from the reference's rounded values in the same
// direction for the length of a sub-tile move the resonance for that long, and its ring time integrates it
// (profiles/r03_sharpness.txt).  MID evaluates exactly those — the blended and jittered formant frequency and
// bandwidth, g = tan_approx, k = bw / freq, a1, a2, a3 — at every sample with the reference's own operation sequence
// (the same bits as the exact kernels) and keeps the fast arithmetic for everything else: amplitudes, turbulence,
// breath and the low-pass factor interpolated, fused multiply-adds in the filter updates, v_rcp in the polyBLEP, the
// sum in tree order.  Measured on 3 000 random voice tables (oracle model, profiles/r04_middle_tier.txt): at most
// 16 * 2^-23 from the reference at ANY sharpness, where the interpolating tier reaches 95 and plain double precision 150.
template <int L, int T, int WAVES, int MIN_WAVES_PER_SIMD, bool STREAM, bool HALF, bool ANYBL, int NFA = NF,
          bool PIPE = false, bool FAST = false, int PQP = 2, bool SPLIT = false, bool MID = false>
__global__ __launch_bounds__(64 * WAVES, MIN_WAVES_PER_SIMD) void synth_kernel(const SynthArgs A)
{
#include "synth_kernel_state.h"

#include "synth_kernel_general_step.h"

#include "synth_kernel_calm_steps.h"

#include "synth_kernel_pipe.h"

#include "synth_kernel_scalar_packed.h"


#include "synth_kernel_fast.h"
#include "synth_kernel_flush.h"

#include "synth_kernel_fast_tile.h"

#include "synth_kernel_split.h"

#include "synth_kernel_tile_loop.h"

    if constexpr (SPLIT) {
        // the utterance's length comes from the lane that saw it end — the chain returned None, or the row was
        // full — inside its own chunk: a lane that stopped at the next chunk's first sample has only paused, and
        // an utterance that ended before this lane's chunk began belongs to an earlier lane
        if (slot_used && done && !paused && n_out >= chunk_lo) {
            if (A.out_len) A.out_len[u] = n_out;
            if (truncated) atomicOr(A.truncated, 1u);
        }
    } else if (emit && j == L - 1 && slot_used) {
        if (A.out_len) A.out_len[u] = n_out;
        if (truncated) atomicOr(A.truncated, 1u);
    }
    if (streaming && A.state && slot_used && emit) {
        StateIO<false> io{A.state, A.state_stride, state_lane};
        visit_state(io);
        if constexpr (LIVE)
            if (A.ring_cap != 0u && j == L - 1 && A.seg_consumed) A.seg_consumed[u] = seg_pos;
    }
    if (emit && lane == 0 && slow_steps) atomicAdd(A.truncated + 1, slow_steps);
    if (emit && lane == 0 && fast_tiles) atomicAdd(A.truncated + 2, fast_tiles);
    if (emit && lane == 0 && general_steps) atomicAdd(A.truncated + 3, general_steps);
#ifdef GRAIL_FAST_PROF
    {
        unsigned long long *prof = reinterpret_cast<unsigned long long *>(A.truncated + 8);
        prof_c[0] = clock64() - prof_start;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (prof_c[k]) atomicAdd(prof + k, prof_c[k]);
            atomicAdd(prof + 31, 1ull);      // waves
        }
        if (prof_lane_levels) atomicAdd(prof + 17, prof_lane_levels);
    }
#endif
}

template <int L, int T, int WAVES, int MINW, bool STREAM, bool HALF, bool ANYBL, int NFA = NF, bool PIPE = false,
          bool FAST = false, int PQP = 2, bool SPLIT = false, bool MID = false>
void start(const SynthArgs &args, dim3 grid, dim3 block, hipStream_t stream)
{
    std::snprintf(g_kernel_name, sizeof g_kernel_name, "synth_kernel<L=%d,T=%d,W=%d,%d,%s%s%sNFA=%d%s%s%s%s%s>", L, T, WAVES,
                  MINW, STREAM ? "STREAM," : "", HALF ? "HALF," : "", ANYBL ? "ANYBL," : "", NFA,
                  PIPE ? ",PIPE" : "", FAST ? ",FAST" : "", PQP == 4 ? ",R16" : PQP == 8 ? ",R32" : "", SPLIT ? ",SPLIT" : "",
                  MID ? ",MID" : "");
    hipLaunchKernelGGL((synth_kernel<L, T, WAVES, MINW, STREAM, HALF, ANYBL, NFA, PIPE, FAST, PQP, SPLIT, MID>), grid, block, 0,
                       stream, args);
}

}  // namespace
}  // namespace grail
