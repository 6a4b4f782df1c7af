// synth_kernels.hip — the fused Selector -> Sequencer -> Jitter -> Synthesize kernel
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

// min(x, x of the lane the DPP control names); lanes without a source keep their own
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t umin_dpp(const uint32_t x)
{
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xF, false);
    return o < x ? o : x;
}
// lane i takes lane i-1's value (within its row of 16 lanes)
__device__ __forceinline__ float dpp_from_lane_below(float x)
{
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111 /* row_shr:1 */, 0xF, 0xF, true));
}

// LDS hand-off between lanes of ONE wave: same-wave DS operations execute in
// order, so only compiler reordering has to be fenced.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the slice of a SynthesisElem that one lane owns: NV vectors of W formants
template <int NV, typename V>
struct Part {
    float frequency;
    V freq[NV], bw[NV], smooth[NV], breath[NV], turb[NV], amp[NV];
};

template <int NV, int W, typename V>
__device__ __forceinline__ void load_part(Part<NV, V> &p, const float *__restrict__ elems,
                                          int row, int f0)
{
    const float *e = elems + (size_t)row * ELEM_FLOATS + f0;
    p.frequency = elems[(size_t)row * ELEM_FLOATS];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const int i = k * W + c;
            vset(p.freq[k], c, e[F_FREQ + i]);
            vset(p.bw[k], c, e[F_BW + i]);
            vset(p.smooth[k], c, e[F_SMOOTH + i]);
            vset(p.breath[k], c, e[F_BREATH + i]);
            vset(p.turb[k], c, e[F_TURB + i]);
            vset(p.amp[k], c, e[F_AMP + i]);
        }
    }
}

// SynthesisElem::silent(), src/lib.rs:367-377
template <int NV, typename V>
__device__ __forceinline__ void silent_part(Part<NV, V> &p)
{
    p.frequency = 0.25f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        p.freq[k] = vsplat(0.25f, p.freq[k]);
        p.bw[k] = vsplat(0.25f, p.bw[k]);
        p.smooth[k] = vsplat(0.25f, p.smooth[k]);
        p.breath[k] = vsplat(0.0f, p.breath[k]);
        p.turb[k] = vsplat(0.0f, p.turb[k]);
        p.amp[k] = vsplat(0.0f, p.amp[k]);
    }
}

// The parallel formant filters of Synthesize::next, src/lib.rs:531-571, for the NV
// formant vectors one lane owns.  SAFE selects the division flavour (same bits).
// Written breadth-first (each step for every k before the next step) so that the NV
// independent dependency chains interleave and hide each other's VALU latency.
//
// NLIVE < NV (quiet step only): vectors k >= NLIVE are "silent" for the whole segment pair —
// amplitude exactly +0 in both blended elems and band-pass state exactly +0 (see
// upper_half_is_silent) — so their v0 is +-0, their band-pass output w1 is exactly +0 and the
// state stays +0 (a1*(+0) + a2*(+-0) = +0, (0 + a2*0) + a3*(+-0) = +0, 2*0 - 0 = +0).  Only
// their one-pole low-pass state (:538) still has to advance; v1 = +0 is returned for the fold.
#define FOR_K _Pragma("unroll") for (int k = 0; k < NV; ++k)
#define FOR_L _Pragma("unroll") for (int k = 0; k < NLIVE; ++k)
// SU = true (quiet step only): the blended smoothness is the same number for all of the lane's
// formants (bit-equal table entries), so 1 - exp_approx(smooth) was evaluated once, as a scalar,
// by the caller (`oml_s`): the same operations on the same operands give the same bits.
// KEEP_LP = false (one-shot kernels, NLIVE < NV): the silent formants can never become audible in
// this launch (their amplitude is 0 in every phoneme of the voice table), so even their low-pass
// state is dead and is not advanced.  Resumable streams keep it (KEEP_LP = true).
template <bool SAFE, int NV, int NLIVE, bool SU, bool KEEP_LP, typename V>
__device__ __forceinline__ void formant_filters(const float saw, const float noise, const float oml_s,
                                                const V (&e_freq)[NV], const V (&e_bw)[NV],
                                                const V (&e_smooth)[NV], const V (&e_breath)[NV],
                                                const V (&e_turb)[NV], const V (&e_amp)[NV],
                                                V (&st_a)[NV], V (&st_b)[NV], V (&st_c)[NV],
                                                V (&v1)[NV])
{
    if constexpr (!SAFE) {
        static_assert(NLIVE == NV, "the IEEE flavour always runs every formant");
        // the rare IEEE-division flavour, one formant vector at a time (fewest live registers)
        FOR_K {
            const V nw = saw * (1.0f - e_breath[k]) + noise * e_breath[k];      // :531
            const V lp = exp_approx(e_smooth[k]);                               // :535
            st_a[k] = st_a[k] + (1.0f - lp) * (nw - st_a[k]);                   // :538
            const V tw = st_a[k] * ((1.0f - e_turb[k]) + noise * e_turb[k]);    // :544-545
            const V v0 = tw * e_amp[k];                                         // :550
            const V g = tan_approx<false>(e_freq[k]);                           // :555
            const V kq = e_bw[k] / e_freq[k];                                   // :558
            const V a1 = vsplat(1.0f, g) / (1.0f + g * (g + kq));               // :560
            const V a2 = g * a1;                                                // :561
            const V a3 = g * a2;                                                // :562
            const V v3 = v0 - st_c[k];                                          // :565
            const V w1 = a1 * st_b[k] + a2 * v3;                                // :566
            const V w2 = (st_c[k] + a2 * st_b[k]) + a3 * v3;                    // :567
            st_b[k] = 2.0f * w1 - st_b[k];                                      // :570
            st_c[k] = 2.0f * w2 - st_c[k];                                      // :571
            v1[k] = w1;
        }
        return;
    } else {
        V num[NLIVE], den[NLIVE], g[NLIVE], kq[NLIVE], a1[NLIVE], y[NLIVE], e[NLIVE], q[NLIVE],
            r[NLIVE], d3[NLIVE], y2[NLIVE], e2[NLIVE], q2[NLIVE], r2[NLIVE];
        const V one = vsplat(1.0f, st_a[0]);
        const V five = vsplat(5.0f, st_a[0]);
        const V m4 = vsplat(-4.0f, st_a[0]);
        // tan_approx numerator / denominator, src/lib.rs:63-70.  In the SAFE operand window
        // (4*a)*b == 4*(a*b) exactly (scaling by 4 commutes with rounding, nothing under- or
        // overflows), so 5 - (4*a)*b == fma(-4, a*b, 5): one rounding of the same real number.
        FOR_L {
            const V x = e_freq[k];
            const V omx = 1.0f - x;
            const V xph = x + 0.5f;
            const V hmx = 0.5f - x;
            const V ox = omx * x;                       // (1-x)*x, shared by both polynomials
            const V ph = xph * hmx;
            num[k] = ox * vfma(m4, ph, five);           // ((1-x)*x) * (5 - (4*(x+.5))*(.5-x))
            den[k] = (xph * vfma(m4, ox, five)) * hmx;  // ((x+.5) * (5 - (4*(1-x))*x)) * (.5-x)
        }
        // g = num/den and kq = bw/freq by div_exact<true>, a1 = 1/d3 by rcp_exact<true>,
        // spelled out step by step across k
        FOR_L { y[k] = vrcp(den[k]); y2[k] = vrcp(e_freq[k]); }
        FOR_L { e[k] = vfma(-den[k], y[k], one); e2[k] = vfma(-e_freq[k], y2[k], one); }
        FOR_L { y[k] = vfma(e[k], y[k], y[k]); y2[k] = vfma(e2[k], y2[k], y2[k]); }
        FOR_L { q[k] = num[k] * y[k]; q2[k] = e_bw[k] * y2[k]; }
        FOR_L { r[k] = vfma(-den[k], q[k], num[k]); r2[k] = vfma(-e_freq[k], q2[k], e_bw[k]); }
        FOR_L { g[k] = vfma(r[k], y[k], q[k]); kq[k] = vfma(r2[k], y2[k], q2[k]); }   // :555, :558
        FOR_L d3[k] = 1.0f + g[k] * (g[k] + kq[k]);                                   // :560
        FOR_L y[k] = vrcp(d3[k]);
        FOR_L e[k] = vfma(-d3[k], y[k], one);
        FOR_L a1[k] = vfma(e[k], y[k], y[k]);
        constexpr int NLP = (NLIVE < NV && !KEEP_LP) ? NLIVE : NV;   // low-pass states to advance
#define FOR_P _Pragma("unroll") for (int k = 0; k < NLP; ++k)
        V nw[NV];
        FOR_P nw[k] = saw * (1.0f - e_breath[k]) + noise * e_breath[k];                   // :531
        if constexpr (SU) {
            FOR_P st_a[k] = st_a[k] + oml_s * (nw[k] - st_a[k]);                          // :535-538
        } else {
            V lp[NV];
            FOR_P lp[k] = exp_approx(e_smooth[k]);                                        // :535
            FOR_P st_a[k] = st_a[k] + (1.0f - lp[k]) * (nw[k] - st_a[k]);                 // :538
        }
#undef FOR_P
        V tw[NLIVE], v0[NLIVE], a2[NLIVE], a3[NLIVE], v3[NLIVE], w1[NLIVE], w2[NLIVE];
        // :544-545  1.0*(1-turb) + noise*turb; the multiply by 1.0 is exact and dropped
        FOR_L tw[k] = st_a[k] * ((1.0f - e_turb[k]) + noise * e_turb[k]);
        FOR_L v0[k] = tw[k] * e_amp[k];                                                   // :550
        FOR_L a2[k] = g[k] * a1[k];                                                       // :561
        FOR_L a3[k] = g[k] * a2[k];                                                       // :562
        FOR_L v3[k] = v0[k] - st_c[k];                                                    // :565
        FOR_L w1[k] = a1[k] * st_b[k] + a2[k] * v3[k];                                    // :566
        FOR_L w2[k] = (st_c[k] + a2[k] * st_b[k]) + a3[k] * v3[k];                        // :567
        FOR_L st_b[k] = 2.0f * w1[k] - st_b[k];                                           // :570
        FOR_L st_c[k] = 2.0f * w2[k] - st_c[k];                                           // :571
        FOR_L v1[k] = w1[k];
#pragma unroll
        for (int k = NLIVE; k < NV; ++k) v1[k] = vsplat(0.0f, st_a[0]);                   // exactly +0
    }
}
#undef FOR_L
#undef FOR_K

// MID kernels: the band-pass coefficients of one sample for the NV formant vectors of a lane — the reference's own
// operation sequence on its own operands (blend :404-414, jitter :305 / :764, tan_approx :555, bw / freq :558,
// a1, a2, a3 :560-562), i.e. the bits the exact kernels compute — written breadth-first across the vectors like
// formant_filters above: the independent chains hide each other's latency and no v_rcp result is consumed by the next
// instruction (each such pair costs a wait state, and a lone wave pays for every issue slot).
template <int NV, typename V>
__device__ __forceinline__ void exact_band_pass_coeffs(const V (&xf)[NV], const V (&yf)[NV], const V (&xb)[NV], const V (&yb)[NV],
                                                       const V (&ffc)[NV], const V (&ffn)[NV], const float alpha, const float oma,
                                                       const float jp, const float jomp, const float d_ffreq,
                                                       V (&a1)[NV], V (&a2)[NV], V (&a3)[NV])
{
#define FOR_K _Pragma("unroll") for (int k = 0; k < NV; ++k)
    V ef[NV], eb[NV], nff[NV], num[NV], den[NV], g[NV], kq[NV], y[NV], e[NV], q[NV], r[NV], d3[NV], y2[NV], e2[NV], q2[NV], r2[NV];
    const V one = vsplat(1.0f, xf[0]), five = vsplat(5.0f, xf[0]), m4 = vsplat(-4.0f, xf[0]);
    FOR_K ef[k] = xf[k] * oma + yf[k] * alpha;                        // :404-414
    FOR_K eb[k] = xb[k] * oma + yb[k] * alpha;
    FOR_K nff[k] = ffc[k] * jomp + ffn[k] * jp;                       // :305
    FOR_K ef[k] = ef[k] + nff[k] * d_ffreq;                           // :764
    FOR_K {
        const V x = ef[k];
        const V omx = 1.0f - x, xph = x + 0.5f, hmx = 0.5f - x;
        const V ox = omx * x, ph = xph * hmx;
        num[k] = ox * vfma(m4, ph, five);                             // see formant_filters: one rounding of the same number
        den[k] = (xph * vfma(m4, ox, five)) * hmx;
    }
    FOR_K { y[k] = vrcp(den[k]); y2[k] = vrcp(ef[k]); }
    FOR_K { e[k] = vfma(-den[k], y[k], one); e2[k] = vfma(-ef[k], y2[k], one); }
    FOR_K { y[k] = vfma(e[k], y[k], y[k]); y2[k] = vfma(e2[k], y2[k], y2[k]); }
    FOR_K { q[k] = num[k] * y[k]; q2[k] = eb[k] * y2[k]; }
    FOR_K { r[k] = vfma(-den[k], q[k], num[k]); r2[k] = vfma(-ef[k], q2[k], eb[k]); }
    FOR_K { g[k] = vfma(r[k], y[k], q[k]); kq[k] = vfma(r2[k], y2[k], q2[k]); }   // :555, :558
    FOR_K d3[k] = 1.0f + g[k] * (g[k] + kq[k]);                                   // :560
    FOR_K y[k] = vrcp(d3[k]);
    FOR_K e[k] = vfma(-d3[k], y[k], one);
    FOR_K a1[k] = vfma(e[k], y[k], y[k]);
    FOR_K a2[k] = g[k] * a1[k];                                                   // :561
    FOR_K a3[k] = g[k] * a2[k];                                                   // :562
#undef FOR_K
}

// Can every division of the coming segment pair take the SAFE path?  Bounds every
// divisor/dividend over the pair: alpha in [0,1] (clk >= 0 for the whole pair once it
// is >= 0 at its first sample, blend_length > 0), the jitter noises in [-1,1] (0 <=
// jitter_frequency <= 1 keeps the noise phase in (0,1]), so that
//   x = formant_freq  in [2^-20, 1/2 - 2^-20]  =>  tan_approx num in [2^-18, 1.25], den in [2^-19, 5]
//   w = formant_bw    in [2^-40, 2^10]         =>  w/x in [2^-39, 2^30],  1+g(g+w/x) in [1, 2^52]
// all inside the proven [2^-60, 2^60] window.  Any NaN fails a comparison => false.
template <int NV, int W, typename V>
__device__ __forceinline__ bool pair_is_safe(const Part<NV, V> &X, const Part<NV, V> &Y, float clk,
                                             float blend_length, float jinc, float d_ffreq,
                                             float d_freq)
{
    constexpr float X_LO = 9.5367431640625e-07f;        // 2^-20
    constexpr float X_HI = 0.5f - 9.5367431640625e-07f;
    constexpr float W_LO = 1.8189894035458565e-12f;     // 2^-39 (2x margin over 2^-40)
    constexpr float W_HI = 512.0f;                      // 2^9   (2x margin under 2^10)
    const float jm = 1.002f * __builtin_fabsf(d_ffreq);
    // carrier frequency (the polyBLEP divisor, src/lib.rs:505/509): in [2^-20, 1]; the
    // dividend is the phase or phase-1, a sum of such frequencies: 0 or >= 2^-24 in magnitude
    const float jf = 1.002f * __builtin_fabsf(d_freq);
    bool ok = (clk >= 0.0f) && (blend_length > 0.0f) && (jinc >= 0.0f) && (jinc <= 1.0f) &&
              (jm <= 1.0f) && (jf <= 1.0f) &&
              (X.frequency * 0.999f - jf >= X_LO) && (Y.frequency * 0.999f - jf >= X_LO) &&
              (X.frequency * 1.001f + jf <= 1.0f) && (Y.frequency * 1.001f + jf <= 1.0f);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const float xf = vget(X.freq[k], c), yf = vget(Y.freq[k], c);
            const float xb = vget(X.bw[k], c), yb = vget(Y.bw[k], c);
            ok = ok && (xf * 0.999f - jm >= X_LO) && (yf * 0.999f - jm >= X_LO) &&
                 (xf * 1.001f + jm <= X_HI) && (yf * 1.001f + jm <= X_HI) &&
                 (xb >= W_LO) && (yb >= W_LO) && (xb <= W_HI) && (yb <= W_HI);
        }
    }
    return ok;
}

// Is the upper half of this lane's formant vectors silent for the coming segment pair?  Then the
// quiet step may skip their band-pass filters (formant_filters<.., NLIVE = NV/2>) and still be
// bit-identical.  Needs, for every such formant: amplitude exactly +0 in both blended elems and
// 0 <= 0.5*jitter_delta_amplitude <= 1/4 (so the jittered amplitude 0*(1-delta) is +0, delta <=
// 1/2), band-pass state b, c exactly +0, and breath / turbulence / smoothness in [0,1] with a
// finite low-pass state (so tw = a*(..) is finite and v0 = tw*(+0) is +-0, never NaN).  Finite,
// positive a1, a2, a3 and a finite saw come from pair_is_safe.
template <int NV, int W, typename V>
__device__ __forceinline__ bool upper_half_is_silent(const Part<NV, V> &X, const Part<NV, V> &Y,
                                                     const V (&st_a)[NV], const V (&st_b)[NV],
                                                     const V (&st_c)[NV], float amp_scale)
{
    bool ok = (amp_scale >= 0.0f) && (amp_scale <= 0.25f);
#pragma unroll
    for (int k = NV / 2; k < NV; ++k) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const float xb = vget(X.breath[k], c), yb = vget(Y.breath[k], c);
            const float xt = vget(X.turb[k], c), yt = vget(Y.turb[k], c);
            const float xs = vget(X.smooth[k], c), ys = vget(Y.smooth[k], c);
            ok = ok && (__float_as_uint(vget(X.amp[k], c)) == 0u) &&
                 (__float_as_uint(vget(Y.amp[k], c)) == 0u) &&
                 (__float_as_uint(vget(st_b[k], c)) == 0u) &&
                 (__float_as_uint(vget(st_c[k], c)) == 0u) &&
                 (xb >= 0.0f) && (xb <= 1.0f) && (yb >= 0.0f) && (yb <= 1.0f) &&
                 (xt >= 0.0f) && (xt <= 1.0f) && (yt >= 0.0f) && (yt <= 1.0f) &&
                 (xs >= 0.0f) && (xs <= 1.0f) && (ys >= 0.0f) && (ys <= 1.0f) &&
                 (__builtin_fabsf(vget(st_a[k], c)) <= 1.152921504606847e18f);   // 2^60
        }
    }
    return ok;
}

// Resumable synthesis: the per-lane state that the reference keeps in
// its Copy iterator structs (Sequencer :839-854, Jitter :724-748, Synthesize :470-488), moved
// between registers and HBM word by word.  Layout: state[word][global lane], coalesced.
template <bool LOAD>
struct StateIO {
    uint32_t *base;
    size_t stride, lane;
    uint32_t w = 0;
    __device__ __forceinline__ uint32_t &slot() { return base[(size_t)(w++) * stride + lane]; }
    __device__ __forceinline__ void operator()(uint32_t &v) { if (LOAD) v = slot(); else slot() = v; }
    __device__ __forceinline__ void operator()(int &v)
    {
        if (LOAD) v = (int)slot(); else slot() = (uint32_t)v;
    }
    __device__ __forceinline__ void operator()(float &v)
    {
        if (LOAD) v = __uint_as_float(slot()); else slot() = __float_as_uint(v);
    }
    __device__ __forceinline__ void operator()(bool &v)
    {
        if (LOAD) v = slot() != 0u; else slot() = v ? 1u : 0u;
    }
    __device__ __forceinline__ void operator()(f2 &v)
    {
        float a = v.x, b = v.y;
        (*this)(a);
        (*this)(b);
        v.x = a;
        v.y = b;
    }
};

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
    static_assert(!FAST || (!PIPE && !HALF), "FAST");
    static_assert(!MID || FAST, "MID is a flavour of the tolerance kernels");
    static_assert(!SPLIT || (FAST && !STREAM && L == 1 && WAVES == 1 && T == 64), "SPLIT");
    static_assert(NFA == NF || (NFA == 4 && !HALF), "NFA");
    static_assert(!PIPE || (WAVES == 4 && NFA / L == 1 && L >= 4 && !HALF && T % 4 == 0), "PIPE");
    constexpr int FPL = NFA / L;         // formants per lane
    constexpr int W = FPL >= 2 ? 2 : 1;  // formants per packed value
    constexpr int NV = FPL / W;          // packed values per lane and field
    typedef typename VecOf<W>::type V;
    constexpr int S = 64 / L;            // utterances per wave
    constexpr int SP = S + 1;            // padded row of the staging tile
    static_assert(T % 4 == 0 && (64 % (T / 4)) == 0, "T");
    // ONE WAVE PER SIMD, by construction.  Every family is laid out for one resident wave per SIMD (a second wave on
    // a SIMD costs as much as it brings), and the host sizes its launches accordingly — but where the waves of a
    // launch LAND is the dispatcher's business: with kernels that fit a SIMD twice (<= 256 registers) it put two
    // waves on some SIMDs and none on others whenever the launch before had left its round-robin state "odd"
    // (a two-lane launch of 1024 waves: 27 ms after another 1024-wave launch, 48 ms after one of 1536 waves or as
    // the first launch of a process; profiles/r04_dispatch.txt).  A wave that owns more than half of the SIMD's 512
    // registers cannot share it: the one-lane kernels do anyway (256 VGPRs + AGPRs); the others claim accumulation
    // registers they never touch.  (PIPE workgroups are placed by their LDS footprint instead.)
    // TWO WAVES PER SIMD (MIN_WAVES_PER_SIMD = 2; lane kernels on two, four and eight lanes per utterance that hold their
    // state in <= 256 registers without a scratch segment): the lone tolerance-mode wave leaves the VALU idle a quarter of
    // the time, and two of them on a SIMD render 20 - 30 % more per second than one after the other (twice as much where
    // events are dense: a slow sample is latency); the exact kernels gain 9 - 15 % on aligned batches and up to 30 % on
    // speech-like ones — where the waves spill (one lane per utterance) they lose 14 % instead (profiles/r04_two_waves.txt,
    // r05_two_waves.txt).  The host asks for these instantiations only for launches of more waves than the device has
    // SIMDs, where the dispatcher's placement has nothing to get wrong.
    if constexpr (L > 1 && !PIPE && MIN_WAVES_PER_SIMD == 1) asm volatile("" ::: "a127");

    // every wave of the block works alone on its own S utterances and its own
    // slice of LDS: there is no inter-wave communication and no block barrier
    // L >= 4: the lanes park all eight band-pass outputs of a sample and the left fold runs at
    // flush time, spread over time steps, instead of a serial chain of L DPP hops per sample
    constexpr bool FOLD_IN_FLUSH = L >= 4;
    constexpr int STAGE_FLOATS = FOLD_IN_FLUSH ? T * S * NFA : T * SP;
    __shared__ float stage_all[PIPE ? 1 : WAVES][STAGE_FLOATS];
    __shared__ uint32_t cnt_all[PIPE ? 1 : WAVES][S];
    const int wave = threadIdx.x / 64;
    float *stage = stage_all[PIPE ? 0 : wave];
    uint32_t *cnt = cnt_all[PIPE ? 0 : wave];
    // PIPE: role 0 renders (owns stage, counts, output), role 1 carries the per-utterance chain,
    // roles 2 and 3 prepare coefficients; `emit` is constant true otherwise
    const int role = PIPE ? wave : 0;
    const bool emit = !PIPE || role == 0;

    const int lane = threadIdx.x % 64;
    const int slot = lane / L;
    const int j = lane % L;
    const int f0 = j * FPL;
    // SPLIT: the waves of the last chunk (longest fast-forward) start first
    const uint32_t split_groups = SPLIT ? (A.n_utt + S - 1) / S : 1u;
    const uint32_t chunk = SPLIT ? A.split_chunks - 1u - blockIdx.x / split_groups : 0u;
    // FOLD (two waves per SIMD, a launch of at most two rounds of the device): every wave is resident from the start, so
    // nothing evens out the SIMDs' loads afterwards, and the launch slots are filled longest utterances first — the
    // workgroups of the second round take their slots in reverse order, so that the SIMD with the longest rows of the first
    // round gets the shortest of the second
    uint32_t block_id = blockIdx.x;
    if constexpr (!SPLIT && !PIPE && !STREAM && MIN_WAVES_PER_SIMD == 2)
        if (A.fold_from != 0u && block_id >= A.fold_from) block_id = gridDim.x - 1u - (block_id - A.fold_from);
    // PIPE, one-shot: a workgroup may hold fewer utterances than it has slots for (SynthArgs::pipe_fill)
    const uint32_t pipe_fill = PIPE && !STREAM && A.pipe_fill != 0u ? A.pipe_fill : (uint32_t)S;
    const uint32_t u0 = SPLIT ? (blockIdx.x % split_groups) * S
                              : PIPE ? blockIdx.x * pipe_fill : (block_id * WAVES + wave) * S;
    // which utterance this slot renders: its position in the launch, or — ragged batches — the host's
    // length-sorted assignment (A.perm), so that the lanes of a wave end together; rows, lengths and
    // per-utterance inputs always belong to utterance `u`.  (A launch may cover a range of the slots only —
    // A.perm then points at the range's first slot and `u` may well exceed A.n_utt: `slot_used` says whether
    // the slot renders, never a comparison of `u`.)
    const bool slot_used = (!PIPE || (uint32_t)slot < pipe_fill) && u0 + slot < A.n_utt;
    const uint32_t u = !slot_used ? A.n_utt : (A.perm ? A.perm[u0 + slot] : u0 + slot);
    bool done = !slot_used;
    if constexpr (SPLIT && GRAIL_SPLIT_SKIP) {
        // a chunk's lane whose utterance ends before the chunk begins (the host's upper bound of its length) has nothing
        // to render — no fast-forward, no warm-up; a wave of such lanes is gone at once.  Rows that differ in length are
        // launched longest first, so the waves of the later chunks are the ones that go, and the host lays out more,
        // shorter chunks than the device has SIMDs for (launch_plan.cpp).
        if (A.len_bound != nullptr && slot_used && chunk > 0u && A.len_bound[u] <= A.split_bounds[chunk]) done = true;
        if (__builtin_amdgcn_ballot_w64(!done) == 0) {
#ifdef GRAIL_FAST_PROF
            if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(A.truncated + 8) + 30, 1ull);
#endif
            return;
        }
    }
    const uint32_t uc = done ? 0u : u;
    __shared__ uint32_t rowid_all[PIPE ? 1 : WAVES][S];
    uint32_t *rowid = rowid_all[PIPE ? 0 : wave];
    if (A.perm && j == L - 1) rowid[slot] = uc;

    uint32_t vid = A.voice_ids ? A.voice_ids[uc] : 0u;
    if (vid >= A.n_voices) vid = 0u;
    const DevVoice VO = A.voices[vid];
    const bool phoneme_mode = A.phoneme_mode != 0;
    const float *__restrict__ elems = A.elems;

    // ---- Sequencer state: IntoSequencer::sequence, src/lib.rs:941-949
    // live streams (STREAM kernels only): the utterance's segments sit in a ring and more may be appended between
    // launches; seg_pos then counts the segments pulled so far and seg_end those appended so far
    // Only the general resumable instantiations (ANYBL: what a live stream always runs — nothing is known about the
    // segments to come) carry the ring code: the lean ones stay what they were (a few instructions more in the general
    // step moved the code of the calm loops and cost the lean one-lane stream kernel 9 %, same instruction counts).
    // Everything else about the ring is worked out where a segment is pulled — a rare path.
    constexpr bool LIVE = STREAM && ANYBL;
    uint32_t seg_pos = (LIVE && A.ring_cap != 0u) ? 0u : A.seg_offsets[uc];
    const uint32_t seg_end = (LIVE && A.ring_cap != 0u) ? A.seg_counts[uc] : A.seg_offsets[uc + 1];
    Seg cur, nxt;
    cur.some = false; cur.elem = -1; cur.length = 0.0f; cur.blend_length = 1.0f; cur.frequency = 0.0f;
    nxt = cur;
    float clk = 0.0f;                        // Sequencer.time
    const float dt = 1.0f / VO.sample_rate;  // :944
    Part<NV, V> X, Y;                        // emitted elem = X*(1-alpha) + Y*alpha
    silent_part(X);
    silent_part(Y);
    float blend_length = 1.0f;
    float inv_blend_length = 1.0f;           // exact when blend_length is +-2^k
    bool blend_pow2 = true;
    bool blend_div_ok = false;               // ANYBL: clk / blend_length may use the short exact division
    bool silent_pair = true;
    bool pair_safe = false;                  // every division of this pair may use div_exact<true>

    // ---- Jitter state: IntoJitter::jitter, src/lib.rs:786-797.  One seed is
    // threaded through the three constructors (2 + 16 + 16 draws), each noise
    // then keeps its own copy of the state.  The three noises share one phase
    // sequence (same start, same increment), kept once.
    uint32_t seed = A.seeds ? A.seeds[uc] : 0u;
    // a resumed stream call loads all of this from its state block: skip the 34 draws
    const bool fresh_start = !(STREAM && A.state && A.resume);
    float fn_cur = 0.0f, fn_next = 0.0f;
    if (fresh_start) {
        fn_cur = lcg_f32(seed);              // ValueNoise::new :228-229
        fn_next = lcg_f32(seed);
    }
    uint32_t fn_state = seed;
    V ff_cur[NV], ff_next[NV], fa_cur[NV], fa_next[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        ff_cur[k] = vsplat(0.0f, ff_cur[k]); ff_next[k] = ff_cur[k];
        fa_cur[k] = ff_cur[k]; fa_next[k] = ff_cur[k];
    }
    uint32_t ff_state = seed;
    if (fresh_start) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {           // ArrayValueNoise::new :275-278
        const float c0 = lcg_f32(seed);
        const float n0 = lcg_f32(seed);
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (i == f0 + k * W + c) { vset(ff_cur[k], c, c0); vset(ff_next[k], c, n0); }
    }
    ff_state = seed;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const float c0 = lcg_f32(seed);
        const float n0 = lcg_f32(seed);
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (i == f0 + k * W + c) { vset(fa_cur[k], c, c0); vset(fa_next[k], c, n0); }
    }
    }
    uint32_t fa_state = seed;
    float jphase = 0.0f;
    const float jinc = VO.jitter_frequency;
    const float d_freq = VO.jitter_delta_frequency;
    const float d_ffreq = VO.jitter_delta_formant_frequency;
    const float amp_scale = 0.5f * VO.jitter_delta_amplitude;   // :769

    // ---- Synthesize state: IntoSynthesize::synthesize, src/lib.rs:587-596
    float phase = 0.0f;
    V st_a[NV], st_b[NV], st_c[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        st_a[k] = vsplat(0.0f, st_a[k]);
        st_b[k] = st_a[k];
        st_c[k] = st_a[k];
    }
    uint32_t noise_seed = 0u;                // :594

    const uint64_t cap = A.cap;              // samples this launch may write per row (<= out_stride)
    const uint32_t cap32 = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;   // n_out is 32-bit
    // where this launch stops rendering an utterance that has not ended: a stream call at its quota, a chunk
    // lane at the first sample of the next chunk (the last chunk runs to the end of the row)
    constexpr bool PAUSES = STREAM || SPLIT;
    const uint32_t chunk_lo = SPLIT ? A.split_bounds[chunk] : 0u;
    const uint32_t pause_at = SPLIT ? (chunk + 1u < A.split_chunks ? A.split_bounds[chunk + 1u] : 0xFFFFFFFFu) : cap32;
    const uint32_t room_end = SPLIT ? (pause_at < cap32 ? pause_at : cap32) : cap32;
    bool paused = false;
    uint32_t n_out = 0;
    uint32_t slow_steps = 0;                 // wave-steps that took the IEEE-division body
    uint32_t fast_tiles = 0, general_steps = 0;   // statistics: tiles rendered by fast_tile, general steps taken
#ifdef GRAIL_FAST_PROF
    unsigned long long prof_c[32] = {};
    unsigned long long prof_t0 = clock64();
    const unsigned long long prof_start = prof_t0;
    unsigned long long prof_lane_levels = 0;
#endif
    bool truncated = false;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(A.out) & 15u) == 0) && ((A.out_stride & 3u) == 0);
    const bool vec16_ok = ((reinterpret_cast<uintptr_t>(A.out_pcm16) & 7u) == 0) && ((A.out_stride & 3u) == 0);

    // false while the lane's segment pair needs the IEEE-division body or has a blend
    // length that is not a power of two: such lanes always take the general step
    bool quiet_ok = false;

    // the chain has returned None (persistent; `done` also covers pauses).  A slot without an utterance counts as finished:
    // it will not render in this launch or any other, so it rides along in calm tiles and runs like an ended utterance (a
    // lone stream in a workgroup laid out for sixteen used to keep its wave out of every calm tile)
    bool finished = !slot_used;
    // one-shot batches: the lane's upper formants have amplitude +0 in every phoneme of the voice table (phoneme
    // batches: looked up here) or in every elem of the batch (caller-built elems: formants 5-8, established by the
    // host at upload — half_capable), so nothing in this launch can ever make them audible
    bool upper_never_live = false;
    if constexpr (HALF && NV >= 2 && !STREAM) {
        if (phoneme_mode) {
            upper_never_live = true;
#pragma unroll
            for (int p = 0; p < NUM_VOICED; ++p)
#pragma unroll
                for (int i = (NV / 2) * W; i < NV * W; ++i)
                    upper_never_live = upper_never_live &&
                        (__float_as_uint(elems[(size_t)(VO.elem_base + p) * ELEM_FLOATS + F_AMP + f0 + i]) == 0u);
        } else {
            upper_never_live = A.half_capable != 0u && f0 + (NV / 2) * W >= NF / 2;
        }
    }
    bool smooth_uniform = false; // this pair: X.smooth and Y.smooth are each one number for all formants
    bool upper_silent = false;   // this pair: the lane's upper NV/2 formant vectors are silent
    auto update_silent = [&]() __attribute__((always_inline)) {
        if constexpr (HALF && NV >= 2)
            upper_silent = A.skip_silent && pair_safe && (STREAM || upper_never_live) &&
                           upper_half_is_silent<NV, W>(X, Y, st_a, st_b, st_c, amp_scale);
        bool su = true;
        const uint32_t xs0 = __float_as_uint(vget(X.smooth[0], 0));
        const uint32_t ys0 = __float_as_uint(vget(Y.smooth[0], 0));
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int c = 0; c < W; ++c)
                su = su && (__float_as_uint(vget(X.smooth[k], c)) == xs0) &&
                     (__float_as_uint(vget(Y.smooth[k], c)) == ys0);
        smooth_uniform = su;
    };

    // (cur, nxt) -> X, Y, blend constants: the match of Sequencer::next resolved once per pair
    auto setup_pair = [&]() __attribute__((always_inline)) {
        // the match at :891-931, resolved once per segment pair
        const bool has_b = cur.elem >= 0;
        const bool has_c = nxt.some && nxt.elem >= 0;
        blend_length = cur.blend_length;
        silent_pair = !has_b && !has_c;
        if (has_b && has_c) {          // c.blend(b, alpha)  :897-903
            load_part<NV, W>(X, elems, nxt.elem, f0);
            load_part<NV, W>(Y, elems, cur.elem, f0);
            X.frequency = nxt.frequency;
            Y.frequency = cur.frequency;
        } else if (has_b) {            // b.copy_silent().blend(b, alpha)  :906-912
            load_part<NV, W>(Y, elems, cur.elem, f0);
            Y.frequency = cur.frequency;
            X = Y;
#pragma unroll
            for (int k = 0; k < NV; ++k) X.amp[k] = vsplat(0.0f, X.amp[k]);
        } else if (has_c) {            // c.blend(c.copy_silent(), alpha)  :915-921
            load_part<NV, W>(X, elems, nxt.elem, f0);
            X.frequency = nxt.frequency;
            Y = X;
#pragma unroll
            for (int k = 0; k < NV; ++k) Y.amp[k] = vsplat(0.0f, Y.amp[k]);
        } else {                       // SynthesisElem::silent()  :924-927
            silent_part(X);
            silent_part(Y);
        }
        // clk / 2^k == clk * 2^-k for every clk (same real number, same rounding)
        const uint32_t blb = __float_as_uint(blend_length);
        const uint32_t ble = (blb >> 23) & 0xFFu;
        blend_pow2 = ((blb & 0x7FFFFFu) == 0u) && ble >= 1u && ble <= 253u;
        inv_blend_length = 1.0f / blend_length;       // IEEE: RN(1/b), what div_exact<true> starts from
        // any other blend length: q = clk*RN(1/b), r = fma(-b, q, clk), q' = fma(r, RN(1/b), q) is the
        // correctly rounded clk/b while b and clk are in the proven window (tools/div_exhaustive.hip);
        // clk <= length, and steps whose clk is below the window take the general step
        if constexpr (ANYBL)
            blend_div_ok = (blend_length >= 0x1p-59f) && (blend_length <= 0x1p59f) &&
                           (cur.length <= 0x1p59f) && (dt >= 0x1p-59f);
    };

    constexpr bool streaming = STREAM;       // a separate instantiation: the one-shot kernel
                                             // carries none of the state traffic or its registers
    // (PIPE: the four waves of a workgroup carry ONE set of utterances — every wave loads the set's state, the rendering
    // wave, whose filters are the live ones, saves it; the block is the lane kernels' of the same L, utterance by utterance:
    // a stream may take either from call to call)
    const size_t state_lane = PIPE ? (size_t)blockIdx.x * 64 + lane : (size_t)(blockIdx.x * WAVES + wave) * 64 + lane;
    auto visit_state = [&](auto &io) __attribute__((always_inline)) {
        io(seg_pos);
        io(cur.some); io(cur.elem); io(cur.length); io(cur.blend_length); io(cur.frequency);
        io(nxt.some); io(nxt.elem); io(nxt.length); io(nxt.blend_length); io(nxt.frequency);
        io(clk); io(pair_safe); io(finished);
        io(fn_cur); io(fn_next); io(fn_state); io(ff_state); io(fa_state); io(jphase);
        io(phase); io(noise_seed);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            io(ff_cur[k]); io(ff_next[k]); io(fa_cur[k]); io(fa_next[k]);
            io(st_a[k]); io(st_b[k]); io(st_c[k]);
        }
    };
    if (streaming && A.state && A.resume && slot_used) {
        StateIO<true> io{A.state, A.state_stride, state_lane};
        visit_state(io);
        done = finished;
        if (cur.some) setup_pair();
        quiet_ok = pair_safe && (blend_pow2 || blend_div_ok);
        update_silent();
    }

    // ---- the general sample step: any lane may be finished, advance a segment, wrap its
    // jitter noise, hit the row capacity, or need the IEEE-division body.
    // CHAIN_ONLY (SPLIT's fast-forward): the per-utterance chain alone — Sequencer, Jitter state, pitch, carrier
    // phase — exactly as below; no formant is evaluated and nothing is staged
    // FAST kernels take the step apart in the slow samples of a mixed tile (fast_slow_sample): mode 2 is the chain part —
    // Sequencer, Jitter state, pitch, carrier phase, the carrier noise, the sample counted — which leaves what the
    // formants need in cv_*; mode 3 is the formant part of the same sample from those values, for a lane whose new
    // segment pair turned out to lie outside the safe window (every other lane goes on in the shared tolerance-mode body).
    float cv_alpha = 1.0f, cv_freq = 0.0f, cv_ph = 0.0f, cv_noise = 0.0f;
    // (an int, not a bool: with a second bool stored `true` next to `done = true` the optimiser merges the two stores into
    // one through a pointer it selects — and both variables live in scratch memory from then on)
    [[maybe_unused]] int cv_live = 0;          // mode 2 rendered a sample (the lane did not end, pause or fill its row in this step)
    auto general_step = [&](const int t, auto chain_only_tag) __attribute__((always_inline)) {
        constexpr int MODE = (int)decltype(chain_only_tag)::value;   // 0: the whole step, 1: CHAIN_ONLY, 2: chain part, 3: formant part
        constexpr bool CHAIN_ONLY = MODE == 1;
        float alpha, oma, frequency;
        if constexpr (MODE == 2) cv_live = 0;
        if constexpr (MODE != 3) {
        if (done) return;
        if (PAUSES && n_out >= pause_at) {   // this launch's share is used up: pause BEFORE advancing
            done = true;
            paused = true;
            return;
        }

        // ================= Sequencer::next, src/lib.rs:859-932
        if constexpr (LIVE) {
            // a live stream whose source has not delivered yet: this step would pull iter.next() (:870, :877-878) and
            // the segment is not in the ring — wait for it (nothing has been touched: the step is taken again, from
            // the same state, by the launch that follows the append).  The source ends only when the host says so.
            if (A.ring_cap != 0u && (clk - dt) < 0.0f) {
                const uint32_t want = (cur.some && nxt.some) ? 1u : (!cur.some && !nxt.some) ? 2u : 0u;
                if (seg_end - seg_pos < want && A.seg_open[uc] != 0u) {
                    done = true;
                    paused = true;
                    return;
                }
            }
        }
        // where segment `pos` of this utterance sits: in its ring (live streams), or at segs[pos]
        const bool ring = LIVE && A.ring_cap != 0u;
        const uint32_t ring_base = ring ? uc * A.ring_cap : 0u;
        const uint32_t ring_mask = ring ? A.ring_cap - 1u : 0xFFFFFFFFu;
        clk -= dt;                                            // :861
        if (__builtin_expect(clk < 0.0f, 0)) {                // :864
            if (cur.some && nxt.some) {                       // :868
                cur = nxt;
                fetch_seg(nxt, A.segs, seg_pos, seg_end, phoneme_mode, VO.elem_base, ring_base, ring_mask);
                clk += cur.length;                            // :873
            } else if (!cur.some && !nxt.some) {              // :876
                fetch_seg(cur, A.segs, seg_pos, seg_end, phoneme_mode, VO.elem_base, ring_base, ring_mask);
                fetch_seg(nxt, A.segs, seg_pos, seg_end, phoneme_mode, VO.elem_base, ring_base, ring_mask);
                if (cur.some) clk += cur.length;              // :881-883
            } else {
                done = true;                                  // :886
                finished = true;
            }
            if (!done && cur.some) {
                setup_pair();
                pair_safe = pair_is_safe<NV, W>(X, Y, clk, blend_length, jinc, d_ffreq, d_freq);
                update_silent();
                quiet_ok = pair_safe && (blend_pow2 || blend_div_ok);
            }
        }
        if (!cur.some) { done = true; finished = true; }      // :930
        if (done) return;
        if (__builtin_expect(n_out >= cap, 0)) {   // the chain would yield another sample: row is full
            truncated = true;
            done = true;
            return;
        }

        // alpha = (time / blend_length).min(1.0)  :899/:908/:917.  A both-silent
        // pair emits silent() itself (:926): alpha = 1 selects Y = silent() exactly
        // (X*0 + Y*1 with finite X).
        float ratio;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!blend_pow2) == 0, 1))
            ratio = clk * inv_blend_length;
        else
            ratio = blend_pow2 ? clk * inv_blend_length : clk / blend_length;
        alpha = __builtin_fminf(ratio, 1.0f);
        alpha = silent_pair ? 1.0f : alpha;
        oma = 1.0f - alpha;

        // SynthesisElem::blend, src/lib.rs:404-414
        frequency = X.frequency * oma + Y.frequency * alpha;
        } else {
            alpha = cv_alpha;
            oma = 1.0f - alpha;
            frequency = cv_freq;
        }
        V e_freq[NV], e_bw[NV], e_smooth[NV], e_breath[NV], e_turb[NV], e_amp[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            e_freq[k] = X.freq[k] * oma + Y.freq[k] * alpha;
            e_smooth[k] = X.smooth[k] * oma + Y.smooth[k] * alpha;
            e_bw[k] = X.bw[k] * oma + Y.bw[k] * alpha;
            e_turb[k] = X.turb[k] * oma + Y.turb[k] * alpha;
            e_breath[k] = X.breath[k] * oma + Y.breath[k] * alpha;
            e_amp[k] = X.amp[k] * oma + Y.amp[k] * alpha;
        }

        // ================= Jitter::next, src/lib.rs:753-777
        if constexpr (MODE != 3) {
        jphase += jinc;                                       // :242 / :291
        if (__builtin_expect(jphase > 1.0f, 0)) {             // :245 / :294
            jphase -= 1.0f;
            fn_cur = fn_next;                                 // :249-250
            fn_next = lcg_f32(fn_state);
            uint32_t s1 = ff_state, s2 = fa_state;
#pragma unroll
            for (int k = 0; k < NV; ++k) { ff_cur[k] = ff_next[k]; fa_cur[k] = fa_next[k]; }
#pragma unroll
            for (int i = 0; i < NF; ++i) {                    // from_func order :301
                const float r1 = lcg_f32(s1);
                const float r2 = lcg_f32(s2);
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int c = 0; c < W; ++c)
                        if (i == f0 + k * W + c) { vset(ff_next[k], c, r1); vset(fa_next[k], c, r2); }
            }
            ff_state = s1;
            fa_state = s2;
        }
        }
        const float jomp = 1.0f - jphase;
        if constexpr (MODE != 3) {
        const float n_freq = fn_cur * jomp + fn_next * jphase;         // :254
        frequency = frequency + n_freq * d_freq;                       // :763
        }
        if constexpr (CHAIN_ONLY) {
            phase += frequency;                                        // :520
            if (phase >= 1.0f) phase -= 1.0f;                          // :523-525
            ++n_out;                                                   // (the carrier noise state follows from n_out)
            return;
        }
        if constexpr (MODE == 2) {
            cv_alpha = alpha;
            cv_freq = frequency;
            cv_ph = phase;
            phase += frequency;                                        // :520
            if (phase >= 1.0f) phase -= 1.0f;                          // :523-525
            cv_noise = lcg_f32(noise_seed);                            // :528
            ++n_out;
            cv_live = 1;
            return;
        }
        const float ph_b = MODE == 3 ? cv_ph : phase;                  // the carrier phase before this sample's step
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const V n_ff = ff_cur[k] * jomp + ff_next[k] * jphase;     // :305
            const V n_fa = fa_cur[k] * jomp + fa_next[k] * jphase;
            e_freq[k] = e_freq[k] + n_ff * d_ffreq;                    // :764
            const V delta = (n_fa + 1.0f) * amp_scale;                 // :768-769
            const V mul = 1.0f - delta;                                // :772
            e_amp[k] = e_amp[k] * mul;                                 // :773
        }

        // ================= Synthesize::next, src/lib.rs:497-578
        // polyBLEP saw: both branches divide by the jittered frequency  :503-514
        const bool head = ph_b < frequency;
        const bool tail = ph_b > (1.0f - frequency);
        float polyblep = 0.0f;
        if (__builtin_expect(head || tail, 0)) {
            const float tt = (head ? ph_b : (ph_b - 1.0f)) / frequency;
            polyblep = head ? ((2.0f * tt - (tt * tt)) - 1.0f)
                            : (((tt * tt) + 2.0f * tt) + 1.0f);
        }
        const float saw = (2.0f * ph_b - 1.0f) - polyblep;             // :517
        float noise;
        if constexpr (MODE == 3) {
            noise = cv_noise;
        } else {
            phase += frequency;                                        // :520
            if (phase >= 1.0f) phase -= 1.0f;                          // :523-525
            noise = lcg_f32(noise_seed);                               // :528
        }

        // events are rare: this step always takes the IEEE-division body (same bits)
        V v1[NV];
        // FAST kernels, the lane's pair inside the safe window (its own decision: a lane's samples never depend
        // on its wave-mates): the per-formant arithmetic of this sample in tolerance mode too (the control
        // flow and the chain above stay the reference's) — reciprocals by v_rcp + one Newton step, fused
        // multiply-adds, v1 = a1 (b + g v3), v2 = c + g v1
        if (FAST && pair_safe) {
            const V one = vsplat(1.0f, V()), five = vsplat(5.0f, V()), m4 = vsplat(-4.0f, V());
            const V nms = vsplat(noise - saw, V()), nm1 = vsplat(noise - 1.0f, V()), sawv = vsplat(saw, V());
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const V oml = 1.0f - exp_approx(e_smooth[k]);                       // :535
                const V nw = vfma(e_breath[k], nms, sawv);                          // :531
                st_a[k] = vfma(oml, nw - st_a[k], st_a[k]);                         // :538
                const V v0 = st_a[k] * (e_amp[k] * vfma(e_turb[k], nm1, one));      // :544-550
                const V v3 = v0 - st_c[k];                                          // :565
                V w1, w2;
                if constexpr (MID) {
                    // the reference's own coefficients (e_freq, e_bw above ARE its blend and jitter), fused updates
                    const V g = tan_approx<true>(e_freq[k]);                        // :555
                    const V kq = div_exact<true>(e_bw[k], e_freq[k]);               // :558
                    const V a1 = rcp_exact<true>(1.0f + g * (g + kq));              // :560
                    const V a2 = g * a1;                                            // :561
                    const V a3 = g * a2;                                            // :562
                    w1 = vfma(a2, v3, a1 * st_b[k]);                                // :566
                    w2 = vfma(a3, v3, vfma(a2, st_b[k], st_c[k]));                  // :567
                } else {
                    const V x = e_freq[k];
                    const V omx = 1.0f - x, xph = x + 0.5f, hmx = 0.5f - x;
                    const V ox = omx * x, ph_ = xph * hmx;
                    const V num = ox * vfma(m4, ph_, five);
                    const V den = (xph * vfma(m4, ox, five)) * hmx;
                    V rd = vrcp(den), rx = vrcp(x);
                    rd = vfma(vfma(-den, rd, one), rd, rd);
                    rx = vfma(vfma(-x, rx, one), rx, rx);
                    const V tg = num * rd;                                          // :555
                    const V kq = e_bw[k] * rx;                                      // :558
                    const V d3 = vfma(tg, tg + kq, one);
                    V a1 = vrcp(d3);
                    a1 = vfma(vfma(-d3, a1, one), a1, a1);                          // :560
                    w1 = a1 * vfma(tg, v3, st_b[k]);                                // :566
                    w2 = vfma(tg, w1, st_c[k]);                                     // :567
                }
                st_b[k] = vfma(vsplat(2.0f, V()), w1, -st_b[k]);                    // :570
                st_c[k] = vfma(vsplat(2.0f, V()), w2, -st_c[k]);                    // :571
                v1[k] = w1;
            }
        } else {
            formant_filters<false, NV, NV, false, true, V>(saw, noise, 0.0f, e_freq, e_bw, e_smooth, e_breath, e_turb, e_amp,
                                          st_a, st_b, st_c, v1);
        }
        if (!pair_safe) ++slow_steps;

        // v1.sum() * 0.5: a left fold from 0.0 over formants 0..7  :574, :123-125,
        // carried down the utterance's L lanes.
        if constexpr (FOLD_IN_FLUSH) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int c = 0; c < W; ++c)
                    if (emit) stage[(t * S + slot) * NFA + f0 + k * W + c] = vget(v1[k], c);
        } else {
            float acc = 0.0f;
#pragma unroll
            for (int step = 0; step < L; ++step) {
                float run = (step == 0) ? 0.0f : dpp_from_lane_below(acc);
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int c = 0; c < W; ++c) run = run + vget(v1[k], c);
                if (NFA < NF && step == L - 1) run = run + 0.0f;   // formants 5-8: literal +0.0 terms
                acc = (j == step) ? run : acc;
            }
            if (j == L - 1) stage[t * SP + slot] = acc * 0.5f;
        }
        if constexpr (MODE != 3) ++n_out;
    };

    // ---- the quiet sample step: taken when a single ballot shows that NO lane of the wave
    // has any of those events at this sample.  Same arithmetic, straight-line: the polyBLEP
    // quotient is evaluated unconditionally with div_exact<true> and selected afterwards.
    // CALM (calm_tag): the step belongs to a calm tile — no lane that still renders
    // can have an event within the tile — so finished-lane masking is not needed, and the carrier
    // noise (the same LCG state in every lane) arrives precomputed in `noise_in`.
    auto quiet_step = [&](auto nlive_tag, auto su_tag, auto calm_tag, const int t, const float clk_next,
                          const float jphase_next, const float noise_in) __attribute__((always_inline)) {
        constexpr int NLIVE = decltype(nlive_tag)::value;   // vectors whose band-pass runs
        constexpr bool SU = decltype(su_tag)::value;        // one smoothness for every formant
        constexpr bool CALM = decltype(calm_tag)::value;
        constexpr bool KEEP_LP = STREAM;                    // silent formants keep their low-pass
        constexpr int NLP = (NLIVE < NV && !KEEP_LP) ? NLIVE : NV;
        if constexpr (!CALM) {
            if (done) return;                                              // finished lanes sit out
        }
        clk = clk_next;                                                    // :861
        float ratio = clk * inv_blend_length;                              // exact quotient for 2^k
        if constexpr (ANYBL) {
            const float rem = vfma(-blend_length, ratio, clk);
            const float quot = vfma(rem, inv_blend_length, ratio);         // RN(clk / blend_length)
            ratio = blend_pow2 ? ratio : quot;
        }
        float alpha = __builtin_fminf(ratio, 1.0f);                        // :899/:908/:917
        alpha = silent_pair ? 1.0f : alpha;
        const float oma = 1.0f - alpha;
        float frequency = X.frequency * oma + Y.frequency * alpha;         // :404-414
        V e_freq[NV], e_bw[NV], e_smooth[NV], e_breath[NV], e_turb[NV], e_amp[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (k < NLP) {
                e_breath[k] = X.breath[k] * oma + Y.breath[k] * alpha;
                e_smooth[k] = SU ? e_breath[k] : X.smooth[k] * oma + Y.smooth[k] * alpha;
            } else {
                e_breath[k] = vsplat(0.0f, e_breath[k]);   // unused
                e_smooth[k] = e_breath[k];
            }
            if (k < NLIVE) {
                e_freq[k] = X.freq[k] * oma + Y.freq[k] * alpha;
                e_bw[k] = X.bw[k] * oma + Y.bw[k] * alpha;
                e_turb[k] = X.turb[k] * oma + Y.turb[k] * alpha;
                e_amp[k] = X.amp[k] * oma + Y.amp[k] * alpha;
            } else {   // silent vectors: no band-pass
                e_freq[k] = e_breath[k]; e_bw[k] = e_breath[k]; e_turb[k] = e_breath[k]; e_amp[k] = e_breath[k];
            }
        }
        float oml_s = 0.0f;
        if constexpr (SU) {   // :404-414, :535 once for all formants (same operands, same bits)
            const float es = vget(X.smooth[0], 0) * oma + vget(Y.smooth[0], 0) * alpha;
            oml_s = 1.0f - exp_approx(es);
        }
        jphase = jphase_next;                                              // :242 / :291, no wrap
        const float jomp = 1.0f - jphase;
        const float n_freq = fn_cur * jomp + fn_next * jphase;             // :254
        frequency = frequency + n_freq * d_freq;                           // :763
#pragma unroll
        for (int k = 0; k < NLIVE; ++k) {
            const V n_ff = ff_cur[k] * jomp + ff_next[k] * jphase;         // :305
            const V n_fa = fa_cur[k] * jomp + fa_next[k] * jphase;
            e_freq[k] = e_freq[k] + n_ff * d_ffreq;                        // :764
            const V delta = (n_fa + 1.0f) * amp_scale;                     // :768-769
            const V mul = 1.0f - delta;                                    // :772
            e_amp[k] = e_amp[k] * mul;                                     // :773
        }
        const bool head = phase < frequency;                               // :503
        const bool tail = phase > (1.0f - frequency);                      // :507
        const float tt = div_exact<true>(head ? phase : (phase - 1.0f), frequency);
        // :506 (2t - t*t) - 1  and  :510 (t*t + 2t) + 1  are both (2t + s*(t*t)) + s with s = -1
        // (head) or +1 (tail): a - b is a + (-b), IEEE addition commutes, s*(t*t) is a sign flip,
        // and 2t is exact (|t| <= 1 here), so fma(2, t, .) rounds the same sum once
        const float tt2 = tt * tt;
        const float s_tt2 = __uint_as_float(__float_as_uint(tt2) ^ (head ? 0x80000000u : 0u));
        const float pb = vfma(2.0f, tt, s_tt2) + (head ? -1.0f : 1.0f);
        const float polyblep = (head | tail) ? pb : 0.0f;
        // :517  2*phase is exact (0 <= phase < 1), so the fma rounds the same difference once
        const float saw = vfma(2.0f, phase, -1.0f) - polyblep;
        // :520-525  `p += f; if p >= 1 { p -= 1 }` == fract(p + f) for 0 <= p < 1, 0 < f <= 1 (pair_is_safe):
        // x - 1 is exact for x in [1, 2), so both branches give the reference's bits in one instruction
        phase = __builtin_amdgcn_fractf(phase + frequency);
        float noise;                                                       // :528
        if constexpr (CALM) noise = noise_in;
        else noise = lcg_f32(noise_seed);
        V v1[NV];
        formant_filters<true, NV, NLIVE, SU, KEEP_LP, V>(saw, noise, oml_s, e_freq, e_bw, e_smooth, e_breath, e_turb,
                                            e_amp, st_a, st_b, st_c, v1);
        if constexpr (FOLD_IN_FLUSH) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int c = 0; c < W; ++c)
                    if (emit) stage[(t * S + slot) * NFA + f0 + k * W + c] = vget(v1[k], c);   // silent: +0
        } else {
            float acc = 0.0f;
#pragma unroll
            for (int step = 0; step < L; ++step) {
                float run = (step == 0) ? 0.0f : dpp_from_lane_below(acc);
#pragma unroll
                for (int k = 0; k < NLIVE; ++k)
#pragma unroll
                    for (int c = 0; c < W; ++c) run = run + vget(v1[k], c);
                // the silent formants' terms are literal +0.0: ((x + 0) + 0) + ... == x + 0
                if (NLIVE < NV || (NFA < NF && step == L - 1)) run = run + 0.0f;
                acc = (j == step) ? run : acc;
            }
            if (j == L - 1) stage[t * SP + slot] = acc * 0.5f;
        }
        if constexpr (!CALM) ++n_out;      // a calm tile adds its T samples at once
    };

    // ---- L = 8 (one formant per lane): the packed slot that holds a second formant for smaller L
    // takes the SAME formant at the NEXT sample instead.  In a calm tile nothing but the carrier
    // phase and the filter state links sample tc to tc+1, so everything else — blend, jitter,
    // tan_approx, the divisions, polyBLEP — is evaluated for both samples at once (.x = tc,
    // .y = tc+1): the same operations on the same operands as two quiet steps, two per issue slot.
    // The per-formant part of two calm samples (.x = tc, .y = tc+1) from their chain values: blend, jitter,
    // coefficients, the two filter steps.
    auto formant_pair = [&](const f2 alpha, const f2 oma, const f2 JP, const f2 jomp, const f2 saw, const f2 NZ,
                            const int tc) __attribute__((always_inline)) {
        if constexpr (W == 1 && NV == 1 && FOLD_IN_FLUSH) {
            // SynthesisElem::blend :404-414, Jitter::next :753-777
            f2 e_freq = X.freq[0] * oma + Y.freq[0] * alpha;
            const f2 e_bw = X.bw[0] * oma + Y.bw[0] * alpha;
            const f2 e_smooth = X.smooth[0] * oma + Y.smooth[0] * alpha;
            const f2 e_breath = X.breath[0] * oma + Y.breath[0] * alpha;
            const f2 e_turb = X.turb[0] * oma + Y.turb[0] * alpha;
            f2 e_amp = X.amp[0] * oma + Y.amp[0] * alpha;
            const f2 n_ff = ff_cur[0] * jomp + ff_next[0] * JP;                // :305
            const f2 n_fa = fa_cur[0] * jomp + fa_next[0] * JP;
            e_freq = e_freq + n_ff * d_ffreq;                                  // :764
            const f2 delta = (n_fa + 1.0f) * amp_scale;                        // :768-769
            e_amp = e_amp * (1.0f - delta);                                    // :772-773
            // Synthesize::next coefficients :535, :555-562 (as formant_filters<true>)
            const f2 oml = 1.0f - exp_approx(e_smooth);
            const f2 omx = 1.0f - e_freq, xph = e_freq + 0.5f, hmx = 0.5f - e_freq;
            const f2 ox = omx * e_freq, ph = xph * hmx;
            const f2 five = vsplat(5.0f, f2()), m4 = vsplat(-4.0f, f2());
            const f2 num = ox * vfma(m4, ph, five);
            const f2 den = (xph * vfma(m4, ox, five)) * hmx;
            const f2 g = div_exact<true>(num, den);                            // :555
            const f2 kq = div_exact<true>(e_bw, e_freq);                       // :558
            const f2 a1 = rcp_exact<true>(1.0f + g * (g + kq));                // :560
            const f2 a2 = g * a1;                                              // :561
            const f2 a3 = g * a2;                                              // :562
            const f2 tmix = (1.0f - e_turb) + NZ * e_turb;                     // :544-545
            const f2 nw = saw * (1.0f - e_breath) + NZ * e_breath;             // :531
            // the filter recurrences :538-571, sample tc then tc+1
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float sa = st_a[0], sb = st_b[0], sc = st_c[0];
                sa = sa + vget(oml, h) * (vget(nw, h) - sa);                   // :538
                const float tw = sa * vget(tmix, h);
                const float v0 = tw * vget(e_amp, h);                          // :550
                const float v3 = v0 - sc;                                      // :565
                const float w1 = vget(a1, h) * sb + vget(a2, h) * v3;          // :566
                const float w2 = (sc + vget(a2, h) * sb) + vget(a3, h) * v3;   // :567
                st_a[0] = sa;
                st_b[0] = 2.0f * w1 - sb;                                      // :570
                st_c[0] = 2.0f * w2 - sc;                                      // :571
                if (emit) stage[((tc + h) * S + slot) * NFA + f0] = w1;   // (PIPE: the rendering wave's filters are the live ones)
            }
        }
    };

    // One formant per lane, eight calm samples: the four lanes of a quad carry the same utterance, so the quad
    // shares the per-utterance chain — quad lane i works out sample pair i (quad_chain), every lane then takes
    // the four pairs' chain values from their lanes and runs its formant through them.
    auto quad_bcast = [](const float x, auto sel_tag) __attribute__((always_inline)) {
        constexpr int I = decltype(sel_tag)::value;
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), I * 0x55, 0xF, 0xF, true));   // quad_perm:[I,I,I,I]
    };
    // Only what is serial — the clock, the jitter phase, the carrier phase — is stepped through all eight
    // samples by every lane (the reference's operations in the reference's order; a lane latches the values of
    // its pair); alpha, pitch, polyBLEP and saw are evaluated once per pair instead of once per lane and pair.
    auto quad_chain = [&](const float noise_of_step, const int first_step, f2 &alpha, f2 &JP, f2 &saw,
                          f2 &NZ) __attribute__((always_inline)) {
        static_assert(L >= 4 || !PIPE, "a quad of lanes per utterance");
        const f2 one2 = vsplat(1.0f, f2());
        const int jq = lane & 3;
        float c = clk, p = jphase;
        f2 CLK = vsplat(0.0f, f2());
        JP = CLK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool me = jq == i;
            c = c - dt;                                                        // :861
            p = p + jinc;                                                      // :242 / :291
            CLK.x = me ? c : CLK.x;
            JP.x = me ? p : JP.x;
            c = c - dt;
            p = p + jinc;
            CLK.y = me ? c : CLK.y;
            JP.y = me ? p : JP.y;
        }
        clk = c;
        jphase = p;
        f2 ratio = CLK * inv_blend_length;
        if constexpr (ANYBL) {
            const f2 rem = vfma(-blend_length * one2, ratio, CLK);
            const f2 quot = vfma(rem, inv_blend_length * one2, ratio);         // RN(clk / blend_length)
            ratio = blend_pow2 ? ratio : quot;
        }
        alpha.x = silent_pair ? 1.0f : __builtin_fminf(ratio.x, 1.0f);         // :899/:908/:917
        alpha.y = silent_pair ? 1.0f : __builtin_fminf(ratio.y, 1.0f);
        const f2 oma = 1.0f - alpha;
        const f2 jomp = 1.0f - JP;
        f2 frequency = X.frequency * oma + Y.frequency * alpha;                // :404-414
        const f2 n_freq = fn_cur * jomp + fn_next * JP;                        // :254
        frequency = frequency + n_freq * d_freq;                               // :763
        // carrier :503-525: the phase goes through the eight samples in order, pitch by pitch
        float ph = phase;
        f2 PH = vsplat(0.0f, f2());
        auto two_steps = [&](auto sel_tag) __attribute__((always_inline)) {
            constexpr int I = decltype(sel_tag)::value;
            const bool me = jq == I;
            PH.x = me ? ph : PH.x;
            ph = __builtin_amdgcn_fractf(ph + quad_bcast(frequency.x, sel_tag));         // see quiet_step
            PH.y = me ? ph : PH.y;
            ph = __builtin_amdgcn_fractf(ph + quad_bcast(frequency.y, sel_tag));
        };
        two_steps(std::integral_constant<int, 0>());
        two_steps(std::integral_constant<int, 1>());
        two_steps(std::integral_constant<int, 2>());
        two_steps(std::integral_constant<int, 3>());
        phase = ph;
        const f2 omf = 1.0f - frequency;
        const bool head0 = PH.x < frequency.x, tail0 = PH.x > omf.x;
        const bool head1 = PH.y < frequency.y, tail1 = PH.y > omf.y;
        const f2 phm1 = PH - 1.0f;
        f2 dividend;
        dividend.x = head0 ? PH.x : phm1.x;
        dividend.y = head1 ? PH.y : phm1.y;
        const f2 tt = div_exact<true>(dividend, frequency);
        const f2 tt2 = tt * tt;
        f2 s_tt2, sgn, polyblep;                                               // see quiet_step
        s_tt2.x = __uint_as_float(__float_as_uint(tt2.x) ^ (head0 ? 0x80000000u : 0u));
        s_tt2.y = __uint_as_float(__float_as_uint(tt2.y) ^ (head1 ? 0x80000000u : 0u));
        sgn.x = head0 ? -1.0f : 1.0f;
        sgn.y = head1 ? -1.0f : 1.0f;
        const f2 pb = vfma(vsplat(2.0f, f2()), tt, s_tt2) + sgn;
        polyblep.x = (head0 | tail0) ? pb.x : 0.0f;
        polyblep.y = (head1 | tail1) ? pb.y : 0.0f;
        saw = vfma(vsplat(2.0f, f2()), PH, -one2) - polyblep;                  // :517
        // the carrier noise of my two samples: lane t of noise_of_step holds the tile's step t
        const int at = first_step + 2 * jq;
        NZ.x = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * at, __float_as_int(noise_of_step)));
        NZ.y = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * at + 4, __float_as_int(noise_of_step)));
    };
    auto time_packed_block = [&](const int tc, const float noise_of_step) __attribute__((always_inline)) {
        if constexpr (W == 1 && NV == 1 && FOLD_IN_FLUSH && L >= 4) {
            f2 alpha, JP, saw, NZ;
            quad_chain(noise_of_step, tc, alpha, JP, saw, NZ);
            auto pair_from = [&](auto sel_tag) __attribute__((always_inline)) {
                constexpr int I = decltype(sel_tag)::value;
                f2 al, jp, sw, nz;
                al.x = quad_bcast(alpha.x, sel_tag); al.y = quad_bcast(alpha.y, sel_tag);
                jp.x = quad_bcast(JP.x, sel_tag); jp.y = quad_bcast(JP.y, sel_tag);
                sw.x = quad_bcast(saw.x, sel_tag); sw.y = quad_bcast(saw.y, sel_tag);
                nz.x = quad_bcast(NZ.x, sel_tag); nz.y = quad_bcast(NZ.y, sel_tag);
                formant_pair(al, 1.0f - al, jp, 1.0f - jp, sw, nz, tc + 2 * I);
            };
            pair_from(std::integral_constant<int, 0>());
            pair_from(std::integral_constant<int, 1>());
            pair_from(std::integral_constant<int, 2>());
            pair_from(std::integral_constant<int, 3>());
        }
    };

    // ---- PIPE: time_packed_steps cut in three, one piece per role, handed on through LDS.
    //   pipe_chain  (wave 1): clock, alpha, jitter phase, pitch blend and jitter, carrier phase, polyBLEP
    //                         and saw of four sample pairs — the per-utterance chain, once for all formants
    //   pipe_coeffs (waves 2, 3; wave 1): blend, jitter, 1-exp(smooth), the low-pass input, the
    //                         turbulence mix, the jittered amplitude, a1 and g from that chain
    //   pipe_render (wave 0): a2 = g a1, a3 = g a2, the two filter recurrence steps and the band-pass outputs
    // Same operations on the same operands in the same order as time_packed_steps.
    // PIPE: a round is 2 * QP sample pairs.  QP = 2: each coefficient wave takes two of its four pairs.  QP = 4
    // (16 samples between barriers; the one in use): the coefficient waves take three pairs
    // each and the chain wave — the lightest stage — the last two of the round it wrote one phase before.
    constexpr int QP = PQP;
    // [round & 1][group of four pairs][q][lane]: lane (quad | pair) holds the pair's chain
    __shared__ float4 chain_all[PIPE ? 2 : 1][PIPE ? (QP + 1) / 2 : 1][PIPE ? 2 : 1][PIPE ? 64 : 1];
    __shared__ float4 ring_all[PIPE ? 2 : 1][PIPE ? 2 * QP : 1][PIPE ? 3 : 1][PIPE ? 64 : 1];
    __shared__ float hand_all[PIPE ? 3 : 1][PIPE ? 64 : 1];
    // One round = groups of four sample pairs, each shared by the quad (quad_chain above).
    auto pipe_chain = [&](float4 (*dst)[64], const float noise_of_step, const int first_step) __attribute__((always_inline)) {
        if constexpr (PIPE) {
            static_assert(!PIPE || QP % 2 == 0, "a quad shares four pairs");
            f2 alpha, JP, saw, NZ;
            quad_chain(noise_of_step, first_step, alpha, JP, saw, NZ);
            dst[0][lane] = make_float4(alpha.x, alpha.y, JP.x, JP.y);
            dst[1][lane] = make_float4(saw.x, saw.y, NZ.x, NZ.y);
        }
    };
    auto pipe_coeffs = [&](const float4 (*src)[64], const int pair, float4 (*dst)[64]) __attribute__((always_inline)) {
        if constexpr (PIPE) {
            const int from = (lane & ~3) | pair;                               // the quad lane that worked out this pair
            const float4 c0 = src[0][from], c2 = src[1][from];
            f2 alpha, JP, saw, NZ;
            alpha.x = c0.x; alpha.y = c0.y; JP.x = c0.z; JP.y = c0.w;
            saw.x = c2.x; saw.y = c2.y; NZ.x = c2.z; NZ.y = c2.w;
            const f2 oma = 1.0f - alpha;                                       // as the chain has them
            const f2 jomp = 1.0f - JP;
            f2 e_freq = X.freq[0] * oma + Y.freq[0] * alpha;                   // :404-414
            const f2 e_bw = X.bw[0] * oma + Y.bw[0] * alpha;
            const f2 e_smooth = X.smooth[0] * oma + Y.smooth[0] * alpha;
            const f2 e_breath = X.breath[0] * oma + Y.breath[0] * alpha;
            const f2 e_turb = X.turb[0] * oma + Y.turb[0] * alpha;
            f2 e_amp = X.amp[0] * oma + Y.amp[0] * alpha;
            const f2 n_ff = ff_cur[0] * jomp + ff_next[0] * JP;                // :305
            const f2 n_fa = fa_cur[0] * jomp + fa_next[0] * JP;
            e_freq = e_freq + n_ff * d_ffreq;                                  // :764
            const f2 delta = (n_fa + 1.0f) * amp_scale;                        // :768-769
            e_amp = e_amp * (1.0f - delta);                                    // :772-773
            const f2 oml = 1.0f - exp_approx(e_smooth);                        // :535
            const f2 omx = 1.0f - e_freq, xph = e_freq + 0.5f, hmx = 0.5f - e_freq;
            const f2 ox = omx * e_freq, ph = xph * hmx;
            const f2 five = vsplat(5.0f, f2()), m4 = vsplat(-4.0f, f2());
            const f2 num = ox * vfma(m4, ph, five);
            const f2 den = (xph * vfma(m4, ox, five)) * hmx;
            const f2 g = div_exact<true>(num, den);                            // :555
            const f2 kq = div_exact<true>(e_bw, e_freq);                       // :558
            const f2 a1 = rcp_exact<true>(1.0f + g * (g + kq));                // :560
            const f2 tmix = (1.0f - e_turb) + NZ * e_turb;                     // :544-545
            const f2 nw = saw * (1.0f - e_breath) + NZ * e_breath;             // :531
            dst[0][lane] = make_float4(oml.x, oml.y, nw.x, nw.y);
            dst[1][lane] = make_float4(tmix.x, tmix.y, e_amp.x, e_amp.y);
            dst[2][lane] = make_float4(a1.x, a1.y, g.x, g.y);                  // a2, a3: the render wave's two products
        }
    };
    auto pipe_render = [&](const float4 (*src)[64], const int tc) __attribute__((always_inline)) {
        if constexpr (PIPE) {
            const float4 q0 = src[0][lane], q1 = src[1][lane], q2 = src[2][lane];
            const float oml[2] = {q0.x, q0.y}, nw[2] = {q0.z, q0.w}, tmix[2] = {q1.x, q1.y};
            const float amp[2] = {q1.z, q1.w}, a1[2] = {q2.x, q2.y}, g[2] = {q2.z, q2.w};
            const float a2[2] = {g[0] * a1[0], g[1] * a1[1]};                  // :561
            const float a3[2] = {g[0] * a2[0], g[1] * a2[1]};                  // :562
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                      // :538-571
                float sa = st_a[0], sb = st_b[0], sc = st_c[0];
                sa = sa + oml[h] * (nw[h] - sa);                               // :538
                const float tw = sa * tmix[h];
                const float v0 = tw * amp[h];                                  // :550
                const float v3 = v0 - sc;                                      // :565
                const float w1 = a1[h] * sb + a2[h] * v3;                      // :566
                const float w2 = (sc + a2[h] * sb) + a3[h] * v3;               // :567
                st_a[0] = sa;
                st_b[0] = 2.0f * w1 - sb;                                      // :570
                st_c[0] = 2.0f * w2 - sc;                                      // :571
                stage[((tc + h) * S + slot) * NFA + f0] = w1;
            }
        }
    };

    // ---- two calm samples per trip, smaller L: the per-utterance chain (clock, alpha, pitch blend and
    // jitter, shared smoothness, polyBLEP, saw) is evaluated for samples tc and tc+1 at once on
    // float2 values (.x = tc, .y = tc+1), exactly as in time_packed_steps; only the carrier phase is
    // carried between the two.  The formant vectors, already packed across formants, then run
    // sample by sample with those scalars.
    // The per-formant part of two calm samples (.x = tc, .y = tc+1) from their chain values, formant vectors
    // packed across formants: blend, jitter, coefficients and filters sample by sample.
    // where the shared low-pass factor is worked out — before the carrier or after it — is the same arithmetic,
    // but it moves the compiler's schedule: the two-lane kernels measure 2 - 3 % faster with it first, the
    // one-lane kernels 2.7 % faster with it last (same-box A/B)
    constexpr bool OML_EARLY = L == 2;
    auto scalar_formant_pair = [&](auto nlive_tag, auto su_tag, const f2 alpha, const f2 oma, const f2 JP,
                                   const f2 jomp, const f2 saw2, const int tc, const float nz0,
                                   const float nz1, const f2 oml_early) __attribute__((always_inline)) {
        constexpr int NLIVE = decltype(nlive_tag)::value;
        constexpr bool SU = decltype(su_tag)::value;
        constexpr bool KEEP_LP = STREAM;
        constexpr int NLP = (NLIVE < NV && !KEEP_LP) ? NLIVE : NV;
        const f2 one2 = vsplat(1.0f, f2());
        f2 oml = one2;
        if constexpr (SU) {   // :404-414, :535 once for all formants (same operands, same bits)
            if constexpr (OML_EARLY) {
                oml = oml_early;
            } else {
                const f2 es = vget(X.smooth[0], 0) * oma + vget(Y.smooth[0], 0) * alpha;
                oml = 1.0f - exp_approx(es);
            }
        }
        V E_freq[2][NV], E_bw[2][NV], E_smooth[2][NV], E_breath[2][NV], E_turb[2][NV], E_amp[2][NV];
        auto blend_h = [&](const int h) __attribute__((always_inline)) {
            const float a = vget(alpha, h), om = vget(oma, h), jp = vget(JP, h), jm = vget(jomp, h);
            V (&e_freq)[NV] = E_freq[h]; V (&e_bw)[NV] = E_bw[h]; V (&e_smooth)[NV] = E_smooth[h]; V (&e_breath)[NV] = E_breath[h]; V (&e_turb)[NV] = E_turb[h]; V (&e_amp)[NV] = E_amp[h];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (k < NLP) {
                    e_breath[k] = X.breath[k] * om + Y.breath[k] * a;
                    e_smooth[k] = SU ? e_breath[k] : X.smooth[k] * om + Y.smooth[k] * a;
                } else {
                    e_breath[k] = vsplat(0.0f, e_breath[k]);   // unused
                    e_smooth[k] = e_breath[k];
                }
                if (k < NLIVE) {
                    e_freq[k] = X.freq[k] * om + Y.freq[k] * a;
                    e_bw[k] = X.bw[k] * om + Y.bw[k] * a;
                    e_turb[k] = X.turb[k] * om + Y.turb[k] * a;
                    e_amp[k] = X.amp[k] * om + Y.amp[k] * a;
                } else {   // silent vectors: no band-pass
                    e_freq[k] = e_breath[k]; e_bw[k] = e_breath[k]; e_turb[k] = e_breath[k]; e_amp[k] = e_breath[k];
                }
            }
#pragma unroll
            for (int k = 0; k < NLIVE; ++k) {
                const V n_ff = ff_cur[k] * jm + ff_next[k] * jp;               // :305
                const V n_fa = fa_cur[k] * jm + fa_next[k] * jp;
                e_freq[k] = e_freq[k] + n_ff * d_ffreq;                        // :764
                const V delta = (n_fa + 1.0f) * amp_scale;                     // :768-769
                const V mul = 1.0f - delta;                                    // :772
                e_amp[k] = e_amp[k] * mul;                                     // :773
            }
        };
        auto filter_h = [&](const int h) __attribute__((always_inline)) {
            const float noise = h == 0 ? nz0 : nz1;
            V (&e_freq)[NV] = E_freq[h]; V (&e_bw)[NV] = E_bw[h]; V (&e_smooth)[NV] = E_smooth[h]; V (&e_breath)[NV] = E_breath[h]; V (&e_turb)[NV] = E_turb[h]; V (&e_amp)[NV] = E_amp[h];
            V v1[NV];
            formant_filters<true, NV, NLIVE, SU, KEEP_LP, V>(vget(saw2, h), noise, vget(oml, h), e_freq, e_bw,
                                                e_smooth, e_breath, e_turb, e_amp, st_a, st_b, st_c, v1);
            const int t = tc + h;
            if constexpr (FOLD_IN_FLUSH) {
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int c = 0; c < W; ++c)
                        stage[(t * S + slot) * NFA + f0 + k * W + c] = vget(v1[k], c);   // silent: +0
            } else {
                float acc = 0.0f;
#pragma unroll
                for (int step = 0; step < L; ++step) {
                    float run = (step == 0) ? 0.0f : dpp_from_lane_below(acc);
#pragma unroll
                    for (int k = 0; k < NLIVE; ++k)
#pragma unroll
                        for (int c = 0; c < W; ++c) run = run + vget(v1[k], c);
                    if (NLIVE < NV || (NFA < NF && step == L - 1)) run = run + 0.0f;
                    acc = (j == step) ? run : acc;
                }
                if (j == L - 1) stage[t * SP + slot] = acc * 0.5f;
            }
        };
        // two formant vectors: the blends of both samples before the filters of the first (measured: the
        // better schedule); four: sample by sample (the register file does not hold both sets)
        if constexpr (NLIVE <= 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) blend_h(h);
#pragma unroll
            for (int h = 0; h < 2; ++h) filter_h(h);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) { blend_h(h); filter_h(h); }
        }
    };
    auto scalar_packed_steps = [&](auto nlive_tag, auto su_tag, const int tc, const float nz0,
                                   const float nz1) __attribute__((always_inline)) {
        const f2 one2 = vsplat(1.0f, f2());
        const float clk0 = clk - dt, clk1 = clk0 - dt;                         // :861
        const float jp0 = jphase + jinc, jp1 = jp0 + jinc;                     // :242 / :291
        clk = clk1;
        jphase = jp1;
        f2 CLK, JP;
        CLK.x = clk0; CLK.y = clk1; JP.x = jp0; JP.y = jp1;
        f2 ratio = CLK * inv_blend_length;
        if constexpr (ANYBL) {
            const f2 rem = vfma(-blend_length * one2, ratio, CLK);
            const f2 quot = vfma(rem, inv_blend_length * one2, ratio);         // RN(clk / blend_length)
            ratio = blend_pow2 ? ratio : quot;
        }
        f2 alpha;                                                              // :899/:908/:917
        alpha.x = silent_pair ? 1.0f : __builtin_fminf(ratio.x, 1.0f);
        alpha.y = silent_pair ? 1.0f : __builtin_fminf(ratio.y, 1.0f);
        const f2 oma = 1.0f - alpha;
        const f2 jomp = 1.0f - JP;
        f2 frequency = X.frequency * oma + Y.frequency * alpha;                // :404-414
        const f2 n_freq = fn_cur * jomp + fn_next * JP;                        // :254
        frequency = frequency + n_freq * d_freq;                               // :763
        f2 oml_early = one2;
        if constexpr (OML_EARLY && decltype(su_tag)::value) {   // :404-414, :535 once for all formants
            const f2 es = vget(X.smooth[0], 0) * oma + vget(Y.smooth[0], 0) * alpha;
            oml_early = 1.0f - exp_approx(es);
        }
        // carrier :503-525
        const f2 omf = 1.0f - frequency;
        const float ph0 = phase;
        const bool head0 = ph0 < frequency.x, tail0 = ph0 > omf.x;
        const float ph1 = __builtin_amdgcn_fractf(ph0 + frequency.x);         // see quiet_step
        const bool head1 = ph1 < frequency.y, tail1 = ph1 > omf.y;
        phase = __builtin_amdgcn_fractf(ph1 + frequency.y);
        f2 PH;
        PH.x = ph0; PH.y = ph1;
        const f2 phm1 = PH - 1.0f;
        f2 dividend;
        dividend.x = head0 ? ph0 : phm1.x;
        dividend.y = head1 ? ph1 : phm1.y;
        const f2 tt = div_exact<true>(dividend, frequency);
        const f2 tt2 = tt * tt;
        f2 s_tt2, sgn, polyblep;                                               // see quiet_step
        s_tt2.x = __uint_as_float(__float_as_uint(tt2.x) ^ (head0 ? 0x80000000u : 0u));
        s_tt2.y = __uint_as_float(__float_as_uint(tt2.y) ^ (head1 ? 0x80000000u : 0u));
        sgn.x = head0 ? -1.0f : 1.0f;
        sgn.y = head1 ? -1.0f : 1.0f;
        const f2 pb = vfma(vsplat(2.0f, f2()), tt, s_tt2) + sgn;
        polyblep.x = (head0 | tail0) ? pb.x : 0.0f;
        polyblep.y = (head1 | tail1) ? pb.y : 0.0f;
        const f2 saw2 = vfma(vsplat(2.0f, f2()), PH, -one2) - polyblep;        // :517
        scalar_formant_pair(nlive_tag, su_tag, alpha, oma, JP, jomp, saw2, tc, nz0, nz1, oml_early);
    };
    // L = 4 with two formants per lane: the quad shares the chain over eight calm samples (quad_chain above)
    auto scalar_packed_block = [&](auto nlive_tag, auto su_tag, const int tc, const float noise_of_step) __attribute__((always_inline)) {
        if constexpr (L >= 4) {
            f2 alpha, JP, saw, NZ;
            quad_chain(noise_of_step, tc, alpha, JP, saw, NZ);
            auto pair_from = [&](auto sel_tag) __attribute__((always_inline)) {
                constexpr int I = decltype(sel_tag)::value;
                f2 al, jp, sw;
                al.x = quad_bcast(alpha.x, sel_tag); al.y = quad_bcast(alpha.y, sel_tag);
                jp.x = quad_bcast(JP.x, sel_tag); jp.y = quad_bcast(JP.y, sel_tag);
                sw.x = quad_bcast(saw.x, sel_tag); sw.y = quad_bcast(saw.y, sel_tag);
                scalar_formant_pair(nlive_tag, su_tag, al, 1.0f - al, jp, 1.0f - jp, sw, tc + 2 * I,
                                    quad_bcast(NZ.x, sel_tag), quad_bcast(NZ.y, sel_tag), vsplat(1.0f, f2()));
            };
            pair_from(std::integral_constant<int, 0>());
            pair_from(std::integral_constant<int, 1>());
            pair_from(std::integral_constant<int, 2>());
            pair_from(std::integral_constant<int, 3>());
        }
    };


    // ---- FAST: tolerance-mode arithmetic.
    // Exact, as everywhere: clk (:861), alpha, the pitch blend and its jitter (:404, :254, :763), the
    // jitter phase (:242) and the carrier phase with its wrap (:520-525) — two samples per packed
    // slot, the same operations on the same operands as the exact kernels.  Within tolerance:
    //   * the polyBLEP quotient (:505/:509) is dividend * v_rcp(frequency);
    //   * the band-pass (:560-571) is used in the algebraically equal form a2 = g a1, a3 = g a2 =>
    //     v1 = a1 (b + g v3),  v2 = c + g v1,  so only a1 and g = tan_approx(x) are needed per sample;
    //   * per formant, everything that is a smooth function of (alpha, jitter phase) — a1, g, the
    //     jittered amplitude G, amplitude x turbulence H, breath, 1 - exp_approx(smooth) — is evaluated
    //     at the ends of SUB-TILES of TS <= 32 samples and interpolated linearly in between.  Alpha and the
    //     jitter phase are linear in time between two events of the lane — a segment advance (:864-888), a noise wrap
    //     (:245), the kink of alpha = min(clk / blend_length, 1) (:899) — and NO SUB-TILE REACHES ACROSS AN EVENT
    //     (fast_horizon): a sub-tile lives in one regime, alpha standing at one or falling with the clock, and takes
    //     its far end from that regime's own formulas.  The end of a sub-tile is the start of the next one.
    //   * the interpolation error is bounded where a lane begins anew behind an event of ITS OWN (fast_level): a relative
    //     change r of a1 or 1 - exp_approx over 32 samples gives an error below r^2/16 <= 2^-23 for r <= 2^-9.5; g =
    //     tan_approx(x) of an x that is linear in time has the curvature of the tangent only, r^2 g^2 / (4 (1 + g^2))
    //     (checked numerically for the reference's rational function, whose own curvature dominates below x = 0.02:
    //     its change is weighed by min(max(2.5 g, 0.1), 2)); G and H are products of linear functions, error
    //     <= |dA dM| / 4 and |dT dG| / 4 <= 2^-20 absolute.  Faster parameter motion halves TS (error / 4) until it fits,
    //     down to TS = 1: every sample from its own evaluation.  The reference's own front end always emits 0.5 s blends
    //     (Intonator :1070-1071), for which TS = 32.
    //   * :531 as saw + breath (noise - saw), :538 as fma, :544-550 as a (G + H (noise - 1)), the
    //     eight-term sum (:574) in tree order.
    // BATCH INVARIANCE.  Where a lane's sub-tiles begin and end, their length, its smoothness flavour, whether a
    // sample of it is stepped by the packed chain or by the reference's control flow — all of it follows from the lane's
    // own state on the utterance's own grid of T-sample tiles; the wave decides only which COPY of the code runs (the
    // tight loops of a tile in which every lane is calm, the plain pairs of a mixed tile, its slow samples), and the
    // copies perform the same operations on a lane's values.  The samples of an utterance therefore do not depend on
    // which utterances share its wave.
    struct FastEnds {
        V a1[NV], tg[NV], g[NV], h[NV], b[NV], om[NV];   // tg = tan_approx(x), g = amplitude
        float oml;
    };
    struct FastAux {
        V ap[NV], mu[NV], tb[NV];
    };
    FastEnds FS;             // the interpolated quantities at the first sample of the lane's sub-tile
    FastEnds FD;             // their per-sample slopes over the lane's current sub-tile
    f2 FTI = vsplat(0.0f, f2());   // position of the next sample pair inside the lane's sub-tile: (i, i + 1)
    int fast_have = -1;      // the flavour (1: shared smoothness, 0: per formant) of the run FS belongs to; -1: no run
    int fast_shift = 0;      // the lane's sub-tile length is 32 >> fast_shift (5: one sample), chosen where it begins anew
    int fast_sub_left = 0;   // samples of the lane's current sub-tile still to render (0: between sub-tiles — FS holds the
                             // values of the next sample, the slopes are due)
    float fast_sub_len = 32.0f;   // length of the lane's current sub-tile (one that begins between grid points, or in front
                                  // of an event, is shorter than 32 >> fast_shift)
    constexpr int FAST_TS0 = 32;
    static_assert(!FAST || T % FAST_TS0 == 0, "whole sub-tiles");
    // the lane's regime at a sample with clock c: alpha stands at one (both sides silent: alpha = 1, :926; or the
    // quotient is above one), or falls with the clock
    auto fast_flat_at = [&](const float c) __attribute__((always_inline)) -> bool {
        return silent_pair | (c * inv_blend_length > 1.0f);
    };
    // the smooth quantities `after` samples from the state (clk, jphase) along the lane's regime (the clock and the
    // jitter phase extrapolated: they only feed continuous functions here).  SLOPE: e receives (value - FS) * scale instead.
    auto fast_endpoint = [&](auto su_tag, auto slope_tag, const float after, const float scale, const bool flat, FastEnds &e,
                             FastAux &x) __attribute__((always_inline)) {
        constexpr bool SU = decltype(su_tag)::value;
        constexpr bool SLOPE = decltype(slope_tag)::value;
        const V one = vsplat(1.0f, V());
        const V five = vsplat(5.0f, V()), m4 = vsplat(-4.0f, V());
        const float c = clk - after * dt;
        const float jp = jphase + after * jinc;
        const float alpha = flat ? 1.0f : c * inv_blend_length;
        const float oma = 1.0f - alpha, jomp = 1.0f - jp;
        auto put = [&](V &dst, const V &start, const V value) __attribute__((always_inline)) {
            if constexpr (SLOPE) dst = (value - start) * scale;
            else dst = value;
        };
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            V ef = vfma(Y.freq[k], vsplat(alpha, V()), X.freq[k] * oma);
            const V eb = vfma(Y.bw[k], vsplat(alpha, V()), X.bw[k] * oma);
            const V et = vfma(Y.turb[k], vsplat(alpha, V()), X.turb[k] * oma);
            const V ea = vfma(Y.amp[k], vsplat(alpha, V()), X.amp[k] * oma);
            const V nff = vfma(ff_next[k], vsplat(jp, V()), ff_cur[k] * jomp);
            const V nfa = vfma(fa_next[k], vsplat(jp, V()), fa_cur[k] * jomp);
            ef = vfma(nff, vsplat(d_ffreq, V()), ef);
            const V mul = vfma(nfa + 1.0f, vsplat(-amp_scale, V()), one);
            if constexpr (MID) {
                // (a1, a2, a3 come from the reference's own sequence at every sample — nothing to interpolate)
                const V gg = ea * mul;
                e.a1[k] = one;
                e.tg[k] = one;
                (void)eb; (void)five; (void)m4;
                put(e.g[k], FS.g[k], gg);
                put(e.h[k], FS.h[k], et * gg);
                x.ap[k] = ea;
                x.mu[k] = mul;
                x.tb[k] = et;
                continue;
            }
            const V omx = 1.0f - ef, xph = ef + 0.5f, hmx = 0.5f - ef;
            const V ox = omx * ef, ph = xph * hmx;
            const V num = ox * vfma(m4, ph, five);
            const V den = (xph * vfma(m4, ox, five)) * hmx;
            // g = num / den (:555), k = bw / x (:558), a1 = 1 / (1 + g (g + k)) (:560): v_rcp + one
            // Newton step each (correctly rounded reciprocals; the quotients are within an ulp)
            V rd = vrcp(den), rx = vrcp(ef);
            rd = vfma(vfma(-den, rd, one), rd, rd);
            rx = vfma(vfma(-ef, rx, one), rx, rx);
            const V tg = num * rd;
            const V kq = eb * rx;
            const V d3 = vfma(tg, tg + kq, one);
            V r3 = vrcp(d3);
            r3 = vfma(vfma(-d3, r3, one), r3, r3);
            const V gg = ea * mul;
            put(e.a1[k], FS.a1[k], r3);
            put(e.tg[k], FS.tg[k], tg);
            put(e.g[k], FS.g[k], gg);
            put(e.h[k], FS.h[k], et * gg);
            x.ap[k] = ea;
            x.mu[k] = mul;
            x.tb[k] = et;
        }
        float oml_here = 1.0f;
        if constexpr (SU) {
            const float es = __builtin_fmaf(vget(Y.smooth[0], 0), alpha, vget(X.smooth[0], 0) * oma);
            oml_here = 1.0f - exp_approx(es);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const V br = vfma(Y.breath[k], vsplat(alpha, V()), X.breath[k] * oma);
            if constexpr (!SU) {
                put(e.b[k], FS.b[k], br);
                const V es = vfma(Y.smooth[k], vsplat(alpha, V()), X.smooth[k] * oma);
                put(e.om[k], FS.om[k], 1.0f - exp_approx(es));
            } else {
                // shared smoothness: the low-pass is used as a' = (1-k) a + k saw + (k breath)(noise - saw),
                // so the interpolated per-formant quantity is k * breath
                put(e.b[k], FS.b[k], br * oml_here);
                e.om[k] = one;
            }
        }
        e.oml = SU ? (SLOPE ? (oml_here - FS.oml) * scale : oml_here) : 1.0f;
    };
    // The error guard: how many halvings of the 32-sample sub-tile the motion of the lane's parameters asks for, from the
    // values FS at a sample, the slopes FD towards a point `span` samples later and the factors of G and H at both
    // (xs, xe).  0 .. 4: sub-tiles of 32 .. 2 samples; 5: faster than two samples can follow (or not a number): every
    // sample from its own evaluation.
    auto fast_level = [&](auto su_tag, const FastAux &xs, const FastAux &xe, const float span) __attribute__((always_inline)) -> int {
        constexpr bool SU = decltype(su_tag)::value;
        constexpr int TS0 = FAST_TS0;
        const float to32 = (float)TS0 * __builtin_amdgcn_rcpf(span);      // (span <= 32: exact where it matters, 32 / 32)
        float ra = 0.0f, rg = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int c = 0; c < W; ++c) {
                // relative change of a1 and g over 32 samples; 32^2 x the products of slopes behind G and H
                if constexpr (!MID) {
                    ra = __builtin_fmaxf(ra, __builtin_fabsf(vget(FD.a1[k], c)) * (float)TS0 *
                                                 __builtin_amdgcn_rcpf(vget(FS.a1[k], c)));
                    // (g: the curvature of the tangent, not of a reciprocal — see above)
                    const float tg0 = vget(FS.tg[k], c);
                    const float weight = __builtin_fminf(__builtin_fmaxf(2.5f * tg0, 0.1f), 2.0f);
                    ra = __builtin_fmaxf(ra, __builtin_fabsf(vget(FD.tg[k], c)) * (float)TS0 *
                                                 __builtin_amdgcn_rcpf(tg0) * weight);
                }
                rg = __builtin_fmaxf(rg, __builtin_fabsf((vget(xe.ap[k], c) - vget(xs.ap[k], c)) *
                                                         (vget(xe.mu[k], c) - vget(xs.mu[k], c))) * (to32 * to32));
                rg = __builtin_fmaxf(rg, __builtin_fabsf((vget(xe.tb[k], c) - vget(xs.tb[k], c)) * to32 *
                                                         vget(FD.g[k], c) * (float)TS0));
            }
        if constexpr (SU) {
            ra = __builtin_fmaxf(ra, __builtin_fabsf(FD.oml) * (float)TS0 * __builtin_amdgcn_rcpf(FS.oml));
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int c = 0; c < W; ++c)
                    ra = __builtin_fmaxf(ra, __builtin_fabsf(vget(FD.om[k], c)) * (float)TS0 *
                                                 __builtin_amdgcn_rcpf(vget(FS.om[k], c)));
        }
        // halvings needed: r / 2^s <= 2^-9.5 (error ~ r^2 / 16), |.| / 4 / 4^s <= 2^-20
        const int la = __builtin_amdgcn_frexp_expf(ra * GRAIL_FAST_A_SCALE);
        // (the second tier serves voices of any sharpness, whose resonances multiply what the amplitudes are off by: it keeps
        // the bound of 2^-22 — a voice of sharpness 195 deviates by 22.5 * 2^-23 with it and by 45.7 with 2^-20)
        const int lg = (__builtin_amdgcn_frexp_expf(rg * (MID ? 1048576.0f : GRAIL_FAST_G_SCALE)) + 1) >> 1;
        int level = la > lg ? la : lg;
        level = level < 0 ? 0 : level;
        if (!(ra == ra) || !(rg == rg)) level = 5;                                 // NaN: not here
        level = level > 5 ? 5 : level;
#ifdef GRAIL_FAST_FORCE_LEVEL0
        level = 0;
#endif
        // the L lanes of an utterance hold different formants: they take the largest of their levels (they run
        // in lockstep — the per-sample sum goes down the lanes — and all of them begin anew together)
#pragma unroll
        for (int m = 1; m < L; m <<= 1) {
            const int o = __shfl_xor(level, m);
            level = o > level ? o : level;
        }
        return level;
    };
    // the lane can render in tolerance mode at all: its segment pair inside the safe window, and pitch < 1/2 (fast_pair's
    // polyBLEP needs the head and tail tests to exclude each other)
    auto fast_lane_ok = [&]() __attribute__((always_inline)) -> bool {
        return !done & quiet_ok & (dt > 0.0f) &
               (__builtin_fmaxf(X.frequency, Y.frequency) + __builtin_fabsf(d_freq) < 0.5f);
    };
    // How many further steps from the state (c, p, n_done) — the clock and the jitter phase of the sample stepped last,
    // the samples rendered so far — are certainly free of events of this lane: the clock stays >= 0 (no segment advance,
    // :864), the noise phase stays <= 1 (no wrap, :245 / :294), the row and this launch's share of it have room, and — a
    // lane on the flat side of the kink of alpha = min(clk / blend_length, 1) — the quotient stays above one.  Step k
    // has the clock c - k dt.  The serial f32 clock strays from that line by up to half an ulp of itself per step, always
    // the same way inside a binade: next to dt that is nothing where the answer is small (a clock of a few dt), and where
    // the clock is compared with the blend length (seconds, possibly) 17 ulp cover the 33 steps a sub-tile can ask about:
    // that much and a quarter step are taken off (the safe side: a sub-tile that ends early costs a slow sample).  0 .. 127.
    auto fast_horizon = [&](const float c, const float p, const uint32_t n_done, const bool flat) __attribute__((always_inline)) -> int {
        const float rdt = __builtin_amdgcn_rcpf(dt);
        float e = c * rdt - 0.01f;
        e = __builtin_fminf(e, (1.0f - p) * __builtin_amdgcn_rcpf(jinc) - 0.01f);      // (jinc = 0: never; NaN is ignored by min)
        if (flat & !silent_pair) {
            const float stray = 17.0f * __builtin_ldexpf(1.0f, __builtin_amdgcn_frexp_expf(c) - 24) * rdt;
            e = __builtin_fminf(e, (c - blend_length) * rdt - (0.25f + stray));
        }
        const int h = e >= 127.0f ? 127 : (e > 0.0f ? (int)e : 0);                      // (NaN: 0)
        const uint32_t room = room_end > n_done ? room_end - n_done : 0u;
        return room < (uint32_t)h ? (int)room : h;
    };
    // the lane's sub-tile length by its level, and how far the next point of its grid is from step t of the tile.
    // Level 5 — the lane's parameters move faster than the line through two samples two apart can follow — keeps the
    // sub-tiles of two samples but takes BOTH from their own evaluation: the start afresh, the slope towards the second
    // sample (fast_refresh, fast_restart); its end value is never used.
    auto fast_grid_left = [&](const int t) __attribute__((always_inline)) -> int {
        const int tsl = FAST_TS0 >> (fast_shift > 4 ? 4 : fast_shift);
        return tsl - (t & (tsl - 1));
    };
    // ---- the per-utterance chain of samples tc, tc+1: exact (see scalar_packed_steps).  Advances
    // clk, jphase and phase; returns the phases before the two samples and their pitch.
    // CLAMP = false: the caller has shown that clk / blend_length <= 1 for every sample of the tile (the clock
    // only falls inside a calm tile), so min(ratio, 1) is the ratio itself.
    f2 chain_alpha = vsplat(0.0f, f2()), chain_jp = vsplat(0.0f, f2());   // MID: alpha and jitter phase of the pair just stepped
    auto chain_pair = [&](auto clamp_tag, f2 &PH, f2 &frequency) __attribute__((always_inline)) {
        constexpr bool CLAMP = decltype(clamp_tag)::value;
        const f2 one2 = vsplat(1.0f, f2());
        // a both-silent pair emits silent() itself (alpha = 1, :926): its reciprocal blend length is replaced
        // by +inf, the clock is positive in a calm tile, and min(+inf, 1) = 1 — no select per sample
        const float inv_bl = (!ANYBL && silent_pair) ? __builtin_inff() : inv_blend_length;
        const float clk0 = clk - dt, clk1 = clk0 - dt;                     // :861
        const float jp0 = jphase + jinc, jp1 = jp0 + jinc;                 // :242 / :291
        clk = clk1;
        jphase = jp1;
        f2 CLK, JP;
        CLK.x = clk0; CLK.y = clk1; JP.x = jp0; JP.y = jp1;
        f2 ratio = CLK * inv_bl;
        if constexpr (ANYBL) {
            const f2 rem = vfma(-blend_length * one2, ratio, CLK);
            const f2 quot = vfma(rem, inv_blend_length * one2, ratio);     // RN(clk / blend_length)
            ratio = blend_pow2 ? ratio : quot;
        }
        f2 alpha;                                                          // :899/:908/:917
        if constexpr (!CLAMP) {
            alpha = ratio;
        } else if constexpr (ANYBL) {
            alpha.x = silent_pair ? 1.0f : __builtin_fminf(ratio.x, 1.0f);
            alpha.y = silent_pair ? 1.0f : __builtin_fminf(ratio.y, 1.0f);
        } else {
            alpha.x = __builtin_fminf(ratio.x, 1.0f);
            alpha.y = __builtin_fminf(ratio.y, 1.0f);
        }
        const f2 oma = 1.0f - alpha;
        const f2 jomp = 1.0f - JP;
        if constexpr (MID) {
            chain_alpha = alpha;
            chain_jp = JP;
        }
        frequency = X.frequency * oma + Y.frequency * alpha;               // :404-414
        const f2 n_freq = fn_cur * jomp + fn_next * JP;                    // :254
        frequency = frequency + n_freq * d_freq;                           // :763
        // :520-525  `p += f; if p >= 1 { p -= 1 }` is fract(p + f) for 0 <= p < 1, 0 < f <= 1: both
        // branches are exact (x - 1 for x in [1, 2) loses nothing)
        const float ph0 = phase;
        const float ph1 = __builtin_amdgcn_fractf(ph0 + frequency.x);
        phase = __builtin_amdgcn_fractf(ph1 + frequency.y);
        PH.x = ph0; PH.y = ph1;
    };
    // A lane between two sub-tiles (fast_sub_left == 0, FS holds the values of its next sample, step t of the tile) takes
    // new slopes — BEFORE that sample is stepped: to the next point of its grid, or as far as its next samples are
    // certainly free of events of its own (fast_horizon), whichever is nearer.  If not even the next sample is — an
    // event of the lane is due — nothing happens here: the lane takes a slow sample and begins anew behind the event
    // (fast_restart).  A lane that follows every sample by itself (level 5) asks the guard again at every point of the
    // 32-sample grid, the same way.
    auto fast_refresh = [&](auto su_tag, const int t) __attribute__((always_inline)) {
        const bool flat = fast_flat_at(clk);            // the regime of the sample stepped last: that of the next ones, or none of them is free
        const int hz = fast_horizon(clk, jphase, n_out, flat);
        const int n_grid = fast_grid_left(t);
        int n = n_grid < hz ? n_grid : hz;
        if (fast_shift >= 5 && (t & (FAST_TS0 - 1)) == 0) n = 0;
        if (n >= 1) {
            FastAux xe;
            const bool own = fast_shift >= 5;            // (level 5: both samples from their own evaluation)
            if (own) fast_endpoint(su_tag, std::false_type(), 1.0f, 1.0f, flat, FS, xe);
            // (1 / n by IEEE division: exactly 2^-k for the sub-tiles on the grid)
            fast_endpoint(su_tag, std::true_type(), own ? 2.0f : (float)(n + 1), own ? 1.0f : 1.0f / (float)n, flat, FD, xe);
            fast_sub_left = n;
            fast_sub_len = (float)n;
            FTI.x = 0.0f; FTI.y = 1.0f;
        }
    };
    // A lane begins anew AT the sample it has just stepped (step t of the tile; clk, jphase, n_out are that sample's):
    // behind an event of its own — the segment pair, the noises or the regime of alpha are new — or wherever it has no run.
    // The values at this sample, the slopes towards the next point of the 32-sample grid or as far as the regime reaches,
    // the error guard and with it the lane's sub-tile length.
    auto fast_restart = [&](auto su_tag, const int t) __attribute__((always_inline)) {
        constexpr bool SU = decltype(su_tag)::value;
        const bool flat = fast_flat_at(clk);
        const int reach = 1 + fast_horizon(clk, jphase, n_out, flat);     // this sample and the free ones behind it
        const int g0 = FAST_TS0 - (t & (FAST_TS0 - 1));
        const int far0 = g0 < reach ? g0 : reach;
        FastAux xs, xe;
        fast_endpoint(su_tag, std::false_type(), 0.0f, 1.0f, flat, FS, xs);
        int far = far0, n = far0;
        // (a loop so that the far end's code exists once: a second trip where the guard asks for a shorter sub-tile)
#pragma unroll 1
        for (int trip = 0; trip < 2; ++trip) {
            fast_endpoint(su_tag, std::true_type(), (float)far, 1.0f / (float)far, flat, FD, xe);
            if (trip == 1) break;
            fast_shift = fast_level(su_tag, xs, xe, (float)far0);
            const int n_grid = fast_grid_left(t);
            n = n_grid < reach ? n_grid : reach;
            const int far1 = fast_shift >= 5 ? 1 : n;     // (level 5: the slope towards the sub-tile's second sample)
            if (far1 == far) break;
            far = far1;
        }
        fast_sub_left = n;
        fast_sub_len = (float)n;
        FTI.x = 0.0f; FTI.y = 1.0f;
        fast_have = SU ? 1 : 0;
    };
    // the sub-tile's end is the next one's start: start + TS * slope (the end value the slopes were
    // made from, to within an ulp; every sub-tile's end is evaluated afresh, so nothing accumulates)
    auto fast_subtile_end = [&](auto su_tag) __attribute__((always_inline)) {
        constexpr bool SU = decltype(su_tag)::value;
        const float fts = fast_sub_len;
        fast_sub_left = 0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            FS.a1[k] = vfma(FD.a1[k], vsplat(fts, V()), FS.a1[k]);
            FS.tg[k] = vfma(FD.tg[k], vsplat(fts, V()), FS.tg[k]);
            FS.g[k] = vfma(FD.g[k], vsplat(fts, V()), FS.g[k]);
            FS.h[k] = vfma(FD.h[k], vsplat(fts, V()), FS.h[k]);
            FS.b[k] = vfma(FD.b[k], vsplat(fts, V()), FS.b[k]);
            if constexpr (!SU) FS.om[k] = vfma(FD.om[k], vsplat(fts, V()), FS.om[k]);
        }
        if constexpr (SU) FS.oml = __builtin_fmaf(FD.oml, fts, FS.oml);
    };
    // polyBLEP :503-517 of two samples without branches or selects: with d_h = f - p (> 0: the head test
    // p < f) and d_t = p - (1 - f) (> 0: the tail test p > 1 - f; never both), u = max(d_h, d_t, 0) / f
    // is 1 - t for the head (:505) and 1 + t for the tail (:509), and the correction is -u^2 or
    // +u^2 (:506, :510) — zero when neither test holds.  d_h - d_t = 1 - 2p = -(2p - 1): the sign
    // of the uncorrected saw says which.  Same tests as the reference, quotient by v_rcp.
    auto fast_saw = [&](const f2 PH, const f2 frequency) __attribute__((always_inline)) -> f2 {
        const f2 one2 = vsplat(1.0f, f2());
        const f2 omf = 1.0f - frequency;
        const f2 d_h = frequency - PH, d_t = PH - omf;
        f2 u;
        u.x = __builtin_fmaxf(__builtin_fmaxf(d_h.x, d_t.x), 0.0f);
        u.y = __builtin_fmaxf(__builtin_fmaxf(d_h.y, d_t.y), 0.0f);
        u = u * vrcp(frequency);
        const f2 saw_nb = vfma(vsplat(2.0f, f2()), PH, -one2);             // 2 p - 1
        f2 su;    // u with the sign of -saw_nb: + for the head (saw + u^2), - for the tail (saw - u^2)
        su.x = __uint_as_float((__float_as_uint(u.x) & 0x7FFFFFFFu) | (~__float_as_uint(saw_nb.x) & 0x80000000u));
        su.y = __uint_as_float((__float_as_uint(u.y) & 0x7FFFFFFFu) | (~__float_as_uint(saw_nb.y) & 0x80000000u));
        return vfma(su, u, saw_nb);                                        // :517
    };
    // the formants of NH samples tc .. (tc + NH - 1) of the lane, coefficients by interpolation at the positions FTI;
    // nz / nm: the carrier noise of the samples and noise - 1
    auto fast_formants = [&](auto su_tag, auto nh_tag, const int tc, const f2 saw2, const float nz0, const float nz1,
                             const float nm0, const float nm1_) __attribute__((always_inline)) {
        constexpr bool SU = decltype(su_tag)::value;
        constexpr int NH = decltype(nh_tag)::value;
        const f2 one2 = vsplat(1.0f, f2());
        f2 keep2 = one2, ksaw2 = one2;          // shared smoothness: 1 - k and k * saw of both samples
        if constexpr (SU) {
            const f2 k2 = vfma(vsplat(FD.oml, f2()), FTI, vsplat(FS.oml, f2()));
            keep2 = 1.0f - k2;
            ksaw2 = k2 * saw2;
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float ti = vget(FTI, h), saw = vget(saw2, h);
            const V tiv = vsplat(ti, V());
            const V nms = vsplat((h == 0 ? nz0 : nz1) - saw, V());
            const V nm1 = vsplat(h == 0 ? nm0 : nm1_, V());
            const V sawv = vsplat(saw, V());
            V acc = vsplat(0.0f, V());
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const V b = vfma(FD.b[k], tiv, FS.b[k]);
                if constexpr (SU) {             // :531 + :538:  a' = (1-k) a + k saw + (k breath)(noise - saw)
                    st_a[k] = vfma(b, nms, vfma(vsplat(vget(keep2, h), V()), st_a[k], vsplat(vget(ksaw2, h), V())));
                } else {
                    const V nw = vfma(b, nms, sawv);                        // :531
                    const V oml_v = vfma(FD.om[k], tiv, FS.om[k]);
                    st_a[k] = vfma(oml_v, nw - st_a[k], st_a[k]);           // :538
                }
            }
            if constexpr (MID) {
                // this sample's coefficients as the reference has them, from its own blend weights (:899-903, :242)
                V a1x[NV], a2x[NV], a3x[NV];
                const float al_h = vget(chain_alpha, h), jp_h = vget(chain_jp, h);
                exact_band_pass_coeffs<NV, V>(X.freq, Y.freq, X.bw, Y.bw, ff_cur, ff_next, al_h, 1.0f - al_h, jp_h, 1.0f - jp_h,
                                              d_ffreq, a1x, a2x, a3x);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const V g_ = vfma(FD.g[k], tiv, FS.g[k]);
                    const V h_ = vfma(FD.h[k], tiv, FS.h[k]);
                    const V v0 = st_a[k] * vfma(h_, nm1, g_);                   // :544-550
                    const V v3 = v0 - st_c[k];                                  // :565
                    const V w1 = vfma(a2x[k], v3, a1x[k] * st_b[k]);            // :566
                    const V w2 = vfma(a3x[k], v3, vfma(a2x[k], st_b[k], st_c[k]));   // :567
                    st_b[k] = vfma(vsplat(2.0f, V()), w1, -st_b[k]);            // :570
                    st_c[k] = vfma(vsplat(2.0f, V()), w2, -st_c[k]);            // :571
                    acc = k == 0 ? w1 : acc + w1;        // (tree order; the first term needs no 0 +)
                }
            } else {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const V a1 = vfma(FD.a1[k], tiv, FS.a1[k]);
                const V tg = vfma(FD.tg[k], tiv, FS.tg[k]);
                const V g_ = vfma(FD.g[k], tiv, FS.g[k]);
                const V h_ = vfma(FD.h[k], tiv, FS.h[k]);
                const V v0 = st_a[k] * vfma(h_, nm1, g_);                   // :544-550
                const V v3 = v0 - st_c[k];                                  // :565
                const V w1 = a1 * vfma(tg, v3, st_b[k]);                    // :566  a1 b + a2 v3
                const V w2 = vfma(tg, w1, st_c[k]);                         // :567  c + a2 b + a3 v3
                st_b[k] = vfma(vsplat(2.0f, V()), w1, -st_b[k]);            // :570
                st_c[k] = vfma(vsplat(2.0f, V()), w2, -st_c[k]);            // :571
                acc = k == 0 ? w1 : acc + w1;            // (tree order; the first term needs no 0 +)
            }
            }
            float part = vget(acc, 0);
            if constexpr (W == 2) part = part + vget(acc, 1);
            const int t_ = tc + h;
            if constexpr (FOLD_IN_FLUSH) {
                // the flush folds NFA parked values per sample: this lane's partial sum, then zeros
                stage[(t_ * S + slot) * NFA + f0] = part;
#pragma unroll
                for (int i = 1; i < FPL; ++i) stage[(t_ * S + slot) * NFA + f0 + i] = 0.0f;
            } else {
                float tot = part;
#pragma unroll
                for (int step = 1; step < L; ++step) tot = dpp_from_lane_below(tot) + part;
                if (j == L - 1) stage[t_ * SP + slot] = tot * 0.5f;
            }
        }
        FTI = FTI + (float)NH;
    };
    // two samples tc, tc + 1 of the lane: the chain, polyBLEP, the formants with interpolated coefficients
    auto fast_pair = [&](auto su_tag, const int tc, const float nz0, const float nz1, const float nm0,
                         const float nm1_) __attribute__((always_inline)) {
        f2 PH, frequency;
        chain_pair(std::true_type(), PH, frequency);
        const f2 saw2 = fast_saw(PH, frequency);
        fast_formants(su_tag, std::integral_constant<int, 2>(), tc, saw2, nz0, nz1, nm0, nm1_);
    };
    // ---- the staged tile's rows to memory: row `slot` holds samples [base_, base_ + T), mine_ of them valid
    // (the general flush; the main loop below has a shortcut for the usual full tile of the lane kernels)
    auto flush_rows = [&](const uint32_t base_, const uint32_t mine_) __attribute__((always_inline)) {
        constexpr int ROW_LANES = T / 4;
        constexpr int ROWS_PER_IT = 64 / ROW_LANES;
        const int rl = lane % ROW_LANES;
        const int rr = lane / ROW_LANES;
        if (emit && j == L - 1) cnt[slot] = mine_;
        if constexpr (PIPE) __syncthreads();
        else wave_lds_sync();
        const int r_first = PIPE ? wave * ROWS_PER_IT : 0;
        constexpr int R_STEP = PIPE ? ROWS_PER_IT * WAVES : ROWS_PER_IT;
#pragma unroll 1
        for (int r0 = r_first; r0 < S; r0 += R_STEP) {
            const int r = r0 + rr;
            if (ROWS_PER_IT > S && r >= S) continue;
            const uint32_t c = cnt[r];
            const int t0 = rl * 4;
            if ((uint32_t)t0 < c) {
                const uint64_t at = (uint64_t)(A.perm ? rowid[r] : u0 + r) * A.out_stride + base_ + t0;
                auto sample_at = [&](const int tt) __attribute__((always_inline)) -> float {
                    if constexpr (FOLD_IN_FLUSH) {
                        // v1.sum() * 0.5: the left fold from 0.0 over formants 0..7  :574, :123-125
                        const float *p = stage + (tt * S + r) * NFA;
                        float run = 0.0f;
#pragma unroll
                        for (int f = 0; f < NFA; ++f) run = run + p[f];
                        if (NFA < NF) run = run + 0.0f;   // formants 5-8: literal +0.0 terms
                        return run * 0.5f;
                    } else {
                        return stage[tt * SP + r];
                    }
                };
                const float s0 = sample_at(t0 + 0);
                const float s1 = sample_at(t0 + 1);
                const float s2 = sample_at(t0 + 2);
                const float s3 = sample_at(t0 + 3);
                if (A.out_pcm16) {
                    // the WAV sink's `(x * i16::MAX as f32) as i16` (examples/cli.rs:49) on the way out
                    int16_t *dst = A.out_pcm16 + at;
                    const int p0 = pcm16_from_f32(s0), p1 = pcm16_from_f32(s1);
                    const int p2 = pcm16_from_f32(s2), p3 = pcm16_from_f32(s3);
                    if (vec16_ok && (uint32_t)(t0 + 4) <= c) {
                        *reinterpret_cast<uint2 *>(dst) =
                            make_uint2((uint32_t)(p0 & 0xFFFF) | ((uint32_t)p1 << 16),
                                       (uint32_t)(p2 & 0xFFFF) | ((uint32_t)p3 << 16));
                    } else {
                        dst[0] = (int16_t)p0;
                        if ((uint32_t)(t0 + 1) < c) dst[1] = (int16_t)p1;
                        if ((uint32_t)(t0 + 2) < c) dst[2] = (int16_t)p2;
                        if ((uint32_t)(t0 + 3) < c) dst[3] = (int16_t)p3;
                    }
                    continue;
                }
                float *dst = A.out + at;
                if (vec_ok && (uint32_t)(t0 + 4) <= c) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(s0, s1, s2, s3);
                } else {
                    dst[0] = s0;
                    if ((uint32_t)(t0 + 1) < c) dst[1] = s1;
                    if ((uint32_t)(t0 + 2) < c) dst[2] = s2;
                    if ((uint32_t)(t0 + 3) < c) dst[3] = s3;
                }
            }
        }
        if constexpr (PIPE) __syncthreads();     // the rendering wave may not park the next tile before all have read
        else wave_lds_sync();
    };

    // ---- FAST: one tile of T steps.  Every lane decides for itself (see BATCH INVARIANCE above).
    auto fast_render_tile = [&](auto) __attribute__((always_inline)) {   // (generic: instantiated by FAST kernels only)
        static_assert(T <= 64, "the horizon of a calm tile is written for T <= 64");
        PROF_ADD(8);     // (flush and everything else between two tiles)
        // lanes that will not render again in this launch (chain exhausted, row full, no utterance) ride along
        // in the tight loops: what they compute is never read and their sample count stands still.  A lane
        // that has PAUSED (stream quota, end of its chunk) keeps its state: it is not idle.
        const bool idle = done && !paused;
        // shared smoothness: all formants of the utterance, whichever of its L lanes holds them
        auto flavour_now = [&]() __attribute__((always_inline)) -> int {
            if constexpr (L > 1) {
                const uint64_t su_mask = __builtin_amdgcn_ballot_w64(smooth_uniform);
                return ((su_mask >> (lane & ~(L - 1))) & ((1ull << L) - 1ull)) == ((1ull << L) - 1ull) ? 1 : 0;
            } else {
                return smooth_uniform ? 1 : 0;
            }
        };
        int flavour = flavour_now();
        const bool ok0 = fast_lane_ok();
        // the carrier noise of the T steps, lane l taking step l (closed-form skip-ahead of the LCG :36-55), where every
        // rendering lane begins the tile in the same state (seed 0 in every utterance, :594, and lanes in step: all but
        // live streams whose utterances waited for their source at different times)
        const uint64_t rendering = __builtin_amdgcn_ballot_w64(!idle);
        if (rendering == 0) return;
        const uint32_t tile_seed = (uint32_t)__builtin_amdgcn_readlane((int)noise_seed, __builtin_ctzll(rendering));
        const bool seeds_agree = __builtin_amdgcn_ballot_w64(!idle & (noise_seed != tile_seed)) == 0;
        const uint32_t ahead = (uint32_t)(lane < T ? lane : T - 1) + 1u;
        const uint32_t sk = tile_seed * LCG_SKIP.mul[ahead] + LCG_SKIP.add[ahead];
        const float noise_of_lane = (__uint_as_float((sk >> 9) | 0x3F800000u) - 1.5f) * 2.0f;
        bool any_slow = false;
        PROF_ADD(9);
        // One loop, in which the wave either renders a RUN of plain samples or ONE slow sample.
        // PLAIN RUN: while every rendering lane is inside a sub-tile of its run the wave renders pairs with fast_pair
        // — tight loops without lane predicates, idle lanes riding along, as many samples at once as every lane's sub-tile
        // still holds; a lane whose sub-tile has ended takes new slopes at the top of the loop (fast_refresh: one end-point
        // evaluation under the lane's predicate).  Sub-tiles never reach across an event of their lane, so "inside a
        // sub-tile" is all there is to test.  SLOW SAMPLE: a lane whose next sample is not certainly free of events —
        // fast_refresh gave it no sub-tile — or that has no run sends the wave through one sample by the chain part of the
        // general step (the reference's control flow: a segment advance, a noise wrap, the end of the row happen here and
        // nowhere else), a new beginning for the lanes that need one (fast_restart, behind their event), and the formants
        // of all lanes in the one tolerance-mode body.  What the wave pays for an event of one lane is that one sample
        // and the lane's two end points.
        // All of it is decided from the lane's own state, and a sample's arithmetic is the same in a pair and alone.
        // SHARED: the tile's carrier noise is one sequence for all lanes (seeds_agree) and comes from noise_of_lane; the
        // lanes' own generator states are set where the wave takes a slow sample (seed_at)
        const float nm1_of_lane = noise_of_lane - 1.0f;
        auto plain_run = [&](auto su_tag, auto shared_tag, int &t, const int t_end) __attribute__((always_inline)) {
            constexpr bool SHARED = decltype(shared_tag)::value;
#pragma unroll 1
            for (; t < t_end; t += 2) {
                if constexpr (SHARED) {
                    const float nz0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, noise_of_lane), t));
                    const float nz1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, noise_of_lane), t + 1));
                    const float nm0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nm1_of_lane), t));
                    const float nm1_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nm1_of_lane), t + 1));
                    fast_pair(su_tag, t, nz0, nz1, nm0, nm1_);
                } else {
                    const float nz0 = lcg_f32(noise_seed), nz1 = lcg_f32(noise_seed);   // :528, the lane's own draws
                    fast_pair(su_tag, t, nz0, nz1, nz0 - 1.0f, nz1 - 1.0f);
                }
            }
        };
        // the generator state of step t of the tile (SHARED)
        auto seed_at = [&](const int t) __attribute__((always_inline)) -> uint32_t {
            return t == 0 ? tile_seed : (uint32_t)__builtin_amdgcn_readlane((int)sk, t - 1);
        };
        int t = 0;
        bool ok_lane = ok0;              // fast_lane_ok() and the flavour change in slow samples only
        // the plain runs of the lanes of one flavour: new slopes for a lane between two sub-tiles whose next samples are
        // certainly free of events, then as many samples as every rendering lane still has inside its sub-tile — as a
        // power of two (the sub-tiles sit on power-of-two grids): that many go through without a test
        auto plain_loop = [&](auto su_tag, int &t) __attribute__((always_inline)) {
            constexpr int FL = decltype(su_tag)::value ? 1 : 0;
            const bool idle_now = done && !paused;               // (neither this nor `has_run` changes inside the loop)
            const bool has_run = ok_lane & (fast_have == FL);
#pragma unroll 1
            while (T - t >= 2) {
                const bool need = has_run & (fast_sub_left == 0);
                if (__builtin_amdgcn_ballot_w64(need) != 0) {
                    PROF_CNT(12, 1);
                    if (need) fast_refresh(su_tag, t);
                }
                const int left = idle_now ? 64 : (has_run ? fast_sub_left : 0);
                if (__builtin_amdgcn_ballot_w64(left < 2) != 0) break;
                int m = 2;
                if (__builtin_amdgcn_ballot_w64(left < 4) == 0) {
                    m = 4;
                    if (__builtin_amdgcn_ballot_w64(left < 8) == 0) {
                        m = 8;
                        if (__builtin_amdgcn_ballot_w64(left < 16) == 0) m = __builtin_amdgcn_ballot_w64(left < 32) == 0 ? 32 : 16;
                    }
                }
                const int room_t = (T - t) & ~1;
                m = m < room_t ? m : room_t;
                const int t_end = t + m;
                if (seeds_agree) plain_run(su_tag, std::true_type(), t, t_end);
                else plain_run(su_tag, std::false_type(), t, t_end);
                PROF_CNT(10, m >> 1);
                fast_sub_left -= m;
                n_out += idle_now ? 0u : (uint32_t)m;
                if (__builtin_amdgcn_ballot_w64(!idle_now & (fast_sub_left == 0)) != 0) {
                    if (!idle_now & (fast_sub_left == 0)) fast_subtile_end(su_tag);
                }
            }
        };
#pragma unroll 1
        while (t < T) {
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;     // nobody renders any more in this launch
            // (a flavour none of the rendering lanes has a run of: its loop would leave at once)
            if (__builtin_amdgcn_ballot_w64(!done & ok_lane & (fast_have == 1)) != 0) plain_loop(std::true_type(), t);
            if (__builtin_amdgcn_ballot_w64(!done & ok_lane & (fast_have == 0)) != 0) plain_loop(std::false_type(), t);
            PROF_ADD(2);
            if (t >= T) break;
            // ---- one slow sample
            PROF_CNT(11, 1);
            ++general_steps;
            any_slow = true;
            if (seeds_agree) {
                const uint32_t s_ = seed_at(t);
                if (!done) noise_seed = s_;
            }
            // the chain part of the general step, every lane; the formant part of the same step for a lane outside the
            // safe window (before the step, or behind the advance it has just taken): the reference's arithmetic where
            // it has to be.  (Taken apart for every lane: the whole step in one piece at this place costs the kernel
            // several hundred bytes of scratch memory — the register allocator's doing, measured.)
            cv_live = 0;
            if (!done) general_step(t, std::integral_constant<int, 2>());
            const bool live = cv_live != 0;
            const bool ok_after = fast_lane_ok();
            const bool ok_post = live & ok_lane & ok_after;
            const bool direct = live & !(ok_lane & ok_after);
            if (__builtin_amdgcn_ballot_w64(direct) != 0) {
                if (direct) {
                    general_step(t, std::integral_constant<int, 3>());
                    fast_have = -1;
                }
            }
            ok_lane = ok_after;
            flavour = flavour_now();                                // (a segment advance may have changed it)
            PROF_ADD(7);
            // a new beginning behind the lane's event, or wherever it has no run
            const bool anew = ok_post & ((fast_have != flavour) | (fast_sub_left == 0));
            if (__builtin_amdgcn_ballot_w64(anew) != 0) {
                PROF_CNT(13, 1); PROF_CNT(16, __popcll(__builtin_amdgcn_ballot_w64(anew)));
                if (anew) {
                    if (flavour) fast_restart(std::true_type(), t);
                    else fast_restart(std::false_type(), t);
#ifdef GRAIL_FAST_PROF
                    prof_lane_levels += (unsigned long long)fast_shift;
#endif
                }
            }
            PROF_ADD(5);
            // the formants of the sample, every lane in the one body
            if (__builtin_amdgcn_ballot_w64(ok_post) != 0) {
                if (ok_post) {
                    f2 PH, frequency;
                    PH.x = cv_ph; PH.y = cv_ph;
                    frequency.x = cv_freq; frequency.y = cv_freq;
                    const f2 saw2 = fast_saw(PH, frequency);
                    if constexpr (MID) {
                        chain_alpha.x = cv_alpha; chain_alpha.y = cv_alpha;
                        chain_jp.x = jphase; chain_jp.y = jphase;
                    }
                    const float nm = cv_noise - 1.0f;
                    if (flavour) fast_formants(std::true_type(), std::integral_constant<int, 1>(), t, saw2, cv_noise, cv_noise, nm, nm);
                    else fast_formants(std::false_type(), std::integral_constant<int, 1>(), t, saw2, cv_noise, cv_noise, nm, nm);
                    fast_sub_left -= 1;
                    if (fast_sub_left == 0) {
                        if (flavour) fast_subtile_end(std::true_type());
                        else fast_subtile_end(std::false_type());
                    }
                }
            }
            ++t;
            PROF_ADD(6);
        }
        if (seeds_agree && t >= T) {
            const uint32_t s_ = seed_at(T);
            if (!done) noise_seed = s_;
        }
        if (!any_slow) { ++fast_tiles; PROF_CNT(15, 1); }
    };

    // ---- SPLIT: fast-forward the exact per-utterance chain to where this chunk's filters start
    uint32_t base0 = 0;
    uint32_t reset_at = 0;       // the tile at which this lane's filters start from zero state
    if constexpr (SPLIT) {
        // (a voice's phonemes decide its warm-up; a batch of caller-built elems brings its own)
        uint32_t w = slot_used ? (A.split_warmup != 0u ? A.split_warmup : VO.warmup) : 0u, w_max = w;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)w_max, m);
            w_max = o > w_max ? o : w_max;
        }
        w_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)w_max);
        base0 = chunk_lo > w_max ? chunk_lo - w_max : 0u;
        reset_at = chunk_lo > w ? chunk_lo - w : 0u;
        for (;;) {
            if (__builtin_amdgcn_ballot_w64(!done & (n_out < base0)) == 0) break;
            const bool calm = !done & quiet_ok & (dt > 0.0f) & (clk > (float)(T + 8) * dt) &
                              (jphase + (float)(T + 1) * jinc < 0.999f) & (n_out + (uint32_t)T <= base0);
            if (__builtin_amdgcn_ballot_w64(!(calm | done)) == 0) {
                // (the usual tile: no lane's alpha needs its clamp — the blend is still under way)
                const bool below_one = !silent_pair & ((clk - dt) * inv_blend_length <= 1.0f) & (ANYBL ? blend_pow2 : true);
                if (__builtin_amdgcn_ballot_w64(!(below_one | done)) == 0) {
#pragma unroll 4
                    for (int tc = 0; tc < T; tc += 2) {
                        f2 PH, frequency;
                        chain_pair(std::false_type(), PH, frequency);
                    }
                } else {
#pragma unroll 4
                    for (int tc = 0; tc < T; tc += 2) {
                        f2 PH, frequency;
                        chain_pair(std::true_type(), PH, frequency);
                    }
                }
                n_out += done ? 0u : (uint32_t)T;
            } else {
                // A tile in which some lane has an event, pair by pair (as the mixed tile of the rendering loop):
                // a lane without an event of its own in the pair takes the packed chain step, the others the
                // reference's control flow; the usual pairs — nobody has one — in a tight loop of their own.  The
                // lanes of the wave move in lockstep (two samples per pair), so n_out < base0 holds for all of them
                // until the tile ends.
                auto pair_calm = [&]() __attribute__((always_inline)) -> bool {
                    return !done & quiet_ok & (dt > 0.0f) & (clk > 2.5f * dt) & (jphase + 2.01f * jinc < 1.0f);
                };
                int t = 0;
#pragma unroll 1
                while (t < T) {
#pragma unroll 1
                    for (; t < T; t += 2) {
                        if (__builtin_amdgcn_ballot_w64(!(pair_calm() | done)) != 0) break;
                        f2 PH, frequency;
                        chain_pair(std::true_type(), PH, frequency);
                        n_out += done ? 0u : 2u;
                    }
                    if (t >= T) break;
                    if (pair_calm()) {
                        f2 PH, frequency;
                        chain_pair(std::true_type(), PH, frequency);
                        n_out += 2u;
                    } else {
                        general_step(t, std::true_type());
                        general_step(t + 1, std::true_type());
                    }
                    t += 2;
                }
            }
        }
        // the carrier noise state after n_out draws from seed 0 (:594): s -> 16807 s + 1 composed n_out times
        {
            uint32_t mul = 16807u, add = 1u, acc = 0u;
#pragma unroll 1
            for (int b = 0; b < 32; ++b) {
                if ((n_out >> b) & 1u) acc = acc * mul + add;
                add = add * (mul + 1u);
                mul = mul * mul;
            }
            noise_seed = acc;
        }
    }

    for (uint32_t base = base0;; base += T) {
        if constexpr (SPLIT) {
            // this lane's warm-up starts here (lanes of other voices may have started theirs earlier)
            if (chunk > 0u && base == reset_at) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    st_a[k] = vsplat(0.0f, st_a[k]);
                    st_b[k] = st_a[k];
                    st_c[k] = st_a[k];
                }
                fast_have = -1;
            }
        }
        int t = 0;
        if constexpr (FAST) {
            fast_render_tile(0);
            t = T;
        }
        // MIXED_RUNS: a tile in which some lane has an event still renders the samples between the events by the calm
        // tile's own loops (packed_run), as many at once as every lane is certain to stay without one; the tile's carrier
        // noise is then one sequence for all lanes (mixed_shared), drawn once as in a calm tile, and a lane's own generator
        // state is brought up to date where the wave takes single steps (mixed_stale)
        // (not the one-lane kernels: the eight-formant ones fill the register file as they are and the second copy of the
        // loop costs them a scratch segment; the four-formant ones lose 6 % of the headline — 43.6 instead of 40.9 ms, the
        // calm loop's registers — for nothing: 91.5 against 89.9 ms on the speech-like corpus; profiles/r05_mixed_runs.txt.
        // The pipelined workgroups take it too: their four waves render the run redundantly, as they do the single steps)
        constexpr bool MIXED_RUNS = GRAIL_MIXED_RUNS && !FAST && (!PIPE || GRAIL_MIXED_RUNS_PIPE) && GRAIL_SCALAR_PACK && (L > 1 || GRAIL_MIXED_RUNS_L1);
        bool mixed_shared = false, mixed_stale = false;
        uint32_t mixed_seed = 0u, mixed_sk = 0u;
        float mixed_noise = 0.0f;
        if constexpr (!FAST) PROF_ADD(8);
        while (t < T) {
            // a run of quiet steps: a tight inner loop, so the loop-carried state keeps its
            // registers from one sample to the next.  Two flavours of the same loop: every
            // formant vector live, or (all lanes agree) the upper half silent for this pair.
            // A calm tile: for every lane that is still rendering, the clock stays >= 0, the
            // jitter phase <= 1 and the row has room for the T steps of the tile.  Lanes that
            // will not render again in this launch (chain exhausted, row full, no utterance) ride
            // along: what they compute is never read and their sample count stands still, so
            // nothing of theirs is flushed.  clk >= m*dt
            // implies RN(clk - dt) >= (m - 1.01)*dt (RN is monotone), so clk > (T+8)*dt leaves
            // > 7*dt after T <= 64 steps; the phase grows by at most jinc*(1 + 2^-23) per step.
            bool calm_tile = false;
            int pipe_tiles = 1;                            // PIPE: calm tiles the pipeline runs through in one go
            const bool idle = STREAM ? finished : done;   // a paused stream lane resumes: not idle
            uint32_t tile_seed = 0u;                       // the carrier-noise state the tile starts from
            if (t == 0) {
                static_assert(T <= 64, "calm-tile margins are written for T <= 64");
                const uint64_t busy = __builtin_amdgcn_ballot_w64(!idle);
                if (busy != 0) {
                    tile_seed = (uint32_t)__builtin_amdgcn_readlane((int)noise_seed, __builtin_ctzll(busy));
                    bool calm = !done & quiet_ok & (dt > 0.0f) & (clk > (float)(T + 8) * dt) &
                                (jphase + (float)(T + 1) * jinc < 0.999f) &
                                (cap32 - n_out >= (uint32_t)T) & (noise_seed == tile_seed);
                    calm_tile = __builtin_amdgcn_ballot_w64(!(calm | idle)) == 0;
                    if constexpr (MIXED_RUNS) {
                        if (!calm_tile && __builtin_amdgcn_ballot_w64(!idle & (noise_seed != tile_seed)) == 0) {
                            const uint32_t ahead = (uint32_t)(lane < T ? lane : T - 1) + 1u;
                            mixed_shared = true;
                            mixed_seed = tile_seed;
                            mixed_sk = tile_seed * LCG_SKIP.mul[ahead] + LCG_SKIP.add[ahead];
                            mixed_noise = (__uint_as_float((mixed_sk >> 9) | 0x3F800000u) - 1.5f) * 2.0f;
                        }
                    }
                    if constexpr (PIPE) {
                        // how many calm tiles in a row (every wave of the workgroup finds the same number): the
                        // pipeline then runs through them without draining.  The margins of the single tile
                        // for N = k T steps: each step lowers the bound on the clock by at most 1.01 dt.
                        pipe_tiles = 1;
                        if (calm_tile) {
#pragma unroll 1
                            for (int k = 2; k <= PIPE_MAX_TILES; ++k) {
                                const float nsteps = (float)(k * T);
                                const bool ok = (clk > (nsteps * 1.0125f + 8.0f) * dt) &
                                                (jphase + (nsteps + 1.0f) * jinc < 0.999f) &
                                                (cap32 - n_out >= (uint32_t)(k * T));
                                if (__builtin_amdgcn_ballot_w64(!(ok | idle)) != 0) break;
                                pipe_tiles = k;
                            }
                        }
                    }
                }
            }
            // PIPE: `n_rounds` rounds of the pipeline from step t0 of the tile — the calm tile's three stages (chain wave two
            // rounds ahead, coefficient waves one, the rendering wave), one barrier per phase — for a stretch of a tile with an
            // event in which nobody has one (MIXED_RUNS below).  No tile boundary inside, so no flush; at the end every wave takes
            // over the clocks the chain wave arrived at, as after a calm tile.
            auto pipe_rounds = [&](const int t0, const int n_rounds, const float noise_of_lane) __attribute__((always_inline)) {
                if constexpr (PIPE) {
                    constexpr int SPR = 4 * QP;
                    constexpr int CHAIN_PAIRS = QP == 8 ? (L == 4 ? PIPE_CP8_L4 : PIPE_CP8_L8) : QP >= 4 ? QP / 2 : 0;
                    constexpr int PAIRS_PER_COEF_WAVE = QP >= 4 ? (2 * QP - CHAIN_PAIRS + 1) / 2 : QP;
                    constexpr int PAIRS_OF_LAST_WAVE = QP >= 4 ? 2 * QP - CHAIN_PAIRS - PAIRS_PER_COEF_WAVE : QP;
#pragma unroll 1
                    for (int ph_ = -2; ph_ < n_rounds; ++ph_) {
                        if (role == 1) {
                            const int m = ph_ + 2;
                            if (m < n_rounds) {
#pragma unroll
                                for (int g = 0; g < QP / 2; ++g) pipe_chain(chain_all[m & 1][g], noise_of_lane, t0 + SPR * m + 8 * g);
                            }
                            if constexpr (CHAIN_PAIRS > 0) {     // and the last pairs of the round before
                                const int mc = ph_ + 1;
                                if (mc >= 0 && mc < n_rounds) {
#pragma unroll
                                    for (int pair = 2 * QP - CHAIN_PAIRS; pair < 2 * QP; ++pair)
                                        pipe_coeffs(chain_all[mc & 1][pair / 4], pair % 4, ring_all[mc & 1][pair]);
                                }
                            }
                        } else if (role >= 2) {
                            const int m = ph_ + 1;
                            if (m >= 0 && m < n_rounds) {
#pragma unroll
                                for (int q = 0; q < PAIRS_PER_COEF_WAVE; ++q) {
                                    const int pair = QP >= 4 ? PAIRS_PER_COEF_WAVE * (role - 2) + q : 2 * q + (role - 2);
                                    if (PAIRS_OF_LAST_WAVE == PAIRS_PER_COEF_WAVE || q < PAIRS_OF_LAST_WAVE || role == 2)
                                        pipe_coeffs(chain_all[m & 1][pair / 4], pair % 4, ring_all[m & 1][pair]);
                                }
                            }
                        } else if (ph_ >= 0) {
#pragma unroll
                            for (int q = 0; q < 2 * QP; ++q) pipe_render(ring_all[ph_ & 1][q], t0 + SPR * ph_ + 2 * q);
                        }
                        __syncthreads();
                    }
                    if (role == 1) {
                        hand_all[0][lane] = clk;
                        hand_all[1][lane] = jphase;
                        hand_all[2][lane] = phase;
                    }
                    __syncthreads();
                    if (role != 1) {
                        clk = hand_all[0][lane];
                        jphase = hand_all[1][lane];
                        phase = hand_all[2][lane];
                    }
                }
            };
            // samples [t0, t1) of the tile by the calm tile's loops (MIXED_RUNS below): nobody has an event among them
            auto packed_run = [&](auto nlive_tag, auto su_tag, const int t0, const int t1,
                                  const float noise_of_lane) __attribute__((always_inline)) {
                // (PIPE: the four waves of the workgroup hold the same utterances and render the run redundantly, as they do
                // the single steps — the same values parked four times)
                if constexpr (W == 1 && NV == 1 && FOLD_IN_FLUSH) {
#pragma unroll 1
                    for (int tc = t0; tc < t1; tc += 8) time_packed_block(tc, noise_of_lane);
                } else if constexpr (L >= 4) {
#pragma unroll 1
                    for (int tc = t0; tc < t1; tc += 8) scalar_packed_block(nlive_tag, su_tag, tc, noise_of_lane);
                } else {
#pragma unroll 1
                    for (int tc = t0; tc < t1; tc += 2) {
                        const float nz0 = __builtin_bit_cast(
                            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, noise_of_lane), tc));
                        const float nz1 = __builtin_bit_cast(
                            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, noise_of_lane), tc + 1));
                        scalar_packed_steps(nlive_tag, su_tag, tc, nz0, nz1);
                    }
                }
            };
            auto quiet_run = [&](auto nlive_tag, auto su_tag) __attribute__((always_inline)) {
                if (calm_tile) {
                    // no lane can have an event before the tile ends: no per-step ballot.  The
                    // carrier noise of the T steps is drawn here, lane l taking step l (closed-form
                    // skip-ahead of the LCG :36-55; wrapping u32 arithmetic is exact).
                    const uint32_t ahead = (uint32_t)(lane < T ? lane : T - 1) + 1u;
                    uint32_t sk = tile_seed * LCG_SKIP.mul[ahead] + LCG_SKIP.add[ahead];
                    const float noise_of_lane = (__uint_as_float((sk >> 9) | 0x3F800000u) - 1.5f) * 2.0f;
                    // two steps per trip halve the loop overhead; with all four formant vectors
                    // live the doubled body no longer fits the register file (measured: slower)
                    constexpr int STEPS_PER_TRIP = 2;
                    static_assert(T % STEPS_PER_TRIP == 0, "whole trips");
                    if constexpr (PIPE) {
                        // Rounds of two sample pairs, three stages one round apart: in phase p wave 1
                        // writes the chain of round p+2, waves 2 and 3 turn the chain of round p+1 into
                        // coefficients (one pair each), wave 0 renders round p; one barrier per phase.
                        constexpr int SPR = 4 * QP;              // samples per round
                        constexpr int ROUNDS = T / SPR;
                        // who turns the round's 2 QP pairs into coefficients: the two coefficient waves and, for a few
                        // pairs, the chain wave, so that the three stages take about the same time
                        // (rounds of 32 samples, 16 pairs: with 4 pairs the chain wave was the slowest stage — 7.39 ms for
                        // config 2 against 6.51 with 2 and 7.26 with none; profiles/r03_pipe_waves.txt)
                        constexpr int CHAIN_PAIRS = QP == 8 ? (L == 4 ? PIPE_CP8_L4 : PIPE_CP8_L8) : QP >= 4 ? QP / 2 : 0;
                        // the first coefficient wave takes the odd pair, if there is one
                        constexpr int PAIRS_PER_COEF_WAVE = QP >= 4 ? (2 * QP - CHAIN_PAIRS + 1) / 2 : QP;
                        constexpr int PAIRS_OF_LAST_WAVE = QP >= 4 ? 2 * QP - CHAIN_PAIRS - PAIRS_PER_COEF_WAVE : QP;
                        static_assert(QP < 4 || PAIRS_OF_LAST_WAVE >= 1, "every coefficient wave has a pair");
                        static_assert(T % SPR == 0, "whole rounds");
                        // Consecutive calm tiles (pipe_tiles of them) go through without draining the pipeline:
                        // when the rendering wave has parked a tile all four waves flush it, then carry on.
                        // The chain wave is two rounds ahead: it draws the next tile's carrier noise itself.
                        uint32_t sk_chain = sk;
                        float noise_chain = noise_of_lane;
                        const int all_rounds = pipe_tiles * ROUNDS;
#pragma unroll 1
                        for (int ph_ = -2; ph_ < all_rounds; ++ph_) {
                            if (role == 1) {
                                const int m = ph_ + 2;
                                if (m < all_rounds) {
                                    const int ml = m % ROUNDS;
                                    if (ml == 0 && m > 0) {          // on to the next tile: its noise, T draws further
                                        const uint32_t seed_next = (uint32_t)__builtin_amdgcn_readlane((int)sk_chain, T - 1);
                                        sk_chain = seed_next * LCG_SKIP.mul[ahead] + LCG_SKIP.add[ahead];
                                        noise_chain = (__uint_as_float((sk_chain >> 9) | 0x3F800000u) - 1.5f) * 2.0f;
                                    }
#pragma unroll
                                    for (int g = 0; g < QP / 2; ++g) pipe_chain(chain_all[m & 1][g], noise_chain, SPR * ml + 8 * g);
                                }
                                if constexpr (CHAIN_PAIRS > 0) {     // and the last pairs of the round before
                                    const int mc = ph_ + 1;
                                    if (mc >= 0 && mc < all_rounds) {
#pragma unroll
                                        for (int pair = 2 * QP - CHAIN_PAIRS; pair < 2 * QP; ++pair)
                                            pipe_coeffs(chain_all[mc & 1][pair / 4], pair % 4, ring_all[mc & 1][pair]);
                                    }
                                }
                            } else if (role >= 2) {
                                const int m = ph_ + 1;
                                if (m >= 0 && m < all_rounds) {
#pragma unroll
                                    for (int q = 0; q < PAIRS_PER_COEF_WAVE; ++q) {
                                        const int pair = QP >= 4 ? PAIRS_PER_COEF_WAVE * (role - 2) + q : 2 * q + (role - 2);
                                        if (PAIRS_OF_LAST_WAVE == PAIRS_PER_COEF_WAVE || q < PAIRS_OF_LAST_WAVE || role == 2)
                                            pipe_coeffs(chain_all[m & 1][pair / 4], pair % 4, ring_all[m & 1][pair]);
                                    }
                                }
                            } else if (ph_ >= 0) {
#pragma unroll
                                for (int q = 0; q < 2 * QP; ++q)
                                    pipe_render(ring_all[ph_ & 1][q], SPR * (ph_ % ROUNDS) + 2 * q);
                            }
                            __syncthreads();
                            if (ph_ >= 0 && ph_ % ROUNDS == ROUNDS - 1 && ph_ != all_rounds - 1) {
                                // a tile inside the run is complete: what the main loop does after a calm tile
                                n_out += idle ? 0u : (uint32_t)T;
                                noise_seed = (uint32_t)__builtin_amdgcn_readlane((int)sk, T - 1);
                                sk = noise_seed * LCG_SKIP.mul[ahead] + LCG_SKIP.add[ahead];
                                flush_rows(base, n_out > base ? n_out - base : 0u);
                                base += T;
                            }
                        }
                        // every wave takes over the clocks the chain wave arrived at
                        if (role == 1) {
                            hand_all[0][lane] = clk;
                            hand_all[1][lane] = jphase;
                            hand_all[2][lane] = phase;
                        }
                        __syncthreads();
                        if (role != 1) {
                            clk = hand_all[0][lane];
                            jphase = hand_all[1][lane];
                            phase = hand_all[2][lane];
                        }
                    } else if constexpr (W == 1 && NV == 1 && FOLD_IN_FLUSH) {
                        static_assert(!(W == 1 && NV == 1 && FOLD_IN_FLUSH) || (L >= 4 && T % 8 == 0), "blocks of eight, quads");
#pragma unroll 1
                        for (int tc = 0; tc < T; tc += 8) time_packed_block(tc, noise_of_lane);
                    } else if constexpr (GRAIL_SCALAR_PACK && STEPS_PER_TRIP == 2 && L >= 4 && T % 8 == 0) {
#pragma unroll 1
                        for (int tc = 0; tc < T; tc += 8) scalar_packed_block(nlive_tag, su_tag, tc, noise_of_lane);
                    } else if constexpr (GRAIL_SCALAR_PACK && STEPS_PER_TRIP == 2) {
#pragma unroll 1
                        for (int tc = 0; tc < T; tc += 2) {
                            const float nz0 = __builtin_bit_cast(
                                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, noise_of_lane), tc));
                            const float nz1 = __builtin_bit_cast(
                                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, noise_of_lane), tc + 1));
                            scalar_packed_steps(nlive_tag, su_tag, tc, nz0, nz1);
                        }
                    } else {
#pragma unroll 1
                        for (int tc = 0; tc < T; tc += STEPS_PER_TRIP) {
#pragma unroll
                            for (int h = 0; h < STEPS_PER_TRIP; ++h) {
                                const float nz = __builtin_bit_cast(
                                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, noise_of_lane), tc + h));
                                quiet_step(nlive_tag, su_tag, std::true_type(), tc + h, clk - dt, jphase + jinc, nz);
                            }
                        }
                    }
                    t = T;
                    n_out += idle ? 0u : (uint32_t)T;
                    noise_seed = (uint32_t)__builtin_amdgcn_readlane((int)sk, T - 1);
                    return;
                }
                // A tile in which some lane has an event.  MIXED_RUNS: the samples between the events still go through
                // the calm tile's own loop, as many at once — 2, 4, ... 32 — as every lane is certain to stay without one:
                // a lane's event-free horizon in steps is min(clk / dt, (1 - jphase) / jinc, room in its row), here from
                // two v_rcp_f32 with the margins of the calm-tile test (clk > (N + 1) dt leaves > 0.6 dt after N <= 32
                // steps, each of which lowers the bound by at most 1.01 dt; the phase grows by at most jinc (1 + 2^-23) per
                // step).  Lanes that will not render again ride along as in a calm tile; a paused stream lane keeps its
                // state: it is not idle and has no horizon.  A NaN anywhere fails the >= tests: single steps.
                constexpr int RUN_STEP = L >= 4 ? 8 : 2;
                while (t < T) {
                    if constexpr (MIXED_RUNS) {
                        const bool idle_now = STREAM ? finished : done;
                        // (a lane that has paused, or whose segment pair needs the general step, has no horizon: single steps)
                        if (mixed_shared && T - t >= RUN_STEP &&
                            __builtin_amdgcn_ballot_w64(!(idle_now | (!done & quiet_ok))) == 0) {
                            const float by_clock = clk * __builtin_amdgcn_rcpf(dt) - 1.5f;
                            const float by_phase = (0.9999f - jphase) * __builtin_amdgcn_rcpf(jinc) - 0.5f;
                            float horizon = __builtin_fminf(__builtin_fminf(by_clock, by_phase), (float)(cap32 - n_out));
                            const bool lane_ok = !done & quiet_ok & (dt >= 0x1p-50f) & (n_out < cap32);
                            horizon = idle_now ? 64.0f : (lane_ok ? horizon : 0.0f);
                            // the wave's horizon: the smallest of the lanes' (max(NaN, 0) is 0; six DPP steps leave the
                            // minimum over the lanes in lane 63)
                            uint32_t steps = (uint32_t)__builtin_fminf(__builtin_fmaxf(horizon, 0.0f), 64.0f);
                            steps = umin_dpp<0x111, 0xF>(steps);    // row_shr:1
                            steps = umin_dpp<0x112, 0xF>(steps);    // row_shr:2
                            steps = umin_dpp<0x114, 0xF>(steps);    // row_shr:4
                            steps = umin_dpp<0x118, 0xF>(steps);    // row_shr:8
                            steps = umin_dpp<0x142, 0xA>(steps);    // row_bcast:15
                            steps = umin_dpp<0x143, 0xC>(steps);    // row_bcast:31
                            const int wave_steps = __builtin_amdgcn_readlane((int)steps, 63) & ~(RUN_STEP - 1);
                            if (wave_steps > 0) {
                                const int room_t = (T - t) & ~(RUN_STEP - 1);
                                int m = wave_steps < room_t ? wave_steps : room_t;
                                if constexpr (PIPE) {
                                    // whole rounds go through the pipeline (every wave finds the same count), the rest of
                                    // the run in the next trip through the loop, redundantly in all four waves
                                    constexpr int SPR = 4 * QP;
                                    if (GRAIL_PIPE_PARTIAL && m >= SPR) {
                                        m = (m / SPR) * SPR;
                                        pipe_rounds(t, m / SPR, mixed_noise);
                                    } else {
                                        packed_run(nlive_tag, su_tag, t, t + m, mixed_noise);
                                    }
                                } else {
                                    packed_run(nlive_tag, su_tag, t, t + m, mixed_noise);
                                }
                                PROF_CNT(10, m >> 1);
                                n_out += idle_now ? 0u : (uint32_t)m;
                                t += m;
                                mixed_stale = true;
                                continue;
                            }
                        }
                        if (mixed_stale) {      // single steps draw from the lane's own generator
                            const uint32_t s_ = t == 0 ? mixed_seed : (uint32_t)__builtin_amdgcn_readlane((int)mixed_sk, t - 1);
                            if (!done) noise_seed = s_;
                            mixed_stale = false;
                        }
                    }
                    const float clk_next = clk - dt;
                    const float jphase_next = jphase + jinc;
                    // bitwise on purpose: no short-circuit, so no exec-mask regions
                    // ANYBL: a blend length that is not 2^k also sends a clk below the division
                    // window (2^-59, or zero) to the general step
                    const float clk_floor = (ANYBL && !blend_pow2) ? 0x1p-59f : 0.0f;
                    const bool eventful = !done & (!quiet_ok | (clk_next < clk_floor) |
                                                   (jphase_next > 1.0f) | (n_out >= cap32));
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(eventful) != 0, 0)) break;
                    quiet_step(nlive_tag, su_tag, std::false_type(), t, clk_next, jphase_next, 0.0f);
                    PROF_CNT(12, 1);
                    ++t;
                }
                if constexpr (MIXED_RUNS) {
                    if (mixed_stale && t >= T) {
                        const uint32_t s_ = (uint32_t)__builtin_amdgcn_readlane((int)mixed_sk, T - 1);
                        if (!done) noise_seed = s_;
                        mixed_stale = false;
                    }
                }
            };
            PROF_ADD(9);
            const bool all_su = __builtin_amdgcn_ballot_w64(!done & !smooth_uniform) == 0;
            typedef std::integral_constant<int, NV> FullTag;
            bool half = false;
            if constexpr (HALF && NV >= 2) {
                typedef std::integral_constant<int, NV / 2> HalfTag;
                half = __builtin_amdgcn_ballot_w64(!done & !upper_silent) == 0;
                if (half && all_su) quiet_run(HalfTag(), std::true_type());
                else if (half) quiet_run(HalfTag(), std::false_type());
            }
            if (!half) {
                if (all_su) quiet_run(FullTag(), std::true_type());
                else quiet_run(FullTag(), std::false_type());
            }
            if (calm_tile) { PROF_ADD(2); PROF_CNT(15, 1); PROF_CNT(16, all_su ? 1 : 0); PROF_CNT(17, half ? 1 : 0); }
            else { PROF_ADD(3); PROF_CNT(18, all_su ? 1 : 0); PROF_CNT(19, half ? 1 : 0); PROF_CNT(20, 1); }
            if (t < T) {
                ++general_steps;
                general_step(t, std::false_type());
                quiet_ok = pair_safe && (blend_pow2 || blend_div_ok);
                ++t;
                PROF_ADD(7); PROF_CNT(11, 1);
            }
        }

        // ---- flush the staged tile: row `slot` holds samples [base, base+T)
        // PIPE: the rendering wave parked the tile; all four waves (identical state, same decisions) flush a
        // share of its rows each instead of waiting for wave 0 to do it alone, between two workgroup barriers
        const uint32_t mine = n_out > base ? n_out - base : 0u;
        constexpr int ROW_LANES = T / 4;
        constexpr int ROWS_PER_IT = 64 / ROW_LANES;
        const int rl = lane % ROW_LANES;
        const int rr = lane / ROW_LANES;
        if constexpr (SPLIT) {
            // warm-up tiles: the filters are still converging, nothing is stored (the lanes of the chunk before
            // render these samples); the stage is per wave and the next tile simply overwrites it
            if (base < chunk_lo) {
                if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                continue;
            }
        }
        if constexpr (!FOLD_IN_FLUSH && ROWS_PER_IT <= S) {
            // the usual tile: every row of the wave received all T samples and the rows take 16-B stores.
            // No per-row conditions, so the LDS reads of all rows are in flight together (a lone wave has
            // nothing else to hide their latency behind) and the stores follow back to back.
            const bool all_full = __builtin_amdgcn_ballot_w64((j == L - 1) & (mine != (uint32_t)T)) == 0;
            // (not for the one-lane eight-formant kernels: they hold 256 VGPRs and AGPRs besides, and the extra
            // path cost their f32 rows 4 %; their i16 rows take the general loop below)
            constexpr bool PCM_FULL_TILE = !(L == 1 && NFA == NF);
            if (PCM_FULL_TILE && all_full && (A.out_pcm16 ? !vec16_ok : !vec_ok)) {
                // ... and rows that do not start 16-byte aligned (an out_stride that is not a multiple of 4 samples — the
                // rows' own length, 96 006, is the natural one): the same tile with 4-byte stores, lane rl of a row
                // taking samples rl, rl + T/4, ...: every store instruction writes runs of T/4 consecutive samples per row
                // (the general flush below cost such strides 14 % of the exact and 33 % of the fast headline kernel; not for
                // the one-lane eight-formant kernels, as for i16 rows: the extra path cost their aligned rows 8 %)
                wave_lds_sync();
                float w[S / ROWS_PER_IT][4];
#pragma unroll
                for (int i = 0; i < S / ROWS_PER_IT; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[i][q] = stage[(rl + q * ROW_LANES) * SP + i * ROWS_PER_IT + rr];
#pragma unroll
                for (int i = 0; i < S / ROWS_PER_IT; ++i) {
                    const uint64_t row = A.perm ? rowid[i * ROWS_PER_IT + rr] : u0 + i * ROWS_PER_IT + rr;
                    if (A.out_pcm16) {      // (2-byte stores, the WAV sink's conversion on the way out)
                        int16_t *dst = A.out_pcm16 + row * A.out_stride + base + rl;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[q * ROW_LANES] = (int16_t)pcm16_from_f32(w[i][q]);
                    } else {
                        float *dst = A.out + row * A.out_stride + base + rl;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[q * ROW_LANES] = w[i][q];
                    }
                }
                wave_lds_sync();
                if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                continue;
            }
            if (all_full && (A.out_pcm16 ? (PCM_FULL_TILE && vec16_ok) : vec_ok)) {
                wave_lds_sync();
                float4 v[S / ROWS_PER_IT];
#pragma unroll
                for (int i = 0; i < S / ROWS_PER_IT; ++i) {
                    const int r = i * ROWS_PER_IT + rr;
                    const int t0 = rl * 4;
                    v[i] = make_float4(stage[(t0 + 0) * SP + r], stage[(t0 + 1) * SP + r], stage[(t0 + 2) * SP + r],
                                       stage[(t0 + 3) * SP + r]);
                }
                if (PCM_FULL_TILE && A.out_pcm16) {
                    // the WAV sink's `(x * i16::MAX as f32) as i16` (examples/cli.rs:49) on the way out: 8-byte stores
#pragma unroll
                    for (int i = 0; i < S / ROWS_PER_IT; ++i) {
                        const int p0 = pcm16_from_f32(v[i].x), p1 = pcm16_from_f32(v[i].y);
                        const int p2 = pcm16_from_f32(v[i].z), p3 = pcm16_from_f32(v[i].w);
                        const uint64_t row = A.perm ? rowid[i * ROWS_PER_IT + rr] : u0 + i * ROWS_PER_IT + rr;
                        *reinterpret_cast<uint2 *>(A.out_pcm16 + row * A.out_stride + base + rl * 4) =
                            make_uint2((uint32_t)(p0 & 0xFFFF) | ((uint32_t)p1 << 16),
                                       (uint32_t)(p2 & 0xFFFF) | ((uint32_t)p3 << 16));
                    }
                } else if (A.perm) {     // (two copies of the loop: the usual one without any LDS look-up)
#pragma unroll
                    for (int i = 0; i < S / ROWS_PER_IT; ++i)
                        *reinterpret_cast<float4 *>(A.out + (uint64_t)rowid[i * ROWS_PER_IT + rr] * A.out_stride + base +
                                                    rl * 4) = v[i];
                } else {
#pragma unroll
                    for (int i = 0; i < S / ROWS_PER_IT; ++i)
                        *reinterpret_cast<float4 *>(A.out + (uint64_t)(u0 + i * ROWS_PER_IT + rr) * A.out_stride + base +
                                                    rl * 4) = v[i];
                }
                wave_lds_sync();
                if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                continue;
            }
        }
        flush_rows(base, mine);
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
    }

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
