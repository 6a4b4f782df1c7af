// scan_kernels.hip — time-parallel synthesis for SMALL batches in fast (tolerance) arithmetic, gfx950.
//
// The lane-per-utterance kernels of synth_kernels.hip need tens of thousands of utterances to fill an
// MI355X; with a few hundred most SIMDs idle and the time per batch is the serial length of one
// utterance.  This kernel turns the mapping around: ONE WORKGROUP PER UTTERANCE, LANES = TIME.  A tile is
// up to 64 consecutive samples, one per lane.
//
//   wave 0 ("chain")      the per-utterance state of the reference, EXACT: Sequencer clock and segment
//                         advances (src/lib.rs:859-932), jitter phase, wraps and redraws (:240-306,
//                         :753-777), the pitch track, the carrier phase with its wrap (:520-525), the
//                         carrier-noise LCG (:36-55, closed-form skip-ahead).  The clock `clk -= dt` and the
//                         jitter phase `p += inc` are serial f32 accumulations; inside one binade they
//                         move by a constant quantum (the increment rounded to that binade's grid), so
//                         lane j gets its value as one fma, exactly, and a tile simply ends where the
//                         binade (or a tie case, or an event) ends.  The carrier phase is the one truly
//                         serial quantity: fract(p + f_j) handed down the lanes with DPP wave_shr:1.
//   waves 1..NP ("pairs") two formants each (packed f32).  Every lane evaluates its sample's filter
//                         coefficients directly (no interpolation), then the recurrences are solved for the
//                         whole tile by inclusive scans over the lanes: the one-pole low-pass (:538) is the
//                         affine map a -> (1-k) a + k x, the Cytomic SVF (:565-571) the 2x2 affine map
//                           [b'; c'] = [[2 a1 - 1, -2 a2], [2 a2, 1 - 2 a3]] [b; c] + v0 [2 a2; 2 a3],
//                         composed as (M2, u2) o (M1, u1) = (M2 M1, M2 u1 + u2) in six DPP steps
//                         (row_shr 1/2/4/8, row_bcast 15/31) — north_star's "first-order-section parallel
//                         scan".  The tile's last state is the next tile's start.
//   wave 0 again          adds the pairs' band-pass outputs (:574) and stores the tile.
//
// Three pipeline stages one tile apart, LDS buffers in between, one workgroup barrier per tile.
// Tolerance mode only (the scans reassociate the recurrences); the discontinuous state is the reference's
// to the bit, so lengths and every boundary / wrap / saw edge sit where the reference puts them.
// The host only sends batches here whose every parameter is inside the proven-safe window
// (grail_api.cpp scan_ok): no NaN / Inf special cases exist on this path.
#include <cstdio>

#include <type_traits>

#include "device_common.h"
#include "kernels.h"
#include "pcm16.h"

namespace grail {
namespace {

constexpr int TL = 64;   // samples per tile = lanes per wave

struct TileIn {
    float alpha[TL], jp[TL], saw[TL], nz[TL];
    int n;        // valid samples (lanes) of this tile
    int epoch;    // which parameter block applies
    int pad[2];
};

// what the pair waves need of the current segment pair and jitter period (written by the chain wave)
struct ParamBlock {
    float X[ELEM_FLOATS], Y[ELEM_FLOATS];      // emitted elem = X (1 - alpha) + Y alpha
    float ffc[NF], ffn[NF], fac[NF], fan[NF];  // formant-frequency / amplitude noise: current, next
    float d_ffreq, amp_scale;
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp(float old, float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ f2 dpp(f2 old, f2 x)
{
    f2 r;
    r.x = dpp<CTRL, ROW_MASK>(old.x, x.x);
    r.y = dpp<CTRL, ROW_MASK>(old.y, x.y);
    return r;
}

// one level of the inclusive scans: combine with the element CTRL lanes earlier in time
//   low-pass   (P, Q):  a -> P a + Q
//   band-pass  (M, U):  s -> M s + U
struct ScanElem {
    f2 P, Q;
    f2 m11, m12, m21, m22, u1, u2;
};
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_level(ScanElem &e)
{
    const f2 one = vsplat(1.0f, f2()), zero = vsplat(0.0f, f2());
    const f2 eP = dpp<CTRL, ROW_MASK>(one, e.P), eQ = dpp<CTRL, ROW_MASK>(zero, e.Q);
    const f2 e11 = dpp<CTRL, ROW_MASK>(one, e.m11), e12 = dpp<CTRL, ROW_MASK>(zero, e.m12);
    const f2 e21 = dpp<CTRL, ROW_MASK>(zero, e.m21), e22 = dpp<CTRL, ROW_MASK>(one, e.m22);
    const f2 eu1 = dpp<CTRL, ROW_MASK>(zero, e.u1), eu2 = dpp<CTRL, ROW_MASK>(zero, e.u2);
    e.Q = vfma(e.P, eQ, e.Q);
    e.P = e.P * eP;
    const f2 n11 = vfma(e.m12, e21, e.m11 * e11), n12 = vfma(e.m12, e22, e.m11 * e12);
    const f2 n21 = vfma(e.m22, e21, e.m21 * e11), n22 = vfma(e.m22, e22, e.m21 * e12);
    e.u1 = vfma(e.m12, eu2, vfma(e.m11, eu1, e.u1));
    e.u2 = vfma(e.m22, eu2, vfma(e.m21, eu1, e.u2));
    e.m11 = n11; e.m12 = n12; e.m21 = n21; e.m22 = n22;
}

template <int NP>   // pair waves: 2 (formants 5-8 proven dead, see live4_ok) or 4
__global__ __launch_bounds__(64 * (NP + 1)) void scan_kernel(const SynthArgs A)
{
    __shared__ TileIn tin[2];
    __shared__ ParamBlock par[2];
    __shared__ float yout[2][NP][TL];
    __shared__ int last_tile;                 // index of the utterance's last tile, known once the chain ends

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t u = A.perm ? A.perm[blockIdx.x] : blockIdx.x;   // longest utterances first (ragged batches)
    if (threadIdx.x == 0) last_tile = 0x7fffffff;
    __syncthreads();

    uint32_t vid = A.voice_ids ? A.voice_ids[u] : 0u;
    if (vid >= A.n_voices) vid = 0u;
    const DevVoice VO = A.voices[vid];
    const float *__restrict__ elems = A.elems;

    if (wave == 0) {
        // =============================== the chain wave ===============================
        // every lane carries the same per-utterance state (wave-uniform values in vector registers)
        uint32_t seg_pos = A.seg_offsets[u];
        const uint32_t seg_end = A.seg_offsets[u + 1];
        Seg cur, nxt;
        cur.some = false; cur.elem = -1; cur.length = 0.0f; cur.blend_length = 1.0f; cur.frequency = 0.0f;
        nxt = cur;
        float clk = 0.0f;
        const float dt = 1.0f / VO.sample_rate;                     // :944
        float xf = 0.25f, yf = 0.25f;                               // pitch of X and Y
        float inv_bl = 1.0f;
        bool silent_pair = true;
        int x_row = -1, y_row = -1;                                 // table rows behind X and Y (-1: silent())
        bool x_mute = false, y_mute = false;                        // copy_silent(): amplitudes zeroed

        uint32_t seed = A.seeds ? A.seeds[u] : 0u;                  // IntoJitter::jitter :786-797
        float fn_cur = lcg_f32(seed), fn_next = lcg_f32(seed);
        uint32_t fn_state = seed;
        // lane i < 8 holds formant i's noise values; lanes 8.. hold copies (i & 7)
        float ff_cur = 0.0f, ff_next = 0.0f, fa_cur = 0.0f, fa_next = 0.0f;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const float c0 = lcg_f32(seed), n0 = lcg_f32(seed);
            if ((lane & 7) == i) { ff_cur = c0; ff_next = n0; }
        }
        uint32_t ff_state = seed;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const float c0 = lcg_f32(seed), n0 = lcg_f32(seed);
            if ((lane & 7) == i) { fa_cur = c0; fa_next = n0; }
        }
        uint32_t fa_state = seed;
        float jphase = 0.0f;
        const float jinc = VO.jitter_frequency;
        const float d_freq = VO.jitter_delta_frequency;

        float phase = 0.0f;
        uint32_t noise_seed = 0u;                                   // :594
        const uint32_t skip_mul = LCG_SKIP.mul[lane + 1], skip_add = LCG_SKIP.add[lane + 1];

        const uint64_t cap = A.cap;
        const uint32_t cap32 = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
        uint32_t n_out = 0;
        bool finished = false, truncated = false;
        int epoch = 0;
        bool params_dirty = true;
        int hist_n[2] = {0, 0};
        uint32_t hist_at[2] = {0u, 0u};

        for (int step = 0;; ++step) {
            // ---- stage 3: add up and store the tile the pair waves finished in the previous step
            if (step >= 2 && step - 2 <= last_tile) {
                const int b = (step - 2) & 1;
                float y = 0.0f;
#pragma unroll
                for (int w = 0; w < NP; ++w) y += yout[b][w][lane];
                if (lane < hist_n[b]) {
                    const uint64_t at = (uint64_t)u * A.out_stride + hist_at[b] + (uint32_t)lane;
                    const float sample = y * 0.5f;                  // :574
                    if (A.out_pcm16) A.out_pcm16[at] = (int16_t)pcm16_from_f32(sample);
                    else A.out[at] = sample;
                }
            }
            // ---- stage 1: the next tile of the chain
            if (!finished) {
                TileIn &ti = tin[step & 1];
                // the first sample of the tile: the reference's own control flow
                float clk_first = clk - dt;                                     // :861
                if (clk_first < 0.0f) {                                         // :864
                    if (cur.some && nxt.some) {                                 // :868
                        cur = nxt;
                        fetch_seg(nxt, A.segs, seg_pos, seg_end, true, VO.elem_base);
                        clk_first += cur.length;                                // :873
                    } else if (!cur.some && !nxt.some) {                        // :876
                        fetch_seg(cur, A.segs, seg_pos, seg_end, true, VO.elem_base);
                        fetch_seg(nxt, A.segs, seg_pos, seg_end, true, VO.elem_base);
                        if (cur.some) clk_first += cur.length;                  // :881-883
                    } else {
                        finished = true;                                        // :886
                    }
                    if (!finished && cur.some) {                                // the match at :891-931
                        const bool has_b = cur.elem >= 0, has_c = nxt.some && nxt.elem >= 0;
                        silent_pair = !has_b && !has_c;
                        inv_bl = 1.0f / cur.blend_length;                       // blend lengths are +-2^k here
                        if (has_b && has_c) { x_row = nxt.elem; y_row = cur.elem; xf = nxt.frequency; yf = cur.frequency; x_mute = y_mute = false; }
                        else if (has_b) { x_row = y_row = cur.elem; xf = yf = cur.frequency; x_mute = true; y_mute = false; }
                        else if (has_c) { x_row = y_row = nxt.elem; xf = yf = nxt.frequency; x_mute = false; y_mute = true; }
                        else { x_row = y_row = -1; xf = yf = 0.25f; x_mute = y_mute = false; }
                        params_dirty = true;
                    }
                }
                if (!cur.some) finished = true;                                 // :930
                if (!finished && n_out >= cap) { truncated = true; finished = true; }
                if (finished) {
                    if (lane == 0) last_tile = step - 1;
                } else {
                    float jp_first = jphase + jinc;                             // :242 / :291
                    if (jp_first > 1.0f) {                                      // :245 / :294
                        jp_first -= 1.0f;
                        fn_cur = fn_next;
                        fn_next = lcg_f32(fn_state);
                        ff_cur = ff_next;
                        fa_cur = fa_next;
                        uint32_t s1 = ff_state, s2 = fa_state;
#pragma unroll
                        for (int i = 0; i < NF; ++i) {                          // from_func order :301
                            const float r1 = lcg_f32(s1), r2 = lcg_f32(s2);
                            if ((lane & 7) == i) { ff_next = r1; fa_next = r2; }
                        }
                        ff_state = s1;
                        fa_state = s2;
                        params_dirty = true;
                    }
                    if (params_dirty) {
                        ++epoch;
                        ParamBlock &pb = par[epoch & 1];
                        if (lane < ELEM_FLOATS) {
                            // SynthesisElem::silent() :367-377, copy_silent() :454-459
                            const float sil = lane == 0 ? 0.25f : (lane < F_BREATH ? 0.25f : 0.0f);
                            float xv = x_row >= 0 ? elems[(size_t)x_row * ELEM_FLOATS + lane] : sil;
                            float yv = y_row >= 0 ? elems[(size_t)y_row * ELEM_FLOATS + lane] : sil;
                            if (lane >= F_AMP) { xv = x_mute ? 0.0f : xv; yv = y_mute ? 0.0f : yv; }
                            pb.X[lane] = xv;
                            pb.Y[lane] = yv;
                        }
                        if (lane < NF) { pb.ffc[lane] = ff_cur; pb.ffn[lane] = ff_next; pb.fac[lane] = fa_cur; pb.fan[lane] = fa_next; }
                        if (lane == 0) { pb.d_ffreq = VO.jitter_delta_formant_frequency; pb.amp_scale = 0.5f * VO.jitter_delta_amplitude; }
                        params_dirty = false;
                    }
                    // ---- the clock and the jitter phase of samples 1.. by closed form.  From a known value v0
                    // (lane b) the next two are plain serial steps v1, v2; from there on the sequence moves by
                    // the quantum q = v1 - v2 as long as the values stay in the binade of v2 (RN(v - d) =
                    // v - RN_grid(d) when v lies on the result's grid) and d is not exactly half-way between two
                    // grid points.  Where the binade ends the same construction starts again from the last
                    // exact lane (up to two more times per tile), so tiles are cut by events, not by binades.
                    // `step` = -dt for the clock, +jinc for the jitter phase; `need_same_start`: an increasing
                    // sequence is regular only if v1 already lies in v2's binade (else v1 is off v2's grid).
                    auto extend = [&](float &v, int &nv, const float step, const bool need_same_start) __attribute__((always_inline)) {
                        // lanes < nv hold exact values; make lanes >= nv exact as far as one binade reaches
                        const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), nv - 1));
                        const float v1 = v0 + step, v2 = v1 + step, q = v2 - v1;
                        const int rel = lane - nv;                       // 0: v1, 1: v2, k >= 2: v2 + (k - 1) q
                        const float cand = rel <= 0 ? v1 : rel == 1 ? v2 : __builtin_fmaf((float)(rel - 1), q, v2);
                        const uint32_t e2 = __float_as_uint(v2) >> 23;
                        const float ulp = __uint_as_float(e2 > 23u ? (e2 - 23u) << 23 : 0u);
                        const bool regular = (__builtin_fabsf(step - q) != 0.5f * ulp) && e2 > 24u &&
                                             (!need_same_start || (__float_as_uint(v1) >> 23) == e2);
                        const bool good = lane < nv || rel <= 1 || (regular && (__float_as_uint(cand) >> 23) == e2);
                        v = lane >= nv ? cand : v;
                        const uint64_t bad_ = ~__builtin_amdgcn_ballot_w64(good);
                        nv = bad_ ? __builtin_ctzll(bad_) : TL;
                    };
                    float cj = clk_first, pj = jp_first;
                    int nc = 1, np_ = 1;
#pragma unroll 1
                    for (int pass = 0; pass < 3 && (nc < TL || np_ < TL); ++pass) {
                        if (nc < TL) extend(cj, nc, -dt, false);
                        if (np_ < TL) extend(pj, np_, jinc, true);
                    }
                    // the tile ends before the first event: a clock below zero (:864) or a phase above one (:245)
                    const bool ok = lane == 0 || (lane < nc && lane < np_ && cj >= 0.0f && pj <= 1.0f);
                    const uint64_t bad = ~__builtin_amdgcn_ballot_w64(ok);
                    int n = bad ? __builtin_ctzll(bad) : TL;
                    const uint32_t room = cap32 - n_out;
                    n = (uint32_t)n > room ? (int)room : n;
                    // ---- per lane: alpha, the pitch (exact: :404-414, :254, :763)
                    float alpha = __builtin_fminf(cj * inv_bl, 1.0f);           // :899/:908/:917
                    alpha = silent_pair ? 1.0f : alpha;
                    const float oma = 1.0f - alpha, jomp = 1.0f - pj;
                    float frequency = xf * oma + yf * alpha;
                    const float n_freq = fn_cur * jomp + fn_next * pj;
                    frequency = frequency + n_freq * d_freq;
                    // ---- the carrier phase: p_j = fract(p_{j-1} + f_{j-1}), exact (:520-525), handed down
                    // the lanes: after k rounds lanes <= k hold their phase
                    // lane 0 has no lane below: the shifted-in value is 0 there (bound_ctrl) and its addend is
                    // the phase the tile starts from, fract(0 + phase) = phase.  Rounds beyond n - 1 only touch
                    // lanes >= n, so the trip count is rounded up to whole groups of eight.
                    const float f_below = dpp<0x138, 0xF>(0.0f, frequency);     // wave_shr:1
                    const float addend = lane == 0 ? phase : f_below;
                    float ph = phase;
                    for (int k = 1; k < n; k += 8) {
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            ph = __builtin_amdgcn_fractf(
                                __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ph), 0x138, 0xF, 0xF, true)) + addend);
                    }
                    // ---- saw with polyBLEP (:503-517), quotient by v_rcp (tolerance)
                    const bool head = ph < frequency, tail = ph > (1.0f - frequency);
                    const float tt = (head ? ph : ph - 1.0f) * __builtin_amdgcn_rcpf(frequency);
                    const float pb_ = head ? ((2.0f * tt - tt * tt) - 1.0f) : ((tt * tt + 2.0f * tt) + 1.0f);
                    const float saw = __builtin_fmaf(2.0f, ph, -1.0f) - ((head | tail) ? pb_ : 0.0f);
                    // ---- carrier noise :528: lane j is j + 1 draws after the tile's start state
                    const uint32_t sk = noise_seed * skip_mul + skip_add;
                    const float nz = (__uint_as_float((sk >> 9) | 0x3F800000u) - 1.5f) * 2.0f;
                    ti.alpha[lane] = alpha;
                    ti.jp[lane] = pj;
                    ti.saw[lane] = saw;
                    ti.nz[lane] = nz;
                    if (lane == 0) { ti.n = n; ti.epoch = epoch; }
                    // ---- carry the chain to the tile's end
                    const int last = n - 1;
                    clk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cj), last));
                    jphase = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pj), last));
                    const float ph_l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ph), last));
                    const float f_l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, frequency), last));
                    phase = __builtin_amdgcn_fractf(ph_l + f_l);
                    noise_seed = (uint32_t)__builtin_amdgcn_readlane((int)sk, last);
                    hist_n[step & 1] = n;
                    hist_at[step & 1] = n_out;
                    n_out += (uint32_t)n;
                }
            }
            __syncthreads();
            if (step >= 1 && step - 1 > last_tile) break;     // the tile stored above was the last one
        }
        if (lane == 0) {
            if (A.out_len) A.out_len[u] = n_out;
            if (truncated) atomicOr(A.truncated, 1u);
        }
        return;
    }

    // =============================== a pair wave: formants f0, f0 + 1 ===============================
    const int pw = wave - 1;
    const int f0 = 2 * pw;
    f2 a_in = vsplat(0.0f, f2()), b_in = a_in, c_in = a_in;       // filter states at the tile's start
    f2 Xf, Xb, Xs, Xr, Xt, Xa, Yf, Yb, Ys, Yr, Yt, Ya, ffc, ffn, fac, fan;
    Xf = Xb = Xs = Xr = Xt = Xa = Yf = Yb = Ys = Yr = Yt = Ya = ffc = ffn = fac = fan = a_in;
    float d_ffreq = 0.0f, amp_scale = 0.0f;
    int have_epoch = -1;
    const f2 one = vsplat(1.0f, f2()), zero = vsplat(0.0f, f2());
    const f2 five = vsplat(5.0f, f2()), m4 = vsplat(-4.0f, f2());
    for (int step = 0;; ++step) {
        if (step >= 1 && step - 1 <= last_tile) {
            const TileIn &ti = tin[(step - 1) & 1];
            const int n = ti.n;
            if (ti.epoch != have_epoch) {
                have_epoch = ti.epoch;
                const ParamBlock &pb = par[have_epoch & 1];
                auto two = [&](const float *p, int off) __attribute__((always_inline)) { f2 r; r.x = p[off + f0]; r.y = p[off + f0 + 1]; return r; };
                Xf = two(pb.X, F_FREQ); Xb = two(pb.X, F_BW); Xs = two(pb.X, F_SMOOTH);
                Xr = two(pb.X, F_BREATH); Xt = two(pb.X, F_TURB); Xa = two(pb.X, F_AMP);
                Yf = two(pb.Y, F_FREQ); Yb = two(pb.Y, F_BW); Ys = two(pb.Y, F_SMOOTH);
                Yr = two(pb.Y, F_BREATH); Yt = two(pb.Y, F_TURB); Ya = two(pb.Y, F_AMP);
                ffc = two(pb.ffc, 0); ffn = two(pb.ffn, 0); fac = two(pb.fac, 0); fan = two(pb.fan, 0);
                d_ffreq = pb.d_ffreq;
                amp_scale = pb.amp_scale;
            }
            const bool live = lane < n;
            const float alpha = ti.alpha[lane], jp = ti.jp[lane], saw = ti.saw[lane], nz = ti.nz[lane];
            const float oma = 1.0f - alpha, jomp = 1.0f - jp;
            const f2 al = vsplat(alpha, f2()), jpv = vsplat(jp, f2());
            // SynthesisElem::blend :404-414 and Jitter::next :763-773 with fused multiply-adds
            f2 ef = vfma(Yf, al, Xf * oma);
            const f2 eb = vfma(Yb, al, Xb * oma);
            const f2 es = vfma(Ys, al, Xs * oma);
            const f2 er = vfma(Yr, al, Xr * oma);
            const f2 et = vfma(Yt, al, Xt * oma);
            const f2 ea = vfma(Ya, al, Xa * oma);
            const f2 nff = vfma(ffn, jpv, ffc * jomp);
            const f2 nfa = vfma(fan, jpv, fac * jomp);
            ef = vfma(nff, vsplat(d_ffreq, f2()), ef);
            const f2 G = ea * vfma(nfa + 1.0f, vsplat(-amp_scale, f2()), one);
            const f2 H = et * G;
            // tan_approx :63-70, k :558, a1 :560 — reciprocals by v_rcp + one Newton step
            const f2 omx = 1.0f - ef, xph = ef + 0.5f, hmx = 0.5f - ef;
            const f2 ox = omx * ef, ph_ = xph * hmx;
            const f2 num = ox * vfma(m4, ph_, five);
            const f2 den = (xph * vfma(m4, ox, five)) * hmx;
            f2 rd = vrcp(den), rx = vrcp(ef);
            rd = vfma(vfma(-den, rd, one), rd, rd);
            rx = vfma(vfma(-ef, rx, one), rx, rx);
            const f2 tg = num * rd;
            const f2 kq = eb * rx;
            const f2 d3 = vfma(tg, tg + kq, one);
            f2 a1 = vrcp(d3);
            a1 = vfma(vfma(-d3, a1, one), a1, a1);
            const f2 oml = 1.0f - exp_approx(es);                          // :535
            const f2 nw = vfma(er, vsplat(nz - saw, f2()), vsplat(saw, f2()));   // :531
            // ---- the one-pole low-pass :538 as an affine scan
            ScanElem e;
            e.P = live ? 1.0f - oml : one;
            e.Q = live ? oml * nw : zero;
            // the band-pass needs v0 = a (G + H (noise - 1)) :544-550, i.e. the low-pass result first:
            // scan the low-pass alone, then build the band-pass elements and scan those
            {
                f2 P = e.P, Q = e.Q;
                auto lvl = [&](auto ctrl, auto rmask) __attribute__((always_inline)) {
                    constexpr int C = decltype(ctrl)::value, R = decltype(rmask)::value;
                    const f2 eP = dpp<C, R>(one, P), eQ = dpp<C, R>(zero, Q);
                    Q = vfma(P, eQ, Q);
                    P = P * eP;
                };
                lvl(std::integral_constant<int, 0x111>(), std::integral_constant<int, 0xF>());
                lvl(std::integral_constant<int, 0x112>(), std::integral_constant<int, 0xF>());
                lvl(std::integral_constant<int, 0x114>(), std::integral_constant<int, 0xF>());
                lvl(std::integral_constant<int, 0x118>(), std::integral_constant<int, 0xF>());
                lvl(std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xA>());
                lvl(std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xC>());
                e.P = P;
                e.Q = Q;
            }
            const f2 a_t = vfma(e.P, a_in, e.Q);                           // low-pass state after this sample
            const f2 v0 = a_t * vfma(H, vsplat(nz - 1.0f, f2()), G);       // :544-550
            // ---- the band-pass :560-571 as a 2x2 affine scan
            const f2 A2 = a1 * tg;                                         // a2 = g a1
            const f2 A3 = A2 * tg;                                         // a3 = g a2
            e.m11 = live ? vfma(vsplat(2.0f, f2()), a1, -one) : one;
            e.m12 = live ? -2.0f * A2 : zero;
            e.m21 = live ? 2.0f * A2 : zero;
            e.m22 = live ? vfma(vsplat(-2.0f, f2()), A3, one) : one;
            e.u1 = live ? (2.0f * A2) * v0 : zero;
            e.u2 = live ? (2.0f * A3) * v0 : zero;
            e.P = one;
            e.Q = zero;
            scan_level<0x111, 0xF>(e);
            scan_level<0x112, 0xF>(e);
            scan_level<0x114, 0xF>(e);
            scan_level<0x118, 0xF>(e);
            scan_level<0x142, 0xA>(e);
            scan_level<0x143, 0xC>(e);
            const f2 b_t = vfma(e.m12, c_in, vfma(e.m11, b_in, e.u1));    // states after this sample
            const f2 c_t = vfma(e.m22, c_in, vfma(e.m21, b_in, e.u2));
            // the output uses the states BEFORE the sample: the lane below, or the tile's start
            const f2 b_p = dpp<0x138, 0xF>(b_in, b_t), c_p = dpp<0x138, 0xF>(c_in, c_t);
            const f2 w1 = vfma(A2, v0 - c_p, a1 * b_p);                    // :566
            float y_lane = live ? w1.x + w1.y : 0.0f;
            if (A.resume) {   // development aid ("scan_debug" option): a chain quantity instead of the audio
                const float dv = A.resume == 1 ? alpha : A.resume == 2 ? jp : A.resume == 3 ? saw : A.resume == 4 ? nz :
                                 A.resume == 5 ? a1.x : A.resume == 6 ? tg.x : A.resume == 7 ? a_t.x :
                                 A.resume == 8 ? v0.x : A.resume == 9 ? ef.x : A.resume == 10 ? G.x :
                                 A.resume == 12 ? ef.y : A.resume == 13 ? G.y : A.resume == 14 ? a1.y :
                                 A.resume == 15 ? tg.y : A.resume == 16 ? a_t.y : A.resume == 17 ? v0.y : w1.x;
                y_lane = (pw == 0 && live) ? 2.0f * dv : 0.0f;
                if (A.resume >= 100) {      // 100 + f: the band-pass output of formant f alone
                    const int f = (int)A.resume - 100;
                    y_lane = (live && f / 2 == pw) ? 2.0f * (f % 2 ? w1.y : w1.x) : 0.0f;
                }
            }
            yout[(step - 1) & 1][pw][lane] = y_lane;
            // the tile's end (lanes >= n carried identities) is the next tile's start
            // (components go through named floats: bit-casting `v.y` of a by-value vector argument in place
            // made hipcc 7.2 read lane 63 of v.x for both halves)
            auto last_of = [](const float vx, const float vy) __attribute__((always_inline)) {
                f2 r;
                r.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx), 63));
                r.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy), 63));
                return r;
            };
            a_in = last_of(a_t.x, a_t.y);
            b_in = last_of(b_t.x, b_t.y);
            c_in = last_of(c_t.x, c_t.y);
        }
        __syncthreads();
        if (step >= 1 && step - 1 > last_tile) break;
    }
}

}  // namespace

hipError_t launch_scan(const SynthArgs &args, hipStream_t stream)
{
    if (args.n_utt == 0) return hipSuccess;
    if (args.live4) hipLaunchKernelGGL((scan_kernel<2>), dim3(args.n_utt), dim3(192), 0, stream, args);
    else hipLaunchKernelGGL((scan_kernel<4>), dim3(args.n_utt), dim3(320), 0, stream, args);
    return hipGetLastError();
}

}  // namespace grail
