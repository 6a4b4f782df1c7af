// scan_kernels.hip — time-parallel synthesis for small and mid-size batches in fast (tolerance) arithmetic, gfx950.
//
// The lane-per-utterance kernels of synth_kernels.hip need tens of thousands of utterances to fill an
// MI355X; with a few thousand most SIMDs idle and the time per batch is the serial length of one
// utterance.  This kernel turns the mapping around: ONE WORKGROUP PER UTTERANCE, LANES = TIME.
// The unit of work is a SUPER-TILE of up to 512 consecutive samples without an event inside.
//
//   chain wave            the per-utterance state of the reference, EXACT: Sequencer clock and segment
//                         advances (src/lib.rs:859-932), jitter phase, wraps and redraws (:240-306,
//                         :753-777), the pitch track.  It walks the super-tile in tiles of 64 samples, one per
//                         lane.  The clock `clk -= dt` and the jitter phase `p += inc` are serial f32
//                         accumulations; inside one binade they move by a constant quantum (the increment
//                         rounded to that binade's grid), so lane j gets its value as one fma, exactly;
//                         quantum and binade are kept from tile to tile and derived afresh only after an
//                         event or where the binade ends.
//   carrier               the carrier phase with its wrap (:520-525) is the one truly serial quantity:
//                         fract(p + f_j) handed down the lanes with DPP wave_shr:1; then saw with polyBLEP
//                         (:503-517) and the carrier-noise LCG (:36-55, closed-form skip-ahead).  In the chain
//                         wave (two-stage workgroups, many utterances) or on a wave of its own, one super-tile
//                         behind (three-stage workgroups, few utterances: the phase loop bounds the time).
//   filter wave           one super-tile further behind, formant pair by formant pair (two formants = one
//                         packed f32 vector; two pairs when formants 5-8 are provably dead, else four).
//                         Lane j owns the EIGHT consecutive samples 8j .. 8j+7: it evaluates their
//                         coefficients (at its first and last sample, interpolated in between under the
//                         fast-tile error guard; directly where the guard fails), composes its eight steps of
//                         each recurrence into one affine map, the 64 maps are combined by an inclusive scan
//                         over the lanes (six DPP steps: row_shr 1/2/4/8, row_bcast 15/31 — north_star's
//                         "first-order-section parallel scan"), and from the state the scan hands it the
//                         lane runs its eight samples with the plain recurrence.  The one-pole low-pass
//                         (:538) is the map a -> (1-k) a + k x, the Cytomic SVF (:565-571) the 2x2 affine map
//                           [b'; c'] = [[2 a1 - 1, -2 a2], [2 a2, 1 - 2 a3]] [b; c] + v0 [2 a2; 2 a3],
//                         composed as (M2, u2) o (M1, u1) = (M2 M1, M2 u1 + u2).  Work per sample: 8/8 of a
//                         serial filter step + 1/8 of a scan, instead of a whole scan per 64 samples.
//                         The pairs' band-pass outputs add up in formant order (:574) in registers and the
//                         lane stores its eight samples (a wave writes 2 KB runs).
//
// Pipeline stages one super-tile apart, double- or triple-buffered LDS in between, ONE workgroup barrier per
// super-tile.  Two waves per utterance keep eight workgroups resident per CU, so that mid-size batches
// (thousands of utterances) are bound by instruction issue, not by the latency of the serial chain.
// Tolerance mode only (the scans reassociate the recurrences); the discontinuous state is the reference's
// to the bit, so lengths and every boundary / wrap / saw edge sit where the reference puts them.  The host
// only sends batches here whose every parameter is inside the proven-safe window (voice_analysis.cpp
// scan_voice_ok): no NaN / Inf special cases exist on this path.
#include <cstdio>

#include <type_traits>

#include "device_common.h"
#include "kernels.h"
#include "pcm16.h"

namespace grail {
namespace {

constexpr int TL = 64;        // lanes per wave = samples per chain tile
constexpr int CK = 8;         // consecutive samples owned by one lane of the filter wave
constexpr int ST = TL * CK;   // samples per super-tile

struct __attribute__((aligned(16))) TileIn {
    float alpha[ST], jp[ST], saw[ST], nz[ST];
};
struct TileMeta {
    int n;          // samples of the super-tile
    int epoch;      // which parameter block applies
    uint32_t at;    // row position of its first sample
    int pad;
};

// what the filter wave needs of the current segment pair and jitter period (written by the chain wave)
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
// wave-uniform by construction; says so to the compiler (scalar registers, scalar branches)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }

// One level of the inclusive scans over the lanes: combine with the element CTRL lanes earlier in time
// (lanes without a source take the identity through DPP's `old` operand).
//   low-pass   (P, Q):  a -> P a + Q
//   band-pass  (M, U):  s -> M s + U
struct LpMap {
    f2 P, Q;
};
struct BpMap {
    f2 m11, m12, m21, m22, u1, u2;
};
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_level(LpMap &e)
{
    const f2 one = vsplat(1.0f, f2()), zero = vsplat(0.0f, f2());
    const f2 eP = dpp<CTRL, ROW_MASK>(one, e.P), eQ = dpp<CTRL, ROW_MASK>(zero, e.Q);
    e.Q = vfma(e.P, eQ, e.Q);
    e.P = e.P * eP;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_level(BpMap &e)
{
    const f2 one = vsplat(1.0f, f2()), zero = vsplat(0.0f, f2());
    const f2 e11 = dpp<CTRL, ROW_MASK>(one, e.m11), e12 = dpp<CTRL, ROW_MASK>(zero, e.m12);
    const f2 e21 = dpp<CTRL, ROW_MASK>(zero, e.m21), e22 = dpp<CTRL, ROW_MASK>(one, e.m22);
    const f2 eu1 = dpp<CTRL, ROW_MASK>(zero, e.u1), eu2 = dpp<CTRL, ROW_MASK>(zero, e.u2);
    const f2 n11 = vfma(e.m12, e21, e.m11 * e11), n12 = vfma(e.m12, e22, e.m11 * e12);
    const f2 n21 = vfma(e.m22, e21, e.m21 * e11), n22 = vfma(e.m22, e22, e.m21 * e12);
    e.u1 = vfma(e.m12, eu2, vfma(e.m11, eu1, e.u1));
    e.u2 = vfma(e.m22, eu2, vfma(e.m21, eu1, e.u2));
    e.m11 = n11; e.m12 = n12; e.m21 = n21; e.m22 = n22;
}
template <typename MAP>
__device__ __forceinline__ void scan_lanes(MAP &e)
{
    scan_level<0x111, 0xF>(e);   // row_shr:1
    scan_level<0x112, 0xF>(e);   // row_shr:2
    scan_level<0x114, 0xF>(e);   // row_shr:4
    scan_level<0x118, 0xF>(e);   // row_shr:8
    scan_level<0x142, 0xA>(e);   // row_bcast:15 into rows 1 and 3
    scan_level<0x143, 0xC>(e);   // row_bcast:31 into rows 2 and 3
}

typedef float vf4u __attribute__((ext_vector_type(4), aligned(4)));       // 16-byte stores at 4-byte alignment
typedef short vs8u __attribute__((ext_vector_type(8), aligned(2)));

// NP: formant pairs, 2 (formants 5-8 proven dead, see live4_ok) or 4.
// SPLIT: the carrier phase, saw and carrier noise run on a wave of their own, between the chain wave and the
// filter wave (three pipeline stages): the serial phase loop is half of the chain wave's time, and with few
// utterances the time per batch IS the chain wave's time.
// amdgpu_waves_per_eu(4): 128 VGPRs (a handful spilled), so that eight two-wave workgroups fit a CU.
template <int NP, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 192 : 128) __attribute__((amdgpu_waves_per_eu(4))) void scan_kernel(const SynthArgs A)
{
    constexpr int NBUF = SPLIT ? 3 : 2;        // super-tiles in flight
    constexpr int W_FILT = SPLIT ? 2 : 1;      // the filter wave; it works W_FILT super-tiles behind the chain
    __shared__ TileIn tin[NBUF];
    // one block per super-tile in flight: an epoch opens at most once per super-tile, the filter wave is W_FILT
    // super-tiles behind the chain wave, so epochs e .. e + W_FILT can be live at once (with two blocks the chain
    // wave of the three-stage flavour overwrote the block the filter wave was still reading whenever three
    // consecutive super-tiles each opened an epoch: segments of a few milliseconds)
    __shared__ ParamBlock par[NBUF];
    __shared__ TileMeta meta[4];
    __shared__ int last_tile;                 // index of the utterance's last super-tile, known once the chain ends

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform, and the compiler knows
    const uint32_t u = A.perm ? A.perm[blockIdx.x] : blockIdx.x;   // longest utterances first (ragged batches)
    if (threadIdx.x == 0) last_tile = 0x7fffffff;
    __syncthreads();

    uint32_t vid = A.voice_ids ? A.voice_ids[u] : 0u;
    if (vid >= A.n_voices) vid = 0u;
    const DevVoice VO = A.voices[vid];
    const float *__restrict__ elems = A.elems;

    // ---- the carrier: phase, saw, noise of up to 64 consecutive samples (lane = sample), from their pitch.
    // Runs in the chain wave, or in the phase wave of the SPLIT flavour.
    float phase = 0.0f;
    uint32_t noise_seed = 0u;                                       // :594
    const uint32_t skip_mul = LCG_SKIP.mul[lane + 1], skip_add = LCG_SKIP.add[lane + 1];
    auto carrier = [&](const float frequency, const int n, float &saw, float &nz) __attribute__((always_inline)) {
        // the carrier phase: p_j = fract(p_{j-1} + f_{j-1}), exact (:520-525), handed down the lanes: after
        // k rounds lanes <= k hold their phase.  Lane 0 has no lane below: the shifted-in value is 0 there
        // (bound_ctrl) and its addend is the phase the tile starts from, fract(0 + phase) = phase.  Rounds
        // beyond n - 1 only touch lanes >= n, so the trip count is rounded up to whole groups of eight.
        const float f_below = dpp<0x138, 0xF>(0.0f, frequency);     // wave_shr:1
        const float addend = lane == 0 ? phase : f_below;
        float ph = phase;
        for (int k = 1; k < n; k += 8) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                ph = __builtin_amdgcn_fractf(
                    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ph), 0x138, 0xF, 0xF, true)) + addend);
        }
        // saw with polyBLEP (:503-517), quotient by v_rcp (tolerance)
        const bool head = ph < frequency, tail = ph > (1.0f - frequency);
        const float tt = (head ? ph : ph - 1.0f) * __builtin_amdgcn_rcpf(frequency);
        const float pb_ = head ? ((2.0f * tt - tt * tt) - 1.0f) : ((tt * tt + 2.0f * tt) + 1.0f);
        saw = __builtin_fmaf(2.0f, ph, -1.0f) - ((head | tail) ? pb_ : 0.0f);
        // carrier noise :528: lane j is j + 1 draws after the tile's start state
        const uint32_t sk = noise_seed * skip_mul + skip_add;
        nz = (__uint_as_float((sk >> 9) | 0x3F800000u) - 1.5f) * 2.0f;
        // carry both to the tile's end
        const int last = n - 1;
        const float ph_l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ph), last));
        const float f_l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, frequency), last));
        phase = __builtin_amdgcn_fractf(ph_l + f_l);
        noise_seed = (uint32_t)__builtin_amdgcn_readlane((int)sk, last);
    };

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
        float inv_bl = 1.0f, bl = 1.0f;                             // blend length: +-2^k (times the exact reciprocal) or any
        bool bl_pow2 = true;
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

        const float lane_p1 = (float)(lane + 1);

        // closed forms kept from tile to tile: signed quantum, binade (biased exponent), still valid?
        float c_q = 0.0f, j_q = 0.0f;
        uint32_t c_e2 = 0u, j_e2 = 0u;
        bool c_reg = false, j_reg = false;

        const uint64_t cap = A.cap;
        const uint32_t cap32 = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
        uint32_t n_out = 0;
        bool finished = false, truncated = false;
        int epoch = 0;
        bool params_dirty = true;
        uint32_t quick_tiles = 0, derived_tiles = 0;   // statistics: tiles on the kept closed forms / derived afresh

        for (int step = 0;; ++step) {
            if (!finished) {
                TileIn &ti = tin[step % NBUF];
                const uint32_t at0 = n_out;
                int S = 0;                                                      // samples of this super-tile so far
                while (S <= ST - TL) {
                    // ---- the clock and the jitter phase of the tile by closed form.  From a known value v0
                    // the next two are plain serial steps v1, v2; from there on the sequence moves by the
                    // quantum q = v2 - v1 as long as the values stay in the binade of v2 (RN(v - d) =
                    // v - RN_grid(d) when v lies on the result's grid) and d is not exactly half-way between two
                    // grid points.  (q, binade) stay valid from tile to tile, so the usual tile is QUICK: one
                    // fma per sequence and a test that all 64 values are still in the binade — which also says
                    // that no clock went below zero (:864) and no phase above one (:245): no event in the tile.
                    float cj = 0.0f, pj = 0.0f;
                    int n = TL;
                    bool c_ok = false, j_ok = false;                            // this tile's 64 values by the kept form?
                    if (uni(cap32 - n_out >= (uint32_t)TL)) {
                        if (uni(c_reg)) {
                            cj = __builtin_fmaf(lane_p1, c_q, clk);
                            // (inside the binade: its lowest value 2^e itself is not — see `extend`)
                            c_ok = __builtin_amdgcn_ballot_w64((__float_as_uint(cj) >> 23) == c_e2 &&
                                                               (__float_as_uint(cj) & 0x7FFFFFu) != 0u) == ~0ull;
                        }
                        if (uni(j_reg)) {
                            pj = __builtin_fmaf(lane_p1, j_q, jphase);
                            j_ok = __builtin_amdgcn_ballot_w64((__float_as_uint(pj) >> 23) == j_e2) == ~0ull;
                        }
                    }
                    const bool quick = c_ok && j_ok;
                    if (quick) {
                        ++quick_tiles;
                    } else {
                        // ---- the first sample of the tile: the reference's own control flow.  An event ends the
                        // super-tile that is under way (its parameter block applies to every sample of it)
                        float clk_first = clk - dt;                                 // :861
                        float jp_first = jphase + jinc;                             // :242 / :291
                        if (uni(S > 0 && (clk_first < 0.0f || jp_first > 1.0f || n_out >= cap))) break;
                        if (uni(clk_first < 0.0f)) {                                // :864
                            if (cur.some && nxt.some) {                             // :868
                                cur = nxt;
                                fetch_seg(nxt, A.segs, seg_pos, seg_end, A.phoneme_mode != 0u, VO.elem_base);
                                clk_first += cur.length;                            // :873
                            } else if (!cur.some && !nxt.some) {                    // :876
                                fetch_seg(cur, A.segs, seg_pos, seg_end, A.phoneme_mode != 0u, VO.elem_base);
                                fetch_seg(nxt, A.segs, seg_pos, seg_end, A.phoneme_mode != 0u, VO.elem_base);
                                if (cur.some) clk_first += cur.length;              // :881-883
                            } else {
                                finished = true;                                    // :886
                            }
                            if (!finished && cur.some) {                            // the match at :891-931
                                const bool has_b = cur.elem >= 0, has_c = nxt.some && nxt.elem >= 0;
                                silent_pair = !has_b && !has_c;
                                bl = cur.blend_length;
                                const uint32_t blb = __float_as_uint(bl), ble = (blb >> 23) & 0xFFu;
                                bl_pow2 = uni(((blb & 0x7FFFFFu) == 0u) && ble >= 1u && ble <= 253u);
                                inv_bl = 1.0f / bl;                                 // exact when the length is +-2^k
                                if (has_b && has_c) { x_row = nxt.elem; y_row = cur.elem; xf = nxt.frequency; yf = cur.frequency; x_mute = y_mute = false; }
                                else if (has_b) { x_row = y_row = cur.elem; xf = yf = cur.frequency; x_mute = true; y_mute = false; }
                                else if (has_c) { x_row = y_row = nxt.elem; xf = yf = nxt.frequency; x_mute = false; y_mute = true; }
                                else { x_row = y_row = -1; xf = yf = 0.25f; x_mute = y_mute = false; }
                                params_dirty = true;
                            }
                            c_reg = false;
                        }
                        if (!cur.some) finished = true;                             // :930
                        if (!finished && n_out >= cap) { truncated = true; finished = true; }
                        if (uni(finished)) break;
                        if (uni(jp_first > 1.0f)) {                                 // :245 / :294
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
                            j_reg = false;
                        }
                        if (params_dirty) {
                            ++epoch;
                            ParamBlock &pb = par[epoch % NBUF];
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
                        // Where the binade ends, or after an event, the closed form is derived afresh from the last
                        // exact lane (up to six times per tile: the first tile after a jitter wrap crosses six
                        // binades).  `step` = -dt for the clock, +jinc for the jitter
                        // phase; `need_same_start`: an increasing sequence is regular only if v1 already lies in
                        // v2's binade (else v1 is off v2's grid).
                        auto extend = [&](float &v, int &nv, const float step_, const bool need_same_start,
                                          float &q_out, uint32_t &e2_out, bool &reg_out) __attribute__((always_inline)) {
                            // lanes < nv hold exact values; make lanes >= nv exact as far as one binade reaches
                            const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), nv - 1));
                            const float v1 = v0 + step_, v2 = v1 + step_, q = v2 - v1;
                            const int rel = lane - nv;                       // 0: v1, 1: v2, k >= 2: v2 + (k - 1) q
                            const float cand = rel <= 0 ? v1 : rel == 1 ? v2 : __builtin_fmaf((float)(rel - 1), q, v2);
                            const uint32_t e2 = __float_as_uint(v2) >> 23;
                            const float ulp = __uint_as_float(e2 > 23u ? (e2 - 23u) << 23 : 0u);
                            const bool regular = (__builtin_fabsf(step_ - q) != 0.5f * ulp) && e2 > 24u &&
                                                 (!need_same_start || (__float_as_uint(v1) >> 23) == e2);
                            // A falling sequence that lands exactly on 2^e has left the binade: the exact difference lies
                            // just below 2^e, where the grid is twice as fine, and rounds to a value there (the quantum was
                            // rounded to the coarse grid above) — 0x3d000000 instead of 0x3cffffff, one segment in a few
                            // hundred; the wrong clock moved alpha and with it the pitch of a blend by an ulp, and the
                            // carrier phase drifted by 1e-6 (found by tools/fuzz_soak.sh, round 3).
                            const bool inside = (__float_as_uint(cand) >> 23) == e2 &&
                                                (step_ > 0.0f || (__float_as_uint(cand) & 0x7FFFFFu) != 0u);
                            const bool good = lane < nv || rel <= 1 || (regular && inside);
                            v = lane >= nv ? cand : v;
                            const uint64_t bad_ = ~__builtin_amdgcn_ballot_w64(good);
                            const int nv_before = nv;
                            nv = bad_ ? __builtin_ctzll(bad_) : TL;
                            // the closed form carries over to the next tile if lane 63 was reached inside v2's binade
                            q_out = q;
                            e2_out = e2;
                            reg_out = regular && nv == TL && nv_before <= TL - 2;
                        };
                        // (a sequence whose kept form held for all 64 lanes — it then had no event either — stays as
                        // it is: usually only one of the two leaves its binade)
                        ++derived_tiles;
                        int nc = TL, np_ = TL;
                        if (!c_ok) { cj = clk_first; nc = 1; c_reg = false; }
                        if (!j_ok) { pj = jp_first; np_ = 1; j_reg = false; }
#pragma unroll 1
                        for (int pass = 0; pass < 6 && (nc < TL || np_ < TL); ++pass) {
                            if (nc < TL) extend(cj, nc, -dt, false, c_q, c_e2, c_reg);
                            if (np_ < TL) extend(pj, np_, jinc, true, j_q, j_e2, j_reg);
                        }
                        // the tile ends before the first event: a clock below zero (:864) or a phase above one (:245)
                        const bool ok = lane == 0 || (lane < nc && lane < np_ && cj >= 0.0f && pj <= 1.0f);
                        const uint64_t bad = ~__builtin_amdgcn_ballot_w64(ok);
                        n = bad ? __builtin_ctzll(bad) : TL;
                        const uint32_t room = cap32 - n_out;
                        n = uni((uint32_t)n > room ? (int)room : n);
                        if (n <= 0) break;
                        // the closed forms carry over only if the value the next tile starts from lies in their
                        // binades (a tile cut short may end in the binade before) and the phase binade is below one
                        const float c_last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cj), n - 1));
                        const float j_last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pj), n - 1));
                        c_reg = c_reg && (__float_as_uint(c_last) >> 23) == c_e2 && (__float_as_uint(c_last) & 0x7FFFFFu) != 0u;
                        j_reg = j_reg && (__float_as_uint(j_last) >> 23) == j_e2 && j_e2 < 127u;
                    }
                    // ---- per lane: alpha, the pitch (exact: :404-414, :254, :763)
                    // clk / 2^k == clk * 2^-k for every clk; any other blend length takes the IEEE quotient
                    float ratio = cj * inv_bl;
                    if (!bl_pow2) ratio = cj / bl;
                    float alpha = __builtin_fminf(ratio, 1.0f);                 // :899/:908/:917
                    alpha = silent_pair ? 1.0f : alpha;
                    const float oma = 1.0f - alpha, jomp = 1.0f - pj;
                    float frequency = xf * oma + yf * alpha;
                    const float n_freq = fn_cur * jomp + fn_next * pj;
                    frequency = frequency + n_freq * d_freq;
                    // lanes >= n write beyond the tile: the next tile overwrites them, or nobody reads them
#ifdef GRAIL_SCAN_DEBUG
                    ti.alpha[S + lane] = A.resume == 9 ? cj : alpha;        // development probe 9: the clock itself
#else
                    ti.alpha[S + lane] = alpha;
#endif
                    ti.jp[S + lane] = pj;
                    if constexpr (SPLIT) {
                        ti.saw[S + lane] = frequency;                           // the phase wave turns it into the saw
                    } else {
                        float saw, nz;
                        carrier(frequency, n, saw, nz);
                        ti.saw[S + lane] = saw;
                        ti.nz[S + lane] = nz;
                    }
                    // ---- carry the chain to the tile's end
                    const int last = n - 1;
                    clk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cj), last));
                    jphase = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pj), last));
                    n_out += (uint32_t)n;
                    S += n;
                }
                if (S > 0) {
                    if (lane == 0) { TileMeta &m = meta[step & 3]; m.n = S; m.epoch = epoch; m.at = at0; }
                } else {
                    finished = true;                    // nothing left (or no room): the previous super-tile was the last
                    if (lane == 0) last_tile = step - 1;
                }
            }
            __syncthreads();
            if (step - W_FILT >= last_tile) break;      // the filter wave stored the last super-tile in this step
        }
        if (lane == 0) {
            if (A.out_len) A.out_len[u] = n_out;
            if (truncated) atomicOr(A.truncated, 1u);
            atomicAdd(A.truncated + 2, quick_tiles);
            atomicAdd(A.truncated + 3, derived_tiles);
        }
        return;
    }

    if constexpr (SPLIT) {
        if (wave == 1) {
            // =============================== the phase wave ===============================
            // one super-tile behind the chain wave: 64 samples at a time, whatever tiles the chain cut
            for (int step = 0;; ++step) {
                if (step >= 1 && step - 1 <= last_tile) {
                    const int S = uni(meta[(step - 1) & 3].n);
                    TileIn &ti = tin[(step - 1) % NBUF];
                    float f_next = ti.saw[lane];                               // the pitch, one tile ahead of its use
                    for (int r = 0; r < S; r += TL) {
                        const int n = S - r < TL ? S - r : TL;
                        const float frequency = f_next;                        // lanes >= n: stale values, unused
                        if (r + TL < S) f_next = ti.saw[r + TL + lane];
                        float saw = 0.0f, nz = 0.0f;
#ifdef GRAIL_SCAN_DEBUG
                        if (A.resume != 205) carrier(frequency, n, saw, nz);   // 205: development probe, the chain wave alone
#else
                        carrier(frequency, n, saw, nz);
#endif
                        ti.saw[r + lane] = saw;
                        ti.nz[r + lane] = nz;
                    }
                }
                __syncthreads();
                if (step - W_FILT >= last_tile) break;
            }
            return;
        }
    }

    // =============================== the filter wave ===============================
    // filter states at the super-tile's start, formant f in lane f of three registers
    float st_a = 0.0f, st_b = 0.0f, st_c = 0.0f;
    // the parameter block of the epoch, one value per lane: X[lane], Y[lane], the four noise arrays
    float vX = 0.0f, vY = 0.0f, vN = 0.0f;
    float d_ffreq = 0.0f, amp_scale = 0.0f;
    int have_epoch = -1;
    const f2 one = vsplat(1.0f, f2()), zero = vsplat(0.0f, f2());
    const f2 five = vsplat(5.0f, f2()), m4 = vsplat(-4.0f, f2()), two_ = vsplat(2.0f, f2());
    auto at_lane = [](const float v, const int l) __attribute__((always_inline)) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    };
    auto put_lane = [&](float &v, const int l, const float x) __attribute__((always_inline)) { v = lane == l ? x : v; };

    // one super-tile: FULL = all 512 samples present (no lane needs identity padding)
    auto super_tile = [&](const TileIn &ti, const TileMeta m, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int S = m.n;
        const int mine = S - CK * lane;                                    // my samples: k < mine
        float y[CK];
#pragma unroll
        for (int k = 0; k < CK; ++k) y[k] = 0.0f;
#pragma unroll 1
        for (int p = 0; p < NP; ++p) {
            const int f0 = 2 * p;
            // every pair reads the tile's samples again (eight registers per sample kept across a pair's
            // whole computation would cost a wave of occupancy): the offset is opaque to the optimiser
            int off = CK * lane;
            asm volatile("" : "+v"(off));
            const float4 *pa = reinterpret_cast<const float4 *>(&ti.alpha[off]);
            const float4 *pj = reinterpret_cast<const float4 *>(&ti.jp[off]);
            const float4 *ps = reinterpret_cast<const float4 *>(&ti.saw[off]);
            const float4 *pn = reinterpret_cast<const float4 *>(&ti.nz[off]);
            f2 a_in, b_in, c_in;
            a_in.x = at_lane(st_a, f0); a_in.y = at_lane(st_a, f0 + 1);
            b_in.x = at_lane(st_b, f0); b_in.y = at_lane(st_b, f0 + 1);
            c_in.x = at_lane(st_c, f0); c_in.y = at_lane(st_c, f0 + 1);
            auto two = [&](const float v, const int off) __attribute__((always_inline)) { f2 r; r.x = at_lane(v, off + f0); r.y = at_lane(v, off + f0 + 1); return r; };
            const f2 Xf = two(vX, F_FREQ), Xb = two(vX, F_BW), Xs = two(vX, F_SMOOTH), Xr = two(vX, F_BREATH), Xt = two(vX, F_TURB), Xa = two(vX, F_AMP);
            const f2 Yf = two(vY, F_FREQ), Yb = two(vY, F_BW), Ys = two(vY, F_SMOOTH), Yr = two(vY, F_BREATH), Yt = two(vY, F_TURB), Ya = two(vY, F_AMP);
            const f2 ffc = two(vN, 0), ffn = two(vN, NF), fac = two(vN, 2 * NF), fan = two(vN, 3 * NF);
            // v0_k = PW_k a_start + QW_k: my low-pass steps up to sample k composed, times w_k (:544-550)
            f2 PW[CK], QW[CK], A1[CK], TG[CK];
            // ---- coefficients of my eight samples; the low-pass steps composed into one map
            LpMap lp;
            lp.P = one;
            lp.Q = zero;
            struct Coef { f2 keep, breath, a1, tg, G, H; };               // keep = exp_approx(smooth) = 1 - k of :535-538
            // everything that is a smooth function of (alpha, jitter phase), evaluated directly
            auto eval = [&](const float alpha, const float jpk, Coef &c, f2 &ea, f2 &mu, f2 &et) __attribute__((always_inline)) {
                const float oma = 1.0f - alpha, jomp = 1.0f - jpk;
                const f2 av = vsplat(alpha, f2()), jpv = vsplat(jpk, f2());
                // SynthesisElem::blend :404-414 and Jitter::next :763-773 with fused multiply-adds
                f2 ef = vfma(Yf, av, Xf * oma);
                const f2 eb = vfma(Yb, av, Xb * oma);
                const f2 es = vfma(Ys, av, Xs * oma);
                c.breath = vfma(Yr, av, Xr * oma);
                et = vfma(Yt, av, Xt * oma);
                ea = vfma(Ya, av, Xa * oma);
                const f2 nff = vfma(ffn, jpv, ffc * jomp);
                const f2 nfa = vfma(fan, jpv, fac * jomp);
                ef = vfma(nff, vsplat(d_ffreq, f2()), ef);
                mu = vfma(nfa + 1.0f, vsplat(-amp_scale, f2()), one);
                c.G = ea * mu;
                c.H = et * c.G;
                // tan_approx :63-70, k :558, a1 :560 — reciprocals by v_rcp + one Newton step
                const f2 omx = 1.0f - ef, xph = ef + 0.5f, hmx = 0.5f - ef;
                const f2 ox = omx * ef, ph_ = xph * hmx;
                const f2 num = ox * vfma(m4, ph_, five);
                const f2 den = (xph * vfma(m4, ox, five)) * hmx;
                f2 rd = vrcp(den), rx = vrcp(ef);
                rd = vfma(vfma(-den, rd, one), rd, rd);
                rx = vfma(vfma(-ef, rx, one), rx, rx);
                c.tg = num * rd;
                const f2 kq = eb * rx;
                const f2 d3 = vfma(c.tg, c.tg + kq, one);
                f2 a1 = vrcp(d3);
                c.a1 = vfma(vfma(-d3, a1, one), a1, a1);
                c.keep = exp_approx(es);                                       // :535
            };
            // one sample: the low-pass step composed onto my earlier ones, the band-pass coefficients kept
            auto sample = [&](const int k, Coef c, const float swk, const float nzk) __attribute__((always_inline)) {
                const f2 nw = vfma(c.breath, vsplat(nzk - swk, f2()), vsplat(swk, f2()));   // :531
                f2 pk = c.keep, qk = vfma(-c.keep, nw, nw);                    // a -> keep a + (1 - keep) nw :538
                f2 w = vfma(c.H, vsplat(nzk - 1.0f, f2()), c.G);               // v0 = a w :544-550
                if constexpr (!FULL) {
                    const bool live = k < mine;                                // identity steps beyond the end
                    pk = live ? pk : one;
                    qk = live ? qk : zero;
                    c.a1 = live ? c.a1 : one;
                    c.tg = live ? c.tg : zero;
                    w = live ? w : zero;
                }
                lp.Q = vfma(pk, lp.Q, qk);
                lp.P = lp.P * pk;
                PW[k] = lp.P * w; QW[k] = lp.Q * w; A1[k] = c.a1; TG[k] = c.tg;
            };
            auto part = [](const float4 v, const int i) __attribute__((always_inline)) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; };
            float4 a4, j4, s4, n4;
            // In a full super-tile alpha and the jitter phase are linear in time, so the coefficients are
            // evaluated at my first and last sample and interpolated in between — under fast_tile's error
            // guard (synth_kernels.hip): no kink of alpha = min(clk / blend, 1) among my samples, relative
            // change of a1, g and the low-pass factor r <= 2^-9.5 (error r^2 / 16 <= 2^-23), the products of
            // linear functions G and H within 2^-22.  Seven samples apart instead of 32: the guard only
            // fails for parameters moving ~20x faster than the reference's front end makes them, and then
            // (any lane of the wave) every sample is evaluated directly.
            bool interpolate = false;
            Coef C0, D;
            if constexpr (FULL) {
                Coef C7;
                f2 ea0, mu0, et0, ea7, mu7, et7;
                const float al0 = ti.alpha[off], al7 = ti.alpha[off + CK - 1];
                eval(al0, ti.jp[off], C0, ea0, mu0, et0);
                eval(al7, ti.jp[off + CK - 1], C7, ea7, mu7, et7);
                D.keep = C7.keep - C0.keep; D.breath = C7.breath - C0.breath; D.a1 = C7.a1 - C0.a1;
                D.tg = C7.tg - C0.tg; D.G = C7.G - C0.G; D.H = C7.H - C0.H;
                float ra = 0.0f, rg = 0.0f;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    ra = __builtin_fmaxf(ra, __builtin_fabsf(vget(D.a1, c)) * __builtin_amdgcn_rcpf(vget(C0.a1, c)));
                    ra = __builtin_fmaxf(ra, __builtin_fabsf(vget(D.tg, c)) * __builtin_amdgcn_rcpf(vget(C0.tg, c)));
                    ra = __builtin_fmaxf(ra, __builtin_fabsf(vget(D.keep, c)) * __builtin_amdgcn_rcpf(1.0f - vget(C0.keep, c)));
                    rg = __builtin_fmaxf(rg, __builtin_fabsf((vget(ea7, c) - vget(ea0, c)) * (vget(mu7, c) - vget(mu0, c))));
                    rg = __builtin_fmaxf(rg, __builtin_fabsf((vget(et7, c) - vget(et0, c)) * vget(D.G, c)));
                }
                const bool kink = (al0 == 1.0f) != (al7 == 1.0f);
                const bool good = !kink && ra <= 0.0013810679f && rg <= 9.5367431640625e-07f;   // 2^-9.5, 2^-20
                interpolate = __builtin_amdgcn_ballot_w64(good) == ~0ull;
            }
            if (interpolate) {
                const f2 seventh = vsplat(1.0f / 7.0f, f2());
                D.keep = D.keep * seventh; D.breath = D.breath * seventh; D.a1 = D.a1 * seventh;
                D.tg = D.tg * seventh; D.G = D.G * seventh; D.H = D.H * seventh;
#pragma unroll
                for (int k = 0; k < CK; ++k) {
                    if (k % 4 == 0) { s4 = ps[k / 4]; n4 = pn[k / 4]; }
                    Coef c = C0;
                    if (k > 0) {
                        const f2 kf = vsplat((float)k, f2());
                        c.keep = vfma(kf, D.keep, C0.keep); c.breath = vfma(kf, D.breath, C0.breath);
                        c.a1 = vfma(kf, D.a1, C0.a1); c.tg = vfma(kf, D.tg, C0.tg);
                        c.G = vfma(kf, D.G, C0.G); c.H = vfma(kf, D.H, C0.H);
                    }
                    sample(k, c, part(s4, k % 4), part(n4, k % 4));
                }
            } else {
#pragma unroll
                for (int k = 0; k < CK; ++k) {
                    if (k % 4 == 0) { a4 = pa[k / 4]; j4 = pj[k / 4]; s4 = ps[k / 4]; n4 = pn[k / 4]; }
                    Coef c;
                    f2 ea, mu, et;
                    eval(part(a4, k % 4), part(j4, k % 4), c, ea, mu, et);
                    sample(k, c, part(s4, k % 4), part(n4, k % 4));
                }
            }
            // ---- the low-pass :538: scan the lanes' maps; my samples start from the state of the lane below
            const f2 myP = lp.P, myQ = lp.Q;
            scan_lanes(lp);
            const f2 Pb = dpp<0x138, 0xF>(one, lp.P), Qb = dpp<0x138, 0xF>(zero, lp.Q);   // the lanes before me
            const f2 a_start = vfma(Pb, a_in, Qb);
            const f2 a_end = vfma(myP, a_start, myQ);
            BpMap bp;
            bp.m11 = one; bp.m12 = zero; bp.m21 = zero; bp.m22 = one; bp.u1 = zero; bp.u2 = zero;
#pragma unroll
            for (int k = 0; k < CK; ++k) {
                const f2 v0 = vfma(PW[k], a_start, QW[k]);
                PW[k] = v0;
                // the band-pass step :560-571 as a 2x2 affine map, composed onto my earlier steps
                const f2 A2 = A1[k] * TG[k];                                   // a2 = g a1
                const f2 A3 = A2 * TG[k];                                      // a3 = g a2
                const f2 m11 = vfma(two_, A1[k], -one), m21 = A2 + A2, m22 = vfma(-two_, A3, one);
                const f2 u1 = m21 * v0, u2 = (A3 + A3) * v0;                   // m12 = -m21
                const f2 n11 = vfma(-m21, bp.m21, m11 * bp.m11), n12 = vfma(-m21, bp.m22, m11 * bp.m12);
                const f2 n21 = vfma(m22, bp.m21, m21 * bp.m11), n22 = vfma(m22, bp.m22, m21 * bp.m12);
                const f2 nu1 = vfma(-m21, bp.u2, vfma(m11, bp.u1, u1));
                const f2 nu2 = vfma(m22, bp.u2, vfma(m21, bp.u1, u2));
                bp.m11 = n11; bp.m12 = n12; bp.m21 = n21; bp.m22 = n22; bp.u1 = nu1; bp.u2 = nu2;
            }
            scan_lanes(bp);
            const f2 e11 = dpp<0x138, 0xF>(one, bp.m11), e12 = dpp<0x138, 0xF>(zero, bp.m12);
            const f2 e21 = dpp<0x138, 0xF>(zero, bp.m21), e22 = dpp<0x138, 0xF>(one, bp.m22);
            const f2 eu1 = dpp<0x138, 0xF>(zero, bp.u1), eu2 = dpp<0x138, 0xF>(zero, bp.u2);
            f2 b = vfma(e12, c_in, vfma(e11, b_in, eu1));
            f2 c = vfma(e22, c_in, vfma(e21, b_in, eu2));
#pragma unroll
            for (int k = 0; k < CK; ++k) {
                const f2 v3 = PW[k] - c;                                       // :565
                const f2 v1 = A1[k] * vfma(TG[k], v3, b);                      // :566 with a2 = g a1
                const f2 v2 = vfma(TG[k], v1, c);                              // :567 with a2 = g a1, a3 = g a2
                b = vfma(two_, v1, -b);                                        // :570
                c = vfma(two_, v2, -c);                                        // :571
                const float yp = v1.x + v1.y;
                y[k] += yp;                                                    // :574, in formant order
            }
#ifdef GRAIL_SCAN_DEBUG
            if (A.resume && A.resume < 200 && p == 0) {   // development aid ("scan_debug" option): a chain quantity instead of the audio
#pragma unroll
                for (int k = 0; k < CK; ++k) {
                    const float dv = (A.resume == 1 || A.resume == 9) ? ti.alpha[CK * lane + k] : A.resume == 2 ? ti.jp[CK * lane + k] :
                                     A.resume == 3 ? ti.saw[CK * lane + k] : A.resume == 4 ? ti.nz[CK * lane + k] :
                                     A.resume == 5 ? A1[k].x : A.resume == 6 ? TG[k].x : PW[k].x;
                    y[k] = 2.0f * dv;
                }
            }
#endif
            // the super-tile's end (lanes beyond it carried identities) is the next one's start
            put_lane(st_a, f0, lane63(a_end.x)); put_lane(st_a, f0 + 1, lane63(a_end.y));
            put_lane(st_b, f0, lane63(b.x)); put_lane(st_b, f0 + 1, lane63(b.y));
            put_lane(st_c, f0, lane63(c.x)); put_lane(st_c, f0 + 1, lane63(c.y));
#ifdef GRAIL_SCAN_DEBUG
            if (A.resume && A.resume < 200) break;
#endif
        }
        // ---- my eight samples of the row
        const uint64_t at = (uint64_t)u * A.out_stride + m.at + (uint32_t)(CK * lane);
        if constexpr (FULL) {
            if (A.out_pcm16) {
                vs8u v;
#pragma unroll
                for (int k = 0; k < CK; ++k) v[k] = (short)pcm16_from_f32(y[k] * 0.5f);
                *reinterpret_cast<vs8u *>(A.out_pcm16 + at) = v;
            } else {
#pragma unroll
                for (int h = 0; h < CK / 4; ++h) {
                    vf4u v;
                    v.x = y[4 * h] * 0.5f; v.y = y[4 * h + 1] * 0.5f; v.z = y[4 * h + 2] * 0.5f; v.w = y[4 * h + 3] * 0.5f;
                    *reinterpret_cast<vf4u *>(A.out + at + 4 * h) = v;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < CK; ++k) {
                if (k < mine) {
                    const float sample = y[k] * 0.5f;                          // :574
                    if (A.out_pcm16) A.out_pcm16[at + k] = (int16_t)pcm16_from_f32(sample);
                    else A.out[at + k] = sample;
                }
            }
        }
    };

    for (int step = 0;; ++step) {
#ifdef GRAIL_SCAN_DEBUG
        const bool probe_skip = A.resume >= 200 && (A.resume & 1);    // development probe: the filter wave idle
#else
        constexpr bool probe_skip = false;
#endif
        if (step >= W_FILT && step - W_FILT <= last_tile && !probe_skip) {
            const TileMeta m = meta[(step - W_FILT) & 3];
            if (m.epoch != have_epoch) {
                have_epoch = m.epoch;
                const ParamBlock &pb = par[have_epoch % NBUF];
                vX = pb.X[lane < ELEM_FLOATS ? lane : 0];
                vY = pb.Y[lane < ELEM_FLOATS ? lane : 0];
                vN = pb.ffc[lane & 31];                  // ffc, ffn, fac, fan: 4 x NF consecutive floats
                d_ffreq = uni(pb.d_ffreq);
                amp_scale = uni(pb.amp_scale);
            }
            const TileIn &ti = tin[(step - W_FILT) % NBUF];
            if (m.n == ST) super_tile(ti, m, std::true_type());
            else super_tile(ti, m, std::false_type());
        }
        __syncthreads();
        if (step - W_FILT >= last_tile) break;
    }
}

}  // namespace

hipError_t launch_scan(const SynthArgs &args, hipStream_t stream)
{
    if (args.n_utt == 0) return hipSuccess;
    // few utterances: the time per batch is the chain's; three-stage workgroups halve it.  Many: two waves
    // per utterance keep more utterances resident per CU (args.pipe: the host's choice, launch_plan.cpp)
    if (args.pipe) {
        if (args.live4) hipLaunchKernelGGL((scan_kernel<2, true>), dim3(args.n_utt), dim3(192), 0, stream, args);
        else hipLaunchKernelGGL((scan_kernel<4, true>), dim3(args.n_utt), dim3(192), 0, stream, args);
    } else {
        if (args.live4) hipLaunchKernelGGL((scan_kernel<2, false>), dim3(args.n_utt), dim3(128), 0, stream, args);
        else hipLaunchKernelGGL((scan_kernel<4, false>), dim3(args.n_utt), dim3(128), 0, stream, args);
    }
    return hipGetLastError();
}

}  // namespace grail
