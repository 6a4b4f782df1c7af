// synth_kernel_parts.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// what the kernel's body is written with: DPP hand-offs, the packed parameter record (Part), the reference's per-formant
// filter step (formant_filters, src/lib.rs:531-571), the band-pass coefficients, the operand-window checks of the short
// division, the stream state mover.  Namespace level.
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
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
