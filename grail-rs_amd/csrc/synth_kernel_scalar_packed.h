// synth_kernel_scalar_packed.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// two calm samples per trip for L < 8: the per-utterance chain on float2 values, formant vectors sample by sample.
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
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
