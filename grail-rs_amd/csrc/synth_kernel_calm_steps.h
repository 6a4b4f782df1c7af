// synth_kernel_calm_steps.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// quiet_step and the time-packed calm steps of one formant per lane (formant_pair, quad_chain, time_packed_block).
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
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
