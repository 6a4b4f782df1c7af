// synth_kernel_pipe.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// PIPE: the calm steps cut in three stages, one per wave role, handed on through LDS.
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
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
