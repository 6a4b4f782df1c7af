// synth_kernel_general_step.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// general_step: the literal control flow of the reference for one sample (any lane may have an event).
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
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
