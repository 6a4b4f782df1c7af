// synth_kernel_fast.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// FAST: the tolerance-mode arithmetic — sub-tile end points, error guard, the packed chain, the formant update.
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
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
