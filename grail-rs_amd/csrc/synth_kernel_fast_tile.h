// synth_kernel_fast_tile.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// FAST: one tile of T steps, every lane deciding for itself (fast_render_tile).
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
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
