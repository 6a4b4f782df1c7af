// synth_kernel_tile_loop.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// the tile loop: calm tiles, tiles with events (runs between them), flush, the end of the launch.
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
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
