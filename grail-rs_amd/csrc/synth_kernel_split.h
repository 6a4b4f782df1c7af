// synth_kernel_split.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// SPLIT: fast-forward of the exact per-utterance chain to where a chunk's filters start.
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
    // ---- SPLIT: fast-forward the exact per-utterance chain to where this chunk's filters start
    uint32_t base0 = 0;
    uint32_t reset_at = 0;       // the tile at which this lane's filters start from zero state
    if constexpr (SPLIT) {
        // (a voice's phonemes decide its warm-up; a batch of caller-built elems brings its own)
        uint32_t w = slot_used ? (A.split_warmup != 0u ? A.split_warmup : VO.warmup) : 0u, w_max = w;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)w_max, m);
            w_max = o > w_max ? o : w_max;
        }
        w_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)w_max);
        base0 = chunk_lo > w_max ? chunk_lo - w_max : 0u;
        reset_at = chunk_lo > w ? chunk_lo - w : 0u;
        for (;;) {
            if (__builtin_amdgcn_ballot_w64(!done & (n_out < base0)) == 0) break;
            const bool calm = !done & quiet_ok & (dt > 0.0f) & (clk > (float)(T + 8) * dt) &
                              (jphase + (float)(T + 1) * jinc < 0.999f) & (n_out + (uint32_t)T <= base0);
            if (__builtin_amdgcn_ballot_w64(!(calm | done)) == 0) {
                // (the usual tile: no lane's alpha needs its clamp — the blend is still under way)
                const bool below_one = !silent_pair & ((clk - dt) * inv_blend_length <= 1.0f) & (ANYBL ? blend_pow2 : true);
                if (__builtin_amdgcn_ballot_w64(!(below_one | done)) == 0) {
#pragma unroll 4
                    for (int tc = 0; tc < T; tc += 2) {
                        f2 PH, frequency;
                        chain_pair(std::false_type(), PH, frequency);
                    }
                } else {
#pragma unroll 4
                    for (int tc = 0; tc < T; tc += 2) {
                        f2 PH, frequency;
                        chain_pair(std::true_type(), PH, frequency);
                    }
                }
                n_out += done ? 0u : (uint32_t)T;
            } else {
                // A tile in which some lane has an event, pair by pair (as the mixed tile of the rendering loop):
                // a lane without an event of its own in the pair takes the packed chain step, the others the
                // reference's control flow; the usual pairs — nobody has one — in a tight loop of their own.  The
                // lanes of the wave move in lockstep (two samples per pair), so n_out < base0 holds for all of them
                // until the tile ends.
                auto pair_calm = [&]() __attribute__((always_inline)) -> bool {
                    return !done & quiet_ok & (dt > 0.0f) & (clk > 2.5f * dt) & (jphase + 2.01f * jinc < 1.0f);
                };
                int t = 0;
#pragma unroll 1
                while (t < T) {
#pragma unroll 1
                    for (; t < T; t += 2) {
                        if (__builtin_amdgcn_ballot_w64(!(pair_calm() | done)) != 0) break;
                        f2 PH, frequency;
                        chain_pair(std::true_type(), PH, frequency);
                        n_out += done ? 0u : 2u;
                    }
                    if (t >= T) break;
                    if (pair_calm()) {
                        f2 PH, frequency;
                        chain_pair(std::true_type(), PH, frequency);
                        n_out += 2u;
                    } else {
                        general_step(t, std::true_type());
                        general_step(t + 1, std::true_type());
                    }
                    t += 2;
                }
            }
        }
        // the carrier noise state after n_out draws from seed 0 (:594): s -> 16807 s + 1 composed n_out times
        {
            uint32_t mul = 16807u, add = 1u, acc = 0u;
#pragma unroll 1
            for (int b = 0; b < 32; ++b) {
                if ((n_out >> b) & 1u) acc = acc * mul + add;
                add = add * (mul + 1u);
                mul = mul * mul;
            }
            noise_seed = acc;
        }
    }
