// synth_kernel_flush.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// flush_rows: the staged tile's rows to memory (f32 or i16 PCM, vector or scalar stores).
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
    // ---- the staged tile's rows to memory: row `slot` holds samples [base_, base_ + T), mine_ of them valid
    // (the general flush; the main loop below has a shortcut for the usual full tile of the lane kernels)
    auto flush_rows = [&](const uint32_t base_, const uint32_t mine_) __attribute__((always_inline)) {
        constexpr int ROW_LANES = T / 4;
        constexpr int ROWS_PER_IT = 64 / ROW_LANES;
        const int rl = lane % ROW_LANES;
        const int rr = lane / ROW_LANES;
        if (emit && j == L - 1) cnt[slot] = mine_;
        if constexpr (PIPE) __syncthreads();
        else wave_lds_sync();
        const int r_first = PIPE ? wave * ROWS_PER_IT : 0;
        constexpr int R_STEP = PIPE ? ROWS_PER_IT * WAVES : ROWS_PER_IT;
#pragma unroll 1
        for (int r0 = r_first; r0 < S; r0 += R_STEP) {
            const int r = r0 + rr;
            if (ROWS_PER_IT > S && r >= S) continue;
            const uint32_t c = cnt[r];
            const int t0 = rl * 4;
            if ((uint32_t)t0 < c) {
                const uint64_t at = (uint64_t)(A.perm ? rowid[r] : u0 + r) * A.out_stride + base_ + t0;
                auto sample_at = [&](const int tt) __attribute__((always_inline)) -> float {
                    if constexpr (FOLD_IN_FLUSH) {
                        // v1.sum() * 0.5: the left fold from 0.0 over formants 0..7  :574, :123-125
                        const float *p = stage + (tt * S + r) * NFA;
                        float run = 0.0f;
#pragma unroll
                        for (int f = 0; f < NFA; ++f) run = run + p[f];
                        if (NFA < NF) run = run + 0.0f;   // formants 5-8: literal +0.0 terms
                        return run * 0.5f;
                    } else {
                        return stage[tt * SP + r];
                    }
                };
                const float s0 = sample_at(t0 + 0);
                const float s1 = sample_at(t0 + 1);
                const float s2 = sample_at(t0 + 2);
                const float s3 = sample_at(t0 + 3);
                if (A.out_pcm16) {
                    // the WAV sink's `(x * i16::MAX as f32) as i16` (examples/cli.rs:49) on the way out
                    int16_t *dst = A.out_pcm16 + at;
                    const int p0 = pcm16_from_f32(s0), p1 = pcm16_from_f32(s1);
                    const int p2 = pcm16_from_f32(s2), p3 = pcm16_from_f32(s3);
                    if (vec16_ok && (uint32_t)(t0 + 4) <= c) {
                        *reinterpret_cast<uint2 *>(dst) =
                            make_uint2((uint32_t)(p0 & 0xFFFF) | ((uint32_t)p1 << 16),
                                       (uint32_t)(p2 & 0xFFFF) | ((uint32_t)p3 << 16));
                    } else {
                        dst[0] = (int16_t)p0;
                        if ((uint32_t)(t0 + 1) < c) dst[1] = (int16_t)p1;
                        if ((uint32_t)(t0 + 2) < c) dst[2] = (int16_t)p2;
                        if ((uint32_t)(t0 + 3) < c) dst[3] = (int16_t)p3;
                    }
                    continue;
                }
                float *dst = A.out + at;
                if (vec_ok && (uint32_t)(t0 + 4) <= c) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(s0, s1, s2, s3);
                } else {
                    dst[0] = s0;
                    if ((uint32_t)(t0 + 1) < c) dst[1] = s1;
                    if ((uint32_t)(t0 + 2) < c) dst[2] = s2;
                    if ((uint32_t)(t0 + 3) < c) dst[3] = s3;
                }
            }
        }
        if constexpr (PIPE) __syncthreads();     // the rendering wave may not park the next tile before all have read
        else wave_lds_sync();
    };
