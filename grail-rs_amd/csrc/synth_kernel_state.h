// synth_kernel_state.h — a FRAGMENT of synth_kernel.h (included there, in this order, nowhere else; not a header of its own):
// the body's prologue: which utterance a lane renders, the iterator state of the chain (Sequencer :839-854, Jitter
// :724-748, Synthesize :470-488) in registers, setup_pair, the resumable state block.
// The cut is textual: every instantiation unit preprocesses to the token stream it had as one file.
    static_assert(!FAST || (!PIPE && !HALF), "FAST");
    static_assert(!MID || FAST, "MID is a flavour of the tolerance kernels");
    static_assert(!SPLIT || (FAST && !STREAM && L == 1 && WAVES == 1 && T == 64), "SPLIT");
    static_assert(NFA == NF || (NFA == 4 && !HALF), "NFA");
    static_assert(!PIPE || (WAVES == 4 && NFA / L == 1 && L >= 4 && !HALF && T % 4 == 0), "PIPE");
    constexpr int FPL = NFA / L;         // formants per lane
    constexpr int W = FPL >= 2 ? 2 : 1;  // formants per packed value
    constexpr int NV = FPL / W;          // packed values per lane and field
    typedef typename VecOf<W>::type V;
    constexpr int S = 64 / L;            // utterances per wave
    constexpr int SP = S + 1;            // padded row of the staging tile
    static_assert(T % 4 == 0 && (64 % (T / 4)) == 0, "T");
    // ONE WAVE PER SIMD, by construction.  Every family is laid out for one resident wave per SIMD (a second wave on
    // a SIMD costs as much as it brings), and the host sizes its launches accordingly — but where the waves of a
    // launch LAND is the dispatcher's business: with kernels that fit a SIMD twice (<= 256 registers) it put two
    // waves on some SIMDs and none on others whenever the launch before had left its round-robin state "odd"
    // (a two-lane launch of 1024 waves: 27 ms after another 1024-wave launch, 48 ms after one of 1536 waves or as
    // the first launch of a process; profiles/r04_dispatch.txt).  A wave that owns more than half of the SIMD's 512
    // registers cannot share it: the one-lane kernels do anyway (256 VGPRs + AGPRs); the others claim accumulation
    // registers they never touch.  (PIPE workgroups are placed by their LDS footprint instead.)
    // TWO WAVES PER SIMD (MIN_WAVES_PER_SIMD = 2; lane kernels on two, four and eight lanes per utterance that hold their
    // state in <= 256 registers without a scratch segment): the lone tolerance-mode wave leaves the VALU idle a quarter of
    // the time, and two of them on a SIMD render 20 - 30 % more per second than one after the other (twice as much where
    // events are dense: a slow sample is latency); the exact kernels gain 9 - 15 % on aligned batches and up to 30 % on
    // speech-like ones — where the waves spill (one lane per utterance) they lose 14 % instead (profiles/r04_two_waves.txt,
    // r05_two_waves.txt).  The host asks for these instantiations only for launches of more waves than the device has
    // SIMDs, where the dispatcher's placement has nothing to get wrong.
    if constexpr (L > 1 && !PIPE && MIN_WAVES_PER_SIMD == 1) asm volatile("" ::: "a127");

    // every wave of the block works alone on its own S utterances and its own
    // slice of LDS: there is no inter-wave communication and no block barrier
    // L >= 4: the lanes park all eight band-pass outputs of a sample and the left fold runs at
    // flush time, spread over time steps, instead of a serial chain of L DPP hops per sample
    constexpr bool FOLD_IN_FLUSH = L >= 4;
    constexpr int STAGE_FLOATS = FOLD_IN_FLUSH ? T * S * NFA : T * SP;
    __shared__ float stage_all[PIPE ? 1 : WAVES][STAGE_FLOATS];
    __shared__ uint32_t cnt_all[PIPE ? 1 : WAVES][S];
    const int wave = threadIdx.x / 64;
    float *stage = stage_all[PIPE ? 0 : wave];
    uint32_t *cnt = cnt_all[PIPE ? 0 : wave];
    // PIPE: role 0 renders (owns stage, counts, output), role 1 carries the per-utterance chain,
    // roles 2 and 3 prepare coefficients; `emit` is constant true otherwise
    const int role = PIPE ? wave : 0;
    const bool emit = !PIPE || role == 0;

    const int lane = threadIdx.x % 64;
    const int slot = lane / L;
    const int j = lane % L;
    const int f0 = j * FPL;
    // SPLIT: the waves of the last chunk (longest fast-forward) start first
    const uint32_t split_groups = SPLIT ? (A.n_utt + S - 1) / S : 1u;
    const uint32_t chunk = SPLIT ? A.split_chunks - 1u - blockIdx.x / split_groups : 0u;
    // FOLD (two waves per SIMD, a launch of at most two rounds of the device): every wave is resident from the start, so
    // nothing evens out the SIMDs' loads afterwards, and the launch slots are filled longest utterances first — the
    // workgroups of the second round take their slots in reverse order, so that the SIMD with the longest rows of the first
    // round gets the shortest of the second
    uint32_t block_id = blockIdx.x;
    if constexpr (!SPLIT && !PIPE && !STREAM && MIN_WAVES_PER_SIMD == 2)
        if (A.fold_from != 0u && block_id >= A.fold_from) block_id = gridDim.x - 1u - (block_id - A.fold_from);
    // PIPE, one-shot: a workgroup may hold fewer utterances than it has slots for (SynthArgs::pipe_fill)
    const uint32_t pipe_fill = PIPE && !STREAM && A.pipe_fill != 0u ? A.pipe_fill : (uint32_t)S;
    const uint32_t u0 = SPLIT ? (blockIdx.x % split_groups) * S
                              : PIPE ? blockIdx.x * pipe_fill : (block_id * WAVES + wave) * S;
    // which utterance this slot renders: its position in the launch, or — ragged batches — the host's
    // length-sorted assignment (A.perm), so that the lanes of a wave end together; rows, lengths and
    // per-utterance inputs always belong to utterance `u`.  (A launch may cover a range of the slots only —
    // A.perm then points at the range's first slot and `u` may well exceed A.n_utt: `slot_used` says whether
    // the slot renders, never a comparison of `u`.)
    const bool slot_used = (!PIPE || (uint32_t)slot < pipe_fill) && u0 + slot < A.n_utt;
    const uint32_t u = !slot_used ? A.n_utt : (A.perm ? A.perm[u0 + slot] : u0 + slot);
    bool done = !slot_used;
    if constexpr (SPLIT && GRAIL_SPLIT_SKIP) {
        // a chunk's lane whose utterance ends before the chunk begins (the host's upper bound of its length) has nothing
        // to render — no fast-forward, no warm-up; a wave of such lanes is gone at once.  Rows that differ in length are
        // launched longest first, so the waves of the later chunks are the ones that go, and the host lays out more,
        // shorter chunks than the device has SIMDs for (launch_plan.cpp).
        if (A.len_bound != nullptr && slot_used && chunk > 0u && A.len_bound[u] <= A.split_bounds[chunk]) done = true;
        if (__builtin_amdgcn_ballot_w64(!done) == 0) {
#ifdef GRAIL_FAST_PROF
            if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(A.truncated + 8) + 30, 1ull);
#endif
            return;
        }
    }
    const uint32_t uc = done ? 0u : u;
    __shared__ uint32_t rowid_all[PIPE ? 1 : WAVES][S];
    uint32_t *rowid = rowid_all[PIPE ? 0 : wave];
    if (A.perm && j == L - 1) rowid[slot] = uc;

    uint32_t vid = A.voice_ids ? A.voice_ids[uc] : 0u;
    if (vid >= A.n_voices) vid = 0u;
    const DevVoice VO = A.voices[vid];
    const bool phoneme_mode = A.phoneme_mode != 0;
    const float *__restrict__ elems = A.elems;

    // ---- Sequencer state: IntoSequencer::sequence, src/lib.rs:941-949
    // live streams (STREAM kernels only): the utterance's segments sit in a ring and more may be appended between
    // launches; seg_pos then counts the segments pulled so far and seg_end those appended so far
    // Only the general resumable instantiations (ANYBL: what a live stream always runs — nothing is known about the
    // segments to come) carry the ring code: the lean ones stay what they were (a few instructions more in the general
    // step moved the code of the calm loops and cost the lean one-lane stream kernel 9 %, same instruction counts).
    // Everything else about the ring is worked out where a segment is pulled — a rare path.
    constexpr bool LIVE = STREAM && ANYBL;
    uint32_t seg_pos = (LIVE && A.ring_cap != 0u) ? 0u : A.seg_offsets[uc];
    const uint32_t seg_end = (LIVE && A.ring_cap != 0u) ? A.seg_counts[uc] : A.seg_offsets[uc + 1];
    Seg cur, nxt;
    cur.some = false; cur.elem = -1; cur.length = 0.0f; cur.blend_length = 1.0f; cur.frequency = 0.0f;
    nxt = cur;
    float clk = 0.0f;                        // Sequencer.time
    const float dt = 1.0f / VO.sample_rate;  // :944
    Part<NV, V> X, Y;                        // emitted elem = X*(1-alpha) + Y*alpha
    silent_part(X);
    silent_part(Y);
    float blend_length = 1.0f;
    float inv_blend_length = 1.0f;           // exact when blend_length is +-2^k
    bool blend_pow2 = true;
    bool blend_div_ok = false;               // ANYBL: clk / blend_length may use the short exact division
    bool silent_pair = true;
    bool pair_safe = false;                  // every division of this pair may use div_exact<true>

    // ---- Jitter state: IntoJitter::jitter, src/lib.rs:786-797.  One seed is
    // threaded through the three constructors (2 + 16 + 16 draws), each noise
    // then keeps its own copy of the state.  The three noises share one phase
    // sequence (same start, same increment), kept once.
    uint32_t seed = A.seeds ? A.seeds[uc] : 0u;
    // a resumed stream call loads all of this from its state block: skip the 34 draws
    const bool fresh_start = !(STREAM && A.state && A.resume);
    float fn_cur = 0.0f, fn_next = 0.0f;
    if (fresh_start) {
        fn_cur = lcg_f32(seed);              // ValueNoise::new :228-229
        fn_next = lcg_f32(seed);
    }
    uint32_t fn_state = seed;
    V ff_cur[NV], ff_next[NV], fa_cur[NV], fa_next[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        ff_cur[k] = vsplat(0.0f, ff_cur[k]); ff_next[k] = ff_cur[k];
        fa_cur[k] = ff_cur[k]; fa_next[k] = ff_cur[k];
    }
    uint32_t ff_state = seed;
    if (fresh_start) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {           // ArrayValueNoise::new :275-278
        const float c0 = lcg_f32(seed);
        const float n0 = lcg_f32(seed);
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (i == f0 + k * W + c) { vset(ff_cur[k], c, c0); vset(ff_next[k], c, n0); }
    }
    ff_state = seed;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const float c0 = lcg_f32(seed);
        const float n0 = lcg_f32(seed);
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (i == f0 + k * W + c) { vset(fa_cur[k], c, c0); vset(fa_next[k], c, n0); }
    }
    }
    uint32_t fa_state = seed;
    float jphase = 0.0f;
    const float jinc = VO.jitter_frequency;
    const float d_freq = VO.jitter_delta_frequency;
    const float d_ffreq = VO.jitter_delta_formant_frequency;
    const float amp_scale = 0.5f * VO.jitter_delta_amplitude;   // :769

    // ---- Synthesize state: IntoSynthesize::synthesize, src/lib.rs:587-596
    float phase = 0.0f;
    V st_a[NV], st_b[NV], st_c[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        st_a[k] = vsplat(0.0f, st_a[k]);
        st_b[k] = st_a[k];
        st_c[k] = st_a[k];
    }
    uint32_t noise_seed = 0u;                // :594

    const uint64_t cap = A.cap;              // samples this launch may write per row (<= out_stride)
    const uint32_t cap32 = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;   // n_out is 32-bit
    // where this launch stops rendering an utterance that has not ended: a stream call at its quota, a chunk
    // lane at the first sample of the next chunk (the last chunk runs to the end of the row)
    constexpr bool PAUSES = STREAM || SPLIT;
    const uint32_t chunk_lo = SPLIT ? A.split_bounds[chunk] : 0u;
    const uint32_t pause_at = SPLIT ? (chunk + 1u < A.split_chunks ? A.split_bounds[chunk + 1u] : 0xFFFFFFFFu) : cap32;
    const uint32_t room_end = SPLIT ? (pause_at < cap32 ? pause_at : cap32) : cap32;
    bool paused = false;
    uint32_t n_out = 0;
    uint32_t slow_steps = 0;                 // wave-steps that took the IEEE-division body
    uint32_t fast_tiles = 0, general_steps = 0;   // statistics: tiles rendered by fast_tile, general steps taken
#ifdef GRAIL_FAST_PROF
    unsigned long long prof_c[32] = {};
    unsigned long long prof_t0 = clock64();
    const unsigned long long prof_start = prof_t0;
    unsigned long long prof_lane_levels = 0;
#endif
    bool truncated = false;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(A.out) & 15u) == 0) && ((A.out_stride & 3u) == 0);
    const bool vec16_ok = ((reinterpret_cast<uintptr_t>(A.out_pcm16) & 7u) == 0) && ((A.out_stride & 3u) == 0);

    // false while the lane's segment pair needs the IEEE-division body or has a blend
    // length that is not a power of two: such lanes always take the general step
    bool quiet_ok = false;

    // the chain has returned None (persistent; `done` also covers pauses).  A slot without an utterance counts as finished:
    // it will not render in this launch or any other, so it rides along in calm tiles and runs like an ended utterance (a
    // lone stream in a workgroup laid out for sixteen used to keep its wave out of every calm tile)
    bool finished = !slot_used;
    // one-shot batches: the lane's upper formants have amplitude +0 in every phoneme of the voice table (phoneme
    // batches: looked up here) or in every elem of the batch (caller-built elems: formants 5-8, established by the
    // host at upload — half_capable), so nothing in this launch can ever make them audible
    bool upper_never_live = false;
    if constexpr (HALF && NV >= 2 && !STREAM) {
        if (phoneme_mode) {
            upper_never_live = true;
#pragma unroll
            for (int p = 0; p < NUM_VOICED; ++p)
#pragma unroll
                for (int i = (NV / 2) * W; i < NV * W; ++i)
                    upper_never_live = upper_never_live &&
                        (__float_as_uint(elems[(size_t)(VO.elem_base + p) * ELEM_FLOATS + F_AMP + f0 + i]) == 0u);
        } else {
            upper_never_live = A.half_capable != 0u && f0 + (NV / 2) * W >= NF / 2;
        }
    }
    bool smooth_uniform = false; // this pair: X.smooth and Y.smooth are each one number for all formants
    bool upper_silent = false;   // this pair: the lane's upper NV/2 formant vectors are silent
    auto update_silent = [&]() __attribute__((always_inline)) {
        if constexpr (HALF && NV >= 2)
            upper_silent = A.skip_silent && pair_safe && (STREAM || upper_never_live) &&
                           upper_half_is_silent<NV, W>(X, Y, st_a, st_b, st_c, amp_scale);
        bool su = true;
        const uint32_t xs0 = __float_as_uint(vget(X.smooth[0], 0));
        const uint32_t ys0 = __float_as_uint(vget(Y.smooth[0], 0));
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int c = 0; c < W; ++c)
                su = su && (__float_as_uint(vget(X.smooth[k], c)) == xs0) &&
                     (__float_as_uint(vget(Y.smooth[k], c)) == ys0);
        smooth_uniform = su;
    };

    // (cur, nxt) -> X, Y, blend constants: the match of Sequencer::next resolved once per pair
    auto setup_pair = [&]() __attribute__((always_inline)) {
        // the match at :891-931, resolved once per segment pair
        const bool has_b = cur.elem >= 0;
        const bool has_c = nxt.some && nxt.elem >= 0;
        blend_length = cur.blend_length;
        silent_pair = !has_b && !has_c;
        if (has_b && has_c) {          // c.blend(b, alpha)  :897-903
            load_part<NV, W>(X, elems, nxt.elem, f0);
            load_part<NV, W>(Y, elems, cur.elem, f0);
            X.frequency = nxt.frequency;
            Y.frequency = cur.frequency;
        } else if (has_b) {            // b.copy_silent().blend(b, alpha)  :906-912
            load_part<NV, W>(Y, elems, cur.elem, f0);
            Y.frequency = cur.frequency;
            X = Y;
#pragma unroll
            for (int k = 0; k < NV; ++k) X.amp[k] = vsplat(0.0f, X.amp[k]);
        } else if (has_c) {            // c.blend(c.copy_silent(), alpha)  :915-921
            load_part<NV, W>(X, elems, nxt.elem, f0);
            X.frequency = nxt.frequency;
            Y = X;
#pragma unroll
            for (int k = 0; k < NV; ++k) Y.amp[k] = vsplat(0.0f, Y.amp[k]);
        } else {                       // SynthesisElem::silent()  :924-927
            silent_part(X);
            silent_part(Y);
        }
        // clk / 2^k == clk * 2^-k for every clk (same real number, same rounding)
        const uint32_t blb = __float_as_uint(blend_length);
        const uint32_t ble = (blb >> 23) & 0xFFu;
        blend_pow2 = ((blb & 0x7FFFFFu) == 0u) && ble >= 1u && ble <= 253u;
        inv_blend_length = 1.0f / blend_length;       // IEEE: RN(1/b), what div_exact<true> starts from
        // any other blend length: q = clk*RN(1/b), r = fma(-b, q, clk), q' = fma(r, RN(1/b), q) is the
        // correctly rounded clk/b while b and clk are in the proven window (tools/div_exhaustive.hip);
        // clk <= length, and steps whose clk is below the window take the general step
        if constexpr (ANYBL)
            blend_div_ok = (blend_length >= 0x1p-59f) && (blend_length <= 0x1p59f) &&
                           (cur.length <= 0x1p59f) && (dt >= 0x1p-59f);
    };

    constexpr bool streaming = STREAM;       // a separate instantiation: the one-shot kernel
                                             // carries none of the state traffic or its registers
    // (PIPE: the four waves of a workgroup carry ONE set of utterances — every wave loads the set's state, the rendering
    // wave, whose filters are the live ones, saves it; the block is the lane kernels' of the same L, utterance by utterance:
    // a stream may take either from call to call)
    const size_t state_lane = PIPE ? (size_t)blockIdx.x * 64 + lane : (size_t)(blockIdx.x * WAVES + wave) * 64 + lane;
    auto visit_state = [&](auto &io) __attribute__((always_inline)) {
        io(seg_pos);
        io(cur.some); io(cur.elem); io(cur.length); io(cur.blend_length); io(cur.frequency);
        io(nxt.some); io(nxt.elem); io(nxt.length); io(nxt.blend_length); io(nxt.frequency);
        io(clk); io(pair_safe); io(finished);
        io(fn_cur); io(fn_next); io(fn_state); io(ff_state); io(fa_state); io(jphase);
        io(phase); io(noise_seed);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            io(ff_cur[k]); io(ff_next[k]); io(fa_cur[k]); io(fa_next[k]);
            io(st_a[k]); io(st_b[k]); io(st_c[k]);
        }
    };
    if (streaming && A.state && A.resume && slot_used) {
        StateIO<true> io{A.state, A.state_stride, state_lane};
        visit_state(io);
        done = finished;
        if (cur.some) setup_pair();
        quiet_ok = pair_safe && (blend_pow2 || blend_div_ok);
        update_silent();
    }
