// streams.cpp — resumable streams (the per-utterance state stays in HBM between pulls) and live streams (segments
// appended while samples are pulled: the lazy source of examples/interactive.rs:31-38).
#include "api_internal.hpp"

using namespace grail;
using namespace grail::host;

extern "C" {

int grail_stream_open(grail_ctx *ctx, const grail_batch *batch, grail_stream **out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if ((rc = check_ready(ctx, batch))) return rc;
    grail_stream *s = new (std::nothrow) grail_stream();
    if (!s) return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->batch = batch;
    s->half_capable = batch_half_capable(ctx, batch);
    s->any_blend = batch->any_blend;
    s->live4 = batch_live4_any_blend(ctx, batch);      // (the lean resumable kernels exist for every blend length)
    s->voices_epoch = ctx->voices_epoch;
    s->L = ctx->lanes_option ? ctx->lanes_option : auto_lanes_per_utt(batch->n_utt, ctx_simds(ctx));
    if (s->live4 && !ctx->lanes_option && ctx->pipeline_option &&
        (int64_t)(((uint64_t)batch->n_utt + 15) / 16) <= pipe4_groups(ctx)) {
        s->L = 4;                            // the pipelined workgroups of four formants (stream_next), 16 utterances each
    } else if (s->live4) {
        if (s->L == 8) s->live4 = false;     // eight lanes per utterance need eight formants to lay out
        else if (!ctx->lanes_option)         // same rule over four formants: the widest one-wave-per-SIMD mapping
            s->L = ((uint64_t)batch->n_utt * 4 + 63) / 64 <= ctx_simds(ctx) ? 4 : ((uint64_t)batch->n_utt * 2 + 63) / 64 <= ctx_simds(ctx) ? 2 : 1;
    }
    s->lanes = state_lanes(batch->n_utt, s->L);
    const size_t bytes = (size_t)state_words(s->L) * s->lanes * sizeof(uint32_t);
    hipError_t e = hipMalloc((void **)&s->d_state, bytes ? bytes : 4);
    if (e != hipSuccess) {
        delete s;
        return hip_fail(e, "stream state allocation");
    }
    *out = s;
    return GRAIL_OK;
}

static int stream_next(grail_ctx *ctx, grail_stream *stream, uint32_t max_samples, float *out_dev,
                       int16_t *out_pcm16_dev, uint64_t out_stride, uint32_t *out_len_dev)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!stream) return fail(GRAIL_ERR_INVALID_ARG, "stream is NULL");
    const grail_batch *batch = stream->batch;
    if ((rc = check_ready(ctx, batch))) return rc;
    if (max_samples > out_stride) return fail(GRAIL_ERR_INVALID_ARG, "max_samples exceeds out_stride");
    if (batch->n_utt == 0) return GRAIL_OK;
    if (!out_dev && !out_pcm16_dev && max_samples) return fail(GRAIL_ERR_INVALID_ARG, "out_dev is NULL");
    SynthArgs a{};
    a.segs = batch->d_segs;
    a.seg_offsets = batch->d_offsets;
    a.ring_cap = stream->ring_cap;
    a.seg_counts = stream->d_counts;
    a.seg_open = stream->d_open;
    a.seg_consumed = stream->d_consumed;
    a.voice_ids = batch->d_voice_ids;
    a.seeds = batch->d_seeds;
    a.perm = batch->d_perm;
    a.elems = batch->phoneme_mode ? ctx->d_voice_elems : batch->d_elems;
    a.voices = ctx->d_voices;
    a.out = out_dev;
    a.out_pcm16 = out_pcm16_dev;
    a.out_len = out_len_dev;
    a.truncated = ctx->d_truncated;
    a.out_stride = out_stride;
    a.cap = max_samples;
    a.n_utt = batch->n_utt;
    a.n_voices = (uint32_t)ctx->voices.size();
    a.phoneme_mode = batch->phoneme_mode ? 1u : 0u;
    a.skip_silent = ctx->skip_silent_option ? 1u : 0u;
    if (stream->voices_epoch != ctx->voices_epoch)
        return fail(GRAIL_ERR_INVALID_ARG, "the voice table changed since the stream was opened");
    a.half_capable = stream->half_capable ? 1u : 0u;
    a.any_blend = stream->any_blend ? 1u : 0u;
    a.live4 = stream->live4 ? 1u : 0u;
    // (may change between calls: both flavours share the state layout)
    // (sharper voices: the second tier has one-lane kernels only — streams on a wider mapping run the exact kernels)
    const int tier = fast_tier(ctx, batch);
    a.fast = tier == 1 ? 1u : (tier == 2 && stream->L == 1) ? 2u : 0u;
    // A few hundred to a few thousand streams leave most SIMDs idle on the lane kernels: they take the pipelined
    // workgroups (four waves share 16 / 8 utterances; synth_kernel<..., PIPE>) wherever a one-shot batch of this size
    // would — same state block as the lane kernels of the stream's mapping, exact arithmetic (a tolerance request is
    // served by them too: their bits are the reference's, and at these sizes they are the faster kernels)
    const uint64_t cus = (uint64_t)ctx->cus;
    if (ctx->pipeline_option && !ctx->lanes_option) {
        if (stream->live4 && stream->L == 4 && (int64_t)(((uint64_t)batch->n_utt + 15) / 16) <= pipe4_groups(ctx))
            a.pipe = ctx->pipe_round32 && ((uint64_t)batch->n_utt + 15) / 16 <= cus ? 2u : 1u;
        else if (!stream->live4 && stream->L == 8 && (int64_t)(((uint64_t)batch->n_utt + 7) / 8) <= pipe8_groups(ctx))
            a.pipe = ctx->pipe_round32 && ((uint64_t)batch->n_utt + 7) / 8 <= cus ? 2u : 1u;
        if (a.pipe) a.fast = 0u;
    }
    a.state = stream->d_state;
    a.state_stride = stream->lanes;
    a.resume = stream->started ? 1u : 0u;
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    hipError_t e = launch_synth(a, stream->L, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "synth kernel launch");
    ctx->last_kernel = last_kernel_name();
    ctx->last_formants = a.live4 ? 4 : 8;
    ctx->last_lanes = stream->L;
    ctx->last_pipe = a.pipe ? 1 : 0;
    ctx->last_fast = (int)a.fast;
    ctx->last_blocks = 1;
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->have_timing = true;
    stream->started = true;
    return GRAIL_OK;
}

int grail_stream_next_async(grail_ctx *ctx, grail_stream *stream, uint32_t max_samples,
                            float *out_dev, uint64_t out_stride, uint32_t *out_len_dev)
{
    return stream_next(ctx, stream, max_samples, out_dev, nullptr, out_stride, out_len_dev);
}

int grail_stream_next_pcm16_async(grail_ctx *ctx, grail_stream *stream, uint32_t max_samples,
                                  int16_t *out_dev, uint64_t out_stride, uint32_t *out_len_dev)
{
    return stream_next(ctx, stream, max_samples, nullptr, out_dev, out_stride, out_len_dev);
}

int grail_stream_close(grail_ctx *ctx, grail_stream *stream)
{
    if (!stream) return GRAIL_OK;
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (stream->d_state) (void)hipFree(stream->d_state);
    if (stream->d_counts) (void)hipFree(stream->d_counts);
    if (stream->d_open) (void)hipFree(stream->d_open);
    if (stream->d_consumed) (void)hipFree(stream->d_consumed);
    if (stream->d_new) (void)hipFree(stream->d_new);
    if (stream->d_new_elems) (void)hipFree(stream->d_new_elems);
    if (stream->d_new_offs) (void)hipFree(stream->d_new_offs);
    for (int i = 0; i < 2; ++i) {
        if (stream->h_stage[i]) (void)hipHostFree(stream->h_stage[i]);
        if (stream->ev_stage[i]) (void)hipEventDestroy(stream->ev_stage[i]);
    }
    if (stream->own) {
        free_batch_buffers(stream->own);
        delete stream->own;
    }
    delete stream;
    return GRAIL_OK;
}

// ---- live streams: the lazy source of examples/interactive.rs:31-38 -------------------------------------------------
int grail_stream_open_live(grail_ctx *ctx, uint32_t n_utt, const uint32_t *voice_ids, const uint32_t *jitter_seeds,
                           uint32_t ring_segments, int caller_built_elems, grail_stream **out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_utt == 0) return fail(GRAIL_ERR_INVALID_ARG, "a live stream needs at least one utterance");
    if (ring_segments == 0) ring_segments = 64;
    if (ring_segments < 4 || (ring_segments & (ring_segments - 1)) != 0 || ring_segments > 65536)
        return fail(GRAIL_ERR_INVALID_ARG, "ring_segments must be a power of two, 4 .. 65536 (0: 64)");
    if ((uint64_t)n_utt * ring_segments > 0x7FFFFFFFull) return fail(GRAIL_ERR_INVALID_ARG, "n_utt x ring_segments exceeds 2^31");
    if (ctx->voices.empty() || !ctx->d_voices) return fail(GRAIL_ERR_NO_VOICES, "call grail_set_voices first");
    grail_batch *b = new (std::nothrow) grail_batch();
    grail_stream *s = new (std::nothrow) grail_stream();
    if (!b || !s) {
        delete b;
        delete s;
        return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    s->own = b;
    s->batch = b;
    s->ring_cap = ring_segments;
    b->phoneme_mode = !caller_built_elems;
    b->any_blend = true;          // what will be appended is not known: the general instantiations
    b->plain = false;
    b->n_utt = n_utt;
    b->max_voice_id = 0;
    b->used_voices.assign(1, 0u);
    if (voice_ids) {
        for (uint32_t u = 0; u < n_utt; ++u) b->max_voice_id = std::max(b->max_voice_id, voice_ids[u]);
        b->used_voices.assign(voice_ids, voice_ids + n_utt);
        std::sort(b->used_voices.begin(), b->used_voices.end());
        b->used_voices.erase(std::unique(b->used_voices.begin(), b->used_voices.end()), b->used_voices.end());
    }
    const size_t ring_rows = (size_t)n_utt * ring_segments;
    hipError_t e = hipSuccess;
    auto zeroed = [&](void **p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 4);
        if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes ? bytes : 4, ctx->stream);
    };
    zeroed((void **)&b->d_segs, ring_rows * sizeof(DevSeg));
    if (caller_built_elems) zeroed((void **)&b->d_elems, ring_rows * ELEM_FLOATS * sizeof(float));
    zeroed((void **)&s->d_counts, (size_t)n_utt * 4);
    zeroed((void **)&s->d_consumed, (size_t)n_utt * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_open, (size_t)n_utt * 4);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)s->d_open, 1, n_utt, ctx->stream);
    if (e == hipSuccess && voice_ids) {
        e = hipMalloc((void **)&b->d_voice_ids, (size_t)n_utt * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(b->d_voice_ids, voice_ids, (size_t)n_utt * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess && jitter_seeds) {
        e = hipMalloc((void **)&b->d_seeds, (size_t)n_utt * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(b->d_seeds, jitter_seeds, (size_t)n_utt * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    s->half_capable = batch_half_capable(ctx, b);
    s->any_blend = true;
    s->live4 = false;
    s->voices_epoch = ctx->voices_epoch;
    s->L = ctx->lanes_option ? ctx->lanes_option : auto_lanes_per_utt(n_utt, ctx_simds(ctx));
    s->lanes = state_lanes(n_utt, s->L);
    const size_t bytes = (size_t)state_words(s->L) * s->lanes * sizeof(uint32_t);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_state, bytes ? bytes : 4);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // voice_ids / jitter_seeds are the caller's
    if (e != hipSuccess) {
        const int st = hip_fail(e, "live stream allocation");
        const std::string keep = last_error();
        grail_stream_close(ctx, s);
        last_error() = keep;
        return st;
    }
    s->appended.assign(n_utt, 0u);
    s->consumed.assign(n_utt, 0u);
    s->open.assign(n_utt, 1);
    if (caller_built_elems) {
        s->last_elem.resize(n_utt);
        s->last_has.assign(n_utt, 0);
    }
    *out = s;
    return GRAIL_OK;
}

// common part of the two append calls: room in the rings, upload, scatter on the device
static int live_append(grail_ctx *ctx, grail_stream *s, const std::vector<DevSeg> &segs, const float *elems,
                       const uint32_t *seg_offsets)
{
    const uint32_t n_utt = s->own->n_utt, cap = s->ring_cap;
    const uint32_t n_new = seg_offsets[n_utt];
    if (n_new == 0) return GRAIL_OK;
    // The Sequencer holds on to its current and next segment (and their elems in the ring are re-read when a call
    // resumes): a ring keeps the last two segments pulled besides everything pending.
    uint32_t full_at = 0;
    auto fits = [&]() {
        for (uint32_t u = 0; u < n_utt; ++u) {
            const uint32_t add = seg_offsets[u + 1] - seg_offsets[u];
            if (add && (uint64_t)s->appended[u] - s->consumed[u] + add + 2u > cap) {
                full_at = u;
                return false;
            }
        }
        return true;
    };
    if (!fits()) {
        // what the host knows of the Sequencers' progress is a lower bound: ask the device
        HIP_TRY(hipMemcpyAsync(s->consumed.data(), s->d_consumed, (size_t)n_utt * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (!fits()) {
            char msg[200];
            std::snprintf(msg, sizeof msg, "the segment ring of utterance %u of the live stream is full (%u pending of %u): pull "
                          "samples first (or open the stream with a larger ring_segments); nothing was appended",
                          full_at, s->appended[full_at] - s->consumed[full_at], cap);
            return fail(GRAIL_ERR_BUFFER_TOO_SMALL, msg);
        }
    }
    for (uint32_t u = 0; u < n_utt; ++u)
        if (seg_offsets[u + 1] > seg_offsets[u] && !s->open[u])
            return fail(GRAIL_ERR_INVALID_ARG, "an utterance of the live stream has been finished: nothing can be appended to it");
    hipError_t e = hipSuccess;
    if (s->new_cap < n_new || (elems && !s->d_new_elems)) {
        const size_t cap_new = std::max<size_t>(std::max<size_t>(n_new, 2 * s->new_cap), 64);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (s->d_new) (void)hipFree(s->d_new);
        if (s->d_new_elems) (void)hipFree(s->d_new_elems);
        s->d_new = nullptr;
        s->d_new_elems = nullptr;
        s->new_cap = 0;
        e = hipMalloc((void **)&s->d_new, cap_new * sizeof(DevSeg));
        if (e == hipSuccess && elems) e = hipMalloc((void **)&s->d_new_elems, cap_new * ELEM_FLOATS * sizeof(float));
        if (e == hipSuccess && !s->d_new_offs) e = hipMalloc((void **)&s->d_new_offs, ((size_t)n_utt + 1) * 4);
        if (e != hipSuccess) return hip_fail(e, "grail_stream_append staging");
        s->new_cap = cap_new;
    }
    // The caller's (and this function's) buffers go through a pinned buffer of the stream's own, so that the call can return
    // with its copies queued: with tens of thousands of live streams fed a phoneme at a time a synchronisation per append
    // — behind every kernel queued so far — was the pace of the whole session.
    const size_t b_segs = (size_t)n_new * sizeof(DevSeg), b_offs = ((size_t)n_utt + 1) * 4;
    const size_t b_elems = elems ? (size_t)n_new * ELEM_FLOATS * sizeof(float) : 0;
    const int slot = s->stage_next;
    if (s->stage_busy[slot]) {
        HIP_TRY(hipEventSynchronize(s->ev_stage[slot]));          // the append before last has left this buffer
        s->stage_busy[slot] = false;
    }
    if (s->h_stage_cap[slot] < b_segs + b_offs + b_elems) {
        if (s->h_stage[slot]) (void)hipHostFree(s->h_stage[slot]);
        s->h_stage[slot] = nullptr;
        s->h_stage_cap[slot] = 0;
        const size_t want = 2 * (b_segs + b_offs + b_elems);
        e = hipHostMalloc(&s->h_stage[slot], want, hipHostMallocDefault);
        if (e != hipSuccess) return hip_fail(e, "grail_stream_append pinned staging");
        s->h_stage_cap[slot] = want;
    }
    if (!s->ev_stage[slot]) HIP_TRY(hipEventCreateWithFlags(&s->ev_stage[slot], hipEventDisableTiming));
    char *h = static_cast<char *>(s->h_stage[slot]);
    std::memcpy(h, segs.data(), b_segs);
    std::memcpy(h + b_segs, seg_offsets, b_offs);
    if (elems) std::memcpy(h + b_segs + b_offs, elems, b_elems);
    e = hipMemcpyAsync(s->d_new, h, b_segs, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_new_offs, h + b_segs, b_offs, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && elems) e = hipMemcpyAsync(s->d_new_elems, h + b_segs + b_offs, b_elems, hipMemcpyHostToDevice, ctx->stream);
    // (stream order: behind every kernel that still reads the rings, ahead of every kernel that will)
    if (e == hipSuccess)
        e = launch_ring_append(s->own->d_segs, s->own->d_elems, s->d_counts, cap, s->d_new, elems ? s->d_new_elems : nullptr,
                               s->d_new_offs, n_utt, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(s->ev_stage[slot], ctx->stream);
    if (e != hipSuccess) {
        // some of the copies may be queued and still reading the pinned buffer, which the next append would overwrite (the
        // slot is not marked busy and stage_next does not move): wait for them before handing the failure back.  Whether
        // the scatter ran is unknown; `appended` is left alone, so the host's view stays a lower bound and the stream is
        // best closed by the caller.
        (void)hipStreamSynchronize(ctx->stream);
        return hip_fail(e, "grail_stream_append");
    }
    s->stage_busy[slot] = true;
    s->stage_next = slot ^ 1;
    for (uint32_t u = 0; u < n_utt; ++u) s->appended[u] += seg_offsets[u + 1] - seg_offsets[u];
    return GRAIL_OK;
}

static int live_check(grail_ctx *ctx, grail_stream *stream, const uint32_t *seg_offsets, bool elems)
{
    if (!stream || !stream->own) return fail(GRAIL_ERR_INVALID_ARG, "not a live stream (grail_stream_open_live)");
    if (stream->own->phoneme_mode == elems)
        return fail(GRAIL_ERR_INVALID_ARG, elems ? "the live stream takes PhonemeElems (grail_stream_append)"
                                                 : "the live stream takes SequenceElems (grail_stream_append_elems)");
    if (stream->voices_epoch != ctx->voices_epoch)
        return fail(GRAIL_ERR_INVALID_ARG, "the voice table changed since the stream was opened");
    uint32_t n = 0;
    return check_offsets(seg_offsets, stream->own->n_utt, &n);
}

int grail_stream_append(grail_ctx *ctx, grail_stream *stream, const grail_phoneme_elem *segs, const uint32_t *seg_offsets)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if ((rc = live_check(ctx, stream, seg_offsets, false))) return rc;
    const uint32_t n_new = seg_offsets[stream->own->n_utt];
    if (n_new && !segs) return fail(GRAIL_ERR_INVALID_ARG, "segs is NULL");
    std::vector<DevSeg> ds(n_new);
    for (uint32_t i = 0; i < n_new; ++i) {
        if (segs[i].phoneme < 0 || segs[i].phoneme >= GRAIL_PH_COUNT)
            return fail(GRAIL_ERR_INVALID_ARG, "phoneme discriminant out of range");
        std::memcpy(&ds[i], &segs[i], sizeof(DevSeg));
    }
    return live_append(ctx, stream, ds, nullptr, seg_offsets);
}

int grail_stream_append_elems(grail_ctx *ctx, grail_stream *stream, const grail_sequence_elem *segs,
                              const uint32_t *seg_offsets)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if ((rc = live_check(ctx, stream, seg_offsets, true))) return rc;
    const uint32_t n_utt = stream->own->n_utt, n_new = seg_offsets[n_utt];
    if (n_new && !segs) return fail(GRAIL_ERR_INVALID_ARG, "segs is NULL");
    std::vector<DevSeg> ds(n_new);
    std::vector<float> elems((size_t)(n_new ? n_new : 1) * ELEM_FLOATS);
    for (uint32_t i = 0; i < n_new; ++i) {
        ds[i].elem = segs[i].has_elem ? 0 : -1;        // (the device writes the ring row)
        ds[i].length = segs[i].length;
        ds[i].blend_length = segs[i].blend_length;
        ds[i].frequency = segs[i].elem.frequency;
        std::memcpy(&elems[(size_t)i * ELEM_FLOATS], &segs[i].elem, sizeof(grail_synthesis_elem));
    }
    // the sharpness fast arithmetic is served up to: every two consecutive elems of an utterance, the seam to what was
    // appended before included (grail_batch_upload_elems does the same over a closed list)
    double sharp = stream->own->elems_sharpness;
    std::vector<grail_synthesis_elem> last = stream->last_elem;
    std::vector<uint8_t> has = stream->last_has;
    for (uint32_t u = 0; u < n_utt; ++u)
        for (uint32_t i = seg_offsets[u]; i < seg_offsets[u + 1]; ++i) {
            if (segs[i].has_elem) {
                grail_synthesis_elem pair[2] = {segs[i].elem, segs[i].elem};
                size_t n_pair = 1;
                if (has[u]) pair[n_pair++] = last[u];
                sharp = std::fmax(sharp, elems_sharpness(pair, n_pair));
                last[u] = segs[i].elem;
            }
            has[u] = segs[i].has_elem ? 1 : 0;
        }
    rc = live_append(ctx, stream, ds, elems.data(), seg_offsets);
    if (rc) return rc;
    stream->own->elems_sharpness = sharp;
    stream->last_elem.swap(last);
    stream->last_has.swap(has);
    return GRAIL_OK;
}

int grail_stream_finish(grail_ctx *ctx, grail_stream *stream, const uint8_t *which)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!stream || !stream->own) return fail(GRAIL_ERR_INVALID_ARG, "not a live stream (grail_stream_open_live)");
    const uint32_t n_utt = stream->own->n_utt;
    std::vector<uint32_t> open(n_utt);
    for (uint32_t u = 0; u < n_utt; ++u) {
        if (!which || which[u]) stream->open[u] = 0;
        open[u] = stream->open[u];
    }
    HIP_TRY(hipMemcpyAsync(stream->d_open, open.data(), (size_t)n_utt * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GRAIL_OK;
}

int grail_stream_pending(grail_ctx *ctx, grail_stream *stream, uint32_t *pending)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!stream || !stream->own) return fail(GRAIL_ERR_INVALID_ARG, "not a live stream (grail_stream_open_live)");
    if (!pending) return fail(GRAIL_ERR_INVALID_ARG, "pending is NULL");
    const uint32_t n_utt = stream->own->n_utt;
    HIP_TRY(hipMemcpyAsync(stream->consumed.data(), stream->d_consumed, (size_t)n_utt * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (uint32_t u = 0; u < n_utt; ++u) pending[u] = stream->appended[u] - stream->consumed[u];
    return GRAIL_OK;
}

}  // extern "C"
