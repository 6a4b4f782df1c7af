// node.cpp — one call, the whole node (grail_node_*; SURVEY.md §8e "one process, 8 devices").
// The reference host makes one call for its whole job (examples/cli.rs:175-184) and every utterance carries its own
// state (src/lib.rs:470-488, 724-748, 839-854): a batch shards over the GPUs of a node with no exchange step.  A node
// is a grail_ctx and a host thread per device; a call hands every thread the contiguous shard of grail_shard_range and
// the thread runs the ordinary one-context call on it, into its slice of the caller's host buffer.  The only collective
// is the broadcast of the voice table (comm.cpp).  Nothing here touches a sample.
#include <chrono>
#include <functional>

#include "api_internal.hpp"

using namespace grail;
using namespace grail::host;

namespace {

// One host thread per device slot: HIP's current device is per thread, and eight shards must be uploaded, launched and
// drained concurrently.  A job is a closure that returns a status; the message of a failure is the thread's own
// grail_last_error(), carried back to the caller's.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int status = GRAIL_OK;
    std::string error;
    float ms = 0.0f;

    void main()
    {
        for (;;) {
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has_job || quit; });
                if (!has_job) return;
                f = std::move(job);
                has_job = false;
            }
            const auto t0 = std::chrono::steady_clock::now();
            last_error().clear();
            const int rc = f();
            const float took = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            {
                std::lock_guard<std::mutex> lk(m);
                status = rc;
                error = rc ? last_error() : std::string();
                ms = took;
                done = true;
            }
            cv.notify_all();
        }
    }
    void start() { th = std::thread([this] { main(); }); }
    void submit(std::function<int()> f)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(f);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return status;
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

}  // namespace

struct grail_node {
    std::vector<int> devices;
    std::vector<grail_ctx *> ctxs;
    std::vector<Worker *> workers;
    std::vector<float> last_ms;
    bool comm_ready = false;
    int voices_without_rccl = 0;
};

namespace {

// run job(i) on every slot's thread (slots with skip[i] sit the call out), wait for all, merge the outcomes
int run_all(grail_node *node, const std::function<int(uint32_t)> &job, const std::vector<uint8_t> *skip = nullptr)
{
    const uint32_t n = (uint32_t)node->ctxs.size();
    node->last_ms.assign(n, 0.0f);
    for (uint32_t i = 0; i < n; ++i)
        if (!skip || !(*skip)[i]) node->workers[i]->submit([&job, i] { return job(i); });
    int hard = GRAIL_OK, soft = GRAIL_OK;
    std::string msg;
    for (uint32_t i = 0; i < n; ++i) {
        if (skip && (*skip)[i]) continue;
        const int rc = node->workers[i]->wait();
        node->last_ms[i] = node->workers[i]->ms;
        if (rc == GRAIL_ERR_BUFFER_TOO_SMALL) {
            if (!soft) {
                soft = rc;
                if (!hard) msg = "device[" + std::to_string(i) + "] = " + std::to_string(node->devices[i]) + ": " + node->workers[i]->error;
            }
        } else if (rc && !hard) {
            hard = rc;
            msg = "device[" + std::to_string(i) + "] = " + std::to_string(node->devices[i]) + ": " + node->workers[i]->error;
        }
    }
    if (hard) return fail(hard, msg);
    if (soft) return fail(soft, msg);
    return GRAIL_OK;
}

struct ShardView {
    grail_node_shard s;
    std::vector<uint32_t> offs;
};

int shard_views(grail_node *node, const uint32_t *seg_offsets, uint32_t n_utt, std::vector<ShardView> &views,
                std::vector<uint8_t> &skip)
{
    uint32_t n_segs = 0;
    int rc = check_offsets(seg_offsets, n_utt, &n_segs);
    if (rc) return rc;
    const uint32_t n = (uint32_t)node->ctxs.size();
    views.resize(n);
    skip.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        grail_node_shard probe;
        if ((rc = grail_node_shard_of(seg_offsets, n_utt, i, n, &probe, nullptr, 0))) return rc;
        views[i].offs.resize((size_t)probe.rows + 1);
        if ((rc = grail_node_shard_of(seg_offsets, n_utt, i, n, &views[i].s, views[i].offs.data(), probe.rows + 1))) return rc;
        skip[i] = probe.rows == 0;
    }
    return GRAIL_OK;
}

int check_node(grail_node *node, uint32_t flags)
{
    if (!node) return fail(GRAIL_ERR_INVALID_ARG, "node is NULL");
    if (flags & GRAIL_OUT_DEVICE)
        return fail(GRAIL_ERR_INVALID_ARG, "a node call writes host memory: GRAIL_OUT_DEVICE names no one device");
    return GRAIL_OK;
}

}  // namespace

extern "C" {

int grail_node_shard_of(const uint32_t *seg_offsets, uint64_t n_utt, uint32_t index, uint32_t n_devices,
                        grail_node_shard *shard, uint32_t *rebased_offsets, uint64_t cap)
{
    if (!shard || n_devices == 0 || index >= n_devices) return fail(GRAIL_ERR_INVALID_ARG, "grail_node_shard_of: bad index / n_devices / shard");
    if (n_utt && !seg_offsets) return fail(GRAIL_ERR_INVALID_ARG, "seg_offsets is NULL");
    uint64_t b = 0, e = 0;
    grail_shard_range(n_utt, index, n_devices, &b, &e);
    shard->first_row = b;
    shard->rows = e - b;
    shard->first_seg = seg_offsets ? seg_offsets[b] : 0u;
    shard->n_segs = seg_offsets ? seg_offsets[e] - seg_offsets[b] : 0u;
    if (rebased_offsets) {
        if (cap < shard->rows + 1) return fail(GRAIL_ERR_BUFFER_TOO_SMALL, "rebased_offsets holds fewer than rows + 1 entries");
        for (uint64_t r = 0; r <= shard->rows; ++r) rebased_offsets[r] = seg_offsets ? seg_offsets[b + r] - shard->first_seg : 0u;
    }
    return GRAIL_OK;
}

int grail_node_create(const int *devices, uint32_t n_devices, grail_node **out)
{
    if (!out) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_devices == 0 || n_devices > 1024) return fail(GRAIL_ERR_INVALID_ARG, "n_devices must be 1 .. 1024");
    grail_node *node = new (std::nothrow) grail_node();
    if (!node) return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
    node->devices.resize(n_devices);
    for (uint32_t i = 0; i < n_devices; ++i) node->devices[i] = devices ? devices[i] : (int)i;
    node->ctxs.assign(n_devices, nullptr);
    node->last_ms.assign(n_devices, 0.0f);
    for (uint32_t i = 0; i < n_devices; ++i) {
        Worker *w = new (std::nothrow) Worker();
        if (!w) {
            grail_node_destroy(node);
            return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
        }
        node->workers.push_back(w);
        w->start();
    }
    // every context is created on the thread that will drive it
    const int rc = run_all(node, [node](uint32_t i) { return grail_create(node->devices[i], &node->ctxs[i]); });
    if (rc) {
        const std::string keep = last_error();
        grail_node_destroy(node);
        return fail(rc, keep);
    }
    *out = node;
    return GRAIL_OK;
}

int grail_node_destroy(grail_node *node)
{
    if (!node) return GRAIL_OK;
    for (size_t i = 0; i < node->workers.size(); ++i) {
        Worker *w = node->workers[i];
        grail_ctx *ctx = i < node->ctxs.size() ? node->ctxs[i] : nullptr;
        if (ctx) {
            w->submit([ctx] { return grail_destroy(ctx); });
            (void)w->wait();
        }
        w->stop();
        delete w;
    }
    delete node;
    return GRAIL_OK;
}

uint32_t grail_node_size(const grail_node *node) { return node ? (uint32_t)node->ctxs.size() : 0u; }

int grail_node_context(grail_node *node, uint32_t index, grail_ctx **ctx)
{
    if (!node || !ctx || index >= node->ctxs.size()) return fail(GRAIL_ERR_INVALID_ARG, "grail_node_context: bad node / index / ctx");
    *ctx = node->ctxs[index];
    return GRAIL_OK;
}

int grail_node_set_voices(grail_node *node, const grail_voice *voices, uint32_t n_voices)
{
    if (!node) return fail(GRAIL_ERR_INVALID_ARG, "node is NULL");
    if (!voices || n_voices == 0) return fail(GRAIL_ERR_INVALID_ARG, "no voices given");
    if (node->voices_without_rccl)
        return run_all(node, [&](uint32_t i) { return grail_set_voices(node->ctxs[i], voices, n_voices); });
    if (!node->comm_ready) {
        const int rc = comm_init_all(node->ctxs.data(), node->devices.data(), (uint32_t)node->ctxs.size());
        if (rc) return rc;
        node->comm_ready = true;
    }
    // slot 0 holds the table; one ncclBroadcast on every slot's stream carries it from its HBM to the others'
    std::vector<uint8_t> only0(node->ctxs.size(), 1);
    only0[0] = 0;
    int rc = run_all(node, [&](uint32_t i) { return grail_set_voices(node->ctxs[i], voices, n_voices); }, &only0);
    if (rc) return rc;
    return run_all(node, [&](uint32_t i) { return grail_broadcast_voices(node->ctxs[i], n_voices, 0); });
}

int grail_node_set_option(grail_node *node, const char *name, int64_t value)
{
    if (!node || !name) return fail(GRAIL_ERR_INVALID_ARG, "NULL argument");
    if (std::strcmp(name, "node_voices_without_rccl") == 0) {
        node->voices_without_rccl = value ? 1 : 0;
        return GRAIL_OK;
    }
    for (grail_ctx *ctx : node->ctxs) {
        const int rc = grail_set_option(ctx, name, value);
        if (rc) return rc;
    }
    return GRAIL_OK;
}

int grail_node_get_option(grail_node *node, const char *name, int64_t *value)
{
    if (!node || !name || !value) return fail(GRAIL_ERR_INVALID_ARG, "NULL argument");
    if (std::strcmp(name, "node_devices") == 0) {
        *value = (int64_t)node->ctxs.size();
        return GRAIL_OK;
    }
    if (std::strcmp(name, "node_voices_without_rccl") == 0) {
        *value = node->voices_without_rccl;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "node_rccl_ranks") == 0) {
        uint32_t least = 0xFFFFFFFFu;
        for (grail_ctx *ctx : node->ctxs) {
            uint32_t ranks = 0;
            const int rc = grail_comm_info(ctx, &ranks, nullptr);
            if (rc) return rc;
            least = std::min(least, ranks);
        }
        *value = (int64_t)least;
        return GRAIL_OK;
    }
    return grail_get_option(node->ctxs[0], name, value);
}

int grail_node_synthesize_batch(grail_node *node, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                                const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt, float *out,
                                uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    int rc = check_node(node, flags);
    if (rc) return rc;
    std::vector<ShardView> v;
    std::vector<uint8_t> skip;
    if ((rc = shard_views(node, seg_offsets, n_utt, v, skip))) return rc;
    return run_all(node, [&](uint32_t i) {
        const grail_node_shard &s = v[i].s;
        return grail_synthesize_batch(node->ctxs[i], segs ? segs + s.first_seg : nullptr, v[i].offs.data(),
                                      voice_ids ? voice_ids + s.first_row : nullptr,
                                      jitter_seeds ? jitter_seeds + s.first_row : nullptr, (uint32_t)s.rows,
                                      out ? out + s.first_row * out_stride : nullptr, out_stride,
                                      out_len ? out_len + s.first_row : nullptr, flags);
    }, &skip);
}

int grail_node_synthesize_batch_elems(grail_node *node, const grail_sequence_elem *segs, const uint32_t *seg_offsets,
                                      const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt,
                                      float *out, uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    int rc = check_node(node, flags);
    if (rc) return rc;
    std::vector<ShardView> v;
    std::vector<uint8_t> skip;
    if ((rc = shard_views(node, seg_offsets, n_utt, v, skip))) return rc;
    return run_all(node, [&](uint32_t i) {
        const grail_node_shard &s = v[i].s;
        return grail_synthesize_batch_elems(node->ctxs[i], segs ? segs + s.first_seg : nullptr, v[i].offs.data(),
                                            voice_ids ? voice_ids + s.first_row : nullptr,
                                            jitter_seeds ? jitter_seeds + s.first_row : nullptr, (uint32_t)s.rows,
                                            out ? out + s.first_row * out_stride : nullptr, out_stride,
                                            out_len ? out_len + s.first_row : nullptr, flags);
    }, &skip);
}

int grail_node_synthesize_batch_pcm16(grail_node *node, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                                      const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt,
                                      int16_t *out, uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    int rc = check_node(node, flags);
    if (rc) return rc;
    std::vector<ShardView> v;
    std::vector<uint8_t> skip;
    if ((rc = shard_views(node, seg_offsets, n_utt, v, skip))) return rc;
    return run_all(node, [&](uint32_t i) {
        const grail_node_shard &s = v[i].s;
        return grail_synthesize_batch_pcm16(node->ctxs[i], segs ? segs + s.first_seg : nullptr, v[i].offs.data(),
                                            voice_ids ? voice_ids + s.first_row : nullptr,
                                            jitter_seeds ? jitter_seeds + s.first_row : nullptr, (uint32_t)s.rows,
                                            out ? out + s.first_row * out_stride : nullptr, out_stride,
                                            out_len ? out_len + s.first_row : nullptr, flags);
    }, &skip);
}

int grail_node_synthesize_batch_device(grail_node *node, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                                       const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt,
                                       float *const *out_dev, uint64_t out_stride, uint32_t *out_len)
{
    if (!node) return fail(GRAIL_ERR_INVALID_ARG, "node is NULL");
    if (!out_dev) return fail(GRAIL_ERR_INVALID_ARG, "out_dev is NULL");
    std::vector<ShardView> v;
    std::vector<uint8_t> skip;
    int rc = shard_views(node, seg_offsets, n_utt, v, skip);
    if (rc) return rc;
    for (size_t i = 0; i < v.size(); ++i)
        if (!skip[i] && !out_dev[i] && out_stride)
            return fail(GRAIL_ERR_INVALID_ARG, "out_dev[" + std::to_string(i) + "] is NULL and the slot has rows to render");
    return run_all(node, [&](uint32_t i) {
        const grail_node_shard &s = v[i].s;
        return grail_synthesize_batch(node->ctxs[i], segs ? segs + s.first_seg : nullptr, v[i].offs.data(),
                                      voice_ids ? voice_ids + s.first_row : nullptr,
                                      jitter_seeds ? jitter_seeds + s.first_row : nullptr, (uint32_t)s.rows, out_dev[i],
                                      out_stride, out_len ? out_len + s.first_row : nullptr, GRAIL_OUT_DEVICE);
    }, &skip);
}

int grail_node_say_batch(grail_node *node, const char *const *texts_utf8, uint32_t n_texts, const uint32_t *voice_ids,
                         const uint32_t *jitter_seeds, float *out, uint64_t out_stride, uint32_t *out_len,
                         uint32_t flags)
{
    int rc = check_node(node, flags);
    if (rc) return rc;
    std::vector<grail_phoneme_elem> segs;
    std::vector<uint32_t> offs;
    // (every context holds the same table: slot 0's host copy serves the transcription)
    if ((rc = say_segments(node->ctxs[0]->voices, texts_utf8, n_texts, voice_ids, segs, offs))) return rc;
    return grail_node_synthesize_batch(node, segs.data(), offs.data(), voice_ids, jitter_seeds, n_texts, out, out_stride,
                                       out_len, flags);
}

int grail_node_lengths(grail_node *node, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                       const uint32_t *voice_ids, uint32_t n_utt, uint32_t max_len, uint32_t *out_len)
{
    if (!node) return fail(GRAIL_ERR_INVALID_ARG, "node is NULL");
    if (n_utt && !out_len) return fail(GRAIL_ERR_INVALID_ARG, "out_len is NULL");
    std::vector<ShardView> v;
    std::vector<uint8_t> skip;
    int rc = shard_views(node, seg_offsets, n_utt, v, skip);
    if (rc) return rc;
    return run_all(node, [&](uint32_t i) {
        const grail_node_shard &s = v[i].s;
        grail_batch *b = nullptr;
        int r = grail_batch_upload(node->ctxs[i], segs ? segs + s.first_seg : nullptr, v[i].offs.data(),
                                   voice_ids ? voice_ids + s.first_row : nullptr, nullptr, (uint32_t)s.rows, &b);
        if (r) return r;
        r = grail_batch_lengths(node->ctxs[i], b, max_len, out_len + s.first_row);
        const std::string keep = last_error();
        grail_batch_free(node->ctxs[i], b);
        last_error() = keep;
        return r;
    }, &skip);
}

int grail_node_last_shard_ms(grail_node *node, float *ms, uint32_t cap)
{
    if (!node || (cap && !ms)) return fail(GRAIL_ERR_INVALID_ARG, "NULL argument");
    for (uint32_t i = 0; i < cap && i < node->last_ms.size(); ++i) ms[i] = node->last_ms[i];
    return GRAIL_OK;
}

int grail_node_host_alloc(grail_node *node, size_t bytes, void **out)
{
    if (!node || !out) return fail(GRAIL_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    int rc = bind(node->ctxs[0]);
    if (rc) return rc;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable));
    return GRAIL_OK;
}

int grail_node_host_free(grail_node *node, void *ptr)
{
    if (!node) return fail(GRAIL_ERR_INVALID_ARG, "node is NULL");
    int rc = bind(node->ctxs[0]);
    if (rc) return rc;
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return GRAIL_OK;
}

}  // extern "C"
