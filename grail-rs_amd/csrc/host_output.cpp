// host_output.cpp — the one-call forms of the C ABI and their host destinations.
#include "api_internal.hpp"

using namespace grail;
using namespace grail::host;

// ---- the one-call forms with a host destination: render and copy back, overlapped -------------
// Rows are rendered in blocks (kernel on ctx->stream into one of two device buffers) while the
// previous block travels to the host on a second stream.  A destination that is pinned /
// registered host memory (grail_host_alloc, hipHostMalloc, hipHostRegister) receives the
// device-to-host copies directly; a pageable destination is fed through a ring of pinned staging
// buffers that copier threads empty into it (one memcpy thread cannot keep up with PCIe Gen5).
// Same bytes as the device-resident result; rows end in zeros.
namespace {

constexpr size_t PIECE_BYTES = 32u << 20;   // pinned staging granularity
constexpr int N_PIECES = 12;                // ring size
constexpr int N_COPIERS = 8;               // memcpy threads for a pageable destination

struct HostPipe {
    hipStream_t copy_stream = nullptr;
    hipEvent_t rendered[2] = {nullptr, nullptr};   // block in dev[i] is complete (on ctx->stream)
    hipEvent_t drained[2] = {nullptr, nullptr};    // dev[i] has been copied out (on copy_stream)
    hipEvent_t landed[N_PIECES] = {};              // pinned piece i holds its data
    void *dev[2] = {nullptr, nullptr};
    size_t dev_bytes = 0;
    void *pin[N_PIECES] = {};
    bool have_pins = false;
};

void pipe_destroy(HostPipe *p)
{
    if (!p) return;
    for (int i = 0; i < 2; ++i) {
        if (p->dev[i]) (void)hipFree(p->dev[i]);
        if (p->rendered[i]) (void)hipEventDestroy(p->rendered[i]);
        if (p->drained[i]) (void)hipEventDestroy(p->drained[i]);
    }
    for (int i = 0; i < N_PIECES; ++i) {
        if (p->pin[i]) (void)hipHostFree(p->pin[i]);
        if (p->landed[i]) (void)hipEventDestroy(p->landed[i]);
    }
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    delete p;
}

// created on first use and kept in the context: pinned allocations cost tens of milliseconds
int pipe_get(grail_ctx *ctx, size_t block_bytes, bool need_pins, HostPipe **out)
{
    HostPipe *p = (HostPipe *)ctx->host_pipe;
    if (!p) {
        // built in a local and published to the context only when every stream and event exists: a
        // half-built pipe left behind by a failed create would make later calls use null handles
        p = new (std::nothrow) HostPipe();
        if (!p) return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
        hipError_t e = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; ++i) {
            e = hipEventCreateWithFlags(&p->rendered[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p->drained[i], hipEventDisableTiming);
        }
        for (int i = 0; i < N_PIECES && e == hipSuccess; ++i)
            e = hipEventCreateWithFlags(&p->landed[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            pipe_destroy(p);
            return hip_fail(e, "host-output pipe");
        }
        ctx->host_pipe = p;
    }
    if (p->dev_bytes < block_bytes) {
        HIP_TRY(hipStreamSynchronize(p->copy_stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 2; ++i) {
            if (p->dev[i]) (void)hipFree(p->dev[i]);
            p->dev[i] = nullptr;
        }
        p->dev_bytes = 0;
        for (int i = 0; i < 2; ++i) HIP_TRY(hipMalloc(&p->dev[i], block_bytes));
        p->dev_bytes = block_bytes;
    }
    if (need_pins && !p->have_pins) {
        for (int i = 0; i < N_PIECES; ++i) HIP_TRY(hipHostMalloc(&p->pin[i], PIECE_BYTES, hipHostMallocDefault));
        p->have_pins = true;
    }
    *out = p;
    return GRAIL_OK;
}

bool is_pinned_host(const void *ptr)
{
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof attr);
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();     // a plain malloc pointer is "invalid value": not an error here
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// the copier side of the pinned ring: each job is one piece that has been ENQUEUED for copy-out
struct CopyJob {
    int piece;
    char *dst;
    size_t bytes;
};
struct CopyRing {
    std::mutex m;
    std::condition_variable cv_job, cv_free;
    std::deque<CopyJob> jobs;
    bool piece_busy[N_PIECES] = {};
    bool closing = false;
    hipError_t error = hipSuccess;
};

void copier_main(int device, HostPipe *p, CopyRing *r)
{
    (void)hipSetDevice(device);
    for (;;) {
        CopyJob job;
        {
            std::unique_lock<std::mutex> lk(r->m);
            r->cv_job.wait(lk, [&] { return !r->jobs.empty() || r->closing; });
            if (r->jobs.empty()) return;
            job = r->jobs.front();
            r->jobs.pop_front();
        }
        const hipError_t e = hipEventSynchronize(p->landed[job.piece]);
        if (e == hipSuccess) std::memcpy(job.dst, p->pin[job.piece], job.bytes);
        {
            std::lock_guard<std::mutex> lk(r->m);
            if (e != hipSuccess && r->error == hipSuccess) r->error = e;
            r->piece_busy[job.piece] = false;
        }
        r->cv_free.notify_all();
    }
}

// ELEM = 4: f32 rows, 2: i16 PCM rows
int render_to_host(grail_ctx *ctx, grail_batch *b, uint32_t n_utt, void *out, size_t elem, uint64_t out_stride,
                   uint32_t *out_len)
{
    const size_t row_bytes = (size_t)out_stride * elem;
    // (copier threads memcpy into `out`: a NULL destination must fail here, not fault there)
    if (!out && n_utt && out_stride) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    uint32_t *d_len = nullptr;
    hipError_t e = hipSuccess;
    if (n_utt) e = hipMalloc((void **)&d_len, (size_t)n_utt * sizeof(uint32_t));
    if (e != hipSuccess) return hip_fail(e, "out_len allocation");
    int rc = GRAIL_OK, sync_rc = GRAIL_OK;
    if (n_utt && row_bytes) {
        // block = up to 4096 rows and 2 GB: big enough for the kernel to outrun PCIe (a 4096-utterance
        // launch renders > 100 GB/s of PCM), small enough for two of them to sit beside the batch
        uint64_t rows = std::min<uint64_t>(4096, std::max<uint64_t>(1, (2ull << 30) / row_bytes));
        rows = std::min<uint64_t>(rows, n_utt);
        const bool direct = is_pinned_host(out);
        HostPipe *p = nullptr;
        rc = pipe_get(ctx, rows * row_bytes, !direct, &p);
        CopyRing ring;
        std::vector<std::thread> copiers;
        if (!rc && !direct)
            for (int i = 0; i < N_COPIERS; ++i) copiers.emplace_back(copier_main, ctx->device, p, &ring);
        int piece_next = 0;
        uint32_t blk = 0;
        for (uint64_t first = 0; !rc && first < n_utt; first += rows, ++blk) {
            const uint32_t count = (uint32_t)std::min<uint64_t>(rows, n_utt - first);
            const int slot = blk & 1;
            const size_t bytes = (size_t)count * row_bytes;
            // the kernel may not overwrite dev[slot] before its previous contents have left
            if (blk >= 2) e = hipStreamWaitEvent(ctx->stream, p->drained[slot], 0);
            if (e == hipSuccess) e = hipMemsetAsync(p->dev[slot], 0, bytes, ctx->stream);
            if (e != hipSuccess) { rc = hip_fail(e, "block set-up"); break; }
            rc = synthesize_rows(ctx, b, elem == 4 ? (float *)p->dev[slot] : nullptr,
                                 elem == 2 ? (int16_t *)p->dev[slot] : nullptr, out_stride, d_len + first,
                                 (uint32_t)first, count, rows < n_utt ? (uint32_t)rows : 0u);   // (one block: plan freely)
            if (rc) break;
            e = hipEventRecord(p->rendered[slot], ctx->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(p->copy_stream, p->rendered[slot], 0);
            char *dst = (char *)out + (size_t)first * row_bytes;
            if (e == hipSuccess && direct) {
                e = hipMemcpyAsync(dst, p->dev[slot], bytes, hipMemcpyDeviceToHost, p->copy_stream);
            } else if (e == hipSuccess) {
                for (size_t off = 0; off < bytes && e == hipSuccess; off += PIECE_BYTES) {
                    const size_t n = std::min(PIECE_BYTES, bytes - off);
                    const int piece = piece_next;
                    piece_next = (piece_next + 1) % N_PIECES;
                    {
                        std::unique_lock<std::mutex> lk(ring.m);
                        ring.cv_free.wait(lk, [&] { return !ring.piece_busy[piece]; });
                        ring.piece_busy[piece] = true;
                        if (ring.error != hipSuccess) e = ring.error;
                    }
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(p->pin[piece], (char *)p->dev[slot] + off, n, hipMemcpyDeviceToHost,
                                           p->copy_stream);
                    if (e == hipSuccess) e = hipEventRecord(p->landed[piece], p->copy_stream);
                    {
                        std::lock_guard<std::mutex> lk(ring.m);
                        if (e == hipSuccess) ring.jobs.push_back(CopyJob{piece, dst + off, n});
                        else ring.piece_busy[piece] = false;
                    }
                    ring.cv_job.notify_one();
                }
            }
            if (e == hipSuccess) e = hipEventRecord(p->drained[slot], p->copy_stream);
            if (e != hipSuccess) rc = hip_fail(e, "device-to-host pipeline");
        }
        {
            std::lock_guard<std::mutex> lk(ring.m);
            ring.closing = true;
        }
        ring.cv_job.notify_all();
        for (auto &t : copiers) t.join();
        if (p) {
            e = hipStreamSynchronize(p->copy_stream);
            if (!rc && e != hipSuccess) rc = hip_fail(e, "device-to-host pipeline");
        }
        if (!rc && ring.error != hipSuccess) rc = hip_fail(ring.error, "device-to-host pipeline");
    }
    if (!rc) {
        sync_rc = grail_sync(ctx);
        if (sync_rc != GRAIL_OK && sync_rc != GRAIL_ERR_BUFFER_TOO_SMALL) rc = sync_rc;
    }
    if (!rc && out_len && n_utt) {
        e = hipMemcpyAsync(out_len, d_len, (size_t)n_utt * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hip_fail(e, "out_len copy");
    }
    if (d_len) (void)hipFree(d_len);
    return rc ? rc : sync_rc;
}

}  // namespace

namespace grail {
namespace host {

void pipe_destroy_opaque(void *p) { pipe_destroy((HostPipe *)p); }

// the front of examples/cli.rs:176-179 for n texts: transcribe + intonate with the voice each text names
int say_segments(const std::vector<grail_voice> &voices, const char *const *texts_utf8, uint32_t n_texts,
                 const uint32_t *voice_ids, std::vector<grail_phoneme_elem> &segs, std::vector<uint32_t> &offs)
{
    if (voices.empty()) return fail(GRAIL_ERR_NO_VOICES, "call grail_set_voices first");
    if (n_texts && !texts_utf8) return fail(GRAIL_ERR_INVALID_ARG, "texts is NULL");
    segs.clear();
    offs.assign(1, 0u);
    for (uint32_t i = 0; i < n_texts; ++i) {
        const uint32_t vid = voice_ids ? voice_ids[i] : 0u;
        if (vid >= voices.size()) return fail(GRAIL_ERR_INVALID_ARG, "voice id out of range");
        if (!texts_utf8[i]) return fail(GRAIL_ERR_INVALID_ARG, "a text is NULL");
        uint32_t n = 0;
        grail_text_to_phoneme_elems(&voices[vid], texts_utf8[i], nullptr, 0, &n);
        const size_t base = segs.size();
        segs.resize(base + n);
        int rc = grail_text_to_phoneme_elems(&voices[vid], texts_utf8[i], segs.data() + base, n, &n);
        if (rc) return fail(rc, "transcription failed");
        offs.push_back((uint32_t)segs.size());
    }
    return GRAIL_OK;
}

}  // namespace host
}  // namespace grail

extern "C" {

static int run_one_call(grail_ctx *ctx, grail_batch *b, uint32_t n_utt, float *out,
                        uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    if (!(flags & GRAIL_OUT_DEVICE)) return render_to_host(ctx, b, n_utt, out, sizeof(float), out_stride, out_len);
    int rc = GRAIL_OK;
    uint32_t *d_len = nullptr;
    hipError_t e = hipSuccess;
    if (n_utt) e = hipMalloc((void **)&d_len, (size_t)n_utt * sizeof(uint32_t));
    if (e != hipSuccess) rc = hip_fail(e, "output allocation");
    if (!rc) rc = grail_batch_synthesize_async(ctx, b, out, out_stride, d_len);
    int sync_rc = GRAIL_OK;
    if (!rc) {
        sync_rc = grail_sync(ctx);
        if (sync_rc != GRAIL_OK && sync_rc != GRAIL_ERR_BUFFER_TOO_SMALL) rc = sync_rc;
    }
    if (!rc && out_len && n_utt) {
        e = hipMemcpyAsync(out_len, d_len, (size_t)n_utt * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hip_fail(e, "out_len copy");
    }
    if (d_len) (void)hipFree(d_len);
    return rc ? rc : sync_rc;
}

int grail_synthesize_batch(grail_ctx *ctx, const grail_phoneme_elem *segs,
                           const uint32_t *seg_offsets, const uint32_t *voice_ids,
                           const uint32_t *jitter_seeds, uint32_t n_utt, float *out,
                           uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    grail_batch *b = nullptr;
    int rc = grail_batch_upload(ctx, segs, seg_offsets, voice_ids, jitter_seeds, n_utt, &b);
    if (rc) return rc;
    rc = run_one_call(ctx, b, n_utt, out, out_stride, out_len, flags);
    const std::string keep = last_error();
    grail_batch_free(ctx, b);
    last_error() = keep;
    return rc;
}

int grail_synthesize_batch_elems(grail_ctx *ctx, const grail_sequence_elem *segs,
                                 const uint32_t *seg_offsets, const uint32_t *voice_ids,
                                 const uint32_t *jitter_seeds, uint32_t n_utt, float *out,
                                 uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    grail_batch *b = nullptr;
    int rc = grail_batch_upload_elems(ctx, segs, seg_offsets, voice_ids, jitter_seeds, n_utt, &b);
    if (rc) return rc;
    rc = run_one_call(ctx, b, n_utt, out, out_stride, out_len, flags);
    const std::string keep = last_error();
    grail_batch_free(ctx, b);
    last_error() = keep;
    return rc;
}

int grail_synthesize_batch_pcm16(grail_ctx *ctx, const grail_phoneme_elem *segs,
                                 const uint32_t *seg_offsets, const uint32_t *voice_ids,
                                 const uint32_t *jitter_seeds, uint32_t n_utt, int16_t *out,
                                 uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    grail_batch *b = nullptr;
    int rc = grail_batch_upload(ctx, segs, seg_offsets, voice_ids, jitter_seeds, n_utt, &b);
    if (rc) return rc;
    int sync_rc = GRAIL_OK;
    // the conversion is part of the synthesis kernel's tile flush: 2 B per sample of HBM and PCIe traffic
    if (!(flags & GRAIL_OUT_DEVICE)) {
        rc = render_to_host(ctx, b, n_utt, out, sizeof(int16_t), out_stride, out_len);
    } else {
        uint32_t *d_len = nullptr;
        hipError_t e = hipSuccess;
        if (n_utt) e = hipMalloc((void **)&d_len, (size_t)n_utt * sizeof(uint32_t));
        if (e != hipSuccess) rc = hip_fail(e, "pcm16 output allocation");
        if (!rc) rc = grail_batch_synthesize_pcm16_async(ctx, b, out, out_stride, d_len);
        if (!rc) {
            sync_rc = grail_sync(ctx);
            if (sync_rc != GRAIL_OK && sync_rc != GRAIL_ERR_BUFFER_TOO_SMALL) rc = sync_rc;
        }
        if (!rc && out_len && n_utt) {
            e = hipMemcpyAsync(out_len, d_len, (size_t)n_utt * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = hip_fail(e, "out_len copy");
        }
        if (d_len) (void)hipFree(d_len);
    }
    const std::string keep = last_error();
    grail_batch_free(ctx, b);
    last_error() = keep;
    return rc ? rc : sync_rc;
}

int grail_say_batch(grail_ctx *ctx, const char *const *texts_utf8, uint32_t n_texts,
                    const uint32_t *voice_ids, const uint32_t *jitter_seeds, float *out,
                    uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    std::vector<grail_phoneme_elem> segs;
    std::vector<uint32_t> offs;
    int rc = say_segments(ctx->voices, texts_utf8, n_texts, voice_ids, segs, offs);
    if (rc) return rc;
    return grail_synthesize_batch(ctx, segs.data(), offs.data(), voice_ids, jitter_seeds, n_texts, out,
                                  out_stride, out_len, flags);
}

}  // extern "C"
