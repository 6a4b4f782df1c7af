// grail_api.cpp — the C ABI (include/grail_hip.h) over the HIP kernels: contexts, voice tables, options, HBM-resident
// batches, synchronisation and the device utilities.  Host orchestration only (with voice_analysis.cpp, launch_plan.cpp,
// synthesize.cpp, streams.cpp, host_output.cpp, comm.cpp): no arithmetic of the hot path happens here and
// there is no CPU fallback: without a HIP device every compute call fails.
#include "api_internal.hpp"

using namespace grail;
using namespace grail::host;

static_assert(sizeof(grail_synthesis_elem) == 196, "SynthesisElem is 49 x f32");
static_assert(sizeof(grail_phoneme_elem) == sizeof(DevSeg), "PhonemeElem uploads as-is");
static_assert(offsetof(grail_phoneme_elem, phoneme) == offsetof(DevSeg, elem), "layout");
static_assert(offsetof(grail_phoneme_elem, frequency) == offsetof(DevSeg, frequency), "layout");
static_assert(sizeof(grail_voice) == 4 + 2 * 196 + 5 * 4, "Voice layout");
static_assert(GRAIL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace grail {
namespace host {

namespace {
thread_local std::string g_last_error;
// Voice-table epochs are drawn from ONE counter for the whole process: a batch (or a stream) remembers the epoch of the
// table something was computed against, and a batch may be rendered by another context than the one that uploaded it —
// two contexts that each installed one table must not both be "epoch 1".
std::atomic<uint64_t> g_epoch{0};
}

std::string &last_error() { return g_last_error; }

int fail(int status, const std::string &msg)
{
    g_last_error = msg;
    return status;
}

int hip_fail(hipError_t e, const char *what)
{
    if (e == hipErrorOutOfMemory)
        return fail(GRAIL_ERR_OUT_OF_MEMORY, std::string(what) + ": " + hipGetErrorString(e));
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
        return fail(GRAIL_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    return fail(GRAIL_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

int bind(grail_ctx *ctx)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    return GRAIL_OK;
}

void free_batch_buffers(grail_batch *b)
{
    free_plan_cache(b->plan_cache);
    b->plan_cache = nullptr;
    for (grail_batch &g : b->groups) {          // (views: they share the device buffers below)
        free_plan_cache(g.plan_cache);
        g.plan_cache = nullptr;
    }
    b->groups.clear();
    if (b->d_segs) (void)hipFree(b->d_segs);
    if (b->d_offsets) (void)hipFree(b->d_offsets);
    if (b->d_voice_ids) (void)hipFree(b->d_voice_ids);
    if (b->d_seeds) (void)hipFree(b->d_seeds);
    if (b->d_perm) (void)hipFree(b->d_perm);
    for (PackedPerm &pp : b->packed)
        if (pp.d_perm) (void)hipFree(pp.d_perm);
    b->packed.clear();
    if (b->d_len_bound) (void)hipFree(b->d_len_bound);
    if (b->d_elems) (void)hipFree(b->d_elems);
}

// clk / 2^k is clk * 2^-k exactly; any other blend length needs the division code of the kernel
bool blend_is_pow2(float blend_length)
{
    uint32_t bits;
    std::memcpy(&bits, &blend_length, sizeof bits);
    const uint32_t e = (bits >> 23) & 0xFFu;
    return (bits & 0x7FFFFFu) == 0u && e >= 1u && e <= 253u;
}

int check_offsets(const uint32_t *seg_offsets, uint32_t n_utt, uint32_t *n_segs)
{
    if (!seg_offsets) return fail(GRAIL_ERR_INVALID_ARG, "seg_offsets is NULL");
    for (uint32_t u = 0; u < n_utt; ++u)
        if (seg_offsets[u + 1] < seg_offsets[u])
            return fail(GRAIL_ERR_INVALID_ARG, "seg_offsets must be non-decreasing");
    *n_segs = seg_offsets[n_utt];
    return GRAIL_OK;
}

// What the launch policy asks of an utterance's segments (the batch's fields of the same names are these over all rows)
struct RowStats {
    bool plain = true;          // every length / blend length / pitch finite, blend lengths > 0
    bool any_blend = false;     // some blend length is not +-2^k
    float min_length = INFINITY, min_pitch = INFINITY, seconds = 0.0f;
    uint32_t segs = 0, kinks = 0;   // segments; those whose alpha = min(clk / blend_length, 1) has a kink (blend_length < length)
    double bound_samples = 0.0;     // grail_length_bound of the row at the table's highest sample rate (INFINITY: none)
};

// What one segment can last at most, in samples.  Every advance adds the segment's length to the clock (:873 / :882), every
// sample takes dt off it in f32 (:861): a step lowers the clock by at least dt - ulp(length) / 2 (the clock is at most the
// length while the segment lasts), so the segment is over after length / (dt - ulp / 2) steps, + 1 for the step that finds
// the clock negative, + 1 for a segment that leaves it negative and still takes its sample.  At 192 kHz a segment of 16 s
// lasts up to 22 % longer than its nominal 3 072 000 samples (ulp(16) / 2 = 0.18 dt) — in the reference as here.
static double segment_bound(float length, double dt)
{
    if (!(length > 0.0f)) return 2.0;
    int e = 0;
    (void)std::frexp(length, &e);                       // length = m 2^e, m in [0.5, 1): ulp = 2^(e - 24)
    const double half_ulp = std::ldexp(1.0, e - 25);
    if (!(dt > 2.0 * half_ulp)) return INFINITY;        // (the clock may not move at all)
    return (double)length / (dt - half_ulp) + 2.0;
}

// An upper bound of every utterance's length in samples, on the device (time-split kernels: a chunk's lane whose utterance
// ends before the chunk begins has nothing to render and says so at once — batches whose rows differ in length then take
// more, shorter chunks; launch_plan.cpp): grail_length_bound at the table's highest sample rate, + a tile.
int upload_len_bound(grail_ctx *ctx, grail_batch *b, const std::vector<RowStats> &rows, uint32_t n_utt)
{
    if (!b->plain || n_utt == 0 || ctx->voices.empty()) return GRAIL_OK;
    std::vector<uint32_t> bound(n_utt);
    for (uint32_t u = 0; u < n_utt; ++u) {
        const double samples = rows[u].bound_samples + 64.0;
        bound[u] = samples < 4.0e9 ? (uint32_t)samples : 0xFFFFFFFFu;
    }
    int rc = upload(&b->d_len_bound, bound.data(), n_utt, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    b->len_bound_epoch = ctx->voices_epoch;
    b->len_bound_known = true;
    return GRAIL_OK;
}

// Ragged batches: the lanes of a wave run in lockstep, so a wave lasts as long as its longest utterance.
// Launch slots are therefore filled in order of decreasing length (sum of the segment lengths, in seconds
// — close enough to the sample count for sorting): the utterances of a wave end together, and when the
// batch is larger than the machine the longest waves start first.  Results do not depend on the slot
// (batch invariance), rows stay where the caller put them.  Aligned batches (all sums equal) keep the
// identity assignment and pay nothing.
// Rows the lean kernel families cannot take (grail_batch::groups) go last, whatever their length: the batch is then
// planned as two batches.  (Judged against the voice table of this moment: the bounds involve its sample rates and
// pitch jitter; a batch uploaded before any table, or rendered with another one, is planned as one.)
int upload_length_order(grail_ctx *ctx, grail_batch *b, const std::vector<RowStats> &rows, uint32_t n_utt)
{
    if (n_utt < 2 || !ctx->sort_option) return GRAIL_OK;
    std::vector<uint8_t> outlier(n_utt, 0);
    uint32_t n_out = 0;
    if (!ctx->voices.empty()) {
        for (uint32_t u = 0; u < n_utt; ++u) {
            const RowStats &r = rows[u];
            const bool lean = r.plain && r.min_length >= 2.0f * ctx->max_dt &&
                              r.min_pitch * 0.999f - 1.002f * ctx->max_pitch_jitter >= 9.5367431640625e-07f;
            outlier[u] = lean ? 0 : 1;
            n_out += outlier[u];
        }
        if (n_out == n_utt) n_out = 0;                  // nothing to separate them from
    }
    float lo = rows[0].seconds, hi = rows[0].seconds;
    for (uint32_t u = 1; u < n_utt; ++u) {
        lo = std::fmin(lo, rows[u].seconds);
        hi = std::fmax(hi, rows[u].seconds);
    }
    const bool ragged = hi - lo > 0.002f;           // (aligned, or NaN lengths: nothing to gain from sorting)
    if (!ragged && n_out == 0) return GRAIL_OK;
    std::vector<uint32_t> perm(n_utt);
    for (uint32_t u = 0; u < n_utt; ++u) perm[u] = u;
    auto key = [&](uint32_t u) { return std::isfinite(rows[u].seconds) ? rows[u].seconds : 0.0f; };   // (a strict weak order)
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t c) {
        if (n_out && outlier[a] != outlier[c]) return outlier[a] < outlier[c];
        return ragged && key(a) > key(c);
    });
    int rc = upload(&b->d_perm, perm.data(), n_utt, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_out) {
        // two views of the batch: same device buffers, each with the summary of its own rows
        b->groups.assign(2, *b);
        for (int g = 0; g < 2; ++g) {
            grail_batch &v = b->groups[g];
            v.groups.clear();
            v.plan_cache = nullptr;
            v.n_utt = g ? n_out : n_utt - n_out;
            v.plain = true;
            v.any_blend = false;
            v.min_length = INFINITY;
            v.min_pitch = INFINITY;
            v.max_seconds = 0.0f;        // (used_voices: the batch's — a superset of the group's)
        }
        for (uint32_t u = 0; u < n_utt; ++u) {
            grail_batch &v = b->groups[outlier[u]];
            const RowStats &r = rows[u];
            v.plain = v.plain && r.plain;
            v.any_blend = v.any_blend || r.any_blend;
            if (r.min_length < v.min_length) v.min_length = r.min_length;
            if (r.min_pitch < v.min_pitch) v.min_pitch = r.min_pitch;
            if (r.seconds > v.max_seconds) v.max_seconds = r.seconds;
        }
        b->groups_epoch = ctx->voices_epoch;
    }
    // What ragged_plan() and ragged_cost() weigh the lane mappings with: per 8 launch slots the longest row and the rows'
    // segment and kink counts — of the whole batch in launch order, and, where a few rows the lean kernel families cannot take
    // were put last, of the OTHER rows by themselves too: the first view of the batch is the corpus, and one zero-length
    // segment among 65 536 speech-like utterances must not cost the rest their plan.  (A view is planned by its rows' lengths
    // only if they differ THEMSELVES: an aligned batch with a few odd rows is ragged only through those, and its first group
    // keeps the aligned plan — the model prices every lane's events as its own, which an aligned group's are not.)
    auto summarise = [&](grail_batch &t, const uint32_t n_rows) {
        const size_t n_gran = ((size_t)n_rows + 7) / 8;
        t.granule_samples.assign(n_gran, 0.0f);
        t.granule_segs.assign(n_gran, 0u);
        t.granule_kinks.assign(n_gran, 0u);
        for (uint32_t s = 0; s < n_rows; ++s) {
            const RowStats &r = rows[perm[s]];
            const float samples = key(perm[s]) * ctx->max_rate;
            if (samples > t.granule_samples[s / 8]) t.granule_samples[s / 8] = samples;
            t.granule_segs[s / 8] += r.segs;
            t.granule_kinks[s / 8] += r.kinks;
        }
    };
    if (ragged && n_out) {
        float llo = INFINITY, lhi = -INFINITY;
        for (uint32_t u = 0; u < n_utt; ++u)
            if (!outlier[u]) {
                llo = std::fmin(llo, rows[u].seconds);
                lhi = std::fmax(lhi, rows[u].seconds);
            }
        if (lhi - llo > 0.002f) {
            summarise(b->groups[0], n_utt - n_out);
            summarise(*b, n_utt);
        }
    } else if (ragged) {
        summarise(*b, n_utt);
    }
    b->perm_host.swap(perm);          // (after the views were copied off the batch: they share d_perm, only the root keeps this)
    return GRAIL_OK;
}

int upload_common(grail_ctx *ctx, grail_batch *b, const uint32_t *seg_offsets,
                  const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt)
{
    int rc;
    if ((rc = upload(&b->d_offsets, seg_offsets, (size_t)n_utt + 1, ctx->stream))) return rc;
    b->max_voice_id = 0;
    b->used_voices.assign(1, 0u);                     // no ids: voice 0 for all
    if (voice_ids) {
        for (uint32_t u = 0; u < n_utt; ++u)
            if (voice_ids[u] > b->max_voice_id) b->max_voice_id = voice_ids[u];
        if (n_utt) {
            b->used_voices.assign(voice_ids, voice_ids + n_utt);
            std::sort(b->used_voices.begin(), b->used_voices.end());
            b->used_voices.erase(std::unique(b->used_voices.begin(), b->used_voices.end()), b->used_voices.end());
        }
        if ((rc = upload(&b->d_voice_ids, voice_ids, n_utt, ctx->stream))) return rc;
    }
    if (jitter_seeds)
        if ((rc = upload(&b->d_seeds, jitter_seeds, n_utt, ctx->stream))) return rc;
    b->n_utt = n_utt;
    // host buffers may be freed by the caller right after we return
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GRAIL_OK;
}

int install_voices(grail_ctx *ctx, const grail_voice *voices, uint32_t n_voices)
{
    if (!voices || n_voices == 0) return fail(GRAIL_ERR_INVALID_ARG, "no voices given");
    std::vector<DevVoice> dv(n_voices);
    std::vector<float> elems((size_t)n_voices * NUM_VOICED * ELEM_FLOATS);
    for (uint32_t v = 0; v < n_voices; ++v) {
        dv[v].sample_rate = voices[v].sample_rate;
        dv[v].jitter_frequency = voices[v].jitter_frequency;
        dv[v].jitter_delta_frequency = voices[v].jitter_delta_frequency;
        dv[v].jitter_delta_formant_frequency = voices[v].jitter_delta_formant_frequency;
        dv[v].jitter_delta_amplitude = voices[v].jitter_delta_amplitude;
        dv[v].elem_base = v * NUM_VOICED;
        dv[v].warmup = voice_warmup(voices[v]);
        dv[v].pad = 0;
        std::memcpy(&elems[(size_t)v * NUM_VOICED * ELEM_FLOATS], voices[v].phonemes,
                    sizeof(grail_synthesis_elem) * NUM_VOICED);
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // kernels may still read the old table
    ctx->voices.clear();                          // a failure below leaves "no voice table set"
    if (ctx->d_voices) (void)hipFree(ctx->d_voices);
    if (ctx->d_voice_elems) (void)hipFree(ctx->d_voice_elems);
    ctx->d_voices = nullptr;
    ctx->d_voice_elems = nullptr;
    HIP_TRY(hipMalloc((void **)&ctx->d_voices, dv.size() * sizeof(DevVoice)));
    HIP_TRY(hipMalloc((void **)&ctx->d_voice_elems, elems.size() * sizeof(float)));
    // on the context's own (non-blocking) stream: ordered with the kernels that read the table
    HIP_TRY(hipMemcpyAsync(ctx->d_voices, dv.data(), dv.size() * sizeof(DevVoice), hipMemcpyHostToDevice,
                           ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->d_voice_elems, elems.data(), elems.size() * sizeof(float),
                           hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // dv / elems are locals
    ctx->voices.assign(voices, voices + n_voices);
    ctx->voices_epoch = ++g_epoch;
    bool silent = true;
    for (uint32_t v = 0; v < n_voices; ++v)
        for (int p = 0; p < NUM_VOICED; ++p)
            for (int i = NF / 2; i < NF; ++i) {
                uint32_t bits;
                std::memcpy(&bits, &voices[v].phonemes[p].formant_amp[i], sizeof bits);
                silent = silent && bits == 0u;
            }
    ctx->voices_upper_silent = silent;
    ctx->voices_live4_ok = true;
    ctx->voices_scan_ok = true;
    for (uint32_t v = 0; v < n_voices; ++v) ctx->voices_scan_ok = ctx->voices_scan_ok && scan_voice_ok(voices[v]);
    ctx->max_dt = 0.0f;
    ctx->max_pitch_jitter = 0.0f;
    ctx->max_rate = 0.0f;
    ctx->max_warmup = 0;
    ctx->voices_split_ok = true;
    ctx->voices_sharpness = 0.0;
    ctx->voice_sharpness.assign(n_voices, 0.0);
    for (uint32_t v = 0; v < n_voices; ++v) {
        ctx->voice_sharpness[v] = elems_sharpness(voices[v].phonemes, NUM_VOICED);
        ctx->voices_sharpness = std::fmax(ctx->voices_sharpness, ctx->voice_sharpness[v]);
    }
    ctx->voice_info.assign(n_voices, grail_ctx::VoiceInfo());
    for (uint32_t v = 0; v < n_voices; ++v) {
        grail_ctx::VoiceInfo &vi = ctx->voice_info[v];
        vi.upper_silent = true;
        for (int p = 0; p < NUM_VOICED; ++p)
            for (int i = NF / 2; i < NF; ++i) {
                uint32_t bits;
                std::memcpy(&bits, &voices[v].phonemes[p].formant_amp[i], sizeof bits);
                vi.upper_silent = vi.upper_silent && bits == 0u;
            }
        vi.live4_ok = live4_ok(voices[v]);
        vi.scan_ok = scan_voice_ok(voices[v]);
        vi.warmup = dv[v].warmup;
        vi.split_ok = dv[v].warmup != 0u && voices[v].sample_rate > 0.0f && std::isfinite(voices[v].sample_rate);
        ctx->voices_live4_ok = ctx->voices_live4_ok && vi.live4_ok;
        ctx->voices_split_ok = ctx->voices_split_ok && vi.split_ok;
        if (dv[v].warmup > ctx->max_warmup) ctx->max_warmup = dv[v].warmup;
        if (voices[v].sample_rate > ctx->max_rate) ctx->max_rate = voices[v].sample_rate;
        const float dt = 1.0f / voices[v].sample_rate;
        if (!(dt <= ctx->max_dt)) ctx->max_dt = dt;       // NaN-proof max
        const float pj = std::fabs(voices[v].jitter_delta_frequency);
        if (!(pj <= ctx->max_pitch_jitter)) ctx->max_pitch_jitter = pj;
    }
    return GRAIL_OK;
}

int check_ready(grail_ctx *ctx, const grail_batch *batch)
{
    if (!batch) return fail(GRAIL_ERR_INVALID_ARG, "batch is NULL");
    if (ctx->voices.empty() || !ctx->d_voices || !ctx->d_voice_elems)   // also after a failed upload
        return fail(GRAIL_ERR_NO_VOICES, "call grail_set_voices first");
    if (batch->max_voice_id >= ctx->voices.size())
        return fail(GRAIL_ERR_INVALID_ARG, "a voice id exceeds the voice table");
    return GRAIL_OK;
}

}  // namespace host
}  // namespace grail

extern "C" {

int grail_abi_version(void) { return GRAIL_ABI_VERSION; }

const char *grail_status_string(int status)
{
    switch (status) {
    case GRAIL_OK: return "ok";
    case GRAIL_ERR_INVALID_ARG: return "invalid argument";
    case GRAIL_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case GRAIL_ERR_HIP: return "HIP runtime error";
    case GRAIL_ERR_BUFFER_TOO_SMALL: return "an utterance did not end within out_stride samples";
    case GRAIL_ERR_OUT_OF_MEMORY: return "out of device memory";
    case GRAIL_ERR_RCCL: return "RCCL error";
    case GRAIL_ERR_NO_VOICES: return "no voice table set";
    default: return "unknown status";
    }
}

const char *grail_last_error(void) { return g_last_error.c_str(); }

int grail_device_count(int *count)
{
    if (!count) return fail(GRAIL_ERR_INVALID_ARG, "count is NULL");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
    *count = n;
    return GRAIL_OK;
}

int grail_device_pci_bus_id(grail_ctx *ctx, char *out, size_t cap)
{
    if (!ctx || !out || cap < 16) return fail(GRAIL_ERR_INVALID_ARG, "grail_device_pci_bus_id: NULL argument or cap < 16");
    out[0] = 0;
    HIP_TRY(hipDeviceGetPCIBusId(out, (int)cap, ctx->device));
    return GRAIL_OK;
}

int grail_create(int device, grail_ctx **out)
{
    if (!out) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(GRAIL_ERR_NO_DEVICE,
                    std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count 0"));
    if (device < 0 || device >= n) return fail(GRAIL_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    // the kernels in this library are gfx950 code objects only, and the launch policy is laid out for CDNA4's
    // 4 SIMDs x 64 lanes per compute unit: any other architecture is "no usable device"
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GRAIL_ERR_NO_DEVICE, std::string("device ") + std::to_string(device) + " is " + prop.gcnArchName +
                                             ": this library holds gfx950 (MI355X) kernels only");
    if (prop.multiProcessorCount <= 0) return fail(GRAIL_ERR_NO_DEVICE, "the device reports no compute units");
    grail_ctx *ctx = new (std::nothrow) grail_ctx();
    if (!ctx) return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
    ctx->device = device;
    ctx->cus = ctx->device_cus = prop.multiProcessorCount;
    hipError_t err;
    if ((err = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (err = hipEventCreate(&ctx->ev_start)) != hipSuccess ||
        (err = hipEventCreate(&ctx->ev_stop)) != hipSuccess ||
        (err = hipMalloc((void **)&ctx->d_truncated, TRUNCATED_WORDS * sizeof(uint32_t))) != hipSuccess ||
        (err = hipMemsetAsync(ctx->d_truncated, 0, TRUNCATED_WORDS * sizeof(uint32_t), ctx->stream)) != hipSuccess ||
        (err = hipStreamSynchronize(ctx->stream)) != hipSuccess) {
        grail_destroy(ctx);
        return hip_fail(err, "grail_create");
    }
    *out = ctx;
    return GRAIL_OK;
}

int grail_destroy(grail_ctx *ctx)
{
    if (!ctx) return GRAIL_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    pipe_destroy_opaque(ctx->host_pipe);
    if (ctx->d_voices) (void)hipFree(ctx->d_voices);
    if (ctx->d_voice_elems) (void)hipFree(ctx->d_voice_elems);
    if (ctx->d_truncated) (void)hipFree(ctx->d_truncated);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GRAIL_OK;
}

int grail_set_voices(grail_ctx *ctx, const grail_voice *voices, uint32_t n_voices)
{
    int rc = bind(ctx);
    if (rc) return rc;
    return install_voices(ctx, voices, n_voices);
}

int grail_get_voices(grail_ctx *ctx, grail_voice *voices, uint32_t cap, uint32_t *n_voices)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    if (n_voices) *n_voices = (uint32_t)ctx->voices.size();
    if (voices)
        for (uint32_t i = 0; i < cap && i < ctx->voices.size(); ++i) voices[i] = ctx->voices[i];
    return GRAIL_OK;
}

uint64_t grail_length_bound(const float *segment_lengths, uint32_t n_segments, float sample_rate)
{
    if (!(sample_rate > 0.0f) || (n_segments && !segment_lengths)) return UINT64_MAX;
    const double dt = (double)(1.0f / sample_rate);
    double samples = 0.0;
    for (uint32_t i = 0; i < n_segments; ++i) {
        if (!std::isfinite(segment_lengths[i])) return UINT64_MAX;
        samples += segment_bound(segment_lengths[i], dt);
    }
    return samples < 1.8e19 ? (uint64_t)samples : UINT64_MAX;
}

int grail_set_option(grail_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return fail(GRAIL_ERR_INVALID_ARG, "NULL argument");
    ++ctx->options_epoch;
    if (std::strcmp(name, "lanes_per_utterance") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
            return fail(GRAIL_ERR_INVALID_ARG, "lanes_per_utterance must be 0, 1, 2, 4 or 8");
        ctx->lanes_option = (int)value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "skip_silent_formants") == 0) {
        ctx->skip_silent_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "small_batch_pipeline") == 0) {
        ctx->pipeline_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "arithmetic") == 0) {
        if (value != 0 && value != 1 && value != 2)
            return fail(GRAIL_ERR_INVALID_ARG, "arithmetic must be 0 (exact), 1 (fast) or 2 (fast, exact coefficients)");
        ctx->fast_option = (int)value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "fast_exact_coefficients") == 0) {
        ctx->mid_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "fast_sharpness_limit_exact_coefficients") == 0) {
        if (value < 0) return fail(GRAIL_ERR_INVALID_ARG, "negative limit");
        ctx->mid_limit = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_parallel_scan") == 0) {
        ctx->scan_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "packed_launch_order") == 0) {
        ctx->packed_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "sort_by_length") == 0) {    // applies to batches uploaded afterwards
        ctx->sort_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split") == 0) {
        ctx->split_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split_chunks") == 0) {
        if (value < 0 || value > SPLIT_MAX_CHUNKS) return fail(GRAIL_ERR_INVALID_ARG, "time_split_chunks must be 0 (auto) .. 64");
        ctx->split_chunks = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split_span_samples") == 0) {
        if (value < 0 || value > 0x7fffffff) return fail(GRAIL_ERR_INVALID_ARG, "time_split_span_samples out of range");
        ctx->split_span = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split_ff_cost_permille") == 0) {
        if (value < 0 || value > 1000) return fail(GRAIL_ERR_INVALID_ARG, "time_split_ff_cost_permille must be 0 .. 1000");
        ctx->split_ff_permille = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split_min_utterances") == 0) {   // -1: the cost model decides
        if (value < -1) return fail(GRAIL_ERR_INVALID_ARG, "negative limit");
        ctx->split_min_utts = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "fast_sharpness_limit") == 0) {
        if (value < 0) return fail(GRAIL_ERR_INVALID_ARG, "negative limit");
        ctx->fast_limit = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline4_max_groups") == 0) {   // tuning: four-formant batches, 16 utterances per workgroup
        ctx->pipe4_max_groups = value;                       // (-1: two per compute unit)
        return GRAIL_OK;
    }
    if (std::strcmp(name, "assume_compute_units") == 0) {   // plan for so many compute units (0: what the device reports)
        if (value < 0 || value > 4096) return fail(GRAIL_ERR_INVALID_ARG, "assume_compute_units must be 0 (the device's) .. 4096");
        ctx->cus = value ? (int)value : ctx->device_cus;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "composite_launches") == 0) {
        ctx->composite_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "ragged_plan") == 0) {
        ctx->ragged_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "two_waves_per_simd") == 0) {
        ctx->two_waves_option = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline_spread") == 0) {
        ctx->pipe_spread = value ? 1 : 0;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "row_groups") == 0) {
        if (value < 0 || value > 2) return fail(GRAIL_ERR_INVALID_ARG, "row_groups must be 0 (off), 1 (by cost) or 2 (always)");
        ctx->row_groups_option = (int)value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline8_max_groups") == 0) {   // tuning: 0 keeps eight-formant batches off the pipeline
        ctx->pipe8_max_groups = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline_round32") == 0) {       // tuning (A/B)
        if (value < 0 || value > 2) return fail(GRAIL_ERR_INVALID_ARG, "pipeline_round32 must be 0 (never), 1 (aligned batches) or 2 (any batch)");
        ctx->pipe_round32 = (int)value;
        return GRAIL_OK;
    }
#ifdef GRAIL_SCAN_DEBUG
    if (std::strcmp(name, "scan_debug") == 0) {       // development builds only (-DGRAIL_SCAN_DEBUG): see scan_kernels.hip
        ctx->scan_debug = (int)value;
        return GRAIL_OK;
    }
#endif
    if (std::strcmp(name, "time_parallel_scan_split_max_utterances") == 0) {   // -1: 6 per compute unit
        if (value < -1) return fail(GRAIL_ERR_INVALID_ARG, "negative limit");
        ctx->scan_split_max = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_parallel_scan_max_utterances") == 0) {         // -1: 34 per compute unit
        if (value < -1) return fail(GRAIL_ERR_INVALID_ARG, "negative limit");
        ctx->scan_max_utts = value;
        return GRAIL_OK;
    }
    return fail(GRAIL_ERR_INVALID_ARG, std::string("unknown option ") + name);
}

int grail_get_option(grail_ctx *ctx, const char *name, int64_t *value)
{
    if (!ctx || !name || !value) return fail(GRAIL_ERR_INVALID_ARG, "NULL argument");
    if (std::strcmp(name, "lanes_per_utterance") == 0) {
        *value = ctx->lanes_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "skip_silent_formants") == 0) {
        *value = ctx->skip_silent_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "arithmetic") == 0) {
        *value = ctx->fast_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_parallel_scan") == 0) {
        *value = ctx->scan_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_parallel_scan_max_utterances") == 0) {   // as set (-1: 34 per compute unit), so that
        *value = ctx->scan_max_utts;                                       // get / set restores exactly
        return GRAIL_OK;
    }
    if (std::strcmp(name, "compute_units") == 0) {             // read-only: what the launch policy plans for
        *value = ctx->cus;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "assume_compute_units") == 0) {      // 0: the device's own count is in force
        *value = ctx->cus == ctx->device_cus ? 0 : ctx->cus;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "composite_launches") == 0) {
        *value = ctx->composite_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "row_groups") == 0) {
        *value = ctx->row_groups_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "ragged_plan") == 0) {
        *value = ctx->ragged_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "two_waves_per_simd") == 0) {
        *value = ctx->two_waves_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline_spread") == 0) {
        *value = ctx->pipe_spread;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline4_max_groups") == 0) {
        *value = ctx->pipe4_max_groups;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline8_max_groups") == 0) {
        *value = ctx->pipe8_max_groups;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline_round32") == 0) {
        *value = ctx->pipe_round32;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "packed_launch_order") == 0) {
        *value = ctx->packed_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "last_launch_packed") == 0) {
        *value = ctx->last_packed;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "sort_by_length") == 0) {
        *value = ctx->sort_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "last_launch_fast") == 0) {          // read-only: some block of the last launch ran tolerance arithmetic
        *value = ctx->last_fast;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "last_launch_blocks") == 0) {        // read-only: kernel launches the last synthesis call was cut into
        *value = ctx->last_blocks;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split") == 0) {
        *value = ctx->split_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split_chunks") == 0) {
        *value = ctx->split_chunks;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split_span_samples") == 0) {
        *value = ctx->split_span;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split_ff_cost_permille") == 0) {
        *value = ctx->split_ff_permille;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_split_min_utterances") == 0) {
        *value = ctx->split_min_utts;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "fast_sharpness_limit") == 0) {
        *value = ctx->fast_limit;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "fast_arithmetic_served") == 0) {    // read-only: the tier "arithmetic" = 1 gets for the voice table
        *value = fast_tier_for(ctx, nullptr, 1);                // as a whole: 1 interpolating, 2 exact coefficients, 0 exact kernels
        return GRAIL_OK;
    }
    if (std::strcmp(name, "fast_exact_coefficients") == 0) {
        *value = ctx->mid_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "fast_sharpness_limit_exact_coefficients") == 0) {
        *value = ctx->mid_limit;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "last_launch_chunks") == 0) {        // read-only: chunks per utterance (0: not time-split)
        *value = ctx->last_split;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "time_parallel_scan_split_max_utterances") == 0) {
        *value = ctx->scan_split_max;
        return GRAIL_OK;
    }
#ifdef GRAIL_FAST_PROF
    if (std::strncmp(name, "debug_prof_", 11) == 0) {          // debug builds: counter k of the tolerance-mode tile loop
        const int k = std::atoi(name + 11);
        if (k < 0 || k >= 32) return fail(GRAIL_ERR_INVALID_ARG, "debug_prof_<k>: k in 0 .. 31");
        unsigned long long v = 0;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(&v, reinterpret_cast<unsigned long long *>(ctx->d_truncated + 8) + k, sizeof v, hipMemcpyDeviceToHost));
        *value = (int64_t)v;
        return GRAIL_OK;
    }
#endif
    if (std::strcmp(name, "slow_division_wave_steps") == 0) {  // read-only statistic
        *value = (int64_t)ctx->slow_steps;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "fast_wave_tiles") == 0) {           // read-only: wave-tiles rendered in fast arithmetic
        *value = (int64_t)ctx->fast_tiles;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "general_wave_steps") == 0) {        // read-only: wave-steps through the general step
        *value = (int64_t)ctx->general_steps;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "last_launch_formants") == 0) {      // read-only: 4 or 8 laid out over the lanes
        *value = ctx->last_formants;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "last_launch_lanes") == 0) {         // read-only: lanes per utterance chosen
        *value = ctx->last_lanes;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "small_batch_pipeline") == 0) {
        *value = ctx->pipeline_option;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "last_launch_pipelined") == 0) {     // read-only
        *value = ctx->last_pipe;
        return GRAIL_OK;
    }
    return fail(GRAIL_ERR_INVALID_ARG, std::string("unknown option ") + name);
}

int grail_batch_upload(grail_ctx *ctx, const grail_phoneme_elem *segs, const uint32_t *seg_offsets,
                       const uint32_t *voice_ids, const uint32_t *jitter_seeds, uint32_t n_utt,
                       grail_batch **out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    uint32_t n_segs = 0;
    if ((rc = check_offsets(seg_offsets, n_utt, &n_segs))) return rc;
    if (n_segs && !segs) return fail(GRAIL_ERR_INVALID_ARG, "segs is NULL");
    for (uint32_t i = 0; i < n_segs; ++i)
        if (segs[i].phoneme < 0 || segs[i].phoneme >= GRAIL_PH_COUNT)
            return fail(GRAIL_ERR_INVALID_ARG, "phoneme discriminant out of range");
    // per utterance, then over the batch: what the launch policy asks of the segments
    std::vector<RowStats> rows(n_utt);
    // (the device's dt of the table's highest sample rate, as the kernels have it: an f32 reciprocal)
    const double min_dt = ctx->max_rate > 0.0f ? (double)(1.0f / ctx->max_rate) : 0.0;
    for (uint32_t u = 0; u < n_utt; ++u) {
        RowStats &r = rows[u];
        for (uint32_t i = seg_offsets[u]; i < seg_offsets[u + 1]; ++i) {
            r.plain = r.plain && std::isfinite(segs[i].length) && std::isfinite(segs[i].blend_length) &&
                      std::isfinite(segs[i].frequency) && segs[i].blend_length > 0.0f;
            r.any_blend = r.any_blend || !blend_is_pow2(segs[i].blend_length);
            if (segs[i].length < r.min_length) r.min_length = segs[i].length;
            const float pitch = std::fmin(segs[i].frequency, 0.5f);   // copy_with_frequency :445-450
            if (pitch < r.min_pitch) r.min_pitch = pitch;
            r.seconds += segs[i].length;
            r.bound_samples += segment_bound(segs[i].length, min_dt);
            r.segs += 1u;
            r.kinks += segs[i].blend_length < segs[i].length ? 1u : 0u;
        }
    }
    grail_batch *b = new (std::nothrow) grail_batch();
    if (!b) return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->phoneme_mode = true;
    b->plain = true;
    b->min_length = INFINITY;
    b->min_pitch = INFINITY;
    for (const RowStats &r : rows) {
        b->plain = b->plain && r.plain;
        b->any_blend = b->any_blend || r.any_blend;
        if (r.min_length < b->min_length) b->min_length = r.min_length;
        if (r.min_pitch < b->min_pitch) b->min_pitch = r.min_pitch;
        if (r.seconds > b->max_seconds) b->max_seconds = r.seconds;
    }
    b->n_segs = n_segs;
    if ((rc = upload(&b->d_segs, segs, n_segs, ctx->stream)) ||
        (rc = upload_common(ctx, b, seg_offsets, voice_ids, jitter_seeds, n_utt)) ||
        (rc = upload_len_bound(ctx, b, rows, n_utt)) || (rc = upload_length_order(ctx, b, rows, n_utt))) {
        free_batch_buffers(b);
        delete b;
        return rc;
    }
    *out = b;
    return GRAIL_OK;
}

int grail_batch_upload_elems(grail_ctx *ctx, const grail_sequence_elem *segs,
                             const uint32_t *seg_offsets, const uint32_t *voice_ids,
                             const uint32_t *jitter_seeds, uint32_t n_utt, grail_batch **out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    uint32_t n_segs = 0;
    if ((rc = check_offsets(seg_offsets, n_utt, &n_segs))) return rc;
    if (n_segs && !segs) return fail(GRAIL_ERR_INVALID_ARG, "segs is NULL");
    // Split the SequenceElems into the 16-B segment records and the elem table.
    std::vector<DevSeg> ds(n_segs);
    std::vector<float> elems((size_t)(n_segs ? n_segs : 1) * ELEM_FLOATS);
    for (uint32_t i = 0; i < n_segs; ++i) {
        ds[i].elem = segs[i].has_elem ? (int32_t)i : -1;
        ds[i].length = segs[i].length;
        ds[i].blend_length = segs[i].blend_length;
        ds[i].frequency = segs[i].elem.frequency;
        std::memcpy(&elems[(size_t)i * ELEM_FLOATS], &segs[i].elem, sizeof(grail_synthesis_elem));
    }
    // (what the launch policy asks of the segments, per utterance and over the batch, as for phoneme batches; a
    // caller-built elem keeps its frequency as it is, copy_with_frequency's min(f, 0.5) :445-450 belongs to the Selector)
    std::vector<RowStats> rows(n_utt);
    // (the device's dt of the table's highest sample rate, as the kernels have it: an f32 reciprocal)
    const double min_dt = ctx->max_rate > 0.0f ? (double)(1.0f / ctx->max_rate) : 0.0;
    for (uint32_t u = 0; u < n_utt; ++u) {
        RowStats &r = rows[u];
        for (uint32_t i = seg_offsets[u]; i < seg_offsets[u + 1]; ++i) {
            r.plain = r.plain && std::isfinite(segs[i].length) && std::isfinite(segs[i].blend_length) &&
                      std::isfinite(segs[i].elem.frequency) && segs[i].blend_length > 0.0f;
            r.any_blend = r.any_blend || !blend_is_pow2(segs[i].blend_length);
            if (segs[i].length < r.min_length) r.min_length = segs[i].length;
            if (segs[i].elem.frequency < r.min_pitch) r.min_pitch = segs[i].elem.frequency;
            r.seconds += segs[i].length;
            r.bound_samples += segment_bound(segs[i].length, min_dt);
            r.segs += 1u;
            r.kinks += segs[i].blend_length < segs[i].length ? 1u : 0u;
        }
    }
    grail_batch *b = new (std::nothrow) grail_batch();
    if (!b) return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->phoneme_mode = false;
    b->plain = true;
    b->min_length = INFINITY;
    b->min_pitch = INFINITY;
    for (const RowStats &r : rows) {
        b->plain = b->plain && r.plain;
        b->any_blend = b->any_blend || r.any_blend;
        if (r.min_length < b->min_length) b->min_length = r.min_length;
        if (r.min_pitch < b->min_pitch) b->min_pitch = r.min_pitch;
        if (r.seconds > b->max_seconds) b->max_seconds = r.seconds;
    }
    b->n_segs = n_segs;
    // The elems of the batch, interned by their formant arrays (everything but the pitch, which none of the analyses
    // below looks at): a corpus names a few dozen distinct parameter sets in hundreds of thousands of segments.
    struct FormantsHash {
        size_t operator()(const float *p) const
        {
            uint64_t h = 0x9E3779B97F4A7C15ull, w;
            for (int i = 0; i < 24; ++i) {                   // 48 floats after the frequency
                std::memcpy(&w, p + 1 + 2 * i, sizeof w);
                h = (h ^ w) * 0xFF51AFD7ED558CCDull;
                h ^= h >> 29;
            }
            return (size_t)h;
        }
    };
    struct FormantsEq {
        bool operator()(const float *a, const float *c) const { return std::memcmp(a + 1, c + 1, 48 * sizeof(float)) == 0; }
    };
    std::unordered_map<const float *, uint32_t, FormantsHash, FormantsEq> ids;
    std::vector<grail_synthesis_elem> distinct;
    std::vector<uint32_t> id_of(n_segs, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < n_segs; ++i) {
        if (!segs[i].has_elem) continue;
        const auto it = ids.emplace((const float *)&segs[i].elem, (uint32_t)distinct.size());
        if (it.second) distinct.push_back(segs[i].elem);
        id_of[i] = it.first->second;
    }
    // the sharpness of the batch (elems_sharpness): parameters only ever blend between the elems of two consecutive
    // segments of an utterance (Sequencer::next :897-921), so every such pair is judged like a voice of two phonemes
    // (once per distinct pair)
    std::unordered_map<uint64_t, double> pair_sharpness;
    for (uint32_t u = 0; u < n_utt; ++u)
        for (uint32_t i = seg_offsets[u]; i < seg_offsets[u + 1]; ++i) {
            if (!segs[i].has_elem) continue;
            const uint32_t a = id_of[i];
            const uint32_t c = (i + 1 < seg_offsets[u + 1] && segs[i + 1].has_elem) ? id_of[i + 1] : a;
            const uint64_t key = ((uint64_t)std::min(a, c) << 32) | std::max(a, c);
            auto it = pair_sharpness.find(key);
            if (it == pair_sharpness.end()) {
                const grail_synthesis_elem pair[2] = {distinct[a], distinct[c]};
                it = pair_sharpness.emplace(key, elems_sharpness(pair, a == c ? 1 : 2)).first;
            }
            b->elems_sharpness = std::fmax(b->elems_sharpness, it->second);
        }
    if ((rc = upload(&b->d_segs, ds.data(), n_segs, ctx->stream)) ||
        (rc = upload(&b->d_elems, elems.data(), elems.size(), ctx->stream)) ||
        (rc = upload_common(ctx, b, seg_offsets, voice_ids, jitter_seeds, n_utt)) ||
        (rc = upload_len_bound(ctx, b, rows, n_utt)) || (rc = upload_length_order(ctx, b, rows, n_utt))) {
        free_batch_buffers(b);
        delete b;
        return rc;
    }
    // the warm-up length of the time-split fast kernels for THESE elems (elems_warmup: the slowest filter over the distinct
    // elems of the batch, the formant-frequency jitter of the voices it names), valid for the voice table of this moment;
    // 0: the batch does not qualify and fast arithmetic renders it with the lane kernels
    if (!ctx->voices.empty() && b->max_voice_id < ctx->voices.size()) {
        double jd = 0.0;
        bool rates_ok = true;
        for (const uint32_t v : b->used_voices) {
            jd = std::fmax(jd, std::fabs((double)ctx->voices[v].jitter_delta_formant_frequency));
            rates_ok = rates_ok && ctx->voices[v].sample_rate > 0.0f && std::isfinite(ctx->voices[v].sample_rate);
        }
        // can formants 5-8 be left out (the four-formant kernels)?  As for a voice table (live4_ok): the scalars of the
        // voices named, and every distinct elem of the batch
        bool voices4 = true;
        for (const uint32_t v : b->used_voices) {
            grail_voice scalars = ctx->voices[v];
            for (int ph = 0; ph < NUM_VOICED; ++ph) grail_elem_silent(&scalars.phonemes[ph]);     // (only the scalars count here)
            voices4 = voices4 && live4_ok(scalars);
        }
        b->elems_live4_ok = voices4 && live4_elems_ok(distinct.data(), distinct.size(), (float)jd);
        b->elems_warmup = rates_ok ? elems_warmup(distinct.data(), distinct.size(), jd) : 0u;
        // ... and whether the scan kernel may take them (its window has no IEEE fallback; pitches as the Selector
        // would have left them: at most 1/2)
        float max_pitch = 0.0f;
        for (uint32_t i = 0; i < n_segs; ++i) max_pitch = std::fmax(max_pitch, segs[i].elem.frequency);
        b->elems_scan_ok = rates_ok && max_pitch <= 0.5f && scan_elems_ok(distinct.data(), distinct.size(), (float)jd);
        b->elems_warmup_epoch = ctx->voices_epoch;
        for (grail_batch &g : b->groups) {          // (the views were made before these were known)
            g.elems_warmup = b->elems_warmup;
            g.elems_warmup_epoch = b->elems_warmup_epoch;
            g.elems_live4_ok = b->elems_live4_ok;
            g.elems_scan_ok = b->elems_scan_ok;
        }
    }
    *out = b;
    return GRAIL_OK;
}

int grail_batch_free(grail_ctx *ctx, grail_batch *batch)
{
    if (!batch) return GRAIL_OK;
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_batch_buffers(batch);
    delete batch;
    return GRAIL_OK;
}

uint32_t grail_batch_size(const grail_batch *batch) { return batch ? batch->n_utt : 0; }

int grail_batch_lengths(grail_ctx *ctx, const grail_batch *batch, uint32_t max_len,
                        uint32_t *out_len)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, batch))) return rc;
    if (!out_len && batch->n_utt) return fail(GRAIL_ERR_INVALID_ARG, "out_len is NULL");
    if (batch->n_utt == 0) return GRAIL_OK;
    uint32_t *d_len = nullptr;
    HIP_TRY(hipMalloc((void **)&d_len, (size_t)batch->n_utt * sizeof(uint32_t)));
    LenArgs a{};
    a.segs = batch->d_segs;
    a.seg_offsets = batch->d_offsets;
    a.voice_ids = batch->d_voice_ids;
    a.voices = ctx->d_voices;
    a.out_len = d_len;
    a.n_utt = batch->n_utt;
    a.n_voices = (uint32_t)ctx->voices.size();
    a.max_len = max_len;
    hipError_t e = launch_lengths(a, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(out_len, d_len, (size_t)batch->n_utt * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_len);
    if (e != hipSuccess) return hip_fail(e, "grail_batch_lengths");
    return GRAIL_OK;
}

int grail_sync(grail_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc) return rc;
    // flag read-back and reset travel on the stream the kernels run on (a non-blocking stream has
    // no implicit ordering with the null stream)
    uint32_t flags[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(flags, ctx->d_truncated, sizeof flags, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint32_t flag = flags[0];
    // the statistics counters are cumulative on the device (u32, wrapping): the host takes differences, so the usual
    // call costs one copy and one synchronisation; only a truncation flag has to be cleared
    ctx->slow_steps += (uint32_t)(flags[1] - ctx->seen_counters[1]);
    ctx->fast_tiles += (uint32_t)(flags[2] - ctx->seen_counters[2]);
    ctx->general_steps += (uint32_t)(flags[3] - ctx->seen_counters[3]);
    for (int i = 1; i < 4; ++i) ctx->seen_counters[i] = flags[i];
    if (flag) {
        HIP_TRY(hipMemsetAsync(ctx->d_truncated, 0, sizeof(uint32_t), ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (flag) {
        return fail(GRAIL_ERR_BUFFER_TOO_SMALL,
                    "at least one utterance did not end within out_stride samples");
    }
    return GRAIL_OK;
}

const char *grail_last_kernel_name(grail_ctx *ctx) { return ctx ? ctx->last_kernel.c_str() : "none"; }

int grail_last_kernel_ms(grail_ctx *ctx, float *ms)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ms) return fail(GRAIL_ERR_INVALID_ARG, "ms is NULL");
    if (!ctx->have_timing) return fail(GRAIL_ERR_INVALID_ARG, "no kernel has been launched");
    HIP_TRY(hipEventSynchronize(ctx->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
    return GRAIL_OK;
}

int grail_pcm16_async(grail_ctx *ctx, const float *in_dev, uint64_t in_stride,
                      const uint32_t *len_dev, uint32_t n_utt, uint32_t max_len, int16_t *out_dev,
                      uint64_t out_stride)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (n_utt && (!in_dev || !len_dev || !out_dev)) return fail(GRAIL_ERR_INVALID_ARG, "NULL buffer");
    hipError_t e = launch_pcm16(in_dev, in_stride, len_dev, n_utt, max_len, out_dev, out_stride,
                                ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "pcm16 kernel launch");
    return GRAIL_OK;
}

int grail_batch_digest(grail_ctx *ctx, const float *in_dev, uint64_t in_stride,
                       const uint32_t *len_dev, uint32_t n_utt, uint64_t *sums, float *maxabs,
                       uint32_t *nonfinite)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (n_utt == 0) return GRAIL_OK;
    if (!in_dev || !len_dev || !sums || !maxabs || !nonfinite)
        return fail(GRAIL_ERR_INVALID_ARG, "NULL buffer");
    unsigned long long *d_s = nullptr;
    float *d_m = nullptr;
    uint32_t *d_b = nullptr;
    hipError_t e = hipMalloc((void **)&d_s, (size_t)n_utt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_m, (size_t)n_utt * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_b, (size_t)n_utt * 4);
    if (e == hipSuccess) e = launch_digest(in_dev, in_stride, len_dev, n_utt, d_s, d_m, d_b, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sums, d_s, (size_t)n_utt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(maxabs, d_m, (size_t)n_utt * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(nonfinite, d_b, (size_t)n_utt * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_s) (void)hipFree(d_s);
    if (d_m) (void)hipFree(d_m);
    if (d_b) (void)hipFree(d_b);
    if (e != hipSuccess) return hip_fail(e, "grail_batch_digest");
    return GRAIL_OK;
}

int grail_batch_compare(grail_ctx *ctx, const float *a_dev, const float *b_dev, uint64_t stride,
                        const uint32_t *len_a_dev, const uint32_t *len_b_dev, uint32_t n_utt, float *maxdiff,
                        double *sumsq, uint32_t *mismatches)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (n_utt == 0) return GRAIL_OK;
    if (!a_dev || !b_dev || !len_a_dev || !len_b_dev || !maxdiff || !sumsq || !mismatches)
        return fail(GRAIL_ERR_INVALID_ARG, "NULL buffer");
    float *d_m = nullptr;
    double *d_q = nullptr;
    uint32_t *d_b = nullptr;
    hipError_t e = hipMalloc((void **)&d_m, (size_t)n_utt * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_q, (size_t)n_utt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_b, (size_t)n_utt * 4);
    if (e == hipSuccess)
        e = launch_compare(a_dev, b_dev, stride, len_a_dev, len_b_dev, n_utt, d_m, d_q, d_b, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(maxdiff, d_m, (size_t)n_utt * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sumsq, d_q, (size_t)n_utt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches, d_b, (size_t)n_utt * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_m) (void)hipFree(d_m);
    if (d_q) (void)hipFree(d_q);
    if (d_b) (void)hipFree(d_b);
    if (e != hipSuccess) return hip_fail(e, "grail_batch_compare");
    return GRAIL_OK;
}

int grail_device_alloc(grail_ctx *ctx, size_t bytes, void **out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return GRAIL_OK;
}

int grail_device_free(grail_ctx *ctx, void *ptr)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (ptr) HIP_TRY(hipFree(ptr));
    return GRAIL_OK;
}

int grail_host_alloc(grail_ctx *ctx, size_t bytes, void **out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out) return fail(GRAIL_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return GRAIL_OK;
}

int grail_host_free(grail_ctx *ctx, void *ptr)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return GRAIL_OK;
}

int grail_memcpy_d2h(grail_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GRAIL_OK;
}

int grail_memcpy_h2d(grail_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GRAIL_OK;
}

int grail_memset_d(grail_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GRAIL_OK;
}

}  // extern "C"
