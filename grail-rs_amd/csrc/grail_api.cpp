// grail_api.cpp — the C ABI (include/grail_hip.h) over the HIP kernels.
// Host orchestration only: contexts, HBM-resident batches, launches, RCCL
// broadcast of the voice table.  No arithmetic of the hot path happens here and
// there is no CPU fallback: without a HIP device every compute call fails.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/grail_hip.h"
#include "kernels.h"

using namespace grail;

static_assert(sizeof(grail_synthesis_elem) == 196, "SynthesisElem is 49 x f32");
static_assert(sizeof(grail_phoneme_elem) == sizeof(DevSeg), "PhonemeElem uploads as-is");
static_assert(offsetof(grail_phoneme_elem, phoneme) == offsetof(DevSeg, elem), "layout");
static_assert(offsetof(grail_phoneme_elem, frequency) == offsetof(DevSeg, frequency), "layout");
static_assert(sizeof(grail_voice) == 4 + 2 * 196 + 5 * 4, "Voice layout");
static_assert(GRAIL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {

thread_local std::string g_last_error;

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

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

// RCCL is loaded on first use so the library loads (and its symbols can be
// checked) on hosts without a GPU stack that can initialise RCCL.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        // The process environment is the host application's: nothing is set here.  A single-node
        // launcher that wants RCCL to skip the InfiniBand / interface probing (up to 2 minutes on a
        // box without a network) exports NCCL_IB_DISABLE=1 NCCL_SOCKET_IFNAME=lo itself, as bench.py
        // and the tests do.
        // The ROCm installation's RCCL by absolute path first: a bare "librccl.so" would be
        // satisfied by any copy the host process already holds (PyTorch wheels bundle one that
        // is bound to their own private HIP runtime, not to the one this library links).
        const char *env = getenv("GRAIL_RCCL_PATH");
        if (env && *env) x.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) x.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) x.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) x.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) return x;
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.handle, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.handle, "ncclCommInitRank");
        x.Broadcast = (decltype(x.Broadcast))dlsym(x.handle, "ncclBroadcast");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.handle, "ncclCommDestroy");
        x.CommCount = (decltype(x.CommCount))dlsym(x.handle, "ncclCommCount");
        x.CommUserRank = (decltype(x.CommUserRank))dlsym(x.handle, "ncclCommUserRank");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.handle, "ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.Broadcast && x.CommDestroy;
        return x;
    }();
    return r;
}

int rccl_fail(ncclResult_t r, const char *what)
{
    const char *s = rccl().GetErrorString ? rccl().GetErrorString(r) : "?";
    return fail(GRAIL_ERR_RCCL, std::string(what) + ": " + s);
}

}  // namespace

struct grail_ctx {
    int device = 0;
    int cus = 256;                    // compute units the launch policy plans for (hipDeviceProp_t::multiProcessorCount;
                                      // "assume_compute_units" overrides it): every capacity of the policy is a multiple
    int device_cus = 256;             // ... what the device reported
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool have_timing = false;
    std::vector<grail_voice> voices;  // host copy of the table
    DevVoice *d_voices = nullptr;
    float *d_voice_elems = nullptr;   // [n_voices * NUM_VOICED][49]
    bool voices_upper_silent = false; // every voice: formants 5-8 have amplitude +0 in every phoneme
    bool voices_live4_ok = false;     // ... and parameters that keep their output at exactly +0 (live4_ok)
    bool voices_scan_ok = false;      // every formant of every voice inside the safe window (scan_voice_ok)
    int scan_debug = 0;
    int sort_option = 1;              // ragged batches: fill launch slots in order of decreasing length
    int64_t pipe8_max_groups = -1;    // eight-formant pipelined workgroups: up to so many (-1: two per CU)
    int64_t pipe4_max_groups = -1;    // four-formant pipelined workgroups (16 utterances each): up to so many (-1: two per CU)
    int scan_option = 1;              // fast arithmetic: small batches go to the time-parallel scan kernel
    int64_t scan_max_utts = -1;       // ... up to this many utterances (x 4/7 with eight live formants; -1: 34 per CU = 8704)
    int64_t scan_split_max = -1;      // ... and up to this many with the carrier phase on a wave of its own (-1: 6 per CU = 1536)
    int composite_option = 1;         // a batch may be cut into blocks with a kernel family each (plan_blocks)
    double voices_sharpness = INFINITY;   // the largest predicted fast-mode deviation of the table, units of 2^-23
    std::vector<double> voice_sharpness;  // ... per voice (a batch is judged by the voices it uses)
    int64_t fast_limit = (int64_t)GRAIL_FAST_SHARPNESS_LIMIT;   // "fast_sharpness_limit": fast kernels up to this
    int mid_option = 1;               // "fast_exact_coefficients": sharper voices get the second tolerance tier (MID)
    int64_t mid_limit = (int64_t)GRAIL_FAST_SHARPNESS_LIMIT_EXACT_COEFFICIENTS;   // ... up to this sharpness
    bool voices_split_ok = false;     // every voice has a warm-up length (voice_warmup): time-split fast kernels
    uint32_t max_warmup = 0;          // ... the longest of them
    float max_rate = 0.0f;            // highest sample rate of the table
    int split_option = 1;             // fast arithmetic: mid-size batches split every utterance's time axis over lanes
    int64_t split_chunks = 0;         // ... into this many chunks (0: as many as fill the machine)
    int64_t split_span = 0;           // ... laid out over this many samples (0: the batch's longest utterance)
    int64_t split_ff_permille = 165;  // ... cost of a fast-forwarded sample against a rendered one
    int64_t split_min_utts = -1;      // ... -1: the cost model picks between the scan kernel, the time-split kernels and the lane
                                      // kernels (family_cost; 2 s utterances: 1 024 of them 2.00 (scan) against 3.13 ms (split),
                                      // 1 536: 3.22 / 3.14, 2 048: 3.33 / 3.13, profiles/r03_small_batch.txt); >= 0: batches smaller
                                      // than this (x 5/6 with eight live formants) stay with the scan kernel, whatever their length
    int last_split = 0;               // chunks of the last launch (statistics; 0: not time-split)
    int last_fast = 0;                // the last launch ran tolerance arithmetic in some block
    int last_blocks = 0;              // kernel launches the last synthesis call was cut into
    float max_dt = 0.0f;              // largest 1/sample_rate of the table
    float max_pitch_jitter = 0.0f;    // largest |jitter_delta_frequency| of the table
    int last_formants = 8, last_lanes = 0, last_pipe = 0;   // what the last synthesis launch used (statistics)
    uint32_t *d_truncated = nullptr;  // [0] truncation flag, [1] slow-path wave-steps, [2] fast wave-tiles, [3] general wave-steps
    uint64_t slow_steps = 0;          // of the kernels synced so far
    uint64_t fast_tiles = 0, general_steps = 0;
    uint32_t seen_counters[4] = {0, 0, 0, 0};   // d_truncated[1..3] as last read: the device counters only ever grow
    int lanes_option = 0;             // 0 = auto
    int skip_silent_option = 1;       // skip band-pass filters of provably silent formants
    int pipeline_option = 1;          // small qualifying batches: producer/consumer workgroups
    int pipe_round32 = 1;             // ... with rounds of 32 samples while one workgroup per CU suffices (8.20 -> 7.86 ms for config 2)
    uint64_t voices_epoch = 0;        // bumped by every install_voices
    uint64_t options_epoch = 0;       // bumped by every grail_set_option (a batch caches its launch plan against both)
    int fast_option = 0;              // "arithmetic": 0 exact (bit-identical), 1 fast (stated tolerance: the tier the voices'
                                      // sharpness allows), 2 fast with the reference's own coefficients (MID) whatever the voices
    std::string last_kernel = "none"; // instantiation of the last synthesis launch
    ncclComm_t comm = nullptr;
    uint32_t comm_rank = 0, comm_world = 1;
    void *host_pipe = nullptr;        // HostPipe: streams, events and buffers of the host-output path
};

struct grail_stream {
    const grail_batch *batch = nullptr;
    uint32_t *d_state = nullptr;   // [state_words(L)][lanes]
    uint64_t lanes = 0;
    int L = 1;
    bool started = false;
    // the kernel flavour, fixed when the stream is opened (the state layout follows it)
    bool live4 = false, half_capable = false, any_blend = false;
    uint64_t voices_epoch = 0;
    // live streams (grail_stream_open_live): the stream owns its batch, whose segments sit in per-utterance rings
    grail_batch *own = nullptr;
    uint32_t ring_cap = 0;            // segments per utterance ring (a power of two); 0: not a live stream
    uint32_t *d_counts = nullptr;     // [n_utt] segments appended so far
    uint32_t *d_open = nullptr;       // [n_utt] 1 while the utterance's source may deliver more
    uint32_t *d_consumed = nullptr;   // [n_utt] segments the Sequencer has pulled (written by the kernels)
    std::vector<uint32_t> appended;   // host copy of d_counts
    std::vector<uint32_t> consumed;   // what the host last read of d_consumed (a lower bound)
    std::vector<uint8_t> open;        // host copy of d_open
    std::vector<grail_synthesis_elem> last_elem;   // elem mode: the last elem appended per utterance (sharpness of the next pair)
    std::vector<uint8_t> last_has;
    // staging of an append (kept: an interactive front end appends a phoneme every half second for hours)
    DevSeg *d_new = nullptr;
    float *d_new_elems = nullptr;
    uint32_t *d_new_offs = nullptr;
    size_t new_cap = 0;
};

struct grail_batch {
    DevSeg *d_segs = nullptr;
    uint32_t *d_offsets = nullptr;
    uint32_t *d_voice_ids = nullptr;
    uint32_t *d_seeds = nullptr;
    uint32_t *d_perm = nullptr;   // ragged batches: launch slot -> utterance, longest first
    float *d_elems = nullptr;  // elem mode only
    uint32_t n_utt = 0;
    uint32_t n_segs = 0;
    uint32_t max_voice_id = 0;
    bool phoneme_mode = true;
    bool any_blend = false;    // some segment's blend length is not +-2^k (selects the kernel)
    bool plain = false;        // every length / blend length / pitch finite, blend lengths > 0
    float max_seconds = 0.0f;  // longest utterance: sum of its segment lengths
    float min_length = 0.0f;   // shortest segment (plain batches)
    float min_pitch = 0.0f;    // lowest frequency.min(0.5) of any segment (plain batches)
    double elems_sharpness = 0.0;   // elem mode: predicted fast-mode deviation of the caller's elems (elems_sharpness())
    std::vector<uint32_t> used_voices;   // the distinct voice ids of the batch, ascending
    // the launch plan of the last synthesis call of this batch (plan_blocks lays out time-split grids by bisection: a
    // fraction of a millisecond of host time, which a one-millisecond kernel should not pay at every launch)
    mutable struct PlanCache *plan_cache = nullptr;
};

// the SIMDs and lanes the policy plans for: 4 SIMDs per compute unit, 64 lanes per wavefront.  Every family is laid out
// for ONE resident wave per SIMD (a second wave on a SIMD costs as much as it brings: profiles/r01_lanes_sweep.txt), so
// all capacities below are multiples of the compute-unit count hipGetDeviceProperties reports (a partitioned MI355X —
// CPX, 32 CUs — plans for 32, not 256); "assume_compute_units" overrides it for tests.
static inline uint64_t ctx_simds(const grail_ctx *ctx) { return 4ull * (uint64_t)ctx->cus; }
static inline uint64_t ctx_lanes(const grail_ctx *ctx) { return 256ull * (uint64_t)ctx->cus; }
static inline int64_t pipe4_groups(const grail_ctx *ctx) { return ctx->pipe4_max_groups < 0 ? 2 * (int64_t)ctx->cus : ctx->pipe4_max_groups; }
static inline int64_t pipe8_groups(const grail_ctx *ctx) { return ctx->pipe8_max_groups < 0 ? 2 * (int64_t)ctx->cus : ctx->pipe8_max_groups; }
static inline int64_t scan_max_utts(const grail_ctx *ctx) { return ctx->scan_max_utts < 0 ? 34 * (int64_t)ctx->cus : ctx->scan_max_utts; }
static inline int64_t scan_split_max(const grail_ctx *ctx) { return ctx->scan_split_max < 0 ? 6 * (int64_t)ctx->cus : ctx->scan_split_max; }

namespace {

int bind(grail_ctx *ctx)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    return GRAIL_OK;
}

template <typename Tp>
int upload(Tp **dst, const void *src, size_t count, hipStream_t stream)
{
    *dst = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void **)dst, count * sizeof(Tp)));
    if (src) HIP_TRY(hipMemcpyAsync(*dst, src, count * sizeof(Tp), hipMemcpyHostToDevice, stream));
    return GRAIL_OK;
}

void free_plan_cache(struct PlanCache *p);

void free_batch_buffers(grail_batch *b)
{
    free_plan_cache(b->plan_cache);
    b->plan_cache = nullptr;
    if (b->d_segs) (void)hipFree(b->d_segs);
    if (b->d_offsets) (void)hipFree(b->d_offsets);
    if (b->d_voice_ids) (void)hipFree(b->d_voice_ids);
    if (b->d_seeds) (void)hipFree(b->d_seeds);
    if (b->d_perm) (void)hipFree(b->d_perm);
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

// Ragged batches: the lanes of a wave run in lockstep, so a wave lasts as long as its longest utterance.
// Launch slots are therefore filled in order of decreasing length (sum of the segment lengths, in seconds
// — close enough to the sample count for sorting): the utterances of a wave end together, and when the
// batch is larger than the machine the longest waves start first.  Results do not depend on the slot
// (batch invariance), rows stay where the caller put them.  Aligned batches (all sums equal) keep the
// identity assignment and pay nothing.
int upload_length_order(grail_ctx *ctx, grail_batch *b, const std::vector<float> &seconds, uint32_t n_utt)
{
    if (n_utt < 2 || !ctx->sort_option) return GRAIL_OK;
    float lo = seconds[0], hi = seconds[0];
    for (uint32_t u = 1; u < n_utt; ++u) {
        lo = std::fmin(lo, seconds[u]);
        hi = std::fmax(hi, seconds[u]);
    }
    if (!(hi - lo > 0.002f)) return GRAIL_OK;       // aligned (or NaN lengths): nothing to gain
    std::vector<uint32_t> perm(n_utt);
    for (uint32_t u = 0; u < n_utt; ++u) perm[u] = u;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t c) { return seconds[a] > seconds[c]; });
    int rc = upload(&b->d_perm, perm.data(), n_utt, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
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

// Can formants 5-8 of this voice be left out of a one-shot render altogether?  They must contribute
// exactly +0.0 to every sample of the reference's own arithmetic, whatever the segments are (given
// alpha in [0,1], i.e. no segment shorter than two samples — checked per batch):
//   amplitude exactly +0 in every phoneme, 0 <= jitter_delta_amplitude/2 <= 1/4  => v0 = tw * (+0) = +-0
//   breath, turbulence, smoothness in [0,1]                                      => the low-pass state and tw stay finite
//   frequency and bandwidth inside pair_is_safe's window with the jitter margin  => finite g, k and 0 < a1, a2, a3 < inf
// and then w1 = a1*(+0) + a2*(+-0) = +0 and the band-pass state never leaves +0 ("Silent
// formants").  SynthesisElem::silent() (0.25 / 0.25 / 0.25 / 0 / 0 / 0) satisfies all of it.
bool live4_ok(const grail_voice &v)
{
    constexpr float X_LO = 9.5367431640625e-07f, X_HI = 0.5f - 9.5367431640625e-07f;
    constexpr float W_LO = 1.8189894035458565e-12f, W_HI = 512.0f;
    const float amp_scale = 0.5f * v.jitter_delta_amplitude;
    const float jm = 1.002f * std::fabs(v.jitter_delta_formant_frequency);
    bool ok = (amp_scale >= 0.0f) && (amp_scale <= 0.25f) && (jm <= 1.0f) &&
              (v.jitter_frequency >= 0.0f) && (v.jitter_frequency <= 1.0f) &&
              (v.sample_rate > 0.0f) && std::isfinite(v.sample_rate) &&
              std::isfinite(v.jitter_delta_frequency);
    for (int p = 0; p < NUM_VOICED && ok; ++p) {
        const grail_synthesis_elem &e = v.phonemes[p];
        for (int i = NF / 2; i < NF && ok; ++i) {
            uint32_t bits;
            std::memcpy(&bits, &e.formant_amp[i], sizeof bits);
            const float f = e.formant_freq[i], w = e.formant_bw[i];
            ok = bits == 0u && e.formant_breath[i] >= 0.0f && e.formant_breath[i] <= 1.0f &&
                 e.formant_turb[i] >= 0.0f && e.formant_turb[i] <= 1.0f &&
                 e.formant_smooth[i] >= 0.0f && e.formant_smooth[i] <= 1.0f &&
                 (f * 0.999f - jm >= X_LO) && (f * 1.001f + jm <= X_HI) && (w >= W_LO) && (w <= W_HI);
        }
    }
    return ok;
}

// Can this voice go through the time-parallel scan kernel (fast arithmetic, small batches)?  That path
// has no IEEE-division fallback: every formant of every phoneme must sit inside pair_is_safe's window
// with the jitter margin, and the jitter parameters must be sane.  SynthesisElem::silent() qualifies.
bool scan_voice_ok(const grail_voice &v)
{
    constexpr float X_LO = 9.5367431640625e-07f, X_HI = 0.5f - 9.5367431640625e-07f;
    constexpr float W_LO = 1.8189894035458565e-12f, W_HI = 512.0f;
    const float amp_scale = 0.5f * v.jitter_delta_amplitude;
    const float jm = 1.002f * std::fabs(v.jitter_delta_formant_frequency);
    bool ok = std::isfinite(amp_scale) && (jm <= 1.0f) && (v.jitter_frequency >= 0.0f) &&
              (v.jitter_frequency <= 0.25f) && (v.sample_rate > 0.0f) && std::isfinite(v.sample_rate) &&
              std::isfinite(v.jitter_delta_frequency);
    for (int p = 0; p < NUM_VOICED && ok; ++p) {
        const grail_synthesis_elem &e = v.phonemes[p];
        for (int i = 0; i < NF && ok; ++i) {
            const float f = e.formant_freq[i], w = e.formant_bw[i];
            ok = std::isfinite(e.formant_amp[i]) && std::isfinite(e.formant_breath[i]) &&
                 std::isfinite(e.formant_turb[i]) && e.formant_smooth[i] >= 0.0f && e.formant_smooth[i] <= 1.0f &&
                 (f * 0.999f - jm >= X_LO) && (f * 1.001f + jm <= X_HI) && (w >= W_LO) && (w <= W_HI);
        }
    }
    return ok;
}

// Time-split fast kernels: how many samples until a filter state that started from zero is within 2^-21 of the
// state the reference would have (relative to the state's size, which is below full scale)?  The chain of
// Synthesize::next per formant is a one-pole low-pass with factor exp_approx(smooth) = (1 - smooth)^5 (:535-538)
// and the trapezoidal state-variable band-pass (:555-571), whose poles are the bilinear images
// z = (1 + s) / (1 - s) of s = g (-k/2 +- sqrt(k^2/4 - 1)), g = tan_approx(freq), k = bw / freq; for k < 2,
// |z|^2 = (1 - g k + g^2) / (1 + g k + g^2) ~ exp(-2 pi bw).  The slowest of them over every phoneme (blends
// move the parameters between phonemes and towards silent()'s 0.25 / 0.25 / 0.25, which decays at once) over the range
// the formant-frequency jitter moves the band-pass through, with a 5 % margin, gives the length; where the low-pass and
// the band-pass decay at nearly the same rate the cascade's n rho^n is solved for instead of rho^n.  Formants that are silent in
// every phoneme have nothing to converge.  0: the voice does not qualify (a parameter outside the window, or a
// warm-up longer than 16384 samples).
uint32_t voice_warmup(const grail_voice &v)
{
    // per-sample decay rate of the band-pass envelope at formant frequency f, bandwidth w (0: not a decaying filter)
    auto svf_rate = [](double f, double w) -> double {
        if (!(f > 0.0 && f < 0.5)) return 0.0;
        const double g = ((1 - f) * f * (5 - 4 * (f + 0.5) * (0.5 - f))) / ((f + 0.5) * (5 - 4 * (1 - f) * f) * (0.5 - f));
        const double k = w / f;
        double z;
        if (k < 2.0) {
            z = std::sqrt((1 - g * k + g * g) / (1 + g * k + g * g));
        } else {
            const double root = std::sqrt(k * k / 4 - 1);
            const double s1 = g * (-k / 2 + root), s2 = g * (-k / 2 - root);
            z = std::fmax(std::fabs((1 + s1) / (1 - s1)), std::fabs((1 + s2) / (1 - s2)));
        }
        return (z > 0.0 && z < 1.0) ? -std::log(z) : 0.0;
    };
    const double eps = 1.0 / 2097152.0;                     // 2^-21
    const double jd = std::fabs((double)v.jitter_delta_formant_frequency);
    if (!std::isfinite(jd)) return 0;
    double longest = 0.0;                                   // samples
    bool any = false;
    for (int i = 0; i < NF; ++i) {
        bool audible = false;
        for (int p = 0; p < NUM_VOICED; ++p) audible = audible || !(v.phonemes[p].formant_amp[i] == 0.0f);
        if (!audible) continue;
        any = true;
        for (int p = 0; p < NUM_VOICED; ++p) {
            const grail_synthesis_elem &e = v.phonemes[p];
            const double f = e.formant_freq[i], w = e.formant_bw[i], sm = e.formant_smooth[i];
            if (!(f > 0.0 && f < 0.5 && w > 0.0 && sm > 0.0 && sm < 1.0) || !std::isfinite(w)) return 0;
            // the formant-frequency jitter moves the band-pass by up to +-jitter_delta_formant_frequency (Jitter::next
            // :764 adds noise in [-1, 1] times it): the slowest decay over that range
            double l_bp = svf_rate(f, w);
            for (const double ff : {f - jd, f + jd})
                if (ff > 0.0 && ff < 0.5) l_bp = std::fmin(l_bp, svf_rate(ff, w));
            const double l_lp = -5.0 * std::log1p(-sm);                          // (1 - smooth)^5 per sample
            if (!(l_bp > 0.0) || !(l_lp > 0.0)) return 0;
            const double slow = std::fmin(l_bp, l_lp), gap = std::fabs(l_bp - l_lp);
            // The low-pass feeds the band-pass: what is left of a wrong start after n samples is bounded by
            // rho^n + sum_j rho_bp^(n-1-j) rho_lp^j, i.e. by (1 + m) rho^n with m = min(n, 1 / |rate difference|).  Far
            // apart (every shipped voice: 0.17 against 0.004 per sample) m is a few samples' worth and the 5 % margin
            // covers it; when the two rates are within a fifth of each other the residual decays like n rho^n and the
            // length is solved for that.
            double n = std::log(1.0 / eps) / (0.95 * slow);
            if (gap <= 0.2 * std::fmax(l_bp, l_lp))
                for (int it = 0; it < 4; ++it) n = std::log((1.0 + std::fmin(n, 1.0 / std::fmax(gap, 1e-12))) / eps) / (0.95 * slow);
            longest = std::fmax(longest, n);
        }
    }
    if (!any) return 64;                                    // nothing audible: any state is the right one
    if (!(longest <= 16384.0)) return 0;
    return ((uint32_t)std::ceil(longest) + 63u) / 64u * 64u;
}

// Fast arithmetic and sharp resonances.  The fast kernels interpolate the filter coefficients of Synthesize::next
// (:555-562) between points evaluated with fused and reordered operations; the reference rounds every operation
// anew at every sample.  A rounding-level difference of a coefficient that lasts for a sub-tile moves the
// resonance of a band-pass by that much of its centre frequency and its damping by that much of one, i.e. the
// output by (difference) x Q resp. x (ring time) of the formant's amplitude — the same amplification the
// reference's own rounding gets (its binary32 rendering is about a third as far from its formulas in double
// precision).  Measured (tools/q_sweep.py: bandwidth sweeps of the shipped voices; a frequency x bandwidth grid of
// single formants; tools/sharpness_data.py: 1 400 random tables; profiles/r03_sharpness.txt): the deviation of a
// single formant that carries all of the amplitude is ~ 2 500 / bandwidth [Hz at 48 kHz] * 2^-23 up to 2.4 kHz
// and grows with the square of the frequency above that; it is proportional to the formant's share of the
// amplitudes, and the formants add up in quadrature.  Hence
//     E_i = share_i * (0.0709 / bw_i) * (1 + (f_i / 0.075)^2)     (f, bw in cycles per sample, as in the elems;
//     S   = sqrt(sum_i E_i^2)                                       share, bw, f: the worst of the phonemes)
// which, scaled as it is, lies above 99.5 % of the random tables' measured deviations and within a factor 1.45
// below the rest; voices::generic() has S = 24 (measured 13 - 20), the bench presets 20 - 22 (11 - 20).
// GRAIL_FAST_TOLERANCE = 64 * 2^-23 is therefore a promise the fast kernels can keep only up to a sharpness: the
// host serves fast arithmetic for S <= GRAIL_FAST_SHARPNESS_LIMIT = 28 (worst measured among those, 3 000 random tables:
// 24; at 32 one table in 3 000 reached 56) and
// renders sharper tables with the exact kernels (their bits satisfy the tolerance trivially).
// Returns S in units of 2^-23 of max(1, peak); +inf for parameters outside the window of the formulas.
double elems_sharpness(const grail_synthesis_elem *elems, size_t n)
{
    double share[NF] = {0}, sens[NF] = {0};
    for (size_t p = 0; p < n; ++p) {
        double total = 0.0;
        for (int i = 0; i < NF; ++i) total += std::fabs((double)elems[p].formant_amp[i]);
        if (!std::isfinite(total)) return INFINITY;
        for (int i = 0; i < NF; ++i)
            if (total > 0.0) share[i] = std::fmax(share[i], std::fabs((double)elems[p].formant_amp[i]) / total);
    }
    double sum = 0.0;
    for (int i = 0; i < NF; ++i) {
        if (share[i] == 0.0) continue;             // never audible: nothing rings
        for (size_t p = 0; p < n; ++p) {
            const double f = elems[p].formant_freq[i], w = elems[p].formant_bw[i];
            if (!(f > 0.0 && f < 0.5 && w > 0.0) || !std::isfinite(w)) return INFINITY;
            sens[i] = std::fmax(sens[i], (0.0709 / w) * (1.0 + (f / 0.075) * (f / 0.075)));
        }
        sum += (share[i] * sens[i]) * (share[i] * sens[i]);
    }
    return std::sqrt(sum);
}
// Is fast arithmetic served for this batch?  Caller-built elems are judged themselves; a phoneme batch by the sharpest of
// the voices IT USES (one sharp voice in the table does not take fast arithmetic away from batches that never name it);
// without a batch: by the whole table.
double batch_sharpness(const grail_ctx *ctx, const grail_batch *batch)
{
    if (!batch) return ctx->voices_sharpness;
    if (!batch->phoneme_mode) return batch->elems_sharpness;
    double s = 0.0;
    for (const uint32_t v : batch->used_voices) s = std::fmax(s, v < ctx->voice_sharpness.size() ? ctx->voice_sharpness[v] : INFINITY);
    return s;
}
// Which arithmetic a batch is rendered in when "arithmetic" asks for a tolerance mode: 1 = the interpolating tier (up
// to "fast_sharpness_limit"), 2 = the reference's own band-pass coefficients at every sample (MID; sharper voices, up to
// "fast_sharpness_limit_exact_coefficients"), 0 = the exact kernels (sharper still, or the tier switched off).
int fast_tier_for(const grail_ctx *ctx, const grail_batch *batch, int arithmetic)
{
    if (!arithmetic) return 0;
    const double s = batch_sharpness(ctx, batch);
    if (arithmetic == 1 && s <= (double)ctx->fast_limit) return 1;
    if ((ctx->mid_option || arithmetic == 2) && s <= (double)ctx->mid_limit) return 2;
    return 0;
}
int fast_tier(const grail_ctx *ctx, const grail_batch *batch) { return fast_tier_for(ctx, batch, ctx->fast_option); }
// The chunk grid of a time-split launch: K chunks over `span` samples.  Chunk k's lane fast-forwards the chain over
// b[k] - W samples (cost r per sample, in units of a rendered sample), warms up over W and renders b[k+1] - b[k]:
// the bounds are spaced so that all lanes take the same time (T below, by bisection).  Bounds are multiples
// of 64; the last one is left to the caller (the row capacity).  false: K chunks do not fit.
bool split_grid(uint32_t span, uint32_t warmup, int K, double r, uint32_t *b)
{
    auto lay = [&](double T, double *out) {
        double at = 0.0;
        for (int k = 0; k < K; ++k) {
            out[k] = at;
            const double before = k ? r * std::fmax(at - warmup, 0.0) + std::fmin((double)warmup, at) : 0.0;
            const double len = T - before;
            if (len < 64.0) return -1.0;
            at += len;
        }
        return at;
    };
    double lo = 0.0, hi = (double)span + warmup + 64.0, pos[SPLIT_MAX_CHUNKS + 1];
    if (lay(hi, pos) < (double)span) return false;
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        const double end = lay(mid, pos);
        if (end < 0.0 || end < (double)span) lo = mid;
        else hi = mid;
    }
    if (lay(hi, pos) < 0.0) return false;
    b[0] = 0;
    for (int k = 1; k < K; ++k) {
        b[k] = ((uint32_t)pos[k] + 32u) / 64u * 64u;
        if (b[k] <= b[k - 1]) return false;
    }
    return b[K - 1] < span;
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
    ++ctx->voices_epoch;
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
    for (uint32_t v = 0; v < n_voices; ++v) {
        ctx->voices_live4_ok = ctx->voices_live4_ok && live4_ok(voices[v]);
        ctx->voices_split_ok = ctx->voices_split_ok && dv[v].warmup != 0u && voices[v].sample_rate > 0.0f &&
                               std::isfinite(voices[v].sample_rate);
        if (dv[v].warmup > ctx->max_warmup) ctx->max_warmup = dv[v].warmup;
        if (voices[v].sample_rate > ctx->max_rate) ctx->max_rate = voices[v].sample_rate;
        const float dt = 1.0f / voices[v].sample_rate;
        if (!(dt <= ctx->max_dt)) ctx->max_dt = dt;       // NaN-proof max
        const float pj = std::fabs(voices[v].jitter_delta_frequency);
        if (!(pj <= ctx->max_pitch_jitter)) ctx->max_pitch_jitter = pj;
    }
    return GRAIL_OK;
}

}  // namespace

static void pipe_destroy_opaque(void *p);

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

float grail_fast_sharpness(const grail_voice *voice)
{
    return voice ? (float)elems_sharpness(voice->phonemes, NUM_VOICED) : INFINITY;
}

uint32_t grail_time_split_warmup(const grail_voice *voice) { return voice ? voice_warmup(*voice) : 0u; }

int grail_time_split_grid(uint32_t span_samples, uint32_t warmup, uint32_t chunks, uint32_t ff_cost_permille,
                          uint32_t *bounds)
{
    if (!bounds) return fail(GRAIL_ERR_INVALID_ARG, "bounds is NULL");
    if (chunks < 2u || chunks > (uint32_t)SPLIT_MAX_CHUNKS)
        return fail(GRAIL_ERR_INVALID_ARG, "chunks must be 2..64");
    if (ff_cost_permille > 1000u) return fail(GRAIL_ERR_INVALID_ARG, "ff_cost_permille must be 0..1000");
    uint32_t b[SPLIT_MAX_CHUNKS + 1];
    if (!split_grid(span_samples, warmup, (int)chunks, 1e-3 * (double)ff_cost_permille, b))
        return fail(GRAIL_ERR_INVALID_ARG, "so many chunks do not fit the span");
    std::memcpy(bounds, b, sizeof(uint32_t) * chunks);
    return GRAIL_OK;
}

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
        (err = hipMalloc((void **)&ctx->d_truncated, 4 * sizeof(uint32_t))) != hipSuccess ||
        (err = hipMemsetAsync(ctx->d_truncated, 0, 4 * sizeof(uint32_t), ctx->stream)) != hipSuccess ||
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
    if (ctx->comm && rccl().ok) rccl().CommDestroy(ctx->comm);
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
    if (std::strcmp(name, "pipeline8_max_groups") == 0) {   // tuning: 0 keeps eight-formant batches off the pipeline
        ctx->pipe8_max_groups = value;
        return GRAIL_OK;
    }
    if (std::strcmp(name, "pipeline_round32") == 0) {       // tuning (A/B)
        ctx->pipe_round32 = value ? 1 : 0;
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
    bool any_blend = false;
    for (uint32_t i = 0; i < n_segs && !any_blend; ++i) any_blend = !blend_is_pow2(segs[i].blend_length);
    bool plain = true;
    float min_length = INFINITY, min_pitch = INFINITY;
    for (uint32_t i = 0; i < n_segs; ++i) {
        plain = plain && std::isfinite(segs[i].length) && std::isfinite(segs[i].blend_length) &&
                std::isfinite(segs[i].frequency) && segs[i].blend_length > 0.0f;
        if (segs[i].length < min_length) min_length = segs[i].length;
        const float pitch = std::fmin(segs[i].frequency, 0.5f);   // copy_with_frequency :445-450
        if (pitch < min_pitch) min_pitch = pitch;
    }
    for (uint32_t i = 0; i < n_segs; ++i)
        if (segs[i].phoneme < 0 || segs[i].phoneme >= GRAIL_PH_COUNT)
            return fail(GRAIL_ERR_INVALID_ARG, "phoneme discriminant out of range");
    grail_batch *b = new (std::nothrow) grail_batch();
    if (!b) return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->phoneme_mode = true;
    b->any_blend = any_blend;
    b->plain = plain;
    b->min_length = min_length;
    b->min_pitch = min_pitch;
    b->n_segs = n_segs;
    std::vector<float> seconds(n_utt, 0.0f);
    for (uint32_t u = 0; u < n_utt; ++u)
        for (uint32_t i = seg_offsets[u]; i < seg_offsets[u + 1]; ++i) seconds[u] += segs[i].length;
    for (uint32_t u = 0; u < n_utt; ++u)
        if (seconds[u] > b->max_seconds) b->max_seconds = seconds[u];
    if ((rc = upload(&b->d_segs, segs, n_segs, ctx->stream)) ||
        (rc = upload_common(ctx, b, seg_offsets, voice_ids, jitter_seeds, n_utt)) ||
        (rc = upload_length_order(ctx, b, seconds, n_utt))) {
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
    bool any_blend = false;
    for (uint32_t i = 0; i < n_segs; ++i) {
        ds[i].elem = segs[i].has_elem ? (int32_t)i : -1;
        ds[i].length = segs[i].length;
        ds[i].blend_length = segs[i].blend_length;
        any_blend = any_blend || !blend_is_pow2(segs[i].blend_length);
        ds[i].frequency = segs[i].elem.frequency;
        std::memcpy(&elems[(size_t)i * ELEM_FLOATS], &segs[i].elem, sizeof(grail_synthesis_elem));
    }
    grail_batch *b = new (std::nothrow) grail_batch();
    if (!b) return fail(GRAIL_ERR_OUT_OF_MEMORY, "host allocation failed");
    b->phoneme_mode = false;
    b->any_blend = any_blend;
    b->n_segs = n_segs;
    // the sharpness of the batch (elems_sharpness): parameters only ever blend between the elems of two consecutive
    // segments of an utterance (Sequencer::next :897-921), so every such pair is judged like a voice of two phonemes
    for (uint32_t u = 0; u < n_utt; ++u)
        for (uint32_t i = seg_offsets[u]; i < seg_offsets[u + 1]; ++i) {
            if (!segs[i].has_elem) continue;
            grail_synthesis_elem pair[2] = {segs[i].elem, segs[i].elem};
            size_t n_pair = 1;
            if (i + 1 < seg_offsets[u + 1] && segs[i + 1].has_elem) pair[n_pair++] = segs[i + 1].elem;
            b->elems_sharpness = std::fmax(b->elems_sharpness, elems_sharpness(pair, n_pair));
        }
    std::vector<float> seconds(n_utt, 0.0f);
    for (uint32_t u = 0; u < n_utt; ++u)
        for (uint32_t i = seg_offsets[u]; i < seg_offsets[u + 1]; ++i) seconds[u] += segs[i].length;
    for (uint32_t u = 0; u < n_utt; ++u)
        if (seconds[u] > b->max_seconds) b->max_seconds = seconds[u];
    if ((rc = upload(&b->d_segs, ds.data(), n_segs, ctx->stream)) ||
        (rc = upload(&b->d_elems, elems.data(), elems.size(), ctx->stream)) ||
        (rc = upload_common(ctx, b, seg_offsets, voice_ids, jitter_seeds, n_utt)) ||
        (rc = upload_length_order(ctx, b, seconds, n_utt))) {
        free_batch_buffers(b);
        delete b;
        return rc;
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

static int check_ready(grail_ctx *ctx, const grail_batch *batch)
{
    if (!batch) return fail(GRAIL_ERR_INVALID_ARG, "batch is NULL");
    if (ctx->voices.empty() || !ctx->d_voices || !ctx->d_voice_elems)   // also after a failed upload
        return fail(GRAIL_ERR_NO_VOICES, "call grail_set_voices first");
    if (batch->max_voice_id >= ctx->voices.size())
        return fail(GRAIL_ERR_INVALID_ARG, "a voice id exceeds the voice table");
    return GRAIL_OK;
}

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

// ---- which kernel family renders a block of rows, what a block costs, how a batch is cut into blocks --------------

struct Family {
    int L = 1;                 // lanes per utterance (lane kernels, pipelined workgroups)
    uint32_t pipe = 0;         // exact pipelined workgroups: 1 = rounds of 16 samples, 2 = rounds of 32
    uint32_t live4 = 0;        // formants 5-8 not laid out
    uint32_t fast = 0;         // tolerance arithmetic
    int split_k = 0;           // time-split kernels: chunks per utterance (0: not time-split)
    uint32_t split_bounds[SPLIT_MAX_CHUNKS + 1] = {};
    bool scan = false;         // the time-parallel scan kernel
    uint32_t scan_pipe = 0;    // ... its three-stage flavour
};

// the longest utterance of the batch in samples, as far as the host knows it (the f32 clock adds a few per segment)
static double batch_span(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride)
{
    double span = std::ceil((double)batch->max_seconds * ctx->max_rate) + 64.0;
    if (!(span >= 64.0)) span = 64.0;                         // NaN / negative lengths
    return std::fmin(span, (double)(out_stride ? out_stride : 1));
}

// Cost model of the planner, in milliseconds per SAMPLE OF THE LONGEST UTTERANCE for one round of a family (a round:
// as many rows as give every SIMD one wave).  Calibrated on 2 s utterances at 48 kHz, one MI355X
// (profiles/r03_small_batch.txt, profiles/r04_duration_sweep.txt); only the ratios matter.  Indexed [L = 1, 2, 4, 8].
constexpr double MID_MS_4 = 32.6, MID_MS_8 = 57.2;     // 65 536 x 2 s, the MID kernels (profiles/r04_middle_tier.txt)
static double lane_ms_per_sample(bool fast, bool live4, int L)
{
    static const double exact4[4] = {40.6, 26.9, 16.3, 18.2}, exact8[4] = {77.1, 43.5, 25.8, 15.7};
    static const double fast4[4] = {15.6, 13.0, 12.1, 11.7}, fast8[4] = {23.2, 19.9, 13.3, 11.7};
    const int i = L == 1 ? 0 : L == 2 ? 1 : L == 4 ? 2 : 3;
    return (fast ? (live4 ? fast4 : fast8) : (live4 ? exact4 : exact8))[i] / 96006.0;
}
// ... of the second tolerance tier (MID, one lane per utterance)
static double mid_ms_per_sample(bool live4) { return (live4 ? MID_MS_4 : MID_MS_8) / 96006.0; }

// what launching `rows` rows with family f costs (model milliseconds)
static double family_cost(const grail_ctx *ctx, const Family &f, uint32_t rows, double span)
{
    const double cus = (double)ctx->cus, lanes = (double)ctx_lanes(ctx);
    if (f.scan) {
        // one workgroup per utterance; g = workgroups per compute unit.  Three-stage flavour: the latency of one
        // utterance's chain up to ~2 per CU, then ~0.53 ms per workgroup and CU (2 s); two-stage: 0.41 (four live
        // formants) / 0.65 (eight)
        const double g = std::ceil((double)rows / cus);
        const double ms2s = f.scan_pipe ? (f.live4 ? std::fmax(1.14, 0.53 * g) : std::fmax(1.67, 0.75 * g))
                                        : (f.live4 ? 0.41 * g + 0.1 : 0.65 * g + 0.2);
        return ms2s * span / 96006.0;
    }
    if (f.split_k) {
        // every lane takes as long as the first chunk's, which renders split_bounds[1] samples and nothing else
        const double rounds = std::ceil((double)rows * f.split_k / lanes);
        // (+ 0.12 ms: what a launch of chunk lanes costs before any of them renders — short utterances see it)
        if (f.fast == 2u) return rounds * ((double)f.split_bounds[1] * mid_ms_per_sample(f.live4 != 0) + 0.12);
        return rounds * ((double)f.split_bounds[1] * (f.live4 ? 15.7 : 23.3) / 96006.0 + 0.12);
    }
    if (f.pipe) {
        const double groups = std::ceil((double)rows / (f.live4 ? 16.0 : 8.0));
        const double per_cu = std::ceil(groups / cus);
        // rounds of 32: one workgroup per CU; rounds of 16: two per CU are resident together, further ones queue
        const double ms2s = f.pipe == 2 ? (f.live4 ? 6.5 : 7.3) * per_cu : 11.2 * std::ceil(per_cu / 2.0);
        return ms2s * span / 96006.0;
    }
    const double rounds = std::ceil((double)rows * f.L / lanes);
    if (f.fast == 2u) return rounds * span * mid_ms_per_sample(f.live4 != 0);
    return rounds * span * lane_ms_per_sample(f.fast != 0, f.live4 != 0, f.L);
}

static bool batch_half_capable(const grail_ctx *ctx, const grail_batch *batch)
{
    return ctx->skip_silent_option && batch->phoneme_mode && ctx->voices_upper_silent;
}

// formants 5-8 left out altogether: the table qualifies (live4_ok); every segment is at least
// two samples long, so the Sequencer clock never goes negative and alpha stays in [0,1]; and
// every pitch stays >= 2^-20 under the pitch jitter, so the polyBLEP quotient and with it the
// saw every formant is fed from stay finite (a dead formant fed +-inf would emit NaN)
static bool batch_live4_any_blend(const grail_ctx *ctx, const grail_batch *batch)
{
    return batch_half_capable(ctx, batch) && ctx->voices_live4_ok && batch->plain &&
           batch->min_length >= 2.0f * ctx->max_dt &&
           batch->min_pitch * 0.999f - 1.002f * ctx->max_pitch_jitter >= 9.5367431640625e-07f;
}
// ... and (the lane kernels' four-formant instantiations) every blend length a power of two
static bool batch_live4(const grail_ctx *ctx, const grail_batch *batch)
{
    return batch_live4_any_blend(ctx, batch) && !batch->any_blend;
}

// The family a block of `fam` rows of this batch takes.  Exact arithmetic: the widest mapping that still gives every
// SIMD at most one wave (pipelined workgroups, then 8 / 4 / 2 / 1 lanes per utterance).  Fast arithmetic: the cheapest
// of the scan kernel, the time-split kernels and the fast lane kernels by the cost model above (which follows the
// utterances' length: a time-split pays a warm-up per chunk, the scan kernel the latency of one utterance's chain),
// unless an option pins the choice.
static void choose_family(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride, uint32_t fam, Family &f,
                          bool exact_only = false)
{
    const uint64_t simds = ctx_simds(ctx), lanes = ctx_lanes(ctx), cus = (uint64_t)ctx->cus;
    f = Family();
    f.live4 = batch_live4(ctx, batch) ? 1u : 0u;
    // fast arithmetic is served up to a sharpness of the resonances (elems_sharpness); beyond it the exact kernels run
    // ... in the tier the sharpness allows: 1 = coefficients interpolated, 2 = the reference's own coefficients (MID)
    f.fast = exact_only ? 0u : (uint32_t)fast_tier(ctx, batch);
    // (MID kernels exist one-shot with one lane per utterance, and time-split: a pinned wider mapping gets the exact kernels)
    if (f.fast == 2u && ctx->lanes_option > 1) f.fast = 0u;
    // (the fast lane kernels have four-formant instantiations for every blend length)
    if (f.fast && batch_live4_any_blend(ctx, batch)) f.live4 = 1u;
    int L = ctx->lanes_option ? ctx->lanes_option : auto_lanes_per_utt(fam, simds);
    if (f.fast == 2u) L = 1;          // (before the four-formant layout is decided: eight lanes would give it up)
    // small batches leave SIMDs idle: four-wave workgroups (one wave renders 16 utterances, one carries
    // the per-utterance chain, two prepare the filter coefficients), up to two per CU (tools/pipe4_range.py:
    // 11.5 ms up to 4 096 utterances, 15.7 up to 8 192 where the lane kernels take 18.0; three per CU lose)
    const bool want_pipe4 = batch_live4(ctx, batch) && !ctx->lanes_option && ctx->pipeline_option &&
                            (int64_t)(((uint64_t)fam + 15) / 16) <= pipe4_groups(ctx);
    const bool want_pipe8 = !batch_live4(ctx, batch) && !ctx->lanes_option && ctx->pipeline_option && !batch->any_blend &&
                            (int64_t)(((uint64_t)fam + 7) / 8) <= pipe8_groups(ctx);
    // one workgroup per CU suffices: rounds of 32 samples instead of 16 (pipe = 2)
    const uint32_t pipe4_kind = ctx->pipe_round32 && ((uint64_t)fam + 15) / 16 <= cus ? 2u : 1u;
    const uint32_t pipe8_kind = ctx->pipe_round32 && ((uint64_t)fam + 7) / 8 <= cus ? 2u : 1u;
    if (want_pipe4 && !f.fast) {
        f.pipe = pipe4_kind;
        L = 4;
    } else if (want_pipe8 && !f.fast) {
        f.pipe = pipe8_kind;                              // eight formants: 8 utterances per workgroup
        L = 8;
    }
    // eight lanes per utterance need eight formants to lay out; for batches that small the
    // 8-lane kernel is also the fastest (18.2 against 18.8 ms: half the rows to flush per wave)
    if (f.live4 && !f.pipe && L == 8) f.live4 = 0u;
    if (f.live4 && !f.pipe && !ctx->lanes_option) {
        // same rule as auto_lanes_per_utt — the widest mapping with one wave per SIMD — over 4 formants
        L = ((uint64_t)fam * 4 + 63) / 64 <= simds ? 4 : ((uint64_t)fam * 2 + 63) / 64 <= simds ? 2 : 1;
    }
    // voices whose upper formants are never audible but that do not qualify for the 4-formant
    // kernels: one lane per utterance runs the half-live loop and ties two lanes per utterance,
    // whose second lane would only hold silent formants
    if (!ctx->lanes_option && !f.live4 && L == 2 && batch_half_capable(ctx, batch)) L = 1;
    f.L = L;
    if (!f.fast) return;

    const double span = batch_span(ctx, batch, out_stride);
    const bool l4ab = batch_live4_any_blend(ctx, batch);
    // fast arithmetic, mid-size batches: one lane per utterance would leave most of the machine idle, so the time
    // axis of every utterance is cut into chunks with a lane each (synth_kernel<..., SPLIT>): as many chunks as
    // fill the machine, laid out over the batch's longest utterance so that all lanes finish together
    Family split = f;
    if (ctx->split_option && !ctx->lanes_option && batch->phoneme_mode && ctx->voices_split_ok && batch->plain &&
        out_stride <= 0xFFFFFFFFull && (ctx->split_chunks >= 2 || ctx->split_chunks == 0)) {
        const double sp = ctx->split_span ? std::fmin((double)ctx->split_span, (double)out_stride) : span;
        int K = ctx->split_chunks ? (int)ctx->split_chunks : (int)std::min<uint64_t>(lanes / fam, SPLIT_MAX_CHUNKS);
        K = (int)std::fmin((double)K, sp / 512.0);
        // (a fast-forwarded sample costs the same whatever is rendered afterwards; a rendered sample of eight live
        // formants costs 1.5 x one of four; 0.8 from a sweep, profiles/r03_small_batch.txt)
        const double ff_cost = 1e-3 * (double)ctx->split_ff_permille * (l4ab ? 1.0 : 0.8) * (f.fast == 2u ? 0.6 : 1.0);
        // the largest K <= K whose chunks fit (a chunk must render at least a tile): fitting is monotone in K
        if (K >= 2 && !split_grid((uint32_t)sp, ctx->max_warmup, K, ff_cost, split.split_bounds)) {
            int lo = 1, hi = K;                  // lo fits (or is 1), hi does not
            while (hi - lo > 1) {
                const int mid = (lo + hi) / 2;
                if (split_grid((uint32_t)sp, ctx->max_warmup, mid, ff_cost, split.split_bounds)) lo = mid;
                else hi = mid;
            }
            K = lo;
            if (K >= 2) (void)split_grid((uint32_t)sp, ctx->max_warmup, K, ff_cost, split.split_bounds);
        }
        if (K >= 2) {
            split.split_k = K;
            split.split_bounds[K] = (uint32_t)out_stride;
            split.live4 = l4ab ? 1u : 0u;
            split.pipe = 0u;
            split.L = 1;
        }
    }
    // fast arithmetic, few utterances: one workgroup per utterance with the time axis across the lanes and the
    // filter recurrences solved by parallel scans (scan_kernels.hip).  Needs every parameter inside the safe window
    // (no IEEE fallback).
    Family scan = f;
    if (f.fast == 1u && ctx->scan_option && !ctx->lanes_option && (int64_t)fam * (l4ab ? 4 : 7) <= 4 * scan_max_utts(ctx) &&
        batch->phoneme_mode && ctx->voices_scan_ok && batch->plain && batch->min_length >= 2.0f * ctx->max_dt &&
        batch->min_pitch * 0.999f - 1.002f * ctx->max_pitch_jitter >= 9.5367431640625e-07f) {
        scan.scan = true;
        scan.live4 = l4ab ? 1u : 0u;                          // (the scan kernel takes any blend length)
        // three-stage workgroups for few utterances (tools/scan_split_crossover.py: up to ~1500 with four
        // live formants, half that with eight, where the filter wave is the slower stage either way)
        scan.scan_pipe = (int64_t)fam * (scan.live4 ? 1 : 2) <= scan_split_max(ctx) ? 1u : 0u;
        scan.pipe = 0u;
    }
    // which of them: a pinned grid or an explicit "time_split_min_utterances" decide as they always did; otherwise
    // the cost model does (2 s utterances: the scan kernel up to ~1 500 of them, the time-split kernels up to half the
    // machine's lanes, the lane kernels beyond; shorter utterances move the first crossover up — a chunk's warm-up
    // does not shrink with the utterance)
    bool take_split = false, take_scan = false;
    if (split.split_k && ctx->split_chunks >= 2) {
        take_split = true;
    } else if (ctx->split_min_utts >= 0) {
        take_split = split.split_k && (int64_t)fam * 6 >= ctx->split_min_utts * (l4ab ? 6 : 5);
        take_scan = !take_split && scan.scan;
    } else {
        const double c_lane = family_cost(ctx, f, fam, span);
        const double c_split = split.split_k ? family_cost(ctx, split, fam, span) : INFINITY;
        const double c_scan = scan.scan ? family_cost(ctx, scan, fam, span) : INFINITY;
        take_split = c_split <= c_scan && c_split < c_lane;
        take_scan = !take_split && c_scan < c_lane;
    }
    if (take_split) f = split;
    else if (take_scan) f = scan;
    if (f.fast == 2u && !ctx->lanes_option && ctx->split_chunks < 2) {
        // The second tier costs 0.8 of the exact one-lane kernel (0.64 - 0.8 time-split): where the exact kernels have a
        // wider mapping to fill the machine with — mid-size batches of voices that do not qualify for time-splitting —
        // they are the faster way to the same tolerance (their bits satisfy it trivially).
        Family exact;
        choose_family(ctx, batch, out_stride, fam, exact, true);
        if (family_cost(ctx, exact, fam, span) <= family_cost(ctx, f, fam, span)) f = exact;
        return;
    }
    if (take_split || take_scan) return;
    // fast arithmetic asked for, but the batch takes neither the scan kernel nor the time-split kernels (caller-built
    // elems, a voice outside their windows, an option switched off) and is small enough for the pipelined exact
    // workgroups: those are faster than the fast lane kernels there (8.1 - 11.5 against 12.4 ms), and exact bits
    // satisfy the tolerance trivially
    if (want_pipe4 || want_pipe8) {
        f.fast = 0u;
        f.live4 = batch_live4(ctx, batch) ? 1u : 0u;
        f.pipe = want_pipe4 ? pipe4_kind : pipe8_kind;
        f.L = want_pipe4 ? 4 : 8;
    }
}

// One launch: `count` launch slots from slot `slot0` of the rows [first, first + n_rows) the caller renders, with
// family f.  out_dev / out_len_dev point at row `first`.  use_perm: the batch's length-sorted slot order applies
// (whole-batch calls): slot s renders utterance perm[s], and every per-utterance array is indexed by the utterance.
static int launch_block(grail_ctx *ctx, const grail_batch *batch, const Family &f, float *out_dev, int16_t *out_pcm16_dev,
                        uint64_t out_stride, uint32_t *out_len_dev, uint32_t first, uint32_t slot0, uint32_t count,
                        bool use_perm)
{
    SynthArgs a{};
    const uint32_t row0 = use_perm ? 0u : first + slot0;      // the utterance that index 0 of the launch's arrays is
    const uint64_t out_shift = use_perm ? 0ull : (uint64_t)slot0 * out_stride;
    a.out_pcm16 = out_pcm16_dev ? out_pcm16_dev + out_shift : nullptr;
    a.segs = batch->d_segs;
    a.seg_offsets = batch->d_offsets + row0;       // the offsets themselves are absolute into segs
    a.voice_ids = batch->d_voice_ids ? batch->d_voice_ids + row0 : nullptr;
    a.seeds = batch->d_seeds ? batch->d_seeds + row0 : nullptr;
    a.perm = use_perm ? batch->d_perm + slot0 : nullptr;
    a.elems = batch->phoneme_mode ? ctx->d_voice_elems : batch->d_elems;
    a.voices = ctx->d_voices;
    a.out = out_dev ? out_dev + out_shift : nullptr;
    a.out_len = out_len_dev ? out_len_dev + (use_perm ? 0u : slot0) : nullptr;
    a.truncated = ctx->d_truncated;
    a.out_stride = out_stride;
    a.cap = out_stride;
    a.n_utt = count;
    a.n_voices = (uint32_t)ctx->voices.size();
    a.phoneme_mode = batch->phoneme_mode ? 1u : 0u;
    a.skip_silent = ctx->skip_silent_option ? 1u : 0u;
    a.half_capable = batch_half_capable(ctx, batch) ? 1u : 0u;
    a.any_blend = batch->any_blend ? 1u : 0u;
    a.live4 = f.live4;
    a.fast = f.fast;
    a.pipe = f.pipe;
    hipError_t e;
    if (f.scan) {
        a.resume = (uint32_t)ctx->scan_debug;
        a.pipe = f.scan_pipe;
        e = launch_scan(a, ctx->stream);
        ctx->last_kernel = a.live4 ? (a.pipe ? "scan_kernel<pairs=2,SPLIT,FAST>" : "scan_kernel<pairs=2,FAST>")
                                   : (a.pipe ? "scan_kernel<pairs=4,SPLIT,FAST>" : "scan_kernel<pairs=4,FAST>");
    } else {
        if (f.split_k) {
            a.split_chunks = (uint32_t)f.split_k;
            std::memcpy(a.split_bounds, f.split_bounds, sizeof a.split_bounds);
        }
        e = launch_synth(a, f.L, ctx->stream);
        ctx->last_kernel = last_kernel_name();
    }
    if (e != hipSuccess) return hip_fail(e, "synth kernel launch");
    return GRAIL_OK;
}

struct Block {
    uint32_t rows;
    Family f;
};

// Cut `rows` rows into blocks, each rendered by the family that suits ITS size, so that the time of a batch is not a
// step function of its size: a family fills the machine with a fixed number of rows (one wave per SIMD), one row more
// costs a whole further round of it — 65 537 utterances took two rounds of the one-lane kernel (81 ms) where one
// round and a pipelined workgroup launch (40.6 + 6.5 ms) do.  Candidates: the whole of it in one launch; or a full
// block of one of the families' capacities (as many rounds as fit for the one-lane kernels) followed by the best
// plan for the rest.  Exact arithmetic is mapping-invariant, so the cut never changes a bit; in fast arithmetic a row's
// samples follow the family of ITS block (include/grail_hip.h, "Determinism contract").
struct Planner {
    const grail_ctx *ctx;
    const grail_batch *batch;
    uint64_t out_stride;
    double span;
    static constexpr double LAUNCH_MS = 0.05;  // what a further launch costs by itself (measured: 0.02 - 0.06 ms)
    // (choose_family lays out time-split grids by bisection: every size is looked at once)
    std::map<uint32_t, std::pair<Family, double>> families;
    std::map<uint32_t, std::pair<double, std::vector<Block>>> plans;

    const std::pair<Family, double> &family(uint32_t rows)
    {
        auto it = families.find(rows);
        if (it != families.end()) return it->second;
        std::pair<Family, double> e;
        choose_family(ctx, batch, out_stride, rows, e.first);
        e.second = family_cost(ctx, e.first, rows, span);
        return families.emplace(rows, e).first->second;
    }
    const std::pair<double, std::vector<Block>> &plan(uint32_t rows, int depth)
    {
        auto it = plans.find(rows);
        if (it != plans.end()) return it->second;
        const std::pair<Family, double> &whole = family(rows);
        double best = whole.second;
        std::vector<Block> best_plan{Block{rows, whole.first}};
        if (depth < 4) {
            const uint64_t lanes = ctx_lanes(ctx), cus = (uint64_t)ctx->cus;
            // the capacities at which some family is exactly full (largest first: of two plans of equal cost the
            // one with the larger head wins)
            const uint64_t caps[] = {lanes, lanes / 2, lanes / 4, lanes / 8, 32 * cus, 16 * cus, 8 * cus};
            uint64_t seen = 0;
            for (const uint64_t c : caps) {
                if (c == 0 || c >= rows || c == seen) continue;
                seen = c;
                const uint32_t m = c == lanes ? (uint32_t)(rows / c) : 1u;
                const uint32_t head = (uint32_t)(m * c);
                const std::pair<Family, double> &fc = family(head);
                if (fc.second + LAUNCH_MS >= best) continue;
                const std::pair<double, std::vector<Block>> &rest = plan(rows - head, depth + 1);
                if (fc.second + LAUNCH_MS + rest.first < best) {
                    best = fc.second + LAUNCH_MS + rest.first;
                    best_plan.assign(1, Block{head, fc.first});
                    best_plan.insert(best_plan.end(), rest.second.begin(), rest.second.end());
                }
            }
        }
        return plans.emplace(rows, std::make_pair(best, best_plan)).first->second;
    }
};

static double plan_blocks(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride, uint32_t rows, double span,
                          std::vector<Block> &out)
{
    Planner p{ctx, batch, out_stride, span, {}, {}};
    const std::pair<double, std::vector<Block>> &best = p.plan(rows, 0);
    out = best.second;
    if (out.size() > 1) {
        // launch order: the block that is cheapest PER ROW first (in practice: the largest).  Two reasons.  Length-
        // sorted (ragged) batches hand out their slots longest first and a block lasts as long as its longest
        // utterance: with lengths falling by g per slot, moving a block of r rows and per-sample cost c behind one of
        // r', c' saves g (c r' - c' r) — the long utterances belong where a ROW costs least.  And a small block leaves
        // most of the machine idle for milliseconds: the large kernel behind it then starts on lowered clocks and
        // loses 2.5 - 3 ms (65 537 utterances: 50.3 ms with the single utterance first, profiles/r04_tail.txt).
        std::stable_sort(out.begin(), out.end(), [&](const Block &x, const Block &y) {
            return family_cost(ctx, x.f, x.rows, span) * (double)y.rows < family_cost(ctx, y.f, y.rows, span) * (double)x.rows;
        });
    }
    return best.first;
}

struct PlanCache {
    uint64_t key[6];
    std::vector<Block> plan;
};
namespace {
void free_plan_cache(PlanCache *p) { delete p; }
}

// Rows [first, first + count) of the batch (count = 0: all of it).  out_dev / out_len_dev point at the
// first row RENDERED, i.e. the caller has already applied the row offset to them.
// family_rows: the number of rows the kernel family is chosen for (0 = count).  A caller that renders a batch in
// row blocks passes its block size for every block, the short last one included: in fast arithmetic a row's
// samples depend on the family (lane mapping, chunk grid, scan kernel), and so they depend neither on the row's
// position nor on n_utt modulo the block size.
static int synthesize_rows(grail_ctx *ctx, const grail_batch *batch, float *out_dev,
                           int16_t *out_pcm16_dev, uint64_t out_stride, uint32_t *out_len_dev,
                           uint32_t first = 0, uint32_t count = 0, uint32_t family_rows = 0)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, batch))) return rc;
    if (batch->n_utt == 0) return GRAIL_OK;
    if (!out_dev && !out_pcm16_dev && out_stride) return fail(GRAIL_ERR_INVALID_ARG, "out_dev is NULL");
    if (first > batch->n_utt || count > batch->n_utt - first) return fail(GRAIL_ERR_INVALID_ARG, "row range");
    if (count == 0) count = batch->n_utt - first;
    if (count == 0) return GRAIL_OK;
    // the length-sorted slot assignment covers the whole batch: row-block launches keep launch order
    const bool use_perm = first == 0 && count == batch->n_utt && batch->d_perm;
    std::vector<Block> plan;
    const uint64_t key[6] = {count, out_stride, family_rows, ctx->options_epoch, ctx->voices_epoch,
                             (uint64_t)(uintptr_t)ctx ^ (batch->phoneme_mode ? 0ull : (uint64_t)(batch->elems_sharpness * 1024.0))};
    if (batch->plan_cache && std::memcmp(batch->plan_cache->key, key, sizeof key) == 0) {
        plan = batch->plan_cache->plan;
    } else {
        // one launch when the caller fixes the family (row blocks, a pinned lane mapping or chunk grid) or asks for it
        const bool single = family_rows != 0 || !ctx->composite_option || ctx->lanes_option || ctx->split_chunks >= 2;
        if (single) {
            Family f;
            choose_family(ctx, batch, out_stride, family_rows > count ? family_rows : count, f);
            plan.push_back(Block{count, f});
        } else {
            plan_blocks(ctx, batch, out_stride, count, batch_span(ctx, batch, out_stride), plan);
        }
        if (!batch->plan_cache) batch->plan_cache = new (std::nothrow) PlanCache();
        if (batch->plan_cache) {
            std::memcpy(batch->plan_cache->key, key, sizeof key);
            batch->plan_cache->plan = plan;
        }
    }
    size_t main_block = 0;                     // the block with the most rows: the one the statistics describe
    for (size_t i = 1; i < plan.size(); ++i)
        if (plan[i].rows > plan[main_block].rows) main_block = i;
    const Family f0 = plan[main_block].f;
    ctx->last_split = f0.split_k;
    ctx->last_formants = f0.live4 ? 4 : 8;
    ctx->last_lanes = f0.scan ? 0 : f0.L;
    ctx->last_pipe = f0.pipe && !f0.scan ? 1 : 0;
    ctx->last_fast = 0;
    ctx->last_blocks = (int)plan.size();
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    uint32_t slot0 = 0;
    std::string first_kernel;
    for (size_t i = 0; i < plan.size(); ++i) {
        const Block &b = plan[i];
        rc = launch_block(ctx, batch, b.f, out_dev, out_pcm16_dev, out_stride, out_len_dev, first, slot0, b.rows, use_perm);
        if (rc) return rc;
        if (i == main_block) first_kernel = ctx->last_kernel;
        if ((int)b.f.fast > ctx->last_fast) ctx->last_fast = (int)b.f.fast;
        slot0 += b.rows;
    }
    ctx->last_kernel = first_kernel;            // the largest block's instantiation names the launch
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->have_timing = true;
    return GRAIL_OK;
}

int grail_plan_blocks(uint32_t compute_units, int arithmetic, int live_formants, uint32_t warmup, uint32_t rows,
                      uint32_t span_samples, grail_plan_block *blocks, uint32_t cap, uint32_t *n_blocks)
{
    if (!n_blocks) return fail(GRAIL_ERR_INVALID_ARG, "n_blocks is NULL");
    *n_blocks = 0;
    if (compute_units == 0 || compute_units > 4096) return fail(GRAIL_ERR_INVALID_ARG, "compute_units must be 1 .. 4096");
    if (live_formants != 4 && live_formants != 8) return fail(GRAIL_ERR_INVALID_ARG, "live_formants must be 4 or 8");
    if (arithmetic != 0 && arithmetic != 1 && arithmetic != 2) return fail(GRAIL_ERR_INVALID_ARG, "arithmetic must be 0, 1 or 2");
    if (rows == 0) return GRAIL_OK;
    // a context and a batch as choose_family sees them: default options, a voice table that qualifies for every
    // family (four or eight live formants), a plain phoneme batch with power-of-two blend lengths
    grail_ctx ctx;
    ctx.cus = ctx.device_cus = (int)compute_units;
    ctx.fast_option = arithmetic;
    ctx.voices_sharpness = 0.0;
    ctx.voices_upper_silent = ctx.voices_live4_ok = live_formants == 4;
    ctx.voices_scan_ok = true;
    ctx.voices_split_ok = warmup != 0u;
    ctx.max_warmup = warmup;
    ctx.max_rate = 1.0f;                  // max_seconds below is in samples
    ctx.max_dt = 1.0f;
    grail_batch batch;
    batch.n_utt = rows;
    batch.phoneme_mode = true;
    batch.plain = true;
    batch.max_seconds = (float)span_samples;
    batch.min_length = 1e9f;
    batch.min_pitch = 0.25f;
    const uint64_t stride = ((uint64_t)span_samples + 64u + 63u) / 64u * 64u;
    std::vector<Block> plan;
    plan_blocks(&ctx, &batch, stride, rows, batch_span(&ctx, &batch, stride), plan);
    *n_blocks = (uint32_t)plan.size();
    for (uint32_t i = 0; i < plan.size() && i < cap && blocks; ++i) {
        const Family &f = plan[i].f;
        blocks[i].rows = plan[i].rows;
        blocks[i].lanes_per_utterance = f.scan ? 0u : (uint32_t)f.L;
        blocks[i].pipelined = f.scan ? 0u : f.pipe;
        blocks[i].chunks = (uint32_t)f.split_k;
        blocks[i].scan = f.scan ? (f.scan_pipe ? 2u : 1u) : 0u;
        blocks[i].fast = f.fast;
        blocks[i].formants = f.live4 ? 4u : 8u;
        blocks[i].model_ms = (float)family_cost(&ctx, f, plan[i].rows, batch_span(&ctx, &batch, stride));
    }
    return GRAIL_OK;
}

int grail_batch_synthesize_async(grail_ctx *ctx, const grail_batch *batch, float *out_dev,
                                 uint64_t out_stride, uint32_t *out_len_dev)
{
    return synthesize_rows(ctx, batch, out_dev, nullptr, out_stride, out_len_dev);
}

int grail_batch_synthesize_pcm16_async(grail_ctx *ctx, const grail_batch *batch, int16_t *out_dev,
                                       uint64_t out_stride, uint32_t *out_len_dev)
{
    return synthesize_rows(ctx, batch, nullptr, out_dev, out_stride, out_len_dev);
}

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
    s->live4 = batch_live4(ctx, batch);
    s->voices_epoch = ctx->voices_epoch;
    s->L = ctx->lanes_option ? ctx->lanes_option : auto_lanes_per_utt(batch->n_utt, ctx_simds(ctx));
    if (s->live4) {
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
    a.state = stream->d_state;
    a.state_stride = stream->lanes;
    a.resume = stream->started ? 1u : 0u;
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    hipError_t e = launch_synth(a, stream->L, ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "synth kernel launch");
    ctx->last_kernel = last_kernel_name();
    ctx->last_formants = a.live4 ? 4 : 8;
    ctx->last_lanes = stream->L;
    ctx->last_pipe = 0;
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
        const std::string keep = g_last_error;
        grail_stream_close(ctx, s);
        g_last_error = keep;
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
    auto fits = [&]() {
        for (uint32_t u = 0; u < n_utt; ++u) {
            const uint32_t add = seg_offsets[u + 1] - seg_offsets[u];
            if (add && (uint64_t)s->appended[u] - s->consumed[u] + add + 2u > cap) return false;
        }
        return true;
    };
    if (!fits()) {
        // what the host knows of the Sequencers' progress is a lower bound: ask the device
        HIP_TRY(hipMemcpyAsync(s->consumed.data(), s->d_consumed, (size_t)n_utt * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (!fits())
            return fail(GRAIL_ERR_BUFFER_TOO_SMALL, "a segment ring of the live stream is full: pull samples first (or open the "
                                                    "stream with a larger ring_segments)");
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
    e = hipMemcpyAsync(s->d_new, segs.data(), (size_t)n_new * sizeof(DevSeg), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_new_offs, seg_offsets, ((size_t)n_utt + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && elems)
        e = hipMemcpyAsync(s->d_new_elems, elems, (size_t)n_new * ELEM_FLOATS * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    // (stream order: behind every kernel that still reads the rings, ahead of every kernel that will)
    if (e == hipSuccess)
        e = launch_ring_append(s->own->d_segs, s->own->d_elems, s->d_counts, cap, s->d_new, elems ? s->d_new_elems : nullptr,
                               s->d_new_offs, n_utt, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // the host buffers are the caller's / locals
    if (e != hipSuccess) return hip_fail(e, "grail_stream_append");
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

static void pipe_destroy_opaque(void *p) { pipe_destroy((HostPipe *)p); }

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
    const std::string keep = g_last_error;
    grail_batch_free(ctx, b);
    g_last_error = keep;
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
    const std::string keep = g_last_error;
    grail_batch_free(ctx, b);
    g_last_error = keep;
    return rc;
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
    const std::string keep = g_last_error;
    grail_batch_free(ctx, b);
    g_last_error = keep;
    return rc ? rc : sync_rc;
}

int grail_say_batch(grail_ctx *ctx, const char *const *texts_utf8, uint32_t n_texts,
                    const uint32_t *voice_ids, const uint32_t *jitter_seeds, float *out,
                    uint64_t out_stride, uint32_t *out_len, uint32_t flags)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    if (ctx->voices.empty()) return fail(GRAIL_ERR_NO_VOICES, "call grail_set_voices first");
    if (n_texts && !texts_utf8) return fail(GRAIL_ERR_INVALID_ARG, "texts is NULL");
    std::vector<grail_phoneme_elem> segs;
    std::vector<uint32_t> offs(1, 0u);
    for (uint32_t i = 0; i < n_texts; ++i) {
        const uint32_t vid = voice_ids ? voice_ids[i] : 0u;
        if (vid >= ctx->voices.size()) return fail(GRAIL_ERR_INVALID_ARG, "voice id out of range");
        if (!texts_utf8[i]) return fail(GRAIL_ERR_INVALID_ARG, "a text is NULL");
        uint32_t n = 0;
        grail_text_to_phoneme_elems(&ctx->voices[vid], texts_utf8[i], nullptr, 0, &n);
        const size_t base = segs.size();
        segs.resize(base + n);
        int rc = grail_text_to_phoneme_elems(&ctx->voices[vid], texts_utf8[i], segs.data() + base, n, &n);
        if (rc) return fail(rc, "transcription failed");
        offs.push_back((uint32_t)segs.size());
    }
    return grail_synthesize_batch(ctx, segs.data(), offs.data(), voice_ids, jitter_seeds, n_texts, out,
                                  out_stride, out_len, flags);
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

int grail_comm_unique_id(uint8_t id[GRAIL_UNIQUE_ID_BYTES])
{
    if (!id) return fail(GRAIL_ERR_INVALID_ARG, "id is NULL");
    if (!rccl().ok) return fail(GRAIL_ERR_RCCL, "librccl.so could not be loaded");
    ncclUniqueId uid;
    ncclResult_t r = rccl().GetUniqueId(&uid);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    std::memcpy(id, uid.internal, GRAIL_UNIQUE_ID_BYTES);
    return GRAIL_OK;
}

int grail_comm_init(grail_ctx *ctx, const uint8_t id[GRAIL_UNIQUE_ID_BYTES], uint32_t rank,
                    uint32_t world)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!id || world == 0 || rank >= world) return fail(GRAIL_ERR_INVALID_ARG, "bad rank/world/id");
    if (!rccl().ok) return fail(GRAIL_ERR_RCCL, "librccl.so could not be loaded");
    if (ctx->comm) {
        rccl().CommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, GRAIL_UNIQUE_ID_BYTES);
    ncclResult_t r = rccl().CommInitRank(&ctx->comm, (int)world, uid, (int)rank);
    if (r != ncclSuccess) {
        ctx->comm = nullptr;
        return rccl_fail(r, "ncclCommInitRank");
    }
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return GRAIL_OK;
}

int grail_broadcast_voices(grail_ctx *ctx, uint32_t n_voices, uint32_t root)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ctx->comm) return fail(GRAIL_ERR_RCCL, "call grail_comm_init first");
    if (n_voices == 0 || root >= ctx->comm_world) return fail(GRAIL_ERR_INVALID_ARG, "bad n_voices/root");
    if (ctx->comm_rank == root && ctx->voices.size() != n_voices)
        return fail(GRAIL_ERR_INVALID_ARG, "root's voice table does not hold n_voices voices");
    const size_t bytes = (size_t)n_voices * sizeof(grail_voice);
    void *d_blob = nullptr;
    HIP_TRY(hipMalloc(&d_blob, bytes));
    hipError_t e = hipSuccess;
    if (ctx->comm_rank == root)
        e = hipMemcpyAsync(d_blob, ctx->voices.data(), bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(d_blob);
        return hip_fail(e, "voice blob upload");
    }
    // one ncclBroadcast over xGMI: root's HBM -> every rank's HBM
    ncclResult_t r = rccl().Broadcast(d_blob, d_blob, bytes, ncclUint8, (int)root, ctx->comm, ctx->stream);
    if (r != ncclSuccess) {
        (void)hipFree(d_blob);
        return rccl_fail(r, "ncclBroadcast");
    }
    std::vector<grail_voice> got(n_voices);
    e = hipMemcpyAsync(got.data(), d_blob, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_blob);
    if (e != hipSuccess) return hip_fail(e, "voice blob download");
    if (ctx->comm_rank == root) return GRAIL_OK;  // already installed
    return install_voices(ctx, got.data(), n_voices);
}

int grail_comm_info(grail_ctx *ctx, uint32_t *ranks, uint32_t *rank)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    if (ranks) *ranks = 0;
    if (rank) *rank = 0;
    if (!ctx->comm) return GRAIL_OK;              // no communicator: 0 ranks
    if (!rccl().ok || !rccl().CommCount || !rccl().CommUserRank)
        return fail(GRAIL_ERR_RCCL, "librccl.so lacks ncclCommCount / ncclCommUserRank");
    int n = 0, r = 0;
    ncclResult_t e = rccl().CommCount(ctx->comm, &n);
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommCount");
    e = rccl().CommUserRank(ctx->comm, &r);
    if (e != ncclSuccess) return rccl_fail(e, "ncclCommUserRank");
    if (ranks) *ranks = (uint32_t)n;
    if (rank) *rank = (uint32_t)r;
    return GRAIL_OK;
}

int grail_comm_destroy(grail_ctx *ctx)
{
    if (!ctx) return fail(GRAIL_ERR_INVALID_ARG, "ctx is NULL");
    if (ctx->comm && rccl().ok) rccl().CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    return GRAIL_OK;
}

}  // extern "C"
