// api_internal.hpp — what the host translation units of the library share (internal; host code only: the kernel
// units include kernels.h, never this).  grail_api.cpp: contexts, options, batches, voices; voice_analysis.cpp: what a
// voice table qualifies for; launch_plan.cpp: kernel families, cost model, block planner; synthesize.cpp: launches;
// streams.cpp: resumable and live streams; host_output.cpp: the one-call forms with a host destination; comm.cpp: RCCL.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
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
#include <unordered_map>
#include <vector>

#include "../../include/grail_hip.h"
#include "kernels.h"

// (the opaque types of the C ABI are global; everything else the units share lives in grail::host)
struct PlanCache;                     // synthesize.cpp

// words of the device block behind grail_ctx::d_truncated: the flag and three statistics counters; debug builds
// (-DGRAIL_FAST_PROF) keep 32 u64 profile counters behind word 8
#ifdef GRAIL_FAST_PROF
constexpr size_t TRUNCATED_WORDS = 8 + 64;
#else
constexpr size_t TRUNCATED_WORDS = 4;
#endif

struct grail_ctx {
    int device = 0;
    int cus = 256;                    // compute units the launch policy plans for (hipDeviceProp_t::multiProcessorCount;
                                      // "assume_compute_units" overrides it): every capacity of the policy is a multiple
    int device_cus = 256;             // ... what the device reported
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool have_timing = false;
    std::vector<grail_voice> voices;  // host copy of the table
    grail::DevVoice *d_voices = nullptr;
    float *d_voice_elems = nullptr;   // [n_voices * NUM_VOICED][49]
    // what each voice of the table qualifies for (a batch is judged by the voices IT names: used_voices); the table-wide
    // flags below are what a context without per-voice records (grail_plan_blocks) goes by
    struct VoiceInfo {
        bool upper_silent = false, live4_ok = false, scan_ok = false, split_ok = false;
        uint32_t warmup = 0;
    };
    std::vector<VoiceInfo> voice_info;
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
    int ragged_option = 1;            // length-sorted batches: lane mappings weighed by the rows' lengths (ragged_plan)
    int pipe_spread = 1;              // pipelined workgroups on rows that differ in length: few utterances per workgroup (pipe_fill_for)
    int two_waves_option = 1;         // tolerance-mode lane kernels on 2 / 4 / 8 lanes: two waves per SIMD where a launch has more waves than SIMDs
    int packed_option = 1;            // launches of more one-wave-per-SIMD workgroups than the device has room for: launch slots in packed order
    int composite_option = 1;         // a batch may be cut into blocks with a kernel family each (plan_blocks)
    int row_groups_option = 1;        // rows the lean families cannot take are planned apart: 1 where the cost model says so, 2 always, 0 never
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
    int last_packed = 0;              // ... blocks of it launched in packed order
    uint32_t *d_truncated = nullptr;  // [0] truncation flag, [1] slow-path wave-steps, [2] fast wave-tiles, [3] general wave-steps
    uint64_t slow_steps = 0;          // of the kernels synced so far
    uint64_t fast_tiles = 0, general_steps = 0;
    uint32_t seen_counters[4] = {0, 0, 0, 0};   // d_truncated[1..3] as last read: the device counters only ever grow
    int lanes_option = 0;             // 0 = auto
    int skip_silent_option = 1;       // skip band-pass filters of provably silent formants
    int pipeline_option = 1;          // small qualifying batches: producer/consumer workgroups
    int pipe_round32 = 1;             // ... with rounds of 32 samples while one workgroup per CU suffices (8.20 -> 7.86 ms for config 2)
    uint64_t voices_epoch = 0;        // set by every install_voices: unique in the process, not per context
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
    grail::DevSeg *d_new = nullptr;
    float *d_new_elems = nullptr;
    uint32_t *d_new_offs = nullptr;
    size_t new_cap = 0;
    // ... on the host side two pinned buffers in turn, each with the event behind its last upload: an append returns as
    // soon as its copies and its scatter kernel are queued, and waits only for the append before last (not for every
    // kernel queued on the stream) before it writes into a buffer again
    void *h_stage[2] = {nullptr, nullptr};
    size_t h_stage_cap[2] = {0, 0};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    bool stage_busy[2] = {false, false};
    int stage_next = 0;
};

// A block of a ragged batch in PACKED launch order (launch_plan.cpp, "The workgroup dispatcher"): the slot -> utterance table
// of rows [slot0, slot0 + rows) with its workgroups re-ordered, on the device.  Kept by the batch (never overwritten: a
// kernel of another context may still be reading it), freed with it.
struct PackedPerm {
    const void *view = nullptr;       // the (view of the) batch the block belongs to
    uint32_t slot0 = 0, rows = 0, per_block = 0, cus = 0;
    uint32_t family = 0;              // L | fast << 8 | live4 << 16: what the workgroups' costs were priced for
    uint32_t *d_perm = nullptr;       // [rows]; nullptr: the plain order is as good (remembered, so that it is not packed again)
    double model_ms = 0.0, plain_ms = 0.0;
};

struct grail_batch {
    grail::DevSeg *d_segs = nullptr;
    uint32_t *d_offsets = nullptr;
    uint32_t *d_voice_ids = nullptr;
    uint32_t *d_seeds = nullptr;
    uint32_t *d_perm = nullptr;   // ragged batches: launch slot -> utterance, longest first
    std::vector<uint32_t> perm_host;          // ... its host copy (the root batch only; packed launch orders are cut from it)
    mutable std::vector<PackedPerm> packed;   // ... blocks of it in packed launch order, made at their first launch
    uint32_t *d_len_bound = nullptr;   // per utterance: an upper bound of its length in samples (plain batches; time-split kernels)
    uint64_t len_bound_epoch = 0;      // ... for the voice table of this epoch (its highest sample rate); epochs are unique
                                       // in the process, so a context other than the uploader never matches
    bool len_bound_known = false;      // (the planner's question; grail_plan_ragged_blocks answers it without a device)
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
    uint32_t elems_warmup = 0;      // elem mode: warm-up length of the time-split kernels over the batch's distinct elems and the
                                    // jitter of the voices it names (elems_warmup()); 0: the batch does not qualify
    uint64_t elems_warmup_epoch = 0;   // ... computed against this voice table (ctx->voices_epoch)
    bool elems_live4_ok = false;    // elem mode: formants 5-8 of every elem (and the voices named) can be left out (live4_ok); same epoch
    bool elems_scan_ok = false;     // elem mode: every elem inside the scan kernel's window (scan_elems_ok), pitches <= 1/2; same epoch
    std::vector<uint32_t> used_voices;   // the distinct voice ids of the batch, ascending
    // the launch plan of the last synthesis call of this batch (plan_blocks lays out time-split grids by bisection: a
    // fraction of a millisecond of host time, which a one-millisecond kernel should not pay at every launch)
    mutable PlanCache *plan_cache = nullptr;
    // Row groups (whole-batch launches of length-sorted batches).  A few rows that the lean kernel families cannot take —
    // a segment shorter than two samples, a non-finite length or pitch — would cost the whole batch its four-formant
    // kernels, pipelined workgroups and fast families, because those are gated on the batch's worst row.  Such rows are
    // put LAST in the slot order and the batch is planned as two batches that share the device buffers: groups[0] = the
    // lean rows (slots [0, groups[0].n_utt)), groups[1] = the rest.  Valid for the voice table they were judged against.
    std::vector<grail_batch> groups;
    uint64_t groups_epoch = 0;
    // Ragged (length-sorted) batches, per granule of 8 consecutive launch slots: the longest row in samples (at the
    // context's highest rate), the rows' segments and their kinks of alpha (blend_length < length) — what ragged_plan()
    // weighs the lane mappings with.  Empty for aligned batches.
    std::vector<float> granule_samples;
    std::vector<uint32_t> granule_segs, granule_kinks;
};

namespace grail {
namespace host {

// errors: the status is returned, the message kept per thread for grail_last_error()
int fail(int status, const std::string &msg);
int hip_fail(hipError_t e, const char *what);
std::string &last_error();

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return ::grail::host::hip_fail(e_, #expr); \
    } while (0)

// the SIMDs and lanes the policy plans for: 4 SIMDs per compute unit, 64 lanes per wavefront.  Every family is laid out
// for ONE resident wave per SIMD (a second wave on a SIMD costs as much as it brings: profiles/r01_lanes_sweep.txt), so
// all capacities below are multiples of the compute-unit count hipGetDeviceProperties reports (a partitioned MI355X —
// CPX, 32 CUs — plans for 32, not 256); "assume_compute_units" overrides it for tests.
inline uint64_t ctx_simds(const grail_ctx *ctx) { return 4ull * (uint64_t)ctx->cus; }
inline uint64_t ctx_lanes(const grail_ctx *ctx) { return 256ull * (uint64_t)ctx->cus; }
inline int64_t pipe4_groups(const grail_ctx *ctx) { return ctx->pipe4_max_groups < 0 ? 2 * (int64_t)ctx->cus : ctx->pipe4_max_groups; }
inline int64_t pipe8_groups(const grail_ctx *ctx) { return ctx->pipe8_max_groups < 0 ? 2 * (int64_t)ctx->cus : ctx->pipe8_max_groups; }
inline int64_t scan_max_utts(const grail_ctx *ctx) { return ctx->scan_max_utts < 0 ? 34 * (int64_t)ctx->cus : ctx->scan_max_utts; }
inline int64_t scan_split_max(const grail_ctx *ctx) { return ctx->scan_split_max < 0 ? 6 * (int64_t)ctx->cus : ctx->scan_split_max; }

int bind(grail_ctx *ctx);

template <typename Tp>
int upload(Tp **dst, const void *src, size_t count, hipStream_t stream)
{
    *dst = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void **)dst, count * sizeof(Tp)));
    if (src) HIP_TRY(hipMemcpyAsync(*dst, src, count * sizeof(Tp), hipMemcpyHostToDevice, stream));
    return GRAIL_OK;
}

// grail_api.cpp
void free_batch_buffers(grail_batch *b);
bool blend_is_pow2(float blend_length);
int check_offsets(const uint32_t *seg_offsets, uint32_t n_utt, uint32_t *n_segs);
int install_voices(grail_ctx *ctx, const grail_voice *voices, uint32_t n_voices);
int check_ready(grail_ctx *ctx, const grail_batch *batch);

// voice_analysis.cpp: what a voice qualifies for (four-formant kernels, scan kernel, time-split warm-up), the predicted
// deviation of fast arithmetic (sharpness), the tier a batch is served in, the chunk grid of a time-split launch
bool live4_ok(const grail_voice &v);
bool live4_elems_ok(const grail_synthesis_elem *elems, size_t n_elems, float jitter_delta_formant_frequency);
bool scan_voice_ok(const grail_voice &v);
bool scan_elems_ok(const grail_synthesis_elem *elems, size_t n_elems, float jitter_delta_formant_frequency);
uint32_t voice_warmup(const grail_voice &v);
uint32_t elems_warmup(const grail_synthesis_elem *elems, size_t n_elems, double jitter_delta_formant_frequency);
double elems_sharpness(const grail_synthesis_elem *elems, size_t n);
double batch_sharpness(const grail_ctx *ctx, const grail_batch *batch);
int fast_tier_for(const grail_ctx *ctx, const grail_batch *batch, int arithmetic);
int fast_tier(const grail_ctx *ctx, const grail_batch *batch);
bool split_grid(uint32_t span, uint32_t warmup, int K, double r, uint32_t *b);

// launch_plan.cpp: which kernel family renders a block of rows, what it costs, how a batch is cut into blocks
struct Family {
    int L = 1;                 // lanes per utterance (lane kernels, pipelined workgroups)
    uint32_t pipe = 0;         // exact pipelined workgroups: 1 = rounds of 16 samples, 2 = rounds of 32
    uint32_t live4 = 0;        // formants 5-8 not laid out
    bool half = false;         // one lane per utterance, exact, eight formants laid out but 5-8 silent: the half-live loops
    uint32_t fast = 0;         // tolerance arithmetic
    int split_k = 0;           // time-split kernels: chunks per utterance (0: not time-split)
    uint32_t split_bounds[SPLIT_MAX_CHUNKS + 1] = {};
    uint32_t split_active = 0; // ... rows that differ in length: how many (wave, chunk) pairs have anything to render (0: all)
    bool scan = false;         // the time-parallel scan kernel
    uint32_t scan_pipe = 0;    // ... its three-stage flavour
};

struct Block {
    uint32_t rows;
    Family f;
};

// lanes per utterance when the option is 0 (auto): the widest mapping that gives each of `simds` SIMDs at most one wave
int auto_lanes_per_utt(uint32_t n_utt, uint64_t simds);
double batch_span(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride);
double family_cost(const grail_ctx *ctx, const Family &f, uint32_t rows, double span);
// pipelined workgroups, rows that differ in length: utterances per workgroup of a launch of `rows` rows (0: every slot)
uint32_t pipe_fill_for(const grail_ctx *ctx, const grail_batch *batch, const Family &f, uint32_t rows);
// a launch of `rows` rows with family f takes the instantiations built for two waves per SIMD (SynthArgs::cohabit)
bool family_cohabits(const grail_ctx *ctx, const Family &f, uint32_t rows);
bool batch_half_capable(const grail_ctx *ctx, const grail_batch *batch);
bool batch_live4_any_blend(const grail_ctx *ctx, const grail_batch *batch);
bool batch_live4(const grail_ctx *ctx, const grail_batch *batch);
void choose_family(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride, uint32_t fam, Family &f,
                   bool exact_only = false, int pin_lanes = 0);
// ragged batches: what a block costs given the lengths and events of ITS rows; the whole-batch plan weighed against
// one launch of each lane mapping with as many rounds as it takes (launch_plan.cpp)
double ragged_cost(const grail_ctx *ctx, const grail_batch *batch, const Family &f, uint32_t slot0, uint32_t rows, double span);
// The launch order of the workgroups of such a block (one wave per SIMD, more workgroups than the device holds at once):
// true and order[position] = workgroup (in the plain, longest-first numbering) when the packed order is worth it by the
// dispatcher's model; *plain_ms / *packed_ms: the model's makespans.  rows_per_block: what a workgroup renders.
bool packed_launch_order(const grail_ctx *ctx, const grail_batch *batch, const Family &f, uint32_t slot0, uint32_t rows, double span,
                         std::vector<uint32_t> *order, uint32_t *rows_per_block, double *plain_ms, double *packed_ms);
void ragged_plan(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride, uint32_t rows, std::vector<Block> &plan);
double plan_blocks(const grail_ctx *ctx, const grail_batch *batch, uint64_t out_stride, uint32_t rows, double span,
                   std::vector<Block> &out, bool exact_only = false);

// synthesize.cpp: one launch per block; a batch's cached plan
void free_plan_cache(PlanCache *p);
int synthesize_rows(grail_ctx *ctx, const grail_batch *batch, float *out_dev, int16_t *out_pcm16_dev, uint64_t out_stride,
                    uint32_t *out_len_dev, uint32_t first = 0, uint32_t count = 0, uint32_t family_rows = 0);

// host_output.cpp / comm.cpp: what grail_destroy releases
void pipe_destroy_opaque(void *p);
void comm_release(grail_ctx *ctx);
// host_output.cpp: texts -> PhonemeElems (grail_say_batch, grail_node_say_batch)
int say_segments(const std::vector<grail_voice> &voices, const char *const *texts_utf8, uint32_t n_texts,
                 const uint32_t *voice_ids, std::vector<grail_phoneme_elem> &segs, std::vector<uint32_t> &offs);
// comm.cpp: one communicator over the contexts of a node, formed inside the process (ncclCommInitAll)
int comm_init_all(grail_ctx *const *ctxs, const int *devices, uint32_t n);

}  // namespace host
}  // namespace grail
