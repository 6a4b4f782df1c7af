// synthesize.cpp — a batch's rows to kernel launches: the cached block plan of a batch, one launch per block
// (launch_plan.cpp chose the families), the device-resident entry points.
#include <mutex>

#include "api_internal.hpp"

using namespace grail;
using namespace grail::host;

// the launch plan of the last synthesis call of a batch (grail_batch::plan_cache), with what it was made for
struct PlanCache {
    uint64_t key[6];
    std::vector<Block> plan;
};

namespace grail {
namespace host {

// One launch: `count` launch slots from slot `slot0` of the rows [first, first + n_rows) the caller renders, with
// family f.  out_dev / out_len_dev point at row `first`.  use_perm: the batch's length-sorted slot order applies
// (whole-batch calls): slot s renders utterance perm[s], and every per-utterance array is indexed by the utterance.
// The block's slot -> utterance table in PACKED launch order (launch_plan.cpp, "The workgroup dispatcher"), made at the
// block's first launch and kept by the root batch; nullptr: the plain order (it is as good, or the block does not qualify).
static const uint32_t *packed_perm_of(grail_ctx *ctx, const grail_batch *root, const grail_batch *view, const Family &f,
                                      uint64_t out_stride, uint32_t slot0, uint32_t count, int *rc)
{
    *rc = GRAIL_OK;
    if (!ctx->packed_option || root->perm_host.size() < (size_t)slot0 + count) return nullptr;
    const uint32_t family = (uint32_t)f.L | f.fast << 8 | f.live4 << 16;
    static std::mutex lock;                    // (one batch may be rendered by several contexts, each on a thread of its own)
    std::lock_guard<std::mutex> hold(lock);
    for (const PackedPerm &pp : root->packed)
        if (pp.view == view && pp.slot0 == slot0 && pp.rows == count && pp.family == family && pp.cus == (uint32_t)ctx->cus)
            return pp.d_perm;
    if (root->packed.size() >= 8) return nullptr;      // (a batch launched under ever new plans: stop collecting tables)
    PackedPerm pp;
    pp.view = view;
    pp.slot0 = slot0;
    pp.rows = count;
    pp.family = family;
    pp.cus = (uint32_t)ctx->cus;
    std::vector<uint32_t> order;
    // (a view's granules count from ITS first slot: the second row group's slots start behind the first's)
    const uint32_t view_slot0 = view == root || root->groups.size() != 2 || view == &root->groups[0] ? slot0 : slot0 - root->groups[0].n_utt;
    if (packed_launch_order(ctx, view, f, view_slot0, count, batch_span(ctx, view, out_stride), &order, &pp.per_block, &pp.plain_ms,
                            &pp.model_ms)) {
        std::vector<uint32_t> perm(count);
        const uint32_t *src = root->perm_host.data() + slot0;
        size_t at = 0;
        for (const uint32_t b : order) {
            const size_t lo = (size_t)b * pp.per_block, hi = std::min<size_t>(lo + pp.per_block, count);
            for (size_t i = lo; i < hi; ++i) perm[at++] = src[i];
        }
        hipError_t e = at == count ? hipMalloc((void **)&pp.d_perm, (size_t)count * sizeof(uint32_t)) : hipErrorInvalidValue;
        if (e == hipSuccess) e = hipMemcpyAsync(pp.d_perm, perm.data(), (size_t)count * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // (`perm` is a local; other contexts may use the table next)
        if (e != hipSuccess) {
            if (pp.d_perm) (void)hipFree(pp.d_perm);
            *rc = hip_fail(e, "packed launch order");
            return nullptr;
        }
    }
    root->packed.push_back(pp);
    return pp.d_perm;
}

static int launch_block(grail_ctx *ctx, const grail_batch *root, const grail_batch *batch, const Family &f, float *out_dev,
                        int16_t *out_pcm16_dev, uint64_t out_stride, uint32_t *out_len_dev, uint32_t first, uint32_t slot0,
                        uint32_t count, bool use_perm)
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
    if (use_perm) {
        int rc = GRAIL_OK;
        const uint32_t *packed = packed_perm_of(ctx, root, batch, f, out_stride, slot0, count, &rc);
        if (rc) return rc;
        if (packed) {
            a.perm = packed;
            ++ctx->last_packed;
        }
    }
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
    a.cohabit = family_cohabits(ctx, f, count) ? 1u : 0u;
    a.len_bound = batch->d_len_bound && batch->len_bound_epoch == ctx->voices_epoch ? batch->d_len_bound + row0 : nullptr;
    a.pipe_fill = pipe_fill_for(ctx, batch, f, count);
    a.fold_from = 0u;
    if (a.cohabit) {
        // at most two rounds of the device: the workgroups of the second take their launch slots in reverse order
        const uint32_t waves_per_block = f.L >= 4 ? 4u : 1u, per_block = (64u / (uint32_t)f.L) * waves_per_block;
        const uint32_t blocks = (count + per_block - 1u) / per_block, round = (uint32_t)ctx_simds(ctx) / waves_per_block;
        if (blocks > round && blocks <= 2u * round) a.fold_from = round;
    }
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
            a.split_warmup = batch->phoneme_mode ? 0u : batch->elems_warmup;
            std::memcpy(a.split_bounds, f.split_bounds, sizeof a.split_bounds);
        }
        e = launch_synth(a, f.L, ctx->stream);
        ctx->last_kernel = last_kernel_name();
    }
    if (e != hipSuccess) return hip_fail(e, "synth kernel launch");
    return GRAIL_OK;
}

void free_plan_cache(PlanCache *p) { delete p; }

// Rows [first, first + count) of the batch (count = 0: all of it).  out_dev / out_len_dev point at the
// first row RENDERED, i.e. the caller has already applied the row offset to them.
// family_rows: the number of rows the kernel family is chosen for (0 = count).  A caller that renders a batch in
// row blocks passes its block size for every block, the short last one included: in fast arithmetic a row's
// samples depend on the family (lane mapping, chunk grid, scan kernel), and so they depend neither on the row's
// position nor on n_utt modulo the block size.
int synthesize_rows(grail_ctx *ctx, const grail_batch *batch, float *out_dev,
                           int16_t *out_pcm16_dev, uint64_t out_stride, uint32_t *out_len_dev,
                           uint32_t first, uint32_t count, uint32_t family_rows)
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
    // the launch plan of `rows` rows of a (view of the) batch, cached with what it was made for
    auto plan_of = [&](const grail_batch *view, const uint32_t rows) {
        std::vector<Block> plan;
        const uint64_t key[6] = {rows, out_stride, family_rows, ctx->options_epoch, ctx->voices_epoch,
                                 (uint64_t)(uintptr_t)ctx ^ (view->phoneme_mode ? 0ull : (uint64_t)(view->elems_sharpness * 1024.0))};
        // (one batch may be rendered by several contexts, each on a thread of its own: the cache is read and replaced
        // under a lock — the plans are a few dozen bytes, the lock is held for a copy)
        static std::mutex cache_lock;
        {
            std::lock_guard<std::mutex> hold(cache_lock);
            if (view->plan_cache && std::memcmp(view->plan_cache->key, key, sizeof key) == 0) return view->plan_cache->plan;
        }
        // one launch when the caller fixes the family (row blocks, a pinned lane mapping or chunk grid) or asks for it
        const bool single = family_rows != 0 || !ctx->composite_option || ctx->lanes_option || ctx->split_chunks >= 2;
        if (single) {
            Family f;
            choose_family(ctx, view, out_stride, family_rows > rows ? family_rows : rows, f);
            plan.push_back(Block{rows, f});
        } else {
            plan_blocks(ctx, view, out_stride, rows, batch_span(ctx, view, out_stride), plan);
            // length-sorted batches: the plan weighed against each lane mapping in as many rounds as it takes, by the
            // lengths and events of the rows (launch_plan.cpp, "Ragged batches")
            // (the batch itself, or the first of its row groups: those rows hold launch slots 0 .. rows - 1 of the sorted order)
            if (use_perm && (view == batch || (batch->groups.size() == 2 && view == &batch->groups[0])))
                ragged_plan(ctx, view, out_stride, rows, plan);
        }
        {
            std::lock_guard<std::mutex> hold(cache_lock);
            if (!view->plan_cache) view->plan_cache = new (std::nothrow) PlanCache();
            if (view->plan_cache) {
                std::memcpy(view->plan_cache->key, key, sizeof key);
                view->plan_cache->plan = plan;
            }
        }
        return plan;
    };
    // Row groups (grail_batch::groups): the rows the lean families cannot take sit last in the slot order and are planned
    // as a batch of their own, so that a few of them do not decide the kernels of all.  Whole-batch launches only, and
    // only while the voice table is the one the rows were judged against.
    struct Part {
        const grail_batch *view;
        Block block;
    };
    std::vector<Part> plan;
    for (const Block &b : plan_of(batch, count)) plan.push_back(Part{batch, b});
    if (use_perm && family_rows == 0 && ctx->row_groups_option && batch->groups.size() == 2 && batch->groups_epoch == ctx->voices_epoch) {
        // ... where that is cheaper by the cost model: a separate launch for four odd rows behind a full round of the
        // one-lane kernel costs more than it saves (53.7 against 46.7 ms), behind 20 000 rows it does not
        // (by the rows' own lengths and events where the view has them — a ragged corpus: ragged_cost falls back to the
        // one-round price of the family otherwise)
        auto cost_of = [&](const std::vector<Part> &parts) {
            double c = 0.0;
            uint32_t at = 0;
            const grail_batch *of = nullptr;
            for (const Part &p : parts) {
                if (p.view != of) at = 0;        // (a view's blocks follow each other from its first slot)
                of = p.view;
                c += ragged_cost(ctx, p.view, p.block.f, at, p.block.rows, batch_span(ctx, p.view, out_stride)) + 0.05;
                at += p.block.rows;
            }
            return c;
        };
        std::vector<Part> grouped;
        for (const grail_batch &g : batch->groups)
            for (const Block &b : plan_of(&g, g.n_utt)) grouped.push_back(Part{&g, b});
        if (ctx->row_groups_option == 2 || cost_of(grouped) < cost_of(plan)) plan.swap(grouped);
    }
    size_t main_block = 0;                     // the block with the most rows: the one the statistics describe
    for (size_t i = 1; i < plan.size(); ++i)
        if (plan[i].block.rows > plan[main_block].block.rows) main_block = i;
    const Family f0 = plan[main_block].block.f;
    ctx->last_split = f0.split_k;
    ctx->last_formants = f0.live4 ? 4 : 8;
    ctx->last_lanes = f0.scan ? 0 : f0.L;
    ctx->last_pipe = f0.pipe && !f0.scan ? 1 : 0;
    ctx->last_packed = 0;
    ctx->last_fast = 0;
    ctx->last_blocks = (int)plan.size();
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    uint32_t slot0 = 0;
    std::string first_kernel;
    for (size_t i = 0; i < plan.size(); ++i) {
        const Block &b = plan[i].block;
        rc = launch_block(ctx, batch, plan[i].view, b.f, out_dev, out_pcm16_dev, out_stride, out_len_dev, first, slot0, b.rows, use_perm);
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


}  // namespace host
}  // namespace grail

extern "C" {

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

}  // extern "C"
