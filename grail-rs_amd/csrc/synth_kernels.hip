// synth_kernels.hip — dispatch of the fused Selector -> Sequencer -> Jitter -> Synthesize kernel (synth_kernel.h)
// to its instantiation units (synth_inst_*.hip), the Sequencer-clock pre-pass, and the launch geometry.
#include <cstdio>

#include "device_common.h"
#include "kernels.h"
#include "synth_launch.h"

namespace grail {

namespace {

// Sequencer clock only (src/lib.rs:861-888, :930): how many elems the
// Sequencer yields, one utterance per lane.
__global__ __launch_bounds__(64) void lengths_kernel(const LenArgs A)
{
    const uint32_t u = blockIdx.x * 64u + threadIdx.x;
    if (u >= A.n_utt) return;
    uint32_t vid = A.voice_ids ? A.voice_ids[u] : 0u;
    if (vid >= A.n_voices) vid = 0u;
    const float dt = 1.0f / A.voices[vid].sample_rate;
    uint32_t pos = A.seg_offsets[u];
    const uint32_t end = A.seg_offsets[u + 1];
    bool cur_some = false, nxt_some = false;
    float cur_len = 0.0f, nxt_len = 0.0f;
    float clk = 0.0f;
    uint32_t n = 0;
    while (n < A.max_len) {
        clk -= dt;
        if (clk < 0.0f) {
            if (cur_some && nxt_some) {
                cur_len = nxt_len;
                nxt_some = pos < end;
                if (nxt_some) nxt_len = A.segs[pos++].length;
                clk += cur_len;
            } else if (!cur_some && !nxt_some) {
                cur_some = pos < end;
                if (cur_some) cur_len = A.segs[pos++].length;
                nxt_some = pos < end;
                if (nxt_some) nxt_len = A.segs[pos++].length;
                if (cur_some) clk += cur_len;
            } else {
                break;
            }
        }
        if (!cur_some) break;
        ++n;
    }
    A.out_len[u] = n;
}

}  // namespace

uint32_t state_words(int L)
{
    const int fpl = NF / L;
    return 22u + 7u * (uint32_t)fpl;   // visit_state: 22 scalars + 7 values per formant
}

static void geometry(int L, uint32_t &per_block, uint32_t &threads)
{
    // must mirror the <L, T, WAVES, MINW> table of the instantiation units
    const int waves = L >= 4 ? 4 : 1;
    per_block = (64u / (uint32_t)L) * (uint32_t)waves;
    threads = 64u * (uint32_t)waves;
}

uint64_t state_lanes(uint32_t n_utt, int L)
{
    uint32_t per_block, threads;
    geometry(L, per_block, threads);
    const uint64_t blocks = (n_utt + per_block - 1) / per_block;
    return blocks * threads;
}

// what the last launch_synth call started, for the bench line and the profile bookkeeping
thread_local char g_kernel_name[96] = "none";
const char *last_kernel_name() { return g_kernel_name; }

hipError_t launch_synth(const SynthArgs &args, int L, hipStream_t stream)
{
    if (args.n_utt == 0) return hipSuccess;
    if (args.pipe && !args.fast) {
        if (args.live4) launch_pipe4(args, stream);
        else launch_pipe8(args, stream);
        return hipGetLastError();
    }
    if (args.split_chunks) {
        if (!args.fast || args.state || L != 1 || args.split_chunks > (uint32_t)SPLIT_MAX_CHUNKS) return hipErrorInvalidValue;
        if (args.fast == 2u) launch_split_mid(args, stream);
        else launch_split(args, stream);
        return hipGetLastError();
    }
    if (args.fast == 2u && L != 1) return hipErrorInvalidValue;   // MID: one lane per utterance
    switch (L) {
    case 1: args.fast == 2u ? launch_mid_l1(args, stream) : args.fast ? launch_fast_l1(args, stream) : launch_exact_l1(args, stream); break;
    case 2: args.fast ? launch_fast_l2(args, stream) : launch_exact_l2(args, stream); break;
    case 4: args.fast ? launch_fast_l4(args, stream) : launch_exact_l4(args, stream); break;
    case 8: args.fast ? launch_fast_l8(args, stream) : launch_exact_l8(args, stream); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_lengths(const LenArgs &args, hipStream_t stream)
{
    if (args.n_utt == 0) return hipSuccess;
    const dim3 grid((args.n_utt + 63) / 64), block(64);
    hipLaunchKernelGGL(lengths_kernel, grid, block, 0, stream, args);
    return hipGetLastError();
}

}  // namespace grail
