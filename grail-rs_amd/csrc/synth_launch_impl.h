// synth_launch_impl.h — which instantiation of synth_kernel a launch takes (internal; included by the
// synth_inst_*.hip units only).
#pragma once

#include "synth_kernel.h"
#include "synth_launch.h"

namespace grail {
namespace {

template <int L, int WAVES>
dim3 lane_grid(const SynthArgs &args)
{
    const uint32_t per_block = (64u / L) * WAVES;
    return dim3((args.n_utt + per_block - 1) / per_block);
}

template <int L, int T, int WAVES, int MINW>
void launch_one_exact(const SynthArgs &args, hipStream_t stream)
{
    const dim3 grid = lane_grid<L, WAVES>(args), block(64 * WAVES);
    // one-shot launches of more waves than the device has SIMDs (args.cohabit): the instantiations built for two waves per
    // SIMD — two lanes per utterance with four formants laid out, four lanes with either layout; the same operations in the
    // same order in 256 registers.  (Two lanes with eight formants would spill, eight lanes are never asked for so many waves.)
    if constexpr ((L == 2 || L == 4) && MINW == 1) {
        if (args.cohabit && !args.state && (L == 4 || args.live4)) {
            launch_one_exact<L, T, WAVES, 2>(args, stream);
            return;
        }
    }
    if constexpr (L <= 4 && MINW == 1) {
        if (args.state && args.live4) {
            // the lean resumable instantiations: four formants laid out, any blend length
            if (args.any_blend) start<L, T, WAVES, MINW, true, false, true, 4>(args, grid, block, stream);
            else start<L, T, WAVES, MINW, true, false, false, 4>(args, grid, block, stream);
            return;
        }
    }
    if constexpr (L <= 4) {
        if (!args.state && args.live4) {
            // L = 4 parks 4 floats per sample instead of 8: room for the 64-step tiles of L = 8
            // (any blend length: the four-formant layout does not depend on how alpha is divided out)
            constexpr int T4 = L == 4 ? 64 : T;
            if (args.any_blend) start<L, T4, WAVES, MINW, false, false, true, 4>(args, grid, block, stream);
            else start<L, T4, WAVES, MINW, false, false, false, 4>(args, grid, block, stream);
            return;
        }
    }
    if constexpr (MINW == 1) {
        if (args.state) {
            // resumable streams: the lean instantiation when the batch allows it (chosen when the stream is
            // opened: the state layout follows the formant layout), the general one otherwise
            if (!args.any_blend && !args.half_capable)
                start<L, T, WAVES, MINW, true, false, false>(args, grid, block, stream);
            else
                start<L, T, WAVES, MINW, true, true, true>(args, grid, block, stream);
            return;
        }
    }
    if constexpr (MINW == 1 || L == 4) {
        if (args.any_blend)
            start<L, T, WAVES, MINW, false, true, true>(args, grid, block, stream);
        else if (L == 1 && args.half_capable)
            start<L, T, WAVES, MINW, false, true, false>(args, grid, block, stream);
        else
            start<L, T, WAVES, MINW, false, false, false>(args, grid, block, stream);
    }
}

template <int L, int T, int WAVES, int MINW>
void launch_one_fast(const SynthArgs &args, hipStream_t stream)
{
    const dim3 grid = lane_grid<L, WAVES>(args), block(64 * WAVES);
    // one-shot launches of more waves than the device has SIMDs (args.cohabit): the instantiations built for two waves per
    // SIMD — the same operations in the same order, 256 registers instead of 384.  (Two lanes per utterance with eight
    // formants laid out would spill: those stay as they are.)
    if constexpr (L >= 2 && MINW == 1) {
        if (args.cohabit && !args.state && (L > 2 || args.live4)) {
            launch_one_fast<L, T, WAVES, 2>(args, stream);
            return;
        }
    }
    // one-shot fast kernels at L = 1: 64-step tiles (half as many flushes, calm tests and coefficient end
    // points per sample: 18.1 -> 17.1 ms on the headline batch; the exact kernel measures slower with them,
    // 43.1 against 40.4 ms, same box)
    constexpr int TF = L == 1 ? 64 : T;
    if constexpr (L <= 4) {
        if constexpr (MINW == 1) {       // (resumable kernels: one wave per SIMD only)
            if (args.state && args.live4) {
                if (args.any_blend) start<L, T, WAVES, MINW, true, false, true, 4, false, true>(args, grid, block, stream);
                else start<L, T, WAVES, MINW, true, false, false, 4, false, true>(args, grid, block, stream);
                return;
            }
        }
        if (!args.state && args.live4) {
            // four live formants (any blend length: the host sets live4 for those too in fast mode).  L = 4
            // parks 4 floats per sample instead of 8: room for the 64-step tiles of L = 8
            constexpr int T4 = L == 4 ? 64 : T;
            constexpr int TT = L == 1 ? TF : T4;
            if (args.any_blend) start<L, TT, WAVES, MINW, false, false, true, 4, false, true>(args, grid, block, stream);
            else start<L, TT, WAVES, MINW, false, false, false, 4, false, true>(args, grid, block, stream);
            return;
        }
    }
    if constexpr (MINW == 1 || L > 2) {     // (two lanes with eight formants laid out: no two-wave instantiation)
        if (!args.state) {
            // (no half-live loops in tolerance mode: which formants a wave skips would be a decision of the wave,
            // and a lane's samples may not depend on its wave-mates)
            if (args.any_blend) start<L, TF, WAVES, MINW, false, false, true, NF, false, true>(args, grid, block, stream);
            else start<L, TF, WAVES, MINW, false, false, false, NF, false, true>(args, grid, block, stream);
            return;
        }
    }
    // resumable streams in tolerance mode (chunks concatenate to the one-shot rendering within the
    // tolerance, not bit for bit: the interpolation ends restart with every call)
    if constexpr (MINW == 1) start<L, T, WAVES, MINW, true, false, true, NF, false, true>(args, grid, block, stream);
}

// time-split fast kernels: one lane per (utterance, chunk), 64-thread workgroups, chunk-major
template <int NFA_, bool ANYBL_>
void launch_one_split(const SynthArgs &args, hipStream_t stream)
{
    const dim3 grid(((args.n_utt + 63u) / 64u) * args.split_chunks), block(64);
    start<1, 64, 1, 1, false, false, ANYBL_, NFA_, false, true, 2, true>(args, grid, block, stream);
}

// the second tolerance tier (MID: the reference's own filter coefficients at every sample), one lane per utterance
template <int NFA_, bool ANYBL_>
void launch_one_mid(const SynthArgs &args, hipStream_t stream)
{
    const dim3 grid = lane_grid<1, 1>(args), block(64);
    start<1, 64, 1, 1, false, false, ANYBL_, NFA_, false, true, 2, false, true>(args, grid, block, stream);
}
template <int NFA_, bool ANYBL_>
void launch_one_split_mid(const SynthArgs &args, hipStream_t stream)
{
    const dim3 grid(((args.n_utt + 63u) / 64u) * args.split_chunks), block(64);
    start<1, 64, 1, 1, false, false, ANYBL_, NFA_, false, true, 2, true, true>(args, grid, block, stream);
}

}  // namespace
}  // namespace grail
