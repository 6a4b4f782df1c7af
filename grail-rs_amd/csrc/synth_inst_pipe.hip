// synth_inst_pipe.hip — synth_kernel instantiations: the four-wave pipelined workgroups of small exact batches.
// Render, chain, 2 x coefficients share 16 utterances (four live formants, 4 lanes each) or 8 utterances (eight
// formants, 8 lanes each); rounds of 16 samples (73 KB of LDS: two workgroups fit a CU).
#include "synth_launch_impl.h"

namespace grail {
// Resumable (args.state: grail_stream_*, closed batches and live streams): the general instantiation — any blend length,
// segments possibly in a ring — with the state block of the lane kernels of the same L.
void launch_pipe4(const SynthArgs &args, hipStream_t stream)
{
    const uint32_t fill = args.pipe_fill != 0u && !args.state ? args.pipe_fill : 16u;
    const dim3 grid((args.n_utt + fill - 1) / fill), block(256);
    if (args.state) {
        if (args.pipe == 2) start<4, 64, 4, 1, true, false, true, 4, true, false, 8>(args, grid, block, stream);
        else start<4, 64, 4, 1, true, false, true, 4, true, false, 4>(args, grid, block, stream);
        return;
    }
    // args.pipe == 2: rounds of 32 samples (131 KB of LDS: one workgroup per CU, half as many barriers)
    // (any_blend: the chain wave divides clk by the blend length instead of multiplying by 2^-k, nothing else differs)
    if (args.any_blend) {
        if (args.pipe == 2) start<4, 64, 4, 1, false, false, true, 4, true, false, 8>(args, grid, block, stream);
        else start<4, 64, 4, 1, false, false, true, 4, true, false, 4>(args, grid, block, stream);
        return;
    }
    if (args.pipe == 2) start<4, 64, 4, 1, false, false, false, 4, true, false, 8>(args, grid, block, stream);
    else start<4, 64, 4, 1, false, false, false, 4, true, false, 4>(args, grid, block, stream);
}
void launch_pipe8(const SynthArgs &args, hipStream_t stream)
{
    const uint32_t fill = args.pipe_fill != 0u && !args.state ? args.pipe_fill : 8u;
    const dim3 grid((args.n_utt + fill - 1) / fill), block(256);
    if (args.state) {
        if (args.pipe == 2) start<8, 64, 4, 1, true, false, true, NF, true, false, 8>(args, grid, block, stream);
        else start<8, 64, 4, 1, true, false, true, NF, true, false, 4>(args, grid, block, stream);
        return;
    }
    if (args.any_blend) {
        if (args.pipe == 2) start<8, 64, 4, 1, false, false, true, NF, true, false, 8>(args, grid, block, stream);
        else start<8, 64, 4, 1, false, false, true, NF, true, false, 4>(args, grid, block, stream);
        return;
    }
    if (args.pipe == 2) start<8, 64, 4, 1, false, false, false, NF, true, false, 8>(args, grid, block, stream);
    else start<8, 64, 4, 1, false, false, false, NF, true, false, 4>(args, grid, block, stream);
}
}  // namespace grail
