// synth_inst_mid.hip — synth_kernel instantiations: the second tolerance tier (MID: the reference's own band-pass
// coefficients at every sample, fast arithmetic elsewhere; see MID in synth_kernel.h), one lane per utterance, one-shot,
// four or eight formants, any blend length or powers of two, and resumable (streams); its time-split form is in
// synth_inst_split_mid.hip.
#include "synth_launch_impl.h"

namespace grail {
void launch_mid_l1(const SynthArgs &args, hipStream_t stream)
{
    if (args.state) {
        // resumable (streams of sharp voices, one lane per utterance): the lean four-formant instantiation when the
        // stream was opened for it, the general one otherwise — the state layout is the exact kernels'
        const dim3 grid = lane_grid<1, 1>(args), block(64);
        // (a stream opened for the four-formant layout keeps it whatever arithmetic a call runs: any blend length)
        if (args.live4 && args.any_blend) start<1, 32, 1, 1, true, false, true, 4, false, true, 2, false, true>(args, grid, block, stream);
        else if (args.live4) start<1, 32, 1, 1, true, false, false, 4, false, true, 2, false, true>(args, grid, block, stream);
        else start<1, 32, 1, 1, true, false, true, NF, false, true, 2, false, true>(args, grid, block, stream);
        return;
    }
    if (args.live4) {
        if (args.any_blend) launch_one_mid<4, true>(args, stream);
        else launch_one_mid<4, false>(args, stream);
    } else {
        if (args.any_blend) launch_one_mid<NF, true>(args, stream);
        else launch_one_mid<NF, false>(args, stream);
    }
}
}  // namespace grail
