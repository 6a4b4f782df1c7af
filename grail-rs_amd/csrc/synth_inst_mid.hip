// synth_inst_mid.hip — synth_kernel instantiations: the second tolerance tier (MID: the reference's own band-pass
// coefficients at every sample, fast arithmetic elsewhere; see MID in synth_kernel.h), one lane per utterance, one-shot,
// four or eight formants, any blend length or powers of two; and its time-split form.
#include "synth_launch_impl.h"

namespace grail {
void launch_mid_l1(const SynthArgs &args, hipStream_t stream)
{
    if (args.live4) {
        if (args.any_blend) launch_one_mid<4, true>(args, stream);
        else launch_one_mid<4, false>(args, stream);
    } else {
        if (args.any_blend) launch_one_mid<NF, true>(args, stream);
        else launch_one_mid<NF, false>(args, stream);
    }
}
}  // namespace grail
