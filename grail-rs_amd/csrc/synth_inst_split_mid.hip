// synth_inst_split_mid.hip — synth_kernel instantiations: the time-split kernels of the second tolerance tier (MID).
#include "synth_launch_impl.h"

namespace grail {
void launch_split_mid(const SynthArgs &args, hipStream_t stream)
{
    if (args.live4) {
        if (args.any_blend) launch_one_split_mid<4, true>(args, stream);
        else launch_one_split_mid<4, false>(args, stream);
    } else {
        if (args.any_blend) launch_one_split_mid<NF, true>(args, stream);
        else launch_one_split_mid<NF, false>(args, stream);
    }
}
}  // namespace grail
