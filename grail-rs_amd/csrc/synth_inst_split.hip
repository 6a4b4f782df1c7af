// synth_inst_split.hip — synth_kernel instantiations: the time-split fast kernels (one lane per utterance and
// chunk of its time axis; see SPLIT in synth_kernel.h), four or eight formants, any blend length or powers of two.
#include "synth_launch_impl.h"

namespace grail {
void launch_split(const SynthArgs &args, hipStream_t stream)
{
    if (args.live4) {
        if (args.any_blend) launch_one_split<4, true>(args, stream);
        else launch_one_split<4, false>(args, stream);
    } else {
        if (args.any_blend) launch_one_split<NF, true>(args, stream);
        else launch_one_split<NF, false>(args, stream);
    }
}
}  // namespace grail
