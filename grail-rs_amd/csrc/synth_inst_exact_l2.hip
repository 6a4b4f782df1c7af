// synth_inst_exact_l2.hip — synth_kernel instantiations: 2 lane(s) per utterance, exact arithmetic.
// <L, T, WAVES, MINW>: 64-thread workgroups are admitted 8 per CU (2 waves per SIMD, measured); L = 4 / 8 use
// 256-thread workgroups so that more waves can be resident.
#include "synth_launch_impl.h"

namespace grail {
void launch_exact_l2(const SynthArgs &args, hipStream_t stream) { launch_one_exact<2, 64, 1, 1>(args, stream); }
}  // namespace grail
