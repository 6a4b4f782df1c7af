// synth_inst_fast_l4.hip — synth_kernel instantiations: 4 lane(s) per utterance, fast arithmetic.
// <L, T, WAVES, MINW>: 64-thread workgroups are admitted 8 per CU (2 waves per SIMD, measured); L = 4 / 8 use
// 256-thread workgroups so that more waves can be resident.
#include "synth_launch_impl.h"

namespace grail {
void launch_fast_l4(const SynthArgs &args, hipStream_t stream) { launch_one_fast<4, 32, 4, 1>(args, stream); }
}  // namespace grail
