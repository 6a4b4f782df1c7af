// synth_inst_exact_l8.hip — synth_kernel instantiations: 8 lane(s) per utterance, exact arithmetic.
// <L, T, WAVES, MINW>: T samples per tile; WAVES waves per workgroup (L = 4 / 8: 256-thread workgroups, whose four
// waves share the workgroup's LDS tiles); MINW = 1: compiled for ONE resident wave per SIMD, all 512 registers a lane
// can have (DESIGN.md §4.1 — a second wave on a SIMD costs more than it brings, profiles/r04_two_waves.txt).
#include "synth_launch_impl.h"

namespace grail {
void launch_exact_l8(const SynthArgs &args, hipStream_t stream) { launch_one_exact<8, 64, 4, 1>(args, stream); }
}  // namespace grail
