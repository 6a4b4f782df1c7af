// synth_launch.h — the instantiation units of synth_kernel (internal).  One translation unit per lane mapping
// and arithmetic, so that the kernel families compile in parallel (synth_kernels.hip dispatches).
#pragma once

#include "kernels.h"

namespace grail {

// exact arithmetic (one-shot and resumable), lanes per utterance 1 / 2 / 4 / 8
void launch_exact_l1(const SynthArgs &args, hipStream_t stream);
void launch_exact_l2(const SynthArgs &args, hipStream_t stream);
void launch_exact_l4(const SynthArgs &args, hipStream_t stream);
void launch_exact_l8(const SynthArgs &args, hipStream_t stream);
// tolerance arithmetic (args.fast)
void launch_fast_l1(const SynthArgs &args, hipStream_t stream);
void launch_fast_l2(const SynthArgs &args, hipStream_t stream);
void launch_fast_l4(const SynthArgs &args, hipStream_t stream);
void launch_fast_l8(const SynthArgs &args, hipStream_t stream);
// tolerance arithmetic, the time axis of every utterance cut into args.split_chunks chunks (one lane each)
void launch_split(const SynthArgs &args, hipStream_t stream);
// the second tolerance tier (args.fast == 2; synth_kernel<..., MID>): one lane per utterance, and its time-split form
void launch_mid_l1(const SynthArgs &args, hipStream_t stream);
void launch_split_mid(const SynthArgs &args, hipStream_t stream);
// the four-wave pipelined workgroups of small exact batches (args.pipe): four or eight live formants
void launch_pipe4(const SynthArgs &args, hipStream_t stream);
void launch_pipe8(const SynthArgs &args, hipStream_t stream);

}  // namespace grail
