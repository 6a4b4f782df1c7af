#!/usr/bin/env python3
"""Where the tolerance-mode kernels spend a launch on a speech-like corpus: the cycle and event counters of a
-DGRAIL_FAST_PROF build (synth_kernel.h PROF_ADD / PROF_CNT), per wave.

usage: GRAIL_HIP_LIB=<path to a library built with EXTRA=-DGRAIL_FAST_PROF> fast_prof.py [n_utt] [--scale=F] [--lanes=L] [--voices=N] [--aligned] [--exact]
(--aligned: the bench corpus instead)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
scale, lanes, n_voices = 1.0, 0, 1
for a in sys.argv[1:]:
    if a.startswith("--scale="):
        scale = float(a[8:])
    if a.startswith("--lanes="):
        lanes = int(a[8:])
    if a.startswith("--voices="):
        n_voices = int(a[9:])
ctx = G.Context(0)
ctx.set_option("time_split", 0)
ctx.set_option("time_parallel_scan", 0)
if lanes:
    ctx.set_option("lanes_per_utterance", lanes)
else:
    ctx.set_option("ragged_plan", 1)
ctx.set_voices(W.single_voice() if n_voices == 1 else W.preset_voices(8))
rng = np.random.default_rng(7)
if "--aligned" in sys.argv:
    segs, offs, vids, seeds = W.make_batch(n, n_voices=n_voices)
    stride = W.max_samples()
else:
    segs, offs, vids, seeds, stride = W.speech_like_batch(n, rng, n_voices=n_voices, scale=scale)
batch = ctx.upload(segs, offs, vids, seeds)
d_out = ctx.device_alloc(n * stride * 4)
d_len = ctx.device_alloc(n * 4)
exact = "--exact" in sys.argv
ctx.set_option("arithmetic", 0 if exact else 1)
NAMES = {0: "total", 2: "plain runs (pairs + slope refreshes)", 3: "loop head (refresh test)", 5: "slow sample: new beginnings (fast_restart)",
         6: "slow sample: formants", 7: "slow sample: chain part of the general step", 8: "flush + between tiles", 9: "tile head"}
CNT = {10: "plain pairs", 11: "slow samples", 12: "slope-refresh executions (wave level)", 13: "new-beginning executions (wave level)",
       15: "tiles without a slow sample", 16: "lanes beginning anew", 17: "sum of their levels"}


if exact:
    NAMES = {0: "total", 2: "calm tiles", 3: "runs of tiles with an event (pairs + single steps)", 7: "general steps",
             8: "flush + between tiles", 9: "tile head"}
    CNT = {15: "calm tiles", 16: "  of them: one smoothness for all formants", 17: "  of them: upper half silent",
           20: "runs in tiles with an event", 18: "  of them: one smoothness", 19: "  of them: upper half silent",
           10: "packed pairs in those tiles", 12: "single quiet steps in those tiles", 11: "general steps"}


def read():
    return [ctx.get_option(f"debug_prof_{k}") for k in range(32)]


batch.synthesize_async(d_out, stride, d_len)
ctx.sync()
before = read()
batch.synthesize_async(d_out, stride, d_len)
ctx.sync()
ms = ctx.last_kernel_ms()
after = read()
d = [a - b for a, b in zip(after, before)]
waves = max(d[31], 1)
print(f"# {ctx.last_kernel_name()}  {n} utterances, scale {scale}, {n_voices} voice(s): {ms:.2f} ms, {waves} waves, L = {ctx.get_option('last_launch_lanes')}, "
      f"fast = {ctx.get_option('last_launch_fast')}")
tot = d[0] / waves
print(f"# cycles per wave (mean over waves): {tot:.3e}")
for k, name in NAMES.items():
    if k:
        print(f"  {name:32s} {d[k] / waves:12.0f} cycles  {100.0 * d[k] / max(d[0], 1):5.1f} %")
for k, name in CNT.items():
    print(f"  {name:44s} {d[k] / waves:10.1f} per wave")
for a, b, what in (((2, 15, "calm tile"), (7, 11, "general step")) if exact else
                   ((2, 10, "plain pair (incl. refreshes)"), (5, 13, "new-beginning execution"), (7, 11, "slow sample: chain"), (6, 11, "slow sample: formants"))):
    if d[b]:
        print(f"  cycles per {what:32s} {d[a] / d[b]:9.0f}")
