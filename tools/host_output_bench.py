#!/usr/bin/env python3
"""End-to-end rate of the one-call host-buffer path (grail_synthesize_batch, GRAIL_OUT_HOST):
upload + kernel + device-to-host copy of the PCM, f32 and i16.  usage: host_output_bench.py [n_utt]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = G.Context(0)
ctx.set_voices(W.single_voice())
segs, offs, vids, seeds = W.make_batch(n)
stride = W.max_samples()
for name, fn in (("f32", ctx.synthesize), ("i16", ctx.synthesize_pcm16)):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        out, lens = fn(segs, offs, vids, seeds, out_stride=stride)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    total = int(lens.astype(np.uint64).sum())
    print(f"{name}: {n} utterances, {total} samples, {out.nbytes / 1e9:.2f} GB to the host in {best * 1e3:.1f} ms "
          f"end to end = {total / best:.3e} samples/s ({out.nbytes / best / 1e9:.1f} GB/s into a pageable numpy buffer)",
          flush=True)
