#!/usr/bin/env python3
"""Kernel ms of one batch size with the library named by GRAIL_HIP_LIB (A/B of experimental builds).
usage: GRAIL_HIP_LIB=... lib_ab.py n [presets]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W

n = int(sys.argv[1])
nv = 8 if len(sys.argv) > 2 else 1
ctx = G.Context(0)
ctx.set_voices(W.preset_voices(8) if nv == 8 else W.single_voice())
stride = W.max_samples()
segs, offs, vids, seeds = W.make_batch(n, n_voices=nv)
batch = ctx.upload(segs, offs, vids, seeds)
d_out = ctx.device_alloc(n * stride * 4)
d_len = ctx.device_alloc(n * 4)
ms = []
for _ in range(8):
    batch.synthesize_async(d_out, stride, d_len)
    ctx.sync()
    ms.append(round(ctx.last_kernel_ms(), 3))
print(os.environ.get("GRAIL_HIP_LIB", "default"), n, "voices", nv, ctx.last_kernel_name(), "min", min(ms), ms)
