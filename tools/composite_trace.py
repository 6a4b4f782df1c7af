#!/usr/bin/env python3
"""Runs composite and single launches of one batch size alternately (for rocprofv3 --kernel-trace: the duration of
every block of a composite launch).   usage: composite_trace.py n [fast] [order]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W

n = int(sys.argv[1])
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 0
order = sys.argv[3] if len(sys.argv) > 3 else "10101010"
ctx = G.Context(0)
ctx.set_voices(W.single_voice())
stride = W.max_samples()
segs, offs, vids, seeds = W.make_batch(n)
batch = ctx.upload(segs, offs, vids, seeds)
d_out = ctx.device_alloc(n * stride * 4)
d_len = ctx.device_alloc(n * 4)
ctx.set_option("arithmetic", fast)
for ch in order:
    ctx.set_option("composite_launches", int(ch))
    batch.synthesize_async(d_out, stride, d_len)
    ctx.sync()
    print(ch, round(ctx.last_kernel_ms(), 2), ctx.get_option("last_launch_blocks"), flush=True)
