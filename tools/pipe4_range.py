#!/usr/bin/env python3
"""Exact mode, four live formants: the four-wave pipelined workgroups beyond one workgroup per CU
("pipeline4_max_groups") against the lane kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
stride = W.max_samples()
ctx.set_voices(W.single_voice())
for n in (4096, 5120, 6144, 8192, 12288, 16384):
    segs, offs, vids, seeds = W.make_batch(n)
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
    row = [f"n={n:5d}:"]
    for groups in (256, 1024):
        ctx.set_option("pipeline4_max_groups", groups)
        ms = []
        for _ in range(3):
            batch.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(ctx.last_kernel_ms())
        row.append(f"{ctx.last_kernel_name()} {min(ms):6.2f} ms")
    print("  ".join(row), flush=True)
    ctx.device_free(d_out); ctx.device_free(d_len); batch.free()
