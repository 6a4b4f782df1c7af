#!/usr/bin/env python3
"""Scan kernel: three-stage (SPLIT) against two-stage workgroups over the batch size — the source of the
default of "time_parallel_scan_split_max_utterances"."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
stride = W.max_samples()
ctx.set_option("arithmetic", 1)
ctx.set_option("time_parallel_scan_max_utterances", 1 << 20)
for n_voices in (1, 8):
    ctx.set_voices(W.single_voice() if n_voices == 1 else W.preset_voices(8))
    for n in (256, 512, 1024, 1536, 2048, 3072, 4096):
        segs, offs, vids, seeds = W.make_batch(n, n_voices=n_voices)
        batch = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
        row = [f"voices={n_voices} n={n:5d}:"]
        for split in (1, 0):
            ctx.set_option("time_parallel_scan_split_max_utterances", (1 << 20) if split else 0)
            ms = []
            for _ in range(3):
                batch.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(ctx.last_kernel_ms())
            row.append(f"{ctx.last_kernel_name()} {min(ms):6.2f} ms")
        print("  ".join(row), flush=True)
        ctx.device_free(d_out); ctx.device_free(d_len); batch.free()
