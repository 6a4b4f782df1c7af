#!/usr/bin/env python3
"""Development probe: scan kernel time, two-stage and three-stage (SPLIT) workgroups, with the filter wave
switched off (scan_debug 201)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
stride = W.max_samples()
for n_voices in (1, 8):
    ctx.set_voices(W.single_voice() if n_voices == 1 else W.preset_voices(8))
    for n in (64, 256, 1024, 4096):
        segs, offs, vids, seeds = W.make_batch(n, n_voices=n_voices)
        batch = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * 4)
        d_len = ctx.device_alloc(n * 4)
        ctx.set_option("arithmetic", 1); ctx.set_option("time_parallel_scan", 1)
        ctx.set_option("time_parallel_scan_max_utterances", 1 << 20)
        row = [f"voices={n_voices} n={n:5d}:"]
        for split in (0, 1):
            ctx.set_option("time_parallel_scan_split_max_utterances", (1 << 20) if split else 0)
            for mode in ((0, 201, 205) if split else (0, 201)):
                ctx.set_option("scan_debug", mode)
                ms = []
                for _ in range(3):
                    batch.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(ctx.last_kernel_ms())
                row.append(f"split {split} mode {mode}: {min(ms):6.2f} ms")
        print("  ".join(row), flush=True)
        ctx.set_option("scan_debug", 0)
        ctx.device_free(d_out); ctx.device_free(d_len); batch.free()
