#!/usr/bin/env python3
"""Kernel time of small and mid-size batches (2 s utterances) in both arithmetic modes: the lane-per-utterance
kernels against the time-parallel scan kernel and the time-split kernels.   usage: small_batch_bench.py [n ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

ctx = G.Context(0)
stride = W.max_samples()
for n_voices in (1, 8):
    ctx.set_voices(W.single_voice() if n_voices == 1 else W.preset_voices(8))
    for n in ([int(a) for a in sys.argv[1:]] or (1, 64, 256, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768)):
        segs, offs, vids, seeds = W.make_batch(n, n_voices=n_voices)
        batch = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * 4)
        d_len = ctx.device_alloc(n * 4)
        row = [f"voices={n_voices} n={n:5d}:"]
        for label, fast, scan, split in (("exact", 0, 1, 0), ("fast lanes", 1, 0, 0), ("fast scan", 1, 1, 0),
                                         ("fast split", 1, 0, 1)):
            if label == "fast scan" and n > 8192:
                continue
            if label == "fast split" and n < 64:
                continue
            ctx.set_option("arithmetic", fast)
            ctx.set_option("time_parallel_scan", scan)
            ctx.set_option("time_parallel_scan_max_utterances", 1 << 20)
            ctx.set_option("time_split", split)
            ctx.set_option("time_split_min_utterances", 0)
            ms = []
            for _ in range(3):
                batch.synthesize_async(d_out, stride, d_len)
                ctx.sync()
                ms.append(ctx.last_kernel_ms())
            chunks = ctx.get_option("last_launch_chunks")
            row.append(f"{label} {min(ms):7.2f} ms ({ctx.last_kernel_name().replace('synth_kernel', 'k')}"
                       + (f" x{chunks}" if chunks else "") + ")")
        print("  ".join(row), flush=True)
        ctx.set_option("arithmetic", 0)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()
