#!/usr/bin/env python3
"""Kernel time of the three arithmetics — exact, fast (coefficients interpolated), fast with the reference's own
coefficients (MID) — on the bench corpus.   usage: mid_bench.py [n ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W

ctx = G.Context(0)
stride = W.max_samples()
for n_voices in (1, 8):
    ctx.set_voices(W.single_voice() if n_voices == 1 else W.preset_voices(8))
    for n in ([int(a) for a in sys.argv[1:]] or (1024, 4096, 16384, 32768, 65536)):
        segs, offs, vids, seeds = W.make_batch(n, n_voices=n_voices)
        batch = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * 4)
        d_len = ctx.device_alloc(n * 4)
        row = [f"voices={n_voices} n={n:6d}:"]
        res = {}
        for mode in (0, 1, 2, 0, 1, 2):
            ctx.set_option("arithmetic", mode)
            ms = []
            for _ in range(3):
                batch.synthesize_async(d_out, stride, d_len)
                ctx.sync()
                ms.append(ctx.last_kernel_ms())
            name = ctx.last_kernel_name().replace("synth_kernel", "k")
            chunks = ctx.get_option("last_launch_chunks")
            res[mode] = (min(ms + [res.get(mode, (1e9,))[0]]), name + (f" x{chunks}" if chunks else ""))
        for mode in (0, 1, 2):
            row.append(f"{('exact', 'fast', 'mid')[mode]} {res[mode][0]:7.2f} ms ({res[mode][1]})")
        row.append(f"mid / exact {res[2][0] / res[0][0]:.2f}")
        print("  ".join(row), flush=True)
        ctx.set_option("arithmetic", 0)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()
