#!/usr/bin/env python3
"""Option "packed_launch_order" off / on: the library's own plan for speech-like batches of more than one round of the device
(kernel ms, best of three), one and eight voices, exact and tolerance arithmetic; with --check the rows of the two
renderings are compared through their on-device digests (the launch order cannot move a bit).
usage: packed_order_ab.py [n_utt ...] [--check]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [65536, 100000, 131072, 160000, 200000]
check = "--check" in sys.argv
ctx = G.Context(0)
print("# speech-like corpus, kernel ms (best of 3), the library's plan; option \"packed_launch_order\" off -> on")
for n_voices in (1, 8):
    ctx.set_voices(W.single_voice() if n_voices == 1 else W.preset_voices(8))
    for n in sizes:
        segs, offs, vids, seeds, stride = W.speech_like_batch(n, np.random.default_rng(7), n_voices=n_voices)
        batch = ctx.upload(segs, offs, vids, seeds)
        d_out, d_len = ctx.device_alloc(n * stride * 4), ctx.device_alloc(n * 4)
        cells = []
        for fast in (0, 1):
            ctx.set_option("arithmetic", fast)
            res, sums = [], []
            for packed in (0, 1):
                ctx.set_option("packed_launch_order", packed)
                ms = []
                for _ in range(3):
                    batch.synthesize_async(d_out, stride, d_len)
                    ctx.sync()
                    ms.append(ctx.last_kernel_ms())
                kern = ctx.last_kernel_name().replace("synth_kernel", "")
                res.append((min(ms), f"{kern} x{ctx.get_option('last_launch_blocks')}" + (" packed" if ctx.get_option("last_launch_packed") else "")))
                if check:
                    sums.append(ctx.digest(d_out, stride, d_len, n)[0].copy())
            same = "" if not check else ("  rows equal" if np.array_equal(sums[0], sums[1]) else "  ROWS DIFFER")
            cells.append(f"{'fast ' if fast else 'exact'} {res[0][0]:7.2f} -> {res[1][0]:7.2f} ms ({res[1][0] / res[0][0]:.3f} x; {res[0][1]} -> {res[1][1]}){same}")
        ctx.set_option("arithmetic", 0)
        ctx.set_option("packed_launch_order", 1)
        print(f"{n:7d} rows, {n_voices} voice(s): " + " | ".join(cells), flush=True)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()
ctx.close()
