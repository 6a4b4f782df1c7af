#!/usr/bin/env python3
"""What a pause before a launch costs the launch: the headline kernel (65 536 utterances x 2 s) repeated with a host
sleep of 0 ... 100 ms between the end of one launch and the start of the next (the device idles in between and lowers
its clocks; a VALU-bound kernel pays for the ramp).   usage: idle_gap_check.py [fast]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W

fast = 1 if len(sys.argv) > 1 else 0
ctx = G.Context(0)
ctx.set_voices(W.single_voice())
ctx.set_option("arithmetic", fast)
stride = W.max_samples()
n = 65536
segs, offs, vids, seeds = W.make_batch(n)
batch = ctx.upload(segs, offs, vids, seeds)
d_out = ctx.device_alloc(n * stride * 4)
d_len = ctx.device_alloc(n * 4)
for _ in range(4):
    batch.synthesize_async(d_out, stride, d_len)
    ctx.sync()
print(f"# {ctx.last_kernel_name()}: kernel ms after a pause of g ms (median of 5; back to back = queued behind the previous launch)")
ms = []
for _ in range(6):
    batch.synthesize_async(d_out, stride, d_len)
batch.synthesize_async(d_out, stride, d_len)
ctx.sync()
print(f"back to back (7 launches queued): last {ctx.last_kernel_ms():7.2f}")
for gap in (0.0, 0.5, 1.0, 2.0, 5.0, 10.0, 20.0, 50.0, 100.0):
    ms = []
    for _ in range(5):
        time.sleep(gap * 1e-3)
        batch.synthesize_async(d_out, stride, d_len)
        ctx.sync()
        ms.append(ctx.last_kernel_ms())
    print(f"pause {gap:6.1f} ms: {sorted(ms)[2]:7.2f}   ({' '.join(f'{x:.2f}' for x in ms)})", flush=True)
