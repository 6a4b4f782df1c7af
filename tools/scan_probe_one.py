#!/usr/bin/env python3
"""Development probe: one batch through the scan kernel.  usage: scan_probe_one.py <n_utt> <scan_debug> [n_voices]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W
n = int(sys.argv[1]); mode = int(sys.argv[2]); nv = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = G.Context(0)
stride = W.max_samples()
ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
segs, offs, vids, seeds = W.make_batch(n, n_voices=nv)
batch = ctx.upload(segs, offs, vids, seeds)
d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
ctx.set_option("arithmetic", 1); ctx.set_option("time_parallel_scan", 1)
ctx.set_option("time_parallel_scan_max_utterances", 1 << 20)
ctx.set_option("scan_debug", mode)
for _ in range(2):
    batch.synthesize_async(d_out, stride, d_len); ctx.sync()
print(ctx.last_kernel_name(), ctx.last_kernel_ms(), "quick tiles", ctx.get_option("fast_wave_tiles"), "derived", ctx.get_option("general_wave_steps"))
