import os, sys
sys.path.insert(0, "grail-rs_amd")
import numpy as np
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
ctx.set_voices(W.single_voice())
stride = W.max_samples()
n = 65536
segs, offs, vids, seeds = W.make_batch(n)
batch = ctx.upload(segs, offs, vids, seeds)
d_out = ctx.device_alloc(n * stride * 4)
d_len = ctx.device_alloc(n * 4)
for fast in (0, 1, 0, 1):
    ctx.set_option("arithmetic", fast)
    ms = []
    for _ in range(8):
        batch.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(round(ctx.last_kernel_ms(), 2))
    print("fast" if fast else "exact", ms, ctx.last_kernel_name())
