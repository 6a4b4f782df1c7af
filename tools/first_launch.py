#!/usr/bin/env python3
"""How long does the FIRST launch of the headline kernel take?  (hipEvent times, no profiler.)
touch 0: rows never written before; touch 1: hipMemset of the rows first; touch 2: another kernel wrote them
first; bench-like: what bench.py does (memset, four launches of a 4096-utterance sub-batch, the real batch).
Under rocprofv3 --kernel-trace the first dispatch of this kernel measures ~78 ms whatever is done before it;
without the profiler it is within 3 % of the steady state as soon as the rows have been touched."""
import sys, os
sys.path.insert(0, "grail-rs_amd")
import numpy as np, grail_hip as G
from grail_hip import workload as W
for touch in (0, 1, 2):
    ctx = G.Context(0)
    ctx.set_voices(W.single_voice())
    n = 65536
    segs, offs, vids, seeds = W.make_batch(n)
    stride = W.max_samples()
    b = ctx.upload(segs, offs, vids, seeds)
    d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
    if touch >= 1: ctx.memset(d_out, 0, n * stride * 4)
    if touch == 2:
        ctx.set_option("arithmetic", 1); b.synthesize_async(d_out, stride, d_len); ctx.sync(); ctx.set_option("arithmetic", 0)
    ms = []
    for i in range(4):
        b.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(round(ctx.last_kernel_ms(), 2))
    print("touch", touch, ms, flush=True)
    ctx.device_free(d_out); ctx.device_free(d_len); b.free(); ctx.close()
# variant: as bench.py does it (memset, four launches of a 4096-utterance sub-batch, then the real batch)
ctx = G.Context(0)
ctx.set_voices(W.single_voice())
n = 65536
segs, offs, vids, seeds = W.make_batch(n)
stride = W.max_samples()
b = ctx.upload(segs, offs, vids, seeds)
d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
ctx.memset(d_out, 0, n * stride * 4)
r = ctx.upload(*W.make_batch(4096))
for _ in range(4):
    r.synthesize_async(d_out, stride, d_len)
ctx.sync()
ms = []
for i in range(4):
    b.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(round(ctx.last_kernel_ms(), 2))
print("bench-like", ms, flush=True)
