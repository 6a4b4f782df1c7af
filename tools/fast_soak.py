#!/usr/bin/env python3
"""Fast mode against exact mode, every sample of full 2-second batches, one batch size per fast kernel family
(scan kernel three-stage / two-stage, lane kernels L = 8 / 4 / 2 / 1), generic voice and 8 presets: the largest
deviation in units of 2^-23, the rms, and that lengths and structure (bad = samples where one side is not finite
or the rows differ in length) agree.  The contract is GRAIL_FAST_TOLERANCE = 64 * 2^-23.
--shards: also shards 1 .. 7 of BASELINE config 5 (what the other seven ranks of an 8-GPU run render)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

ctx = G.Context(0)
stride = W.max_samples()
worst = 0.0
for nv in (1, 8):
    ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
    for n in (256, 1024, 4096, 8192, 12288, 16384, 32768, 65536):
        segs, offs, vids, seeds = W.make_batch(n, n_voices=nv)
        batch = ctx.upload(segs, offs, vids, seeds)
        d_a = ctx.device_alloc(n * stride * 4); d_b = ctx.device_alloc(n * stride * 4)
        l_a = ctx.device_alloc(n * 4); l_b = ctx.device_alloc(n * 4)
        ctx.set_option("arithmetic", 0)
        batch.synthesize_async(d_a, stride, l_a); ctx.sync()
        exact_name = ctx.last_kernel_name()
        ctx.set_option("arithmetic", 1)
        batch.synthesize_async(d_b, stride, l_b); ctx.sync()
        fast_name = ctx.last_kernel_name()
        ctx.set_option("arithmetic", 0)
        md, sq, bad = ctx.compare(d_a, d_b, stride, l_a, l_b, n)
        lens = np.zeros(n, dtype=np.uint32); ctx.d2h(lens, l_a, n * 4)
        k = float(md.max()) * 2.0 ** 23
        rms = float(np.sqrt(sq.sum() / max(int(lens.astype(np.uint64).sum()), 1))) * 2.0 ** 23
        worst = max(worst, k)
        print(f"voices={nv} n={n:6d}: max |fast - exact| = {k:5.1f} * 2^-23, rms {rms:4.2f}, structural mismatches {int(bad.sum())}"
              f"   {fast_name}  vs  {exact_name}", flush=True)
        assert int(bad.sum()) == 0 and k <= G.FAST_TOLERANCE_ULPS
        for d in (d_a, d_b, l_a, l_b):
            ctx.device_free(d)
        batch.free()
# the eight shards of BASELINE config 5 (524 288 utterances, 65 536 per GPU): what ranks 1 .. 7 render
if "--shards" in sys.argv:
    for nv in (1, 8):
        ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
        n = 65536
        d_a = ctx.device_alloc(n * stride * 4); d_b = ctx.device_alloc(n * stride * 4)
        l_a = ctx.device_alloc(n * 4); l_b = ctx.device_alloc(n * 4)
        for shard in range(1, 8):
            segs, offs, vids, seeds = W.make_batch(n, first_utt=shard * n, n_voices=nv)
            batch = ctx.upload(segs, offs, vids, seeds)
            ctx.set_option("arithmetic", 0)
            batch.synthesize_async(d_a, stride, l_a); ctx.sync()
            ctx.set_option("arithmetic", 1)
            batch.synthesize_async(d_b, stride, l_b); ctx.sync()
            ctx.set_option("arithmetic", 0)
            md, sq, bad = ctx.compare(d_a, d_b, stride, l_a, l_b, n)
            k = float(md.max()) * 2.0 ** 23
            worst = max(worst, k)
            print(f"voices={nv} shard {shard} (utterances {shard * n} ..): max |fast - exact| = {k:5.1f} * 2^-23, "
                  f"structural mismatches {int(bad.sum())}", flush=True)
            assert int(bad.sum()) == 0 and k <= G.FAST_TOLERANCE_ULPS
            batch.free()
        for d in (d_a, d_b, l_a, l_b):
            ctx.device_free(d)
print(f"worst {worst:.1f} * 2^-23 (contract: {G.FAST_TOLERANCE_ULPS})")
