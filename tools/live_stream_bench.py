#!/usr/bin/env python3
"""Live streams (grail_stream_open_live / _append: the lazy source of examples/interactive.rs:31-48) next to closed-batch
streams of the same utterances: kernel time per pull of `chunk` samples with everything appended up front, and the cost
of an append call (host time + the ring-scatter kernel) when one segment per utterance is fed between pulls, as an
interactive front end does.   usage: live_stream_bench.py [arithmetic 0|1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

ctx = G.Context(0)
ctx.set_voices(W.single_voice())
fast = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx.set_option("arithmetic", fast)
print("arithmetic:", "fast" if fast else "exact", flush=True)
chunk = 4800
stride = (chunk + 63) // 64 * 64
for n in (1, 256, 4096, 65536):
    segs, offs, vids, seeds = W.make_batch(n)
    per = int(offs[1] - offs[0])                          # segments per utterance (4)
    d_out = ctx.device_alloc(n * stride * 4)
    d_len = ctx.device_alloc(n * 4)

    def pull_all(st, pulls):
        ms = []
        for _ in range(pulls):
            st.next_async(chunk, d_out, stride, d_len)
            ctx.sync()
            ms.append(ctx.last_kernel_ms())
        return ms, ctx.last_kernel_name()

    batch = ctx.upload(segs, offs, vids, seeds)
    st = G.Stream(batch)
    closed, closed_name = pull_all(st, 12)
    st.close()
    batch.free()
    # live, everything appended before the first pull
    live = G.LiveStream(ctx, n, vids, seeds, ring_segments=8)
    live.append(segs, offs)
    live.finish()
    upfront, live_name = pull_all(live, 12)
    live.close()
    # live, one segment per utterance appended between pulls (the source runs just ahead of the Sequencer)
    live = G.LiveStream(ctx, n, vids, seeds, ring_segments=8)
    seg_arr = np.asarray(segs).reshape(n, per)
    one_offs = np.arange(n + 1, dtype=np.uint32)
    fed, app_ms, lazy = 0, [], []
    for p in range(12):
        while fed < per and fed < 2 + (p * chunk) // 24000:      # two segments ahead of the clock
            t0 = time.perf_counter()
            live.append(np.ascontiguousarray(seg_arr[:, fed]), one_offs)
            app_ms.append((time.perf_counter() - t0) * 1e3)
            fed += 1
            if fed == per:
                live.finish()
        live.next_async(chunk, d_out, stride, d_len)
        ctx.sync()
        lazy.append(ctx.last_kernel_ms())
    live.close()
    med = lambda x: float(np.median(x[2:]))
    print(f"n={n:6d} pull of {chunk} samples: closed-batch stream {med(closed):7.3f} ms ({closed_name})   live, appended up front "
          f"{med(upfront):7.3f} ms ({live_name})   live, fed between pulls {med(lazy):7.3f} ms; an append of one segment per "
          f"utterance {float(np.median(app_ms)):6.3f} ms of host time", flush=True)
    ctx.device_free(d_out)
    ctx.device_free(d_len)
