#!/usr/bin/env python3
"""Real-time path (grail_stream_*, the analogue of examples/interactive.rs:31-48): per-chunk cost of
pulling `chunk` samples at a time for n utterances — kernel time, wall time per synchronous call,
wall time per call when calls are queued back to back (one sync per 8 calls: what a double-buffered
audio callback does) — next to the one-shot kernel of the same batch on the same box."""
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
ctx.set_option("time_parallel_scan", 0)          # compare like with like: lane kernels on both sides
print("arithmetic:", "fast" if fast else "exact", flush=True)
for n in (1, 256, 4096, 65536):
    segs, offs, vids, seeds = W.make_batch(n)
    batch = ctx.upload(segs, offs, vids, seeds)
    full = W.max_samples()
    d_full = ctx.device_alloc(n * full * 4)
    d_len = ctx.device_alloc(n * 4)
    one = []
    for _ in range(3):
        batch.synthesize_async(d_full, full, d_len)
        ctx.sync()
        one.append(ctx.last_kernel_ms())
    lens = np.zeros(n, dtype=np.uint32)
    ctx.d2h(lens, d_len, n * 4)
    one_rate = float(lens.astype(np.uint64).sum()) / (min(one) * 1e-3)
    print(f"n={n:6d} one-shot kernel {min(one):7.3f} ms = {one_rate:.3e} samples/s ({ctx.last_kernel_name()})", flush=True)
    ctx.device_free(d_full)
    for chunk in (480, 960, 4800):       # 10 ms, 20 ms, 100 ms at 48 kHz
        stride = (chunk + 63) // 64 * 64
        d_out = ctx.device_alloc(n * stride * 4)
        st = G.Stream(batch)
        calls = min(40, 96000 // chunk)
        kms, wall = [], []
        for i in range(calls // 2):
            t0 = time.perf_counter()
            st.next_async(chunk, d_out, stride, d_len)
            ctx.sync()
            wall.append((time.perf_counter() - t0) * 1e3)
            kms.append(ctx.last_kernel_ms())
        name = ctx.last_kernel_name()
        q = calls - calls // 2
        t0 = time.perf_counter()
        for i in range(q):
            st.next_async(chunk, d_out, stride, d_len)
        ctx.sync()
        queued = (time.perf_counter() - t0) * 1e3 / q
        st.close()
        k, w = float(np.median(kms[2:])), float(np.median(wall[2:]))
        print(f"n={n:6d} chunk={chunk:5d} ({chunk / 48:.0f} ms audio): kernel {k:7.3f} ms ({n * chunk / (k * 1e-3) / one_rate * 100:5.1f} % of "
              f"one-shot), call+sync {w:7.3f} ms ({chunk / 48.0 / w:6.1f}x real time), queued {queued:7.3f} ms/call = "
              f"{n * chunk / (queued * 1e-3):.3e} samples/s ({n * chunk / (queued * 1e-3) / one_rate * 100:5.1f} % of one-shot)  {name}",
              flush=True)
        ctx.device_free(d_out)
    ctx.device_free(d_len)
    batch.free()
