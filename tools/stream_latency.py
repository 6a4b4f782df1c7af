#!/usr/bin/env python3
"""Real-time path (grail_stream_*, the analogue of examples/interactive.rs:31-48): per-chunk cost of
pulling `chunk` samples at a time for n utterances, kernel time and wall time per call, and the
real-time factor (audio seconds produced per wall second per utterance)."""
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
for n in (1, 64, 4096, 65536):
    for chunk in (480, 4800):       # 10 ms, 100 ms at 48 kHz
        segs, offs, vids, seeds = W.make_batch(n)
        batch = ctx.upload(segs, offs, vids, seeds)
        stride = (chunk + 63) // 64 * 64
        d_out = ctx.device_alloc(n * stride * 4)
        d_len = ctx.device_alloc(n * 4)
        st = G.Stream(batch)
        calls = min(40, 96000 // chunk)
        kms, wall = [], []
        for i in range(calls):
            t0 = time.perf_counter()
            st.next_async(chunk, d_out, stride, d_len)
            ctx.sync()
            wall.append((time.perf_counter() - t0) * 1e3)
            kms.append(ctx.last_kernel_ms())
        st.close()
        k, w = float(np.median(kms[2:])), float(np.median(wall[2:]))
        print(f"n={n:6d} chunk={chunk:5d} samples ({chunk / 48:.0f} ms audio): kernel {k:7.3f} ms, "
              f"call+sync {w:7.3f} ms  -> {chunk / 48.0 / w:6.1f}x real time per utterance, "
              f"{n * chunk / (w * 1e-3):.3e} samples/s", flush=True)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()
