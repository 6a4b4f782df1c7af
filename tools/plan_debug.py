#!/usr/bin/env python3
"""The ragged planner's prices next to measurements: speech-like batches, every lane mapping pinned, and what the planner
says each would cost (GRAIL_PLAN_DEBUG=1 makes launch_plan.cpp print its candidates on stderr).
usage (GPU box): GRAIL_PLAN_DEBUG=1 python3 tools/plan_debug.py [n_utt ...] [--fast] 2>&1"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [131072, 160000, 200000]
fast = 1 if "--fast" in sys.argv else 0
ctx = G.Context(0)
for nv in (1, 8):
    ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
    for n in sizes:
        segs, offs, vids, seeds, stride = W.speech_like_batch(n, np.random.default_rng(7), n_voices=nv)
        b = ctx.upload(segs, offs, vids, seeds)
        d_out, d_len = ctx.device_alloc(n * stride * 4), ctx.device_alloc(n * 4)
        ctx.set_option("arithmetic", fast)
        print(f"=== {n} rows, {nv} voice(s), {'fast' if fast else 'exact'}", file=sys.stderr, flush=True)
        for lanes in (0, 1, 2, 4, 8):
            if n * lanes > 8 * 65536 * 4:
                continue
            ctx.set_option("lanes_per_utterance", lanes)
            ms = []
            for _ in range(2):
                b.synthesize_async(d_out, stride, d_len)
                ctx.sync()
                ms.append(ctx.last_kernel_ms())
            print(f"   measured lanes={lanes}: {min(ms):.2f} ms {ctx.last_kernel_name()} x{ctx.get_option('last_launch_blocks')} "
                  f"packed={ctx.get_option('last_launch_packed')}", file=sys.stderr, flush=True)
        ctx.set_option("lanes_per_utterance", 0)
        ctx.set_option("arithmetic", 0)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        b.free()
