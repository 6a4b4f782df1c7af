#!/usr/bin/env python3
"""Kernel time of batches just above a whole round of a kernel family: composite launches against
the single launch they replace, exact and fast arithmetic, with the plan the library chose.
usage: tail_bench.py [n ...]        (2 s utterances at 48 kHz, voices::generic(); add --presets for eight live formants)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

args = [a for a in sys.argv[1:] if not a.startswith("--")]
presets = "--presets" in sys.argv
sizes = [int(a) for a in args] or [4096, 8192, 16384, 32768, 36000, 40000, 49152, 65536, 65537, 70000, 81920, 98304,
                                    131072, 131073, 200000]
ctx = G.Context(0)
voices = W.preset_voices(8) if presets else W.single_voice()
ctx.set_voices(voices)
stride = W.max_samples()
warm = max(G.time_split_warmup(v) for v in voices)
cus = ctx.get_option("compute_units")
print(f"# tail_bench: {'8 presets (eight live formants)' if presets else 'voices::generic() (four live formants)'}, "
      f"2 s utterances, compute_units={cus}; kernel ms = min of 5 after a warm-up, composite and single launches alternating (hipEvents around all launches of the call)")
T = {}
for n in sizes:
    segs, offs, vids, seeds = W.make_batch(n, n_voices=len(voices))
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out = ctx.device_alloc(n * stride * 4)
    d_len = ctx.device_alloc(n * 4)
    row = [f"n={n:6d}"]
    for fast in (0, 1):
        ctx.set_option("arithmetic", fast)
        # warm up both variants, then alternate them: min over 5 (a box drifts by a few per cent within seconds)
        res = {1: [float("inf"), 0], 0: [float("inf"), 0]}
        for rep in range(6):
            for comp in (1, 0):
                ctx.set_option("composite_launches", comp)
                batch.synthesize_async(d_out, stride, d_len)
                ctx.sync()
                if rep:
                    res[comp][0] = min(res[comp][0], ctx.last_kernel_ms())
                res[comp][1] = ctx.get_option("last_launch_blocks")
        plan = G.plan_blocks(n, 96006, fast, 8 if presets else 4, warmup=warm, compute_units=cus)
        T[(n, fast)] = res[1][0]
        row.append(f"{'fast ' if fast else 'exact'} {res[1][0]:7.2f} ms in {res[1][1]} launch(es) "
                   f"[{' + '.join(f'{b.rows}:{b.family()}' for b in plan)}; model {sum(b.model_ms for b in plan):6.2f}]"
                   f" (single launch {res[0][0]:7.2f})")
    ctx.set_option("arithmetic", 0)
    ctx.set_option("composite_launches", 1)
    print("   ".join(row), flush=True)
    ctx.device_free(d_out)
    ctx.device_free(d_len)
    batch.free()
# the bound of VERDICT r3: T(n) <= floor(n / 65536) T(65536) + T_best(n mod 65536) + 0.3 ms
lanes = 256 * cus
for fast in (0, 1):
    if (lanes, fast) not in T:
        continue
    for n in sizes:
        if n <= lanes:
            continue
        rest = n % lanes
        best_rest = min([T[(m, fast)] for m in sizes if m >= rest and m <= lanes and (m, fast) in T] or [float("nan")]) if rest else 0.0
        bound = (n // lanes) * T[(lanes, fast)] + best_rest + 0.3
        print(f"# {'fast ' if fast else 'exact'} n={n:6d}: {T[(n, fast)]:7.2f} ms; bound {n // lanes} x {T[(lanes, fast)]:.2f} + "
              f"T_best({rest}) {best_rest:.2f} + 0.3 = {bound:7.2f}  {'ok' if T[(n, fast)] <= bound else 'ABOVE'}")
