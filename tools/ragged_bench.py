#!/usr/bin/env python3
"""Kernel time on a RAGGED corpus: per-segment lengths drawn from [0.3, 0.7] s and (optionally)
per-voice jitter rates that differ, so segment boundaries and jitter wraps of the 64 utterances of
a wave do not coincide (the bench corpus has them all aligned).
usage: ragged_bench.py [n_utt [sort_by_length [arithmetic [time_split_chunks]]]] [--lanes=L] [--no-ragged-plan]   (A/B)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

pin = [a for a in sys.argv[1:] if a.startswith("--lanes=")]
no_ragged_plan = "--no-ragged-plan" in sys.argv
sys.argv = [a for a in sys.argv if not a.startswith("--")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sort = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fast = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = G.Context(0)
ctx.set_option("sort_by_length", sort)
ctx.set_option("arithmetic", fast)
if pin:
    ctx.set_option("lanes_per_utterance", int(pin[0][8:]))
if no_ragged_plan:
    ctx.set_option("ragged_plan", 0)
if len(sys.argv) > 4:
    ctx.set_option("time_split_chunks", int(sys.argv[4]))      # 0 = the library's own choice
print(f"n = {n} utterances, sort_by_length = {sort}, arithmetic = {'fast' if fast else 'exact'}", flush=True)
rng = np.random.default_rng(1)
for label, ragged_len, ragged_jit, n_voices in (("aligned 1 voice", False, False, 1),
                                                 ("ragged lengths, 1 voice", True, False, 1),
                                                 ("aligned 8 presets", False, False, 8),
                                                 ("ragged lengths, 8 presets", True, False, 8),
                                                 ("ragged lengths + jitter rates, 8 presets", True, True, 8)):
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    if ragged_jit:
        for i, v in enumerate(voices):
            v.jitter_frequency = np.float32((12.0 + i) / 48000.0)
    ctx.set_voices(voices)
    segs, offs, vids, seeds = W.make_batch(n, n_voices=n_voices)
    if ragged_len:
        segs["length"] = rng.uniform(0.3, 0.7, len(segs)).astype(np.float32)
    stride = (int(0.7 * 4 * 48000) + 64 + 63) // 64 * 64
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out = ctx.device_alloc(n * stride * 4)
    d_len = ctx.device_alloc(n * 4)
    ms = []
    for _ in range(3):
        t0, g0 = ctx.get_option("fast_wave_tiles"), ctx.get_option("general_wave_steps")
        batch.synthesize_async(d_out, stride, d_len)
        ctx.sync()
        ms.append(ctx.last_kernel_ms())
        stats = (ctx.get_option("fast_wave_tiles") - t0, ctx.get_option("general_wave_steps") - g0)
    lens = np.zeros(n, dtype=np.uint32)
    ctx.d2h(lens, d_len, lens.nbytes)
    total = int(lens.astype(np.uint64).sum())
    print(f"{label:45s} kernel {min(ms):7.2f} ms  {total / (min(ms) * 1e-3):.3e} samples/s  "
          f"(max row {int(lens.max())} samples, mean {total / n:.0f})  {ctx.last_kernel_name()}"
          + (f"  [per wave: {stats[0] * 64 / n:.0f} tight tiles, {stats[1] * 64 / n:.0f} general steps]" if fast else ""), flush=True)
    ctx.device_free(d_out)
    ctx.device_free(d_len)
    batch.free()
