#!/usr/bin/env python3
"""Kernel time on a SPEECH-LIKE corpus: utterances of 8 - 32 phonemes of 40 - 160 ms each (blends of 30 - 80 ms, pitch
contours of 90 - 220 Hz), so that utterances differ in length by a factor of four and every lane of a wave has a segment
boundary every few thousand samples at a time of its own — next to the bench corpus (4 aligned segments of 0.5 s).
Reports samples/s over the samples actually rendered.
usage: speech_like_bench.py [n_utt] [--blend-is-length] [--lanes=L] [--two-waves=0|1] [--round16] [--no-split] [--no-ragged-plan] [--scale=F] [--long-tail]   (A/B:
pinned lane mapping, no time-split, the one-round launch policy; every length and blend length times F; one utterance in a hundred
of 60 - 80 phonemes among utterances of 4 - 12)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
ctx = G.Context(0)
scale = 1.0
for a in sys.argv[1:]:
    if a.startswith("--scale="):
        scale = float(a[8:])
    if a.startswith("--lanes="):
        ctx.set_option("lanes_per_utterance", int(a[8:]))
    if a.startswith("--two-waves="):
        ctx.set_option("two_waves_per_simd", int(a[12:]))
    if a == "--no-spread":
        ctx.set_option("pipeline_spread", 0)
    if a == "--round32":
        ctx.set_option("pipeline_round32", 2)
    if a == "--round16":
        ctx.set_option("pipeline_round32", 0)
    if a == "--no-split":
        ctx.set_option("time_split", 0)
    if a == "--no-ragged-plan":
        ctx.set_option("ragged_plan", 0)
rng = np.random.default_rng(7)
for n_voices in (1, 8):
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    ctx.set_voices(voices)
    segs, offs, vids, seeds, stride = W.speech_like_batch(n, rng, n_voices=n_voices, scale=scale,
                                                          blend_is_length="--blend-is-length" in sys.argv,
                                                          long_tail="--long-tail" in sys.argv)
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out = ctx.device_alloc(n * stride * 4)
    d_len = ctx.device_alloc(n * 4)
    for fast in (0, 1):
        ctx.set_option("arithmetic", fast)
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
        print(f"speech-like, {n_voices} voice(s), {n} utterances ({total / n / 48000:.2f} s on average, {lens.min() / 48000:.2f} .. {lens.max() / 48000:.2f} s), "
              f"{'fast ' if fast else 'exact'}: {min(ms):7.2f} ms = {total / (min(ms) * 1e-3):.3e} samples/s  ({ctx.last_kernel_name()}, "
              f"{ctx.get_option('last_launch_blocks')} block(s), L = {ctx.get_option('last_launch_lanes')})"
              + (f"  [per wave: {stats[0] * 64 / n:.0f} tight tiles, {stats[1] * 64 / n:.0f} general steps, longest row {lens.max() // 64} tiles]" if fast else ""), flush=True)
    ctx.set_option("arithmetic", 0)
    ctx.device_free(d_out)
    ctx.device_free(d_len)
    batch.free()
