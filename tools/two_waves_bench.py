#!/usr/bin/env python3
"""Two waves per SIMD (option "two_waves_per_simd"), same box, same library: kernel ms of pinned lane mappings with the
option off and on — the bench corpus (65 536 aligned utterances of 2 s) and the speech-like corpus (phonemes of 40 - 160 ms
and of 4 - 16 ms), exact and tolerance arithmetic, one voice (four formants laid out) and eight presets (eight).
usage: two_waves_bench.py [n_utt]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = G.Context(0)
ctx.set_option("time_split", 0)
ctx.set_option("time_parallel_scan", 0)


def corpus(kind, nv):
    if kind == "aligned":
        return W.make_batch(n, n_voices=nv) + (W.max_samples(),)
    return W.speech_like_batch(n, np.random.default_rng(7), n_voices=nv, scale=0.1 if kind.endswith("4 - 16 ms") else 1.0)


print(f"# {n} utterances; kernel ms, best of 3: one wave per SIMD (rounds in turn) -> two waves per SIMD   [kernel of the latter]")
for kind in ("aligned", "speech-like, phonemes of 40 - 160 ms", "speech-like, phonemes of 4 - 16 ms"):
    for nv in (1, 8):
        ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
        segs, offs, vids, seeds, stride = corpus(kind, nv)
        batch = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * 4)
        d_len = ctx.device_alloc(n * 4)
        for fast in (0, 1):
            ctx.set_option("arithmetic", fast)
            for lanes in (2, 4, 8):
                if lanes == 8 and nv == 1:
                    continue
                ctx.set_option("lanes_per_utterance", lanes)
                res = []
                for tw in (0, 1):
                    ctx.set_option("two_waves_per_simd", tw)
                    ms = []
                    for _ in range(3):
                        batch.synthesize_async(d_out, stride, d_len)
                        ctx.sync()
                        ms.append(ctx.last_kernel_ms())
                    res.append((min(ms), ctx.last_kernel_name()))
                built = ",2," in res[1][1]
                print(f"{kind:38s} {nv} voice(s) {'fast ' if fast else 'exact'} {lanes} lanes: {res[0][0]:7.2f} -> {res[1][0]:7.2f} ms"
                      + (f"  = {res[1][0] / res[0][0]:.2f}   {res[1][1]}" if built else "   (no two-wave instantiation)"), flush=True)
        ctx.set_option("arithmetic", 0)
        ctx.set_option("lanes_per_utterance", 0)
        ctx.set_option("two_waves_per_simd", 1)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()
