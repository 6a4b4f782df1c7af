#!/usr/bin/env python3
"""Experiment: a ragged batch rendered as TWO kernels side by side — the longest rows on a wide lane mapping, the rest on one lane
each — on two streams (two contexts of one device), against one launch of one mapping.  Wall time around both launches (host clock
after a device-wide sync: +- 0.1 ms).   usage: two_stream_experiment.py [n_utt]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(7)
a, b = G.Context(0), G.Context(0)
for n_voices in (1, 8):
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    a.set_voices(voices); b.set_voices(voices)
    counts = rng.integers(8, 33, n)
    offs = np.zeros(n + 1, dtype=np.uint32); offs[1:] = np.cumsum(counts)
    k = int(offs[-1])
    segs = np.zeros(k, dtype=G.PHONEME_DTYPE)
    segs["phoneme"] = rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE, G.PH_STOP], k, p=[.4, .4, .12, .08])
    segs["phoneme"][offs[:-1]] = G.PH_SILENCE
    segs["length"] = rng.uniform(0.04, 0.16, k).astype(np.float32)
    segs["blend_length"] = rng.uniform(0.03, 0.08, k).astype(np.float32)
    segs["frequency"] = (rng.uniform(90, 220, k) / 48000.0).astype(np.float32)
    secs = np.add.reduceat(segs["length"].astype(np.float64), offs[:-1])
    order = np.argsort(-secs, kind="stable")
    stride = (int(32 * 0.16 * 48000) + 64 + 63) // 64 * 64
    d_out = a.device_alloc(n * stride * 4); d_len = a.device_alloc(n * 4)

    def sub(rows):
        o = np.zeros(len(rows) + 1, dtype=np.uint32)
        o[1:] = np.cumsum(counts[rows])
        s = np.concatenate([segs[offs[u]:offs[u + 1]] for u in rows])
        return s, o, (rows % n_voices).astype(np.uint32), rows.astype(np.uint32)

    whole = a.upload(*sub(order))
    for fast in (0, 1):
        a.set_option("arithmetic", fast); b.set_option("arithmetic", fast)
        ms = []
        for _ in range(3):
            whole.synthesize_async(d_out, stride, d_len); a.sync(); ms.append(a.last_kernel_ms())
        print(f"{n_voices} voice(s), {n} utterances, {'fast' if fast else 'exact'}: one launch {min(ms):6.1f} ms  ({a.last_kernel_name()})", flush=True)
        for frac, la, lb in ((0.25, 2, 1), (0.375, 2, 1), (0.5, 2, 1), (0.125, 4, 1), (0.25, 4, 2), (0.125, 4, 2)):
            cut = int(n * frac) // 64 * 64
            ba, bb = a.upload(*sub(order[:cut])), b.upload(*sub(order[cut:]))
            a.set_option("lanes_per_utterance", la); b.set_option("lanes_per_utterance", lb)
            best = 1e9
            for _ in range(3):
                a.sync(); b.sync()
                t = time.perf_counter()
                ba.synthesize_async(d_out, stride, d_len)
                bb.synthesize_async(a.ptr_add(d_out, cut * stride * 4) if hasattr(a, "ptr_add") else type(d_out)(d_out.value + cut * stride * 4),
                                    stride, type(d_len)(d_len.value + cut * 4))
                a.sync(); b.sync()
                best = min(best, (time.perf_counter() - t) * 1e3)
            print(f"    longest {cut:6d} rows on {la} lanes beside {n - cut:6d} on {lb}: {best:6.1f} ms wall  (alone: {a.last_kernel_ms():.1f} + {b.last_kernel_ms():.1f})", flush=True)
            a.set_option("lanes_per_utterance", 0); b.set_option("lanes_per_utterance", 0)
            ba.free(); bb.free()
    a.set_option("arithmetic", 0); b.set_option("arithmetic", 0)
    whole.free()
    a.device_free(d_out); a.device_free(d_len)
