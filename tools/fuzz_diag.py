#!/usr/bin/env python3
"""Replays one trial of the fast-mode fuzz tests (tests/test_fast_gpu.py) and says where the largest deviation
sits: fast vs reference (k), reference vs its own formulas in double precision (k0), fast vs double (k64), the
utterance / sample / voice, and that voice's formants.   usage: fuzz_diag.py lanes|split <seed> <trial> [lanes|chunks]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W
import oracle_lib as O

ULP = 2.0 ** -23
kind, seed, want = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
arg = int(sys.argv[4]) if len(sys.argv) > 4 else (1 if kind == "lanes" else 2)


def ov(voices):
    return [O.Voice.from_buffer_copy(bytes(v)) for v in voices]


def worst(a, b, lens):
    k, where = 0.0, None
    for u in range(len(lens)):
        n = int(lens[u])
        if n:
            d = np.abs(a[u, :n].astype(np.float64) - b[u, :n].astype(np.float64))
            i = int(d.argmax())
            if d[i] > k:
                k, where = float(d[i]), (u, i)
    return k / ULP, where


rng = np.random.default_rng(seed)
for trial in range(want + 1):
    voices = []
    if kind == "lanes":
        centres = [np.exp(rng.uniform(np.log(150.0), np.log(12000.0), 8)) for _ in range(3)]
    for i in range(3):
        centre = centres[i] if kind == "lanes" else np.exp(rng.uniform(np.log(150.0), np.log(12000.0), 8))
        v = G.voice_generic(48000.0)
        for p in range(2):
            freq, bw = centre * rng.uniform(0.65, 1.35, 8), rng.uniform(30, 600, 8)
            if trial % 3 != 2:
                bw = np.maximum(bw, freq / 30.0)
            e = G.elem_new_phoneme(freq, bw,
                                   rng.uniform(200, 4000, 8), rng.uniform(0, 1, 8), rng.uniform(0, 1, 8),
                                   rng.uniform(0.0, 1, 8) * (rng.uniform(0, 1, 8) > 0.3) + 1e-3)
            v.phonemes[p] = G.elem_resample(e, 44100.0, 48000.0)
        voices.append(v if trial % 3 == 2 else W.tame_voice(v))
    n_utt = 40 if kind == "lanes" else 70
    utts = []
    for u in range(n_utt):
        n = int(rng.integers(1, 5 if kind == "lanes" else 6))
        blends = [0.0625, 0.125, 0.25, 0.5, 1.0, 0.3, 0.07] if kind == "lanes" else [0.0625, 0.125, 0.25, 0.5, 0.3, 0.07]
        utts.append([(int(rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE])), float(rng.uniform(0.05, 0.3)),
                      float(rng.choice(blends)), float(rng.uniform(80, 400) / 48000.0)) for _ in range(n)])
    segs = G.segments([s for u in utts for s in u])
    offs = np.cumsum([0] + [len(u) for u in utts]).astype(np.uint32)
    vids = rng.integers(0, 3, n_utt).astype(np.uint32)
    seeds = rng.integers(0, 2 ** 32, n_utt, dtype=np.uint64).astype(np.uint32)
    if kind == "split":
        rng.integers(20000, 70000)
stride = 65536 if kind == "lanes" else 81920
ref, ref_len = O.synthesize_batch(ov(voices), segs, offs, vids, seeds, stride)
O.set_precise(True)
r64, _ = O.synthesize_batch(ov(voices), segs, offs, vids, seeds, stride)
O.set_precise(False)
ctx = G.Context(0)
ctx.set_voices(voices)
ctx.set_option("arithmetic", 1)
if kind == "lanes":
    ctx.set_option("lanes_per_utterance", arg)
else:
    ctx.set_option("time_split_chunks", arg)
out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
assert np.array_equal(out_len, ref_len)
scale = max(1.0, float(np.max(np.abs(ref))))
k, where = worst(out, ref, ref_len)
k0, w0 = worst(ref, r64, ref_len)
k64, _ = worst(out, r64, ref_len)
u, i = where
print(f"sharpness {[round(G.fast_sharpness(v), 1) for v in voices]} served {ctx.get_option('fast_arithmetic_served')}")
print(f"{ctx.last_kernel_name()}  scale {scale:.2f}: fast vs ref {k / scale:.1f}  ref vs double {k0 / scale:.1f}  "
      f"fast vs double {k64 / scale:.1f}   at utterance {u} sample {i} of {int(ref_len[u])} (voice {int(vids[u])}), "
      f"ref vs double there {abs(float(ref[u, i]) - float(r64[u, i])) / ULP / scale:.1f}, worst ref-vs-double at {w0}")
if kind == "split":      # the same batch through the lane kernel: how much of it is the time-split's own?
    ctx.set_option("time_split", 0)
    ctx.set_option("time_split_chunks", 0)
    ctx.set_option("time_parallel_scan", 0)
    ctx.set_option("lanes_per_utterance", 1)
    o2, _ = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    k2, w2 = worst(o2, ref, ref_len)
    print(f"  lane kernel ({ctx.last_kernel_name()}): fast vs ref {k2 / scale:.1f} at {w2}; at the time-split's worst sample "
          f"{abs(float(o2[u, i]) - float(ref[u, i])) / ULP / scale:.1f}")
# per-utterance profile of the worst row: error in blocks of 2048 samples
d = np.abs(out[u, :int(ref_len[u])].astype(np.float64) - ref[u, :int(ref_len[u])]) / ULP / scale
d0 = np.abs(ref[u, :int(ref_len[u])].astype(np.float64) - r64[u, :int(ref_len[u])]) / ULP / scale
print("  fast-ref per 2048:", " ".join(f"{d[j:j + 2048].max():.0f}" for j in range(0, len(d), 2048)))
print("  ref-dbl  per 2048:", " ".join(f"{d0[j:j + 2048].max():.0f}" for j in range(0, len(d0), 2048)))
print("  segments:", [(int(s["phoneme"]), round(float(s["length"]), 3), float(s["blend_length"])) for s in segs[offs[u]:offs[u + 1]]])
v = voices[int(vids[u])]
for p in range(2):
    e = v.phonemes[p]
    print("   f", np.round(np.array(e.formant_freq[:]) * 48000).astype(int), "bw", np.round(np.array(e.formant_bw[:]) * 48000).astype(int),
          "amp", np.round(np.array(e.formant_amp[:]), 3))
