#!/usr/bin/env python3
"""The interpolation guard of the tolerance mode under the input that strains it: RANDOM voice tables at the served sharpness
(formants anywhere in 150 Hz .. 12 kHz, Q <= 30, every amplitude / turbulence / breath pattern; bandwidths widened until
grail_fast_sharpness <= the limit) rendered on speech-like segment lists with blends of 3 - 80 ms — every lane's amplitudes,
turbulence and formant frequencies ramping steeply every few hundred samples — on every lane mapping, against the oracle.
Prints the worst |fast - reference| per phoneme scale in units of 2^-23 of max(1, peak) (the contract: 64).
usage: steep_blend_soak.py [tables [seed]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

tables = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = G.Context(0)
ctx.set_option("arithmetic", 1)
ctx.set_option("time_split", 0)
ctx.set_option("time_parallel_scan", 0)
ctx.set_option("ragged_plan", 0)
worst = {}
for t in range(tables):
    voices = []
    centres = np.exp(rng.uniform(np.log(150.0), np.log(12000.0), 8))
    for _ in range(3):
        v = G.voice_generic(48000.0)
        for p in range(2):
            freq = centres * rng.uniform(0.65, 1.35, 8)
            bw = np.maximum(rng.uniform(30, 600, 8), freq / 30.0)
            e = G.elem_new_phoneme(freq, bw, rng.uniform(200, 4000, 8), rng.uniform(0, 1, 8), rng.uniform(0, 1, 8),
                                   rng.uniform(0.0, 1, 8) * (rng.uniform(0, 1, 8) > 0.3) + 1e-3)
            v.phonemes[p] = G.elem_resample(e, 44100.0, 48000.0)
        voices.append(W.tame_voice(v))
    ctx.set_voices(voices)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    for scale in (0.1, 0.25, 1.0):
        n = 96
        segs, offs, vids, seeds, stride = W.speech_like_batch(n, rng, n_voices=3, scale=scale)
        ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, stride)
        for lanes in (1, 2, 4, 8):
            ctx.set_option("lanes_per_utterance", lanes)
            out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            assert "FAST" in ctx.last_kernel_name(), ctx.last_kernel_name()
            assert np.array_equal(out_len, ref_len)
            for u in range(n):
                m = int(ref_len[u])
                if m:
                    d = float(np.max(np.abs(out[u, :m].astype(np.float64) - ref[u, :m]))) / max(1.0, float(np.max(np.abs(ref[u, :m]))))
                    worst[scale] = max(worst.get(scale, 0.0), d * 2.0 ** 23)
    print(f"table {t}: worst so far " + "  ".join(f"x{s}: {w:.1f}" for s, w in sorted(worst.items())), flush=True)
print("steep blends on random voice tables at the served sharpness: worst |fast - reference| / 2^-23 = " +
      ", ".join(f"{w:.1f} (phonemes x {s})" for s, w in sorted(worst.items())) + "; contract 64")
