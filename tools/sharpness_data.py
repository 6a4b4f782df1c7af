#!/usr/bin/env python3
"""Data behind grail_fast_sharpness(): random one-voice tables (the fuzz tests' generator), each rendered in exact
and in fast arithmetic (sharpness limit lifted) on a corpus of random segment lists; one line per table with the
measured deviation (units of 2^-23 of max(1, peak)) and, per formant of both phonemes, frequency, bandwidth,
amplitude.   usage: sharpness_data.py <n_tables> <seed> [arithmetic] > table.jsonl
(arithmetic 1, the default: the interpolating tier; 2: the reference's own coefficients at every sample, MID)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G

n_tables, seed = int(sys.argv[1]), int(sys.argv[2])
arithmetic = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed)
ctx = G.Context(0)
ctx.set_option("fast_sharpness_limit", 1 << 30)
stride = 65536
for t in range(n_tables):
    centre = np.exp(rng.uniform(np.log(150.0), np.log(12000.0), 8))
    v = G.voice_generic(48000.0)
    lo = float(rng.choice([30.0, 50.0, 80.0]))
    for p in range(2):
        freq, bw = centre * rng.uniform(0.65, 1.35, 8), rng.uniform(lo, 600, 8)
        if t % 2 == 0:
            bw = np.maximum(bw, freq / float(rng.choice([20.0, 30.0, 50.0, 80.0])))
        e = G.elem_new_phoneme(freq, bw, rng.uniform(200, 4000, 8), rng.uniform(0, 1, 8), rng.uniform(0, 1, 8),
                               rng.uniform(0.0, 1, 8) * (rng.uniform(0, 1, 8) > 0.3) + 1e-3)
        v.phonemes[p] = G.elem_resample(e, 44100.0, 48000.0)
    ctx.set_voices([v])
    n_utt = 48
    utts = []
    for u in range(n_utt):
        n = int(rng.integers(1, 5))
        utts.append([(int(rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE])), float(rng.uniform(0.05, 0.3)),
                      float(rng.choice([0.0625, 0.125, 0.25, 0.5, 1.0, 0.3, 0.07])),
                      float(rng.uniform(80, 400) / 48000.0)) for _ in range(n)])
    segs = G.segments([s for u in utts for s in u])
    offs = np.cumsum([0] + [len(u) for u in utts]).astype(np.uint32)
    seeds = rng.integers(0, 2 ** 32, n_utt, dtype=np.uint64).astype(np.uint32)
    ctx.set_option("arithmetic", 0)
    ref, ref_len = ctx.synthesize(segs, offs, None, seeds, out_stride=stride)
    ctx.set_option("arithmetic", arithmetic)
    ctx.set_option("lanes_per_utterance", 1)
    out, out_len = ctx.synthesize(segs, offs, None, seeds, out_stride=stride)
    assert ("MID" in ctx.last_kernel_name()) == (arithmetic == 2), ctx.last_kernel_name()
    ctx.set_option("lanes_per_utterance", 0)
    assert np.array_equal(out_len, ref_len)
    scale = max(1.0, float(np.abs(ref).max()))
    k = float(np.abs(out.astype(np.float64) - ref).max()) * 2.0 ** 23 / scale
    row = {"k": k, "scale": scale, "sharpness": G.fast_sharpness(v),
           "phonemes": [{"f": [float(x) for x in v.phonemes[p].formant_freq], "bw": [float(x) for x in v.phonemes[p].formant_bw],
                         "amp": [float(x) for x in v.phonemes[p].formant_amp]} for p in range(2)]}
    print(json.dumps(row), flush=True)
