#!/usr/bin/env python3
"""VERDICT r3 item 4, the measurement before the kernel: how far from the reference is an arithmetic that rounds the
band-pass coefficients g, k, a1, a2, a3 (src/lib.rs:555-562) exactly as the reference does at every sample and is free
everywhere else?  CPU only (oracle/liboracle.so, modes 2 and 3 of orc_set_precise): the random one-voice tables and
corpora of tools/sharpness_data.py (same generator, same seeds), rendered by the oracle in its own arithmetic, with the
reference's coefficients + double precision elsewhere (the best such a tier can do), with the reference's coefficients +
binary32 FMA arithmetic elsewhere (what a kernel would run), and all in double precision (the reference's own rounding
noise).  One JSON line per table.   usage: middle_tier_experiment.py <n_tables> <seed> [first] > out.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G          # host-side parameter algebra only (elem_new_phoneme, resample, sharpness): no GPU
import oracle_lib as O

n_tables, seed = int(sys.argv[1]), int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
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
    if t < first:
        continue            # (the generator has to be stepped through the tables skipped)
    ov = [O.Voice.from_buffer_copy(bytes(v))]
    res = {}
    O.set_precise(0)
    ref, ref_len = O.synthesize_batch(ov, segs, offs, None, seeds, stride)
    scale = max(1.0, float(np.abs(ref).max()))
    for name, mode in (("coef_exact_rest_f64", 2), ("coef_exact_rest_f32_fma", 3), ("all_f64", 1)):
        O.set_precise(mode)
        out, out_len = O.synthesize_batch(ov, segs, offs, None, seeds, stride)
        assert np.array_equal(out_len, ref_len)
        res[name] = float(np.abs(out.astype(np.float64) - ref).max()) * 2.0 ** 23 / scale
    O.set_precise(0)
    res.update({"t": t, "seed": seed, "scale": scale, "sharpness": G.fast_sharpness(v)})
    print(json.dumps(res), flush=True)
