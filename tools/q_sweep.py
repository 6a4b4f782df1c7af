#!/usr/bin/env python3
"""Fast-mode deviation against the sharpness of the formant resonances: the 8 preset voices (and the generic voice)
with every bandwidth divided by 1, 2, 4, 8 (Q = frequency / bandwidth up to 48, 97, 194, 387), 256 utterances x 2 s,
fast against exact rendering through each fast kernel family, in units of 2^-23 relative to max(1, peak), next to
grail_fast_sharpness() — the library's prediction, up to which ("fast_sharpness_limit", default 32) it serves fast
arithmetic at all; the sweep lifts the limit to measure beyond it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

ctx = G.Context(0)
ctx.set_option("fast_sharpness_limit", 1 << 30)      # measure the fast kernels beyond what the library would serve
stride = W.max_samples()
n = 256
for nv in (8, 1):
    for div in (1.0, 1.5, 2.0, 3.0, 4.0, 8.0):
        voices = W.preset_voices(8) if nv == 8 else W.single_voice()
        qmax = 0.0
        for v in voices:
            for p in range(2):
                e = v.phonemes[p]
                for i in range(8):
                    e.formant_bw[i] = e.formant_bw[i] / div
                    if e.formant_amp[i] != 0.0:
                        qmax = max(qmax, e.formant_freq[i] / e.formant_bw[i])
        ctx.set_voices(voices)
        segs, offs, vids, seeds = W.make_batch(n, n_voices=nv)
        ctx.set_option("arithmetic", 0)
        ref, ref_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        scale = max(1.0, float(np.abs(ref).max()))
        sharp = max(G.fast_sharpness(v) for v in voices)
        row = [f"voices={nv} bandwidths / {div:3.1f}  Q <= {qmax:5.1f}  sharpness {sharp:6.1f}"
               f"{' (served)' if sharp <= G.FAST_SHARPNESS_LIMIT else ' (exact kernels by default)'}  peak {scale:5.2f}:"]
        for label, opts in (("scan", {}), ("lanes=1", {"lanes_per_utterance": 1}), ("lanes=8", {"lanes_per_utterance": 8}),
                            ("split x8", {"time_split_chunks": 8})):
            ctx.set_option("arithmetic", 1)
            for k, val in opts.items():
                ctx.set_option(k, val)
            out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
            name = ctx.last_kernel_name()
            for k in opts:
                ctx.set_option(k, 0)
            assert np.array_equal(out_len, ref_len)
            k = float(np.abs(out.astype(np.float64) - ref).max()) * 2.0 ** 23 / scale
            row.append(f"{label} {k:6.1f}" + ("" if ("FAST" in name) else " (exact kernel)"))
        print("  ".join(row), flush=True)

# ---- one formant at a time: frequency x bandwidth, carrying all or a quarter of the amplitude
print("single formant (the other amplitude, if any, on a 1 kHz / 400 Hz formant); random segment lists, lanes=1:")
rng = np.random.default_rng(5)
utts = []
for u in range(64):
    utts.append([(int(rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE])), float(rng.uniform(0.05, 0.3)),
                  float(rng.choice([0.0625, 0.125, 0.25, 0.5, 1.0, 0.3, 0.07])),
                  float(rng.uniform(80, 400) / 48000.0)) for _ in range(int(rng.integers(1, 5)))])
segs = G.segments([s for u in utts for s in u])
offs = np.cumsum([0] + [len(u) for u in utts]).astype(np.uint32)
seeds = rng.integers(0, 2 ** 32, 64, dtype=np.uint64).astype(np.uint32)
for share in (1.0, 0.25):
    for f in (150, 300, 600, 1200, 2400, 4800, 9600, 15000):
        row = []
        for bw in (30, 60, 120, 240):
            v = G.voice_generic(48000.0)
            for p in range(2):
                e = G.elem_new_phoneme(np.array([f * (1.0 if p == 0 else 1.1)] + [1000.0] * 7), np.array([bw] + [400.0] * 7),
                                       np.full(8, 1600.0), np.full(8, 0.3), np.full(8, 0.3),
                                       np.array([share, 1.0 - share] + [0.0] * 6))
                v.phonemes[p] = G.elem_resample(e, 44100.0, 48000.0)
            ctx.set_voices([v])
            ctx.set_option("arithmetic", 0)
            ref, ref_len = ctx.synthesize(segs, offs, None, seeds, out_stride=65536)
            ctx.set_option("arithmetic", 1)
            ctx.set_option("lanes_per_utterance", 1)
            out, out_len = ctx.synthesize(segs, offs, None, seeds, out_stride=65536)
            ctx.set_option("lanes_per_utterance", 0)
            scale = max(1.0, float(np.abs(ref).max()))
            k = float(np.abs(out.astype(np.float64) - ref).max()) * 2.0 ** 23 / scale
            row.append(f"bw {bw:3d} Hz: {k:6.1f} (predicted {G.fast_sharpness(v):6.1f})")
        print(f"  share {share:4.2f}  f {f:5d} Hz   " + "   ".join(row), flush=True)
