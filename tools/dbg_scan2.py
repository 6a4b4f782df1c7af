#!/usr/bin/env python3
"""Chain quantities of the scan kernel (alpha, jitter phase, saw, carrier noise) against a numpy
float32 restatement of the reference chain.  (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W
f32 = np.float32
ctx = G.Context(0)
voices = W.single_voice()
ctx.set_voices(voices)
segs, offs, vids, seeds = W.make_batch(2)
stride = W.max_samples()
ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
u = 1
sg = segs[offs[u]:offs[u + 1]]
tr = O.trace_elems(ov[0], sg, int(seeds[u]), 1)   # jittered elems per sample
freq = tr[:, 0].astype(np.float32)
n = len(freq)
# reference carrier: phase, saw, noise
phase = f32(0); saw = np.zeros(n, dtype=np.float32); nz = np.zeros(n, dtype=np.float32); s = np.uint32(0)
with np.errstate(over="ignore"):
    for i in range(n):
        fr = freq[i]
        if phase < fr:
            t = f32(phase / fr); pb = f32(f32(f32(f32(2) * t) - f32(t * t)) - f32(1))
        elif phase > f32(f32(1) - fr):
            t = f32(f32(phase - f32(1)) / fr); pb = f32(f32(f32(t * t) + f32(f32(2) * t)) + f32(1))
        else:
            pb = f32(0)
        saw[i] = f32(f32(f32(f32(2) * phase) - f32(1)) - pb)
        phase = f32(phase + fr)
        if phase >= 1: phase = f32(phase - f32(1))
        s = np.uint32(s * np.uint32(16807) + np.uint32(1))
        nz[i] = f32(f32((np.uint32((s >> np.uint32(9)) | np.uint32(0x3F800000))).view(np.float32) - f32(1.5)) * f32(2))
fi = 0      # scan_debug modes 5 - 7 show formant 0: a1, g, v0 (1 - 4: alpha, jitter phase, saw, carrier noise)
x = tr[:, 1 + fi].astype(np.float64); bw = tr[:, 9 + fi].astype(np.float64); sm = tr[:, 17 + fi].astype(np.float64)
br = tr[:, 25 + fi].astype(np.float64); tb = tr[:, 33 + fi].astype(np.float64); am = tr[:, 41 + fi].astype(np.float64)
g = ((1 - x) * x * (5 - 4 * (x + .5) * (.5 - x))) / ((x + .5) * (5 - 4 * (1 - x) * x) * (.5 - x))
k = bw / x
a1 = 1 / (1 + g * (g + k))
lp = (1 - sm) ** 5
a_t = np.zeros(n); v0 = np.zeros(n)
a = 0.0
for i in range(n):
    nw = saw[i] * (1 - br[i]) + nz[i] * br[i]
    a = a + (1 - lp[i]) * (nw - a)
    a_t[i] = a
    v0[i] = a * ((1 - tb[i]) + nz[i] * tb[i]) * am[i]
ctx.set_option("arithmetic", 1)
for mode, name, ref in ((3, "saw", saw.astype(np.float64)), (4, "noise", nz.astype(np.float64)), (5, "a1.x", a1), (6, "tg.x", g), (7, "v0.x", v0)):
    ctx.set_option("scan_debug", mode)
    out, ol = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    d = np.abs(out[u, :n].astype(np.float64) - ref)
    bad = np.nonzero(d > 1e-5 * max(1e-3, np.abs(ref).max()))[0]
    print(name, "max diff", d.max(), "peak", np.abs(ref).max(), "first bad", bad[:6])
    if len(bad):
        i = int(bad[0]); lo = max(0, i - 2)
        print("   ref", ref[lo:lo + 6]); print("   out", out[u, lo:lo + 6])
