#!/usr/bin/env python3
"""Fits the ragged planner's model of the tolerance-mode lane kernels (launch_plan.cpp ragged_wave_ms, fast branch) to
pinned-mapping measurements (tools/ragged_fit_collect.sh).  Model of one wave: its longest row's samples at the mapping's
aligned rate x m, plus c per event (segment boundaries and kinks of alpha) of its rows; waves go to the SIMD that falls free
first.  usage: ragged_fit.py gpurun_out/ragged_fit.txt      (no GPU needed: the corpora are regenerated from their seed)"""
import heapq
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
from grail_hip import workload as W

text = open(sys.argv[1]).read()
aligned = {}
for m in re.finditer(r"aligned voices=(\d) L=(\d) n=\d+ fast\s+([\d.]+) ms", text):
    aligned[(int(m.group(1)), int(m.group(2)))] = float(m.group(3))
meas = {}
blocks = re.split(r"# scale ([\d.]+) lanes (\d)\n", text)
for i in range(1, len(blocks) - 2, 3):
    sc, L, body = float(blocks[i]), int(blocks[i + 1]), blocks[i + 2]
    for m in re.finditer(r"speech-like, (\d) voice\(s\).*?fast :\s+([\d.]+) ms", body):
        meas[(int(m.group(1)), L, sc)] = float(m.group(2))


def rows_of(scale):
    rng = np.random.default_rng(7)
    out = {}
    for nv in (1, 8):
        segs, offs, vids, seeds, stride = W.speech_like_batch(65536, rng, n_voices=nv, scale=scale)
        secs = np.add.reduceat(segs["length"].astype(np.float64), offs[:-1])
        n_seg = np.diff(offs.astype(np.int64))
        kinks = np.add.reduceat((segs["blend_length"] < segs["length"]).astype(np.int64), offs[:-1])
        order = np.argsort(-secs, kind="stable")
        out[nv] = (secs[order] * 48000.0, n_seg[order], kinks[order])
    return out


def makespan(samples, events, L, rate, m, c, simds=1024):
    per = 64 // L
    n = len(samples)
    free = []
    span = 0.0
    for w in range(0, n, per):
        t = (samples[w] + 64.0) * rate * m + events[w:w + per].sum() * c
        if len(free) >= simds:
            t += heapq.heappop(free)
        heapq.heappush(free, t)
        span = max(span, t)
    return span


corp = {sc: rows_of(sc) for sc in (1.0, 0.4, 0.1)}
print("# mapping: aligned ms per round; fitted m (rate multiplier on a wave with events), c (ms per event); model / measured per scale")
for nv in (1, 8):
    for L in (1, 2, 4, 8):
        rate = aligned[(nv, L)] / 96006.0
        best = None
        for m in np.arange(1.0, 1.81, 0.02):
            for c in np.arange(0.0, 0.0121, 0.00025):
                err = 0.0
                for sc in (1.0, 0.4, 0.1):
                    s, g, k = corp[sc][nv]
                    mod = makespan(s, (g + k).astype(np.float64), L, rate, m, c)
                    err += (np.log(mod / meas[(nv, L, sc)])) ** 2
                if best is None or err < best[0]:
                    best = (err, m, c)
        _, m, c = best
        line = []
        for sc in (1.0, 0.4, 0.1):
            s, g, k = corp[sc][nv]
            mod = makespan(s, (g + k).astype(np.float64), L, rate, m, c)
            line.append(f"x{sc}: {mod:6.2f} / {meas[(nv, L, sc)]:6.2f}")
        print(f"voices {nv} L {L}: aligned {aligned[(nv, L)]:6.2f}  m = {m:.2f}  c = {c * 1000:.2f} us   " + "   ".join(line))
