#!/usr/bin/env python3
"""Fits the ragged planner's model of the tolerance-mode lane kernels (launch_plan.cpp ragged_wave_ms fast branch, and the
two-waves-per-SIMD simulation of ragged_cost) to pinned-mapping measurements (tools/ragged_fit_collect.sh, once with option
"two_waves_per_simd" = 0 and once with 1).  Model of one wave alone on its SIMD: its longest row's samples at the mapping's
aligned rate x m, plus c per event (segment boundaries and kinks of alpha) of its rows.  One wave per SIMD: waves go to the
SIMD that falls free first.  Two per SIMD: while two are resident each advances at 1 / (2 g) of the lone pace.
usage: ragged_fit.py one_wave.txt [two_waves.txt]      (no GPU needed: the corpora are regenerated from their seed)"""
import heapq
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
from grail_hip import workload as W


def parse(path):
    text = open(path).read()
    aligned, meas = {}, {}
    for m in re.finditer(r"aligned voices=(\d) L=(\d) n=\d+ fast\s+([\d.]+) ms", text):
        aligned[(int(m.group(1)), int(m.group(2)))] = float(m.group(3))
    blocks = re.split(r"# scale ([\d.]+) lanes (\d)\n", text)
    for i in range(1, len(blocks) - 2, 3):
        sc, L, body = float(blocks[i]), int(blocks[i + 1]), blocks[i + 2]
        for m in re.finditer(r"speech-like, (\d) voice\(s\).*?fast :\s+([\d.]+) ms", body):
            meas[(int(m.group(1)), L, sc)] = float(m.group(2))
    return aligned, meas


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


def wave_times(samples, events, L, rate, m, c):
    per = 64 // L
    return [(samples[w] + 64.0) * rate * m + events[w:w + per].sum() * c for w in range(0, len(samples), per)]


def makespan_one(times, simds=1024):
    free, span = [], 0.0
    for t in times:
        if len(free) >= simds:
            t += heapq.heappop(free)
        heapq.heappush(free, t)
        span = max(span, t)
    return span


def makespan_two(times, g, simds=1024):
    if len(times) <= simds:
        return makespan_one(times, simds)
    pace2 = 1.0 / (2.0 * g)
    sm = [[0.0, -1.0, -1.0] for _ in range(simds)]     # time, a (shorter), b

    def put(m, w):
        if m[2] < 0:
            m[2] = w
        elif w <= m[2]:
            m[1] = w
        else:
            m[1], m[2] = m[2], w

    def nxt(m):
        return m[0] + m[2] if m[1] < 0 else m[0] + m[1] / pace2

    k = 0
    while k < len(times) and k < 2 * simds:
        put(sm[k % simds], times[k])
        k += 1
    heap = [(nxt(m), i) for i, m in enumerate(sm) if m[2] >= 0]
    heapq.heapify(heap)
    span = 0.0
    while heap:
        _, i = heapq.heappop(heap)
        m = sm[i]
        if m[1] < 0:
            m[0] += m[2]
            m[2] = -1.0
        else:
            m[0] += m[1] / pace2
            m[2] -= m[1]
            m[1] = -1.0
        span = max(span, m[0])
        if k < len(times):
            put(m, times[k])
            k += 1
        if m[2] >= 0:
            heapq.heappush(heap, (nxt(m), i))
    return span


aligned, one = parse(sys.argv[1])
two = parse(sys.argv[2])[1] if len(sys.argv) > 2 else {}
corp = {sc: rows_of(sc) for sc in (1.0, 0.4, 0.1)}
print("# mapping: aligned ms per round; fitted m (rate multiplier on a wave with events), c (per event), g (two waves per SIMD: the pair's "
      "time over twice the lone wave's); model / measured per phoneme scale, one wave per SIMD | two")
for nv in (1, 8):
    for L in (1, 2, 4, 8):
        rate = aligned[(nv, L)] / 96006.0
        best = None
        for m in np.arange(1.0, 1.61, 0.02):
            for c in np.arange(0.0, 0.0121, 0.00025):
                err = 0.0
                for sc in (1.0, 0.4, 0.1):
                    s, sg, k = corp[sc][nv]
                    mod = makespan_one(wave_times(s, (sg + k).astype(np.float64), L, rate, m, c))
                    err += np.log(mod / one[(nv, L, sc)]) ** 2
                if best is None or err < best[0]:
                    best = (err, m, c)
        _, m, c = best
        line = "   ".join(f"x{sc}: {makespan_one(wave_times(corp[sc][nv][0], (corp[sc][nv][1] + corp[sc][nv][2]).astype(np.float64), L, rate, m, c)):6.2f} / {one[(nv, L, sc)]:6.2f}"
                          for sc in (1.0, 0.4, 0.1))
        gtxt = ""
        if two and L >= 2 and not (L == 2 and nv == 8):
            bg = None
            for g in np.arange(0.50, 1.001, 0.01):
                err = 0.0
                for sc in (1.0, 0.4, 0.1):
                    s, sg, k = corp[sc][nv]
                    mod = makespan_two(wave_times(s, (sg + k).astype(np.float64), L, rate, m, c), g)
                    err += np.log(mod / two[(nv, L, sc)]) ** 2
                if bg is None or err < bg[0]:
                    bg = (err, g)
            g = bg[1]
            gtxt = f"  | g = {g:.2f}  " + "   ".join(
                f"x{sc}: {makespan_two(wave_times(corp[sc][nv][0], (corp[sc][nv][1] + corp[sc][nv][2]).astype(np.float64), L, rate, m, c), g):6.2f} / {two[(nv, L, sc)]:6.2f}"
                for sc in (1.0, 0.4, 0.1))
        print(f"voices {nv} L {L}: aligned {aligned[(nv, L)]:6.2f}  m = {m:.2f}  c = {c * 1000:.2f} us   " + line + gtxt)
