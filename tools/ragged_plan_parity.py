#!/usr/bin/env python3
"""The speech-like corpus at full size, every utterance: what option "ragged_plan" changes is the kernel family, never a bit.
Exact arithmetic: per-row digests (sum of the samples' bit patterns, computed on the device) of the rendering with the
one-round launch policy against the rendering with the plan by the rows' lengths and events — equal for every row — and
sampled rows against the oracle, bit for bit.  Fast arithmetic asked for: every sample within GRAIL_FAST_TOLERANCE of the
exact rendering (relative to max(1, the row's peak)).   usage: ragged_plan_parity.py [n_utt]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = G.Context(0)
rng = np.random.default_rng(7)
for n_voices in (1, 8):
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    ctx.set_voices(voices)
    counts = rng.integers(8, 33, n)
    offs = np.zeros(n + 1, dtype=np.uint32)
    offs[1:] = np.cumsum(counts)
    k = int(offs[-1])
    segs = np.zeros(k, dtype=G.PHONEME_DTYPE)
    segs["phoneme"] = rng.choice([G.PH_A, G.PH_E, G.PH_SILENCE, G.PH_STOP], k, p=[.4, .4, .12, .08])
    segs["phoneme"][offs[:-1]] = G.PH_SILENCE
    segs["length"] = rng.uniform(0.04, 0.16, k).astype(np.float32)
    segs["blend_length"] = rng.uniform(0.03, 0.08, k).astype(np.float32)
    segs["frequency"] = (rng.uniform(90, 220, k) / 48000.0).astype(np.float32)
    vids = (np.arange(n) % n_voices).astype(np.uint32)
    seeds = np.arange(n, dtype=np.uint32)
    stride = (int(32 * 0.16 * 48000) + 64 + 63) // 64 * 64
    batch = ctx.upload(segs, offs, vids, seeds)
    d_a = ctx.device_alloc(n * stride * 4); d_b = ctx.device_alloc(n * stride * 4)
    l_a = ctx.device_alloc(n * 4); l_b = ctx.device_alloc(n * 4)
    # exact: one round against the ragged plan
    ctx.set_option("arithmetic", 0)
    ctx.set_option("ragged_plan", 0)
    batch.synthesize_async(d_a, stride, l_a); ctx.sync()
    name0, ms0 = f"{ctx.last_kernel_name()} x {ctx.get_option('last_launch_blocks')} launch(es)", ctx.last_kernel_ms()
    sums0, peak0, bad0 = ctx.digest(d_a, stride, l_a, n)
    ctx.set_option("ragged_plan", 1)
    batch.synthesize_async(d_b, stride, l_b); ctx.sync()
    name1, ms1 = f"{ctx.last_kernel_name()} x {ctx.get_option('last_launch_blocks')} launch(es)", ctx.last_kernel_ms()
    sums1, peak1, bad1 = ctx.digest(d_b, stride, l_b, n)
    lens0 = np.zeros(n, dtype=np.uint32); ctx.d2h(lens0, l_a, n * 4)
    lens1 = np.zeros(n, dtype=np.uint32); ctx.d2h(lens1, l_b, n * 4)
    same = int((sums0 == sums1).sum())
    print(f"voices={n_voices} n={n}: exact, {name0} ({ms0:.1f} ms) against {name1} ({ms1:.1f} ms): {same} of {n} row digests equal, "
          f"lengths equal: {bool(np.array_equal(lens0, lens1))}, non-finite samples {int(bad0.sum())} / {int(bad1.sum())}", flush=True)
    assert same == n and np.array_equal(lens0, lens1) and name0 != name1
    # sampled rows of the ragged-plan rendering against the oracle (the longest, the shortest, 46 others)
    order = np.argsort(lens1)
    rows = sorted(set([int(order[0]), int(order[-1])] + [int(x) for x in np.random.default_rng(1).choice(n, 46, replace=False)]))
    sub_offs = np.zeros(len(rows) + 1, dtype=np.uint32)
    sub = []
    for i, u in enumerate(rows):
        sub.append(segs[offs[u]:offs[u + 1]])
        sub_offs[i + 1] = sub_offs[i] + offs[u + 1] - offs[u]
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    ref, ref_len = O.synthesize_batch(ov, np.concatenate(sub), sub_offs, vids[rows], seeds[rows], stride)
    row = np.zeros(stride, dtype=np.float32)
    for i, u in enumerate(rows):
        ctx.d2h(row, d_b, stride * 4, offset=u * stride * 4)
        m = int(ref_len[i])
        assert lens1[u] == m and np.array_equal(row[:m].view(np.uint32), ref[i, :m].view(np.uint32)), u
    print(f"voices={n_voices} n={n}: {len(rows)} rows (the longest, the shortest, 46 drawn) against the oracle: bit-identical", flush=True)
    # fast arithmetic asked for (ragged plan) against the exact rendering
    ctx.set_option("arithmetic", 1)
    batch.synthesize_async(d_a, stride, l_a); ctx.sync()
    namef, msf, served = ctx.last_kernel_name(), ctx.last_kernel_ms(), ctx.get_option("last_launch_fast")
    ctx.set_option("arithmetic", 0)
    md, sq, bad = ctx.compare(d_b, d_a, stride, l_b, l_a, n)
    rel = md.astype(np.float64) / np.maximum(1.0, peak1.astype(np.float64))
    print(f"voices={n_voices} n={n}: fast asked for, {namef} ({msf:.1f} ms, last_launch_fast = {served}): max |fast - exact| = "
          f"{float(rel.max()) * 2.0 ** 23:.1f} * 2^-23 relative to max(1, peak), structural mismatches {int(bad.sum())}", flush=True)
    assert int(bad.sum()) == 0 and float(rel.max()) <= G.FAST_TOLERANCE
    for d in (d_a, d_b, l_a, l_b):
        ctx.device_free(d)
    batch.free()
print("ragged plan parity: ok")
