#!/usr/bin/env python3
"""How the hardware hands one-wave workgroups to SIMDs (tools/dispatch_order.hip), against the list-scheduling model of
launch_plan.cpp ("waves are handed to the SIMDs in launch order as they fall free").  Workgroups spin for the time the waves
of a speech-like batch would take (scaled), launched (a) longest first and (b) in the packed order of
tools/packed_order_experiment.py; the records say which XCC / CU / SIMD ran each workgroup, from when to when.
Reports: workgroup -> XCC mapping; how many workgroups ran per SIMD; the realised makespan against the dispatcher model's
(packed_order_experiment.dispatch_makespan); for every hand-over after the first round how long the SIMD had been idle; how
many workgroups started out of launch order.  DISPATCH_ORDER_SAVE=dir keeps the records (which XCC / SE / CU / SIMD, when).
usage (GPU box): python3 tools/dispatch_order.py [n_utt ...]"""
import heapq
import importlib.util
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
from grail_hip import workload as W

spec = importlib.util.spec_from_file_location("packed", os.path.join(ROOT, "tools", "packed_order_experiment.py"))
packed = importlib.util.module_from_spec(spec)
spec.loader.exec_module(packed)
BIN = os.path.join(ROOT, "tools", "dispatch_order.bin")


def greedy(seq, slots):
    s = [0.0] * slots
    heapq.heapify(s)
    for c in seq:
        heapq.heappush(s, heapq.heappop(s) + c)
    return max(s)


def ensure_built():
    src = os.path.join(ROOT, "tools", "dispatch_order.hip")
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", src, "-o", BIN])


def run(cost_us, tag, waves=1):
    ensure_built()
    with tempfile.TemporaryDirectory() as d:
        cf, rf = os.path.join(d, "c.u32"), os.path.join(d, "r.u64")
        np.asarray(cost_us, dtype=np.uint32).tofile(cf)
        subprocess.check_call([BIN, cf, rf, str(waves)])
        rec = np.fromfile(rf, dtype=np.uint64).reshape(-1, 4)
    if os.environ.get("DISPATCH_ORDER_SAVE"):
        np.savez_compressed(os.path.join(os.environ["DISPATCH_ORDER_SAVE"], tag.replace(" ", "_").replace(",", "") + ".npz"),
                            rec=rec, cost_us=np.asarray(cost_us))
    t0 = rec[:, 0].astype(np.int64)
    t1 = rec[:, 1].astype(np.int64)
    base = t0.min()
    start, end = (t0 - base) / 100.0, (t1 - base) / 100.0            # microseconds
    hw, xcc = rec[:, 2].astype(np.int64), rec[:, 3].astype(np.int64) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    slot = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + (simd if waves == 1 else 0)
    n = len(cost_us)
    slots = len(np.unique(slot))
    by_xcc = [np.flatnonzero(xcc == x) for x in range(8)]
    rr = all(np.array_equal(b % 8, np.full(len(b), b[0] % 8)) for b in by_xcc if len(b))
    per_slot = np.bincount(np.unique(slot, return_inverse=True)[1])
    # hand-overs: for every workgroup that started after the first round, the idle gap of its SIMD and the number of
    # workgroups with a smaller index (on the same XCC) that started later
    order_t = np.argsort(start, kind="stable")
    late = 0
    for x in range(8):
        b = by_xcc[x]
        if len(b) == 0:
            continue
        st = start[b]                         # b ascending = launch order on this XCC
        # inversions: a later-launched workgroup starting more than 5 us before an earlier-launched one
        run_max = np.maximum.accumulate(st)
        late += int(np.sum(st < run_max - 5.0))
    gaps = []
    for s in np.unique(slot):
        b = np.flatnonzero(slot == s)
        b = b[np.argsort(start[b])]
        gaps += list(start[b[1:]] - end[b[:-1]])
    gaps = np.array(gaps) if gaps else np.zeros(1)
    model = packed.dispatch_makespan(np.asarray(cost_us, dtype=np.float64), slots=32 if waves == 1 else 8)
    print(f"{tag}: {n} workgroups on {slots} {'SIMD' if waves == 1 else 'compute-unit'} slots; workgroup b on XCC b mod 8: {rr}; per SIMD {per_slot.min()} - {per_slot.max()} workgroups; "
          f"makespan {end.max() / 1e3:.2f} ms, dispatcher model (static shader-engine round robin, in order per XCC) {model / 1e3:.2f} ms ({end.max() / model:.3f} x); hand-over gap median "
          f"{np.median(gaps):.1f} us, 99th percentile {np.percentile(gaps, 99):.1f} us; started out of launch order (by > 5 us): {late}", flush=True)
    return end.max()


def main():
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [131072, 160000, 200000]
    if "--waves=4" in sys.argv:
        # workgroups of four waves that each hold a SIMD alone: one workgroup per compute unit, 8 per shader engine
        for n in sizes:
            segs, offs, vids, seeds, stride = W.speech_like_batch(n, np.random.default_rng(7))
            secs = np.add.reduceat(segs["length"].astype(np.float64), offs[:-1])
            L = np.sort(secs * 48000.0)[::-1]
            cost = np.array([L[j * 64] for j in range((n + 63) // 64)])
            us = cost / 10.0
            a = run(us, f"{n} rows as workgroups of 4 waves, longest first", waves=4)
            order, plan, greedy_ms, ideal = packed.planned_block_order(cost, 256, 32)
            b = run(us[order], f"{n} rows as workgroups of 4 waves, packed in 32 pools", waves=4)
            print(f"    packed / longest first = {b / a:.3f}   (dispatcher model, 8 slots per engine: "
                  f"{packed.dispatch_makespan(us[order], slots=8) / packed.dispatch_makespan(us, slots=8):.3f})", flush=True)
        return
    for n in sizes:
        segs, offs, vids, seeds, stride = W.speech_like_batch(n, np.random.default_rng(7))
        secs = np.add.reduceat(segs["length"].astype(np.float64), offs[:-1])
        L = np.sort(secs * 48000.0)[::-1]
        n_jobs = (n + 63) // 64
        cost = np.array([L[j * 64] for j in range(n_jobs)])
        us = cost / 10.0                                   # 192 000 samples -> 19.2 ms
        a = run(us, f"{n} rows, longest first")
        for pools in (8, 32):
            order, plan, greedy_ms, ideal = packed.planned_block_order(cost, 1024, pools)
            b = run(us[order], f"{n} rows, packed in {pools} pools")
            print(f"    packed / longest first = {b / a:.3f}   (dispatcher model: "
                  f"{packed.dispatch_makespan(us[order]) / packed.dispatch_makespan(us):.3f})", flush=True)


if __name__ == "__main__":
    main()
