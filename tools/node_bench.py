#!/usr/bin/env python3
"""The node call on ONE GPU (all this box has): grail_node_synthesize_batch over 1 / 2 / 4 / 8 device slots on device 0 —
contexts and host threads as on a node, the PCIe link and the GPU shared — next to one grail_synthesize_batch, into a pinned
destination.  What it shows: the node layer costs nothing (the shards' kernels and copies overlap; the end-to-end rate is
the one-context rate, bounded by the one link), and every row is the one-context row.  Scaling over GPUs it cannot show.
--device: the rows stay in HBM (grail_node_synthesize_batch_device) — the headline metric's shape, through the node API, in one
process: on a box with several GPUs `--devices=0,1,...` renders n_utt rows PER DEVICE and reports whole-node samples/s.
usage: node_bench.py [n_utt] [--device] [--devices=0,1,...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W

os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384
devices = None
for a in sys.argv[1:]:
    if a.startswith("--devices="):
        devices = [int(x) for x in a[10:].split(",")]
if "--device" in sys.argv:
    devices = devices or [0]
    per = n
    n_total = per * len(devices)
    voices = W.single_voice()
    segs, offs, vids, seeds = W.make_batch(n_total)
    stride = W.max_samples()
    distinct = len(set(devices)) == len(devices)
    with G.Node(devices, voices_without_rccl=not distinct) as node:
        node.set_voices(voices)
        shards = [G.node_shard_of(offs, i, len(devices))[0] for i in range(len(devices))]
        ctxs = [node.context(i) for i in range(len(devices))]
        bufs = [c.device_alloc(int(s.rows) * stride * 4) for c, s in zip(ctxs, shards)]
        for c, s, b in zip(ctxs, shards, bufs):
            c.memset(b, 0, int(s.rows) * stride * 4)
        best, lens = None, None
        for _ in range(4):
            t0 = time.perf_counter()
            lens = node.synthesize_device(segs, offs, vids, seeds, bufs, stride)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        total = int(lens.astype(np.uint64).sum())
        kern = [c.last_kernel_ms() for c in ctxs]
        print(f"node of {len(devices)} device slot(s) {devices} (RCCL ranks {node.get_option('node_rccl_ranks')}), {per} utterances x 2 s per slot, "
              f"rows left in HBM: one call {best * 1e3:.1f} ms (upload + kernels) = {total / best:.3e} samples/s whole node; "
              f"kernel ms per slot {', '.join('%.2f' % k for k in kern)}; shards {', '.join('%.0f' % m for m in node.last_shard_ms())} ms")
        for c, b in zip(ctxs, bufs):
            c.device_free(b)
    sys.exit(0)
voices = W.single_voice()
segs, offs, vids, seeds = W.make_batch(n)
stride = W.max_samples()
with G.Context(0) as ctx:
    ctx.set_voices(voices)
    dst = ctx.host_alloc((n, stride), np.float32)
    lens = np.zeros(n, dtype=np.uint32)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.synthesize_into(dst, lens, segs, offs, vids, seeds)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    ref = dst.copy()
    total = int(lens.astype(np.uint64).sum())
    print(f"{n} utterances x 2 s = {total} samples, {dst.nbytes / 1e9:.2f} GB of f32 rows into pinned host memory")
    print(f"one context, grail_synthesize_batch:            {best * 1e3:7.1f} ms = {dst.nbytes / best / 1e9:5.1f} GB/s = {total / best:.3e} samples/s")
    ctx.host_free(dst)
for slots in (1, 2, 4, 8):
    with G.Node([0] * slots, voices_without_rccl=slots > 1) as node:
        t0 = time.perf_counter()
        node.set_voices(voices)
        t_voices = time.perf_counter() - t0
        dst = node.host_alloc((n, stride), np.float32)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            out, out_len = node.synthesize(segs, offs, vids, seeds, out=dst)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        same = np.array_equal(out.view(np.uint32), ref.view(np.uint32)) and np.array_equal(out_len, lens)
        ms = node.last_shard_ms()
        print(f"node of {slots} slot(s) on device 0 ({'RCCL, ranks ' + str(node.get_option('node_rccl_ranks')) if slots == 1 else 'table per context'}; "
              f"set_voices {t_voices * 1e3:.0f} ms): {best * 1e3:7.1f} ms = {dst.nbytes / best / 1e9:5.1f} GB/s; shards {min(ms):.0f} - {max(ms):.0f} ms; "
              f"rows {'equal' if same else 'DIFFER'}", flush=True)
        node.host_free(dst)
