#!/usr/bin/env python3
"""Caller-built SequenceElems (grail_synthesize_batch_elems): the bench corpus handed over as elems — every
segment carrying the phoneme's elem itself, which is what a caller with a voice of its own does — exact and fast (with and without the
time-split kernels), next to the same batch as PhonemeElems.   usage: elems_split_bench.py [n ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import ctypes as C
import numpy as np
import grail_hip as G
from grail_hip import workload as W

ctx = G.Context(0)
for n_voices in (1, 8):
    voices = W.single_voice() if n_voices == 1 else W.preset_voices(8)
    ctx.set_voices(voices)
    for n in ([int(a) for a in sys.argv[1:]] or (1024, 4096, 16384)):
        segs, offs, vids, seeds = W.make_batch(n, n_voices=n_voices)
        stride = W.max_samples()
        arr = (G.SequenceElem * len(segs))()
        for u in range(n):
            v = voices[int(vids[u])]
            for i in range(int(offs[u]), int(offs[u + 1])):
                ph = int(segs["phoneme"][i])
                arr[i].has_elem = 1 if ph >= G.PH_A else 0
                if ph >= G.PH_A:
                    arr[i].elem = v.phonemes[ph - G.PH_A]
                arr[i].elem.frequency = min(float(segs["frequency"][i]), 0.5)
                arr[i].length = float(segs["length"][i])
                arr[i].blend_length = float(segs["blend_length"][i])
        h = C.c_void_p()
        G._check(G.load().grail_batch_upload_elems(ctx.handle, C.cast(arr, C.c_void_p), offs.ctypes.data, vids.ctypes.data,
                                                   seeds.ctypes.data, n, C.byref(h)))
        ebatch = G.Batch(ctx, h, n)
        pbatch = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * 4)
        d_len = ctx.device_alloc(n * 4)
        row = [f"voices={n_voices} n={n:6d}:"]
        for what, batch, split, fast in (("exact: PhonemeElems", pbatch, 1, 0), ("SequenceElems", ebatch, 1, 0),
                                         ("fast: PhonemeElems", pbatch, 1, 1), ("SequenceElems", ebatch, 1, 1),
                                         ("SequenceElems, time_split = 0", ebatch, 0, 1)):
            ctx.set_option("arithmetic", fast)
            ctx.set_option("time_split", split)
            ms = []
            for _ in range(4):
                batch.synthesize_async(d_out, stride, d_len)
                ctx.sync()
                ms.append(ctx.last_kernel_ms())
            chunks = ctx.get_option("last_launch_chunks")
            row.append(f"{what} {min(ms):6.2f} ms ({ctx.last_kernel_name().replace('synth_kernel', 'k')}{' x%d' % chunks if chunks else ''})")
        ctx.set_option("time_split", 1)
        ctx.set_option("arithmetic", 0)
        print("  ".join(row), flush=True)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        ebatch.free()
        pbatch.free()
