#!/usr/bin/env python3
"""Half-minute utterances (four segments of 2 - 9 s, blends of 0.25 - 8 s) through every kernel family against the
oracle: exact arithmetic bit for bit (pipelined workgroups, lane kernels), fast arithmetic within the tolerance (scan
kernel, lane kernel, time-split kernels with 16 and 64 chunks); generic voice and 8 presets."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W
ULP = 2.0 ** -23
ctx = G.Context(0)
for nv in (1, 8):
    voices = W.single_voice() if nv == 1 else W.preset_voices(8)
    ctx.set_voices(voices)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    rng = np.random.default_rng(3 + nv)
    n_utt = 12
    segs, offs, vids, seeds = W.make_batch(n_utt, n_voices=nv)
    k = len(segs)
    segs["length"] = rng.uniform(2.0, 9.0, k).astype(np.float32)
    segs["blend_length"] = rng.choice([0.5, 2.0, 0.25, 4.0, 8.0], k).astype(np.float32)
    stride = int(4 * 9.0 * 48000) + 64
    ref, ref_len = O.synthesize_batch_threads(ov, segs, offs, vids, seeds, stride, 8)[:2]
    print("voices", nv, "lengths", int(ref_len.min()), "..", int(ref_len.max()), flush=True)
    for fast, opts in ((0, {}), (0, {"lanes_per_utterance": 1}), (0, {"lanes_per_utterance": 8}), (1, {}), (1, {"time_split": 0, "time_parallel_scan": 0, "lanes_per_utterance": 1}),
                       (1, {"time_split_chunks": 16}), (1, {"time_split_chunks": 64})):
        ctx.set_option("arithmetic", fast)
        for a, b in opts.items(): ctx.set_option(a, b)
        out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        name = ctx.last_kernel_name()
        for a in opts: ctx.set_option(a, 1 if a in ("time_split", "time_parallel_scan") else 0)
        assert np.array_equal(out_len, ref_len), (name, out_len, ref_len)
        worst = 0.0
        for u in range(n_utt):
            n = int(ref_len[u])
            if fast == 0:
                assert np.array_equal(out[u, :n].view(np.uint32), ref[u, :n].view(np.uint32)), (name, u)
            else:
                worst = max(worst, float(np.abs(out[u, :n].astype(np.float64) - ref[u, :n]).max()) / max(1.0, float(np.abs(ref[u, :n]).max())))
        print("  ", "fast" if fast else "exact", opts, name, "bit-identical" if not fast else f"worst {worst / ULP:.1f} * 2^-23", flush=True)
        assert worst <= G.FAST_TOLERANCE
ctx.set_option("arithmetic", 0)
