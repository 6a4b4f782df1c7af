#!/usr/bin/env python3
"""Quick parity + timing check of the small-batch pipeline kernels against the oracle (development aid;
the permanent tests are tests/test_parity_gpu.py::test_small_batch_pipeline*)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W

ctx = G.Context(0)
voices = W.single_voice()
ctx.set_voices(voices)
ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
for n, length in ((40, 0.02), (16, 0.5), (100, 0.11)):
    segs, offs, vids, seeds = W.make_batch(n, length=length, blend_length=2.0 ** -6)
    stride = W.max_samples(length=length)
    out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    print("n", n, "pipelined", ctx.get_option("last_launch_pipelined"), "formants", ctx.get_option("last_launch_formants"), flush=True)
    ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, stride)
    ok = np.array_equal(out_len, ref_len)
    bad = [u for u in range(n) if not np.array_equal(out[u, :ref_len[u]].view(np.uint32), ref[u, :ref_len[u]].view(np.uint32))]
    print("  lengths equal", ok, "mismatching utterances", bad[:10], flush=True)
    if bad:
        u = bad[0]; a = out[u, :ref_len[u]].view(np.uint32); b = ref[u, :ref_len[u]].view(np.uint32)
        i = int(np.argmax(a != b)); print("  first diff utt", u, "sample", i, out[u, i], ref[u, i], "n diff", int((a != b).sum()))
