#!/usr/bin/env python3
"""Diagnostic: the failing case of test_random_batches_streamed_in_random_chunks in fast arithmetic — where a streamed
utterance leaves the tolerance (sample index, chunk boundaries)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W
from test_fuzz_gpu import random_batch
from test_stream_gpu import stream_all

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 41
lanes_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1]
rng = np.random.default_rng(seed)
voices = W.preset_voices(8) if seed % 2 else [G.voice_generic(48000.0), G.voice_generic(44100.0)]
ctx = G.Context(0)
ctx.set_voices(voices)
segs, offs, vids, seeds = random_batch(rng, 60, len(voices), 48000.0)
ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, 10048)
ctx.set_option("arithmetic", 1)
for chunks in ([10048],):
    for lanes in lanes_list:
        ctx.set_option("lanes_per_utterance", lanes)
        b = ctx.upload(segs, offs, vids, seeds)
        got = stream_all(ctx, b, len(ref_len), chunks, stride=10048)
        b.free()
        bad = 0
        for u in range(len(ref_len)):
            if len(got[u]) != ref_len[u]:
                print("length", u, len(got[u]), ref_len[u]); continue
            if ref_len[u]:
                want = ref[u, :ref_len[u]]
                d = np.abs(got[u].astype(np.float64) - want)
                if d.max() > G.FAST_TOLERANCE:
                    first = int(np.argmax(d > G.FAST_TOLERANCE))
                    if bad < 6:
                        print(f"lanes {lanes} chunks {chunks[:6]}: utt {u} len {ref_len[u]} max {d.max() * 2**23:.1f} first bad sample {first}  segs {offs[u]}..{offs[u+1]}: "
                              f"{[(int(s['phoneme']), round(float(s['length']) * 48000), round(float(s['blend_length']) * 48000)) for s in segs[offs[u]:offs[u+1]]][:6]}  nan {np.isnan(got[u]).sum()}")
                    bad += 1
        print(f"lanes {lanes} chunks {chunks[:6]}..: {bad} bad of {len(ref_len)}   kernel {ctx.last_kernel_name()}")
# the same batch in one piece (one-shot kernels)
for lanes in lanes_list:
    ctx.set_option("lanes_per_utterance", lanes)
    out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=10048)
    bad = 0
    for u in range(len(ref_len)):
        if ref_len[u]:
            d = np.abs(out[u, :ref_len[u]].astype(np.float64) - ref[u, :ref_len[u]])
            if d.max() > G.FAST_TOLERANCE:
                bad += 1
                if bad < 4:
                    print(f"one-shot lanes {lanes}: utt {u} max {d.max() * 2**23:.1f} first bad {int(np.argmax(d > G.FAST_TOLERANCE))}")
    print(f"one-shot lanes {lanes}: {bad} bad   kernel {ctx.last_kernel_name()}")
