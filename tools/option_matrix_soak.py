#!/usr/bin/env python3
"""Every tuning option at random: 'with "arithmetic" = 0 none of them ever changes a result bit' and with
"arithmetic" = 1 the rows stay within the tolerance (include/grail_hip.h).  Random batches (the fuzz tests' generator:
ragged segment counts, Silence / Stop / Glide, blend lengths that are powers of two and not), random settings of
lanes_per_utterance, small_batch_pipeline, pipeline_round32, pipeline4/8_max_groups, skip_silent_formants,
sort_by_length, time_parallel_scan (+ its two thresholds), time_split (+ chunks, span, cost, minimum), composite_launches,
row_groups, ragged_plan, two_waves_per_simd, packed_launch_order and assume_compute_units (1 - 4 compute units: batches of a few hundred utterances are then cut, grouped
and planned by their rows' lengths the way batches of 100 000 are on the whole device), against the oracle.   usage: option_matrix_soak.py [trials [seed [big]]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import grail_hip as G
import oracle_lib as O
from grail_hip import workload as W
from test_fuzz_gpu import pow2_blend_batch, random_batch

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"       # also batches of thousands: time-split and scan by themselves
ctx = G.Context(0)
DEFAULTS = {"lanes_per_utterance": 0, "small_batch_pipeline": 1, "pipeline_round32": 1, "pipeline4_max_groups": 512,
            "pipeline8_max_groups": 512, "skip_silent_formants": 1, "sort_by_length": 1, "time_parallel_scan": 1,
            "time_parallel_scan_max_utterances": 1536, "time_parallel_scan_split_max_utterances": 1536, "time_split": 1,
            "time_split_chunks": 0, "time_split_span_samples": 0, "time_split_ff_cost_permille": 165,
            "time_split_min_utterances": 1537, "composite_launches": 1, "row_groups": 1, "ragged_plan": 1,
            "assume_compute_units": 0, "two_waves_per_simd": 1, "pipeline_spread": 1, "packed_launch_order": 1}
CHOICES = {"lanes_per_utterance": [0, 0, 1, 2, 4, 8], "small_batch_pipeline": [0, 1], "pipeline_round32": [0, 1, 1, 2],
           "pipeline4_max_groups": [0, 2, 512], "pipeline8_max_groups": [0, 3, 512], "skip_silent_formants": [0, 1],
           "sort_by_length": [0, 1], "time_parallel_scan": [0, 1], "time_parallel_scan_max_utterances": [0, 40, 1536],
           "time_parallel_scan_split_max_utterances": [0, 30, 1536], "time_split": [0, 1], "time_split_chunks": [0, 0, 2, 5],
           "time_split_span_samples": [0, 4096, 9000], "time_split_ff_cost_permille": [0, 165, 900],
           "time_split_min_utterances": [0, 50, 1537], "composite_launches": [0, 1, 1], "row_groups": [0, 1, 2],
           "ragged_plan": [0, 1, 1], "assume_compute_units": [0, 1, 2, 4], "two_waves_per_simd": [0, 1, 1], "pipeline_spread": [0, 1, 1], "packed_launch_order": [0, 1, 1]}
worst, kernels, packed_launches = 0.0, {}, 0
for trial in range(trials):
    nv = int(rng.choice([1, 2, 8]))
    voices = W.single_voice() if nv == 1 else ([G.voice_generic(48000.0), G.voice_generic(44100.0)] if nv == 2 else W.preset_voices(8))
    ctx.set_voices(voices)
    n_utt = int(rng.choice([1, 17, 64, 130, 300] + ([1700, 2600, 4200] if BIG else [])))
    segs, offs, vids, seeds = (pow2_blend_batch(rng, n_utt, len(voices)) if rng.integers(0, 2)
                               else random_batch(rng, n_utt, len(voices), 48000.0))
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    stride = 10048
    ref, ref_len = O.synthesize_batch(ov, segs, offs, vids, seeds, stride)
    for rep in range(3):
        opts = {k: int(rng.choice(v)) for k, v in CHOICES.items()}
        fast = int(rng.integers(0, 2))
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_option("arithmetic", fast)
        out, out_len = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
        name = ctx.last_kernel_name()
        kernels[name] = kernels.get(name, 0) + 1
        packed_launches += 1 if ctx.get_option("last_launch_packed") else 0
        assert np.array_equal(out_len, ref_len), (trial, opts, fast, name)
        for u in range(n_utt):
            n = int(ref_len[u])
            if not fast:
                assert np.array_equal(out[u, :n].view(np.uint32), ref[u, :n].view(np.uint32)), (trial, opts, name, u)
            elif n:
                d = float(np.abs(out[u, :n].astype(np.float64) - ref[u, :n]).max()) / max(1.0, float(np.abs(ref[u, :n]).max()))
                worst = max(worst, d)
                assert d <= G.FAST_TOLERANCE, (trial, opts, name, u, d * 2 ** 23)
    for k, v in DEFAULTS.items():
        ctx.set_option(k, v)
    ctx.set_option("arithmetic", 0)
print(f"{trials} random batches x 3 random option settings: exact rows bit-identical to the oracle, fast rows within "
      f"{worst * 2 ** 23:.1f} * 2^-23; {packed_launches} of the launches took a packed order; kernels used:")
for k, v in sorted(kernels.items()):
    print(f"  {v:4d}  {k}")
