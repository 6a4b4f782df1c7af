#!/bin/bash
# Pinned-mapping measurements behind the ragged planner's model of the tolerance-mode kernels (launch_plan.cpp
# ragged_wave_ms): the speech-like corpus with phonemes of 40 - 160, 16 - 64 and 4 - 16 ms at 65 536 utterances, one / two /
# four / eight lanes per utterance, one and eight voices; and the aligned rate of every mapping (one round of the machine).
# usage: bash tools/ragged_fit_collect.sh > gpurun_out/ragged_fit.txt ; then tools/ragged_fit.py gpurun_out/ragged_fit.txt
cd "$(dirname "$0")/.."
for sc in 1.0 0.4 0.1; do
  for L in 1 2 4 8; do
    echo "# scale $sc lanes $L"
    timeout 300 python3 tools/speech_like_bench.py 65536 --scale=$sc --lanes=$L --no-split
  done
done
python3 - <<'PY'
import os, sys
sys.path.insert(0, "grail-rs_amd")
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
ctx.set_option("time_split", 0); ctx.set_option("time_parallel_scan", 0); ctx.set_option("small_batch_pipeline", 0)
stride = W.max_samples()
for nv in (1, 8):
    ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
    for L in (1, 2, 4, 8):
        n = 65536 // L
        segs, offs, vids, seeds = W.make_batch(n, n_voices=nv)
        b = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
        ctx.set_option("lanes_per_utterance", L)
        for ar in (0, 1):
            ctx.set_option("arithmetic", ar)
            ms = []
            for _ in range(4):
                b.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(ctx.last_kernel_ms())
            print(f"aligned voices={nv} L={L} n={n} {'fast ' if ar else 'exact'} {min(ms):7.2f} ms  {ctx.last_kernel_name()}", flush=True)
        ctx.device_free(d_out); ctx.device_free(d_len); b.free()
PY
