#!/bin/bash
# The ragged plan outside the cells its model was fitted on: phonemes x 0.25 and x 2 (utterances of 0.1 - 1 s and 1 - 7.7 s), three batch
# sizes; the library's choice next to every pinned lane mapping.   usage: bash tools/ragged_plan_sweep.sh > gpurun_out/ragged_plan_sweep.txt
cd "$(dirname "$0")/.."
for sc in 0.25 2.0; do for n in 16384 65536 100000; do
  echo "# ---- phonemes x $sc, $n utterances: the library's choice"
  timeout 300 python tools/speech_like_bench.py $n --scale=$sc
  for L in 1 2 4; do
    echo "# ---- phonemes x $sc, $n utterances: \"lanes_per_utterance\" = $L"
    timeout 300 python tools/speech_like_bench.py $n --scale=$sc --lanes=$L
  done
done; done
