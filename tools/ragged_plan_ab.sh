#!/bin/bash
# A/B of option "ragged_plan" (launch_plan.cpp, "Ragged batches") on the speech-like corpus (tools/speech_like_bench.py) and
# on the ragged bench corpus (tools/ragged_bench.py): every line once with the one-round launch policy, once with the plan
# weighed by the rows' lengths and events.   usage: bash tools/ragged_plan_ab.sh > gpurun_out/ragged_plan.txt
cd "$(dirname "$0")/.."
for n in 8192 16384 32768 49152 65536 100000 131072; do
  echo "# ---- speech-like, $n utterances: \"ragged_plan\" = 0"
  timeout 300 python tools/speech_like_bench.py $n --no-ragged-plan
  echo "# ---- speech-like, $n utterances: \"ragged_plan\" = 1 (default)"
  timeout 300 python tools/speech_like_bench.py $n
done
for ar in 0 1; do
  echo "# ---- ragged bench corpus (segments of 0.3 - 0.7 s), 65 536 utterances: \"ragged_plan\" = 0"
  timeout 300 python tools/ragged_bench.py 65536 1 $ar --no-ragged-plan | grep -v "^n ="
  echo "# ---- ragged bench corpus, 65 536 utterances: \"ragged_plan\" = 1 (default)"
  timeout 300 python tools/ragged_bench.py 65536 1 $ar | grep -v "^n ="
done
