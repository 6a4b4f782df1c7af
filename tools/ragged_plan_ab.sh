#!/bin/bash
# A/B of option "ragged_plan" (launch_plan.cpp, "Ragged batches") on the speech-like corpus (tools/speech_like_bench.py; also with
# every length times 0.4 / 0.1, and on small batches) and on the ragged bench corpus (tools/ragged_bench.py): every line once with the
# one-round launch policy, once with the plan weighed by the rows' lengths and events.
# usage: bash tools/ragged_plan_ab.sh > gpurun_out/ragged_plan.txt
cd "$(dirname "$0")/.."
ab() {   # scale, sizes...
  sc=$1; shift
  for n in "$@"; do
    echo "# ---- phonemes x $sc, $n utterances: \"ragged_plan\" = 0"
    timeout 300 python tools/speech_like_bench.py $n --scale=$sc --no-ragged-plan
    echo "# ---- phonemes x $sc, $n utterances: \"ragged_plan\" = 1 (default)"
    timeout 300 python tools/speech_like_bench.py $n --scale=$sc
  done
}
ab 1.0 256 4096 8192 16384 32768 49152 65536 100000 131072
ab 0.4 256 4096 8192 65536 200000
ab 0.1 256 4096 8192 65536 200000
for ar in 0 1; do
  echo "# ---- ragged bench corpus (segments of 0.3 - 0.7 s), 65 536 utterances: \"ragged_plan\" = 0"
  timeout 300 python tools/ragged_bench.py 65536 1 $ar --no-ragged-plan | grep -v "^n ="
  echo "# ---- ragged bench corpus, 65 536 utterances: \"ragged_plan\" = 1 (default)"
  timeout 300 python tools/ragged_bench.py 65536 1 $ar | grep -v "^n ="
done
