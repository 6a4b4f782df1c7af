#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/exp_plan2.txt; : > $out
timeout 1700 python3 -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|^ERROR|^FAILED|options left|Error" | tail -8 >> $out
for args in "" "--scale=0.1" "--scale=0.4"; do
  echo "# speech-like 65536 $args" >> $out
  python3 tools/speech_like_bench.py 65536 $args 2>&1 | grep "exact\|fast" | cut -c1-240 >> $out
done
for n in 16384 32768 131072 200000; do
  echo "# speech-like $n" >> $out
  python3 tools/speech_like_bench.py $n 2>&1 | grep "exact" | cut -c1-240 >> $out
done
python3 -m pytest tests/test_planner_guard_gpu.py -m gpu -q -s 2>&1 | grep -v "^make\|^g++" | cut -c1-400 >> $out
