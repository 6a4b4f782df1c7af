#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/exp_mixed3.txt; : > $out
timeout 1500 python3 -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_composite_gpu.py tests/test_stream_gpu.py tests/test_live_stream_gpu.py tests/test_sample_rates_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 >> $out
for lib in lib lib_nomix; do
  export GRAIL_HIP_LIB=$GRAFT_REPO_ROOT/grail-rs_amd/$lib/libgrail_hip.so
  echo "##### $lib" >> $out
  python3 bench.py --cpu-utts 0 --other-configs 0 --fast-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['roofline']['kernel'])" >> $out
  python3 bench.py --config 4 --cpu-utts 0 --other-configs 0 --fast-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['ms_per_step'], d['roofline']['kernel'])" >> $out
  for args in "" "--lanes=4" "--lanes=8" "--scale=0.1"; do
    echo "# speech-like $args" >> $out
    python3 tools/speech_like_bench.py 65536 $args 2>&1 | grep exact >> $out
  done
done
for lib in lib lib_nomix; do export GRAIL_HIP_LIB=$GRAFT_REPO_ROOT/grail-rs_amd/$lib/libgrail_hip.so; echo "## mid_bench $lib" >> $out; python3 tools/mid_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" >> $out; done
export GRAIL_HIP_LIB=$GRAFT_REPO_ROOT/grail-rs_amd/lib_prof/libgrail_hip.so
for args in "--lanes=2" "--lanes=4"; do
  echo "##### $args" >> $out
  python3 tools/fast_prof.py 65536 --exact $args 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" >> $out
done
