#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/exp_pipe2.txt; : > $out
timeout 1500 python3 -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_composite_gpu.py tests/test_stream_gpu.py tests/test_live_stream_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 >> $out
for lib in lib lib_nopipe; do
  export GRAIL_HIP_LIB=$GRAFT_REPO_ROOT/grail-rs_amd/$lib/libgrail_hip.so
  echo "##### $lib" >> $out
  python3 bench.py --config 2 --cpu-utts 0 --other-configs 0 --fast-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2', d['ms_per_step'], d['roofline']['kernel'])" >> $out
  python3 bench.py --config 2 --cpu-utts 0 --other-configs 0 --fast-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2', d['ms_per_step'], d['roofline']['kernel'])" >> $out
  for n in 256 4096; do for sc in 1.0 0.4; do
    python3 tools/speech_like_bench.py $n --scale=$sc 2>&1 | grep exact | cut -c1-200 >> $out
  done; done
  python3 tools/stream_latency.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | cut -c1-200 >> $out
  python3 tools/live_stream_bench.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | cut -c1-200 >> $out
done
