#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/exp_mixed2.txt; : > $out
export GRAIL_HIP_LIB=$GRAFT_REPO_ROOT/grail-rs_amd/lib_prof/libgrail_hip.so
for args in "--lanes=2" "--lanes=1" "--lanes=2 --aligned" "--lanes=2 --voices=8" "--lanes=4 --voices=8"; do
  echo "##### $args" >> $out
  python3 tools/fast_prof.py 65536 --exact $args 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" >> $out
done
unset GRAIL_HIP_LIB
timeout 1500 python3 -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_composite_gpu.py tests/test_stream_gpu.py tests/test_live_stream_gpu.py tests/test_sample_rates_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 >> $out
