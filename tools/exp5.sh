cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp5
timeout 1500 python -m pytest tests/test_fast_gpu.py tests/test_fuzz_gpu.py tests/test_stream_gpu.py tests/test_live_stream_gpu.py tests/test_composite_gpu.py -q -m gpu > gpurun_out/exp5/test_gpu.txt 2>&1
tail -3 gpurun_out/exp5/test_gpu.txt
for L in 1 2; do for sc in 1.0 0.1; do timeout 600 python3 tools/speech_like_bench.py 65536 --lanes=$L --scale=$sc | grep "fast "; done; done
for args in "65536 --lanes=1" "65536 --lanes=2" "65536 --lanes=1 --scale=0.1"; do
    GRAIL_HIP_LIB=$PWD/grail-rs_amd/lib_prof/libgrail_hip.so timeout 300 python3 tools/fast_prof.py $args > "gpurun_out/exp5/prof_$(echo $args | tr ' =' '__').txt" 2>&1
done
timeout 300 python3 tools/ab_libs.py $PWD/grail-rs_amd/lib/libgrail_hip.so fast 2>&1 | head -6
