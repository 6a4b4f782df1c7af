cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp4
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/exp4/test_gpu.txt 2>&1
tail -5 gpurun_out/exp4/test_gpu.txt
timeout 600 python3 tools/speech_like_bench.py 65536 --lanes=1 > gpurun_out/exp4/speech_l1.txt 2>&1
timeout 600 python3 tools/speech_like_bench.py 65536 --lanes=2 > gpurun_out/exp4/speech_l2.txt 2>&1
timeout 600 python3 tools/speech_like_bench.py 65536 --scale=0.1 --lanes=1 > gpurun_out/exp4/speech_s01_l1.txt 2>&1
timeout 600 python3 tools/speech_like_bench.py 65536 --scale=0.1 --lanes=2 > gpurun_out/exp4/speech_s01_l2.txt 2>&1
for args in "65536 --lanes=1" "65536 --lanes=2" "65536 --lanes=1 --scale=0.1" "65536 --lanes=2 --voices=8" "65536 --lanes=1 --aligned" "65536 --lanes=1 --aligned --voices=8"; do
    GRAIL_HIP_LIB=$PWD/grail-rs_amd/lib_prof/libgrail_hip.so timeout 300 python3 tools/fast_prof.py $args > "gpurun_out/exp4/prof_$(echo $args | tr ' =' '__').txt" 2>&1
done
timeout 300 python3 tools/ab_libs.py $PWD/grail-rs_amd/lib/libgrail_hip.so fast > gpurun_out/exp4/ab_fast.txt 2>&1
timeout 300 python3 tools/ab_libs.py $PWD/grail-rs_amd/lib/libgrail_hip.so exact > gpurun_out/exp4/ab_exact.txt 2>&1
grep -h "fast \|exact" gpurun_out/exp4/speech*.txt
head -7 gpurun_out/exp4/ab_fast.txt; head -7 gpurun_out/exp4/ab_exact.txt
