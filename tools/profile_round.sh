#!/bin/bash
# The round's evidence, collected on the GPU box: bench lines, rocprofv3 kernel stats of the default
# bench command, PMC counters for every reported workload (tools/collect_counters.py), tool outputs.
# usage (through gpurun): tools/profile_round.sh <tag>     -> gpurun_out/<tag>/
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$root" || exit 1
out=gpurun_out/$tag; mkdir -p $out
python3 tools/collect_counters.py $out/traffic.json > $out/collect_counters.log 2>&1
cp $out/traffic.json profiles/traffic.json            # so that the bench lines below carry the counters
python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
python3 bench.py --config 4 --cpu-utts 0 > $out/bench_n1_config4.json 2>/dev/null
python3 bench.py --config 2 --cpu-utts 0 > $out/bench_n1_config2.json 2>/dev/null
python3 bench.py --mode fast --cpu-utts 0 > $out/bench_n1_fast.json 2>/dev/null
python3 bench.py --pcm16 --cpu-utts 0 > $out/bench_n1_pcm16.json 2>/dev/null
python3 bench.py --verify --cpu-utts 0 --fast-leg 0 > $out/bench_n1_verify.json 2>/dev/null
rm -rf gpurun_out/prof/stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python3 bench.py --cpu-utts 0 > $out/stats.log 2>&1
cp gpurun_out/prof/stats/*/*kernel_stats.csv $out/kernel_stats.csv
rm -rf gpurun_out/prof/stats2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats2 -- python3 bench.py --config 2 --cpu-utts 0 > $out/stats2.log 2>&1
cp gpurun_out/prof/stats2/*/*kernel_stats.csv $out/kernel_stats_config2.csv
# config 4 (all eight formants live), exact (with the fast leg of the default line) and the second tolerance tier
rm -rf gpurun_out/prof/stats4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats4 -- python3 bench.py --config 4 --cpu-utts 0 > $out/stats4.log 2>&1
cp gpurun_out/prof/stats4/*/*kernel_stats.csv $out/kernel_stats_config4.csv
rm -rf gpurun_out/prof/stats5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats5 -- python3 bench.py --mode mid --cpu-utts 0 --fast-leg 0 > $out/stats5.log 2>&1
cp gpurun_out/prof/stats5/*/*kernel_stats.csv $out/kernel_stats_mid.csv
python3 bench.py --mode mid --cpu-utts 0 > $out/bench_n1_mid.json 2>/dev/null
python3 tools/tail_bench.py > $out/tail.txt 2>&1
python3 tools/tail_bench.py --presets 65536 65537 70000 98304 131073 > $out/tail_presets.txt 2>&1
python3 tools/mid_bench.py > $out/mid_bench.txt 2>&1
python3 tools/small_batch_bench.py > $out/small_batch.txt 2>&1
python3 tools/stream_latency.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $out/stream_latency.txt
python3 tools/host_output_bench.py > $out/host_output.txt 2>&1
python3 tools/live_stream_bench.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $out/live_stream.txt
python3 tools/stream_latency.py 1 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $out/stream_latency_fast.txt
# the speech-like corpus (event-dense input): the library's own plan, the one-round layout, pinned mappings, phonemes of 4 - 16 ms
( echo "# ---- 65 536 utterances, the library's plan"; python3 tools/speech_like_bench.py 65536
  echo "# ---- one round, one lane per utterance (option ragged_plan = 0)"; python3 tools/speech_like_bench.py 65536 --no-ragged-plan
  echo "# ---- pinned: two lanes per utterance"; python3 tools/speech_like_bench.py 65536 --lanes=2
  echo "# ---- phonemes of 4 - 16 ms (--scale=0.1), the library's plan"; python3 tools/speech_like_bench.py 65536 --scale=0.1
  echo "# ---- phonemes of 4 - 16 ms, pinned: two lanes per utterance"; python3 tools/speech_like_bench.py 65536 --scale=0.1 --lanes=2
  echo "# ---- blend_length = length (no kinks)"; python3 tools/speech_like_bench.py 65536 --blend-is-length ) > $out/speech_like.txt 2>&1
rm -rf gpurun_out/prof/stats6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats6 -- python3 tools/speech_like_bench.py 65536 > $out/stats6.log 2>&1
cp gpurun_out/prof/stats6/*/*kernel_stats.csv $out/kernel_stats_speech_like.csv
bash tools/pmc_speech_like.sh > $out/pmc_speech_like_raw.txt 2>&1
python3 tools/two_waves_bench.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $out/two_waves.txt
python3 -m pytest tests/test_planner_guard_gpu.py -m gpu -q -s 2>&1 | grep -v "^make\|^g++" > $out/planner_guard.txt
# round 6: the workgroup dispatcher, the packed launch order (option off / on, every row's digest), the planner's prices next to
# the pinned mappings, the node call on one GPU
python3 tools/dispatch_order.py 131072 160000 200000 > $out/dispatch_order_raw.txt 2>&1
python3 tools/packed_order_ab.py --check 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $out/packed_order_ab.txt
( GRAIL_PLAN_DEBUG=1 python3 tools/plan_debug.py 40000 65536 80000 100000 131072 160000 200000 2>&1; GRAIL_PLAN_DEBUG=1 python3 tools/plan_debug.py 40000 65536 80000 100000 131072 160000 200000 --fast 2>&1 ) | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $out/plan_debug_raw.txt
python3 tools/node_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $out/node_bench.txt
python3 -m pytest tests/test_node_gpu.py tests/test_packed_order_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" > $out/node_tests.txt
python3 tools/ragged_bench.py 65536 1 0 > $out/ragged.txt 2>&1
python3 tools/ragged_bench.py 65536 1 1 > $out/ragged_fast.txt 2>&1
python3 tools/ragged_bench.py 131072 1 0 > $out/ragged_131072.txt 2>&1
# SQ counters of the final exact headline and config-4 kernels and of the fast headline kernel
( for cfg in "--config 3" "--config 4" "--config 3 --mode fast"; do
    bash tools/pmc.sh sq1 "$cfg --steps 1 --warmup 0 --fast-leg 0 --ramp 0" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
    bash tools/pmc.sh sq2 "$cfg --steps 1 --warmup 0 --fast-leg 0 --ramp 0" SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
  done ) > $out/pmc_sq.txt 2>&1
# every utterance of full-size batches against the oracle (exact mode), one size per kernel family — the 2 / 4 / 8-lane
# kernels were rebuilt this round (one wave per SIMD by construction), 70 000 is a composite launch
# ... and 131 072 speech-like rows (packed launch order), every utterance
GRAIL_SOAK=1 GRAIL_SOAK_UTTS=131072 GRAIL_SOAK_SPEECH=1.0 python3 -m pytest tests/test_full_parity_soak_gpu.py -m gpu -q -s 2>&1 | grep -E "full parity|passed|failed|skipped" > $out/full_parity_speech_like_131072.txt
( for n in 70000 32768 16384 8192 4096; do GRAIL_SOAK=1 GRAIL_SOAK_UTTS=$n python3 -m pytest tests/test_full_parity_soak_gpu.py -m gpu -q -s 2>&1 | grep -E "full parity|passed|failed"; done ) > $out/full_parity.txt 2>&1
ls -la $out
