#!/bin/bash
# The round's evidence, collected on the GPU box: bench lines, rocprofv3 kernel stats of the default
# bench command, PMC counters for every reported workload (tools/collect_counters.py), tool outputs.
# usage (through gpurun): tools/profile_round.sh <tag>     -> gpurun_out/<tag>/
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
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
python3 tools/small_batch_bench.py > $out/small_batch.txt 2>&1
python3 tools/stream_latency.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $out/stream_latency.txt
python3 tools/host_output_bench.py > $out/host_output.txt 2>&1
python3 tools/ragged_bench.py > $out/ragged.txt 2>&1
./tools/dpp_check.bin > $out/dpp_check.txt 2>&1
python3 tools/scan_split_crossover.py > $out/scan_split.txt 2>&1
./tools/phase_chain_mb.bin > $out/phase_chain.txt 2>&1
ls -la $out
