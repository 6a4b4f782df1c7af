cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r
python bench.py > gpurun_out/r/bench_n1.json 2> gpurun_out/r/bench_n1.err
python bench.py --voices 8 --cpu-utts 0 > gpurun_out/r/bench_n1_config4.json 2>/dev/null
python bench.py --literal --cpu-utts 0 > gpurun_out/r/bench_n1_literal.json 2>/dev/null
python bench.py --pcm16 --cpu-utts 0 > gpurun_out/r/bench_n1_pcm16.json 2>/dev/null
rm -rf gpurun_out/prof/stats; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python3 bench.py --cpu-utts 0 > gpurun_out/r/stats.log 2>&1
cp gpurun_out/prof/stats/*/*kernel_stats.csv gpurun_out/r/kernel_stats.csv
tools/pmc.sh w3 "--steps 2 --warmup 0" WRITE_SIZE > gpurun_out/r/pmc.txt
tools/pmc.sh f3 "--steps 2 --warmup 0" FETCH_SIZE >> gpurun_out/r/pmc.txt
tools/pmc.sh s3 "--steps 1 --warmup 0" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE >> gpurun_out/r/pmc.txt
tools/pmc.sh s4 "--steps 1 --warmup 0 --voices 8" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE >> gpurun_out/r/pmc.txt
tools/pmc.sh s3b "--steps 1 --warmup 0" SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM >> gpurun_out/r/pmc.txt
cat gpurun_out/r/pmc.txt
