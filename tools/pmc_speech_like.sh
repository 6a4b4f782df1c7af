#!/bin/bash
# SQ counters of the kernels the speech-like corpus runs on (the library's plan, 65 536 utterances, one and eight voices, exact
# and tolerance arithmetic): per-kernel averages per dispatch.   usage (on the GPU box): bash tools/pmc_speech_like.sh > out.txt
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$root" || exit 1
rm -rf gpurun_out/prof/speech_pmc; mkdir -p gpurun_out/prof/speech_pmc
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS \
  --output-format csv -d gpurun_out/prof/speech_pmc -- python3 tools/speech_like_bench.py 65536 > gpurun_out/prof/speech_pmc.log 2>&1
grep "speech-like" gpurun_out/prof/speech_pmc.log | cut -c1-220
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/prof/speech_pmc/*/*counter_collection.csv')
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name']
    if 'synth_kernel' not in k:
        continue
    k = k[k.index('synth_kernel'):].split('(')[0]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
print("# per dispatch (launch); <L, T, WAVES, MINW, STREAM, HALF, ANYBL, NFA, PIPE, FAST, ...>")
for k, c in agg.items():
    d = len(n[k]); v = {a: b / d for a, b in c.items()}
    waves = max(v.get('SQ_WAVES', 1.0), 1.0)
    print(f"{k}: {d} launches, waves {waves:.0f}, VALU instructions {v.get('SQ_INSTS_VALU', 0):.4g} ({v.get('SQ_INSTS_VALU', 0) / waves:.4g} per wave), "
          f"SALU {v.get('SQ_INSTS_SALU', 0):.4g}, LDS {v.get('SQ_INSTS_LDS', 0):.4g}, wave cycles {v.get('SQ_WAVE_CYCLES', 0):.4g} "
          f"({v.get('SQ_WAVE_CYCLES', 0) / max(v.get('SQ_INSTS_VALU', 1), 1):.2f} per VALU instruction), VALU active {v.get('SQ_ACTIVE_INST_VALU', 0):.4g}, "
          f"waiting {v.get('SQ_WAIT_ANY', 0):.4g}, busy cycles {v.get('SQ_BUSY_CYCLES', 0):.4g}")
PY
