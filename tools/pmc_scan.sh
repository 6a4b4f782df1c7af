#!/bin/bash
# usage: tools/pmc_scan.sh <tag> "<scan_probe_one args>" COUNTER...   (one rocprofv3 --pmc pass over the scan kernel)
tag=$1; shift; args="$1"; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$root" || exit 1
rm -rf gpurun_out/prof/$tag; mkdir -p gpurun_out/prof/$tag
rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof/$tag -- python3 tools/scan_probe_one.py $args > gpurun_out/prof/$tag.log 2>&1
python3 - <<PY
import csv,collections,glob
f=glob.glob('gpurun_out/prof/$tag/*/*counter_collection.csv')
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(float); n=collections.Counter()
for r in rows:
    if 'scan_kernel' in r['Kernel_Name']:
        agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Dispatch_Id']]+=1
nd=len(n)
print('$tag dispatches',nd, {k:'%.4g'%(v/nd) for k,v in sorted(agg.items())})
PY
