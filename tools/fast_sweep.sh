#!/bin/bash
# kernel times (hipEvents) of both arithmetic modes over configs / lanes-per-utterance, same box
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp; cd "$root" || exit 1; mkdir -p gpurun_out/r2
LANES=${LANES:-"0 1 2 4 8"}
for cfg in ${CONFIGS:-3 4 2}; do for mode in exact fast; do for L in $LANES; do
  python bench.py --config $cfg --mode $mode --fast-leg 0 --lanes $L --cpu-utts 0 --steps 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('config',$cfg,'$mode','lanes',$L,'ms %.2f'%d['roofline']['kernel_ms'],'%.3g samples/s'%d['value'],d['roofline']['kernel'])"
done; done; done
