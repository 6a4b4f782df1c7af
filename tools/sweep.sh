#!/bin/bash
# usage: tools/sweep.sh "<bench args>" variant-or-lanes specs...   e.g.  tools/sweep.sh "--steps 2" L2 L4 V1 V3
args="$1"; shift
for spec in "$@"; do
  case $spec in
    L*) extra="--lanes ${spec#L}";;
    V*) extra="--variant ${spec#V}";;
  esac
  timeout 300 python bench.py $args --cpu-utts 0 $extra 2>&1 | python -c "
import json,sys
l=sys.stdin.read().strip().splitlines()[-1]
try:
  d=json.loads(l); print('$spec', '%.4g samples/s' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'])
except Exception as e: print('$spec fail', l[-300:])
"
done
