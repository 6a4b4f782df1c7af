#!/usr/bin/env python3
"""Where does the scan kernel's output leave the exact kernel's?  (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
segs, offs, vids, seeds = W.make_batch(4, n_voices=nv)
stride = W.max_samples()
ctx.set_option("arithmetic", 0)
ref, rl = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
ctx.set_option("arithmetic", 1)
out, ol = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
print(ctx.last_kernel_name(), rl, ol)
for u in range(4):
    n = int(rl[u])
    d = np.abs(out[u, :n].astype(np.float64) - ref[u, :n])
    bad = np.nonzero(d > 1e-4)[0]
    print(f"utt {u}: max {d.max():.3e}, {len(bad)} samples off by > 1e-4, first at {bad[0] if len(bad) else None}; "
          f"zeros in out where ref != 0: {int(np.count_nonzero((out[u,:n]==0)&(ref[u,:n]!=0)))}")
    if len(bad):
        i = int(bad[0])
        lo = max(0, i - 4)
        print("   idx ", list(range(lo, lo + 12)))
        print("   ref ", np.round(ref[u, lo:lo + 12], 6))
        print("   out ", np.round(out[u, lo:lo + 12], 6))
        # how does the error evolve: sample every 4096
        print("   |d| at", [(int(j), float(f"{d[j]:.2e}")) for j in range(0, n, 8192)])
