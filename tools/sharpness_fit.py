#!/usr/bin/env python3
"""grail_fast_sharpness() against the measured tables of tools/sharpness_data.py: how often and by how much the
prediction lies below the measured deviation, what fraction of the random tables the limit serves and the worst
deviation among those.   usage: sharpness_fit.py tables.jsonl[.gz] ..."""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G


def predicted(row):
    f = np.array([ph["f"] for ph in row["phonemes"]])
    w = np.array([ph["bw"] for ph in row["phonemes"]])
    a = np.abs(np.array([ph["amp"] for ph in row["phonemes"]]))
    share = (a / a.sum(axis=1, keepdims=True)).max(axis=0)
    sens = ((0.0709 / w) * (1.0 + (f / 0.075) ** 2)).max(axis=0)
    return float(np.sqrt(((share * sens)[share > 0] ** 2).sum()))


rows = []
for path in sys.argv[1:]:
    with (gzip.open(path, "rt") if path.endswith(".gz") else open(path)) as fh:
        rows += [json.loads(line) for line in fh]
k = np.array([r["k"] for r in rows])
s = np.array([predicted(r) for r in rows])

print(f"{len(rows)} random one-voice tables; measured deviation (lane kernel, limit lifted): median {np.median(k):.1f}, "
      f"p90 {np.percentile(k, 90):.1f}, p99 {np.percentile(k, 99):.1f}, max {k.max():.1f}  (units of 2^-23 of max(1, peak))")
ratio = k / s
print(f"measured / predicted: median {np.median(ratio):.2f}, p99 {np.percentile(ratio, 99):.2f}, p99.5 {np.percentile(ratio, 99.5):.2f}, "
      f"max {ratio.max():.2f}; tables above the prediction: {(ratio > 1).mean() * 100:.2f} %")
for limit in (24.0, G.FAST_SHARPNESS_LIMIT, 32.0, 40.0, 48.0):
    served = s <= limit
    print(f"limit {limit:4.0f}: serves {served.mean() * 100:4.1f} % of the tables; among them worst measured {k[served].max():5.1f}, "
          f"p99 {np.percentile(k[served], 99):5.1f}" + ("   <- GRAIL_FAST_SHARPNESS_LIMIT" if limit == G.FAST_SHARPNESS_LIMIT else ""))
