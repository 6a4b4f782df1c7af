#!/usr/bin/env python3
"""Per-kernel start / duration table of a rocprofv3 --kernel-trace CSV (synthesis kernels only).  usage: trace_summary.py file.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    if "synth_kernel" not in name and "scan_kernel" not in name:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = name[name.index("synth_kernel") if "synth_kernel" in name else name.index("scan_kernel"):][:72]
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e6:8.3f}"
    print(f"{short:72s} grid {int(r['Grid_Size_X']):8d} start {(s - t0) / 1e6:10.3f} dur {(e - s) / 1e6:8.3f} ms  {gap}")
    prev_end = e
