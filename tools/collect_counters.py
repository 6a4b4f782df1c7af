#!/usr/bin/env python3
"""Collect the PMC evidence behind bench.py's `roofline.traffic` / `roofline_valu` for every reported
workload, each counter group in its own rocprofv3 pass (MI355X_MICROARCH.md, HBM section: WRITE_SIZE is
exact for 16-B-per-lane streaming stores, FETCH_SIZE counts half the bytes of wide coalesced reads on
gfx950 and is doubled), and stamp every entry with the kernel instantiation and the sha of the kernel
sources it was measured on.  bench.py reports an entry only when both match the running build.

Runs ON THE GPU BOX (needs rocprofv3):   python3 tools/collect_counters.py [out.json]
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "traffic.json")
WORKLOADS = [   # key (as bench.py builds it), bench arguments
    ("config3_utts65536", ["--config", "3", "--mode", "exact"]),
    ("config3_utts65536_fast", ["--config", "3", "--mode", "fast"]),
    ("config4_utts65536", ["--config", "4", "--mode", "exact"]),
    ("config4_utts65536_fast", ["--config", "4", "--mode", "fast"]),
    ("config3_utts65536_mid", ["--config", "3", "--mode", "mid"]),
    ("config2_utts4096", ["--config", "2", "--mode", "exact"]),
    ("config2_utts4096_fast", ["--config", "2", "--mode", "fast"]),
    ("config3_utts65536_pcm16", ["--config", "3", "--mode", "exact", "--pcm16"]),
    # the regime the time-parallel scan kernel still serves (fast arithmetic, up to 1 536 utterances)
    ("config3_utts1024_fast", ["--utts", "1024", "--mode", "fast"]),
    # the speech-like corpus of bench.py's `other_configs.speech_like` (not a BASELINE config), as the timed batch
    ("speech_like_utts65536", ["--corpus", "speech", "--mode", "exact"]),
    ("speech_like_utts65536_fast", ["--corpus", "speech", "--mode", "fast"]),
]
PASSES = [["WRITE_SIZE"], ["FETCH_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_BUSY_CYCLES"]]
COMMON = ["--fast-leg", "0", "--other-configs", "0", "--cpu-utts", "0", "--steps", "2", "--warmup", "0", "--ramp", "0"]
env = dict(os.environ, TMPDIR="/tmp")


def pmc_pass(tag, counters, args):
    d = os.path.join(ROOT, "gpurun_out", "prof", tag)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    # the program itself follows `--` (no shell, no env wrapper): the profiler has already initialised the GPU
    cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", "python3",
                                                os.path.join(ROOT, "bench.py")] + args + COMMON
    subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        return {}
    agg, dispatches = {}, set()
    for r in csv.DictReader(open(files[0])):
        if "synth_kernel" in r["Kernel_Name"] or "scan_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            dispatches.add(r["Dispatch_Id"])
    n = max(len(dispatches), 1)
    return {k: v / n for k, v in agg.items()}


entries = {}
for key, args in WORKLOADS:
    line = subprocess.run(["python3", os.path.join(ROOT, "bench.py")] + args + COMMON, cwd=ROOT, env=env,
                          capture_output=True, text=True).stdout.strip().splitlines()[-1]
    b = json.loads(line)
    c = {}
    for i, counters in enumerate(PASSES):
        c.update(pmc_pass(f"{key}_p{i}", counters, args))
    e = {
        "kernel": b["roofline"]["kernel"], "kernel_source_sha": b["roofline"]["kernel_source_sha"],
        "kernel_ms_unprofiled": b["roofline"]["kernel_ms"],
        "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
        "write_size_kb": c.get("WRITE_SIZE"), "fetch_size_kb": c.get("FETCH_SIZE"),
        "hbm_bytes": (c["WRITE_SIZE"] * 1024.0 + 2.0 * c["FETCH_SIZE"] * 1024.0)
        if "WRITE_SIZE" in c and "FETCH_SIZE" in c else None,
        "valu_insts": c.get("SQ_INSTS_VALU"), "salu_insts": c.get("SQ_INSTS_SALU"), "waves": c.get("SQ_WAVES"),
        "command": "rocprofv3 --pmc <group> -- python3 bench.py " + " ".join(args + COMMON),
    }
    entries[key] = e
    print(key, json.dumps(e), flush=True)
doc = {
    "_doc": "HBM bytes per launch = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950: FETCH_SIZE reads half the bytes of a "
            "wide coalesced read; WRITE_SIZE is exact for 16-B-per-lane streaming stores), each counter group collected "
            "in its own rocprofv3 --pmc pass by tools/collect_counters.py; per-dispatch averages over the synthesis "
            "kernel's launches.  bench.py reports an entry only if `kernel_source_sha` and `kernel` match the running build.",
    "entries": entries,
}
os.makedirs(os.path.dirname(OUT), exist_ok=True)
with open(OUT, "w") as f:
    json.dump(doc, f, indent=1)
print("wrote", OUT)
