#!/usr/bin/env python3
"""Same-box A/B of two builds of libgrail_hip.so: kernel time of a few batches with the library in the tree,
with another build (argument: its path; e.g. `make OUT=lib/libgrail_hip_old.so` of an older checkout), and
with the first again.  Kernel times differ by +-3 % between the boxes of the pool, so a comparison of two
builds only means something inside one session.  Each library runs in a process of its own.

usage: ab_libs.py <other libgrail_hip.so> [exact|fast]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "grail-rs_amd"))
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
ctx.set_option("arithmetic", 1 if sys.argv[1] == "fast" else 0)
stride = W.max_samples()
for nv, n in ((1, 65536), (8, 65536), (1, 32768), (1, 16384), (1, 4096), (8, 4096)):
    ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
    segs, offs, vids, seeds = W.make_batch(n, n_voices=nv)
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
    ms = []
    for _ in range(5):
        batch.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(ctx.last_kernel_ms())
    print(sys.argv[2], "voices=%%d n=%%6d  %%7.2f ms  %%s" %% (nv, n, min(ms), ctx.last_kernel_name()), flush=True)
    ctx.device_free(d_out); ctx.device_free(d_len); batch.free()
''' % ROOT

other = os.path.abspath(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "exact"
for label, lib in (("tree ", None), ("other", other), ("tree ", None)):
    env = dict(os.environ)
    if lib:
        env["GRAIL_HIP_LIB"] = lib
    else:
        env.pop("GRAIL_HIP_LIB", None)
    subprocess.run([sys.executable, "-c", WORKER, mode, label], env=env, check=True)
