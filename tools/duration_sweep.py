#!/usr/bin/env python3
"""the library's own choice of kernel family against every pinned family, over utterance lengths
0.25 / 1 / 8 / 30 s and batch sizes 256 ... 65 536, exact and fast arithmetic.  Prints kernel milliseconds (min of 2 after
a warm-up) and auto / best.   usage: duration_sweep.py [--presets] [--seconds 0.25,1,8,30] [--sizes 256,1024,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


presets = "--presets" in sys.argv
seconds = [float(x) for x in arg("--seconds", "0.25,1,8,30").split(",")]
sizes = [int(x) for x in arg("--sizes", "256,1024,4096,16384,65536").split(",")]
ctx = G.Context(0)
voices = W.preset_voices(8) if presets else W.single_voice()
ctx.set_voices(voices)
DEFAULTS = {"arithmetic": 0, "lanes_per_utterance": 0, "time_parallel_scan": 1, "time_split": 1,
            "time_split_min_utterances": -1, "time_parallel_scan_max_utterances": -1, "small_batch_pipeline": 1,
            "composite_launches": 1}
VARIANTS = {
    0: [("auto", {}), ("pipe", {"composite_launches": 0}), ("L8", {"lanes_per_utterance": 8}), ("L4", {"lanes_per_utterance": 4}),
        ("L2", {"lanes_per_utterance": 2}), ("L1", {"lanes_per_utterance": 1})],
    1: [("auto", {}), ("scan", {"time_split": 0, "time_parallel_scan_max_utterances": 1 << 20, "composite_launches": 0}),
        ("split", {"time_parallel_scan": 0, "time_split_min_utterances": 0, "composite_launches": 0}),
        ("L8", {"lanes_per_utterance": 8}), ("L4", {"lanes_per_utterance": 4}), ("L2", {"lanes_per_utterance": 2}),
        ("L1", {"lanes_per_utterance": 1}), ("pipe", {"time_split": 0, "time_parallel_scan": 0, "composite_launches": 0})],
}
print(f"# duration_sweep: {'8 presets (eight live formants)' if presets else 'voices::generic() (four live formants)'}, 48 kHz, "
      f"compute_units={ctx.get_option('compute_units')}; kernel ms, min of 2 after a warm-up (the library's own choice: min of 4, measured first and last); * = the library's choice is within 10 % of the best")
worst = 0.0
for sec in seconds:
    nseg = max(2, int(round(sec / 0.5)))
    seg_len = sec / nseg
    stride = W.max_samples(segments=nseg, length=seg_len)
    for n in sizes:
        pcm16 = n * stride * 4 > 200e9
        if n * stride * 2 > 200e9:
            print(f"{sec:5.2f} s x {n:6d}: skipped (the rows would not fit the device)")
            continue
        segs, offs, vids, seeds = W.make_batch(n, n_voices=len(voices), segments=nseg, length=seg_len,
                                               blend_length=min(0.5, 2.0 ** np.floor(np.log2(seg_len))))
        batch = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * (2 if pcm16 else 4))
        d_len = ctx.device_alloc(n * 4)
        for fast in (0, 1):
            res = {}
            for name, opts in VARIANTS[fast]:
                for k, v in DEFAULTS.items():
                    ctx.set_option(k, v)
                ctx.set_option("arithmetic", fast)
                for k, v in opts.items():
                    ctx.set_option(k, v)
                # pinned lane mappings that would need more than four rounds are not contenders
                if name.startswith("L") and n * int(name[1:]) > 4 * 65536:
                    continue
                ms = []
                for rep in range(3):
                    if pcm16:
                        batch.synthesize_pcm16_async(d_out, stride, d_len)
                    else:
                        batch.synthesize_async(d_out, stride, d_len)
                    ctx.sync()
                    if rep:
                        ms.append(ctx.last_kernel_ms())
                kern = ctx.last_kernel_name().replace("synth_kernel", "k")
                blocks = ctx.get_option("last_launch_blocks")
                chunks = ctx.get_option("last_launch_chunks")
                what = ("scan" if "scan" in kern else "split%d" % chunks if chunks else "pipe" if "PIPE" in kern
                        else "L%d" % ctx.get_option("last_launch_lanes")) + ("" if blocks == 1 else "+%d" % (blocks - 1))
                # a variant that fell through to another family is that family: keep the label honest
                res[name] = (min(ms), what)
            # the library's own choice once more at the end: the first variant of a cell starts behind the upload
            # (an idle device lowers its clocks: +5 - 7 % on the first launches)
            for k, v in DEFAULTS.items():
                ctx.set_option(k, v)
            ctx.set_option("arithmetic", fast)
            for rep in range(2):
                (batch.synthesize_pcm16_async if pcm16 else batch.synthesize_async)(d_out, stride, d_len)
                ctx.sync()
                res["auto"] = (min(res["auto"][0], ctx.last_kernel_ms()), res["auto"][1])
            best = min(v[0] for k, v in res.items())
            ratio = res["auto"][0] / best
            worst = max(worst, ratio)
            cells = "  ".join(f"{k}[{v[1]}] {v[0]:8.3f}" for k, v in res.items())
            print(f"{sec:5.2f} s x {n:6d} {'fast ' if fast else 'exact'}{' i16' if pcm16 else ''}: auto/best {ratio:5.2f}{'*' if ratio <= 1.10 else ' '}  {cells}",
                  flush=True)
        ctx.device_free(d_out)
        ctx.device_free(d_len)
        batch.free()
print(f"# worst auto / best over the sweep: {worst:.2f}")
