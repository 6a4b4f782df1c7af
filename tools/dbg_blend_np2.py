import os, sys
sys.path.insert(0, "grail-rs_amd")
import numpy as np
import grail_hip as G
from grail_hip import workload as W
n = 65536
ctx = G.Context(0)
for bl in (0.5, 0.3):
    for nv in (1, 8):
        voices = W.single_voice() if nv == 1 else W.preset_voices(8)
        ctx.set_voices(voices)
        segs, offs, vids, seeds = W.make_batch(n, n_voices=nv, blend_length=bl)
        stride = W.max_samples()
        batch = ctx.upload(segs, offs, vids, seeds)
        d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
        ms = []
        for _ in range(2):
            batch.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(ctx.last_kernel_ms())
        print(f"blend_length {bl} voices {nv}: {min(ms):.2f} ms slow_division_wave_steps={ctx.get_option('slow_division_wave_steps')}", flush=True)
        ctx.device_free(d_out); ctx.device_free(d_len); batch.free()
