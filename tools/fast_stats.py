import sys, os
sys.path.insert(0, "grail-rs_amd")
import numpy as np, grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0); ctx.set_voices(W.single_voice())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536; stride = W.max_samples()
segs, offs, vids, seeds = W.make_batch(n)
b = ctx.upload(segs, offs, vids, seeds)
d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
ctx.set_option("arithmetic", 1); ctx.set_option("lanes_per_utterance", 1)
for i in range(3):
    t0 = ctx.get_option("fast_wave_tiles"); g0 = ctx.get_option("general_wave_steps")
    b.synthesize_async(d_out, stride, d_len); ctx.sync()
    print(ctx.last_kernel_name(), ctx.last_kernel_ms(), "fast tiles", ctx.get_option("fast_wave_tiles") - t0, "general wave steps", ctx.get_option("general_wave_steps") - g0)
