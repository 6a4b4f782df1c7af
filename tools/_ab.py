import os, sys
sys.path.insert(0, "/root/repo/grail-rs_amd")
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
stride = W.max_samples()
for nv, n in ((1, 65536), (8, 65536), (1, 32768)):
    ctx.set_voices(W.single_voice() if nv == 1 else W.preset_voices(8))
    segs, offs, vids, seeds = W.make_batch(n, n_voices=nv)
    batch = ctx.upload(segs, offs, vids, seeds)
    d_out = ctx.device_alloc(n * stride * 4); d_len = ctx.device_alloc(n * 4)
    ms = []
    for _ in range(5):
        batch.synthesize_async(d_out, stride, d_len); ctx.sync(); ms.append(ctx.last_kernel_ms())
    print("old" if os.environ.get("GRAIL_HIP_LIB") else "new", nv, n, f"{min(ms):.2f} ms", ctx.last_kernel_name(), flush=True)
    ctx.device_free(d_out); ctx.device_free(d_len); batch.free()
