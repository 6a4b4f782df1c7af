#!/usr/bin/env python3
"""i16 rows against `(x * 32767) as i16` of the f32 rows of the same utterances, full 2-second utterances through
the pipelined workgroups (four and eight formants) and the one-lane kernel inside a 70 000-utterance batch."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "grail-rs_amd"))
import numpy as np
import grail_hip as G
from grail_hip import workload as W
ctx = G.Context(0)
for nv, n in ((1, 2048), (8, 1024), (1, 70000)):
    voices = W.single_voice() if nv == 1 else W.preset_voices(8)
    ctx.set_voices(voices)
    segs, offs, vids, seeds = W.make_batch(n, n_voices=nv)
    stride = W.max_samples()
    take = min(n, 2048)
    f32, n32 = ctx.synthesize(segs[:offs[take]], offs[:take + 1], vids[:take], seeds[:take], out_stride=stride)
    if n > 2048:   # the same utterances inside a big batch (one lane per utterance), rows read back from the device
        b = ctx.upload(segs, offs, vids, seeds)
        d = ctx.device_alloc(n * stride * 2); dl = ctx.device_alloc(n * 4)
        b.synthesize_pcm16_async(d, stride, dl); ctx.sync()
        i16 = np.zeros((take, stride), dtype=np.int16); ctx.d2h(i16, d, take * stride * 2)
        n16 = np.zeros(take, dtype=np.uint32); ctx.d2h(n16, dl, take * 4)
        name = ctx.last_kernel_name()
        ctx.device_free(d); ctx.device_free(dl); b.free()
    else:
        i16, n16 = ctx.synthesize_pcm16(segs, offs, vids, seeds, out_stride=stride)
        name = ctx.last_kernel_name()
    assert np.array_equal(n32[:take], n16[:take])
    want = np.clip(np.trunc(f32.astype(np.float32) * np.float32(32767.0)), -32768, 32767).astype(np.int16)
    bad = 0
    for u in range(take):
        bad += int(np.count_nonzero(i16[u, :n16[u]] != want[u, :n16[u]]))
    print(f"voices={nv} n={n}: {take} utterances x {int(n16[0])} samples, i16 rows vs converted f32 rows: {bad} differences  ({name})")
    assert bad == 0
