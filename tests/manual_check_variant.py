"""Quick parity + timing check of a kernel variant on the GPU box: tests/manual_check_variant.py <variant>"""
import sys
sys.path.insert(0, 'grail-rs_amd'); sys.path.insert(0, 'tests')
import numpy as np
import grail_hip as G, oracle_lib as O
from grail_hip import workload as W
from test_parity_gpu import edge_case_batch
variant = int(sys.argv[1])
ctx = G.Context(0)
ctx.set_option("kernel_variant", variant)
ok = True
for name, voices, (segs, offs, vids, seeds), stride in [
    ("short150", W.single_voice(), W.make_batch(150, length=0.03, blend_length=0.03), W.max_samples(length=0.03)),
    ("presets", W.preset_voices(8), W.make_batch(200, n_voices=8, length=0.05, blend_length=0.05), W.max_samples(length=0.05)),
]:
    ctx.set_voices(voices)
    out, ln = ctx.synthesize(segs, offs, vids, seeds, out_stride=stride)
    ov = [O.Voice.from_buffer_copy(bytes(v)) for v in voices]
    ref, rl = O.synthesize_batch(ov, segs, offs, vids, seeds, stride)
    same = np.array_equal(ln, rl) and all(np.array_equal(out[u, :ln[u]].view(np.uint32), ref[u, :rl[u]].view(np.uint32)) for u in range(len(ln)))
    print(name, "bit-identical" if same else "MISMATCH"); ok &= same
segs, offs = edge_case_batch(48000.0)
n = len(offs) - 1
seeds = np.arange(n, dtype=np.uint32) * 977
voices = W.single_voice(); ctx.set_voices(voices)
with np.errstate(all="ignore"):
    out, ln = ctx.synthesize(segs, offs, None, seeds, out_stride=20032)
    ref, rl = O.synthesize_batch([O.Voice.from_buffer_copy(bytes(voices[0]))], segs, offs, None, seeds, 20032)
same = np.array_equal(ln, rl) and all(np.array_equal(out[u, :ln[u]].view(np.uint32), ref[u, :rl[u]].view(np.uint32)) for u in range(n))
print("edge cases", "bit-identical" if same else "MISMATCH", ln.tolist() if not same else ""); ok &= same
print("ALL OK" if ok else "FAILED")
