import sys
sys.path.insert(0,'grail-rs_amd'); sys.path.insert(0,'tests')
import numpy as np, grail_hip as G, oracle_lib as O
from grail_hip import workload as W
ctx=G.Context(0); voices=W.single_voice(); ctx.set_voices(voices)
ov=[O.Voice.from_buffer_copy(bytes(v)) for v in voices]
ctx.set_option("arithmetic",1)
# empty batch
out,n=ctx.synthesize(np.zeros(0,dtype=G.PHONEME_DTYPE),[0],out_stride=64); print("empty ok", out.shape, n)
# no voice ids / seeds, scan kernel
segs,offs,vids,seeds=W.make_batch(5,length=0.03,blend_length=0.03125)
out,n=ctx.synthesize(segs,offs,None,None,out_stride=8192); print(ctx.last_kernel_name())
ref,rl=O.synthesize_batch(ov,segs,offs,None,None,8192)
print("null ids/seeds: lens equal",np.array_equal(n,rl),"max diff ulp",np.abs(out-ref).max()/2**-23)
# elems batch in fast mode
elems=[]
for s in segs:
    e=G.SequenceElem(); ph=int(s["phoneme"])
    if ph>=G.PH_A:
        e.has_elem=1; e.elem=voices[0].phonemes[ph-G.PH_A]; e.elem.frequency=min(float(s["frequency"]),0.5)
    else:
        e.has_elem=0
    e.length=float(s["length"]); e.blend_length=float(s["blend_length"]); elems.append(e)
o2,n2=ctx.synthesize_elems(elems,offs,None,None,out_stride=8192); print(ctx.last_kernel_name())
print("elems fast: lens equal",np.array_equal(n2,rl),"max diff ulp",np.abs(o2-ref).max()/2**-23)
ctx.set_option("arithmetic",0)
o3,n3=ctx.synthesize_elems(elems,offs,None,None,out_stride=8192); print(ctx.last_kernel_name(), "exact elems equal oracle", np.array_equal(o3.view(np.uint32),ref.view(np.uint32)))
# one-sample cap (out_stride tiny) in fast scan
ctx.set_option("arithmetic",1)
o4,n4=ctx.synthesize(segs,offs,vids,seeds,out_stride=64,allow_truncation=True); print("tiny stride", n4, ctx.last_kernel_name())
print("done")
