import sys; sys.path.insert(0,'grail-rs_amd'); sys.path.insert(0,'tests')
import grail_hip as G, numpy as np
from grail_hip import workload as W
ctx=G.Context(0); ctx.set_voices(W.single_voice())
segs,offs,v,s=W.make_batch(256, length=0.05, blend_length=0.05)
out,ln=ctx.synthesize(segs,offs,v,s,out_stride=W.max_samples(length=0.05))
print('len',ln[:4],'slow steps',ctx.get_option('slow_division_wave_steps'))
