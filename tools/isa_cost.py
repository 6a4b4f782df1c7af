#!/usr/bin/env python3
"""Static cost estimate of a straight-line hot path in gfx950 ISA.
usage: isa_cost.py file.s start:end [start:end ...]   (1-based inclusive line ranges)
Cost table (SIMD pipe cycles per wave64 instruction) measured with tools/valu_microbench.hip."""
import re, sys, collections
COST2 = {'v_mul_f32','v_add_f32','v_sub_f32','v_subrev_f32','v_fma_f32','v_fmac_f32','v_mac_f32'}
COST8 = {'v_rcp_f32','v_rsq_f32','v_sqrt_f32','v_exp_f32','v_log_f32'}
def cost(op):
    base = re.sub(r'_(e32|e64|dpp|sdwa)$','',op)
    if base in COST2: return 2
    if base in COST8: return 8
    if base.startswith('v_'): return 4
    return 0
def main():
    f=sys.argv[1]; lines=open(f).read().splitlines()
    tot=collections.Counter(); n=collections.Counter()
    for r in sys.argv[2:]:
        a,b=map(int,r.split(':'))
        for l in lines[a-1:b]:
            t=l.strip().split()
            if not t or t[0].startswith(('.',';')) or t[0].endswith(':'): continue
            op=t[0]; n[op]+=1; tot[op]+=cost(op)
    valu=sum(c for o,c in n.items() if o.startswith('v_'))
    print('instructions',sum(n.values()),'valu',valu,'pipe cycles',sum(tot.values()))
    for o,c in n.most_common(40): print(f'  {o:28s} n={c:4d} cyc={tot[o]}')
main()
