#!/usr/bin/env python3
"""List the hot (v_pk-heavy, < 700 lines) loops of a kernel with their instruction mix.
usage: loops.py file.s kernel-substring"""
import collections
import re
import sys

f, key = sys.argv[1], sys.argv[2]
lines = open(f).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and re.match(r'^_Z\S+:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end + 1]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))
for a, b in sorted(loops):
    if not (sum(1 for l in body[a:b + 1] if 'v_pk_' in l) > 50 and b - a < 700):
        continue
    cnt = collections.Counter()
    for l in body[a:b + 1]:
        t = l.strip().split()
        if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'):
            continue
        cnt[t[0]] += 1
    print(f'loop {a}-{b}: {sum(cnt.values())} instrs, pk {sum(c for o, c in cnt.items() if o.startswith("v_pk_"))}, '
          f'rcp {cnt["v_rcp_f32_e32"]}, accread {cnt["v_accvgpr_read_b32"]}, accwrite {cnt["v_accvgpr_write_b32"]}, '
          f'mov {cnt["v_mov_b32_e32"] + cnt["v_mov_b64_e32"]}, nop {cnt["s_nop"]}, salu {sum(c for o, c in cnt.items() if o.startswith("s_"))}')
