#!/usr/bin/env python3
"""Find the hot inner loop (the innermost backward branch enclosing > 50 v_pk instructions) of a
kernel in a .s file and print its instruction mix.   usage: hotloop.py file.s kernel-substring"""
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


def npk(a, b):
    return sum(1 for l in body[a:b + 1] if 'v_pk_' in l)


best = None
for a, b in loops:
    if npk(a, b) > 50 and (best is None or (b - a) < (best[1] - best[0])):
        best = (a, b)
a, b = best
cnt = collections.Counter()
for l in body[a:b + 1]:
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'):
        continue
    cnt[t[0]] += 1
valu = sum(c for o, c in cnt.items() if o.startswith('v_'))
salu = sum(c for o, c in cnt.items() if o.startswith('s_'))
print(f'loop lines {a}-{b}: {sum(cnt.values())} instrs, {valu} VALU, {salu} SALU, '
      f'pk {sum(c for o, c in cnt.items() if o.startswith("v_pk_"))}, rcp {cnt["v_rcp_f32_e32"]}, '
      f'mov {sum(c for o, c in cnt.items() if o.startswith("v_mov"))}, nop {cnt["s_nop"]}, '
      f'scratch {sum(c for o, c in cnt.items() if "scratch" in o)}, ds {sum(c for o, c in cnt.items() if o.startswith("ds_"))}')
