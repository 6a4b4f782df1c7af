#!/usr/bin/env python3
"""Loops of each kernel in a gfx950 .s file (make asm UNIT=...): the innermost loops (largest VALU count first), then
the largest enclosing ones — where the hot loops are and how many instructions a trip issues.   usage: isa_loops.py file.s [kernel_index]"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
starts = [i for i, l in enumerate(lines) if re.match(r'^\s*\.type\s+_ZN5grail.*synth_kernel.*@function', l)]
for ki, st in enumerate(starts):
    en = starts[ki + 1] if ki + 1 < len(starts) else len(lines)
    if len(sys.argv) > 2 and int(sys.argv[2]) != ki:
        continue
    labels = {}
    for i in range(st, en):
        m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
        if m:
            labels[m.group(1)] = i
    loops = []
    for i in range(st, en):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', lines[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            body = lines[a:i + 1]
            c = lambda pat: sum(1 for l in body if re.match(pat, l))
            loops.append((a + 1, i + 1, c(r'\s+v_'), c(r'\s+v_readlane'), c(r'\s+v_rcp'), c(r'\s+v_pk_'), c(r'\s+s_'),
                          c(r'\s+ds_'), c(r'\s+v_accvgpr|\s+scratch_|\s+buffer_')))
    print("kernel", ki, re.sub(r'.*synth_kernelI', 'synth_kernel<', lines[st])[:80])
    # innermost loops first (a loop that contains no other: where the time goes), then the largest enclosing ones —
    # a listing of the largest loops alone hides the small hot ones behind the big rare ones
    inner = [lp for lp in loops if not any(o is not lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    for title, sel in (("innermost", sorted(inner, key=lambda x: -x[2])[:12]),
                       ("enclosing", sorted([lp for lp in loops if lp not in inner], key=lambda x: -x[2])[:4])):
        for lp in sel:
            print("   %s lines %6d-%6d: valu %4d (readlane %d, rcp %d, packed %d)  salu %3d  lds %2d  accvgpr/scratch %d" % ((title,) + lp))
