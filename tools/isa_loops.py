#!/usr/bin/env python3
"""Loops of each kernel in a gfx950 .s file (make asm UNIT=...), largest VALU count first: where the hot loops are and
how many instructions a trip issues.   usage: isa_loops.py file.s [kernel_index]"""
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
    for lp in sorted(loops, key=lambda x: -x[2])[:8]:
        print("   lines %6d-%6d: valu %4d (readlane %d, rcp %d, packed %d)  salu %3d  lds %2d  accvgpr/scratch %d" % lp)
