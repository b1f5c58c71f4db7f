#!/usr/bin/env python3
"""Static instruction histogram of k_pk_tab2's inner loops, from the gfx950 assembly scripts/pk_isa.sh writes.

    scripts/pk_isa.sh 'k_pk_tab2ILi64ELi4ELi2E' /tmp/tab2.s && python3 scripts/pk_tab2_isa_budget.py /tmp/tab2.s

Every innermost loop (a backward branch to a label with no other label in between) longer than 100 instructions is
listed with its opcode classes; profiles/r05_pk_tab2_isa_budget.txt multiplies the loops the benchmark's problem runs
(variants PKV_CROSS_CORE and PKV_AUTO_CORE, node rule on) by their trip counts and by the issue cost of each class."""
import collections
import re
import sys

CLASSES = [
    ('fp64 fma/mul/add', re.compile(r'v_(fma|fmac|mul|add)_f64')),
    ('fp64 rsq/rcp (quarter rate)', re.compile(r'v_(rsq|rcp|sqrt)_f64')),
    ('fp64 rndne/ldexp/cvt (exp range reduction)', re.compile(r'v_(rndne_f64|ldexp_f64|cvt_i32_f64|cvt_f64_i32|frexp)')),
    ('other VALU (moves, address arithmetic, compares)', re.compile(r'v_')),
    ('LDS reads', re.compile(r'ds_')),
    ('global loads', re.compile(r'global_load|buffer_load')),
    ('waitcnt / nop', re.compile(r's_waitcnt|s_nop')),
    ('other scalar', re.compile(r's_')),
]


def main(path):
    lines = open(path).read().split('\n')
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r'^(\.LBB[0-9_]+):', ln)
        if m:
            labels[m.group(1)] = i
    label_lines = sorted(labels.values())
    loops = []
    for i, ln in enumerate(lines):
        m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)', ln)
        if not m or labels.get(m.group(1), i + 1) > i:
            continue
        start = labels[m.group(1)]
        if any(start < x < i for x in label_lines):
            continue                                    # not innermost
        body = [l.split()[0] for l in lines[start + 1:i + 1] if l.startswith('\t') and not l.lstrip().startswith((';', '.'))]
        if len(body) > 100:
            loops.append((start + 1, i + 1, body))
    for start, end, body in loops:
        hist = collections.OrderedDict((name, 0) for name, _ in CLASSES)
        for op in body:
            for name, rx in CLASSES:
                if rx.match(op):
                    hist[name] += 1
                    break
        kind = ('extra nodes (one exponential per walker and node)' if hist[CLASSES[2][0]] else 'midpoints (recurrence)')
        print(f'loop at lines {start}-{end}: {len(body)} instructions, {kind}')
        for name, n in hist.items():
            print(f'    {n:5d}  {name}')
        top = collections.Counter(body).most_common(8)
        print('    opcodes: ' + ', '.join(f'{n} {op}' for op, n in top))


if __name__ == '__main__':
    main(sys.argv[1])
