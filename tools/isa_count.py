#!/usr/bin/env python3
"""Instruction counts of the loops of one kernel, from the ISA hipcc emits (make -C lightcurve_fitting_amd/csrc asm
writes csrc/lcf_hip.s).  Used to state, in DESIGN.md and bench.py, how many vector-ALU instructions the SHIPPED
band-sum loop issues per Planck sample (the unit of roofline.frac).
Usage: python tools/isa_count.py [mangled-name prefix]   (default: k_solo<5, 1, true, 2>)"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else '_ZN12_GLOBAL__N_16k_soloILi5ELi1ELb1ELi2ELb0E'
lines = open(os.path.join(ROOT, 'lightcurve_fitting_amd', 'csrc', 'lcf_hip.s')).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(prefix)][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))


def count(a, b):
    c = collections.Counter()
    for l in body[a:b + 1]:
        t = l.strip().split(' ')[0]
        if t.startswith('v_'):
            c['valu_f64' if 'f64' in t else 'valu_other'] += 1
        elif t.startswith('s_'):
            c['salu'] += 1
        elif t.startswith('ds_'):
            c['lds'] += 1
        elif t.startswith(('global_', 'scratch_', 'buffer_', 'flat_')):
            c['vmem'] += 1
    return c


print(f'{prefix}: {len(body)} lines, {len(loops)} loops')
for a, b in sorted(loops):
    c = count(a, b)
    if c['valu_f64'] + c['valu_other'] < 10:
        continue
    rcp = sum('v_rcp_f64' in l for l in body[a:b + 1])
    rd = sum(bool(re.search(r'ds_read_b128|ds_read2_b64', l)) for l in body[a:b + 1])
    print(f'  lines {a:6d}-{b:6d}: valu f64 {c["valu_f64"]:4d}, other valu {c["valu_other"]:4d}, salu {c["salu"]:4d}, '
          f'lds {c["lds"]:3d}, vmem {c["vmem"]:3d}, v_rcp_f64 {rcp}, 16-byte LDS reads {rd}')
