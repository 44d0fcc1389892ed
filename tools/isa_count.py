#!/usr/bin/env python3
"""Vector-ALU instruction counts of the hot loops, from the ISA hipcc emits for the shipped library.

The Makefile of lightcurve_fitting_amd/csrc builds with -save-temps and runs this on the device assembly; the result
(csrc/liblcf_hip.isa.json) is what bench.py's roofline.frac is computed from -- the unit of work of every bench line
is counted in the binary that runs, not typed into bench.py.

    python tools/isa_count.py --asm <lcf_hip-...-gfx950.s> --json        # what the build does
    python tools/isa_count.py [--asm ...] [mangled-name prefix]          # human-readable loop listing of one kernel

What is counted (all: vector-ALU instructions, FP64 and other, per unit):
  quad_main / quad_safe   one trip of the band-sum loop over four Planck samples (generic k_solo<5,...>: the loop with
                          one v_rcp_f64, LDS table reads and no vector-memory access; `safe` = the e^-x form)
  point_lean              one interpolated data point of the model-specialised kernel: the straight-line block with the
                          epoch's 24 coefficient reads (ds_read_b128) and its interval (v_fract_f64), divided by 6 points
  state_lean              one log-space thermal state of that kernel: the blocks between the barrier behind the serial
                          head and the point block that hold the logarithm and the exponential
  log_lean                the logarithm alone (the block of state_lean with v_frexp_mant_f64): the part of a thermal
                          state every model with a log-space state pays
Models outside the specialised kernels (the companion-shocking fit) are priced with point_lean per interpolated point and
log_lean per state: what every such point / state executes at least -- a lower bound, as roofline.frac is meant to be.
"""
import argparse
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GENERIC = '_ZN12_GLOBAL__N_16k_soloILi5ELi1ELb1ELi2ELb0ELi0E'
LEAN = '_ZN12_GLOBAL__N_16k_soloILi5ELi1ELb1ELi2ELb0ELi1E'
#: the kernel the headline TIMES: the same inlined half-step inside the loop of the resident launch (its registers are
#: allocated differently -- it spills -- so it is counted itself)
RUN = '_ZN12_GLOBAL__N_110k_solo_runILi5ELi1ELb1ELi2ELi1ELb0E'


def kernel_body(lines, prefix):
    start = [i for i, l in enumerate(lines) if l.startswith(prefix)]
    if not start:
        raise SystemExit(f'isa_count: no kernel {prefix} in the assembly')
    start = start[0]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    return [l for l in lines[start:end] if l.strip() and not l.strip().startswith(';')]


def count(instrs):
    c = collections.Counter()
    for l in instrs:
        t = l.strip().split(' ')[0]
        if t.startswith('v_'):
            c['valu_f64' if 'f64' in t else 'valu_other'] += 1
        elif t.startswith('s_'):
            c['salu'] += 1
        elif t.startswith('ds_'):
            c['lds'] += 1
        elif t.startswith(('global_', 'scratch_', 'buffer_', 'flat_')):
            c['vmem'] += 1
    c['valu'] = c['valu_f64'] + c['valu_other']
    return c


def loops_of(body):
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    out = []
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return sorted(out)


def blocks_of(body):
    """Basic blocks in layout order: (first line index, instructions)."""
    blocks, cur, first = [], [], 0
    for i, l in enumerate(body):
        if re.match(r'^\.LBB\d+_\d+:', l):
            if cur:
                blocks.append((first, cur))
            cur, first = [], i
            continue
        if not cur:
            first = i
        cur.append(l.strip())
        if re.match(r's_(c?branch|endpgm|swappc|setpc)', l.strip()):
            blocks.append((first, cur))
            cur = []
    if cur:
        blocks.append((first, cur))
    return blocks


def quad_loops(body):
    """(main, safe) VALU counts of the four-sample band-sum loops: innermost loops with exactly one v_rcp_f64, LDS reads
    and no vector-memory access; the e^x form is the shorter one in FP64 instructions."""
    found = []
    for a, b in loops_of(body):
        seg = body[a:b + 1]
        c = count(seg)
        if sum('v_rcp_f64' in l for l in seg) == 1 and c['vmem'] == 0 and c['lds'] >= 4 and c['valu'] < 120:
            found.append((c['valu_f64'], c['valu']))
    if not found:
        raise SystemExit('isa_count: no four-sample band-sum loop found')
    found = sorted(set(found))
    return found[0][1], found[-1][1]


def lean_counts(body):
    blocks = blocks_of(body)
    idx = [k for k, (_, b) in enumerate(blocks)
           if sum('ds_read_b128' in x for x in b) == 24 and any('v_fract_f64' in x for x in b)]
    if len(idx) != 1:
        raise SystemExit(f'isa_count: expected one lean point block, found {len(idx)}')
    k = idx[0]
    point = count(blocks[k][1])['valu'] / 6.
    # the thermal state: back from the point block to the barrier behind the head, the blocks with FP64 arithmetic
    state, log_only = 0, 0
    j = k - 1
    while j >= 0 and not any(x.startswith('s_barrier') for x in blocks[j][1]):
        c = count(blocks[j][1])
        if c['valu_f64'] >= 4:
            state += c['valu']
            if any('v_frexp_mant_f64' in x for x in blocks[j][1]):
                log_only += c['valu']
        j -= 1
    if not log_only:
        raise SystemExit('isa_count: no logarithm between the barrier and the point block')
    return point, state, log_only


def report(lines):
    generic, lean = kernel_body(lines, GENERIC), kernel_body(lines, LEAN)
    main, safe = quad_loops(generic)
    point, state, log_only = lean_counts(lean)
    out = {'quad_main': main, 'quad_safe': safe, 'point_lean': point, 'state_lean': state, 'log_lean': log_only,
           'kernels': {'generic': GENERIC, 'lean': LEAN}}
    try:   # the resident launch's own counts (bench.py prefers them for the kernel it times)
        rp, rs, rl = lean_counts(kernel_body(lines, RUN))
        out.update({'point_run': rp, 'state_run': rs, 'log_run': rl})
        out['kernels']['run'] = RUN
    except SystemExit as exc:
        out['run_note'] = f'k_solo_run not counted ({exc}): its points and states are priced with the lean kernel\'s counts'
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--asm', default=os.path.join(ROOT, 'lightcurve_fitting_amd', 'csrc', 'lcf_hip.s'))
    ap.add_argument('--json', action='store_true')
    ap.add_argument('prefix', nargs='?', default=LEAN)
    args = ap.parse_args()
    lines = open(args.asm).read().split('\n')
    if args.json:
        print(json.dumps(report(lines), indent=1))
        return
    body = kernel_body(lines, args.prefix)
    print(f'{args.prefix}: {len(body)} lines')
    for a, b in loops_of(body):
        c = count(body[a:b + 1])
        if c['valu'] < 10:
            continue
        rcp = sum('v_rcp_f64' in l for l in body[a:b + 1])
        rd = sum(bool(re.search(r'ds_read_b128|ds_read2_b64', l)) for l in body[a:b + 1])
        print(f'  loop  {a:6d}-{b:6d}: valu f64 {c["valu_f64"]:4d}, other valu {c["valu_other"]:4d}, salu {c["salu"]:4d}, '
              f'lds {c["lds"]:3d}, vmem {c["vmem"]:3d}, v_rcp_f64 {rcp}, 16-byte LDS reads {rd}')
    for first, b in blocks_of(body):
        c = count(b)
        if c['valu'] >= 40:
            print(f'  block {first:6d}: valu f64 {c["valu_f64"]:4d}, other valu {c["valu_other"]:4d}, lds {c["lds"]:3d}, '
                  f'vmem {c["vmem"]:3d}')


if __name__ == '__main__':
    sys.exit(main())
