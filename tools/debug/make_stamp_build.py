#!/usr/bin/env python3
"""Diagnostic build of the library with s_memtime stamps inside k_fused (never shipped).
Writes build_variants/liblcf_stamps.so; tools/debug/read_stamps.py prints the median ticks per segment.
Stamps are taken by thread 0 of the part-0 workgroups of the first 64 slots:
  0 kernel entry | 1 draw record loaded | 2 accept tests done | 3 proposal ready | 4 logarithms | 5 coefficients |
  6 serial part written to LDS | 7 after the barrier (tables staged) | 8 thermal states in LDS | 9 points done |
  10 partial sum stored"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, 'lightcurve_fitting_amd/csrc/lcf_hip.hip')).read()


def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a[:70])
    s = s.replace(a, b, 1)


DECL = ('__device__ unsigned long long g_stamps[64 * 12];\n__device__ int g_stamp_on;\n'
        '#define STAMP(k) do { if (stamp_on) { unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" '
        ': "=s"(t_) :: "memory"); if (threadIdx.x == 0) g_stamps[stamp_slot * 12 + (k)] = t_; } } while (0)\n')
rep("// The serial part of a half-step for slot i, executed by ONE wave", DECL + "// The serial part of a half-step for slot i, executed by ONE wave")
# step_serial: stamps only when called from a stamped workgroup (flag passed through a device global set per launch)
rep("    const int prev_wid = (have_prev && primary) ? prev_draws[i].wid : 0;\n",
    "    const bool stamp_on = primary && mine && sq != nullptr && i < 64 && have_prev;\n    const int stamp_slot = i;\n"
    "    const int prev_wid = (have_prev && primary) ? prev_draws[i].wid : 0;\n")
rep("    if (have_next) dr = draws[i];\n", "    if (have_next) dr = draws[i];\n    STAMP(1);\n")
rep("    if (lane == 0 && rslot >= 0) {  // commit of the previous half-step's slot i", "    STAMP(2);\n    if (lane == 0 && rslot >= 0) {  // commit of the previous half-step's slot i")
rep("        const double lg = log(arg);  // one logarithm per lane, all at once", "        STAMP(3);\n        const double lg = log(arg);  // one logarithm per lane, all at once")
rep("        double c[kNCoef];\n        walker_coefficients(pb, q, lq, c);", "        STAMP(4);\n        double c[kNCoef];\n        walker_coefficients(pb, q, lq, c);\n        STAMP(5);")
# k_fused
rep("    const int part = blockIdx.x / n_own, i = lo + blockIdx.x % n_own;\n    const bool reddened = pb.model == kShockCooling3;\n",
    "    const int part = blockIdx.x / n_own, i = lo + blockIdx.x % n_own;\n    const bool reddened = pb.model == kShockCooling3;\n"
    "    const bool stamp_on = part == 0 && i < 64 && have_prev;\n    const int stamp_slot = i;\n    STAMP(0);\n")
rep("    __syncthreads();\n    if (sc[kNCoef] == -INFINITY) return;  // prior excludes the proposal: likelihood skipped (fitting.py:125)\n    double cs[kNCoef];",
    "    STAMP(6);\n    __syncthreads();\n    STAMP(7);\n    if (sc[kNCoef] == -INFINITY) return;  // prior excludes the proposal: likelihood skipped (fitting.py:125)\n    double cs[kNCoef];")
rep("    if (THERM || reddened) __syncthreads();\n    const double term = points_loop<VARIANT, 0, true, THERM>(pb, part, 0, sq, cs, lth, e0, ltab, fdesc, ExpTab{exptab},\n                                                             nullptr, nullptr);\n",
    "    if (THERM || reddened) __syncthreads();\n    STAMP(8);\n    const double term = points_loop<VARIANT, 0, true, THERM>(pb, part, 0, sq, cs, lth, e0, ltab, fdesc, ExpTab{exptab},\n                                                             nullptr, nullptr);\n    STAMP(9);\n")
rep("    store_part_sum(term, red, sm.part2[g & 1] + (size_t)i * pb.n_parts + part);\n}\n",
    "    store_part_sum(term, red, sm.part2[g & 1] + (size_t)i * pb.n_parts + part);\n    STAMP(10);\n}\n")
s += ('\nextern "C" int lcf_debug_read_stamps(unsigned long long* out) {\n    return (int)hipMemcpyFromSymbol(out, '
      'HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 12);\n}\n')
os.makedirs(os.path.join(ROOT, 'build_variants'), exist_ok=True)
tmp = os.path.join(ROOT, 'lightcurve_fitting_amd/csrc/_stamps_tmp.hip')
open(tmp, 'w').write(s)
try:
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                    '-I' + os.path.join(ROOT, 'include'), '-Wno-unused-value', '-ffp-contract=on', '-shared', '-o',
                    os.path.join(ROOT, 'build_variants/liblcf_stamps.so'), tmp,
                    os.path.join(ROOT, 'lightcurve_fitting_amd/csrc/lcf_sed.hip')], check=True)
finally:
    os.remove(tmp)
print('built build_variants/liblcf_stamps.so')
