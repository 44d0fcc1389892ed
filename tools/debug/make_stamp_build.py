#!/usr/bin/env python3
"""Diagnostic build of the library with s_memtime stamps inside k_step (never shipped).
Writes build_variants/liblcf_stamps.so; tools/debug/read_stamps.py prints the median cycles per segment."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, 'lightcurve_fitting_amd/csrc/lcf_hip.hip')).read()
s = src
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b, 1)
rep("// ND > 0: the walker dimension is a compile-time constant", "__device__ unsigned long long g_stamps_decl_marker;\n// ND > 0: the walker dimension is a compile-time constant")
rep("template <int ND>\n__device__ inline void step_body(", "__device__ unsigned long long g_stamps[64 * 12];\n#define STAMP(k) do { if (stamp_on) { unsigned long long t_; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_) :: \"memory\"); if (threadIdx.x == 0) g_stamps[i * 12 + (k)] = t_; } } while (0)\ntemplate <int ND>\n__device__ inline void step_body(")
rep("    const int i = bid / n_echunks, ec = bid % n_echunks;\n    const bool in_shard = do_thermal && i >= lo && i < hi;", "    const int i = bid / n_echunks, ec = bid % n_echunks;\n    const bool stamp_on = ec == 0 && i < 64 && have_prev && have_next;\n    STAMP(0);\n    const bool in_shard = do_thermal && i >= lo && i < hi;")
rep("        if (have_next) dr = draws[i];\n", "        if (have_next) dr = draws[i];\n        STAMP(1);\n")
rep("        if (lane == 0 && rslot >= 0) {  // commit of the previous half-step's slot i", "        STAMP(2);\n        if (lane == 0 && rslot >= 0) {  // commit of the previous half-step's slot i")
rep("        if (have_next) {\n            double q[kMaxDim], lq[kMaxDim];", "        STAMP(3);\n        if (have_next) {\n            double q[kMaxDim], lq[kMaxDim];")
rep("            const double lg = log(arg);  // one logarithm per lane, all at once", "            STAMP(4);\n            const double lg = log(arg);  // one logarithm per lane, all at once")
rep("            double c[kNCoef];\n            walker_coefficients(pb, q, lq, c);", "            STAMP(5);\n            double c[kNCoef];\n            walker_coefficients(pb, q, lq, c);\n            STAMP(6);")
rep("            if (lane == 0) {\n                for (int k = 0; k < kNCoef; ++k) sc[k] = c[k];", "            STAMP(7);\n            if (lane == 0) {\n                for (int k = 0; k < kNCoef; ++k) sc[k] = c[k];")
rep("    if (!have_next || !in_shard) return;\n    __syncthreads();", "    STAMP(8);\n    if (!have_next || !in_shard) return;\n    __syncthreads();\n    STAMP(9);")
rep("    therm[(size_t)i * pb.n_epochs + ep] = make_double2(T > 0. ? 1. / T : 0., pref);\n}\n\ntemplate <int ND>\n__global__ __launch_bounds__(kBlock) void k_step(", "    therm[(size_t)i * pb.n_epochs + ep] = make_double2(T > 0. ? 1. / T : 0., pref);\n    STAMP(10);\n}\n\ntemplate <int ND>\n__global__ __launch_bounds__(kBlock) void k_step(")
s += '\nextern "C" int lcf_debug_read_stamps(unsigned long long* out) {\n    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 12);\n}\n'
os.makedirs(os.path.join(ROOT, 'build_variants'), exist_ok=True)
tmp = os.path.join(ROOT, 'lightcurve_fitting_amd/csrc/_stamps_tmp.hip')
open(tmp, 'w').write(s)
try:
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                    '-Wno-unused-value', '-ffp-contract=on', '-shared', '-o', os.path.join(ROOT, 'build_variants/liblcf_stamps.so'),
                    tmp, os.path.join(ROOT, 'lightcurve_fitting_amd/csrc/lcf_sed.hip')], check=True)
finally:
    os.remove(tmp)
