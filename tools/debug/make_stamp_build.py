#!/usr/bin/env python3
"""Diagnostic build of the library with s_memtime stamps inside k_solo (-DLCF_STAMPS; never shipped; see DESIGN.md
section 5).  Writes build_variants/liblcf_stamps.so; tools/debug/read_stamps.py prints the median ticks per segment.
Stamps of the first 64 workgroups, by the first lane of wave 0:
  0 kernel entry | 1 draw record loaded | 2 walker rows loaded, proposal formed | 3 logarithms | 4 coefficients |
  5 priors, serial part written to LDS | 6 after the barrier (tables staged) | 7 thermal states in LDS |
  8 points done | 9 wave sums in LDS | 10 accept test + commit done
and of wave 1:  11 its share of the table staging done | 12 its points done"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.makedirs(os.path.join(ROOT, 'build_variants'), exist_ok=True)
csrc = os.path.join(ROOT, 'lightcurve_fitting_amd', 'csrc')
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-DLCF_STAMPS', '-DLCF_DEV_BUILD',
                '-I' + os.path.join(ROOT, 'include'), '-Wno-unused-value', '-ffp-contract=on', '-mllvm', '-disable-machine-licm', '-shared', '-o',
                os.path.join(ROOT, 'build_variants/liblcf_stamps.so'), os.path.join(csrc, 'lcf_hip.hip'),
                os.path.join(csrc, 'lcf_sed.hip')], check=True)
print('built build_variants/liblcf_stamps.so')
