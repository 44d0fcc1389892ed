"""Median s_memtime ticks between the stamps of tools/debug/make_stamp_build.py (run with
LCF_HIP_LIB=build_variants/liblcf_stamps.so on the GPU box)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from lightcurve_fitting_amd import engine as E  # noqa: E402
from lightcurve_fitting_amd.sampler import EnsembleSampler  # noqa: E402

model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
nw = int(os.environ.get('LCF_STAMP_WALKERS', '1024'))
s = EnsembleSampler(nw, 5, eng, seed=1)
nst = int(os.environ.get('LCF_STAMP_STEPS', '50'))
s.run_mcmc(bench.initial_walkers(nw), nst, store=False)
lib = E.load_library()
buf = (C.c_ulonglong * (64 * 16))()
lib.lcf_debug_read_stamps.argtypes = [C.c_void_p]
lib.lcf_debug_read_stamps(buf)
a = np.array(buf[:], dtype=np.int64).reshape(64, 16)
names = ['entry->draw record', 'walker rows + proposal', 'logarithms', 'coefficients', 'priors + LDS publish',
         'barrier', 'coefficients to SGPRs + thermal states', 'points', 'wave sums + barrier', 'accept + commit']
d = np.diff(a[:, :11], axis=1)
for n, v in zip(names, np.median(d, axis=0)):
    print(f'{n:45s} {v:9.0f}')
print('total', np.median(a[:, 10] - a[:, 0]), ' spread of entry stamps over the 64 workgroups', np.ptp(a[:, 0]))
print('wave 1: staging done at', np.median(a[:, 11] - a[:, 0]), ' points done at', np.median(a[:, 12] - a[:, 0]),
      '(wave 0:', np.median(a[:, 8] - a[:, 0]), ')')
print('points loop, wave 0: operands of the first iteration after', np.median(a[:, 13] - a[:, 7]), ' first iteration done after',
      np.median(a[:, 14] - a[:, 7]), ' all iterations', np.median(a[:, 8] - a[:, 7]))
print('device ms per step', s.last_run_ms / nst)
