"""Mean life of a workgroup of the half-step kernel, phase by phase, over ALL half-steps of a run (the sums of the
-DLCF_STAMPS build: tools/debug/make_stamp_build.py; run with LCF_HIP_LIB=build_variants/liblcf_stamps.so on the GPU box).
    python tools/debug/read_stamp_sums.py [walkers=1024] [steps=640]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from lightcurve_fitting_amd import engine as E  # noqa: E402
from lightcurve_fitting_amd.sampler import EnsembleSampler  # noqa: E402

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 640
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
s = EnsembleSampler(nw, 5, eng, seed=1)
s.run_mcmc(bench.initial_walkers(nw), 64, store=False)
lib = E.load_library()
acc, cnt = (C.c_ulonglong * 1024)(), (C.c_ulonglong * 1024)()
lib.lcf_debug_read_stamp_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.lcf_debug_read_stamp_sums(acc, cnt, 1)
s.run_mcmc(None, nst, store=True)
lib.lcf_debug_read_stamp_sums(acc, cnt, 0)
a = np.array(acc[:], dtype=np.float64).reshape(64, 16)
c = np.array(cnt[:], dtype=np.float64).reshape(64, 16)
names = ['(loop) -> entry', 'entry -> draw record', 'rows polled + proposal', 'logarithms', 'coefficients',
         'priors + LDS publish', 'barrier', 'entry of the column phase', 'columns', 'wave sums + barrier', 'accept + commit']
per = np.where(c > 0, a / np.maximum(c, 1), np.nan)          # ticks per passage
share = a / (2. * nst)                                        # ticks per half-step (a stamp not passed adds nothing)
print(f'{nw} walkers, {nst} steps, kernel {s._native.last_run_kernel()}; device {1e3 * s.last_run_ms / (2 * nst):.3f} us per half-step')
print(f'{"phase":32s} {"ticks/half-step":>16s} {"ticks/passage":>14s} {"passed":>8s}   (mean over 64 workgroups; 2.4 GHz ticks)')
for k, n in enumerate(names):
    print(f'{n:32s} {np.nanmean(share[:, k]):16.0f} {np.nanmean(per[:, k]):14.0f} {np.mean(c[:, k]) / (2 * nst):8.2f}')
tot = share[:, :11].sum(axis=1)
print(f'sum {tot.mean():.0f} ticks = {tot.mean() / 2400:.2f} us per half-step (min {tot.min():.0f}, max {tot.max():.0f})')
