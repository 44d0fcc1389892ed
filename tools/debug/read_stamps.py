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
         'barrier', 'coefficients to SGPRs', 'columns (states + points)', 'wave sums + barrier', 'accept + commit']
d = np.diff(a[:, :11], axis=1)
for n, v in zip(names, np.median(d, axis=0)):
    print(f'{n:45s} {v:9.0f}')
tot = a[:, 10] - a[:, 0]
print('total: median', np.median(tot), ' min', tot.min(), ' p90', np.percentile(tot, 90), ' max', tot.max())
print('per phase min:', ' '.join(f'{v:.0f}' for v in d.min(axis=0)))
print('per phase p10:', ' '.join(f'{v:.0f}' for v in np.percentile(d, 10, axis=0)))
print('per phase mean:', ' '.join(f'{v:.0f}' for v in d.mean(axis=0)))
print('per phase p90:', ' '.join(f'{v:.0f}' for v in np.percentile(d, 90, axis=0)))
print('per phase max:', ' '.join(f'{v:.0f}' for v in d.max(axis=0)))
print('wave 1: staging done at', np.median(a[:, 11] - a[:, 0]), ' columns done at', np.median(a[:, 12] - a[:, 0]),
      '(wave 0:', np.median(a[:, 8] - a[:, 0]), ')')
print('columns, wave 0: thermal state after', np.median(a[:, 13] - a[:, 7]), ' first three filters after',
      np.median(a[:, 14] - a[:, 7]), ' next three after', np.median(a[:, 15] - a[:, 7]), ' all', np.median(a[:, 8] - a[:, 7]))
print('device ms per step', s.last_run_ms / nst)

# chip-wide timeline of the last two launches (100 MHz wall clock, the same on every XCD)
wb = (C.c_ulonglong * (2 * 1024 * 2))()
lib.lcf_debug_read_wall.argtypes = [C.c_void_p]
lib.lcf_debug_read_wall(wb)
w = np.array(wb[:], dtype=np.int64).reshape(2, 1024, 2)[:, :nw // 2, :]
order = np.argsort([w[0, :, 0].min(), w[1, :, 0].min()])
first, second = w[order[0]], w[order[1]]
for name, x in (('earlier launch', first), ('later launch', second)):
    e, q = x[:, 0], x[:, 1]
    print(f'{name}: entries spread over {(e.max() - e.min()) / 100:.2f} us, exits over {(q.max() - q.min()) / 100:.2f} us, '
          f'first entry -> last exit {(q.max() - e.min()) / 100:.2f} us, median lifetime {np.median(q - e) / 100:.2f} us')
print(f'last exit of the earlier launch -> first entry of the later one: {(second[:, 0].min() - first[:, 1].max()) / 100:.2f} us; '
      f'first entry -> first entry: {(second[:, 0].min() - first[:, 0].min()) / 100:.2f} us')
