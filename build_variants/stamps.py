import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from lightcurve_fitting_amd import engine as E
from lightcurve_fitting_amd.sampler import EnsembleSampler
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
s = EnsembleSampler(1024, 5, eng, seed=1)
s.run_mcmc(bench.initial_walkers(1024), 50, store=False)
lib = E.load_library()
buf = (C.c_ulonglong * (64*12))()
lib.lcf_debug_read_stamps.argtypes=[C.c_void_p]; lib.lcf_debug_read_stamps(buf)
a = np.array(buf[:], dtype=np.int64).reshape(64,12)
d = np.diff(a[:, :11], axis=1)
print('median cycles per segment (s_memtime ticks):', np.median(d, axis=0))
print('total', np.median(a[:,10]-a[:,0]), 'to stamp8', np.median(a[:,8]-a[:,0]))
