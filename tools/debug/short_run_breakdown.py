"""Where the wall time of a 20-step run goes: Python around the native call, the native call, the device."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd.sampler import EnsembleSampler
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
x0 = bench.initial_walkers(1024)
bench.half_step_kernel_ms(eng, 1024, x0, 3)
s = EnsembleSampler(1024, 5, eng, seed=1)
s.run_mcmc(x0, 5, store=False)
nat = s._native
orig = nat.run
spent = {}
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); spent['native'] = time.perf_counter() - t; return r
nat.run = timed
rows = []
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.run_mcmc(None, 20, store=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((1e6 * (t2 - t0), 1e6 * (t1 - t0), 1e6 * spent['native'], 1e3 * s.last_run_ms))
for r in rows:
    print('wall %.0f us | run_mcmc %.0f | native call %.0f | device (events around the 40 launches) %.0f' % r)
