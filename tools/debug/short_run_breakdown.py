"""Where the wall time of a 20-step run goes: Python around the native calls, the native calls, the device."""
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
spent = {}
def wrap(name):
    orig = getattr(nat, name)
    def timed(*a, **k):
        t = time.perf_counter(); r = orig(*a, **k); spent[name] = spent.get(name, 0.) + time.perf_counter() - t; return r
    setattr(nat, name, timed)
for name in ('run', 'snapshot'):
    wrap(name)
for i in range(8):
    spent.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.run_mcmc(None, 20, store=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('wall %.0f us | run_mcmc %.0f | native run %.0f, snapshot %.0f | device %.0f' %
          (1e6 * (t2 - t0), 1e6 * (t1 - t0), 1e6 * spent['run'], 1e6 * spent.get('snapshot', 0.), 1e3 * s.last_run_ms))
