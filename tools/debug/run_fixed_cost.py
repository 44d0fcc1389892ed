"""Fixed cost of a run: wall time of run_mcmc(None, k) for small k (device part = 2 k launches of 12.5 us)."""
import os, sys, time, gc
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd.sampler import EnsembleSampler
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
x0 = bench.initial_walkers(1024)
gc.collect(); gc.disable()
bench.half_step_kernel_ms(eng, 1024, x0, 3)
s = EnsembleSampler(1024, 5, eng, seed=1)
s.run_mcmc(x0, 5, store=False)
for k in (1, 1, 2, 5, 10, 20, 20):
    w, n, d = [], [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run_mcmc(None, k, store=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        w.append(1e6 * (t2 - t0)); n.append(1e6 * (t1 - t0)); d.append(1e3 * s.last_run_ms)
    print(f'{k:3d} steps: wall {np.median(w):6.0f} us, run_mcmc {np.median(n):6.0f}, device {np.median(d):6.0f} -> fixed {np.median(w) - np.median(d):5.0f} us')
