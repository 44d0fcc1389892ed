"""Where the wall time of a SHORT run goes (the driver times 20 steps): host time of each native call of one
run_mcmc(None, n) at the benchmark shape, against the device time of the run.  Run on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
import torch  # noqa: E402
from lightcurve_fitting_amd.sampler import EnsembleSampler  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
s = EnsembleSampler(1024, 5, eng, seed=1)
s.run_mcmc(bench.initial_walkers(1024), 50, store=False)
ns = s._native
rows = []
for rep in range(8):
    torch.cuda.synchronize()
    time.sleep(0.002 * (rep % 2))   # every other repetition starts from an idle (down-clocked) device
    t0 = time.perf_counter()
    ns.run_async(s._steps_done, n, 'random', False)
    t1 = time.perf_counter()
    ns.wait()
    t2 = time.perf_counter()
    acc = ns.naccepted()
    x, lp = ns.get_state()
    t3 = time.perf_counter()
    s._steps_done += n
    rows.append((1e6 * (t1 - t0), 1e6 * (t2 - t1), 1e6 * (t3 - t2), 1e3 * ns.last_run_ms()))
    t4 = time.perf_counter()
    s.run_mcmc(None, n, store=False)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    rows[-1] += (1e6 * (t5 - t4), 1e3 * s.last_run_ms)
print(f'{n} steps: [enqueue us, wait us, fetch us, device us | run_mcmc wall us, its device us]')
for r in rows:
    print('  '.join(f'{v:9.1f}' for v in r))
