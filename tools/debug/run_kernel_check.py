"""k_solo_run (resident workgroups, one launch per block of half-steps) against k_solo (a launch per half-step):
the same chain bit for bit, and the time per half-step of both."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd.sampler import EnsembleSampler
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
if os.environ.get('WORKLOAD') == 'companion':
    model, lc, priors, _ = bench.build_companion(0)
    x0, nd = bench.companion_walkers(nw), 8
else:
    model, lc, priors = bench.build_problem(0)
    x0, nd = bench.initial_walkers(nw), 5
eng = model.engine_for(lc, priors=priors)
chains = {}
KERNELS = os.environ.get('KERNELS', 'solo,auto').split(',')
for kern in KERNELS:
    s = EnsembleSampler(nw, nd, eng, seed=7)
    used = s._native.set_half_step_kernel(kern)
    s.reserve_chain(steps)
    s.run_mcmc(x0, min(70, steps), store=True)
    chains[kern] = (s.get_chain().copy(), s.get_log_prob().copy(), s._native.naccepted().copy())
    for n in (20, steps):
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.run_mcmc(None, n, store=True)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print('%-5s -> %-5s %5d steps: wall %8.1f us, %6.2f us per half-step (device %6.2f), %.3e walker-steps/s' %
              (kern, s._native.last_run_kernel(), n, 1e6 * best, 1e6 * best / (2 * n), 1e3 * s.last_run_ms / (2 * n), nw * n / best), flush=True)
a, b = chains[KERNELS[0]], chains[KERNELS[-1]]
print('nacc sums', a[2].sum(), b[2].sum(), a[2][:8], b[2][:8], 'accepted moves in the chain', (np.diff(a[0][:, :, 0], axis=0) != 0).sum() + 0)
print('chain equal:', np.array_equal(a[0], b[0]), 'lp equal:', np.array_equal(a[1], b[1]), 'nacc equal:', np.array_equal(a[2], b[2]),
      'max |dx|', np.abs(a[0] - b[0]).max())
