import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd.sampler import EnsembleSampler
nw = 64
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
x0 = bench.initial_walkers(nw)
for n in (1, 2, 3, 4, 8, 33, 70):
    out = {}
    for kern in ('solo', 'auto'):
        s = EnsembleSampler(nw, 5, eng, seed=7)
        s._native.set_half_step_kernel(kern)
        s.run_mcmc(x0, n, store=True)
        out[kern] = s._native.naccepted().copy()
        out[kern + 'x'] = [a.copy() for a in s._native.get_state()]
    same_state = all(np.array_equal(a, b) for a, b in zip(out['solox'], out['autox']))
    d = out['solo'] - out['auto']
    print(n, 'state equal', same_state, 'solo sum', out['solo'].sum(), 'run sum', out['auto'].sum(), 'walkers that differ', np.flatnonzero(d)[:12], d[np.flatnonzero(d)][:12])
