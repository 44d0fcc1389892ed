"""186 000 points in 31 000 epochs (8 parts of ~3900 epochs: 62 KiB of thermal states per workgroup): the one-launch
half-step with more than 64 KiB of LDS against the two-kernel path."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from helpers import lc_dict
from lightcurve_fitting_amd import models as M
from lightcurve_fitting_amd.engine import NativeSampler
from oracle import lcf_oracle as O
rng = np.random.default_rng(99)
epochs = np.sort(rng.uniform(0.4, 30., 31000))
t = np.repeat(epochs, 6); names = list(np.tile(list('UBVgri'), len(epochs)))
bands = [O.band(n) for n in names]
truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
ytrue = O.evaluate(('ShockCooling', O.ShockCoolingOracle(0.)), t, bands, truth)
y, dy = ytrue * (1 + 0.05 * rng.standard_normal(len(t))), 0.05 * ytrue
priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
eng = M.ShockCooling(redshift=0.).engine_for(lc_dict(t, names, y, dy), priors=priors)
for nw in (16, 256, 1024):
    x0 = truth * (1 + 0.002 * rng.standard_normal((nw, 5)))
    for kernel in ('auto', 'fused', 'phases'):
        s = NativeSampler(eng, nw, 5)
        used = s.set_half_step_kernel(kernel)
        s.set_state(x0); s.run(0, 3, 'random', False); s.run(3, 10, 'random', False)
        print(nw, 'walkers', kernel, '->', used, ' ms per step', s.last_run_ms() / 10)
