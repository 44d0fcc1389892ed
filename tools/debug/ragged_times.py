"""configs[1] with every observation at its own time (no shared epochs), as real multi-band photometry is: which kernel
runs and how fast."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd import models as M
from lightcurve_fitting_amd.engine import NativeSampler
for jitter in (0., 1e-3):
    rng = np.random.default_rng(bench.SEED)
    epochs = np.sort(rng.uniform(0.5, 10., bench.N_EPOCHS))
    t = np.repeat(epochs, len(bench.BANDS)) + jitter * rng.uniform(0., 1., bench.N_EPOCHS * len(bench.BANDS))
    names = list(np.tile(bench.BANDS, bench.N_EPOCHS))
    model = M.ShockCooling(redshift=0., n=1.5)
    ytrue = model(t, names, *bench.TRUTH)
    y = ytrue * (1. + 0.05 * rng.standard_normal(len(t)))
    lc = {'MJD': t, 'filter': names, 'lum': y, 'dlum': 0.05 * ytrue}
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    eng = model.engine_for(lc, priors=priors)
    s = NativeSampler(eng, 1024, 5)
    used = s.set_half_step_kernel('auto')
    s.set_state(bench.initial_walkers(1024))
    s.run(0, 50, 'random', False)
    s.run(50, 500, 'random', False)
    print(f'jitter {jitter}: kernel {used}, {1e3 * s.last_run_ms() / 1000:.2f} us per half-step, '
          f'{1024 * 500 / (s.last_run_ms() * 1e-3):.3e} walker-steps/s', flush=True)
