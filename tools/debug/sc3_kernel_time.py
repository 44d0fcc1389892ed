"""Likelihood-kernel time of ShockCooling3 (per-walker reddening, full tables) next to ShockCooling at the
BASELINE configs[1] shape: 512 walkers x 3000 points.  python tools/debug/sc3_kernel_time.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from lightcurve_fitting_amd import models as M  # noqa: E402

model, lc, priors = bench.build_problem(0)
x0 = bench.initial_walkers(512)
eng = model.engine_for(lc, priors=priors)
for variant in (2, 1):
    eng.set_variant(variant)
    print(f'ShockCooling  variant {variant}: {eng.profile_loglike_kernel(x0, 50) * 1e3:.1f} us')
m3 = M.ShockCooling3(redshift=0., n=1.5)
x3 = np.column_stack([x0[:, :4], np.full(512, 20.), np.random.default_rng(0).uniform(0., 0.5, 512), x0[:, 4]])
lc3 = {'MJD': lc['MJD'], 'filter': lc['filter'], 'flux': lc['lum'] * M.c4 / 400., 'dflux': lc['dlum'] * M.c4 / 400.}
e3 = m3.engine_for(lc3)
print(f'ShockCooling3 (full tables + reddening): {e3.profile_loglike_kernel(x3, 50) * 1e3:.1f} us')
