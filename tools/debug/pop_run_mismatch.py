"""Where a population's chains from resident launches (k_pop_run) first differ from those of a launch per half-step
(k_pop): per transient the first (step, walker) that differs.   python tools/debug/pop_run_mismatch.py [transients walkers steps epochs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lightcurve_fitting_amd import models as M  # noqa: E402
from lightcurve_fitting_amd.sampler import PopulationSampler  # noqa: E402

n_tr, nw, steps, n_ep = (int(a) for a in (sys.argv[1:5] + ['4', '64', '6', '60'][len(sys.argv) - 1:]))
problems, x0 = [], {}
for k in range(n_tr):
    rng = np.random.default_rng(300 + k)
    epochs = np.sort(rng.uniform(0.4, 9., n_ep))
    t, names = np.repeat(epochs, 6), list(np.tile(list('UBVgri'), n_ep))
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1]) * rng.uniform(0.9, 1.1, 5)
    m = M.ShockCooling(redshift=0.)
    y = m(t, names, *truth) * (1 + 0.05 * rng.standard_normal(len(t)))
    problems.append((m, {'MJD': t, 'filter': names, 'lum': y, 'dlum': 0.05 * np.abs(y)},
                     [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]))
    x0[k] = truth * (1 + 0.05 * rng.standard_normal((nw, 5)))
out = {}
for form in ('population', 'population-run'):
    if form == 'population':
        os.environ['LCF_NO_POP_RUN'] = '1'
    else:
        os.environ.pop('LCF_NO_POP_RUN', None)
    pop = PopulationSampler(problems, nw, seed=17)
    pop.run_mcmc(x0, steps)
    first = [(pop[k].get_chain(), pop[k].get_log_prob()) for k in range(n_tr)]
    pop.run_mcmc(None, 3)
    print(form, '->', pop[0]._native.last_run_kernel(), pop[0]._native.last_run_launches(), 'launches', flush=True)
    out[form] = [(np.concatenate([first[k][0], pop[k].get_chain()]), np.concatenate([first[k][1], pop[k].get_log_prob()]),
                  pop[k].acceptance_fraction) for k in range(n_tr)]
for k in range(n_tr):
    a, b = out['population'][k], out['population-run'][k]
    bad = np.argwhere(np.any(a[0] != b[0], axis=2) | (a[1] != b[1]))
    print('transient', k, 'same' if len(bad) == 0 else f'{len(bad)} (step, walker) differ, first {bad[0]}: '
          f'{a[0][tuple(bad[0])]} lp {a[1][tuple(bad[0])]} | {b[0][tuple(bad[0])]} lp {b[1][tuple(bad[0])]}',
          '| acceptance same:', np.array_equal(a[2], b[2]))
