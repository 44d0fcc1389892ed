"""Soak of the population runs with resident workgroups (k_pop_run) against a launch per half-step (k_pop): random
populations (2-9 transients; models, filters, epochs, fitted sigma, odd and even walker counts), run lengths from one to
several launches (small blocks of draw records: launches of a few half-steps), few workgroups per transient (several
groups of proposals per workgroup), fewer workgroups than transients (several launches per block of half-steps), the
interpolants from LDS or from L2, and two runs that continue each other.  Chains, log-probabilities and acceptance
counts must agree bit for bit.            python tools/debug/pop_soak.py lo hi"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lightcurve_fitting_amd import models as M  # noqa: E402
from lightcurve_fitting_amd.sampler import PopulationSampler  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, kernels = [], {}
KNOBS = ('LCF_NO_POP_RUN', 'LCF_DRAW_BLOCK', 'LCF_RUN_GRID', 'LCF_POP_ITAB_LDS')
t0 = time.time()
for seed in range(lo, hi):
    rng = np.random.default_rng(91000 + seed)
    n_tr = int(rng.integers(2, 10))
    two = rng.integers(3) == 0
    sigma = rng.integers(4) == 0
    filts = list(rng.choice(['U', 'B', 'V', 'g', 'r', 'i'], int(rng.integers(2, 7)), replace=False))
    nw = int(rng.integers(12, 120))
    truth = np.array([30., 3., 30., 0.2]) if two else np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    pri = ([M.UniformPrior(0., 100.)] * 3 + [M.UniformPrior(-1., 0.29)]) if two else \
        ([M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.29)])
    if sigma:
        pri = pri + [M.UniformPrior(0., 5.)]
    problems, x0 = [], {}
    for k in range(n_tr):
        n_ep = int(rng.integers(3, 200))
        epochs = np.sort(rng.uniform(0.3, 25., n_ep))
        t, names = np.repeat(epochs, len(filts)), list(np.tile(filts, n_ep))
        m = M.ShockCooling2(redshift=0.01) if two else M.ShockCooling(redshift=0.01)
        y = m(t, names, *truth) * (1 + 0.05 * rng.standard_normal(len(t)))
        problems.append((m, {'MJD': t, 'filter': names, 'lum': y, 'dlum': 0.05 * np.abs(y)}, pri) +
                        (({'use_sigma': True},) if sigma else ()))
        x0[k] = np.concatenate([truth, [0.5]] if sigma else [truth]) * (1 + 0.03 * rng.standard_normal((nw, len(pri))))
    n1, n2 = int(rng.integers(1, 80)), int(rng.integers(1, 40))
    env = {}
    if rng.integers(2):
        env['LCF_DRAW_BLOCK'] = str(int(rng.choice([2, 5, 24])))
    if rng.integers(2):
        env['LCF_RUN_GRID'] = str(int(rng.choice([1, 3, 7, 20])))
    if rng.integers(3) == 0:
        env['LCF_POP_ITAB_LDS'] = '0'
    out = {}
    for form in ('population', 'population-run'):
        for name in KNOBS:
            os.environ.pop(name, None)
        os.environ.update(env)
        if form == 'population':
            os.environ['LCF_NO_POP_RUN'] = '1'
        try:
            pop = PopulationSampler(problems, nw, seed=seed)
            pop.run_mcmc(x0, n1)
            first = [(pop[k].get_chain(), pop[k].get_log_prob()) for k in range(n_tr)]
            pop.run_mcmc(None, n2, store=bool(seed % 3))
            used = pop[0]._native.last_run_kernel()
            kernels[used] = kernels.get(used, 0) + 1
            out[form] = [first[k] + ((pop[k].get_chain(), pop[k].get_log_prob()) if seed % 3 else ()) +
                         (pop[k].acceptance_fraction,) + tuple(pop[k]._native.get_state()[:2]) for k in range(n_tr)]
        except Exception as exc:  # noqa: BLE001
            print('seed', seed, form, type(exc).__name__, str(exc)[:300], flush=True)
            out[form] = None
    same = out['population'] is not None and out['population-run'] is not None and all(
        all(np.array_equal(u, v) for u, v in zip(a, b)) for a, b in zip(out['population'], out['population-run']))
    if not same:
        bad.append((seed, n_tr, nw, 'SC2' if two else 'SC', bool(sigma), len(filts), n1, n2, env))
    if seed % 10 == 0:
        print('seed', seed, 'failures', len(bad), kernels, f'{time.time() - t0:.0f}s', flush=True)
print('done', hi - lo, 'cases;', kernels, '; MISMATCHES:', bad)
