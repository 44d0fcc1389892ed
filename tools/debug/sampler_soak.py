"""Soak of the sampler paths (ad hoc, not part of the suite): random light curves, models and ensemble sizes; the
one-launch run must equal the separate phases (k_step + k_thermal + k_points + k_finalize) bit for bit, and the
row-protocol half-steps too."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from lightcurve_fitting_amd import models as M
from lightcurve_fitting_amd.engine import NativeSampler, LcfError

lo, hi = int(sys.argv[1]), int(sys.argv[2])
POOL = ['U', 'B', 'V', 'g', 'r', 'i', 'DLT40', 'z', 'UVW2', 'R', 'I']
bad = []
kernels = {}
NS = int(os.environ.get('SOAK_STEPS', '3'))
t0 = time.time()
for seed in range(lo, hi):
    rng = np.random.default_rng(5000 + seed)
    n_ep = int(rng.integers(2, 700))
    filts = list(rng.choice(POOL, int(rng.integers(1, 7)), replace=False))
    epochs = np.sort(rng.uniform(0.3, 25., n_ep))
    if rng.random() < 0.6:   # multiband grid
        t = np.repeat(epochs, len(filts)); names = list(np.tile(filts, n_ep))
    else:                    # ragged
        t = epochs; names = list(rng.choice(filts, n_ep))
    kind = int(rng.integers(4))
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    if kind == 0:
        m = M.ShockCooling(redshift=0.01, n=float(rng.choice([1.5, 3.])), RW=bool(rng.integers(2)))
        pri = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.29)]
    elif kind == 1:
        m = M.ShockCooling4(redshift=0.01)
        pri = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.29)]
    elif kind == 2:
        m = M.ShockCooling2(redshift=0.01); truth = np.array([30., 3., 20., 0.1])
        pri = [M.UniformPrior(0., 100.)] * 3 + [M.UniformPrior(-1., 0.29)]
    else:
        m = M.ShockCooling3(redshift=0.01); truth = np.array([1.2, 0.5, 3.0, 2.0, 30., 0.1, 0.1])
        pri = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(1., 100.), M.UniformPrior(0., 1.), M.UniformPrior(-1., 0.29)]
    ndim = len(truth)
    use_sigma = bool(rng.integers(2))
    if use_sigma:
        truth = np.append(truth, 0.5); pri = pri + [M.UniformPrior(0., 5.)]; ndim += 1
    q = 'flux' if kind == 3 else 'lum'
    ytrue = m(t, names, *truth[:m.nparams])
    y = ytrue * (1 + 0.05 * rng.standard_normal(len(t))); dy = 0.05 * np.abs(ytrue) + 1e-300
    lc = {'MJD': t, 'filter': names, q: y, 'd' + q: dy}
    nw = 2 * int(rng.integers(ndim, 60))
    try:
        eng = m.engine_for(lc, use_sigma, 'relative', pri)
        x0 = truth * (1 + 0.03 * rng.standard_normal((nw, ndim)))
        a = NativeSampler(eng, nw, seed); a.set_state(x0); a.run(0, NS, 'random', True)
        kernels[a.last_run_kernel()] = kernels.get(a.last_run_kernel(), 0) + 1
        b = NativeSampler(eng, nw, seed); b.set_state(x0); b.begin(0, NS, 'random', True)
        c = NativeSampler(eng, nw, seed); c.set_state(x0); c.begin(0, NS, 'random', True)
        for step in range(NS):
            for half in (0, 1):
                b.propose(step, half); b.evaluate(0, nw // 2); b.accept(step, half)
                c.half_step_rows(step, half, 0, nw // 2); c.accept(step, half)
        b.check(); c.check()
        ca, cb, cc = a.get_chain(), b.get_chain(), c.get_chain()
        ok = np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1]) and np.array_equal(ca[0], cc[0]) \
            and np.array_equal(ca[1], cc[1]) and np.all(np.isfinite(ca[1]))
        if not ok:
            bad.append((seed, kind, len(t), nw, a.one_launch, 'chains differ'))
        for s in (a, b, c):
            s.close()
    except Exception as exc:  # noqa: BLE001
        bad.append((seed, kind, len(t), nw, repr(exc)[:160]))
    if seed % 20 == 0:
        print('seed', seed, 'failures', len(bad), f'{time.time() - t0:.0f}s', flush=True)
print('done', hi - lo, 'cases; kernels of the one-call run:', kernels, '; failures:', bad)
