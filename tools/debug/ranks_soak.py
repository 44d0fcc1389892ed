"""Soak of the sharded protocol (ad hoc): R emulated ranks of one ensemble on one GPU, row protocol, random light
curves / walker counts / rank counts incl. uneven and empty shards; every rank must reproduce the one-GPU chain."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import torch
from lightcurve_fitting_amd import models as M
from lightcurve_fitting_amd.engine import NativeSampler
from lightcurve_fitting_amd.sampler import NativeBackend, shard_bounds

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.time()
side = torch.cuda.Stream()
for seed in range(lo, hi):
    rng = np.random.default_rng(9000 + seed)
    n_ep = int(rng.integers(3, 400))
    filts = list(rng.choice(['U', 'B', 'V', 'g', 'r', 'i'], int(rng.integers(1, 7)), replace=False))
    epochs = np.sort(rng.uniform(0.3, 25., n_ep))
    t = np.repeat(epochs, len(filts)); names = list(np.tile(filts, n_ep))
    m = M.ShockCooling(redshift=0.)
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    y = m(t, names, *truth) * (1 + 0.05 * rng.standard_normal(len(t))); dy = 0.05 * y
    pri = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.29)]
    eng = m.engine_for({'MJD': t, 'filter': names, 'lum': y, 'dlum': dy}, priors=pri)
    nw = 2 * int(rng.integers(5, 80)); nh = nw // 2
    ranks = int(rng.choice([2, 3, 5, 8]))
    proto_rows = bool(rng.integers(2))
    x0 = truth * (1 + 0.03 * rng.standard_normal((nw, 5)))
    ref = NativeSampler(eng, nw, seed); ref.set_state(x0); ref.run(0, 3, 'random', True)
    ss = [NativeSampler(eng, nw, seed) for _ in range(ranks)]
    bs = [NativeBackend(s, rows=proto_rows) for s in ss]
    for s in ss:
        s.set_state(x0); s.begin(0, 3, 'random', True)
    bounds = [shard_bounds(nh, ranks, r)[:2] for r in range(ranks)]
    with torch.cuda.stream(side):
        for step in range(3):
            for half in (0, 1):
                for r, b in enumerate(bs):
                    b.half_step(step, half, *bounds[r])
                views = [b.newlp() for b in bs]
                for r, (a0, a1) in enumerate(bounds):
                    for q in range(ranks):
                        if q != r and a1 > a0:
                            views[q][a0:a1].copy_(views[r][a0:a1])
                for b in bs:
                    b.accept(step, half)
    side.synchronize()
    want = ref.get_chain()
    for r, s in enumerate(ss):
        s.check()
        got = s.get_chain()
        if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
            bad.append((seed, ranks, nw, len(t), proto_rows, r))
            break
    for s in ss + [ref]:
        s.close()
    if seed % 20 == 0:
        print('seed', seed, 'failures', len(bad), f'{time.time() - t0:.0f}s', flush=True)
print('done', hi - lo, 'cases; failures:', bad)
