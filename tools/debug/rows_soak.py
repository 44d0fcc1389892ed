"""Soak of the row-board runs between ranks: 2-3 emulated ranks of one ensemble on one GPU, random light curves (models,
filters, shared and own epochs, fitted sigma), walker counts, run lengths (one to several resident launches; with small
blocks of draw records: launches of a few half-steps and many progress waits), both forms -- resident workgroups
(k_solo_run<..., RANKS>) and a launch per half-step -- and two runs that continue each other.  Every rank must end with
the single-GPU chain, state and acceptance counts, bit for bit.        python tools/debug/rows_soak.py lo hi"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lightcurve_fitting_amd import models as M  # noqa: E402
from lightcurve_fitting_amd.engine import NativeSampler  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, skipped, forms = [], [], {'run': 0, 'solo': 0}
os.environ.setdefault('LCF_PEER_WAIT_S', '1.0')
t0 = time.time()
for seed in range(lo, hi):
    rng = np.random.default_rng(77000 + seed)
    n_ep = int(rng.integers(3, 260))
    filts = list(rng.choice(['U', 'B', 'V', 'g', 'r', 'i'], int(rng.integers(1, 7)), replace=False))
    epochs = np.sort(rng.uniform(0.3, 25., n_ep))
    t, names = np.repeat(epochs, len(filts)), list(np.tile(filts, n_ep))
    if rng.integers(4) == 0:                        # every observation at a time of its own
        t = t + rng.uniform(0., 0.2, len(t))
    two = rng.integers(3) == 0
    sigma = rng.integers(4) == 0
    truth = np.array([30., 3., 30., 0.2]) if two else np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    make = (lambda: M.ShockCooling2(redshift=0.01)) if two else (lambda: M.ShockCooling(redshift=0.01))
    m0 = make()
    y = m0(t, names, *truth) * (1 + 0.05 * rng.standard_normal(len(t)))
    lc = {'MJD': t, 'filter': names, 'lum': y, 'dlum': 0.05 * np.abs(y)}
    pri = ([M.UniformPrior(0., 100.)] * 3 + [M.UniformPrior(-1., 0.29)]) if two else \
        ([M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.29)])
    if sigma:
        pri = pri + [M.UniformPrior(0., 5.)]
    ranks = int(rng.choice([2, 3]))
    nh = ranks * int(rng.integers(3, 30))
    nw = 2 * nh
    nd = len(pri)
    x0 = np.concatenate([truth, [0.5]] if sigma else [truth]) * (1 + 0.03 * rng.standard_normal((nw, nd)))
    n1, n2 = int(rng.integers(1, 50)), int(rng.integers(1, 50))
    block = str(int(rng.choice([3, 7, 40])))
    if rng.integers(2) and not os.environ.get('SOAK_NOBLOCK'):
        os.environ['LCF_DRAW_BLOCK'] = block
    else:
        os.environ.pop('LCF_DRAW_BLOCK', None)
    split = 'random' if rng.integers(4) else 'identity'
    form = 'auto' if rng.integers(3) else 'solo'
    kw = dict(use_sigma=bool(sigma), priors=pri)
    ref = NativeSampler(m0.engine_for(lc, **kw), nw, seed)
    ref.set_half_step_kernel(os.environ.get('SOAK_REF', 'solo'))
    ref.set_state(x0)
    ref.run(0, n1 + n2, split, True)
    want = ref.get_chain()
    # (Emulated ranks are streams of ONE process, which owns four hardware queues: two ranks whose streams land on the same
    # queue run in order, and the rank in front then waits for the one behind it until the bound -- seen for the very
    # first case of a process (SOAK_PAD=1 moves the collisions to every third case).  Such a case ends with the bounded
    # wait's error and is counted as SKIPPED, not as a mismatch; real ranks are processes with queues of their own.)
    pad = [make().engine_for(lc, **kw) for _ in range(int(os.environ.get('SOAK_PAD', '0')))]
    engines = [make().engine_for(lc, **kw) for _ in range(ranks)]
    ss = [NativeSampler(e, nw, seed) for e in engines]
    ok = True
    try:
        ptrs = [s.board_export()[1] for s in ss]
        for r, s in enumerate(ss):
            s.board_connect(ranks, r, local_ptrs=ptrs)
            s.set_state(x0)
            s.run(500, n1 + n2, split, True)      # (size every buffer first: emulated ranks share the host thread)
            s.set_state(x0)
            s.set_half_step_kernel(form)
        chains = [[], []]
        for first, n in ((0, n1), (n1, n2)):
            for s in ss:
                s.run_rows(first, n, split, True, asynchronous=True)
            for s in ss:
                s.wait()
            forms[ss[0].last_run_kernel()] += 1
            got = [s.get_chain() for s in ss]
            for g in got:
                ok = ok and np.array_equal(g[0], want[0][first:first + n]) and np.array_equal(g[1], want[1][first:first + n])
        for s in ss:
            ok = ok and np.array_equal(s.naccepted(), ref.naccepted())
            ok = ok and all(np.array_equal(a, b) for a, b in zip(s.get_state(), ref.get_state()))
    except Exception as exc:  # noqa: BLE001
        if 'was not posted within' in str(exc) or 'waited' in str(exc):
            skipped.append(seed)
            ok = True
        else:
            ok = False
            print('seed', seed, type(exc).__name__, str(exc)[:200], flush=True)
    if not ok:
        bad.append((seed, ranks, nw, len(t), 'SC2' if two else 'SC', bool(sigma), n1, n2, os.environ.get('LCF_DRAW_BLOCK'), split, form))
    for s in ss + [ref]:
        s.close()
    if seed % 20 == 0:
        print('seed', seed, 'failures', len(bad), forms, f'{time.time() - t0:.0f}s', flush=True)
print('done', hi - lo, 'cases;', forms, '; skipped (ranks on one hardware queue):', skipped, '; MISMATCHES:', bad)
