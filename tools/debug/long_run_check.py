"""Resident launches of up to 256 half-steps against a launch per half-step over LONG runs -- several blocks of draw
records, the ring of 512 versions wrapped several times, a run continued twice, ensembles of several slots per workgroup:
chain, log-probabilities and acceptance counts bit for bit.    python tools/debug/long_run_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from lightcurve_fitting_amd.engine import NativeSampler  # noqa: E402

model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
bad = 0
for nw, runs in ((1024, (700, 300)), (100, (1500, 40, 900)), (2048, (400,)), (37, (2000,))):
    x0 = bench.initial_walkers(nw)
    out = {}
    for kern in ('solo', 'auto'):
        s = NativeSampler(eng, nw, 11)
        s.set_half_step_kernel(kern)
        s.set_state(x0)
        first, parts, kernels = 0, [], []
        for n in runs:
            s.run(first, n, 'random', True)
            parts.append(s.get_chain())
            kernels.append((s.last_run_kernel(), s.last_run_launches()))
            first += n
        out[kern] = (parts, s.naccepted(), s.get_state(), kernels)
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(out['solo'][0], out['auto'][0]))
    same = same and np.array_equal(out['solo'][1], out['auto'][1])
    same = same and all(np.array_equal(a, b) for a, b in zip(out['solo'][2], out['auto'][2]))
    bad += 0 if same else 1
    print(nw, 'walkers, runs of', runs, 'steps:', out['auto'][3], 'against', out['solo'][3][0][0], '->', 'same' if same else 'DIFFERENT',
          flush=True)
print('done; differing cases:', bad)
