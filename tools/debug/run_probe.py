"""k_run probe: which run lengths / splits work (chain equal to k_solo's)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from lightcurve_fitting_amd.engine import NativeSampler, LcfError
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
for nw, steps, split in ((40, 1, 'identity'), (40, 8, 'identity'), (40, 9, 'identity'), (40, 8, 'random'), (40, 9, 'random'),
                         (40, 20, 'random'), (1024, 20, 'random'), (41, 20, 'random')):
    x0 = bench.initial_walkers(nw)
    out = {}
    for kernel in ('run', 'solo'):
        s = NativeSampler(eng, nw, 5)
        used = s.set_half_step_kernel(kernel)
        s.set_state(x0)
        try:
            s.run(0, steps, split, True)
            out[kernel] = s.get_chain()[0]
        except LcfError as exc:
            out[kernel] = str(exc)[:150]
        s.close()
    same = isinstance(out['run'], np.ndarray) and np.array_equal(out['run'], out['solo'])
    print(nw, steps, split, used, 'same' if same else out['run'] if isinstance(out['run'], str) else 'DIFFERENT', flush=True)
