"""One single-GPU run of the multiband test problem: k_solo_run ('auto') against k_solo, for a given seed of the start
state and walker count.   python tools/debug/run_repro.py [x0 seed] [walkers] [steps] [prior scenarios]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_half_step_kernels as T  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
nwalkers = int(sys.argv[2]) if len(sys.argv) > 2 else 54
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 9
pb, eng = T._multiband()
x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(seed).standard_normal((nwalkers, 5)))
for rep in range(3):
    refs = {k: T._run(eng, nwalkers, 321, x0, nsteps, k, 'random') for k in ('auto', 'solo')}
    print(rep, 'kernel', refs['auto'][4].last_run_kernel(), 'same chain', np.array_equal(refs['auto'][1], refs['solo'][1]),
          'same counts', np.array_equal(refs['auto'][3], refs['solo'][3]), flush=True)
