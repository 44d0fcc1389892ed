"""walker-steps/s of the headline fit (bench.py's problem) under each half-step kernel choice, same box, same process.
  python tools/debug/kernel_choice_timing.py [steps] [walkers]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from lightcurve_fitting_amd.engine import NativeSampler  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
x0 = bench.initial_walkers(nw)
chains = {}
for kernel in ('auto', 'fused', 'phases', 'auto', 'fused'):
    s = NativeSampler(eng, nw, 11)
    used = s.set_half_step_kernel(kernel)
    s.set_state(x0)
    s.run(0, 50, 'random', False)
    t0 = time.perf_counter()
    s.run(50, steps, 'random', False)
    wall = time.perf_counter() - t0
    ms = s.last_run_ms()
    x, lp = s.get_state()
    chains[used] = (x, lp)
    print(f'{kernel:6s} -> {used:6s}: {nw * steps / wall:.3e} walker-steps/s wall, device {1e3 * ms / (2 * steps):.2f} us per half-step',
          flush=True)
    s.close()
names = list(chains)
for k in names[1:]:
    print(f'{names[0]} == {k}:', np.array_equal(chains[names[0]][0], chains[k][0]) and np.array_equal(chains[names[0]][1], chains[k][1]))
for a in names:
    for b in names:
        if a < b:
            print(f'{a} == {b}:', np.array_equal(chains[a][0], chains[b][0]) and np.array_equal(chains[a][1], chains[b][1]))

# the row-board form of k_solo with ONE rank: what reading rows from tagged words in uncached memory and posting them costs
s = NativeSampler(eng, nw, 11)
ptr = s.board_export()[1]
s.board_connect(1, 0, local_ptrs=[ptr])
s.set_state(x0)
s.run_rows(0, 50, 'random', False)
t0 = time.perf_counter()
s.run_rows(50, steps, 'random', False)
wall = time.perf_counter() - t0
x, lp = s.get_state()
print(f'rows   -> k_solo over its own board: {nw * steps / wall:.3e} walker-steps/s wall, device {1e3 * s.last_run_ms() / (2 * steps):.2f} us '
      f'per half-step; same state as solo: {np.array_equal(x, chains["solo"][0])}')
