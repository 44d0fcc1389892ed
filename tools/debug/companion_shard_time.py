"""configs[2] as specified is STRONG scaling: 4096 walkers over N GPUs, i.e. 2048 / N proposals per rank and half-step.
The row-board driver runs k_solo over a rank's own slots only, so one rank's launch at N GPUs is timed here, on one GPU,
as the half-step of an ensemble of 4096 / N walkers (same kernel, same light curve; the board's polls and posts add the
0.9 us measured for them in round 2).  Prints us per half-step launch and the strong-scaling factor it implies."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd.engine import NativeSampler

model, lc, priors, _ = bench.build_companion(0)
eng = model.engine_for(lc, priors=priors)
base = None
for n_gpus in (1, 2, 4, 8):
    nw = bench.COMPANION_WALKERS // n_gpus
    s = NativeSampler(eng, nw, 3)
    s.set_state(bench.companion_walkers(nw))
    s.run(0, 20, 'random', False)
    best = 1e9
    for rep in range(3):
        s.run(20 + 60 * rep, 60, 'random', False)
        best = min(best, s.last_run_ms() / 120)
    if base is None:
        base = best
    print(f'{n_gpus} GPUs: {nw // 2:5d} proposals per rank and half-step ({s.last_run_kernel()}): {1e3 * best:8.2f} us per launch '
          f'-> strong-scaling factor {base / best:5.2f} (+0.9 us of board traffic: {base / (best + 0.9e-3):5.2f})', flush=True)
    s.close()
