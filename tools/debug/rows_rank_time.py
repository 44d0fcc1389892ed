"""What ONE RANK of a row-board run costs per half-step, measured on one GPU: a sampler connected as the only rank of a
row-board run (its own uncached board, system-scope posts and polls, progress words, the row collection behind every
launch) with as many proposals per half-step as a rank of an N-GPU run has -- against the single-GPU run of the same
ensemble (k_solo_run, the board in ordinary memory, chain written by the workgroups).  What it cannot show: the posts to
the other N - 1 boards (one more store instruction per board in the commit) and the fabric's share of the post -> poll
latency.

    python tools/debug/rows_rank_time.py [workload=mcmc|companion] [walkers ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lightcurve_fitting_amd.engine import NativeSampler  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'mcmc'
sizes = [int(a) for a in sys.argv[2:]] or [1024, 512]
steps = int(os.environ.get('STEPS', '640'))
if workload == 'companion':
    model, lc, priors, _ = bench.build_companion(0)
    walkers = bench.companion_walkers
else:
    model, lc, priors = bench.build_problem(0)
    walkers = bench.initial_walkers
eng = model.engine_for(lc, priors=priors)
for nw in sizes:
    x0 = walkers(nw)
    out = {}
    for form in ('single GPU', 'one rank, resident', 'one rank, launch per half-step'):
        s = NativeSampler(eng, nw, 7)
        if form != 'single GPU':
            s.board_connect(1, 0, local_ptrs=[s.board_export()[1]])
        s.set_half_step_kernel('solo' if form.endswith('half-step') else 'auto')
        s.set_state(x0)
        run = s.run if form == 'single GPU' else s.run_rows
        run(0, 64, 'random', True)
        best = 1e9
        for rep in range(4):
            run(64 + steps * rep, steps, 'random', True)
            best = min(best, s.last_run_ms() / (2 * steps))
        out[form] = (1e3 * best, s.last_run_kernel(), s.last_run_launches(), s.get_state()[1].mean())
        s.close()
    print(workload, nw, 'walkers (%d proposals per half-step):' % (nw // 2),
          '; '.join(f'{k}: {v[0]:.2f} us per half-step ({v[1]}, {v[2]} launches)' for k, v in out.items()),
          '| same state:', len({v[3] for v in out.values()}) == 1, flush=True)
