"""What the 8-GPU scaling of the two sharded workloads comes to ON PAPER, from one GPU: per GPU count N, the time per
half-step of ONE RANK of the row-board run -- a sampler connected as the only rank of such a run (its own uncached
board, system-scope posts and polls, progress words, the row collection behind every launch) with as many proposals per
half-step as a rank of an N-GPU run has -- against the single-GPU run of the whole ensemble.

  configs[2] (companion, default)   STRONG scaling: 4096 walkers over N GPUs, 2048 / N proposals per rank and half-step;
                                    factor = t(1 GPU, 2048 proposals) / t(one rank, 2048 / N proposals)
  configs[1] (`mcmc`)               WEAK scaling: 1024 walkers per GPU, 512 proposals per rank at any N;
                                    factor = N x t(1 GPU, 512 proposals) / t(one rank, 512 proposals)

The row collection behind a launch handles the rows of ALL ranks (every rank keeps the whole chain): one rank alone
collects only its own share, so the share's cost is measured (stored against unstored run) and added N - 1 more times.
What one GPU cannot show: the posts to the other N - 1 boards (one more 16-byte store per board and row, not waited
for) and the fabric's share of the post -> poll latency.          python tools/debug/companion_shard_time.py [mcmc]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from lightcurve_fitting_amd.engine import NativeSampler  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'companion'
if workload == 'companion':
    model, lc, priors, _ = bench.build_companion(0)
    walkers, total, steps = bench.companion_walkers, bench.COMPANION_WALKERS, 64
else:
    model, lc, priors = bench.build_problem(0)
    walkers, total, steps = bench.initial_walkers, bench.WALKERS_PER_GPU, 640
eng = model.engine_for(lc, priors=priors)


def per_half_step(nw, rank, store=True):
    s = NativeSampler(eng, nw, 3)
    if rank:
        s.board_connect(1, 0, local_ptrs=[s.board_export()[1]])
    s.set_state(walkers(nw))
    run = s.run_rows if rank else s.run
    run(0, 32, 'random', store)
    best = 1e9
    for rep in range(3):
        run(32 + steps * rep, steps, 'random', store)
        best = min(best, s.last_run_ms() / (2 * steps))
    kernel, launches = s.last_run_kernel(), s.last_run_launches()
    s.close()
    return 1e3 * best, kernel, launches


t1, k1, _ = per_half_step(total, False)
print(f'{workload}: 1 GPU, {total // 2} proposals per half-step ({k1}): {t1:.2f} us per half-step', flush=True)
for n in (2, 4, 8):
    nw = total // n if workload == 'companion' else total
    t, k, launches = per_half_step(nw, True)
    t_unstored, _, _ = per_half_step(nw, True, store=False)
    own = max(t - t_unstored, 0.)                   # collecting the rank's OWN rows into the chain, per half-step
    t_all = t + (n - 1) * own                       # ... and the other ranks' rows
    factor = t1 / t_all if workload == 'companion' else n * t1 / t_all
    print(f'{n} GPUs: one rank with {nw // 2:5d} proposals per half-step ({k}, {launches} launches of the last run): {t:6.2f} us '
          f'per half-step with its own rows collected ({own:.3f} us of it), {t_all:6.2f} with all ranks\' rows -> '
          f'{"strong" if workload == "companion" else "weak"}-scaling factor {factor:5.2f}', flush=True)
