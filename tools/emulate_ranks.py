"""Per-rank kernel cost of a multi-GPU fit, measured on ONE device.

`ranks` samplers play the ranks of one sharded ensemble (BASELINE configs[1] shape, 1024 walkers per rank) on one
GPU, exactly as tests/test_gpu_sampler.py::test_emulated_multi_rank_run_on_one_gpu does: same seed, every rank
evaluates only its shard through lcf_sampler_half_step_rows, the all-gather is device-to-device copies.  Everything a
rank launches per half-step except RCCL itself is therefore timed; run it under
`rocprofv3 --kernel-trace --stats --output-format csv` for the per-kernel split.

    python tools/emulate_ranks.py [ranks=8] [steps=50]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import torch
    import bench
    from lightcurve_fitting_amd.engine import NativeSampler
    from lightcurve_fitting_amd.sampler import NativeBackend, shard_bounds
    model, lc, priors = bench.build_problem(0)
    eng = model.engine_for(lc, priors=priors)
    nw = 1024 * ranks
    nh = nw // 2
    x0 = bench.initial_walkers(nw)
    samplers = [NativeSampler(eng, nw, 1234) for _ in range(ranks)]
    backs = [NativeBackend(s, rows=True) for s in samplers]  # rows of partial sums travel, as in the native run
    bounds = [shard_bounds(nh, ranks, r)[:2] for r in range(ranks)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st = side.cuda_stream
        for first, n in ((0, 5), (5, steps)):  # warm-up, then the timed run
            for s in samplers:
                if first == 0:
                    s.set_state(x0)
                s.begin(first, n, 'random', False)
            side.synchronize()
            t0 = time.perf_counter()
            for step in range(first, first + n):
                for half in (0, 1):
                    for r, b in enumerate(backs):
                        b.half_step(step, half, *bounds[r])
                    views = [b.newlp() for b in backs]
                    for r, (lo, hi) in enumerate(bounds):
                        for q in range(ranks):
                            if q != r:
                                views[q][lo:hi].copy_(views[r][lo:hi], non_blocking=True)
                    for s in samplers:
                        s.accept(step, half, st)
            side.synchronize()
            dt = time.perf_counter() - t0
    for s in samplers:
        s.check()
    a = samplers[0].get_state()[0]
    assert all(np.array_equal(a, s.get_state()[0]) for s in samplers[1:]), 'emulated ranks diverged'
    print(f'{ranks} emulated ranks x 1024 walkers: {dt / (2 * steps) * 1e6:.1f} us per half-step for all ranks '
          f'(copies included), {dt / (2 * steps * ranks) * 1e6:.1f} us per rank')


if __name__ == '__main__':
    main()
