#!/usr/bin/env python3
"""Run only the dominant kernel of a bench workload for a few steps: the target of the rocprofv3 --pmc passes of
tools/collect_profiles.sh (and of A/B timings of kernel variants).

    python tools/prof_kernel.py [mcmc|companion|population|sed] [steps=10]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else 'mcmc'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    import torch  # noqa: F401
    from lightcurve_fitting_amd.engine import NativeSampler
    if workload == 'mcmc':
        model, lc, priors = bench.build_problem(0)
        eng = model.engine_for(lc, priors=priors)
        s = NativeSampler(eng, bench.WALKERS_PER_GPU, bench.SEED + 7)
        s.set_state(bench.initial_walkers(bench.WALKERS_PER_GPU))
        s.run(0, steps, 'random', False)
        # (`steps` and `roofline.half_steps_per_launch`: what tools/collect_profiles.sh reads to average the right launches
        # and to say how many half-steps a profiled launch covered)
        print(json.dumps({'workload': workload, 'kernel': s.set_half_step_kernel('auto'), 'launches': s.last_run_launches(),
                          'kernel_ms': s.last_run_ms() / s.last_run_launches(), 'steps': steps,
                          'roofline': {'half_steps_per_launch': 2. * steps / s.last_run_launches()}}))
    elif workload == 'companion':
        model, lc, priors, _ = bench.build_companion(0)
        eng = model.engine_for(lc, priors=priors)
        s = NativeSampler(eng, bench.COMPANION_WALKERS, bench.SEED + 7)
        s.set_state(bench.companion_walkers(bench.COMPANION_WALKERS))
        s.run(0, steps, 'random', False)
        print(json.dumps({'workload': workload, 'kernel': s.set_half_step_kernel('auto'), 'launches': s.last_run_launches(),
                          'kernel_ms': s.last_run_ms() / s.last_run_launches()}))
    elif workload == 'population':
        args = bench.parse_args(['--workload', 'population', '--steps', str(steps), '--warmup', '2', '--no-cpu-baseline'])
        bench.run_population(args)
    elif workload == 'sed':
        args = bench.parse_args(['--workload', 'sed', '--steps', str(20 * steps), '--no-cpu-baseline'])
        bench.run_sed(args)
    else:
        raise SystemExit('unknown workload ' + workload)


if __name__ == '__main__':
    main()
