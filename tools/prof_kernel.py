#!/usr/bin/env python3
"""Run only the dominant kernel -- k_fused, one launch per half-step of a 1024-walker ensemble at the configs[1]
shape -- for a few steps: target for the rocprofv3 --pmc passes and for A/B timing of kernel variants.

    python tools/prof_kernel.py [steps=10] [variant=2]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    model, lc, priors = bench.build_problem(0)
    eng = model.engine_for(lc, priors=priors)
    eng.set_variant(variant)
    x0 = bench.initial_walkers(bench.WALKERS_PER_GPU)
    ms = bench.fused_kernel_ms(eng, x0, reps)
    n = bench.WALKERS_PER_GPU // 2
    print(json.dumps({'kernel_ms': ms, 'proposals_per_launch': n, 'variant': variant,
                      'alg_frac': n * bench.ALG_INSTR / (ms * 1e-3) / 1e12 / bench.PEAK_FP64_TINSTR}))


if __name__ == '__main__':
    main()
