#!/usr/bin/env python3
"""Run only the dominant kernel (per-point likelihood, config-2 shape, 512 walkers) a few times: target for rocprofv3
--pmc passes and for A/B timing of kernel variants."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    model, lc, priors = bench.build_problem(0)
    eng = model.engine_for(lc, priors=priors)
    eng.set_variant(variant)
    x0 = bench.initial_walkers(n)
    ms = eng.profile_loglike_kernel(x0, reps=reps)
    samples = n * eng.samples_per_eval
    print(json.dumps({'kernel_ms': ms, 'walkers': n, 'variant': variant, 'real_samples_per_s': samples / ms * 1e3,
                      'alg_frac': n * bench.ALG_INSTR / (ms * 1e-3) / 1e12 / bench.PEAK_FP64_TINSTR}))

if __name__ == '__main__':
    main()
