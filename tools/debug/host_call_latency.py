"""Per-call latency of the host-pointer API (what an emcee `vectorize=True` callable pays per half-step)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
for n in (64, 512, 2048):
    P = bench.initial_walkers(n)
    for _ in range(20):
        eng.log_posterior(P)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.log_posterior(P)
    dt = (time.perf_counter() - t0) / 200
    print(f'{n} walkers: {dt * 1e6:.1f} us per lcf_log_posterior call -> {n / dt:.3g} walker-evals/s')
