"""Walker-steps/s of the single-GPU fit against the ensemble size (configs[1] light curve)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd.engine import NativeSampler
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
for nw in (64, 256, 1024, 2048, 4096, 8192, 16384):
    s = NativeSampler(eng, nw, 3)
    s.set_state(bench.initial_walkers(nw))
    s.run(0, 20, 'random', False)
    s.run(20, 200, 'random', False)
    ms = s.last_run_ms() / 200
    print(f'{nw:6d} walkers: {ms * 1e3:8.1f} us per step  {nw / ms * 1e3 / 1e6:7.2f}e6 walker-steps/s')
    s.close()
