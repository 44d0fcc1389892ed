"""One 600-step run at the benchmark shape, for `rocprofv3 --kernel-trace` (block transitions of the draw records)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from lightcurve_fitting_amd.sampler import EnsembleSampler  # noqa: E402

model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
s = EnsembleSampler(1024, 5, eng, seed=1)
s.run_mcmc(bench.initial_walkers(1024), 20, store=False)
s.run_mcmc(None, int(sys.argv[1]) if len(sys.argv) > 1 else 600, store=False)
print('device ms per step', s.last_run_ms / 600)
