# A/B: device time per half-step of configs[1] with a library variant
for v in "$@"; do
  LCF_HIP_LIB=build_variants/liblcf_$v.so python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import bench, numpy as np
from lightcurve_fitting_amd.engine import NativeSampler
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
s = NativeSampler(eng, 1024, 7)
s.set_state(bench.initial_walkers(1024))
s.run(0, 200, 'random', False)
best = 1e9
for rep in range(5):
    s.run(200 + 1000 * rep, 1000, 'random', False)
    best = min(best, s.last_run_ms() / 2000)
x, lp = s.get_state()
print('$v', 'us per half-step %.3f' % (1e3 * best), 'kernel', s.last_run_kernel(), 'mean lp %.6f' % lp.mean())
PY
done
