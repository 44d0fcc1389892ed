import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd import engine as E, models as M
from lightcurve_fitting_amd.filters import PackedTables
model, lc, priors = bench.build_problem(0)
x0 = bench.initial_walkers(512)
eng = model.engine_for(lc, priors=priors)
print('default', eng.profile_loglike_kernel(x0, 50) * 1e3)
t, filts, y, dy = M._photometry(lc, 'lum')
uniq, idx = M._index_filters(filts)
tabs = PackedTables(uniq, z=0.)
print(tabs.hoff, tabs.htmin, tabs.coff, tabs.ctmin)
for label, htab in (('no hot', None), ('hot', (tabs.hoff, tabs.ha, tabs.hw, tabs.htmin))):
    e = E.Engine(model.model_id, 5, model._consts(), t, y, dy, idx, tabs.off, tabs.a, tabs.w,
                 ctab=(tabs.coff, tabs.ca, tabs.cw, tabs.ctmin), htab=htab)
    e.set_variant(2)
    print(label, e.profile_loglike_kernel(x0, 50) * 1e3, e.log_likelihood(x0[:2]))
