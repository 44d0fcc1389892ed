"""Which entries of the out-of-domain ShockCooling golden case differ in NaN pattern at a band-sum level."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from conftest import golden
from helpers import shockcooling_case
from lightcurve_fitting_amd import models as M
s, lc = shockcooling_case()
m = M.ShockCooling(redshift=0.01)
for v in (3, 2):
    e = m.engine_for(lc); e.set_variant(v)
    got = e.evaluate(s['sce/P']); want = s['sce/sc/y']
    na, nb = np.isnan(got), np.isnan(want)
    bad = np.argwhere(na != nb)
    print('variant', v, 'mismatches', len(bad))
    for w in sorted(set(bad[:, 0])):
        cols = bad[bad[:, 0] == w][:, 1]
        print(' walker', w, 'P', s['sce/P'][w], 'n', len(cols), 'got', got[w, cols[:3]], 'want', want[w, cols[:3]], 't', np.asarray(lc['MJD'])[cols[:3]])
