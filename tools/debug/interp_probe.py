import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from helpers import lc_dict
from lightcurve_fitting_amd import models as M
from oracle import lcf_oracle as O
rng = np.random.default_rng(11)
nw = 64
epochs = np.geomspace(0.3, 60., 200)
t, names = np.repeat(epochs, 6), list(np.tile(list('UBVgri'), len(epochs)))
m = M.ShockCooling2(redshift=0.01)
P = np.column_stack([np.geomspace(0.3, 3000., nw), np.full(nw, 3.), np.full(nw, 20.), rng.uniform(-0.2, 0.5, nw)])
P[3, 0], P[4, 1] = -5., -1.
ytrue = 1e20 * (1 + rng.uniform(0., 1., len(t)))
eng = m.engine_for(lc_dict(t, names, ytrue, 0.05 * ytrue))
out = {}
for v in (3, 2, 0):
    eng.set_variant(v); out[v] = eng.evaluate(P)
orc = O.ShockCoolingOracle(0.01)
for v in (3, 2):
    a, b = out[v], out[0]
    bad = np.argwhere(np.abs(a - b) > 1e-11 * np.maximum(np.abs(b), 1e-300))
    print('variant', v, 'mismatches', len(bad))
    for w, i in bad[:12]:
        T, R = orc.temperature_radius2(t[i:i+1], *P[w])
        print('  walker', w, 'point', i, names[i], 't', t[i], 'T', T, 'got', a[w, i], 'want', b[w, i])
