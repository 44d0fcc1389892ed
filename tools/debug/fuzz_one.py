import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import test_gpu_fuzz as T
from conftest import relerr
from helpers import lc_dict
from lightcurve_fitting_amd import models as M
from oracle import lcf_oracle as O
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
n = int(rng.integers(3, 400))
pool = list(rng.choice(T.ALL, int(rng.integers(1, 9)), replace=False))
t, names, y, dy = T._light_curve(rng, pool, n)
z = float(rng.choice([0., 0.003, 0.05, 0.7]))
bands = [O.band(x) for x in names]
lc = lc_dict(t, names, y, dy)
kw = dict(n=float(rng.choice([1.5, 3.])), RW=bool(rng.integers(2)))
variant = int(rng.integers(3))
print('n', n, 'pool', pool, 'z', z, kw, 'variant', variant)
cases = [
    (M.ShockCooling(redshift=z, **kw), ('ShockCooling', O.ShockCoolingOracle(z, **kw)),
     T._params(rng, [0.1, 0.05, 0.2, 0.1, -5.], [5., 3., 10., 8., 10.], 12, 0.08)),
    (M.ShockCooling2(redshift=z, **kw), ('ShockCooling2', O.ShockCoolingOracle(z, **kw)),
     T._params(rng, [1., 0.1, 1., -5.], [80., 20., 60., 10.], 12, 0.08)),
    (M.ShockCooling4(redshift=z), ('ShockCooling4', O.ShockCooling4Oracle(z)),
     T._params(rng, [0.1, 0.05, 0.2, 0.1, -5.], [5., 3., 10., 8., 10.], 12, 0.08)),
]
for model, orc, P in cases:
    eng = model.engine_for(lc)
    eng.set_variant(variant)
    want_y = O.evaluate(orc, t, bands, P.T).T
    got = eng.evaluate(P)
    bad = np.isnan(got) != np.isnan(want_y)
    print(type(model).__name__, 'nan mismatch count', bad.sum())
    if bad.any():
        w, i = np.argwhere(bad)[0]
        print('  walker', w, 'P', P[w], 'point', i, 't', t[i], names[i], 'got', got[w, i], 'want', want_y[w, i])
        print('  T,R oracle', [x[i] if np.ndim(x) else x for x in (orc[1].temperature_radius(t, *P[w]) if orc[0] != 'ShockCooling2' else orc[1].temperature_radius2(t, *P[w]))])
        T_, R_ = eng.temperature_radius(P[w:w + 1])
        print('  T,R engine', T_[0, i], R_[0, i])
    want = O.log_likelihood(orc, t, bands, y, dy, P.T)
    gl = model.log_likelihood(lc, P)
    bad = np.isnan(gl) != np.isnan(want)
    print('  loglike nan mismatch', bad.sum(), np.argwhere(bad).ravel()[:5])
    sig = np.column_stack([P, rng.uniform(0., 3., len(P))])
    mode = str(rng.choice(['relative', 'absolute']))
    model.engine_for(lc, True, mode).set_variant(variant)
    want = O.log_likelihood(orc, t, bands, y, dy, sig.T, True, mode)
    gl = model.log_likelihood(lc, sig, True, mode)
    bad = np.isnan(gl) != np.isnan(want)
    print('  sigma loglike nan mismatch', bad.sum(), mode, [(sig[k], gl[k], want[k]) for k in np.argwhere(bad).ravel()[:2]])
m3, o3 = M.ShockCooling3(redshift=z, **kw), ('ShockCooling3', O.ShockCoolingOracle(z, **kw))
P = T._params(rng, [0.1, 0.05, 0.2, 0.1, 1., 0., -5.], [5., 3., 10., 8., 100., 1.5, 10.], 12, 0.05)
lc3 = {'MJD': t, 'filter': names, 'flux': y * 1e-47, 'dflux': dy * 1e-47}
eng = m3.engine_for(lc3)
eng.set_variant(min(variant, 1))
got = eng.evaluate(P)
want_y = O.evaluate(o3, t, bands, P.T).T
bad = np.isnan(got) != np.isnan(want_y)
print('SC3 evaluate nan mismatch', bad.sum())
for w in np.unique(np.argwhere(bad)[:, 0])[:3]:
    i = np.argwhere(bad[w]).ravel()[0]
    print('  walker', w, 'P', P[w], 'point', i, 't', t[i], names[i], 'got', got[w, i], 'want', want_y[w, i])
want = O.log_likelihood(o3, t, bands, lc3['flux'], lc3['dflux'], P.T)
gl = m3.log_likelihood(lc3, P)
bad = np.isnan(gl) != np.isnan(want)
print('SC3 loglike nan mismatch', bad.sum(), [(P[k], gl[k], want[k]) for k in np.argwhere(bad).ravel()[:3]])
