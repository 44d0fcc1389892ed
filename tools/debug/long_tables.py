"""Fit through filters with long transmission tables (2MASS, UVOT, TESS, DECam: 4000+ samples in all -> only the
compressed levels are staged in LDS): one-launch path must be available and agree with the oracle."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from conftest import relerr
from lightcurve_fitting_amd import models as M
from lightcurve_fitting_amd.engine import NativeSampler
from oracle import lcf_oracle as O
rng = np.random.default_rng(3)
filts = ['UVW2', 'UVW1', 'U', 'B', 'V', 'g', 'r', 'i', 'J', 'H', 'K', 'TESS', 'z-DECam']
epochs = np.sort(rng.uniform(0.5, 12., 230))
t = np.repeat(epochs, len(filts)); names = list(np.tile(filts, len(epochs)))
m = M.ShockCooling(redshift=0.01)
truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
orc = ('ShockCooling', O.ShockCoolingOracle(0.01))
bands = [O.band(n) for n in names]
ytrue = O.evaluate(orc, t, bands, truth)
y = ytrue * (1 + 0.05 * rng.standard_normal(len(t))); dy = 0.05 * ytrue
pri = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.4)]
lc = {'MJD': t, 'filter': names, 'lum': y, 'dlum': dy}
eng = m.engine_for(lc, priors=pri)
P = truth * (1 + 0.1 * rng.standard_normal((64, 5)))
P[0, 3] = 0.01  # a cold walker: temperatures below the compressed levels' validity for some bands
for v in (2, 1, 0):
    eng.set_variant(v)
    print('variant', v, 'rel err', relerr(eng.log_likelihood(P), O.log_likelihood(orc, t, bands, y, dy, P.T)))
eng.set_variant(2)
s = NativeSampler(eng, 1024, 1)
print('one launch:', s.one_launch, 'points', len(t))
x0 = truth * (1 + 0.05 * rng.standard_normal((1024, 5)))
s.set_state(x0); s.run(0, 20, 'random', False); s.run(20, 100, 'random', False)
print('us per step', s.last_run_ms() / 100 * 1e3)
for kernel in ('fused', 'phases'):
    s2 = NativeSampler(eng, 1024, 1)
    print(kernel, '->', s2.set_half_step_kernel(kernel))
    s2.set_state(x0); s2.run(0, 20, 'random', False); s2.run(20, 100, 'random', False)
    print('   us per step', s2.last_run_ms() / 100 * 1e3)
