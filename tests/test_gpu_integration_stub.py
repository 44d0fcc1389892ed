"""INTEGRATION.md's reference-side binding runs as written: the stub (tools/integration_stub.py, quoted verbatim in
INTEGRATION.md) is executed against stand-ins that expose exactly what it reads from the reference's objects -- an `lc`
with `.data` columns, `Filter.trans` tables, a model with the construction-time constants, priors -- and its
log-posteriors are the oracle's."""
import os
import sys

import numpy as np
import pytest

from conftest import relerr
from helpers import small_problem
from oracle import lcf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_quotes_the_stub_verbatim():
    stub = open(os.path.join(ROOT, 'tools', 'integration_stub.py')).read()
    assert stub in open(os.path.join(ROOT, 'INTEGRATION.md')).read()


@pytest.mark.gpu
def test_reference_side_binding_runs_as_written():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import integration_stub as S
    from lightcurve_fitting_amd.engine import LIB_PATH

    class Col:                       # astropy Column: .data
        def __init__(self, a):
            self.data = np.asarray(a)

    class RefFilter:                 # filters.py: Filter.trans['freq'], ['T_norm_per_freq'] (descending frequency)
        def __init__(self, band):
            self.name = band.name
            self.trans = {'freq': band.freq, 'T_norm_per_freq': band.tnorm}

    class UniformPrior:              # models.py:1066-1075
        def __init__(self, p_min, p_max):
            self.p_min, self.p_max = p_min, p_max

    class RefModel:                  # BaseShockCooling.__init__ (models.py:192-226), n = 1.5
        output_quantity, z = 'lum', 0.
        A, a, alpha, epsilon_1, epsilon_2, L_0, T_0, Tph_to_Tcol = 0.94, 1.67, 0.8, 0.027, 0.086, 2.0e42, 1.61, 1.1

    pb = small_problem()
    filt = {n: RefFilter(O.band(n)) for n in dict.fromkeys(pb['names'])}

    class LC(dict):
        pass
    lc = LC(MJD=Col(pb['t']), lum=Col(pb['y']), dlum=Col(pb['dy']))
    lc['filter'] = [filt[n] for n in pb['names']]
    priors = [UniformPrior(0., 10.)] * 4 + [UniformPrior(-1., 0.5)]
    f = S.make_vectorized_log_posterior(lc, RefModel(), priors, library=LIB_PATH)
    P = pb['truth'] * (1 + 0.05 * np.random.default_rng(2).standard_normal((16, 5)))
    P[3, 0] = 11.                    # outside its prior: -inf, likelihood skipped (fitting.py:125)
    want = np.array([O.log_posterior(('ShockCooling', pb['orc']), pb['t'], pb['bands'], pb['y'], pb['dy'], pb['priors'], p) for p in P])
    got = f(P)
    assert got[3] == -np.inf and relerr(got, want) < 1e-11
    assert relerr(f(P[0]), want[:1]) < 1e-11
