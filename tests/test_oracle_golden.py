"""The CPU oracle against (i) the golden vectors produced from the reference's own code and (ii) the known answers
KA-1..KA-8 of SURVEY.md section 8c.  CPU only."""
import numpy as np
import pytest

from conftest import golden, relerr
from oracle import lcf_oracle as O

TOL = 5e-13  # the oracle restates the same float64 arithmetic; differences are summation-order noise


def test_constants():
    p = golden('primitives')
    got = [O.K_B, O.C3, O.C4, O.C1, O.C2, O.C_NM_THZ * 10., O.SIGMA_SB]
    assert relerr(got, p['const/values']) < 1e-15


def test_bandpass_normalisation_every_filter():
    g = golden('filters')
    for n in g['filt/names']:
        b = O.band(str(n))
        assert relerr(b.freq, g[f'filt/{n}/freq']) < 1e-14
        tn = g[f'filt/{n}/tnorm']
        nz = tn != 0
        assert np.all(b.tnorm[~nz] == 0)
        assert relerr(b.tnorm[nz], tn[nz]) < 1e-13
        assert relerr([b.freq_eff, b.dfreq], g[f'filt/{n}/scalars'][:2]) < 1e-13


def test_ka1_planck():
    got = O.planck(np.array([500., 1000.]), 10., 1.)
    assert relerr(got, [3.5150683096318397e+18, 2.3396733565090606e+18]) < 1e-15
    assert relerr(got, golden('primitives')['planck/ka1']) < 1e-15


def test_ka2_synthesize():
    want = [5.3340427714584846e+19, 5.5889452729878102e+19, 5.2231143459679003e+19, 5.5347843094141714e+19,
            4.8700621798754804e+19, 4.1267907544148402e+19]
    got = [O.synthesize_blackbody(O.band(b), 12., 3., 0.002) for b in 'UBVgri']
    assert relerr(got, want) < 1e-14
    meta = {'U': (13, 858.321721875779, 159.8316216440046), 'g': (89, 648.9823425403824, 159.2704003790686),
            'i': (89, 402.8409598867557, 56.109424354169235)}
    for name, (k, fe, df) in meta.items():
        b = O.band(name)
        assert len(b.freq) == k and relerr([b.freq_eff, b.dfreq], [fe, df]) < 1e-14


def test_synthesize_grid_redshift_cutoff_extremes():
    p = golden('primitives')
    names = [str(x) for x in p['synth/names']]
    T, R = p['synth/T'], p['synth/R']
    for z in (0., 0.002, 0.5):
        got = np.array([[O.synthesize_blackbody(O.band(n), t, r, z) for t, r in zip(T, R)] for n in names])
        assert relerr(got, p[f'synth/z{z}']) < TOL
    got = np.array([[O.synthesize_blackbody(O.band(n), t, r, 0.01, 300.) for t, r in zip(T, R)] for n in names[:6]])
    assert relerr(got, p['synth/cutoff300_z0.01']) < TOL
    got = np.array([[O.synthesize_blackbody(O.band(n), t, 2., 0.) for t in p['synth/extreme_T']] for n in names[:6]])
    assert relerr(got, p['synth/extreme']) < TOL


def test_ka3_ka4_ka5_shock_cooling():
    p = (1.2, 0.5, 3.0, 2.0, 0.1)
    o = O.ShockCoolingOracle(0.)
    T, R = o.temperature_radius(np.array([1., 2., 5.]), *p)
    assert relerr(T, [25.1133428291314, 17.995858867252704, 11.79422022752378]) < 1e-14
    assert relerr(R, [1.9033451113887767, 3.225723951794573, 5.734466075385747]) < 1e-14
    t = np.array([1., 1, 2, 2, 3, 3, 4, 4])
    bands = [O.band(n) for n in ['g', 'r'] * 4]
    y = O.evaluate(('ShockCooling', o), t, bands, p)
    want = [1.1232544949860943e+20, 7.6915334578502517e+19, 1.7052714937336129e+20, 1.2716501723408432e+20,
            1.9343320921715062e+20, 1.5414360329560972e+20, 1.9793247962254144e+20, 1.6692559258905441e+20]
    assert relerr(y, want) < 1e-14
    assert relerr(O.evaluate(('ShockCooling', O.ShockCoolingOracle(0., 3.)), t, bands, p)[:2],
                  [9.172549391053632e+19, 6.329575711722897e+19]) < 1e-14
    assert relerr(O.evaluate(('ShockCooling', O.ShockCoolingOracle(0., RW=True)), t, bands, p)[:2],
                  [1.104883020716401e+20, 7.437472919442057e+19]) < 1e-14
    assert relerr(O.evaluate(('ShockCooling', O.ShockCoolingOracle(0.01)), t, bands, p)[:2],
                  [1.1372923869790041e+20, 7.8038201614070071e+19]) < 1e-14
    assert relerr(O.evaluate(('ShockCooling4', O.ShockCooling4Oracle(0.)), t, bands, p)[:2],
                  [9.9034622225235608e+19, 6.7807688363102355e+19]) < 1e-14
    y2 = O.evaluate(('ShockCooling2', O.ShockCoolingOracle(0.002)), np.array([1., 2, 3]),
                    [O.band(n) for n in 'UBV'], (30., 3., 30., 0.2))
    assert relerr(y2, [4.968025997157724e+19, 6.780310158705343e+19, 6.926003228090998e+19]) < 1e-14
    # KA-5
    yo = np.array(want)
    s = np.array([1., -1] * 4)
    args = (('ShockCooling', o), t, bands, yo * (1 + 0.05 * s), 0.05 * yo)
    assert relerr(O.log_likelihood(*args, np.array(p)), -358.71468035831157) < 1e-13
    assert relerr(O.log_likelihood(*args, np.array([1.0, 0.7, 2.5, 2.5, 0.2])), -416.5037980733944) < 1e-13
    assert relerr(O.log_likelihood(*args, np.array([1.0, 0.7, 2.5, 2.5, 2.5])), -1442.1547722087005) < 1e-13
    assert relerr(O.log_likelihood(*args, np.array(p + (0.5,)), True, 'relative'), -358.8072545635684) < 1e-13
    assert relerr(O.log_likelihood(*args, np.array(p + (0.5,)), True, 'absolute'), -358.92913923329746) < 1e-13
    with pytest.raises(Exception):
        O.log_likelihood(*args, np.array(p), False, 'bogus')


VARIANTS = {'n15': dict(n=1.5), 'n3': dict(n=3.), 'rw': dict(RW=True), 'n3rw': dict(n=3., RW=True)}


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_shock_cooling_blocks(tag):
    s = golden('shockcooling')
    t, bands = s['scb/t'], [O.band(str(n)) for n in s['scb/names']]
    o = O.ShockCoolingOracle(0.01, **VARIANTS[tag])
    T, R = o.temperature_radius(t, *s['scb/P'].T)
    assert relerr(T.T, s[f'scb/{tag}/T']) < TOL and relerr(R.T, s[f'scb/{tag}/R']) < TOL
    assert relerr(O.evaluate(('ShockCooling', o), t, bands, s['scb/P'].T).T, s[f'scb/{tag}/y']) < TOL
    assert relerr(O.log_likelihood(('ShockCooling', o), t, bands, s['scb/y'], s['scb/dy'], s['scb/P'].T),
                  s[f'scb/{tag}/ll']) < TOL
    assert relerr(O.evaluate(('ShockCooling2', o), t, bands, s['scb/P2'].T).T, s[f'scb/{tag}/y2']) < TOL
    assert relerr(O.log_likelihood(('ShockCooling2', o), t, bands, s['scb/y'], s['scb/dy'], s['scb/P2'].T),
                  s[f'scb/{tag}/ll2']) < TOL


def test_shock_cooling_sigma_sc4_and_edges():
    s = golden('shockcooling')
    t, bands = s['scb/t'], [O.band(str(n)) for n in s['scb/names']]
    y, dy = s['scb/y'], s['scb/dy']
    o = O.ShockCoolingOracle(0.01)
    Ps = np.column_stack([s['scb/P'], s['scb/sigma']])
    m = ('ShockCooling', o)
    assert relerr(O.log_likelihood(m, t, bands, y, dy, Ps.T, True, 'relative'), s['scb/n15/ll_rel']) < TOL
    assert relerr(O.log_likelihood(m, t, bands, y, dy, Ps.T, True, 'absolute'), s['scb/n15/ll_abs']) < TOL
    # reference-shaped (per-point loop) and batched forms agree
    a = O.log_likelihood(m, t, bands, y, dy, s['scb/P'][3], reference_shaped=True)
    assert relerr(a, s['scb/n15/ll'][3]) < TOL
    m4 = ('ShockCooling4', O.ShockCooling4Oracle(0.01))
    assert relerr(O.evaluate(m4, t, bands, s['scb/P'].T).T, s['scb/sc4/y']) < TOL
    assert relerr(O.log_likelihood(m4, t, bands, y, dy, s['scb/P'].T), s['scb/sc4/ll']) < TOL
    # out-of-domain parameters: zeros and NaNs exactly where the reference has them
    for tag, mm in (('sc', m), ('sc4', m4)):
        assert relerr(O.evaluate(mm, t, bands, s['sce/P'].T).T, s[f'sce/{tag}/y']) < TOL
        assert relerr(O.log_likelihood(mm, t, bands, y, dy, s['sce/P'].T), s[f'sce/{tag}/ll']) < TOL
    m2 = ('ShockCooling2', o)
    assert relerr(O.evaluate(m2, t, bands, s['sce/P2'].T).T, s['sce/sc2/y']) < TOL
    assert relerr(O.log_likelihood(m2, t, bands, y, dy, s['sce/P2'].T), s['sce/sc2/ll']) < TOL


def test_ka6_ka7_companion():
    T, R = O.kasen_temperature_radius(np.array([2., 5., 20.]), 1., 0.5, 1.2)
    assert relerr(T, [21.049044173239672, 10.323820373101428, 4.635484277247457]) < 1e-14
    assert relerr(R, [2.75525424632749, 8.098984401632642, 27.21120752464609]) < 1e-14
    t = np.repeat([57003., 57010, 57020, 57040], 6)
    bands = [O.band(n) for n in ['U', 'B', 'V', 'g', 'r', 'i'] * 4]
    lum = 1e20 * (1 + 0.1 * np.arange(24))
    m = ('CompanionShocking', O.CompanionShockingOracle(bands, lum, 0.003, 1))
    q = (57001., 0.5, 1.2, 57018., 1.05, 0.95, 0.9, 0.6)
    y = O.evaluate(m, t, bands, q)
    assert relerr(y[:6], [1.6808975459233330e+20, 2.5838140437096995e+20, 2.2476288897286065e+20,
                          2.5162134821899239e+20, 2.0582230618486483e+20, 1.7538132940926642e+20]) < 1e-14
    assert relerr(y[18:], [3.2950548244318814e+19, 7.0722999175115620e+19, 1.6402605486730134e+20,
                           9.7433040808173289e+19, 1.9879981776195777e+20, 2.6559499948788993e+20]) < 1e-14
    assert relerr(O.log_likelihood(m, t, bands, lum, 0.05 * lum, np.array(q)), -4152.2519961946) < 1e-12


def test_companion_blocks_and_edges():
    c = golden('companion')
    t, bands = c['csb/t'], [O.band(str(n)) for n in c['csb/names']]
    lum, dlum = c['csb/lum'], c['csb/dlum']
    for v in (1, 2, 3):
        m = ('CompanionShocking', O.CompanionShockingOracle(bands, lum, 0.003, v))
        P = c[f'csb/P{v}']
        assert relerr(np.array([O.evaluate(m, t, bands, p) for p in P]), c[f'csb/y{v}']) < TOL
        assert relerr(np.array([O.log_likelihood(m, t, bands, lum, dlum, p) for p in P]), c[f'csb/ll{v}']) < TOL
    m = ('CompanionShocking', O.CompanionShockingOracle(bands, lum, 0.003, 1))
    got = np.array([O.log_likelihood(m, t, bands, lum, dlum, np.append(p, 0.7), True, 'absolute') for p in c['csb/P1']])
    assert relerr(got, c['csb/ll1_sigma_abs']) < TOL
    assert relerr(np.array([O.evaluate(m, t, bands, p) for p in c['cse/P1']]), c['cse/y1']) < TOL
    with pytest.raises(Exception, match='No SiFTO template'):
        O.CompanionShockingOracle([O.band('z')], [1.], 0., 1)


def test_config_shapes():
    g = golden('config2')
    bands = [O.band(str(n)) for n in g['cfg2/names']]
    m = ('ShockCooling', O.ShockCoolingOracle(0., 1.5))
    got = O.log_likelihood(m, g['cfg2/t'], bands, g['cfg2/y'], g['cfg2/dy'], g['cfg2/P'].T)
    assert relerr(got, g['cfg2/ll']) < TOL
    assert relerr(O.evaluate(m, g['cfg2/t'], bands, g['cfg2/P'][0]), g['cfg2/yfit0']) < TOL
    g = golden('config3')
    bands = [O.band(str(n)) for n in g['cfg3/names']]
    m = ('CompanionShocking', O.CompanionShockingOracle(bands, g['cfg3/y'], 0.003, 1))
    got = np.array([O.log_likelihood(m, g['cfg3/t'], bands, g['cfg3/y'], g['cfg3/dy'], p) for p in g['cfg3/P'][:4]])
    assert relerr(got, g['cfg3/ll'][:4]) < TOL


def test_ka8_priors_and_bolometric_helpers():
    p = golden('primitives')
    assert relerr(O.pseudo(10., 1., 0.), 1.9045964708399877e+33) < 1e-14
    assert relerr(O.stefan_boltzmann(10., 1.), 3.448780921664817e+33) < 1e-14
    assert relerr(O.mag2flux(-17., 0.05, 34.090065622282225), (2.7291427281800803e+20, 1.256816672512111e+19)) < 1e-14
    assert relerr(O.pseudo(10., 1., 0.), p['misc/pseudo_10_1_0']) < 1e-15
    xs = p['prior/x']
    for key, pr in (('prior/uniform_0_1', (0, 0., 1., 0., 1.)), ('prior/loguniform_0.01_1000', (1, .01, 1000., 0., 1.)),
                    ('prior/gaussian_0_10_0_1', (2, 0., 10., 0., 1.))):
        got = np.array([O.log_prior([pr], [x]) for x in xs])
        assert relerr(got, p[key]) < 1e-15
    assert O.log_prior([(0, 0., 1., 0., 1.)], [0.5]) == 0. and O.log_prior([(0, 0., 1., 0., 1.)], [1.0]) == -np.inf
    assert O.log_prior([(1, .01, 1000., 0., 1.)], [2.]) == -0.6931471805599453
    assert O.log_prior([(2, 0., 10., 0., 1.)], [2.]) == -2.0
    names = [str(x) for x in p['bolo/names']]
    m = ('Blackbody', type('Z', (), {'z': 0.01})())
    got = np.array([O.evaluate(m, None, [O.band(n) for n in names], (t, r)) for t, r in zip(p['bolo/T'], p['bolo/R'])])
    assert relerr(got, p['bolo/y_z0.01']) < TOL


def test_philox_known_answers_and_sampler_statistics():
    assert [int(x) for x in O.philox4x32((0, 0, 0, 0), (0, 0))] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert [int(x) for x in O.philox4x32((f, f, f, f), (f, f))] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(x) for x in O.philox4x32((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0))] \
        == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # the stretch move samples a correlated 3-D Gaussian correctly (mean and covariance)
    rng = np.random.default_rng(0)
    A = rng.standard_normal((3, 3))
    cov = A @ A.T + np.eye(3)
    icov = np.linalg.inv(cov)
    mu = np.array([1., -2., 0.5])

    def lp(x):
        d = np.atleast_2d(x) - mu
        return -0.5 * np.einsum('ni,ij,nj->n', d, icov, d)

    chain, lps, nacc = O.stretch_move_run(lp, mu + rng.standard_normal((40, 3)), 1500, seed=42)
    flat = chain[300:].reshape(-1, 3)
    assert np.all(np.abs(flat.mean(0) - mu) < 0.15)
    assert np.all(np.abs(np.cov(flat.T) - cov) < 0.25 * np.abs(cov).max())
    assert 0.3 < nacc.mean() / 1500 < 0.85
    z, j, lnu = O.stretch_draws(7, 3, 1, np.arange(1000), 500)
    assert z.min() >= 0.5 and z.max() <= 2.0 and j.min() >= 0 and j.max() < 500 and np.all(lnu < 0)


def test_sed_likelihoods():
    """Per-epoch blackbody SED log-likelihoods (bolometric.py:154-164) for (T, R[, sigma]) candidates."""
    g = golden('sed')
    z = float(g['sed/z'])
    m = ('Blackbody', type('Z', (), {'z': z})())
    off = g['sed/ep_off']
    for e in range(len(off) - 1):
        sl = slice(off[e], off[e + 1])
        bands = [O.band(str(n)) for n in g['sed/names'][sl]]
        c = g['sed/cand'][e]
        args = (m, None, bands, g['sed/y'][sl], g['sed/dy'][sl])
        assert relerr(O.log_likelihood(*args, c[:, :2].T), g['sed/ll'][e]) < TOL
        assert relerr(O.log_likelihood(*args, c.T, True, 'relative'), g['sed/ll_rel'][e]) < TOL
        assert relerr(O.log_likelihood(*args, c.T, True, 'absolute'), g['sed/ll_abs'][e]) < TOL


def test_c_oracle_matches_reference_numbers():
    """oracle/lcf_oracle_c.c (plain C, no fast-math) against the reference's log-likelihoods, edge cases included."""
    import os
    import subprocess
    from conftest import ROOT
    if not os.path.exists(os.path.join(ROOT, 'oracle', 'liblcf_oracle.so')):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True, capture_output=True)
    g = golden('config2')
    bands = [O.band(str(n)) for n in g['cfg2/names']]
    got = O.c_shock_cooling_loglike(O.ShockCoolingOracle(0., 1.5), g['cfg2/t'], bands, g['cfg2/y'], g['cfg2/dy'],
                                    g['cfg2/P'], n_threads=2)
    assert relerr(got, g['cfg2/ll']) < TOL
    s = golden('shockcooling')
    bands = [O.band(str(n)) for n in s['scb/names']]
    for tag, kw in VARIANTS.items():
        got = O.c_shock_cooling_loglike(O.ShockCoolingOracle(0.01, **kw), s['scb/t'], bands, s['scb/y'], s['scb/dy'],
                                        s['scb/P'])
        assert relerr(got, s[f'scb/{tag}/ll']) < TOL
    got = O.c_shock_cooling_loglike(O.ShockCoolingOracle(0.01), s['scb/t'], bands, s['scb/y'], s['scb/dy'], s['sce/P'])
    assert relerr(got, s['sce/sc/ll']) < TOL  # zeros and NaNs where the reference has them


def test_fitzpatrick99_known_answer_and_definition():
    """The published law: the README example of the third-party `extinction` package (all printed digits), the
    anchor value at 5470 A, the limits, and continuity where the ultraviolet formula takes over."""
    got = O.fitzpatrick99(np.array([2000., 4000., 8000.]), 1.0, 3.1)
    assert np.array_equal(np.round(got, 8), [2.76225609, 1.42325373, 0.55333671])
    k5 = -5.13540e-02 + 1.00216 * 3.1 - 7.35778e-05 * 3.1 ** 2 - 3.1
    assert O.fitzpatrick99(np.array([5470.]), 3.1, 3.1)[0] == pytest.approx(3.1 + k5, rel=1e-14)
    assert abs(O.fitzpatrick99(np.array([1e9]), 1., 3.1)[0]) < 1e-4
    lo, hi = O.fitzpatrick99(np.array([2700. * (1 + 1e-9), 2700. * (1 - 1e-9)]), 1., 3.1)
    assert abs(lo - hi) < 1e-8
    assert O.fitzpatrick99(np.array([4400.]), 3.1)[0] - O.fitzpatrick99(np.array([5500.]), 3.1)[0] == pytest.approx(1., abs=0.05)


def test_shockcooling3_against_reference_vectors():
    """'a': E(B-V) = 0, the reference's own arithmetic only; 'b': E(B-V) != 0, the reference's code around the
    restated law (tools/refgen/standins/extinction)."""
    g = golden('shockcooling3')
    t, names, P = g['sc3/t'], [str(n) for n in g['sc3/names']], g['sc3/P']
    bands = [O.band(n) for n in names]
    model = ('ShockCooling3', O.ShockCoolingOracle(z=0.012))
    P0 = P.copy()
    P0[:, 5] = 0.
    assert relerr(O.evaluate(model, t, bands, P0.T).T, g['sc3/a_y']) < TOL
    assert relerr(O.evaluate(model, t, bands, P.T).T, g['sc3/b_y']) < TOL
    assert relerr(np.array([O.evaluate(model, t, bands, p, reference_shaped=True) for p in P]), g['sc3/b_y']) < TOL
    y, dy = g['sc3/flux'], g['sc3/dflux']
    assert relerr(O.log_likelihood(model, t, bands, y, dy, P0.T), g['sc3/a_lnl']) < TOL
    assert relerr(O.log_likelihood(model, t, bands, y, dy, P.T), g['sc3/b_lnl']) < TOL
    assert relerr(O.log_likelihood(model, t, bands, y, dy, g['sc3/Ps'].T, use_sigma=True), g['sc3/b_lnl_sigma']) < TOL
    # dense branch of the reference for parameter arrays: (every 15th filter row, ntimes, nwalkers)
    rows = range(0, len(t), 15)
    for j, r in enumerate(rows):
        got = O.evaluate(model, t, [bands[r]] * len(t), P.T)
        assert relerr(got, g['sc3/b_y_block'][j]) < TOL
