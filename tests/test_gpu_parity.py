"""Parity of the HIP engine (through the C ABI) against the golden vectors from the reference and the CPU oracle.
Tolerance: north_star asks for 1e-6 relative on float64; the engine is held to 1e-11 here (measured ~1e-14)."""
import numpy as np
import pytest

from conftest import golden, relerr
from helpers import config2_case, lc_dict, shockcooling_case
from lightcurve_fitting_amd import engine as E, models as M
from oracle import lcf_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-11
VARIANTS = {'n15': dict(n=1.5), 'n3': dict(n=3.), 'rw': dict(RW=True), 'n3rw': dict(n=3., RW=True)}


# band-sum levels (lcf_engine_set_variant): 3 = the interpolants of ln S(ln T), what an engine starts with and what every
# bench line times; 2 = Gauss-compressed tables; 1 = full tables, fused exponentials; 0 = libm, shaped like the reference
@pytest.fixture(params=[3, 2, 1, 0], ids=['interp', 'gauss', 'fast', 'libm'])
def variant(request):
    return request.param


def _eng(model, lc, variant, *a, **k):
    e = model.engine_for(lc, *a, **k)
    e.set_variant(variant)
    return e


def test_native_library_is_the_one_in_tree():
    lib = E.load_library()
    assert lib.lcf_device_count() >= 1
    assert E.LIB_PATH.endswith('lightcurve_fitting_amd/csrc/liblcf_hip.so')


def test_blackbody_to_filters_primitive(variant):
    p = golden('primitives')
    names = [str(x) for x in p['synth/names']]
    for z in (0., 0.002, 0.5):
        # every filter x every (T, R) -> dense branch (models.py:1163-1164)
        got = M.blackbody_to_filters(names, p['synth/T'], p['synth/R'], z=z, variant=variant)
        assert relerr(got, p[f'synth/z{z}']) < TOL
    got = M.blackbody_to_filters(names[:6], p['synth/T'], p['synth/R'], z=0.01, cutoff_freq=300., variant=variant)
    assert relerr(got, p['synth/cutoff300_z0.01']) < TOL
    got = M.blackbody_to_filters(names[:6], p['synth/extreme_T'], np.full(7, 2.), variant=variant)
    assert relerr(got, p['synth/extreme']) < TOL  # exp overflow -> 0, T <= 0 -> 0
    # pointwise branch (models.py:1161-1162) and KA-2
    got = M.blackbody_to_filters(list('UBVgri'), np.full(6, 12.), np.full(6, 3.), z=0.002, variant=variant)
    assert relerr(got, [5.3340427714584846e+19, 5.5889452729878102e+19, 5.2231143459679003e+19,
                        5.5347843094141714e+19, 4.8700621798754804e+19, 4.1267907544148402e+19]) < TOL
    with pytest.raises(Exception, match='same shape'):
        M.blackbody_to_filters(['g'], np.ones(2), np.ones(3))
    # bolometric-style direct (T, R) model (bolometric.py:154-164)
    bn = [str(x) for x in p['bolo/names']]
    bb = M.Blackbody(redshift=0.01)
    got = bb(np.zeros(6), bn, p['bolo/T'], p['bolo/R'])  # (6, 32)
    assert relerr(got.T, p['bolo/y_z0.01']) < TOL


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_shock_cooling_variants(tag, variant):
    s, lc = shockcooling_case()
    m = M.ShockCooling(redshift=0.01, **VARIANTS[tag])
    e = _eng(m, lc, variant)
    T, R = e.temperature_radius(s['scb/P'])
    assert relerr(T, s[f'scb/{tag}/T']) < TOL and relerr(R, s[f'scb/{tag}/R']) < TOL
    assert relerr(e.evaluate(s['scb/P']), s[f'scb/{tag}/y']) < TOL
    assert relerr(m.log_likelihood(lc, s['scb/P']), s[f'scb/{tag}/ll']) < TOL
    assert m.log_likelihood(lc, s['scb/P'][2]) == pytest.approx(s[f'scb/{tag}/ll'][2], rel=TOL)  # 1-D p -> float
    m2 = M.ShockCooling2(redshift=0.01, **VARIANTS[tag])
    e2 = _eng(m2, lc, variant)
    assert relerr(e2.evaluate(s['scb/P2']), s[f'scb/{tag}/y2']) < TOL
    assert relerr(m2.log_likelihood(lc, s['scb/P2']), s[f'scb/{tag}/ll2']) < TOL


def test_sigma_modes_sc4_and_out_of_domain_parameters(variant):
    s, lc = shockcooling_case()
    m = M.ShockCooling(redshift=0.01)
    Ps = np.column_stack([s['scb/P'], s['scb/sigma']])
    _eng(m, lc, variant, True, 'relative')
    assert relerr(m.log_likelihood(lc, Ps, use_sigma=True), s['scb/n15/ll_rel']) < TOL
    _eng(m, lc, variant, True, 'absolute')
    assert relerr(m.log_likelihood(lc, Ps, use_sigma=True, sigma_type='absolute'), s['scb/n15/ll_abs']) < TOL
    with pytest.raises(Exception, match='sigma_type'):
        m.log_likelihood(lc, Ps, use_sigma=True, sigma_type='bogus')
    m4 = M.ShockCooling4(redshift=0.01)
    e4 = _eng(m4, lc, variant)
    T, R = e4.temperature_radius(s['scb/P'])
    assert relerr(T, s['scb/sc4/T']) < TOL and relerr(R, s['scb/sc4/R']) < TOL
    assert relerr(e4.evaluate(s['scb/P']), s['scb/sc4/y']) < TOL
    assert relerr(m4.log_likelihood(lc, s['scb/P']), s['scb/sc4/ll']) < TOL
    # power() semantics: zeros and NaNs exactly where the reference produces them (relerr checks the NaN pattern)
    e = _eng(m, lc, variant)
    assert relerr(e.evaluate(s['sce/P']), s['sce/sc/y']) < TOL
    assert relerr(m.log_likelihood(lc, s['sce/P']), s['sce/sc/ll']) < TOL
    assert relerr(e4.evaluate(s['sce/P']), s['sce/sc4/y']) < TOL
    assert relerr(m4.log_likelihood(lc, s['sce/P']), s['sce/sc4/ll']) < TOL
    m2 = M.ShockCooling2(redshift=0.01)
    e2 = _eng(m2, lc, variant)
    assert relerr(e2.evaluate(s['sce/P2']), s['sce/sc2/y']) < TOL
    assert relerr(m2.log_likelihood(lc, s['sce/P2']), s['sce/sc2/ll']) < TOL


def test_known_answers_ka3_ka4_ka5():
    p = (1.2, 0.5, 3.0, 2.0, 0.1)
    t = np.array([1., 1, 2, 2, 3, 3, 4, 4])
    f = ['g', 'r'] * 4
    want = np.array([1.1232544949860943e+20, 7.6915334578502517e+19, 1.7052714937336129e+20, 1.2716501723408432e+20,
                     1.9343320921715062e+20, 1.5414360329560972e+20, 1.9793247962254144e+20, 1.6692559258905441e+20])
    m = M.ShockCooling(redshift=0.)
    assert relerr(m(t, f, *p), want) < TOL
    T, R = m.temperature_radius(np.array([1., 2., 5.]), *p)
    assert relerr(T, [25.1133428291314, 17.995858867252704, 11.79422022752378]) < TOL
    assert relerr(R, [1.9033451113887767, 3.225723951794573, 5.734466075385747]) < TOL
    assert relerr(M.ShockCooling(redshift=0., n=3.)(t, f, *p)[:2], [9.172549391053632e+19, 6.329575711722897e+19]) < TOL
    assert relerr(M.ShockCooling4(redshift=0.)(t, f, *p)[:2], [9.9034622225235608e+19, 6.7807688363102355e+19]) < TOL
    assert relerr(M.ShockCooling2(redshift=0.002)(np.array([1., 2, 3]), list('UBV'), 30., 3., 30., 0.2),
                  [4.968025997157724e+19, 6.780310158705343e+19, 6.926003228090998e+19]) < TOL
    s = np.array([1., -1] * 4)
    lc = lc_dict(t, f, want * (1 + 0.05 * s), 0.05 * want)
    assert m.log_likelihood(lc, np.array(p)) == pytest.approx(-358.71468035831157, rel=TOL)
    assert m.log_likelihood(lc, np.array([1.0, 0.7, 2.5, 2.5, 2.5])) == pytest.approx(-1442.1547722087005, rel=TOL)
    assert m.log_likelihood(lc, np.array(p + (0.5,)), use_sigma=True) == pytest.approx(-358.8072545635684, rel=TOL)
    assert m.log_likelihood(lc, np.array(p + (0.5,)), use_sigma=True, sigma_type='absolute') == \
        pytest.approx(-358.92913923329746, rel=TOL)
    # dense grid call: (nfilters, ntimes[, nwalkers]) like the reference's plotting path (fitting.py:337-352)
    grid = m(np.array([1., 2., 3., 4.]), ['g', 'r'], *p)
    assert grid.shape == (2, 4) and relerr(grid[0], want[0::2]) < TOL and relerr(grid[1], want[1::2]) < TOL
    grid = m(np.array([1., 2., 3., 4.]), ['g', 'r'], *[np.array([x, x]) for x in p])
    assert grid.shape == (2, 4, 2) and relerr(grid[1, :, 1], want[1::2]) < TOL


def test_companion_shocking_family(variant):
    c = golden('companion')
    lc = lc_dict(c['csb/t'], c['csb/names'], c['csb/lum'], c['csb/dlum'])
    for v, cls in ((1, M.CompanionShocking), (2, M.CompanionShocking2), (3, M.CompanionShocking3)):
        m = cls(lc, redshift=0.003)
        e = _eng(m, lc, variant)
        assert relerr(e.evaluate(c[f'csb/P{v}']), c[f'csb/y{v}']) < TOL
        assert relerr(m.log_likelihood(lc, c[f'csb/P{v}']), c[f'csb/ll{v}']) < TOL
    m = M.CompanionShocking(lc, redshift=0.003)
    _eng(m, lc, variant, True, 'absolute')
    P = np.column_stack([c['csb/P1'], np.full(len(c['csb/P1']), 0.7)])
    assert relerr(m.log_likelihood(lc, P, use_sigma=True, sigma_type='absolute'), c['csb/ll1_sigma_abs']) < TOL
    e = _eng(m, lc, variant)
    assert relerr(e.evaluate(c['cse/P1']), c['cse/y1']) < TOL
    assert relerr(m.log_likelihood(lc, c['cse/P1']), c['cse/ll1']) < TOL
    # KA-6 / KA-7
    T, R = m.temperature_radius(np.array([2., 5., 20.]), 1., 0.5, 1.2)
    assert relerr(T, [21.049044173239672, 10.323820373101428, 4.635484277247457]) < TOL
    assert relerr(R, [2.75525424632749, 8.098984401632642, 27.21120752464609]) < TOL
    t = np.repeat([57003., 57010, 57020, 57040], 6)
    lum = 1e20 * (1 + 0.1 * np.arange(24))
    lc7 = lc_dict(t, ['U', 'B', 'V', 'g', 'r', 'i'] * 4, lum, 0.05 * lum)
    m7 = M.CompanionShocking(lc7, redshift=0.003)
    q = np.array([57001., 0.5, 1.2, 57018., 1.05, 0.95, 0.9, 0.6])
    assert m7.log_likelihood(lc7, q) == pytest.approx(-4152.2519961946, rel=1e-12)
    assert relerr(m7(t, lc7['filter'], *q), c['cs/ka7_y']) < TOL


def test_config2_and_config3_shapes(variant):
    g, lc = config2_case()
    m = M.ShockCooling(redshift=0.)
    e = _eng(m, lc, variant)
    assert e.samples_per_eval == 500 * (11 + 9 + 13 + 87 + 73 + 87)  # zero-weight samples dropped at pack time
    assert relerr(m.log_likelihood(lc, g['cfg2/P']), g['cfg2/ll']) < TOL
    assert relerr(e.evaluate(g['cfg2/P'][:1])[0], g['cfg2/yfit0']) < TOL
    g3 = golden('config3')
    lc3 = lc_dict(g3['cfg3/t'], g3['cfg3/names'], g3['cfg3/y'], g3['cfg3/dy'])
    m3 = M.CompanionShocking(lc3, redshift=0.003)
    _eng(m3, lc3, variant)
    assert relerr(m3.log_likelihood(lc3, g3['cfg3/P']), g3['cfg3/ll']) < TOL


def test_full_size_block_against_oracle_and_properties():
    """BASELINE.json configs[1] at full width: 1024 walkers x 3000 points, checked against the vectorised oracle and
    through size-independent properties."""
    g, lc = config2_case()
    rng = np.random.default_rng(11)
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    P = truth * rng.uniform(0.8, 1.2, (1024, 5))
    m = M.ShockCooling(redshift=0.)
    ll = m.log_likelihood(lc, P)
    bands = [O.band(n) for n in lc['filter']]
    sel = rng.choice(1024, 96, replace=False)
    ref = O.log_likelihood(('ShockCooling', O.ShockCoolingOracle(0.)), lc['MJD'], bands, lc['lum'], lc['dlum'], P[sel].T)
    assert relerr(ll[sel], ref) < TOL
    # (i) bitwise run-to-run determinism, (ii) invariance to the order of walkers, (iii) batch == singles
    assert np.array_equal(ll, m.log_likelihood(lc, P))
    perm = rng.permutation(1024)
    assert np.array_equal(m.log_likelihood(lc, P[perm]), ll[perm])
    assert np.array_equal(np.array([m.log_likelihood(lc, p) for p in P[:5]]), ll[:5])
    # (iv) invariance to the order of the data points (the engine re-sorts by filter): same value to rounding
    shuffle = rng.permutation(len(lc['MJD']))
    lc_s = lc_dict(lc['MJD'][shuffle], np.array(lc['filter'])[shuffle], lc['lum'][shuffle], lc['dlum'][shuffle])
    assert relerr(M.ShockCooling(redshift=0.).log_likelihood(lc_s, P[:64]), ll[:64]) < 1e-13
    # (v) additivity over disjoint subsets of the photometry
    half = np.arange(len(lc['MJD'])) % 2 == 0
    parts = [lc_dict(lc['MJD'][k], np.array(lc['filter'])[k], lc['lum'][k], lc['dlum'][k]) for k in (half, ~half)]
    s = sum(M.ShockCooling(redshift=0.).log_likelihood(q, P[:64]) for q in parts)
    assert relerr(s, ll[:64]) < 1e-13
    # (vi) chi^2 scaling: inflating every uncertainty by c changes lnL by the closed form
    cfac = 1.7
    lc_c = lc_dict(lc['MJD'], lc['filter'], lc['lum'], cfac * lc['dlum'])
    ll_c = M.ShockCooling(redshift=0.).log_likelihood(lc_c, P[:64])
    n = len(lc['MJD'])
    const = -0.5 * np.sum(np.log(2 * np.pi * lc['dlum'] ** 2))
    expect = const - n * np.log(cfac) + (ll[:64] - const) / cfac ** 2
    assert relerr(ll_c, expect) < 1e-12


def test_log_posterior_prior_short_circuit():
    s, lc = shockcooling_case()
    m = M.ShockCooling(redshift=0.01)
    priors = [M.UniformPrior(0., 10.), M.LogUniformPrior(0.01, 10.), M.GaussianPrior(0., 10., 3., 2.),
              M.UniformPrior(0., 10.), M.UniformPrior(-1., 0.5)]
    e = m.engine_for(lc, priors=priors)
    P = s['scb/P'].copy()
    P[3, 0] = 10.0   # on the boundary: strict inequality -> -inf
    P[5, 4] = 0.7    # outside
    got = e.log_posterior(P)
    ll = e.log_likelihood(P)
    for i, p in enumerate(P):
        lp = sum(pr(x) for pr, x in zip(priors, p))
        if np.isinf(lp):
            assert got[i] == -np.inf
        else:
            assert got[i] == pytest.approx(lp + ll[i], rel=1e-14)
    assert got[3] == -np.inf and got[5] == -np.inf and np.isfinite(got[0])
    bands = [O.band(n) for n in lc['filter']]
    desc = [pr.descriptor() for pr in priors]
    ref = np.array([O.log_posterior(('ShockCooling', O.ShockCoolingOracle(0.01)), lc['MJD'], bands, lc['lum'],
                                    lc['dlum'], desc, p) for p in P])
    assert relerr(got, ref) < TOL


def test_edge_shapes_and_errors():
    m = M.ShockCooling(redshift=0.)
    one = lc_dict([2.0], ['g'], [1e20], [1e18])
    bands = [O.band('g')]
    p = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    assert m.log_likelihood(one, p) == pytest.approx(
        float(O.log_likelihood(('ShockCooling', O.ShockCoolingOracle(0.)), [2.0], bands, [1e20], [1e18], p)), rel=TOL)
    e = m.engine_for(one)
    assert e.log_likelihood(np.empty((0, 5))).shape == (0,)
    with pytest.raises(ValueError, match='shape'):
        e.log_likelihood(np.ones((3, 4)))
    # a light curve spanning several chunks and many filters, ragged counts per filter
    rng = np.random.default_rng(3)
    names = rng.choice(['UVW2', 'UVM2', 'UVW1', 'U', 'B', 'V', 'g', 'r', 'i', 'z', 'y', 'J', 'H', 'K', 'w', 'G'], 777)
    t = rng.uniform(0.3, 20., 777)
    y = 1e20 * rng.uniform(0.5, 2., 777)
    lc = lc_dict(t, names, y, 0.1 * y)
    P = p * rng.uniform(0.7, 1.3, (17, 5))
    ref = O.log_likelihood(('ShockCooling', O.ShockCoolingOracle(0.)), t, [O.band(n) for n in names], y, 0.1 * y, P.T)
    assert relerr(m.log_likelihood(lc, P), ref) < TOL
    # very wide tables (JWST, 1482 rows): table slice read from global memory when it exceeds the LDS window
    names = rng.choice(['F2550W', 'F2100W', 'F1800W', 'F444W', 'F356W', 'NUV', 'FUV', 'F070W'], 300)
    t = rng.uniform(0.3, 20., 300)
    lc = lc_dict(t, names, y[:300], 0.1 * y[:300])
    ref = O.log_likelihood(('ShockCooling', O.ShockCoolingOracle(0.)), t, [O.band(n) for n in names], y[:300],
                           0.1 * y[:300], P.T)
    assert relerr(m.log_likelihood(lc, P), ref) < TOL


def test_config1_example_light_curve_end_to_end():
    """BASELINE configs[0]: the reference's tutorial fit on the included SN 2016bkv photometry
    (docs/source/usage.rst:174-200), from magnitudes to log-likelihoods, against the reference's own numbers."""
    from lightcurve_fitting_amd.fitting import lightcurve_mcmc
    from lightcurve_fitting_amd.lightcurve import LC
    g = golden('config1')
    lc = LC({'MJD': g['cfg1/MJD'], 'mag': g['cfg1/mag'], 'dmag': g['cfg1/dmag'], 'filter': g['cfg1/filter'],
             'nondet': g['cfg1/nondet'], 'source': g['cfg1/source']}, meta={'dm': 30.79, 'redshift': 0.002})
    lc.calcAbsMag()
    lc.calcLum()
    early = lc.where(MJD_min=57468., MJD_max=57485.)
    model = M.ShockCooling2(early)
    assert model.z == 0.002
    assert relerr(model.log_likelihood(early, g['cfg1/P1a']), g['cfg1/ll1a']) < TOL
    sel = lc[g['cfg1/sel1b']]
    m5 = M.ShockCooling(sel)
    assert relerr(m5.log_likelihood(sel, g['cfg1/P1b']), g['cfg1/ll1b']) < TOL
    # the documented fit, 64 walkers (configs[0]); finite chain inside the priors, better than the starting box
    priors = [M.UniformPrior(0., 100.)] * 3 + [M.UniformPrior(57468., 57468.7)]
    np.random.seed(1)
    sampler = lightcurve_mcmc(early, model, priors=priors, p_lo=[20., 2., 20., 57468.5], p_up=[50., 5., 50., 57468.7],
                              nwalkers=64, nsteps=100, nsteps_burnin=200)
    assert sampler.flatchain.shape == (6400, 4) and np.all(np.isfinite(sampler.flatchain))
    assert np.all(sampler.flatchain[:, 3] > 57468.) and np.all(sampler.flatchain[:, 3] < 57468.7)
    assert np.median(sampler.flatlnprobability) > np.max(g['cfg1/ll1a'])
    t_max = model.t_max(sampler.flatchain.mean(axis=0))
    assert np.isfinite(t_max)


def test_log_posterior_callable_and_degenerate_inputs():
    """The drop-in seam itself (fitting.py:121-130): one callable, per-walker and vectorised forms; an empty light
    curve is refused; NaN photometry propagates; workspace growth across very different batch sizes."""
    from lightcurve_fitting_amd.fitting import make_log_posterior
    s, lc = shockcooling_case()
    m = M.ShockCooling(redshift=0.01)
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 10.)]
    f = make_log_posterior(lc, m, priors)
    block = f(s['scb/P'])
    assert block.shape == (24,) and isinstance(f(s['scb/P'][0]), float)
    assert np.array_equal(np.array([f(p) for p in s['scb/P']]), block)
    assert relerr(block, s['scb/n15/ll']) < TOL                      # uniform priors add 0 inside their bounds
    assert f(np.array([11., 1., 1., 1., 0.])) == -np.inf
    big = np.tile(s['scb/P'], (300, 1))                                # 7200 walkers: workspace regrowth
    assert np.array_equal(f(big)[:24], block) and np.array_equal(f(big)[-24:], block)
    assert np.array_equal(f(s['scb/P'][:3]), block[:3])
    # no data at all is rejected at the boundary (an engine needs at least one band table)
    with pytest.raises((E.LcfError, ValueError)):
        M.ShockCooling(redshift=0.).make_engine(np.zeros(0), [], np.zeros(0), np.zeros(0))
    # NaN photometry -> NaN likelihood (the reference propagates it the same way)
    bad = lc_dict(lc['MJD'], lc['filter'], lc['lum'].copy(), lc['dlum'])
    bad['lum'][5] = np.nan
    assert np.all(np.isnan(M.ShockCooling(redshift=0.01).log_likelihood(bad, s['scb/P'][:4])))


def test_device_pointer_entry_points_with_torch_tensors():
    """lcf_log_likelihood_dev / lcf_log_posterior_dev: walker block and output stay in HBM (torch tensors), work is
    enqueued on torch's current stream without a host round-trip; results equal the host-pointer entry points."""
    import torch
    s, lc = shockcooling_case()
    m = M.ShockCooling(redshift=0.01)
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    eng = m.engine_for(lc, priors=priors)
    P = np.tile(s['scb/P'], (20, 1))
    want_ll, want_lp = eng.log_likelihood(P), eng.log_posterior(P)
    dP = torch.tensor(P, dtype=torch.float64, device='cuda')
    out = torch.empty(len(P), dtype=torch.float64, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    eng.log_likelihood_dev(len(P), dP.data_ptr(), out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want_ll)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.log_likelihood_dev(len(P), dP.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream,
                               posterior=True)
        doubled = out * 2.  # consumer on the same stream sees the result without any host synchronisation
    side.synchronize()
    assert np.array_equal(out.cpu().numpy(), want_lp)
    assert np.array_equal(doubled.cpu().numpy(), 2. * want_lp)


def test_shockcooling3_distance_and_reddening():
    """ShockCooling3 (models.py:433-496): flux = c4 Lnu(E(B-V)) / d_L^2 with the band-table weights reddened per
    walker in LDS.  'a' vectors: the reference's own arithmetic (E(B-V) = 0); 'b': the reference's code around the
    restated Fitzpatrick-99 law; plus the oracle on a block with per-walker E(B-V), negative and zero values."""
    g = golden('shockcooling3')
    t, names, P = g['sc3/t'], [str(n) for n in g['sc3/names']], g['sc3/P']
    m = M.ShockCooling3(redshift=0.012)
    lc = {'MJD': t, 'filter': names, 'flux': g['sc3/flux'], 'dflux': g['sc3/dflux']}
    P0 = P.copy()
    P0[:, 5] = 0.
    e = m.engine_for(lc)
    assert relerr(e.evaluate(P0), g['sc3/a_y']) < TOL
    assert relerr(e.evaluate(P), g['sc3/b_y']) < TOL
    assert relerr(m.log_likelihood(lc, P0), g['sc3/a_lnl']) < TOL
    assert relerr(m.log_likelihood(lc, P), g['sc3/b_lnl']) < TOL
    assert relerr(m.log_likelihood(lc, g['sc3/Ps'], use_sigma=True), g['sc3/b_lnl_sigma']) < TOL
    with pytest.raises(Exception):
        e.set_variant(2)  # reddened weights: no compressed tables
    # scalar call (pointwise), (T, R) and the reference's dense branch for parameter arrays
    assert relerr(m(t, names, *P[0]), g['sc3/b_y'][0]) < TOL
    T, R = m.temperature_radius(t, *np.array([1.1, 0.6, 2.5, 1.8, 0.05]))
    assert relerr([T, R], g['sc3/b_TR']) < TOL
    block = m(t, names, *P.T)
    assert block.shape == (len(t), len(t), len(P)) and relerr(block[::15], g['sc3/b_y_block']) < TOL
    # reddened blackbody_to_filters (one E(B-V) per call): pointwise and dense
    Tp, Rp = np.linspace(4., 40., 6), np.linspace(1., 3., 6)
    ob = [O.band(n) for n in 'UBVgri']
    got = M.blackbody_to_filters(list('UBVgri'), Tp, Rp, z=0.02, ebv=0.3)
    assert relerr(got, O.blackbody_to_filters_pointwise(ob, Tp, Rp, 0.02, ebv=0.3)) < TOL
    got = M.blackbody_to_filters(['U', 'r'], Tp, Rp, z=0.02, ebv=0.3)
    want = [[O.synthesize_blackbody(O.band(n), a, b, 0.02, ebv=0.3) for a, b in zip(Tp, Rp)] for n in 'Ur']
    assert got.shape == (2, 6) and relerr(got, want) < TOL
    # oracle on a wider block
    rng = np.random.default_rng(8)
    Q = np.array([1.1, 0.6, 2.5, 1.8, 25., 0.15, 0.05]) * (1. + 0.3 * rng.uniform(-1., 1., (64, 7)))
    Q[:, 5] = rng.uniform(-0.2, 1.5, 64)
    Q[0, 5], Q[1, 4] = 0., 1e-3
    bands = [O.band(n) for n in names]
    model = ('ShockCooling3', O.ShockCoolingOracle(z=0.012))
    assert relerr(e.evaluate(Q), O.evaluate(model, t, bands, Q.T).T) < TOL
    assert relerr(e.log_likelihood(Q), O.log_likelihood(model, t, bands, lc['flux'], lc['dflux'], Q.T)) < TOL
    # a short fit through the sampler: 7 parameters, chain equals the oracle-driven stretch move
    from lightcurve_fitting_amd.sampler import EnsembleSampler
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(1., 100.), M.UniformPrior(0., 1.), M.UniformPrior(-1., 0.5)]
    eng = m.engine_for(lc, priors=priors)
    x0 = np.array([1.1, 0.6, 2.5, 1.8, 25., 0.15, 0.05]) * (1. + 0.02 * rng.standard_normal((32, 7)))
    s = EnsembleSampler(32, 7, eng, seed=5)
    s.run_mcmc(x0, 5)
    from helpers import oracle_log_posterior
    fn = oracle_log_posterior(dict(orc=None, model=model, t=t, bands=bands, y=lc['flux'], dy=lc['dflux'],
                                   priors=[p.descriptor() for p in priors]))
    ref, ref_lp, _ = O.stretch_move_run(fn, x0, 5, 5)
    assert relerr(s.get_chain(), ref) < 1e-9 and relerr(s.get_log_prob(), ref_lp) < 1e-9


def test_model_grid_at_plotting_size():
    """The second caller of the kernel: `lightcurve_model_plot` evaluates
    model(xfit[1000], ufilts, *ps[ndim, 100]) -> (nfilt, 1000, 100) (fitting.py:337-352; dense branch,
    models.py:1163-1164)."""
    import time
    rng = np.random.default_rng(12)
    m = M.ShockCooling(redshift=0.01)
    xfit = np.linspace(0.3, 12., 1000)
    ufilts = list('UBVgri')
    ps = np.array([1.2, 0.5, 3.0, 2.0, 0.1]) * (1. + 0.2 * rng.uniform(-1., 1., (100, 5)))
    got = m(xfit, ufilts, *ps.T)                       # first call builds the 6000-point evaluation engine
    t0 = time.perf_counter()
    got = m(xfit, ufilts, *ps.T)
    dt = time.perf_counter() - t0
    assert got.shape == (6, 1000, 100)
    orc = ('ShockCooling', O.ShockCoolingOracle(0.01))
    for j, f in enumerate(ufilts):
        want = O.evaluate(orc, xfit, [O.band(f)] * len(xfit), ps.T)      # (1000, 100)
        assert relerr(got[j], want) < TOL
    print(f'model grid 6 x 1000 x 100: {dt * 1e3:.2f} ms per call (host arrays in and out)')


def test_table_levels_switch_without_a_seam():
    """Dense temperature sweep across the validity thresholds of the compressed tables (hot -> cool -> full): the
    default variant against the libm evaluation of the full tables, per point, for narrow, broad and very broad bands
    at two redshifts; plus the oracle on a coarse subset."""
    from lightcurve_fitting_amd.filters import PackedTables
    names = ['U', 'B', 'g', 'r', 'DLT40', 'UVW2', 'z', 'F2100W']
    T = np.geomspace(0.25, 600., 3000)
    R = np.full_like(T, 2.)
    for z in (0., 0.4):
        bb = M.Blackbody(redshift=z)
        tabs = PackedTables(names, z=z)
        assert np.isfinite(tabs.htmin).sum() >= 4 and np.isfinite(tabs.ctmin).sum() >= 4
        eng, _ = bb._eval_engine(np.zeros(len(names)), names)
        out = {}
        for v in (3, 2, 0):   # 3: interpolated ln S(ln T) between 2 and 256 kK (here through the generic log-space state)
            eng.set_variant(v)
            out[v] = eng.evaluate(np.column_stack([T, R]))        # (3000, nfilters)
        ok = out[0] > 0
        for v in (3, 2):
            assert np.array_equal(ok, out[v] > 0)
            assert np.max(np.abs(out[v][ok] / out[0][ok] - 1.)) < 1e-11
        for j in (0, 2, 4):
            want = O.synthesize_blackbody(O.band(names[j]), T[::150], R[::150], z)
            assert relerr(out[2][::150, j], want) < TOL


@pytest.mark.parametrize('model_name', ['ShockCooling2', 'ShockCooling4', 'CompanionShocking'])
def test_interpolated_level_against_the_sample_tables(model_name):
    """The third table level (ln S(ln T) per filter, log-space thermal states) against the libm sums over the full
    tables and against the oracle: walkers whose temperatures run from far below 2 kK to far above 256 kK, so that whole
    light curves, single waves and single points fall on either side of the interpolants' range; before the explosion;
    out-of-domain parameters."""
    rng = np.random.default_rng(11)
    nw = 64
    if model_name == 'CompanionShocking':
        names8 = ['U', 'B', 'V', 'g', 'r', 'i', 'DLT40', 'unfilt.']
        epochs = 57001. + np.geomspace(0.02, 60., 160)
        t, names = np.repeat(epochs, 8), list(np.tile(names8, len(epochs)))
        lum0 = 2e20 * np.exp(-0.5 * ((t - 57018.) / 12.) ** 2)
        m = M.CompanionShocking(lc_dict(t, names, lum0, 0.05 * lum0), redshift=0.003)
        bands = [O.band(n) for n in names]
        om = ('CompanionShocking', O.CompanionShockingOracle(bands, lum0, z=0.003, variant=1))
        P = np.tile([57001., 0.5, 1.2, 57018., 1.05, 0.95, 0.9, 0.6], (nw, 1))
        P[:, 1] = np.geomspace(1e-3, 1e3, nw)          # a13: T ~ a13^(1/4) -> 4 kK ... 140 kK at one day
        P[:, 0] += rng.uniform(-0.5, 0.5, nw)
        P[5, 2] = -1.                                   # M v^7 <= 0: no shock component
    else:
        epochs = np.geomspace(0.3, 60., 200)
        t, names = np.repeat(epochs, 6), list(np.tile(list('UBVgri'), len(epochs)))
        bands = [O.band(n) for n in names]
        if model_name == 'ShockCooling2':
            m, om = M.ShockCooling2(redshift=0.01), ('ShockCooling2', O.ShockCoolingOracle(0.01))
            # (T_1 from 2 kK: at 60 d that is 0.3 kK; colder still, exp(a / T) overflows in the libm evaluation a few
            # e-folds before the engine's gradual underflow reaches zero -- 1e-280 against 0, of data of 1e20)
            P = np.column_stack([np.geomspace(2., 3000., nw), np.full(nw, 3.), np.full(nw, 20.), rng.uniform(-0.2, 0.5, nw)])
            P[3, 0], P[4, 1] = -5., -1.                 # T_1 < 0 -> no light; L_1 < 0 -> NaN
        else:
            m, om = M.ShockCooling4(redshift=0.01), ('ShockCooling4', O.ShockCooling4Oracle(0.01))
            P = np.column_stack([np.geomspace(0.02, 40., nw), np.full(nw, 0.5), np.full(nw, 3.), np.geomspace(50., 0.01, nw),
                                 rng.uniform(-0.2, 0.5, nw)])
    ytrue = 1e20 * (1 + rng.uniform(0., 1., len(t)))
    lc = lc_dict(t, names, ytrue, 0.05 * ytrue)
    eng = m.engine_for(lc)
    out = {}
    for v in (3, 0):
        eng.set_variant(v)
        out[v] = (eng.evaluate(P), eng.log_likelihood(P))
    assert relerr(out[3][0], out[0][0]) < TOL and relerr(out[3][1], out[0][1]) < TOL   # (NaN patterns included)
    some = [0, 3, 4, 5, 17, 31, 48, 63]
    want = np.array([O.evaluate(om, t, bands, P[k]) for k in some])
    assert relerr(out[3][0][some], want) < TOL
    if model_name == 'ShockCooling2':   # the temperatures really straddle the range, and a wave straddles it too
        T = np.array([O.ShockCoolingOracle(0.01).temperature_radius2(t, *P[k])[0] for k in (0, 20, 40, 63)])
        assert np.nanmin(T) < 1. and np.nanmax(T) > 1000. and np.any((T[1] > 2.) & (T[1] < 256.))
        assert np.any(T[2][:64] > 256.) and np.any(T[2][:64 * 3] < 256.)


def test_long_tables_stage_only_their_compressed_levels():
    """UVOT + optical + 2MASS + TESS + DECam: 4300 table samples do not fit in LDS, their compressed levels (< 200) do;
    points colder than every level read the full table from global memory.  Likelihoods against the oracle for every
    variant, a walker cold enough to need the full tables included, and the one-launch sampler path stays available."""
    from lightcurve_fitting_amd.engine import NativeSampler
    from lightcurve_fitting_amd.filters import PackedTables
    rng = np.random.default_rng(3)
    filts = ['UVW2', 'UVW1', 'U', 'B', 'V', 'g', 'r', 'i', 'J', 'H', 'K', 'TESS', 'z-DECam']
    tabs = PackedTables(filts, z=0.01)
    assert tabs.off[-1] > 3700 > tabs.coff[-1] + tabs.hoff[-1]
    epochs = np.sort(rng.uniform(0.5, 12., 60))
    t = np.repeat(epochs, len(filts))
    names = list(np.tile(filts, len(epochs)))
    m = M.ShockCooling(redshift=0.01)
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    orc = ('ShockCooling', O.ShockCoolingOracle(0.01))
    bands = [O.band(n) for n in names]
    ytrue = O.evaluate(orc, t, bands, truth)
    y, dy = ytrue * (1 + 0.05 * rng.standard_normal(len(t))), 0.05 * ytrue
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.4)]
    eng = m.engine_for(lc_dict(t, names, y, dy), priors=priors)
    P = truth * (1 + 0.1 * rng.standard_normal((32, 5)))
    P[0, 3], P[1, 3] = 0.01, 1e-5      # cold: below the validity of the compressed levels of the bluer bands
    Tcold = O.ShockCoolingOracle(0.01).temperature_radius(t, *P[1])[0]
    assert np.nanmin(Tcold) < 0.5
    want = O.log_likelihood(orc, t, bands, y, dy, P.T)
    for v in (2, 1, 0):
        eng.set_variant(v)
        assert relerr(eng.log_likelihood(P), want) < TOL
    eng.set_variant(2)
    a = NativeSampler(eng, 32, 4)
    assert a.one_launch
    a.set_state(P)
    a.run(0, 4, 'random', True)
    b = NativeSampler(eng, 32, 4)
    b.set_state(P)
    b.begin(0, 4, 'random', True)
    for step in range(4):
        for half in (0, 1):
            b.propose(step, half)
            b.evaluate(0, 16)
            b.accept(step, half)
    b.check()
    assert np.array_equal(a.get_chain()[0], b.get_chain()[0]) and np.array_equal(a.get_chain()[1], b.get_chain()[1])
