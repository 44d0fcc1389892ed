"""Host-side logic and the C-ABI surface.  CPU only: nothing here launches a kernel."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden, relerr
from lightcurve_fitting_amd import engine as E, filters as F, models as M, rng, sampler as S


def test_filter_registry_matches_reference():
    g = golden('filters')
    assert [f.name for f in F.all_filters] == [str(x) for x in g['filt/order']]
    alias = dict(zip([str(k) for k in g['filt/alias_keys']], [str(v) for v in g['filt/alias_vals']]))
    assert {k: v.name for k, v in F.filtdict.items()} == alias
    m0 = g['filt/M0_all']
    mine = np.array([f.M0 for f in F.all_filters])
    assert relerr(mine, m0) < 1e-15
    for n, ch in zip(g['filt/names'], g['filt/chars']):
        assert F.filtdict[str(n)].char == str(ch)


def test_filter_curves_and_packed_tables():
    g = golden('filters')
    for n in g['filt/names']:
        f = F.filtdict[str(n)]
        assert relerr(f.freq, g[f'filt/{n}/freq']) < 1e-14
        tn = g[f'filt/{n}/tnorm']
        nz = tn != 0
        assert relerr(f.T_norm_per_freq[nz], tn[nz]) < 1e-13 and np.all(f.T_norm_per_freq[~nz] == 0)
        assert relerr([f.freq_eff, f.dfreq, f.M0, f.wl_eff, f.dwl], g[f'filt/{n}/scalars']) < 1e-13
        a, w = f.planck_table(z=0.01)
        assert np.all(w > 0) and np.all(a > 0) and len(a) <= f.nsamples
    # packed (a, W) reproduces the reference band integral (computed here with plain NumPy from the tables)
    p = golden('primitives')
    names = [str(x) for x in p['synth/names']]
    tabs = F.PackedTables(names, z=0.002)
    for i, n in enumerate(names):
        a, w = tabs.a[tabs.off[i]:tabs.off[i + 1]], tabs.w[tabs.off[i]:tabs.off[i + 1]]
        mine = p['synth/R'] ** 2 * np.array([np.sum(w / np.expm1(a / t)) for t in p['synth/T']])
        assert relerr(mine, p['synth/z0.002'][i]) < 1e-12
    a, w = F.filtdict['g'].planck_table(z=0.01, cutoff_freq=300.)
    mine = p['synth/R'] ** 2 * np.array([np.sum(w / np.expm1(a / t)) for t in p['synth/T']])
    assert relerr(mine, p['synth/cutoff300_z0.01'][3]) < 1e-12
    with pytest.raises(ValueError):
        F.filtdict['L'].planck_table()
    with pytest.raises(KeyError):
        F.as_filter('no-such-filter')


def test_gauss_compressed_tables_reproduce_the_full_band_sum():
    """Every compressed table (Gauss rule of the full table's discrete measure) against the full sum, on and above its
    validity threshold, and against the reference's synthesize() numbers."""
    names = [f.name for f in F.all_filters if f.filename]
    tabs = F.PackedTables(names, z=0.002)
    n_comp = n_hot = 0
    for i, n in enumerate(names):
        a, w = tabs.a[tabs.off[i]:tabs.off[i + 1]], tabs.w[tabs.off[i]:tabs.off[i + 1]]
        longer = len(a)
        for lvl, (oo, aa, ww, tt, floor) in enumerate([(tabs.coff, tabs.ca, tabs.cw, tabs.ctmin, F.COOL_TMIN),
                                                      (tabs.hoff, tabs.ha, tabs.hw, tabs.htmin, F.HOT_TMIN)]):
            ca, cw = aa[oo[i]:oo[i + 1]], ww[oo[i]:oo[i + 1]]
            if len(ca) == 0:
                assert np.isinf(tt[i])
                continue
            n_comp += lvl == 0
            n_hot += lvl == 1
            # a level must save at least one quad of samples against the next longer table, and hold from its t_min
            assert (len(ca) + 3) // 4 < (longer + 3) // 4 and tt[i] <= floor * 1.13  # (one grid step of margin)
            assert np.all(cw > 0) and np.all(ca > 0) and np.all(np.diff(ca) < 0)
            assert np.sum(cw) == pytest.approx(np.sum(w), rel=1e-13)
            for T in np.concatenate([[max(tt[i], 0.3)], np.geomspace(max(tt[i], 0.3), 5e3, 23)]):
                full, comp = np.sum(w / np.expm1(a / T)), np.sum(cw / np.expm1(ca / T))
                assert abs(comp - full) <= 4e-14 * full, (n, lvl, T)
            longer = len(ca)
        if np.isfinite(tabs.htmin[i]) and np.isfinite(tabs.ctmin[i]):
            assert tabs.htmin[i] >= tabs.ctmin[i]  # the engine tests the hot level first
    assert n_hot >= 10
    assert n_comp >= 50
    p = golden('primitives')
    for i, n in enumerate([str(x) for x in p['synth/names']]):
        j = names.index(F.filtdict[n].name)
        ca, cw = tabs.ca[tabs.coff[j]:tabs.coff[j + 1]], tabs.cw[tabs.coff[j]:tabs.coff[j + 1]]
        ok = p['synth/T'] >= tabs.ctmin[j]
        if len(ca) and ok.any():
            mine = p['synth/R'][ok] ** 2 * np.array([np.sum(cw / np.expm1(ca / t)) for t in p['synth/T'][ok]])
            assert relerr(mine, p['synth/z0.002'][i][ok]) < 1e-12
    x, w = np.linspace(1., 2., 40), np.linspace(1., 3., 40)
    xg, wg = F.gauss_rule(x, w, 6)
    for deg in range(12):  # exact for polynomials of degree < 2m
        assert np.sum(wg * xg ** deg) == pytest.approx(np.sum(w * x ** deg), rel=1e-13)


def test_not_a_knot_spline_matches_scipy_coefficients():
    c = golden('companion')
    lum = c['csb/lum']
    names = [str(x) for x in c['csb/names']]
    lc = {'MJD': c['csb/t'], 'filter': names, 'lum': lum, 'dlum': c['csb/dlum']}
    m = M.CompanionShocking(lc, redshift=0.003)
    for i, n in enumerate(c['csb/spline_filters']):
        mine = m.sifto[F.filtdict[str(n)]]          # (n-1, 4)
        ref = np.moveaxis(c['csb/spline_c'][i], 0, 1)  # scipy: (4, n-1)
        scale = np.abs(ref).max()
        assert np.max(np.abs(mine - ref)) / scale < 1e-12
    with pytest.raises(Exception, match='No SiFTO template'):
        M.CompanionShocking({'MJD': [1.], 'filter': ['z'], 'lum': [1.], 'dlum': [1.]})


def test_priors():
    p = golden('primitives')
    xs = p['prior/x']
    assert relerr(np.array([M.UniformPrior(0., 1.)(x) for x in xs], dtype=float), p['prior/uniform_0_1']) == 0
    assert relerr(np.array([M.LogUniformPrior(0.01, 1000.)(x) for x in xs], dtype=float),
                  p['prior/loguniform_0.01_1000']) < 1e-15
    assert relerr(np.array([M.GaussianPrior(0., 10., 0., 1.)(x) for x in xs], dtype=float),
                  p['prior/gaussian_0_10_0_1']) < 1e-15
    with pytest.raises(ValueError):
        M.LogUniformPrior(-1., 1.)
    assert M.GaussianPrior(0., 10., 3., 2.).descriptor() == (2, 0., 10., 3., 2.)


def test_model_metadata():
    assert M.ShockCooling().nparams == 5 and M.ShockCooling2().nparams == 4 and M.ShockCooling4().nparams == 5
    assert M.ShockCooling(n=3.).epsilon_T == pytest.approx(2 * 0.016 - 0.5)
    assert M.ShockCooling(RW=True).a == 0. and M.ShockCooling(RW=True).Tph_to_Tcol == 1.2
    with pytest.raises(ValueError, match='n can only be 1.5 or 3'):
        M.ShockCooling(n=2.)
    sc3 = M.ShockCooling3(redshift=0.01)
    assert sc3.nparams == 7 and sc3.output_quantity == 'flux' and sc3.reddened and sc3.model_id == 3
    assert M.ShockCooling3.t_max([1., 1., 1., 2., 30., 0.1, 0.5]) == M.ShockCooling.t_max([1., 1., 1., 2., 0.5])
    assert M.ShockCooling3.t_min([1., 1., 1., 2., 30., 0.1, 0.5]) == M.ShockCooling.t_min([1., 1., 1., 2., 0.5])
    m = M.ShockCooling()
    m.input_names.append('\\sigma')
    assert M.ShockCooling().nparams == 5  # instance-level list: no cross-instance leak
    assert (M.k_B, M.c1, M.c2, M.c3, M.c4) == tuple(golden('primitives')['const/values'][[0, 3, 4, 1, 2]])
    assert M.ShockCooling.t_max([1., 1., 1., 2., 0.5]) == pytest.approx(7.4 * 2 ** 0.55 + 0.5)


def test_library_exports_every_declared_symbol():
    """Every function include/lcf.h declares is exported by the built library with the binding's prototype."""
    header = open(os.path.join(ROOT, 'include', 'lcf.h')).read()
    declared = set(re.findall(r'\b(lcf_[a-z_0-9]+)\s*\(', header))
    bound = {name for name, _, _ in E.SIGNATURES}
    assert declared == bound, declared ^ bound
    lib = E.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.lcf_abi_version() == E.LCF_ABI_VERSION
    assert ctypes.sizeof(E.LcfPrior) == 40


def test_engine_fails_loudly_without_gpu():
    lib = E.load_library()
    if lib.lcf_device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(E.LcfError, match='LCF_ERR_NO_DEVICE'):
        M.blackbody_to_filters(['g'], np.array([10.]), np.array([1.]))
    lc = {'MJD': [1., 2.], 'filter': ['g', 'r'], 'lum': [1e20, 1e20], 'dlum': [1e18, 1e18]}
    with pytest.raises(E.LcfError, match='no CPU fallback'):
        M.ShockCooling().log_likelihood(lc, np.array([1., 1., 1., 1., 0.]))


def test_engine_argument_validation_precedes_device_use():
    lib = E.load_library()
    pr = E.LcfProblem()
    h = ctypes.c_void_p()
    assert lib.lcf_engine_create(ctypes.byref(pr), 0, ctypes.byref(h)) == 1  # abi_version 0
    assert b'abi_version' in lib.lcf_last_error()
    assert lib.lcf_engine_create(None, 0, ctypes.byref(h)) == 1
    with pytest.raises(ValueError):
        E.Engine(1, 5, [], [1., 2.], [1.], [1., 1.], [0, 0], [0, 1], [1.], [1.])
    with pytest.raises(Exception, match='sigma_type'):
        M.ShockCooling().make_engine([1.], ['g'], [1.], [1.], sigma_type='bogus')
    assert lib.lcf_log_likelihood(None, 1, None, None) == 1
    assert lib.lcf_sampler_create(None, 4, 0, 2.0, ctypes.byref(h)) == 1


def test_rng_and_shards():
    assert [int(x) for x in rng.philox4x32(0, 0, 0, 0, 0, 0)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    perm = rng.split_permutations(99, 10, 5, 64)
    assert perm.shape == (5, 64) and all(sorted(r) == list(range(64)) for r in perm)
    assert not np.array_equal(perm[0], perm[1])
    assert np.array_equal(perm[2], rng.split_permutations(99, 12, 1, 64)[0])  # keyed by absolute step
    for n, w in ((512, 8), (10, 4), (3, 8), (7, 2)):
        spans = [S.shard_bounds(n, w, r) for r in range(w)]
        covered = [i for lo, hi, _ in spans for i in range(lo, hi)]
        assert covered == list(range(n)) and all(hi - lo <= width for lo, hi, width in spans)


def test_lightcurve_mcmc_argument_checks():
    from lightcurve_fitting_amd.fitting import lightcurve_mcmc
    lc = {'MJD': [1., 2.], 'filter': ['g', 'r'], 'lum': [1e20, 1e20], 'dlum': [1e18, 1e18]}
    m = M.ShockCooling2()
    with pytest.raises(Exception, match='p_up must have length 4'):
        lightcurve_mcmc(lc, m, p_lo=[1, 1, 1, 0], p_up=[2, 2, 2])
    with pytest.raises(Exception, match='priors must have length 4'):
        lightcurve_mcmc(lc, m, priors=[M.UniformPrior(0, 1)], p_lo=[1, 1, 1, 0], p_up=[2, 2, 2, 1])
    with pytest.raises(Exception, match='outside prior'):
        lightcurve_mcmc(lc, m, priors=[M.UniformPrior(0, 1.5)] * 4, p_lo=[1, 1, 1, 0], p_up=[2, 2, 2, 1])
    with pytest.raises(Exception, match='deprecated'):
        lightcurve_mcmc(lc, m, p_lo=[1, 1, 1, 0], p_up=[2, 2, 2, 1], model_kwargs={})


def _example_lc():
    from lightcurve_fitting_amd.lightcurve import LC
    g = golden('config1')
    return g, LC({'MJD': g['cfg1/MJD'], 'mag': g['cfg1/mag'], 'dmag': g['cfg1/dmag'], 'filter': g['cfg1/filter'],
                  'nondet': g['cfg1/nondet'], 'source': g['cfg1/source']})


def test_lightcurve_container_and_luminosity_prep(tmp_path):
    """Host-side data prep before the fit (reference lightcurve.py:271-359, 912-941) on the included example LC."""
    from lightcurve_fitting_amd.lightcurve import LC, flux2mag, mag2flux
    g, lc = _example_lc()
    assert len(lc) == 758 and isinstance(lc['filter'][0], F.Filter)
    lc.calcAbsMag(dm=30.79, redshift=0.002)
    lc.calcLum()
    assert relerr(lc['lum'], g['cfg1/lum']) == 0. and relerr(lc['dlum'], g['cfg1/dlum']) == 0.
    early = lc.where(MJD_min=57468., MJD_max=57485.)
    assert np.array_equal(early['MJD'], g['cfg1/MJD'][g['cfg1/early']]) and len(early) == 149
    assert len(lc.where(filter='B')) == np.sum(g['cfg1/filter'] == 'B')
    assert len(lc.where(filter=['B', 'V'], nondet=False)) == np.sum(np.isin(g['cfg1/filter'], ['B', 'V']) & ~g['cfg1/nondet'])
    assert len(lc.where(filter_not='B')) == len(lc) - len(lc.where(filter='B'))
    # KA-8
    fl, dfl = mag2flux(np.array([-17.]), np.array([0.05]), np.array([F.filtdict['g'].M0]))
    assert relerr([fl[0], dfl[0]], [2.7291427281800803e+20, 1.256816672512111e+19]) < 1e-15
    m, dm_ = flux2mag(fl, dfl, F.filtdict['g'].M0)
    assert m[0] == pytest.approx(-17.) and dm_[0] == pytest.approx(0.05)
    # per-filter extinction dictionaries are applied, or computed from E(B-V) at the effective wavelengths
    lc2 = lc.copy()
    lc2.meta.pop('extinction')
    lc2.calcAbsMag(dm=30.79, extinction={'B': 0.1})
    isB = np.array([f.name == 'B' for f in lc2['filter']])
    assert np.allclose(lc2['absmag'][isB], lc['absmag'][isB] - 0.1) and np.allclose(lc2['absmag'][~isB], lc['absmag'][~isB])
    ext = dict(zip('UBVgri', golden('shockcooling3')['sc3/filter_ext']))  # reference Filter.extinction(0.1, 3.1)
    lc3 = LC({'MJD': [1., 2., 3.], 'mag': [20., 21., 22.], 'dmag': [0.1] * 3, 'filter': ['g', 'U', 'unknown']})
    lc3.calcAbsMag(dm=30., ebv=0.1)
    assert relerr(lc3['absmag'], [20. - 30. - ext['g'], 21. - 30. - ext['U'], 22. - 30.]) < 1e-14
    assert set(lc3.meta['extinction']) == {'g', 'U'} and lc3.meta['hostext'] == {}
    lc4 = LC({'MJD': [1.], 'mag': [20.], 'dmag': [0.1], 'filter': ['g']}, meta={'z': 0.3})
    lc4.calcAbsMag(dm=30., host_ebv=0.1, host_rv=2.5)
    assert lc4['absmag'][0] == pytest.approx(-10. - F.filtdict['g'].extinction(0.1, 2.5, 0.3), rel=1e-15)
    assert F.filtdict['g'].extinction(0.1, 2.5, 0.3) > F.filtdict['g'].extinction(0.1, 2.5, 0.)  # bluer in the host frame
    # ASCII round trip in the example file's two-line fixed-width layout; Swift U/B/V remapping
    path = tmp_path / 'lc.txt'
    path.write_text('   MJD    mag  dmag filter telescope nondet\n------ ------ ----- ------ --------- ------\n'
                    '57000.5  17.5  0.10      U     Swift  False\n57001.5  18.0  0.20      r       LCO   True\n'
                    '57002.5     --  0.20      ?       LCO  False\n')
    t = LC.read(str(path))
    assert t.colnames[:3] == ['MJD', 'mag', 'dmag'] and t['filter'][0].name == 'U_S' and t['filter'][1].name == 'r'
    assert t['filter'][2].name == 'unknown' and t['mag'][2] == 0. and list(t['nondet']) == [False, True, False]


def test_fitzpatrick99_law():
    """Product-side restatement of the Fitzpatrick (1999) law (lightcurve_fitting_amd/extinction.py): the README
    example of the third-party package, agreement with the oracle's independent SciPy-spline version, the per-filter
    extinction the reference computes at the effective wavelength, and the per-sample table the engine uses."""
    from lightcurve_fitting_amd import extinction as X
    from lightcurve_fitting_amd.filters import PackedTables, filtdict
    from oracle import lcf_oracle as O
    got = X.fitzpatrick99(np.array([2000., 4000., 8000.]), 1.0, 3.1)
    assert np.array_equal(np.round(got, 8), [2.76225609, 1.42325373, 0.55333671])
    wave = np.geomspace(912., 6e4, 500)
    for rv in (2.3, 3.1, 5.0):
        assert relerr(X.fitzpatrick99(wave, 0.7, rv), O.fitzpatrick99(wave, 0.7, rv)) < 1e-13
    want = golden('shockcooling3')['sc3/filter_ext']  # reference Filter.extinction(0.1, 3.1) around the same law
    assert relerr([filtdict[f].extinction(0.1, 3.1) for f in 'UBVgri'], want) < 1e-12
    assert filtdict['unknown'].extinction(0.1) is None
    freq = np.array([300., 600., 1200.])
    assert relerr(X.extinction_law(freq, [0.1, 0.2]), O.extinction_law(freq, np.array([0.1, 0.2]))) < 1e-13
    tabs = PackedTables(['U', 'g'], z=0.02, reddening=True, compress=False)
    assert tabs.ext.shape == tabs.a.shape
    b = O.band('g')
    nu = b.freq * 1.02
    e_all = O.fitzpatrick99(O.C_NM_THZ * 10. / nu, 3.1, 3.1)
    sl = slice(tabs.off[1], tabs.off[2])
    assert relerr(np.sort(tabs.ext[sl]), np.sort(e_all[np.isin(np.round(O.C1 * nu, 9), np.round(tabs.a[sl], 9))])) < 1e-12


def test_hot_kernels_neither_spill_nor_lose_occupancy():
    """The compiler's resource report written by the build (csrc/liblcf_hip.resources.txt): the one-launch half-step
    and the likelihood kernel, in the instantiations the benchmark and the fits use, must stay free of scratch memory
    and vector-register spills at 4 waves per SIMD -- a code change that silently costs that loses 25 % (it happened)."""
    import os
    import re
    path = os.path.join(os.path.dirname(E.__file__), 'csrc', 'liblcf_hip.resources.txt')
    if not os.path.exists(path):
        pytest.skip('no resource report next to the library (built without the Makefile)')
    text = open(path).read()
    blocks = {}
    for m in re.finditer(r'Function Name: (\S+)(.*?)(?=Function Name:|\Z)', text, re.S):
        fields = dict(re.findall(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)', m.group(2)))
        blocks[m.group(1)] = fields
    # k_solo<ND = 4..9, fast band sum, epoch-major likelihood, two / four parts, single-GPU and row-board form, generic
    # model> + its model-specialised instantiations (ShockCooling: ND 5, ShockCooling2: ND 4 -- the benchmark kernels),
    # k_fused<ND = 4..9, both thermal modes>, k_points<fast band sum, likelihood mode, tables staged, both thermal modes>
    solo = [k for k in blocks if re.search(r'k_soloILi[4-9]ELi1ELb1ELi[24]ELb[01]ELi[012]E', k)]
    special = [k for k in solo if re.search(r'ELi[12]EEEv', k)]
    hot = solo + [k for k in blocks if re.search(r'k_fusedILi[4-9]ELi1ELb[01]E', k) or
                  re.search(r'k_pointsILi1ELi0ELb1ELb[01]E', k)]
    assert len(solo) == 24 + 8 and len(special) == 8 and len(hot) == 32 + 14, sorted(blocks)[:5]
    for k in hot:
        f = blocks[k]
        assert int(f['ScratchSize']) == 0 and int(f['VGPRs Spill']) == 0, (k, f)
        # scalar registers spill into lanes of a vector register (cheap, but every spill is an instruction in a kernel
        # that is bound by instruction issue): the benchmark kernels stay under one register's worth, the generic ones
        # -- every model's arithmetic and both column fast paths behind run-time switches -- under four
        assert int(f['SGPRs Spill']) <= (64 if k in special else 256), (k, f)
        assert int(f['Occupancy']) >= 4 and int(f['VGPRs']) <= 128, (k, f)
    # k_solo_run (resident workgroups: the same half-step in a loop over up to 64 of them): what lives across the loop
    # costs the benchmark kernels a few registers -- it must stay a few (spilled registers are scratch traffic in every
    # half-step), and the occupancy that lets two workgroups share a CU must hold for every instantiation
    # (... ELb0E: one GPU, ELb1E: one rank of a row-board run)
    runs = [k for k in blocks if re.search(r'k_solo_runILi[4-9]ELi1ELb1ELi[24]ELi[012]ELb[01]E', k)]
    run_special = [k for k in runs if re.search(r'ELi[12]ELb[01]EEEv', k)]
    assert len(runs) == 12 + 4 + 4 and len(run_special) == 4 + 2, sorted(blocks)[:5]
    for k in runs:
        f = blocks[k]
        assert int(f['Occupancy']) >= 4 and int(f['VGPRs']) <= 128, (k, f)
        assert int(f['VGPRs Spill']) <= 32 and int(f['ScratchSize']) <= 128, (k, f)
        # scalar spills: the sampler is read through a pointer since round 4 (as kernel arguments it cost the benchmark
        # kernels 145); what is left are the problem's and the sampler's fields that live across the loop
        assert int(f['SGPRs Spill']) <= (128 if k in run_special else 300), (k, f)
    # population mode's one launch per half-step, in the dimensions with their own instantiation
    pops = [k for k in blocks if re.search(r'k_popILi[4568]ELi1ELi4ELi[012]E', k)]
    assert len(pops) == 4 + 2, sorted(blocks)[:5]
    for k in pops:
        f = blocks[k]
        assert int(f['ScratchSize']) == 0 and int(f['VGPRs Spill']) == 0 and int(f['Occupancy']) >= 4, (k, f)


def test_roofline_basis_is_generated_by_the_build():
    """bench.py's roofline.frac is computed from instruction counts that the BUILD took from the ISA of the library
    (csrc/liblcf_hip.isa.json, tools/isa_count.py) -- not from constants typed into bench.py: the report must exist next
    to the library, hold plausible counts, be what bench.isa_counts() returns, and bench.py must hold no per-unit
    float64 instruction constant of its own."""
    import json
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(os.path.dirname(E.__file__), 'csrc', 'liblcf_hip.isa.json')
    if not os.path.exists(path):
        pytest.skip('no ISA report next to the library (built without the Makefile)')
    doc = json.load(open(path))
    assert 60 <= doc['quad_main'] <= 100 and doc['quad_main'] <= doc['quad_safe']
    assert 18 <= doc['point_lean'] <= 40 and 40 <= doc['log_lean'] <= doc['state_lean'] <= 160
    assert 'k_solo' in doc['kernels']['lean'] and 'k_solo' in doc['kernels']['generic']
    sys.path.insert(0, root)
    import bench
    counts, note = bench.isa_counts()
    assert note is None and counts == {k: float(doc[k]) for k in counts}
    src = open(os.path.join(root, 'bench.py')).read()
    assert not re.search(r'VALU_PER_(QUAD_F64|POINT|EPOCH)\w*\s*=', src)


def test_shipped_table_levels_are_the_built_ones_and_make_packing_cheap(monkeypatch):
    """data/table_levels.npz (tools/pack_table_levels.py): the z = 0 levels of every bandpass -- Gauss-compressed tables
    and interpolants, proved when they were packed.  They are what this tree's code builds (bitwise, for a sample of
    filters: a stale file would not be), every plain table takes its levels from there at any redshift, tables with a
    cut-off are still built, and packing six filters costs milliseconds instead of 0.2 s."""
    import time
    from lightcurve_fitting_amd import filters as F
    names = ['U', 'B', 'V', 'g', 'r', 'i', 'DLT40', 'UVW2']
    fast = F.PackedTables(names, z=0.)
    assert fast.levels_from == ['shipped'] * len(names)
    monkeypatch.setenv('LCF_PACK_LEVELS', '1')
    built = F.PackedTables(names, z=0.)
    monkeypatch.delenv('LCF_PACK_LEVELS')
    assert built.levels_from == ['built'] * len(names)
    for attr in ('ca', 'cw', 'coff', 'ctmin', 'cbound', 'ha', 'hw', 'hoff', 'htmin', 'hbound', 'icoef', 'itmin', 'ibound'):
        assert np.array_equal(getattr(fast, attr), getattr(built, attr), equal_nan=True), attr
    assert fast.iu0 == built.iu0 == np.log(F.INTERP_TMIN)
    every = [f for f in F.all_filters if f.filename]
    assert F.PackedTables(every, z=0.02).levels_from == ['shipped'] * len(every)
    assert F.PackedTables(['g', 'r'], z=0.02, cutoff_freq=1500.).levels_from == ['built'] * 2
    # a redshifted table: nodes, weights and thresholds scale, the interpolants' grid moves by ln(1 + z)
    z = 0.05
    red = F.PackedTables(names, z=z)
    assert np.allclose(red.ca, fast.ca * (1 + z), rtol=1e-15) and np.allclose(red.cw, fast.cw * (1 + z) ** 3, rtol=1e-15)
    assert np.allclose(red.ctmin, fast.ctmin * (1 + z)) and red.iu0 == np.log(F.INTERP_TMIN) + np.log1p(z)
    assert np.allclose(red.icoef[..., :-1], fast.icoef[..., :-1], rtol=0, atol=0, equal_nan=True)
    assert np.allclose(red.icoef[..., -1], fast.icoef[..., -1] + 3 * np.log1p(z), rtol=0, atol=1e-15, equal_nan=True)
    F.PackedTables(list('UBVgri'), z=0.011)                      # (curves read, file open)
    t0 = time.perf_counter()
    F.PackedTables(list('UBVgri'), z=0.012)
    assert time.perf_counter() - t0 < 0.05


def test_kernel_names_follow_the_header():
    """`NativeSampler.last_run_kernel()` names every LCF_KERNEL_* value of include/lcf.h, with the header's numbers."""
    import re
    from lightcurve_fitting_amd import engine
    text = open(os.path.join(ROOT, 'include', 'lcf.h')).read()
    values = {name.lower().replace('_', '-'): int(v) for name, v in re.findall(r'LCF_KERNEL_(\w+)\s*=\s*(\d+)', text)}
    assert len(values) == 7 and values == {name: v for v, name in engine.KERNEL_NAMES.items()}
