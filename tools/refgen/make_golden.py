#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE's own hot-path code on the build host.

The reference (``/root/reference/lightcurve_fitting``) is imported unmodified, with throw-away stand-ins for the
third-party packages this container lacks (``tools/refgen/standins``: astropy units/constants/table, extinction,
emcee, corner).  The stand-ins carry no model arithmetic: they supply CODATA constants, unit conversion factors and
ASCII table reading.  Everything written to ``tests/golden/*.npz`` is plain data: inputs and the reference's
outputs.  The reference itself never travels to the GPU box.

Usage:  python tools/refgen/make_golden.py [--ref /root/reference]
"""
import argparse
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, '..', '..'))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def import_reference(ref_root):
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(HERE, 'standins'))
    sys.path.insert(1, ref_root)
    warnings.simplefilter('ignore')
    from lightcurve_fitting import models, filters, bolometric, lightcurve  # noqa
    from astropy.table import Table
    return models, filters, bolometric, lightcurve, Table


def make_lc(Table, filters, t, names, lum, dlum, **meta):
    """Duck LC: the reference only needs columns, ``.meta``, ``.colnames`` and ``.where(filter=...)``."""

    class DuckLC(Table):
        def where(self, filter=None):
            f = filters.filtdict[filter] if isinstance(filter, str) else filter
            return self[np.array([x == f for x in self['filter']])]

    lc = DuckLC()
    lc['MJD'] = np.asarray(t, dtype=float)
    fobj = np.empty(len(names), dtype=object)
    fobj[:] = [filters.filtdict[n] for n in names]
    lc['filter'] = fobj
    lc['lum'] = np.asarray(lum, dtype=float)
    lc['dlum'] = np.asarray(dlum, dtype=float)
    lc.meta.update(meta)
    return lc


def gen_filters(filters, out):
    names = []
    for f in filters.all_filters:
        if not f.filename:
            continue
        names.append(f.name)
        out[f'filt/{f.name}/freq'] = np.array(f.trans['freq'].value)
        out[f'filt/{f.name}/tnorm'] = np.array(f.trans['T_norm_per_freq'].data)
        out[f'filt/{f.name}/scalars'] = np.array([f.freq_eff.value, f.dfreq.value, f.M0,
                                                  f.wl_eff.value, f.dwl.value])
    out['filt/names'] = np.array(names)
    out['filt/chars'] = np.array([filters.filtdict[n].char for n in names])
    # full alias map
    out['filt/alias_keys'] = np.array(list(filters.filtdict.keys()))
    out['filt/alias_vals'] = np.array([f.name for f in filters.filtdict.values()])
    out['filt/order'] = np.array(filters.Filter.order)
    out['filt/M0_all'] = np.array([f.M0 for f in filters.all_filters])


def gen_constants(models, filters, bolometric, out):
    out['const/values'] = np.array([models.k_B, models.c3, models.c4, models.c1, models.c2, filters.c,
                                    bolometric.sigma_sb])


def gen_planck(models, filters, out):
    rng = np.random.default_rng(1)
    nu = np.array([500., 1000.])
    out['planck/ka1'] = models.planck_fast(nu, 10., 1.)
    T = 10 ** rng.uniform(-0.3, 2.3, 40)
    R = 10 ** rng.uniform(-2, 2, 40)
    names = ['U', 'B', 'V', 'g', 'r', 'i', 'UVW2', 'unfilt.', 'DLT40', 'F2550W', 'F070W', 'NUV', 'z', 'Kepler', 'w']
    out['synth/names'] = np.array(names)
    out['synth/T'] = T
    out['synth/R'] = R
    for z in (0., 0.002, 0.5):
        out[f'synth/z{z}'] = np.array([[filters.filtdict[n].synthesize(models.planck_fast, t, r, np.inf, z=z, ebv=0.)
                                        for t, r in zip(T, R)] for n in names])
    out['synth/cutoff300_z0.01'] = np.array([[filters.filtdict[n].synthesize(models.planck_fast, t, r, 300., z=0.01,
                                                                           ebv=0.) for t, r in zip(T, R)]
                                             for n in names[:6]])
    # extreme temperatures (overflow of exp -> 0; T<=0 -> 0)
    Tx = np.array([1e-3, 0.02, 0.1, 1e3, 1e5, 0., -1.])
    out['synth/extreme_T'] = Tx
    out['synth/extreme'] = np.array([[filters.filtdict[n].synthesize(models.planck_fast, np.float64(t), np.float64(2.),
                                                                   np.inf, z=0., ebv=0.) for t in Tx]
                                     for n in names[:6]])


def sc_params(rng, n, lo=(0.3, 0.1, 0.5, 0.3, -0.3), hi=(3., 2., 8., 5., 0.4)):
    return rng.uniform(lo, hi, (n, 5))


def gen_shockcooling(models, filters, Table, out):
    rng = np.random.default_rng(2)
    p_ka = (1.2, 0.5, 3.0, 2.0, 0.1)
    out['sc/ka3_T'], out['sc/ka3_R'] = models.ShockCooling(redshift=0.).temperature_radius(
        np.array([1., 2., 5.]), *p_ka)
    t = np.array([1., 1, 2, 2, 3, 3, 4, 4])
    names = ['g', 'r'] * 4
    f = [filters.filtdict[n] for n in names]
    out['sc/ka4_y'] = models.ShockCooling(redshift=0.)(t, f, *p_ka)
    out['sc/ka4_n3'] = models.ShockCooling(redshift=0., n=3.)(t, f, *p_ka)
    out['sc/ka4_rw'] = models.ShockCooling(redshift=0., RW=True)(t, f, *p_ka)
    out['sc/ka4_z'] = models.ShockCooling(redshift=0.01)(t, f, *p_ka)
    out['sc/ka4_sc4'] = models.ShockCooling4(redshift=0.)(t, f, *p_ka)
    out['sc/ka4_sc2'] = models.ShockCooling2(redshift=0.002)(np.array([1., 2, 3]),
                                                           [filters.filtdict[n] for n in 'UBV'], 30., 3., 30., 0.2)
    # KA-5 likelihoods
    y = out['sc/ka4_y']
    s = np.array([1., -1] * 4)
    lc = make_lc(Table, filters, t, names, y * (1 + 0.05 * s), 0.05 * y)
    m = models.ShockCooling(redshift=0.)
    out['sc/ka5'] = np.array([
        m.log_likelihood(lc, np.array(p_ka)),
        m.log_likelihood(lc, np.array([1.0, 0.7, 2.5, 2.5, 0.2])),
        m.log_likelihood(lc, np.array([1.0, 0.7, 2.5, 2.5, 2.5])),
        m.log_likelihood(lc, np.array(p_ka + (0.5,)), use_sigma=True, sigma_type='relative'),
        m.log_likelihood(lc, np.array(p_ka + (0.5,)), use_sigma=True, sigma_type='absolute')])

    # randomised blocks: ragged multi-filter light curve, every variant
    npts = 137
    names = rng.choice(['U', 'B', 'V', 'g', 'r', 'i', 'UVW2', 'z', 'DLT40', 'unfilt.'], npts)
    t = np.sort(rng.uniform(0.2, 12., npts))
    fobj = [filters.filtdict[n] for n in names]
    out['scb/t'] = t
    out['scb/names'] = names
    P = sc_params(rng, 24)
    P[-4:, 4] = rng.uniform(1., 6., 4)  # some explosion times inside the data (negative phases -> y_fit = 0)
    out['scb/P'] = P
    truth = models.ShockCooling(redshift=0.01)(t, fobj, *p_ka)
    yobs = truth * (1 + 0.05 * rng.standard_normal(npts))
    dy = 0.05 * truth * rng.uniform(0.5, 2., npts)
    out['scb/y'] = yobs
    out['scb/dy'] = dy
    lc = make_lc(Table, filters, t, names, yobs, dy)
    variants = {'n15': dict(n=1.5), 'n3': dict(n=3.), 'rw': dict(RW=True), 'n3rw': dict(n=3., RW=True)}
    for tag, kw in variants.items():
        m = models.ShockCooling(redshift=0.01, **kw)
        TR = [m.temperature_radius(t, *p) for p in P]
        out[f'scb/{tag}/T'] = np.array([x[0] for x in TR])
        out[f'scb/{tag}/R'] = np.array([x[1] for x in TR])
        out[f'scb/{tag}/y'] = np.array([m(t, fobj, *p) for p in P])
        out[f'scb/{tag}/ll'] = np.array([m.log_likelihood(lc, p) for p in P])
    m = models.ShockCooling(redshift=0.01)
    sig = rng.uniform(0.1, 3., len(P))
    out['scb/sigma'] = sig
    out['scb/n15/ll_rel'] = np.array([m.log_likelihood(lc, np.append(p, s_), use_sigma=True) for p, s_ in zip(P, sig)])
    out['scb/n15/ll_abs'] = np.array([m.log_likelihood(lc, np.append(p, s_), use_sigma=True, sigma_type='absolute')
                                      for p, s_ in zip(P, sig)])
    # ShockCooling2
    P2 = rng.uniform((5., 0.3, 3., -0.3), (60., 10., 40., 0.4), (24, 4))
    P2[-4:, 3] = rng.uniform(1., 6., 4)
    out['scb/P2'] = P2
    for tag, kw in variants.items():
        m2 = models.ShockCooling2(redshift=0.01, **kw)
        out[f'scb/{tag}/y2'] = np.array([m2(t, fobj, *p) for p in P2])
        out[f'scb/{tag}/ll2'] = np.array([m2.log_likelihood(lc, p) for p in P2])
    # ShockCooling4
    m4 = models.ShockCooling4(redshift=0.01)
    TR = [m4.temperature_radius(t, *p) for p in P]
    out['scb/sc4/T'] = np.array([x[0] for x in TR])
    out['scb/sc4/R'] = np.array([x[1] for x in TR])
    out['scb/sc4/y'] = np.array([m4(t, fobj, *p) for p in P])
    out['scb/sc4/ll'] = np.array([m4.log_likelihood(lc, p) for p in P])

    # out-of-domain parameters: power() semantics (zeros, NaNs) -- np.float64 inputs as emcee passes them
    edge = np.array([
        [-1., 1., 1., 1., 0.],  # v_s < 0
        [1., -1., 1., 1., 0.],  # M_env < 0 (t_tr NaN -> no cut-off)
        [1., 1., -1., 1., 0.],  # f_rho_M < 0
        [1., 1., 1., -1., 0.],  # R < 0 -> NaN
        [1., 1., 1., 1., 100.],  # all phases negative
        [0., 1., 1., 1., 0.],  # v_s = 0
        [1., 0., 1., 1., 0.],  # M_env = 0
        [1., 1., 0., 1., 0.],  # f_rho_M = 0
        [1., 1., 1., 0., 0.],  # R = 0
        [-1., -1., 1., 1., 0.],
        [-1., 1., -1., 1., 0.],
        [1., 1., 1., 1., float(t[5])],  # a phase of exactly zero
    ])
    out['sce/P'] = edge
    for tag, mdl in (('sc', models.ShockCooling(redshift=0.01)), ('sc4', models.ShockCooling4(redshift=0.01))):
        out[f'sce/{tag}/y'] = np.array([mdl(t, fobj, *[np.float64(x) for x in p]) for p in edge])
        out[f'sce/{tag}/ll'] = np.array([mdl.log_likelihood(lc, p) for p in edge])
    edge2 = np.array([[-5., 3., 30., 0.2], [30., -3., 30., 0.2], [30., 3., -30., 0.2], [30., 3., 30., 100.],
                      [0., 3., 30., 0.2], [30., 0., 30., 0.2], [30., 3., 0., 0.2]])
    out['sce/P2'] = edge2
    m2 = models.ShockCooling2(redshift=0.01)
    out['sce/sc2/y'] = np.array([m2(t, fobj, *[np.float64(x) for x in p]) for p in edge2])
    out['sce/sc2/ll'] = np.array([m2.log_likelihood(lc, p) for p in edge2])


def gen_shockcooling3(models, filters, Table, out):
    """ShockCooling3 (distance + reddening free, fits 'flux').  'sc3/a_*': E(B-V) = 0, the reference's own arithmetic
    only (A_lambda = 0 whatever the law).  'sc3/b_*': E(B-V) != 0 -- the reference's code path (frames, R_V factor,
    broadcasting: filters.py:14-33, 308-310) around the stand-in's restatement of the Fitzpatrick (1999) law, i.e.
    these vectors pin the plumbing, not the law itself (see standins/extinction)."""
    rng = np.random.default_rng(33)
    npts = 90
    names = rng.choice(['U', 'B', 'V', 'g', 'r', 'i', 'UVW2', 'z'], npts)
    t = np.sort(rng.uniform(0.3, 10., npts))
    f = [filters.filtdict[n] for n in names]
    out['sc3/t'], out['sc3/names'] = t, np.array(names)
    m = models.ShockCooling3(redshift=0.012)
    truth = np.array([1.1, 0.6, 2.5, 1.8, 25., 0.15, 0.05])
    n = 10
    P = truth * (1. + 0.2 * rng.uniform(-1., 1., (n, 7)))
    P[:, 6] = rng.uniform(-0.2, 0.25, n)
    P[1, 4] = -P[1, 4]   # negative distance: dist ** 2 is positive
    P[2, 0] = -0.5       # v_s < 0: power() zeroing
    out['sc3/P'] = P
    P0 = P.copy()
    P0[:, 5] = 0.
    out['sc3/a_y'] = np.array([m(t, f, *p) for p in P0])                   # (n, npts), per-walker scalar calls
    out['sc3/b_y'] = np.array([m(t, f, *p) for p in P])
    # parameter arrays make T two-dimensional, so the reference takes the dense branch (models.py:1161-1164):
    # (nfilters, ntimes, nwalkers) even though len(t) == len(f); every 15th filter row is kept
    block = m(t, f, *[P[:, j] for j in range(7)])
    assert block.shape == (npts, npts, n)
    out['sc3/b_y_block'] = block[::15]
    yflux = m(t, f, *truth)
    sgn = np.where(rng.uniform(size=npts) < 0.5, -1., 1.)
    lc = make_lc(Table, filters, t, names, np.ones(npts), np.ones(npts))
    lc['flux'] = yflux * (1. + 0.04 * sgn)
    lc['dflux'] = 0.04 * yflux
    out['sc3/flux'], out['sc3/dflux'] = np.asarray(lc['flux']), np.asarray(lc['dflux'])
    out['sc3/a_lnl'] = np.array([m.log_likelihood(lc, p) for p in P0])
    out['sc3/b_lnl'] = np.array([m.log_likelihood(lc, p) for p in P])
    Ps = np.column_stack([P, rng.uniform(0.1, 2., n)])
    out['sc3/Ps'] = Ps
    out['sc3/b_lnl_sigma'] = np.array([m.log_likelihood(lc, p, use_sigma=True, sigma_type='relative') for p in Ps])
    out['sc3/b_TR'] = np.array(m.temperature_radius(t, *truth[[0, 1, 2, 3, 6]]))
    out['sc3/filter_ext'] = np.array([filters.filtdict[x].extinction(0.1, 3.1, z=0.) for x in 'UBVgri'])


def gen_config2(models, filters, Table, out, nwalkers=48):
    """BASELINE.json configs[1] photometry and reference log-likelihoods for a walker block."""
    rng = np.random.default_rng(20241024)
    epochs = np.sort(rng.uniform(0.5, 10., 500))
    bands = ['U', 'B', 'V', 'g', 'r', 'i']
    t = np.repeat(epochs, len(bands))
    names = np.tile(bands, len(epochs))
    fobj = [filters.filtdict[n] for n in names]
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    m = models.ShockCooling(redshift=0., n=1.5)
    ytrue = m(t, fobj, *truth)
    y = ytrue * (1. + 0.05 * rng.standard_normal(len(t)))
    dy = 0.05 * ytrue
    lo = truth * 0.8
    hi = truth * 1.2
    lo[4], hi[4] = 0., 0.2
    P = rng.uniform(lo, hi, (nwalkers, 5))
    lc = make_lc(Table, filters, t, names, y, dy)
    out['cfg2/t'] = t
    out['cfg2/names'] = names
    out['cfg2/y'] = y
    out['cfg2/dy'] = dy
    out['cfg2/P'] = P
    out['cfg2/ll'] = np.array([m.log_likelihood(lc, p) for p in P])
    out['cfg2/yfit0'] = m(t, fobj, *P[0])


def gen_companion(models, filters, Table, out):
    rng = np.random.default_rng(3)
    # KA-6
    out['cs/ka6_T'], out['cs/ka6_R'] = models.BaseCompanionShocking.temperature_radius(
        np.array([2., 5., 20.]), 1., 0.5, 1.2)
    # KA-7
    t = np.repeat([57003., 57010, 57020, 57040], 6)
    names = ['U', 'B', 'V', 'g', 'r', 'i'] * 4
    lum = 1e20 * (1 + 0.1 * np.arange(24))
    lc = make_lc(Table, filters, t, names, lum, 0.05 * lum)
    q = (57001., 0.5, 1.2, 57018., 1.05, 0.95, 0.9, 0.6)
    m = models.CompanionShocking(lc, redshift=0.003)
    fobj = [filters.filtdict[n] for n in names]
    out['cs/ka7_y'] = m(t, fobj, *q)
    out['cs/ka7_ll'] = np.array(m.log_likelihood(lc, np.array(q)))

    # randomised: 8 supported filters incl. DLT40 / unfilt.
    npts = 160
    names = rng.choice(['U', 'B', 'V', 'g', 'r', 'i', 'DLT40', 'unfilt.'], npts)
    t = np.sort(rng.uniform(57001., 57110., npts))
    fobj = [filters.filtdict[n] for n in names]
    lum = 1e20 * rng.uniform(0.5, 3., npts)
    dlum = 0.05 * lum * rng.uniform(0.5, 2., npts)
    lc = make_lc(Table, filters, t, names, lum, dlum)
    out['csb/t'] = t
    out['csb/names'] = names
    out['csb/lum'] = lum
    out['csb/dlum'] = dlum
    m1 = models.CompanionShocking(lc, redshift=0.003)
    m2 = models.CompanionShocking2(lc, redshift=0.003)
    m3 = models.CompanionShocking3(lc, redshift=0.003)
    # spline samples + scale factors (one per filter in the light curve)
    xs = np.linspace(-20., 90., 441)
    out['csb/spline_x'] = xs
    ufilts = sorted(set(names))
    out['csb/spline_filters'] = np.array(ufilts)
    out['csb/spline_y'] = np.array([m1.sifto[filters.filtdict[n]](xs) for n in ufilts])
    out['csb/spline_c'] = np.array([m1.sifto[filters.filtdict[n]].c for n in ufilts])
    P1 = rng.uniform((56995., 0.05, 0.2, 57012., 0.8, 0.7, 0.7, 0.3), (57003., 2., 3., 57024., 1.3, 1.3, 1.3, 1.5),
                     (20, 8))
    P1[-3:, 0] = rng.uniform(57010., 57050., 3)  # explosion inside the data
    out['csb/P1'] = P1
    out['csb/y1'] = np.array([m1(t, fobj, *p) for p in P1])
    out['csb/ll1'] = np.array([m1.log_likelihood(lc, p) for p in P1])
    P2 = np.column_stack([P1[:, :5], rng.uniform(-3., 3., (20, 2))])
    out['csb/P2'] = P2
    out['csb/y2'] = np.array([m2(t, fobj, *p) for p in P2])
    out['csb/ll2'] = np.array([m2.log_likelihood(lc, p) for p in P2])
    P3 = P2.copy()
    P3[:, 2] = rng.uniform(0., 180., 20)
    out['csb/P3'] = P3
    out['csb/y3'] = np.array([m3(t, fobj, *p) for p in P3])
    out['csb/ll3'] = np.array([m3.log_likelihood(lc, p) for p in P3])
    out['csb/ll1_sigma_abs'] = np.array([m1.log_likelihood(lc, np.append(p, 0.7), use_sigma=True,
                                                           sigma_type='absolute') for p in P1])
    edge = np.array([
        [57001., -0.5, 1.2, 57018., 1.05, 0.95, 0.9, 0.6],  # a13 < 0: a13**36 > 0 still
        [57001., 0.5, -1.2, 57018., 1.05, 0.95, 0.9, 0.6],  # Mv < 0 -> T = R = 0
        [57001., 0.5, 1.2, 57018., -1.05, 0.95, 0.9, 0.6],  # negative stretch
        [57200., 0.5, 1.2, 57018., 1.05, 0.95, 0.9, 0.6],  # all phases negative
        [57001., 0.5, 1.2, 57300., 1.05, 0.95, 0.9, 0.6],  # template entirely after the data
        [57001., 0., 1.2, 57018., 1.05, 0.95, 0.9, 0.6],  # a13 = 0
    ])
    out['cse/P1'] = edge
    out['cse/y1'] = np.array([m1(t, fobj, *[np.float64(x) for x in p]) for p in edge])
    out['cse/ll1'] = np.array([m1.log_likelihood(lc, p) for p in edge])


def gen_config3(models, filters, Table, out, nwalkers=12):
    """BASELINE.json configs[2] photometry shape, smaller walker block."""
    rng = np.random.default_rng(20241025)
    bands = ['U', 'B', 'V', 'g', 'r', 'i', 'DLT40', 'unfilt.']
    epochs = np.sort(rng.uniform(57001., 57060., 1000))
    t = np.repeat(epochs, len(bands))
    names = np.tile(bands, len(epochs))
    fobj = [filters.filtdict[n] for n in names]
    q = np.array([57001., 0.5, 1.2, 57018., 1.05, 0.95, 0.9, 0.6])
    # a plausible light curve to scale the template by: per-filter peak luminosities
    peak = {'U': 2.1e20, 'B': 2.6e20, 'V': 2.4e20, 'g': 2.5e20, 'r': 2.2e20, 'i': 1.7e20, 'DLT40': 2.2e20,
            'unfilt.': 2.2e20}
    lum0 = np.array([peak[n] for n in names]) * np.exp(-0.5 * ((t - 57018.) / 12.) ** 2)
    lc0 = make_lc(Table, filters, t, names, lum0, 0.05 * lum0)
    m = models.CompanionShocking(lc0, redshift=0.003)
    ytrue = m(t, fobj, *q)
    y = ytrue * (1. + 0.05 * rng.standard_normal(len(t)))
    dy = 0.05 * np.maximum(ytrue, 1e17)
    lc = make_lc(Table, filters, t, names, y, dy)
    m = models.CompanionShocking(lc, redshift=0.003)  # template scaled to the noisy data, as a user would
    P = q * (1 + 0.02 * rng.standard_normal((nwalkers, 8)))
    P[:, 0] = q[0] + 0.5 * rng.standard_normal(nwalkers)
    P[:, 3] = q[3] + 0.5 * rng.standard_normal(nwalkers)
    out['cfg3/t'] = t
    out['cfg3/names'] = names
    out['cfg3/y'] = y
    out['cfg3/dy'] = dy
    out['cfg3/P'] = P
    out['cfg3/ll'] = np.array([m.log_likelihood(lc, p) for p in P])


def gen_config1(models, filters, lightcurve, Table, out, ref_root):
    """BASELINE.json configs[0]: the included example light curve (SN 2016bkv), docs/source/usage.rst:174-200.
    No Milky-Way extinction is applied (the Fitzpatrick-99 package is absent): stated deviation, SURVEY section 8d."""
    path = os.path.join(ref_root, 'lightcurve_fitting', 'example', 'SN2016bkv.txt')
    rows = [ln.split() for ln in open(path) if ln.strip() and set(ln.strip()) - set('- ')]
    header, rows = rows[0], rows[1:]
    col = {h: [r[i] for r in rows] for i, h in enumerate(header)}
    mjd = np.array(col['MJD'], float)
    mag = np.array(col['mag'], float)
    dmag = np.array(col['dmag'], float)
    nondet = np.array([v == 'True' for v in col['nondet']])
    names = np.array(col['filter'])
    out['cfg1/MJD'], out['cfg1/mag'], out['cfg1/dmag'] = mjd, mag, dmag
    out['cfg1/filter'], out['cfg1/nondet'], out['cfg1/source'] = names, nondet, np.array(col['source'])
    dm, z = 30.79, 0.002
    fobj = [filters.filtdict[n] for n in names]
    zp = np.array([f.m0 for f in fobj]) + 90.19
    lum, dlum = lightcurve.mag2flux(mag - dm, dmag, zp, nondet, 3.)
    out['cfg1/lum'], out['cfg1/dlum'] = lum, dlum
    rng = np.random.default_rng(1)
    # 1a: docs example verbatim -- ShockCooling2 on the early light curve
    early = (mjd >= 57468.) & (mjd <= 57485.)
    out['cfg1/early'] = early
    lc = make_lc(Table, filters, mjd[early], names[early], lum[early], dlum[early], redshift=z)
    m = models.ShockCooling2(lc)
    assert m.z == z
    P = rng.uniform([20., 2., 20., 57468.5], [50., 5., 50., 57468.7], (64, 4))
    out['cfg1/P1a'] = P
    out['cfg1/ll1a'] = np.array([m.log_likelihood(lc, p) for p in P])
    # 1b: ShockCooling (5 parameters) on the first 200 rows (by MJD) among filters B, V, i
    sel = np.nonzero(np.isin(names, ['B', 'V', 'i']))[0]
    sel = sel[np.argsort(mjd[sel], kind='stable')][:200]
    out['cfg1/sel1b'] = sel
    lc = make_lc(Table, filters, mjd[sel], names[sel], lum[sel], dlum[sel], redshift=z)
    m = models.ShockCooling(lc)
    P = rng.uniform([0.5, 0.2, 1., 0.5, 57467.], [2., 2., 5., 4., 57468.6], (64, 5))
    out['cfg1/P1b'] = P
    out['cfg1/ll1b'] = np.array([m.log_likelihood(lc, p) for p in P])


def gen_sed(models, filters, out):
    """Per-epoch SED likelihoods as spectrum_mcmc's inner log_posterior computes them (bolometric.py:139-164):
    the reference's own Filter.synthesize(planck_fast, T, R) per filter, then the Gaussian likelihood."""
    rng = np.random.default_rng(6)
    pool = ['U', 'B', 'V', 'g', 'r', 'i', 'UVW2', 'UVM2', 'UVW1', 'z']
    z = 0.01
    ep_off, names, ys, dys, cands, ll, ll_rel, ll_abs = [0], [], [], [], [], [], [], []
    for e in range(12):
        nf = rng.integers(2, 9)
        fl = list(rng.choice(pool, nf, replace=False))
        Tt, Rt = rng.uniform(5., 40.), 10 ** rng.uniform(-0.5, 1.)
        fobj = [filters.filtdict[n] for n in fl]
        ytrue = np.array([f.synthesize(models.planck_fast, Tt, Rt, z=z, ebv=0.) for f in fobj])
        y = ytrue * (1 + 0.05 * rng.standard_normal(nf))
        dy = 0.05 * ytrue * rng.uniform(0.5, 2., nf)
        c = np.column_stack([Tt * rng.uniform(0.5, 2., 24), Rt * rng.uniform(0.5, 2., 24), rng.uniform(0., 3., 24)])
        for (T, R, sg) in c:
            y_fit = np.array([f.synthesize(models.planck_fast, T, R, z=z, ebv=0.) for f in fobj])
            for sigma, dest in ((dy, ll), (np.sqrt(dy ** 2. + (sg * dy) ** 2.), ll_rel),
                                (np.sqrt(dy ** 2. + (sg * np.median(dy)) ** 2.), ll_abs)):
                dest.append(-0.5 * np.sum(np.log(2 * np.pi * sigma ** 2.) + ((y - y_fit) / sigma) ** 2.))
        ep_off.append(ep_off[-1] + nf)
        names += fl
        ys += list(y)
        dys += list(dy)
        cands.append(c)
    out['sed/z'] = np.array(z)
    out['sed/ep_off'] = np.array(ep_off)
    out['sed/names'] = np.array(names)
    out['sed/y'], out['sed/dy'] = np.array(ys), np.array(dys)
    out['sed/cand'] = np.array(cands)
    out['sed/ll'] = np.array(ll).reshape(12, 24)
    out['sed/ll_rel'] = np.array(ll_rel).reshape(12, 24)
    out['sed/ll_abs'] = np.array(ll_abs).reshape(12, 24)


def gen_priors_misc(models, filters, bolometric, lightcurve, out):
    u = models.UniformPrior(0., 1.)
    lu = models.LogUniformPrior(0.01, 1000.)
    g = models.GaussianPrior(0., 10., 0., 1.)
    xs = np.array([0.5, 1.0, 0., -0.1, 2., 0.01, 1000., 999.9, 10., 3.3])
    out['prior/x'] = xs
    out['prior/uniform_0_1'] = np.array([u(x) for x in xs], dtype=float)
    out['prior/loguniform_0.01_1000'] = np.array([lu(x) for x in xs], dtype=float)
    out['prior/gaussian_0_10_0_1'] = np.array([g(x) for x in xs], dtype=float)
    out['misc/pseudo_10_1_0'] = np.array(bolometric.pseudo(10., 1., 0.))
    out['misc/stefan_boltzmann_10_1'] = np.array(bolometric.stefan_boltzmann(10., 1.))
    fl, dfl = lightcurve.mag2flux(np.array([-17.]), np.array([0.05]), np.array([filters.filtdict['g'].M0]))
    out['misc/mag2flux_g'] = np.array([fl[0], dfl[0]])
    # bolometric-style direct (T, R) band photometry + likelihood pieces (bolometric.py:154-164)
    rng = np.random.default_rng(4)
    names = ['U', 'B', 'V', 'g', 'r', 'i']
    T = rng.uniform(1., 100., 32)
    R = 10 ** rng.uniform(-2., 3., 32)
    out['bolo/names'] = np.array(names)
    out['bolo/T'] = T
    out['bolo/R'] = R
    out['bolo/y_z0.01'] = np.array([[filters.filtdict[n].synthesize(models.planck_fast, t, r, z=0.01, ebv=0.,
                                                                  cutoff_freq=np.inf) for n in names]
                                    for t, r in zip(T, R)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--only', default='', help='comma-separated subset of fixture files to regenerate')
    args = ap.parse_args()
    models, filters, bolometric, lightcurve, Table = import_reference(args.ref)
    os.makedirs(GOLD, exist_ok=True)
    jobs = {
        'filters': lambda o: gen_filters(filters, o),
        'primitives': lambda o: (gen_constants(models, filters, bolometric, o), gen_planck(models, filters, o),
                                 gen_priors_misc(models, filters, bolometric, lightcurve, o)),
        'shockcooling': lambda o: gen_shockcooling(models, filters, Table, o),
        'shockcooling3': lambda o: gen_shockcooling3(models, filters, Table, o),
        'companion': lambda o: gen_companion(models, filters, Table, o),
        'config2': lambda o: gen_config2(models, filters, Table, o),
        'config3': lambda o: gen_config3(models, filters, Table, o),
        'config1': lambda o: gen_config1(models, filters, lightcurve, Table, o, args.ref),
        'sed': lambda o: gen_sed(models, filters, o),
    }
    only = set(args.only.split(',')) if args.only else None
    for name, job in jobs.items():
        if only and name not in only:
            continue
        out = {}
        job(out)
        path = os.path.join(GOLD, name + '.npz')
        np.savez_compressed(path, **{k.replace('/', '__'): v for k, v in out.items()})
        print(f'{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
