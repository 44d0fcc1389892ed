"""Per-epoch blackbody SED likelihoods on the MI355X and the small closed-form helpers around them.

Mirrors the part of the reference's ``bolometric.py`` that shares the hot path's primitive
(SURVEY.md section 8 row a11 / "next" row 1):

* :func:`spectrum_log_likelihood` -- the inner ``log_posterior`` of ``spectrum_mcmc`` without its priors
  (``bolometric.py:154-164``): ``[f.synthesize(planck_fast, T, R) for f in filters]`` + Gaussian likelihood, batched
  over (epoch x candidate) on the device, float64 or float32;
* :func:`pseudo` (``bolometric.py:32-59``), :func:`stefan_boltzmann` (``bolometric.py:422-453``): closed-form,
  one NumPy expression each, evaluated on the host once per result table (not a hot loop).

Grouping observations into epochs, ``curve_fit`` starting values, the MCMC orchestration and the output table of
``calculate_bolometric`` are outside the scope of this engine.
"""
import numpy as np

from . import engine as _eng
from .filters import PackedTables, as_filter, c1, c2, filtdict

#: Stefan-Boltzmann constant in W (1000 Rsun)^-2 kK^-4 (bolometric.py:419)
sigma_sb = 2.744452656619892e+28


def stefan_boltzmann(temp, radius, dtemp=None, drad=None, covTR=None):
    """Blackbody luminosity [W] (and its uncertainty) from T [kK] and R [1000 Rsun] (bolometric.py:422-453)."""
    lum = 4 * np.pi * radius ** 2 * sigma_sb * temp ** 4
    if dtemp is None or drad is None or covTR is None:
        return lum
    dlum = 8 * np.pi * sigma_sb * (radius ** 2 * temp ** 8 * drad ** 2 + 4 * radius ** 4 * temp ** 6 * dtemp ** 2
                                   + 4 * radius ** 3 * temp ** 7 * covTR) ** 0.5
    return lum, dlum


def pseudo(temp, radius, z, filter0=filtdict['I'], filter1=filtdict['U'], cutoff_freq=np.inf):
    """Planck spectrum integrated on a 1-THz grid between two filters [W] (bolometric.py:32-59)."""
    freq0 = filter0.freq_eff - filter0.dfreq / 2.
    freq1 = filter1.freq_eff + filter1.dfreq / 2.
    nu = np.arange(freq0, freq1) * (1. + z)
    temp = np.asarray(temp, dtype=float)
    radius = np.asarray(radius, dtype=float)
    with np.errstate(all='ignore'):
        inv_t = np.where(temp > 0., 1. / np.where(temp > 0., temp, 1.), 0.)
        occ = np.exp(c1 * np.multiply.outer(inv_t, nu)) - 1.
        occ = np.where(occ > 0., 1. / np.where(occ > 0., occ, 1.), 0.)
        lnu = c2 * np.multiply.outer(radius ** 2, nu ** 3 * np.minimum(1., cutoff_freq / nu)) * occ
    tw = np.ones(len(nu))
    tw[[0, -1]] = 0.5
    return np.sum(lnu * tw, axis=-1) * 1e12


class SpectrumLikelihood:
    """Observed SEDs of many epochs resident on the GPU; evaluates candidate blackbodies per epoch.

    Parameters
    ----------
    epochs : sequence of (filters, y, dy)
        Per epoch: the filters (objects or aliases), the observed luminosity densities [W/Hz] and uncertainties.
    z : float
        Redshift between the blackbody and the observed filters.
    """

    def __init__(self, epochs, z=0., cutoff_freq=np.inf, device=0):
        filts = [[as_filter(f) for f in e[0]] for e in epochs]
        uniq = list(dict.fromkeys(f for fl in filts for f in fl))
        lookup = {f: i for i, f in enumerate(uniq)}
        tabs = PackedTables(uniq, z=z, cutoff_freq=cutoff_freq)
        self.engine = _eng.SedEngine(tabs.off, tabs.a, tabs.w, device=device,
                                     ctab=(tabs.coff, tabs.ca, tabs.cw, tabs.ctmin),
                                     itab=tabs.interpolants(below=10))   # from 0.94 kK: the priors start at 1 kK
        self.itab_tmin = tabs.interpolants(below=10)[1]
        off = np.concatenate([[0], np.cumsum([len(fl) for fl in filts])])
        idx = np.array([lookup[f] for fl in filts for f in fl], dtype=np.int32)
        y = np.concatenate([np.asarray(e[1], dtype=float) for e in epochs]) if len(epochs) else np.zeros(0)
        dy = np.concatenate([np.asarray(e[2], dtype=float) for e in epochs]) if len(epochs) else np.zeros(0)
        self.engine.set_observations(off, idx, y, dy)
        self.n_epochs = len(epochs)
        self.samples_per_candidate = np.array([sum(tabs.off[lookup[f] + 1] - tabs.off[lookup[f]] for f in fl)
                                               for fl in filts])

    #: arithmetic of the device kernel: 'f64' = float64 through the interpolants of ln S_f(ln T) (the light-curve
    #: engine's default level; sample tables outside their range), 'f64-tables' = float64 sample by sample (the
    #: reference's own sum), 'f32' = float32 sample by sample (BASELINE configs[3] as specified)
    PRECISIONS = {'f64': 2, 'f64-tables': 0, 'f32': 1}

    def __call__(self, candidates, sigma_type='relative', precision='f64', compressed=True):
        """``candidates``: (n_epochs, n_cand, 2|3) of (T, R[, sigma]) -> log-likelihoods (n_epochs, n_cand).
        ``compressed``: use the Gauss-compressed band tables where they are valid (same sums to 2e-14)."""
        if sigma_type not in ('relative', 'absolute'):
            raise Exception('sigma_type must either be "relative" or "absolute"')
        st = _eng.SIGMA_RELATIVE if sigma_type == 'relative' else _eng.SIGMA_ABSOLUTE
        return self.engine.log_likelihood(candidates, st, self.PRECISIONS[precision], compressed)


def spectrum_log_likelihood(filters, y, dy, T, R, z=0., sigma=None, sigma_type='relative', precision='f64'):
    """Log-likelihood of one epoch's SED for arrays of candidate (T, R[, sigma])."""
    T = np.atleast_1d(np.asarray(T, dtype=float))
    cols = [T, np.broadcast_to(np.asarray(R, dtype=float), T.shape)]
    if sigma is not None:
        cols.append(np.broadcast_to(np.asarray(sigma, dtype=float), T.shape))
    like = SpectrumLikelihood([(filters, y, dy)], z=z)
    return like(np.stack(cols, axis=-1)[None], sigma_type, precision)[0]


def blackbody_grid_fit(epochs, z=0., T_grid=None, R_grid=None, precision='f64', device=0):
    """Per-epoch blackbody fit on a dense (T, R) grid: the device evaluates every epoch's log-likelihood surface in
    one launch; the posterior moments under the reference's default priors -- uniform in T, log-uniform in R
    (``bolometric.py:729``) -- give the estimates the reference gets from ``curve_fit`` / a short MCMC per epoch.

    Returns a dict of arrays over epochs: ``temp, radius`` (maximum likelihood), ``temp_mean, dtemp, radius_mean,
    dradius, covTR`` (posterior moments), ``lum, dlum`` (Stefan-Boltzmann at the posterior mean) and ``lnL_max``."""
    T_grid = np.linspace(1., 100., 128) if T_grid is None else np.asarray(T_grid, dtype=float)
    R_grid = np.geomspace(0.01, 1000., 128) if R_grid is None else np.asarray(R_grid, dtype=float)
    like = SpectrumLikelihood(epochs, z=z, device=device)
    TT, RR = np.meshgrid(T_grid, R_grid, indexing='ij')
    cand = np.broadcast_to(np.stack([TT.ravel(), RR.ravel()], axis=-1), (like.n_epochs, TT.size, 2))
    lnl = like(np.ascontiguousarray(cand), precision=precision)                 # (n_epochs, nT * nR)
    best = np.argmax(lnl, axis=1)
    # quadrature weights: uniform prior in T -> dT; log-uniform prior in R -> d(ln R) on the geometric grid
    wT = np.gradient(T_grid)
    wR = np.gradient(np.log(R_grid))
    w = np.exp(lnl - lnl.max(axis=1, keepdims=True)) * np.outer(wT, wR).ravel()
    w /= w.sum(axis=1, keepdims=True)
    Tm, Rm = w @ TT.ravel(), w @ RR.ravel()
    dT = np.sqrt(np.maximum(w @ TT.ravel() ** 2 - Tm ** 2, 0.))
    dR = np.sqrt(np.maximum(w @ RR.ravel() ** 2 - Rm ** 2, 0.))
    cov = w @ (TT.ravel() * RR.ravel()) - Tm * Rm
    lum, dlum = stefan_boltzmann(Tm, Rm, dT, dR, cov)
    return dict(temp=TT.ravel()[best], radius=RR.ravel()[best], temp_mean=Tm, dtemp=dT, radius_mean=Rm, dradius=dR,
                covTR=cov, lum=lum, dlum=dlum, lnL_max=lnl[np.arange(len(best)), best])


def spectrum_mcmc_population(epochs, priors=None, z=0., nwalkers=10, burnin_steps=200, steps=100, T_range=(1., 100.),
                             R_range=(0.01, 1000.), seed=0, use_sigma=False, sigma_type='relative'):
    """``spectrum_mcmc`` (bolometric.py:87-190) for MANY epochs at once: one (T, R[, sigma]) ensemble per epoch, all
    running in lock step on the device (population mode).  Defaults follow the reference (10 walkers, 200 burn-in +
    100 steps, uniform prior on T, log-uniform on R).  Returns ``chains[n_epochs][steps * nwalkers, ndim]``."""
    from .models import Blackbody, LogUniformPrior, UniformPrior
    from .sampler import PopulationSampler
    ndim = 3 if use_sigma else 2
    if priors is None:
        priors = [UniformPrior(*T_range), LogUniformPrior(*R_range)] + ([UniformPrior(0., 10.)] if use_sigma else [])
    if nwalkers < 2 * ndim:
        raise ValueError('nwalkers must be at least 2 * ndim')
    problems, x0 = [], {}
    rng = np.random.default_rng(seed)
    for k, (filts, y, dy) in enumerate(epochs):
        lc = {'MJD': np.zeros(len(y)), 'filter': list(filts), 'lum': np.asarray(y, float), 'dlum': np.asarray(dy, float)}
        model = Blackbody(redshift=z)
        if use_sigma:
            model.input_names.append('\\sigma')
        problems.append((model, lc, priors, dict(use_sigma=use_sigma, sigma_type=sigma_type)))
        # starting guesses: uniform over the prior ranges like the reference (bolometric.py:166)
        cols = [rng.uniform(*T_range, nwalkers), np.exp(rng.uniform(*np.log(R_range), nwalkers))]
        if use_sigma:
            cols.append(rng.uniform(0., 1., nwalkers))
        x0[k] = np.column_stack(cols)
    pop = PopulationSampler(problems, nwalkers, seed=seed)
    pop.run_mcmc(x0, burnin_steps, store=False)
    for s in pop.samplers.values():
        s.reset()
    pop.run_mcmc(None, steps)
    return [pop[k].flatchain for k in range(len(epochs))]
