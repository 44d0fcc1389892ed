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
                                     ctab=(tabs.coff, tabs.ca, tabs.cw, tabs.ctmin))
        off = np.concatenate([[0], np.cumsum([len(fl) for fl in filts])])
        idx = np.array([lookup[f] for fl in filts for f in fl], dtype=np.int32)
        y = np.concatenate([np.asarray(e[1], dtype=float) for e in epochs]) if len(epochs) else np.zeros(0)
        dy = np.concatenate([np.asarray(e[2], dtype=float) for e in epochs]) if len(epochs) else np.zeros(0)
        self.engine.set_observations(off, idx, y, dy)
        self.n_epochs = len(epochs)
        self.samples_per_candidate = np.array([sum(tabs.off[lookup[f] + 1] - tabs.off[lookup[f]] for f in fl)
                                               for fl in filts])

    def __call__(self, candidates, sigma_type='relative', precision='f64', compressed=True):
        """``candidates``: (n_epochs, n_cand, 2|3) of (T, R[, sigma]) -> log-likelihoods (n_epochs, n_cand).
        ``compressed``: use the Gauss-compressed band tables where they are valid (same sums to 2e-14)."""
        if sigma_type not in ('relative', 'absolute'):
            raise Exception('sigma_type must either be "relative" or "absolute"')
        st = _eng.SIGMA_RELATIVE if sigma_type == 'relative' else _eng.SIGMA_ABSOLUTE
        return self.engine.log_likelihood(candidates, st, {'f64': 0, 'f32': 1}[precision], compressed)


def spectrum_log_likelihood(filters, y, dy, T, R, z=0., sigma=None, sigma_type='relative', precision='f64'):
    """Log-likelihood of one epoch's SED for arrays of candidate (T, R[, sigma])."""
    T = np.atleast_1d(np.asarray(T, dtype=float))
    cols = [T, np.broadcast_to(np.asarray(R, dtype=float), T.shape)]
    if sigma is not None:
        cols.append(np.broadcast_to(np.asarray(sigma, dtype=float), T.shape))
    like = SpectrumLikelihood([(filters, y, dy)], z=z)
    return like(np.stack(cols, axis=-1)[None], sigma_type, precision)[0]
