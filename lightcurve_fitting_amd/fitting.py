"""``lightcurve_mcmc`` counterpart (reference fitting.py:16-168): same keyword signature, checks and return
conventions; the ensemble runs on the MI355X instead of inside emcee's Python loop.

Plotting (``show`` / ``save_plot_as``), ``lightcurve_corner`` and ``lightcurve_model_plot`` are outside the hot path
(SURVEY.md section 8): they only consume ``sampler.chain`` / ``sampler.flatchain``, which the returned object provides.
"""
import warnings

import numpy as np

from .models import UniformPrior
from .sampler import EnsembleSampler

PRIOR_WARNING = 'The p_max/p_min keywords are deprecated. Use the priors keyword instead.'
MODEL_KWARGS_WARNING = 'The model_kwargs keyword is deprecated. These are now included in the model intialization.'


def make_log_posterior(lc, model, priors=None, use_sigma=False, sigma_type='relative'):
    """The ``log_posterior`` callable that the reference builds inside ``lightcurve_mcmc`` (fitting.py:121-128), backed
    by the device engine.  ``f(p)`` with ``p`` of shape ``(ndim,)`` returns a float, exactly like the reference's
    closure; with a C-contiguous ``(n, ndim)`` block it returns ``(n,)`` -- the form
    ``emcee.EnsembleSampler(nwalkers, ndim, f, vectorize=True)`` calls once per half-step."""
    engine = model.engine_for(lc, use_sigma=use_sigma, sigma_type=sigma_type, priors=priors)

    def log_posterior(p):
        p = np.asarray(p, dtype=np.float64)
        out = engine.log_posterior(p)
        return float(out[0]) if p.ndim == 1 else out

    log_posterior.engine = engine
    return log_posterior


def _prepare_photometry(lc, model):
    """The column the model is fitted to, derived from the magnitudes when ``lc`` knows how (fitting.py:68-72)."""
    if not hasattr(lc, 'calcAbsMag'):
        return                                           # plain column stores (dict of arrays) carry it already
    for method in {'flux': ('calcFlux',), 'lum': ('calcAbsMag', 'calcLum')}.get(model.output_quantity, ()):
        getattr(lc, method)()


def _bound_vector(values, ndim, fill, *, deprecated=False, name=None, fallback=None):
    """One of the four length-``ndim`` keyword vectors of ``lightcurve_mcmc`` as a float array.

    ``None`` -> ``fallback`` if given, else ``ndim`` copies of ``fill``; a wrong length raises with the reference's
    message (the deprecation text for ``p_min`` / ``p_max``, which also warn when they are used at all)."""
    if values is None:
        return np.full(ndim, fill) if fallback is None else fallback
    if len(values) != ndim:
        raise Exception(PRIOR_WARNING if deprecated else '{} must have length {:d}'.format(name, ndim))
    if deprecated:
        warnings.warn(PRIOR_WARNING)
    return np.asarray(values, dtype=float).copy()


def _resolve_priors(priors, p_min, p_max, ndim):
    if priors is None:
        return [UniformPrior(lo, hi) for lo, hi in zip(p_min, p_max)]
    if len(priors) != ndim:
        raise Exception('priors must have length {:d}'.format(ndim))
    return priors


def _check_start_box(names, priors, p_lo, p_up):
    """The box the walkers start in must lie inside the priors' support (fitting.py:113-119)."""
    for k, (param, prior) in enumerate(zip(names, priors)):
        for label, guess, limit_name, limit, outside in (('p_lo', p_lo[k], 'p_min', prior.p_min, p_lo[k] < prior.p_min),
                                                         ('p_up', p_up[k], 'p_max', prior.p_max, p_up[k] > prior.p_max)):
            if outside:
                raise Exception(f'starting guess for {param} ({label} = {guess}) is outside prior '
                                f'({limit_name} = {limit})')


def lightcurve_mcmc(lc, model, priors=None, p_min=None, p_max=None, p_lo=None, p_up=None,
                    nwalkers=100, nsteps=1000, nsteps_burnin=1000, model_kwargs=None,
                    show=False, save_plot_as='', save_sampler_as='', use_sigma=False, sigma_type='relative',
                    seed=None):
    """Fit an analytical model to observed photometry with an affine-invariant ensemble sampler on the GPU.

    Arguments as in the reference (fitting.py:16-58).  ``seed`` (extension) keys the counter-based RNG; by default
    it is drawn from NumPy's global generator, which also provides the starting guesses (fitting.py:132), so
    ``np.random.seed`` makes a run reproducible exactly as it does for the reference.

    Returns the sampler (``.chain`` (nwalkers, nsteps, ndim), ``.flatchain``, ``.run_mcmc``, ``.reset``).
    """
    import time
    marks = [('call', time.perf_counter())]
    if model_kwargs is not None:
        raise Exception(MODEL_KWARGS_WARNING)
    _prepare_photometry(lc, model)
    if use_sigma and model.input_names[-1] != '\\sigma':   # the intrinsic-scatter parameter joins the model's own
        model.input_names.append('\\sigma')
        model.units.append('')
    ndim = model.nparams

    p_min = _bound_vector(p_min, ndim, -np.inf, deprecated=True)
    p_max = _bound_vector(p_max, ndim, np.inf, deprecated=True)
    p_lo = _bound_vector(p_lo, ndim, None, name='p_lo', fallback=p_min)
    if p_up is None:
        raise Exception('p_up must have length {:d}'.format(ndim))
    p_up = _bound_vector(p_up, ndim, None, name='p_up')
    priors = _resolve_priors(priors, p_min, p_max, ndim)
    _check_start_box(model.input_names, priors, p_lo, p_up)

    # log_posterior of fitting.py:121-128 lives on the device: priors are baked into the engine
    marks.append(('checks', time.perf_counter()))
    engine = model.engine_for(lc, use_sigma=use_sigma, sigma_type=sigma_type, priors=priors)
    marks.append(('engine', time.perf_counter()))
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(np.random.randint(0, 2 ** 31 - 1))
    sampler = EnsembleSampler(nwalkers, ndim, engine, seed=seed)

    start = p_lo + (p_up - p_lo) * np.random.rand(nwalkers, ndim)      # uniform in the starting box
    marks.append(('sampler', time.perf_counter()))
    burned_in = sampler.run_mcmc(start, nsteps_burnin)
    marks.append(('burn_in', time.perf_counter()))
    if show or save_plot_as:
        warnings.warn('chain plots are not produced by the MI355X engine; plot sampler.chain with the reference tools')
    sampler.reset()                                                    # keep only the post-burn-in chain
    sampler.run_mcmc(burned_in.coords, nsteps, skip_initial_state_check=True)
    marks.append(('run', time.perf_counter()))
    #: seconds of this call, phase by phase, up to the chain complete in HBM (it crosses PCIe when it is first read):
    #: argument checks and photometry; the engine (band tables packed on the host + device engine created; ~0 when the
    #: model already holds an engine for this photometry); sampler and starting positions; burn-in; the stored run
    sampler.timings = {name: t - t0 for (name, t), (_, t0) in zip(marks[1:], marks[:-1])}
    sampler.timings['total'] = marks[-1][1] - marks[0][1]
    sampler.timings['engine_parts'] = dict(getattr(engine, 'timings', {}))
    if save_sampler_as:
        print('saving sampler.flatchain as ' + save_sampler_as)
        np.save(save_sampler_as, sampler.flatchain)
    return sampler
