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
    if model_kwargs is not None:
        raise Exception(MODEL_KWARGS_WARNING)

    if hasattr(lc, 'calcAbsMag'):
        if model.output_quantity == 'flux':
            lc.calcFlux()
        elif model.output_quantity == 'lum':
            lc.calcAbsMag()
            lc.calcLum()

    if use_sigma and model.input_names[-1] != '\\sigma':
        model.input_names.append('\\sigma')
        model.units.append('')

    ndim = model.nparams

    if p_min is None:
        p_min = np.tile(-np.inf, ndim)
    elif len(p_min) == ndim:
        p_min = np.array(p_min, float)
        warnings.warn(PRIOR_WARNING)
    else:
        raise Exception(PRIOR_WARNING)

    if p_max is None:
        p_max = np.tile(np.inf, ndim)
    elif len(p_max) == ndim:
        p_max = np.array(p_max, float)
        warnings.warn(PRIOR_WARNING)
    else:
        raise Exception(PRIOR_WARNING)

    if p_lo is None:
        p_lo = p_min
    elif len(p_lo) == ndim:
        p_lo = np.array(p_lo, float)
    else:
        raise Exception('p_lo must have length {:d}'.format(ndim))

    if p_up is not None and len(p_up) == ndim:
        p_up = np.array(p_up, float)
    else:
        raise Exception('p_up must have length {:d}'.format(ndim))

    if priors is None:
        priors = [UniformPrior(p0, p1) for p0, p1 in zip(p_min, p_max)]
    elif len(priors) != ndim:
        raise Exception('priors must have length {:d}'.format(ndim))

    for param, prior, p0, p1 in zip(model.input_names, priors, p_lo, p_up):
        if p0 < prior.p_min:
            raise Exception(f'starting guess for {param} (p_lo = {p0}) is outside prior (p_min = {prior.p_min})')
        if p1 > prior.p_max:
            raise Exception(f'starting guess for {param} (p_up = {p1}) is outside prior (p_max = {prior.p_max})')

    # log_posterior of fitting.py:121-128 lives on the device: priors are baked into the engine
    engine = model.engine_for(lc, use_sigma=use_sigma, sigma_type=sigma_type, priors=priors)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(np.random.randint(0, 2 ** 31 - 1))
    sampler = EnsembleSampler(nwalkers, ndim, engine, seed=seed)

    starting_guesses = np.random.rand(nwalkers, ndim) * (p_up - p_lo) + p_lo
    pos, _, _ = sampler.run_mcmc(starting_guesses, nsteps_burnin)

    if show or save_plot_as:
        warnings.warn('chain plots are not produced by the MI355X engine; plot sampler.chain with the reference tools')

    sampler.reset()
    sampler.run_mcmc(pos, nsteps, skip_initial_state_check=True)
    if save_sampler_as:
        np.save(save_sampler_as, sampler.flatchain)
        print('saving sampler.flatchain as ' + save_sampler_as)

    return sampler
