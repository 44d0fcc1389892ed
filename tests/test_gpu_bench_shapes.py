"""Parity of the SAMPLER at the shapes the benchmark times (BASELINE configs[1], [2], [4]): the chain of the device
against the oracle-driven chain with the same counter-based draws, and the posterior moments of a long device run
against a long oracle run."""
import os
import sys

import numpy as np
import pytest

from conftest import relerr
from lightcurve_fitting_amd.sampler import EnsembleSampler, PopulationSampler
from oracle import lcf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu
N_THREADS = max(1, min(16, os.cpu_count() or 1))


def _config1():
    model, lc, priors = bench.build_problem(0)
    bands = [O.band(n) for n in lc['filter']]
    orc = O.ShockCoolingOracle(0., 1.5)
    lo = np.array([p.p_min for p in priors])
    hi = np.array([p.p_max for p in priors])

    def log_posterior(block):  # uniform priors (strict bounds) + the plain-C restatement of the likelihood
        block = np.atleast_2d(block)
        out = np.full(len(block), -np.inf)
        ok = np.all((block > lo) & (block < hi), axis=1)
        if ok.any():
            out[ok] = O.c_shock_cooling_loglike(orc, lc['MJD'], bands, lc['lum'], lc['dlum'], block[ok], N_THREADS)
        return out
    return model, lc, priors, log_posterior


def test_config1_chain_at_the_benchmark_shape():
    """1024 walkers x 3000 points x 6 steps: every one of the 512 workgroups of a launch commits its own walker; the
    chain must be the oracle's (the C restatement makes the 7000 evaluations cost seconds)."""
    model, lc, priors, log_posterior = _config1()
    eng = model.engine_for(lc, priors=priors)
    x0 = bench.initial_walkers(1024)
    s = EnsembleSampler(1024, 5, eng, seed=bench.SEED)
    s.run_mcmc(x0, 6)
    # (the kernel the headline times: resident workgroups -- not the launch-by-launch repeat of a launch that gave up)
    assert s._native.last_run_kernel() == 'run'
    ref, ref_lp, ref_acc = O.stretch_move_run(log_posterior, x0, 6, bench.SEED)
    assert relerr(s.get_chain(), ref) < 1e-9 and relerr(s.get_log_prob(), ref_lp) < 1e-9
    assert np.array_equal(np.round(s.acceptance_fraction * 6).astype(int), ref_acc) and ref_acc.sum() > 500
    # the same chain from the kernel the multi-GPU run uses
    f = EnsembleSampler(1024, 5, eng, seed=bench.SEED)
    assert f._native.set_half_step_kernel('fused') == 'fused'
    f.run_mcmc(x0, 6)
    assert np.array_equal(f.get_chain(), s.get_chain())


def test_config1_posterior_moments_against_a_long_oracle_chain():
    """Statistics, not draws: from the same burnt-in ensemble (3000 device steps), 150 device steps with seed A and 150
    oracle-driven steps with seed B -- 153 600 samples each -- must describe the same posterior.  Tolerances: means
    within 0.15 posterior standard deviations (f_rho M is barely constrained and mixes slowly: both chains cover the
    same window of the same start, so what remains is their sampling error, ~0.03 sigma at an autocorrelation time
    of ~100 steps), standard deviations within 15 %, correlation coefficients within 0.1."""
    model, lc, priors, log_posterior = _config1()
    eng = model.engine_for(lc, priors=priors)
    burn = EnsembleSampler(1024, 5, eng, seed=11)
    start = burn.run_mcmc(bench.initial_walkers(1024), 3000, store=False)
    dev = EnsembleSampler(1024, 5, eng, seed=12)
    dev.run_mcmc(start.coords, 150)
    ref, _, _ = O.stretch_move_run(log_posterior, start.coords, 150, 13, log_prob0=start.log_prob)
    a, b = dev.get_chain(flat=True), ref.reshape(-1, 5)
    sd = a.std(axis=0)
    assert np.all(np.abs(a.mean(axis=0) - b.mean(axis=0)) < 0.15 * sd), (a.mean(0), b.mean(0), sd)
    assert np.all(np.abs(b.std(axis=0) / sd - 1.) < 0.15), (b.std(0), sd)
    ca, cb = np.corrcoef(a.T), np.corrcoef(b.T)   # (R and v_s are strongly anti-correlated in this model)
    assert np.max(np.abs(ca - cb)) < 0.1, (ca, cb)
    assert 0.1 < dev.acceptance_fraction.mean() < 0.7


def test_config2_chain_at_the_benchmark_shape():
    """CompanionShocking, 8000 points (four parts -> 1024-thread workgroups, one proposal per CU: resident, k_solo_run<8, 1,
    true, 8>), 512 walkers x 2 steps against the oracle-driven chain -- and the same chain from a launch per half-step."""
    model, lc, priors, lum0 = bench.build_companion(0)
    bands = [O.band(n) for n in lc['filter']]
    orc = O.CompanionShockingOracle(bands, lum0, z=0.003, variant=1)
    pri = [p.descriptor() for p in priors]

    def log_posterior(block):
        return np.array([O.log_posterior(('CompanionShocking', orc), lc['MJD'], bands, lc['lum'], lc['dlum'], pri, p)
                         for p in np.atleast_2d(block)])
    eng = model.engine_for(lc, priors=priors)
    x0 = bench.companion_walkers(512)
    s = EnsembleSampler(512, 8, eng, seed=bench.SEED)
    assert s._native.set_half_step_kernel('auto') == 'run'   # (four parts, one proposal per CU: 1024-thread workgroups, resident)
    s.run_mcmc(x0, 2)
    assert s._native.last_run_kernel() == 'run'
    ref, ref_lp, ref_acc = O.stretch_move_run(log_posterior, x0, 2, bench.SEED)
    assert relerr(s.get_chain(), ref) < 1e-9 and relerr(s.get_log_prob(), ref_lp) < 1e-9
    assert np.array_equal(np.round(s.acceptance_fraction * 2).astype(int), ref_acc)
    for kernel in ('fused', 'solo'):
        f = EnsembleSampler(512, 8, eng, seed=bench.SEED)
        assert f._native.set_half_step_kernel(kernel) == kernel
        f.run_mcmc(x0, 2)
        assert np.array_equal(f.get_chain(), s.get_chain())


def test_config4_population_at_the_benchmark_shape():
    """32 transients x 512 walkers in lock step (resident workgroups for all of them: k_pop_run) == 32 independent
    single-transient runs, the same walker positions bit for bit; three of the transients against the oracle chain."""
    from lightcurve_fitting_amd import models as M
    rng = np.random.default_rng(bench.SEED + 5)
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    problems, x0 = [], {}
    for k in range(32):
        truth = bench.TRUTH * rng.uniform(0.8, 1.2, 5)
        epochs = np.sort(rng.uniform(0.5, 10., 100))
        t, names = np.repeat(epochs, 6), list(np.tile(bench.BANDS, 100))
        model = M.ShockCooling(redshift=0.)
        ytrue = model(t, names, *truth)
        lc = {'MJD': t, 'filter': names, 'lum': ytrue * (1 + 0.05 * rng.standard_normal(600)), 'dlum': 0.05 * ytrue}
        problems.append((model, lc, priors))
        x0[k] = truth * rng.uniform(0.9, 1.1, (512, 5))
    pop = PopulationSampler(problems, 512, seed=77)
    pop.run_mcmc(x0, 3)
    assert pop[0]._native.last_run_kernel() == 'population-run'
    orc = O.ShockCoolingOracle(0., 1.5)
    for k in range(32):
        model, lc, pri = problems[k]
        solo = EnsembleSampler(512, 5, model.engine_for(lc, priors=pri), seed=77 + k)
        solo.run_mcmc(x0[k], 3)
        assert np.array_equal(pop[k].get_chain(), solo.get_chain()), k
        # (the population launch merges a proposal's parts into one workgroup: another summation tree, last-bit
        # differences in the log-probabilities)
        np.testing.assert_allclose(pop[k].get_log_prob(), solo.get_log_prob(), rtol=1e-12, atol=1e-9)
        if k in (0, 13, 31):
            bands = [O.band(n) for n in lc['filter']]

            def log_posterior(block, lc=lc, bands=bands):
                block = np.atleast_2d(block)
                out = np.full(len(block), -np.inf)
                ok = np.all((block > [0., 0., 0., 0., -1.]) & (block < [10., 10., 10., 10., 0.5]), axis=1)
                out[ok] = O.c_shock_cooling_loglike(orc, lc['MJD'], bands, lc['lum'], lc['dlum'], block[ok], N_THREADS)
                return out
            ref, ref_lp, _ = O.stretch_move_run(log_posterior, x0[k], 3, 77 + k)
            assert relerr(solo.get_chain(), ref) < 1e-9 and relerr(solo.get_log_prob(), ref_lp) < 1e-9
