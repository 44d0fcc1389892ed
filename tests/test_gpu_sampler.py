"""The device-resident stretch-move sampler against the CPU oracle sampler (same counter-based RNG) and by its
statistics; ``lightcurve_mcmc`` end to end."""
import numpy as np
import pytest

from conftest import relerr
from helpers import lc_dict, oracle_log_posterior, small_problem
from lightcurve_fitting_amd import models as M, rng
from lightcurve_fitting_amd.fitting import lightcurve_mcmc
from lightcurve_fitting_amd.sampler import EnsembleSampler
from oracle import lcf_oracle as O

pytestmark = pytest.mark.gpu


def _setup(nwalkers, seed=1):
    pb = small_problem()
    lc = lc_dict(pb['t'], pb['names'], pb['y'], pb['dy'])
    m = M.ShockCooling(redshift=0.)
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    eng = m.engine_for(lc, priors=priors)
    r = np.random.default_rng(seed)
    x0 = pb['truth'] * (1 + 0.05 * r.standard_normal((nwalkers, 5)))
    return pb, lc, m, eng, x0


@pytest.mark.parametrize('randomize', [True, False])
def test_chain_matches_oracle_sampler(randomize):
    pb, lc, m, eng, x0 = _setup(32)
    s = EnsembleSampler(32, 5, eng, seed=987654321012345, randomize_split=randomize)
    state = s.run_mcmc(x0, 12)
    ref, ref_lp, ref_acc = O.stretch_move_run(oracle_log_posterior(pb), x0, 12, 987654321012345,
                                              randomize_split=randomize)
    # identical accept/reject decisions -> identical chains up to likelihood rounding (1e-14)
    assert relerr(s.get_chain(), ref) < 1e-9
    assert relerr(s.get_log_prob(), ref_lp) < 1e-9
    assert np.array_equal(np.round(s.acceptance_fraction * 12).astype(int), ref_acc)
    assert s.chain.shape == (32, 12, 5) and s.flatchain.shape == (32 * 12, 5)
    assert np.array_equal(s.flatchain[:12], s.get_chain()[:, 0, :])  # walker-major flattening
    coords, lp, _ = state
    assert np.array_equal(coords, s.get_chain()[-1]) and np.array_equal(lp, s.get_log_prob()[-1])
    # continuing from the stored state uses the next RNG steps
    s.run_mcmc(None, 3)
    ref2, _, _ = O.stretch_move_run(oracle_log_posterior(pb), ref[-1], 3, 987654321012345, log_prob0=ref_lp[-1],
                                    randomize_split=randomize, first_step=12)
    assert relerr(s.get_chain()[12:], ref2) < 1e-9
    s.reset()
    assert s.chain.shape == (32, 0, 5)


def test_posterior_statistics_and_determinism():
    pb, lc, m, eng, x0 = _setup(64)
    a = EnsembleSampler(64, 5, eng, seed=5)
    a.run_mcmc(x0, 300)
    b = EnsembleSampler(64, 5, eng, seed=5)
    b.run_mcmc(x0, 300)
    assert np.array_equal(a.get_chain(), b.get_chain())  # bitwise reproducible
    c = EnsembleSampler(64, 5, eng, seed=6)
    c.run_mcmc(x0, 300)
    assert not np.array_equal(a.get_chain(), c.get_chain())
    flat = a.get_chain(discard=150, flat=True)
    assert np.all(np.isfinite(flat))
    # well-constrained combinations recover the truth: R (radius) and t_0
    assert abs(np.median(flat[:, 4]) - pb['truth'][4]) < 0.1
    assert 0.15 < a.acceptance_fraction.mean() < 0.8
    assert a.last_run_ms > 0.


def test_sampler_input_validation():
    pb, lc, m, eng, x0 = _setup(32)
    with pytest.raises(ValueError, match='fewer walkers than twice'):
        EnsembleSampler(8, 5, eng)
    with pytest.raises(ValueError, match='ndim'):
        EnsembleSampler(32, 4, eng)
    s = EnsembleSampler(32, 5, eng)
    with pytest.raises(ValueError, match='dimensions'):
        s.run_mcmc(x0[:10], 2)
    with pytest.raises(ValueError, match='infinite or NaN'):
        bad = x0.copy()
        bad[0, 0] = np.nan
        s.run_mcmc(bad, 2)
    with pytest.raises(ValueError, match='linearly independent'):
        s.run_mcmc(np.tile(x0[0], (32, 1)), 2)
    with pytest.raises(ValueError, match='never been called'):
        EnsembleSampler(32, 5, eng).run_mcmc(None, 2)
    # NaN log-probability (R < 0 inside an unbounded prior) -> ValueError like emcee
    m2 = M.ShockCooling(redshift=0.)
    eng2 = m2.engine_for(lc)  # no priors
    bad = x0.copy()
    bad[:, 3] *= -1
    with pytest.raises(ValueError, match='NaN'):
        EnsembleSampler(32, 5, eng2).run_mcmc(bad, 2)


def test_sharded_phases_equal_fused_run():
    """propose / evaluate(shards) / accept called phase by phase reproduce lcf_sampler_run exactly."""
    from lightcurve_fitting_amd.engine import NativeSampler
    pb, lc, m, eng, x0 = _setup(32)
    perm = rng.split_permutations(77, 0, 5, 32)
    a = NativeSampler(eng, 32, 77)
    a.set_state(x0)
    a.run(0, 5, 'random', True)
    b = NativeSampler(eng, 32, 77)
    b.set_state(x0)
    b.begin(0, 5, perm, True)  # host-provided colouring == device-generated colouring
    for step in range(5):
        for half in (0, 1):
            b.propose(step, half)
            for lo, hi in ((0, 5), (5, 11), (11, 16)):  # three uneven "ranks"
                b.evaluate(lo, hi)
            b.accept(step, half)
    b.check()
    assert np.array_equal(a.get_chain()[0], b.get_chain()[0])
    assert np.array_equal(a.naccepted(), b.naccepted())


def test_lightcurve_mcmc_end_to_end():
    pb, lc, m, eng, x0 = _setup(32)
    np.random.seed(3)
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    lo = pb['truth'] * 0.8
    hi = pb['truth'] * 1.2
    sampler = lightcurve_mcmc(lc, m, priors=priors, p_lo=lo, p_up=hi, nwalkers=32, nsteps=40, nsteps_burnin=60)
    assert sampler.chain.shape == (32, 40, 5) and sampler.flatchain.shape == (1280, 5)
    assert np.all(sampler.flatchain[:, :4] > 0) and np.all(sampler.flatchain[:, :4] < 10)
    m2 = M.ShockCooling(redshift=0.)
    s2 = lightcurve_mcmc(lc, m2, priors=priors + [M.UniformPrior(0., 5.)], p_lo=np.append(lo, 0.1),
                         p_up=np.append(hi, 1.), nwalkers=32, nsteps=10, nsteps_burnin=10, use_sigma=True,
                         sigma_type='absolute')
    assert s2.chain.shape == (32, 10, 6) and m2.input_names[-1] == '\\sigma'


def test_collective_paths_single_rank_rccl():
    """Both multi-GPU code paths with one rank -- (i) the native loop with its own RCCL communicator (dlopen'd,
    bootstrapped over torch.distributed) and (ii) the Python-driven loop over torch.distributed collectives on the
    aliased device buffer -- must reproduce the fused single-GPU run bit for bit."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    pb, lc, m, eng, x0 = _setup(32)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        a = EnsembleSampler(32, 5, eng, seed=4242, force_sharded=True)
        a.run_mcmc(x0, 8)
        a.run_mcmc(None, 4)
        assert a._comm not in (None, False), 'the native RCCL communicator was not created'
        c = EnsembleSampler(32, 5, eng, seed=4242, force_sharded=True, native_collectives=False)
        c.run_mcmc(x0, 8)
        c.run_mcmc(None, 4)
        assert c._comm is False
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    b = EnsembleSampler(32, 5, eng, seed=4242)
    b.run_mcmc(x0, 8)
    b.run_mcmc(None, 4)
    for s in (a, c):
        assert np.array_equal(s.get_chain(), b.get_chain())
        assert np.array_equal(s.get_log_prob(), b.get_log_prob())
        assert np.array_equal(s.acceptance_fraction, b.acceptance_fraction)


def test_population_mode_equals_individual_runs():
    """BASELINE configs[4] shape in small: independent transients, all ensembles in flight at once on their own
    streams; every chain equals the chain of that transient run alone (bitwise)."""
    from lightcurve_fitting_amd.sampler import PopulationSampler, partition
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    problems, x0 = [], {}
    for k in range(5):
        pb = small_problem(npts=40 + 7 * k, seed=20 + k)
        lc = lc_dict(pb['t'], pb['names'], pb['y'], pb['dy'])
        problems.append((M.ShockCooling(redshift=0.), lc, priors))
        x0[k] = pb['truth'] * (1 + 0.05 * np.random.default_rng(k).standard_normal((32, 5)))
    solos = []
    for k, (model, lc, pri) in enumerate(problems):
        solo = EnsembleSampler(32, 5, M.ShockCooling(redshift=0.).engine_for(lc, priors=pri), seed=100 + k)
        solo.run_mcmc(x0[k], 10)
        solo.run_mcmc(None, 5)
        solos.append(solo)
    for batched in (True, False):  # one launch per half-step for all transients / one stream per transient
        pop = PopulationSampler(problems, 32, seed=100)
        assert pop.indices == [0, 1, 2, 3, 4]
        pop.run_mcmc(x0, 10, batched=batched)
        pop.run_mcmc(None, 5, batched=batched)
        for k in range(5):
            assert np.array_equal(pop[k].get_chain(), solos[k].get_chain())
            assert np.array_equal(pop[k].get_log_prob(), solos[k].get_log_prob())
            assert np.array_equal(pop[k].acceptance_fraction, solos[k].acceptance_fraction)
            assert pop[k].chain.shape == (32, 15, 5)
    assert [list(partition(7, 3, r)) for r in range(3)] == [[0, 1, 2], [3, 4, 5], [6]]


def test_large_population_regroups_partial_sums():
    """With thousands of proposals in one launch population mode uses one workgroup per proposal instead of the
    engine's several (fewer, longer partial chi^2 sums): log-probabilities agree with the solo runs to rounding and
    the chains make the same decisions."""
    from lightcurve_fitting_amd.sampler import PopulationSampler
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    problems, x0 = [], {}
    for k in range(12):
        pb = small_problem(npts=600 + k, seed=50 + k)
        lc = lc_dict(pb['t'], pb['names'], pb['y'], pb['dy'])
        problems.append((M.ShockCooling(redshift=0.), lc, priors))
        x0[k] = pb['truth'] * (1 + 0.05 * np.random.default_rng(k).standard_normal((700, 5)))
    pop = PopulationSampler(problems, 700, seed=7)
    pop.run_mcmc(x0, 4)
    for k in (0, 5, 11):
        model, lc, pri = problems[k]
        solo = EnsembleSampler(700, 5, model.engine_for(lc, priors=pri), seed=7 + k)
        solo.run_mcmc(x0[k], 4)
        np.testing.assert_allclose(pop[k].get_log_prob(), solo.get_log_prob(), rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(pop[k].get_chain(), solo.get_chain(), rtol=1e-12)
        assert np.array_equal(pop[k].acceptance_fraction, solo.acceptance_fraction)


@pytest.mark.parametrize('nwalkers', [10, 22, 130])
def test_odd_ensemble_sizes_match_oracle(nwalkers):
    """Walker counts that are not multiples of the wavefront or workgroup size (and the emcee minimum 2 ndim)."""
    pb, lc, m, eng, _ = _setup(32)
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(nwalkers).standard_normal((nwalkers, 5)))
    s = EnsembleSampler(nwalkers, 5, eng, seed=31337)
    s.run_mcmc(x0, 6)
    ref, ref_lp, _ = O.stretch_move_run(oracle_log_posterior(pb), x0, 6, 31337)
    assert relerr(s.get_chain(), ref) < 1e-9 and relerr(s.get_log_prob(), ref_lp) < 1e-9


@pytest.mark.parametrize('protocol', ['rows', 'lp'])
@pytest.mark.parametrize('ranks', [2, 4])
def test_emulated_multi_rank_run_on_one_gpu(ranks, protocol):
    """Every kernel path of a multi-GPU run except RCCL itself: `ranks` samplers play the ranks of one ensemble on
    one device (same seed, each evaluating only its shard, with light proposals for the other shards), the
    all-gather is emulated by device-to-device copies between their buffers -- of each proposal's row of partial sums
    and log-prior ('rows': the protocol of the native lcf_sampler_run_sharded, no finalize launch) or of the finished
    log-posteriors ('lp').  Every emulated rank must end with the chain of the fused single-GPU run, bit for bit."""
    import torch
    from lightcurve_fitting_amd.engine import NativeSampler
    from lightcurve_fitting_amd.sampler import NativeBackend, shard_bounds
    pb, lc, m, eng, x0 = _setup(48)
    nsteps, seed, nh = 7, 99, 24
    ref = NativeSampler(eng, 48, seed)
    ref.set_state(x0)
    ref.run(0, nsteps, 'random', True)
    samplers = [NativeSampler(eng, 48, seed) for _ in range(ranks)]
    backs = [NativeBackend(s, rows=protocol == 'rows') for s in samplers]
    for s in samplers:
        s.set_state(x0)
        s.begin(0, nsteps, 'random', True)
    bounds = [shard_bounds(nh, ranks, r)[:2] for r in range(ranks)]
    side = torch.cuda.Stream()  # a real stream: handle 0 would mean "the engine's own stream" to the ABI
    with torch.cuda.stream(side):
        st = side.cuda_stream
        assert st != 0
        for step in range(nsteps):
            for half in (0, 1):
                for r, b in enumerate(backs):
                    b.half_step(step, half, *bounds[r])
                views = [b.newlp() for b in backs]      # current parity's buffers
                assert views[0].shape == ((nh, 2) if protocol == 'rows' else (nh,))  # 40 points: one part
                for r, (lo, hi) in enumerate(bounds):   # "all-gather": rank r's shard reaches every other rank
                    for q in range(ranks):
                        if q != r and hi > lo:
                            views[q][lo:hi].copy_(views[r][lo:hi])
                for s in samplers:
                    s.accept(step, half, st)
    side.synchronize()
    want_chain, want_lp = ref.get_chain()
    for s in samplers:
        s.check()
        chain, lp = s.get_chain()
        assert np.array_equal(chain, want_chain) and np.array_equal(lp, want_lp)
        assert np.array_equal(s.naccepted(), ref.naccepted())


@pytest.mark.parametrize('model_name', ['ShockCooling', 'ShockCooling4'])
def test_chain_matches_oracle_on_a_multiband_grid(model_name):
    """Shared epochs (thermal states per (walker, epoch) in LDS), three chunks of points in two parts, prior-excluded
    proposals, a walker far outside the model's domain: the one-launch half-step against the oracle-driven sampler,
    and against the two-kernel path of the same library bit for bit (separate phases through the C ABI)."""
    rng = np.random.default_rng(77)
    epochs = np.sort(rng.uniform(0.4, 9., 110))
    t = np.repeat(epochs, 6)
    names = list(np.tile(list('UBVgri'), len(epochs)))
    bands = [O.band(n) for n in names]
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    if model_name == 'ShockCooling':
        m, om = M.ShockCooling(redshift=0.004), ('ShockCooling', O.ShockCoolingOracle(0.004))
    else:
        m, om = M.ShockCooling4(redshift=0.004), ('ShockCooling4', O.ShockCooling4Oracle(0.004))
    ytrue = O.evaluate(om, t, bands, truth)
    y, dy = ytrue * (1 + 0.05 * rng.standard_normal(len(t))), 0.05 * ytrue
    lc = lc_dict(t, names, y, dy)
    priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
    pb = dict(model=om, orc=None, t=t, bands=bands, y=y, dy=dy, priors=[p.descriptor() for p in priors])
    x0 = truth * (1 + 0.05 * rng.standard_normal((40, 5)))   # R near its prior edge: some proposals are excluded
    eng = m.engine_for(lc, priors=priors)
    s = EnsembleSampler(40, 5, eng, seed=2024)
    s.run_mcmc(x0, 8)
    ref, ref_lp, _ = O.stretch_move_run(oracle_log_posterior(pb), x0, 8, 2024)
    assert relerr(s.get_chain(), ref) < 1e-9 and relerr(s.get_log_prob(), ref_lp) < 1e-9
    assert 0 < s.acceptance_fraction.mean() < 1
    from lightcurve_fitting_amd.engine import NativeSampler
    ns = NativeSampler(eng, 40, 2024)
    ns.set_state(x0)
    ns.begin(0, 8, 'random', True)
    for step in range(8):           # propose / evaluate / accept as separate launches: k_step + k_points + k_finalize
        for half in (0, 1):
            ns.propose(step, half)
            ns.evaluate(0, 20)
            ns.accept(step, half)
    ns.check()
    chain, lp = ns.get_chain()
    assert np.array_equal(chain, s.get_chain()) and np.array_equal(lp, s.get_log_prob())


def test_long_light_curve_one_launch_and_phases():
    """186 000 points in 31 000 epochs, 8 parts of ~3900 columns each: the thermal states are the lanes' own (no LDS
    budget per epoch any more), so the run is one launch per half-step (k_solo: each half of the workgroup takes four of
    the 8 parts) -- and the same chain as the separate phases (k_step + k_points), bit for bit; the likelihoods of the
    final ensemble against the oracle."""
    from lightcurve_fitting_amd.engine import NativeSampler
    rng = np.random.default_rng(99)
    epochs = np.sort(rng.uniform(0.4, 30., 31000))
    t = np.repeat(epochs, 6)
    names = list(np.tile(list('UBVgri'), len(epochs)))
    bands = [O.band(n) for n in names]
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    om = ('ShockCooling', O.ShockCoolingOracle(0.))
    ytrue = O.evaluate(om, t, bands, truth)
    y, dy = ytrue * (1 + 0.05 * rng.standard_normal(len(t))), 0.05 * ytrue
    m = M.ShockCooling(redshift=0.)
    priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
    eng = m.engine_for(lc_dict(t, names, y, dy), priors=priors)
    x0 = truth * (1 + 0.002 * rng.standard_normal((16, 5)))
    a = NativeSampler(eng, 16, 5)
    assert a.one_launch and NativeSampler(_setup(8)[3], 8, 1).one_launch
    a.set_state(x0)
    a.run(0, 3, 'random', True)
    assert a.last_run_kernel() == 'solo'   # (more than two parts and 8 proposals: a launch per half-step)
    b = NativeSampler(eng, 16, 5)
    b.set_state(x0)
    b.begin(0, 3, 'random', True)
    for step in range(3):
        for half in (0, 1):
            b.propose(step, half)
            b.evaluate(0, 8)
            b.accept(step, half)
    b.check()
    assert np.array_equal(a.get_chain()[0], b.get_chain()[0]) and np.array_equal(a.get_chain()[1], b.get_chain()[1])
    x, lp = a.get_state()
    pri = [p.descriptor() for p in priors]
    want = [O.log_posterior(om, t, bands, y, dy, pri, xi) for xi in x[:2]]
    assert relerr(lp[:2], want) < 1e-10
