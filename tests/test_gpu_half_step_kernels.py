"""The three ways a single-GPU run can execute a half-step -- one workgroup per proposal that also accepts or rejects
(k_solo), one workgroup per (proposal, part) (k_fused), separate proposal / likelihood launches -- must produce ONE
chain, bit for bit, and that chain must be the oracle-driven one: odd ensembles, runs that cross the blocks in which the
draw records are generated, runs that continue each other, a light curve edited in place."""
import numpy as np
import pytest

from conftest import relerr
from helpers import lc_dict, oracle_log_posterior, small_problem
from lightcurve_fitting_amd import models as M
from lightcurve_fitting_amd.engine import NativeSampler
from lightcurve_fitting_amd.sampler import EnsembleSampler
from oracle import lcf_oracle as O

pytestmark = pytest.mark.gpu

PRIORS = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]


def _small(nwalkers, seed=1):
    pb = small_problem()
    lc = lc_dict(pb['t'], pb['names'], pb['y'], pb['dy'])
    eng = M.ShockCooling(redshift=0.).engine_for(lc, priors=PRIORS)
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(seed).standard_normal((nwalkers, 5)))
    return pb, eng, x0


def _multiband(n_epochs=110, seed=77):
    """Shared epochs, several chunks of points in two parts, R near its prior edge (some proposals are excluded)."""
    rng = np.random.default_rng(seed)
    epochs = np.sort(rng.uniform(0.4, 9., n_epochs))
    t, names = np.repeat(epochs, 6), list(np.tile(list('UBVgri'), n_epochs))
    bands = [O.band(n) for n in names]
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    om = ('ShockCooling', O.ShockCoolingOracle(0.004))
    ytrue = O.evaluate(om, t, bands, truth)
    y, dy = ytrue * (1 + 0.05 * rng.standard_normal(len(t))), 0.05 * ytrue
    priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
    pb = dict(model=om, orc=None, t=t, bands=bands, y=y, dy=dy, priors=[p.descriptor() for p in priors], truth=truth)
    eng = M.ShockCooling(redshift=0.004).engine_for(lc_dict(t, names, y, dy), priors=priors)
    return pb, eng


def _run(eng, nwalkers, seed, x0, nsteps, kernel, split='random'):
    s = NativeSampler(eng, nwalkers, seed)
    used = s.set_half_step_kernel(kernel)
    s.set_state(x0)
    s.run(0, nsteps, split, True)
    chain, lp = s.get_chain()
    return used, chain, lp, s.naccepted(), s


def test_states_outside_the_interpolants_take_the_general_path():
    """Half of the walkers start with an explosion time BEHIND the first epochs (negative phases: no log-space state) and
    every proposal that mixes such a walker with another one lands anywhere in between: waves of the model-specialised
    half-step kernel leave their straight-line path for the out-of-line general one, waves of the generic kernels fall
    back from the interpolants to the sample tables -- one chain for all kernels, and the oracle's."""
    pb, eng = _multiband()
    rng = np.random.default_rng(11)
    x0 = pb['truth'] * (1 + 0.05 * rng.standard_normal((64, 5)))
    x0[::2, 4] = rng.uniform(0.405, 0.49, 32)      # epochs start at 0.4 d; the prior allows up to 0.5
    assert np.any(pb['t'].min() < x0[:, 4])
    runs = {k: _run(eng, 64, 5, x0, 6, k) for k in ('auto', 'solo', 'fused', 'phases')}
    assert runs['auto'][0] == 'run' and runs['solo'][0] == 'solo'
    for k in ('solo', 'fused', 'phases'):
        assert np.array_equal(runs['auto'][1], runs[k][1]) and np.array_equal(runs['auto'][2], runs[k][2])
    ref, ref_lp, ref_acc = O.stretch_move_run(oracle_log_posterior(pb), x0, 6, 5)
    assert relerr(runs['auto'][1], ref) < 1e-9 and relerr(runs['auto'][2], ref_lp) < 1e-9
    assert np.array_equal(runs['auto'][3], ref_acc)
    assert np.any(runs['auto'][1][-1][:, 4] > pb['t'].min())   # (walkers behind the first epoch survive to the end)


@pytest.mark.parametrize('nwalkers', [40, 41])
def test_four_kernels_one_chain(nwalkers):
    pb, eng = _multiband()
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(3).standard_normal((nwalkers, 5)))
    runs = {k: _run(eng, nwalkers, 2024, x0, 9, k) for k in ('auto', 'solo', 'fused', 'phases')}
    assert [runs[k][0] for k in ('auto', 'solo', 'fused', 'phases')] == ['run', 'solo', 'fused', 'phases']
    assert runs['auto'][4].last_run_kernel() == 'run' and runs['solo'][4].last_run_kernel() == 'solo'
    for k in ('solo', 'fused', 'phases'):
        assert np.array_equal(runs['auto'][1], runs[k][1]) and np.array_equal(runs['auto'][2], runs[k][2])
        assert np.array_equal(runs['auto'][3], runs[k][3])
    ref, ref_lp, ref_acc = O.stretch_move_run(oracle_log_posterior(pb), x0, 9, 2024)
    assert relerr(runs['auto'][1], ref) < 1e-9 and relerr(runs['auto'][2], ref_lp) < 1e-9
    assert np.array_equal(runs['auto'][3], ref_acc) and 0 < ref_acc.sum() < 9 * nwalkers


def test_resident_workgroups_with_several_slots_each(monkeypatch):
    """k_solo_run with fewer workgroups than a half-step has proposals (LCF_RUN_GRID: as on a device that holds fewer
    than the ensemble needs): every workgroup takes its slots of a half-step one after the other, still waits only for
    rows of earlier half-steps, and the chain is the one of a launch per half-step -- over several launches (45 steps =
    a block of 32 and one of 13) and with the state, the counts and a continued run read back."""
    monkeypatch.setenv('LCF_RUN_GRID', '7')
    pb, eng = _multiband()
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(4).standard_normal((41, 5)))
    ref = _run(eng, 41, 31, x0, 45, 'solo')
    got = _run(eng, 41, 31, x0, 45, 'auto')
    assert got[0] == 'run' and got[4].last_run_kernel() == 'run' and got[4].last_run_launches() == 2
    assert ref[4].last_run_launches() == 90
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    for a, b in zip(got[4].get_state(), ref[4].get_state()):
        assert np.array_equal(a, b)
    for s in (got[4], ref[4]):
        s.run(45, 5, 'random', True)
    assert np.array_equal(got[4].get_chain()[0], ref[4].get_chain()[0]) and np.array_equal(got[4].naccepted(), ref[4].naccepted())


def test_two_runs_in_flight_keep_one_resident_launch_per_device():
    """Two samplers on engines (streams) of their own, both runs enqueued before either is waited for: the resident
    workgroups of two launches could keep each other off the device, so the second run -- enqueued while the first is in
    flight -- takes a launch per half-step; both chains are the ones of runs that had the device to themselves."""
    (pa, ea), (pb_, eb) = _multiband(seed=77), _multiband(seed=78)
    assert ea is not eb
    rng = np.random.default_rng(8)
    xa = pa['truth'] * (1 + 0.05 * rng.standard_normal((64, 5)))
    xb = pb_['truth'] * (1 + 0.05 * rng.standard_normal((64, 5)))
    ref_a, ref_b = _run(ea, 64, 3, xa, 150, 'solo'), _run(eb, 64, 4, xb, 150, 'solo')
    a, b = NativeSampler(ea, 64, 3), NativeSampler(eb, 64, 4)
    a.set_state(xa)
    b.set_state(xb)
    a.run_async(0, 150, 'random', True)
    b.run_async(0, 150, 'random', True)
    a.wait()
    b.wait()
    assert a.last_run_kernel() == 'run' and b.last_run_kernel() == 'solo'
    assert np.array_equal(a.get_chain()[0], ref_a[1]) and np.array_equal(b.get_chain()[0], ref_b[1])
    assert np.array_equal(a.naccepted(), ref_a[3]) and np.array_equal(b.naccepted(), ref_b[3])
    b.run(150, 10, 'random', True)       # (nothing in flight any more: resident workgroups again)
    assert b.last_run_kernel() == 'run'


def test_a_resident_launch_that_cannot_finish_is_repeated_launch_by_launch(monkeypatch, capfd):
    """k_solo_run's workgroups wait for each other, so all of them must be on the device at once.  When they are not
    (LCF_RUN_TEST_MISSING: the last workgroup is never launched -- as when another process's resident kernel holds the
    compute units), a workgroup that has waited LCF_RESIDENT_WAIT_S (50 ms) for a row looks at the count of the launch's
    workgroups that have started, finds one missing and gives up -- long before LCF_PEER_WAIT_S (5 s, the bound for a row
    of a launch that IS complete); the launch has not touched the state it started from, and the library repeats the
    same steps with a launch per half-step: same chain, same counts, one line on stderr, and the sampler stays with
    k_solo afterwards."""
    import time
    pb, eng = _multiband()
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(6).standard_normal((40, 5)))
    ref = _run(eng, 40, 17, x0, 12, 'solo')
    monkeypatch.setenv('LCF_RUN_TEST_MISSING', '1')
    monkeypatch.delenv('LCF_PEER_WAIT_S', raising=False)
    t0 = time.perf_counter()
    got = _run(eng, 40, 17, x0, 12, 'auto')
    assert time.perf_counter() - t0 < 0.5          # (sampler, state, the launch that gives up, the repeat, the chain)
    assert got[0] == 'run' and got[4].last_run_kernel() == 'solo'
    assert 'repeated with a launch per half-step' in capfd.readouterr().err
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    for a, b in zip(got[4].get_state(), ref[4].get_state()):
        assert np.array_equal(a, b)
    monkeypatch.delenv('LCF_RUN_TEST_MISSING')
    for s in (got[4], ref[4]):
        s.run(12, 4, 'random', True)
    assert got[4].last_run_kernel() == 'solo' and got[4].set_half_step_kernel('auto') == 'solo'
    assert np.array_equal(got[4].get_chain()[0], ref[4].get_chain()[0]) and np.array_equal(got[4].naccepted(), ref[4].naccepted())


@pytest.mark.parametrize('kernel', ['auto', 'solo', 'fused', 'phases'])
@pytest.mark.parametrize('nwalkers,randomize', [(11, True), (13, False), (27, True)])
def test_odd_ensembles_follow_emcee_split(nwalkers, randomize, kernel):
    """An odd ensemble: the larger colour (ceil(n / 2) walkers) moves first against the smaller one, as in emcee's
    red-blue move; the last slot of every second half-step stays empty."""
    pb, eng, x0 = _small(nwalkers, seed=nwalkers)
    _, chain, lp, acc, _ = _run(eng, nwalkers, 4242, x0, 7, kernel, 'random' if randomize else 'identity')
    ref, ref_lp, ref_acc = O.stretch_move_run(oracle_log_posterior(pb), x0, 7, 4242, randomize_split=randomize)
    assert relerr(chain, ref) < 1e-9 and relerr(lp, ref_lp) < 1e-9 and np.array_equal(acc, ref_acc)


def test_long_run_crosses_draw_blocks():
    """300 steps = a short first block of draw records, one full block of 256 and a remainder: nothing may change at
    the seams, for the kernel that needs no slot bookkeeping and for the ones that carry it across blocks."""
    pb, eng, x0 = _small(16, seed=8)
    runs = {k: _run(eng, 16, 99, x0, 300, k) for k in ('auto', 'solo', 'fused', 'phases')}
    for k in ('solo', 'fused', 'phases'):
        assert np.array_equal(runs['auto'][1], runs[k][1]) and np.array_equal(runs['auto'][3], runs[k][3])
    ref, ref_lp, ref_acc = O.stretch_move_run(oracle_log_posterior(pb), x0, 300, 99)
    assert relerr(runs['auto'][1], ref) < 1e-8 and np.array_equal(runs['auto'][3], ref_acc)


@pytest.mark.parametrize('kernel', ['auto', 'solo', 'fused'])
def test_runs_that_continue_each_other(kernel):
    """10 + 10 + 5 steps (the second run adopts the draw records the first one left behind for it; the third changes
    the colouring, so what the second one left is discarded) == the same 25 steps of the oracle."""
    pb, eng, x0 = _small(24, seed=5)
    s = NativeSampler(eng, 24, 606)
    s.set_half_step_kernel(kernel)
    s.set_state(x0)
    chains = []
    for first, n, split in ((0, 10, 'random'), (10, 10, 'random'), (20, 5, 'identity')):
        s.run(first, n, split, True)
        chains.append(s.get_chain()[0])
    fn = oracle_log_posterior(pb)
    a, a_lp, _ = O.stretch_move_run(fn, x0, 20, 606)
    b, _, _ = O.stretch_move_run(fn, a[-1], 5, 606, log_prob0=a_lp[-1], randomize_split=False, first_step=20)
    assert relerr(np.concatenate(chains), np.concatenate([a, b])) < 1e-9
    x, lp = s.get_state()
    assert np.array_equal(x, chains[-1][-1])


def test_state_calls_between_and_after_runs():
    """get_state / naccepted are served from the snapshot a run leaves in host memory; set_state and further runs must
    refresh it."""
    pb, eng, x0 = _small(20, seed=2)
    s = NativeSampler(eng, 20, 1)
    s.set_state(x0)
    x, lp = s.get_state()
    assert np.array_equal(x, x0) and relerr(lp, oracle_log_posterior(pb)(x0)) < 1e-11
    s.run(0, 4, 'random', True)
    x1, lp1 = s.get_state()
    assert np.array_equal(x1, s.get_chain()[0][-1]) and np.array_equal(lp1, s.get_chain()[1][-1])
    acc1 = s.naccepted()
    s.run(4, 3, 'random', False)
    x2, _ = s.get_state()
    assert not np.array_equal(x1, x2) and np.all(s.naccepted() >= acc1)
    s.set_state(x0)
    assert np.array_equal(s.get_state()[0], x0) and not s.naccepted().any()


def test_engine_follows_the_light_curve_content():
    """Model.log_likelihood reads the light curve on every call (reference models.py:116-119): editing a column in
    place, or a new light-curve object, must never be answered from an engine built for other photometry."""
    pb = small_problem()
    lc = lc_dict(pb['t'], pb['names'], pb['y'], pb['dy'])
    m = M.ShockCooling(redshift=0.)
    P = pb['truth'] * (1 + 0.02 * np.random.default_rng(0).standard_normal((6, 5)))
    before = m.log_likelihood(lc, P)
    eng = m.engine_for(lc)
    assert m.engine_for(lc) is eng                       # unchanged content: the same engine
    lc['lum'][:] = lc['lum'] * 1.05                      # in-place edit (what calcAbsMag / calcLum do on a refit)
    after = m.log_likelihood(lc, P)
    assert np.all(before != after)
    om = ('ShockCooling', O.ShockCoolingOracle(0.))
    want = O.log_likelihood(om, pb['t'], pb['bands'], lc['lum'], lc['dlum'], P.T)
    assert relerr(after, want) < 1e-11
    # an engine that left the cache stays usable by whoever holds it (samplers, closures)
    s = EnsembleSampler(12, 5, eng, seed=3)
    for k in range(m.max_bound_engines + 2):
        other = lc_dict(pb['t'], pb['names'], pb['y'] * (1 + 0.01 * (k + 1)), pb['dy'])
        m.log_likelihood(other, P)
    assert all(b.engine is not eng for b in m._bound)
    s.run_mcmc(P[:1] * (1 + 0.01 * np.random.default_rng(1).standard_normal((12, 5))), 3)
    assert np.all(np.isfinite(s.get_log_prob()))
    assert relerr(eng.log_likelihood(P), before) == 0.


@pytest.mark.parametrize('ranks', [2, 3])
def test_peer_mailboxes_emulated_ranks(ranks):
    """The collective-free sharded run: `ranks` samplers on engines (= streams) of their own play the ranks of one
    ensemble; each stores the rows of its shard straight into every mailbox and polls its own.  Their launches run
    concurrently on the device; every rank must reproduce the single-GPU chain bit for bit.  (No more than three: a
    process gets four hardware queues, and streams that share one run in order -- a rank would wait for a rank queued
    behind it until the 0.5 s bound.  Real ranks are processes: tools/peer_ranks_check.py.)"""
    pb, eng = _multiband()
    nwalkers, nsteps = 48, 9
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(4).standard_normal((nwalkers, 5)))
    _, want_chain, want_lp, want_acc, _ = _run(eng, nwalkers, 321, x0, nsteps, 'auto')
    priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
    lc = lc_dict(pb['t'], [b.name for b in pb['bands']], pb['y'], pb['dy'])
    engines = [M.ShockCooling(redshift=0.004).engine_for(lc, priors=priors) for _ in range(ranks)]
    assert len({id(e) for e in engines}) == ranks
    samplers = [NativeSampler(e, nwalkers, 321) for e in engines]
    ptrs = [s.mailbox_export()[1] for s in samplers]
    for r, s in enumerate(samplers):
        s.mailbox_connect(ranks, r, local_ptrs=ptrs)
        # ranks emulated in ONE process share its host thread: an allocation (or free) made while another rank's
        # launch waits for this rank would hold the launches back beyond the 0.5 s bound.  Size every buffer first.
        s.set_state(x0)
        s.run(100, nsteps, 'random', True)
        s.set_state(x0)
    for first, n in ((0, 4), (4, nsteps - 4)):          # two runs: the generations continue across runs
        for s in samplers:
            s.run_peers(first, n, 'random', True, asynchronous=True)
        for s in samplers:
            s.wait()
    for s in samplers:
        chain, lp = s.get_chain()
        assert np.array_equal(chain, want_chain[4:]) and np.array_equal(lp, want_lp[4:])
        assert np.array_equal(s.naccepted(), want_acc)


@pytest.mark.parametrize('form', ['resident', 'per half-step'])
@pytest.mark.parametrize('ranks,nwalkers,split', [(2, 48, 'random'), (3, 54, 'random'), (2, 44, 'identity'), (3, 42, 'random')])
def test_row_boards_emulated_ranks(ranks, nwalkers, split, form):
    """The sharded run in which nothing is replicated: every emulated rank moves ITS share of the walkers -- with resident
    workgroups (k_solo_run<..., RANKS>: a launch per block of half-steps) or with a k_solo launch per half-step -- and
    posts their rows (position, log-posterior, acceptance count) on all boards; state, counts and chain are complete on
    every rank at the end and equal the single-GPU run bit for bit -- also across two runs that continue each other,
    with a fixed colouring, and with 21 slots per half-step shared by 3 ranks (42 walkers)."""
    pb, eng = _multiband()
    nsteps = 9
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(4).standard_normal((nwalkers, 5)))
    _, want_chain, want_lp, want_acc, ref = _run(eng, nwalkers, 321, x0, nsteps, 'auto', split)
    want_x, want_lp_end = ref.get_state()
    priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
    lc = lc_dict(pb['t'], [b.name for b in pb['bands']], pb['y'], pb['dy'])
    engines = [M.ShockCooling(redshift=0.004).engine_for(lc, priors=priors) for _ in range(ranks)]
    samplers = [NativeSampler(e, nwalkers, 321) for e in engines]
    ptrs = [s.board_export()[1] for s in samplers]
    for r, s in enumerate(samplers):
        s.board_connect(ranks, r, local_ptrs=ptrs)
        s.set_state(x0)                       # (size every buffer first: see the mailbox test above)
        s.run(100, nsteps, split, True)
        s.set_state(x0)
        s.set_half_step_kernel('auto' if form == 'resident' else 'solo')
    for first, n in ((0, 4), (4, nsteps - 4)):
        for s in samplers:
            s.run_rows(first, n, split, True, asynchronous=True)
        for s in samplers:
            s.wait()
            assert s.last_run_kernel() == ('run' if form == 'resident' else 'solo')
    for s in samplers:
        chain, lp = s.get_chain()
        assert np.array_equal(chain, want_chain[4:]) and np.array_equal(lp, want_lp[4:])
        assert np.array_equal(s.naccepted(), want_acc)
        x, lp_end = s.get_state()
        assert np.array_equal(x, want_x) and np.array_equal(lp_end, want_lp_end)


@pytest.mark.parametrize('ranks,nwalkers,block,grid', [(2, 40, None, None), (3, 36, '5', None), (2, 44, None, '4')])
def test_resident_row_boards_across_launches_and_draw_blocks(ranks, nwalkers, block, grid, monkeypatch):
    """A row-board run of 150 steps with resident workgroups: five launches per rank (32 steps each; every launch from the
    third on waits for the other ranks' progress words), the seam between the first block of draw records and the next --
    and, with blocks of 5 steps, thirty launches of ten half-steps; with four resident workgroups per rank (LCF_RUN_GRID)
    for its eleven slots of a half-step, every workgroup takes two or three of them in turn.  Chain, state and counts of
    every rank equal the single-GPU run's."""
    if block:
        monkeypatch.setenv('LCF_DRAW_BLOCK', block)
    if grid:
        monkeypatch.setenv('LCF_RUN_GRID', grid)
    pb, eng = _multiband()
    nsteps = 150
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(14).standard_normal((nwalkers, 5)))
    _, want_chain, want_lp, want_acc, ref = _run(eng, nwalkers, 99, x0, nsteps, 'solo')
    priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
    lc = lc_dict(pb['t'], [b.name for b in pb['bands']], pb['y'], pb['dy'])
    engines = [M.ShockCooling(redshift=0.004).engine_for(lc, priors=priors) for _ in range(ranks)]
    samplers = [NativeSampler(e, nwalkers, 99) for e in engines]
    ptrs = [s.board_export()[1] for s in samplers]
    for r, s in enumerate(samplers):
        s.board_connect(ranks, r, local_ptrs=ptrs)
        s.set_state(x0)
        s.run(1000, nsteps, 'random', True)   # (size every buffer first)
        s.set_state(x0)
    for s in samplers:
        s.run_rows(0, nsteps, 'random', True, asynchronous=True)
    for s in samplers:
        s.wait()
        assert s.last_run_kernel() == 'run' and s.last_run_launches() == (30 if block else 5)
    for s in samplers:
        chain, lp = s.get_chain()
        assert np.array_equal(chain, want_chain) and np.array_equal(lp, want_lp)
        assert np.array_equal(s.naccepted(), want_acc)
        for a, b in zip(s.get_state(), ref.get_state()):
            assert np.array_equal(a, b)


@pytest.mark.parametrize('ranks', [2, 3])
def test_peer_mailboxes_between_processes(ranks):
    """Real processes, HIP IPC: tools/peer_ranks_check.py (every rank == the single-GPU chain, mailboxes connected)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'peer_ranks_check.py'), str(ranks), '48', '10'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['every_rank_equals_the_single_gpu_chain'] and line['peer_mailboxes_connected_on_every_rank']


@pytest.mark.parametrize('ranks,steps', [(2, 10), (3, 140)])
def test_resident_row_boards_between_processes(ranks, steps):
    """Real processes, boards mapped through HIP IPC, every rank's workgroups resident for up to 32 steps
    (k_solo_run<..., RANKS>; 140 steps in two runs of 70: three launches each, the third behind the other ranks' progress
    words): every rank ends with the single-GPU chain, state and counts."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'LCF_COLLECTIVE')}
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'peer_ranks_check.py'), str(ranks), '48', str(steps), 'rows'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['every_rank_equals_the_single_gpu_chain'] and line['connected_on_every_rank']
    assert line['half_step_kernel_of_the_last_run'] == ['run'] * ranks
    assert line['its_launches'] == [1 if steps == 10 else 3] * ranks


def test_collective_none_probes_and_picks_a_driver_between_processes():
    """EnsembleSampler(collective='auto') with two real processes: the first run probes rows / peers / all-gather (the ranks
    agree on the fastest one that works), the chain is the single-GPU chain whichever won, and a second sampler of the
    same process group reuses the choice without probing again."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'LCF_COLLECTIVE')}
    env['LCF_PEER_WAIT_S'] = '0.5'     # (two ranks that share a device may starve each other in the row-board probe)
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'peer_ranks_check.py'), '2', '48', '10', 'auto'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['requested'] == 'auto' and line['driver'] == line['probe']['selected']
    assert line['every_rank_equals_the_single_gpu_chain'] and line['connected_on_every_rank']
    assert line['probe'][line['driver']]['ok'] and line['probe']['peers']['replicas_agree']


@pytest.mark.parametrize('nwalkers', [21, 64])
def test_population_one_launch_per_half_step(nwalkers, monkeypatch):
    """Population mode with shared epochs: the transients' workgroups stay for blocks of half-steps and hand each other
    rows through the transients' boards (k_pop_run), or -- LCF_NO_POP_RUN=1 -- ONE launch per half-step for all transients
    (k_pop: a workgroup per four proposals, accept test included).  Their chains are bitwise those of the two-launch path
    (LCF_NO_POP=1), and the oracle-driven chain of a transient; 21 walkers = 11 slots per half-step: a partly filled last
    workgroup and, in every second half-step, an empty slot.  Resident launches also with blocks of 2 steps of draw
    records (several launches per run) and with 2 workgroups per transient (several groups of proposals per workgroup)."""
    from lightcurve_fitting_amd.sampler import PopulationSampler
    problems, x0, pbs = [], {}, []
    for k in range(4):
        rng = np.random.default_rng(300 + k)
        epochs = np.sort(rng.uniform(0.4, 9., 60 + 11 * k))
        t, names = np.repeat(epochs, 6), list(np.tile(list('UBVgri'), len(epochs)))
        bands = [O.band(n) for n in names]
        truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1]) * rng.uniform(0.9, 1.1, 5)
        om = ('ShockCooling', O.ShockCoolingOracle(0.))
        ytrue = O.evaluate(om, t, bands, truth)
        y, dy = ytrue * (1 + 0.05 * rng.standard_normal(len(t))), 0.05 * ytrue
        priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
        pbs.append(dict(model=om, orc=None, t=t, bands=bands, y=y, dy=dy, priors=[p.descriptor() for p in priors]))
        problems.append((M.ShockCooling(redshift=0.), lc_dict(t, names, y, dy), priors))
        x0[k] = truth * (1 + 0.05 * rng.standard_normal((nwalkers, 5)))
    chains = {}
    forms = {'population-run': {}, 'population': {'LCF_NO_POP_RUN': '1'}, 'population-phases': {'LCF_NO_POP': '1'},
             'population-run, blocks of 2 steps': {'LCF_DRAW_BLOCK': '2'},
             'population-run, 8 workgroups': {'LCF_RUN_GRID': '8'},
             'population-run, interpolants from L2': {'LCF_POP_ITAB_LDS': '0'}}
    for form, env in forms.items():
        for name in ('LCF_NO_POP_RUN', 'LCF_NO_POP', 'LCF_DRAW_BLOCK', 'LCF_RUN_GRID', 'LCF_POP_ITAB_LDS'):
            monkeypatch.delenv(name, raising=False)
        for name, value in env.items():
            monkeypatch.setenv(name, value)
        pop = PopulationSampler(problems, nwalkers, seed=17)
        pop.run_mcmc(x0, 6)
        pop.run_mcmc(None, 3)
        assert pop[0]._native.last_run_kernel() == form.split(',')[0]
        if 'blocks' in form:
            assert pop[0]._native.last_run_launches() == 2
        chains[form] = [(pop[k].get_chain(), pop[k].get_log_prob(), pop[k].acceptance_fraction) for k in range(4)]
    for form in forms:
        for a, b in zip(chains[form], chains['population-phases']):
            assert all(np.array_equal(u, v) for u, v in zip(a, b)), form
    ref, ref_lp, _ = O.stretch_move_run(oracle_log_posterior(pbs[2]), x0[2], 9, 17 + 2)
    assert relerr(chains['population-run'][2][0], ref) < 1e-9 and relerr(chains['population-run'][2][1], ref_lp) < 1e-9


@pytest.mark.parametrize('shape', ['ShockCooling2', 'ShockCooling2 + sigma', 'ShockCooling + sigma'])
def test_population_resident_other_models(shape, monkeypatch):
    """k_pop_run for the other model-specialised kernel (ShockCooling2, four parameters) and for the generic one (a fitted
    sigma: walker dimension at run time), odd ensembles, three and four filters: the chains of the two-launch path."""
    from lightcurve_fitting_amd.sampler import PopulationSampler
    two, sigma = shape.startswith('ShockCooling2'), shape.endswith('sigma')
    truth = np.array([30., 3., 30., 0.2]) if two else np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    pri = ([M.UniformPrior(0., 100.)] * 3 + [M.UniformPrior(-1., 0.29)]) if two else \
        ([M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.29)])
    if sigma:
        pri = pri + [M.UniformPrior(0., 5.)]
    problems, x0 = [], {}
    for k in range(3):
        rng = np.random.default_rng(800 + k)
        filts = list('BVgr')[:3 + k % 2]
        epochs = np.sort(rng.uniform(0.4, 20., 30 + 17 * k))
        t, names = np.repeat(epochs, len(filts)), list(np.tile(filts, len(epochs)))
        m = M.ShockCooling2(redshift=0.01) if two else M.ShockCooling(redshift=0.01)
        y = m(t, names, *truth) * (1 + 0.05 * rng.standard_normal(len(t)))
        problems.append((m, lc_dict(t, names, y, 0.05 * np.abs(y)), pri) + (({'use_sigma': True},) if sigma else ()))
        x0[k] = np.concatenate([truth, [0.5]] if sigma else [truth]) * (1 + 0.03 * rng.standard_normal((27, len(pri))))
    out = {}
    for form in ('population-run', 'population-phases'):
        if form == 'population-phases':
            monkeypatch.setenv('LCF_NO_POP', '1')
        pop = PopulationSampler(problems, 27, seed=41)
        pop.run_mcmc(x0, 7)
        pop.run_mcmc(None, 2)
        assert pop[0]._native.last_run_kernel() == form
        out[form] = [(pop[k].get_chain(), pop[k].get_log_prob(), pop[k].acceptance_fraction) for k in range(3)]
    for a, b in zip(out['population-run'], out['population-phases']):
        assert all(np.array_equal(u, v) for u, v in zip(a, b))


def test_population_resident_runs_mix_with_single_runs(monkeypatch):
    """A transient's sampler shares its board of tagged rows, its second set of state buffers and its count of started
    workgroups between the population's resident launches (k_pop_run) and its own (k_solo_run): population run ->
    every ensemble on its own stream -> population run again gives the chains of three population runs with a launch per
    half-step."""
    from lightcurve_fitting_amd.sampler import PopulationSampler
    problems, x0 = [], {}
    for k in range(3):
        rng = np.random.default_rng(700 + k)
        epochs = np.sort(rng.uniform(0.4, 9., 40 + 9 * k))
        t, names = np.repeat(epochs, 6), list(np.tile(list('UBVgri'), len(epochs)))
        truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
        m = M.ShockCooling(redshift=0.)
        y = m(t, names, *truth) * (1 + 0.05 * rng.standard_normal(len(t)))
        problems.append((m, lc_dict(t, names, y, 0.05 * np.abs(y)), [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]))
        x0[k] = truth * (1 + 0.05 * rng.standard_normal((34, 5)))
    out = {}
    for form in ('mixed', 'reference'):
        if form == 'reference':
            monkeypatch.setenv('LCF_NO_POP_RUN', '1')
        pop = PopulationSampler(problems, 34, seed=23)
        pop.run_mcmc(x0, 5)
        pop.run_mcmc(None, 4, batched=(form == 'reference'))
        pop.run_mcmc(None, 3)
        assert pop[0]._native.last_run_kernel() == ('population' if form == 'reference' else 'population-run')
        out[form] = [(pop[k].get_chain(), pop[k].get_log_prob(), pop[k].acceptance_fraction) for k in range(3)]
    for a, b in zip(out['mixed'], out['reference']):
        assert a[0].shape[0] == 12 and all(np.array_equal(u, v) for u, v in zip(a, b))


def test_population_resident_launch_that_is_not_all_there_falls_back(monkeypatch):
    """A resident population launch one of whose workgroups never starts (LCF_RUN_TEST_MISSING: what another process's
    resident kernel on the same GPU does to it) gives up within the residency bound, leaves the transients' states
    untouched, and the same steps run with a launch per half-step: the same chains, well inside a second."""
    import subprocess
    import sys
    import os
    code = """
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from lightcurve_fitting_amd import models as M
from lightcurve_fitting_amd.sampler import PopulationSampler
problems, x0 = [], {}
for k in range(3):
    rng = np.random.default_rng(900 + k)
    epochs = np.sort(rng.uniform(0.4, 9., 50))
    t, names = np.repeat(epochs, 6), list(np.tile(list('UBVgri'), 50))
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    m = M.ShockCooling(redshift=0.)
    y = m(t, names, *truth) * (1 + 0.05 * rng.standard_normal(len(t)))
    problems.append((m, {'MJD': t, 'filter': names, 'lum': y, 'dlum': 0.05 * np.abs(y)},
                     [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]))
    x0[k] = truth * (1 + 0.05 * rng.standard_normal((32, 5)))
os.environ['LCF_NO_POP_RUN'] = '1'
ref = PopulationSampler(problems, 32, seed=5)
ref.run_mcmc(x0, 5)
del os.environ['LCF_NO_POP_RUN']
os.environ['LCF_RUN_TEST_MISSING'] = '1'
pop = PopulationSampler(problems, 32, seed=5)
t0 = time.time()
pop.run_mcmc(x0, 5)
dt = time.time() - t0
assert pop[0]._native.last_run_kernel() == 'population', pop[0]._native.last_run_kernel()
for k in range(3):
    assert np.array_equal(pop[k].get_chain(), ref[k].get_chain())
    assert np.array_equal(pop[k].acceptance_fraction, ref[k].acceptance_fraction)
del os.environ['LCF_RUN_TEST_MISSING']
pop.run_mcmc(None, 2)      # (later population runs of the process: a launch per half-step)
assert pop[0]._native.last_run_kernel() == 'population'
print('OK %%.2f' %% dt)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith('OK'), (out.stdout[-1500:], out.stderr[-3000:])
    assert float(out.stdout.split()[1]) < 1.0, out.stdout
    assert 'resident population launch gave up' in out.stderr


def test_row_boards_companion_shape():
    """The companion model of configs[2] -- 8 parameters, 8000 points in four parts -- at launches of at most one
    workgroup per CU: 1024-thread workgroups, all four parts of a proposal side by side.  One chain, bit for bit, from
    k_solo<8, 1, true, 8> (a launch per half-step), from the resident form k_solo_run<8, 1, true, 8> on one GPU, and from
    two emulated ranks of a row-board run in either form."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    model, lc, priors, _ = bench.build_companion(0)
    nwalkers, nsteps = 40, 4
    x0 = bench.companion_walkers(nwalkers)
    eng = model.engine_for(lc, priors=priors)
    ref = NativeSampler(eng, nwalkers, 9)
    assert ref.set_half_step_kernel('solo') == 'solo'
    ref.set_state(x0)
    ref.run(0, nsteps, 'random', True)
    want_chain, want_lp = ref.get_chain()
    one = NativeSampler(eng, nwalkers, 9)
    assert one.set_half_step_kernel('auto') == 'run'    # (four parts and 20 proposals: resident 1024-thread workgroups)
    one.set_state(x0)
    one.run(0, nsteps, 'random', True)
    assert one.last_run_kernel() == 'run'
    chain, lp = one.get_chain()
    assert np.array_equal(chain, want_chain) and np.array_equal(lp, want_lp) and np.array_equal(one.naccepted(), ref.naccepted())
    for form in ('resident', 'per half-step'):
        # (a model keeps one engine per light curve: two models for the two ranks' engines)
        engines = [bench.build_companion(0)[0].engine_for(lc, priors=priors) for _ in range(2)]
        assert engines[0] is not engines[1]
        samplers = [NativeSampler(e, nwalkers, 9) for e in engines]
        ptrs = [s.board_export()[1] for s in samplers]
        for r, s in enumerate(samplers):
            s.board_connect(2, r, local_ptrs=ptrs)
            s.set_state(x0)
            s.run(100, nsteps, 'random', True)
            s.set_state(x0)
            s.set_half_step_kernel('auto' if form == 'resident' else 'solo')
        for s in samplers:
            s.run_rows(0, nsteps, 'random', True, asynchronous=True)
        for s in samplers:
            s.wait()
            assert s.last_run_kernel() == ('run' if form == 'resident' else 'solo')
        for s in samplers:
            chain, lp = s.get_chain()
            assert np.array_equal(chain, want_chain) and np.array_equal(lp, want_lp)
            assert np.array_equal(s.naccepted(), ref.naccepted())


def test_row_boards_companion_several_slots_per_workgroup(monkeypatch):
    """Ranks whose share exceeds one proposal per CU (configs[2] on two GPUs: 1024 proposals per rank) take the resident
    form with 512-thread workgroups (k_solo_run<8, 1, true, 4, 0, RANKS>) and several slots of a half-step per workgroup:
    two emulated ranks of 260 proposals each on 100 workgroups each, the chain of the single-GPU run bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    model, lc, priors, _ = bench.build_companion(0)
    nwalkers, nsteps = 1040, 3
    x0 = bench.companion_walkers(nwalkers)
    ref = NativeSampler(model.engine_for(lc, priors=priors), nwalkers, 9)
    assert ref.set_half_step_kernel('solo') == 'solo'
    ref.set_state(x0)
    ref.run(0, nsteps, 'random', True)
    want_chain, want_lp = ref.get_chain()
    monkeypatch.setenv('LCF_RUN_GRID', '100')
    engines = [bench.build_companion(0)[0].engine_for(lc, priors=priors) for _ in range(2)]
    samplers = [NativeSampler(e, nwalkers, 9) for e in engines]
    ptrs = [s.board_export()[1] for s in samplers]
    for r, s in enumerate(samplers):
        s.board_connect(2, r, local_ptrs=ptrs)
        s.set_state(x0)
        s.run(100, nsteps, 'random', True)
        s.set_state(x0)
    for s in samplers:
        s.run_rows(0, nsteps, 'random', True, asynchronous=True)
    for s in samplers:
        s.wait()
        assert s.last_run_kernel() == 'run' and s.last_run_launches() == 1
    for s in samplers:
        chain, lp = s.get_chain()
        assert np.array_equal(chain, want_chain) and np.array_equal(lp, want_lp)
        assert np.array_equal(s.naccepted(), ref.naccepted())


@pytest.mark.parametrize('form', ['resident', 'per half-step'])
def test_row_boards_missing_rank_ends_with_an_error(monkeypatch, form):
    """A rank that never runs: the waits of the other one are bounded (LCF_PEER_WAIT_S, here 0.5 s; 5 s by default), the
    run ends with LCF_ERR_STATE, says which row was missing and that the ensemble must be set again -- it does not hang
    the device."""
    import time
    from lightcurve_fitting_amd.engine import LcfError
    monkeypatch.setenv('LCF_PEER_WAIT_S', '0.5')       # (read when a sampler is created)
    pb, eng = _multiband()
    nwalkers = 48
    x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(4).standard_normal((nwalkers, 5)))
    priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
    lc = lc_dict(pb['t'], [b.name for b in pb['bands']], pb['y'], pb['dy'])
    engines = [M.ShockCooling(redshift=0.004).engine_for(lc, priors=priors) for _ in range(2)]
    samplers = [NativeSampler(e, nwalkers, 321) for e in engines]
    ptrs = [s.board_export()[1] for s in samplers]
    for r, s in enumerate(samplers):
        s.board_connect(2, r, local_ptrs=ptrs)
        s.set_state(x0)
        s.set_half_step_kernel('auto' if form == 'resident' else 'solo')
    t0 = time.perf_counter()
    with pytest.raises(LcfError) as err:
        samplers[0].run_rows(0, 6, 'random', True)     # rank 1 never starts
    assert time.perf_counter() - t0 < 5.
    assert samplers[0].last_run_kernel() == ('run' if form == 'resident' else 'solo')
    assert err.value.status == 7 and 'was not posted within 0.5 s' in str(err.value) and 'set_state is required' in str(err.value)
    # the sampler is usable again after a new set_state
    samplers[0].set_state(x0)
    samplers[0].run(0, 3, 'random', True)
    assert np.all(np.isfinite(samplers[0].get_chain()[0]))


@pytest.mark.parametrize('shared_only', [False, True])
def test_light_curve_without_shared_epochs(shared_only, monkeypatch):
    """Real multi-band photometry has no two observations at one time.  By default such a light curve still gets its
    thermal states ahead of the point loop (one "epoch" per point) and with them the fast path -- k_solo, log-space
    states, interpolated band sums; with LCF_SHARED_EPOCHS_ONLY=1 (the round-1 rule) the state is computed inside the
    point loop (k_fused, sample tables).  Both are the oracle-driven chain."""
    if shared_only:
        monkeypatch.setenv('LCF_SHARED_EPOCHS_ONLY', '1')
    pb, eng, x0 = _small(26, seed=12)
    assert len(np.unique(pb['t'])) == len(pb['t'])
    runs = {k: _run(eng, 26, 77, x0, 6, k) for k in ('auto', 'solo', 'fused', 'phases')}
    assert runs['auto'][0] == ('fused' if shared_only else 'run') and runs['solo'][0] == ('fused' if shared_only else 'solo')
    for k in ('solo', 'fused', 'phases'):
        assert np.array_equal(runs['auto'][1], runs[k][1]) and np.array_equal(runs['auto'][3], runs[k][3])
    ref, ref_lp, ref_acc = O.stretch_move_run(oracle_log_posterior(pb), x0, 6, 77)
    assert relerr(runs['auto'][1], ref) < 1e-9 and relerr(runs['auto'][2], ref_lp) < 1e-9
    assert np.array_equal(runs['auto'][3], ref_acc)
