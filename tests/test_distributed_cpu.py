"""The multi-rank half-step protocol (propose -> evaluate shard -> all-gather -> accept) on CPU with gloo,
world_size 2 and 3 (uneven shards), against the single-process oracle sampler.  CPU only."""
import os
import socket

import numpy as np
import pytest

from helpers import OracleBackend, oracle_log_posterior, small_problem


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nwalkers, nsteps, seed, out_dir):
    import torch.distributed as dist
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    from lightcurve_fitting_amd import rng, sampler as S
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pb = small_problem()
        r = np.random.default_rng(1)
        x0 = pb['truth'] * (1 + 0.05 * r.standard_normal((nwalkers, 5)))
        be = OracleBackend(oracle_log_posterior(pb), x0, seed)
        split = rng.split_permutations(seed, 0, nsteps, nwalkers) if rank % 2 else 'random'
        S.ShardedStretchDriver(be).run(0, nsteps, split, True)
        np.save(os.path.join(out_dir, f'chain_{rank}.npy'), be.chain)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,nwalkers', [(2, 14), (3, 14), (2, 16)])  # uneven shards, and the in-place gather
def test_sharded_chain_equals_single_process(tmp_path, world, nwalkers):
    import torch.multiprocessing as mp
    from oracle import lcf_oracle as O
    nsteps, seed = 6, 2024
    mp.spawn(_worker, args=(world, _free_port(), nwalkers, nsteps, seed, str(tmp_path)), nprocs=world, join=True)
    chains = [np.load(tmp_path / f'chain_{r}.npy') for r in range(world)]
    for c in chains[1:]:
        assert np.array_equal(c, chains[0])  # every rank holds the identical ensemble
    pb = small_problem()
    r = np.random.default_rng(1)
    x0 = pb['truth'] * (1 + 0.05 * r.standard_normal((nwalkers, 5)))
    ref, _, _ = O.stretch_move_run(oracle_log_posterior(pb), x0, nsteps, seed)
    assert np.array_equal(chains[0], ref)  # ... which is the single-process chain, bit for bit
    assert not np.array_equal(ref[0], ref[-1])
