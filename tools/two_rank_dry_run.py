"""Dry run of the multi-rank driver on ONE GPU: 2 processes share cuda:0, gloo for the collectives (RCCL refuses two
ranks on one device, which also exercises the agreed fallback from the native communicator).  Standalone on purpose:
it must be started from a process that has NOT touched the GPU (spawning re-execs the interpreter), so it is not a
pytest case.  Usage on the GPU box:  python tools/two_rank_dry_run.py [world=2] [nwalkers=48]
(3 ranks with 50 walkers: unequal shards, the padded staging path of the driver)."""
import os, sys, socket
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np

WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 48


def worker(rank, world, port, out):
    import torch, torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from test_gpu_sampler import _setup
    from lightcurve_fitting_amd.sampler import EnsembleSampler
    pb, lc, m, eng, x0 = _setup(NW)
    s = EnsembleSampler(NW, 5, eng, seed=99)
    s.run_mcmc(x0, 7)
    print(rank, 'native comm:', s._comm, flush=True)
    np.save(f'{out}/chain_{rank}.npy', s.get_chain())
    dist.destroy_process_group()

if __name__ == '__main__':
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    out = 'gpurun_out'
    os.makedirs(out, exist_ok=True)
    mp.spawn(worker, args=(WORLD, port, out), nprocs=WORLD, join=True)
    chains = [np.load(f'{out}/chain_{r}.npy') for r in range(WORLD)]
    from test_gpu_sampler import _setup
    from lightcurve_fitting_amd.sampler import EnsembleSampler
    pb, lc, m, eng, x0 = _setup(NW)
    ref = EnsembleSampler(NW, 5, eng, seed=99); ref.run_mcmc(x0, 7)
    print(f'{WORLD} ranks, {NW} walkers: ranks equal', all(np.array_equal(chains[0], c) for c in chains[1:]),
          'equal to single-GPU', np.array_equal(chains[0], ref.get_chain()))
