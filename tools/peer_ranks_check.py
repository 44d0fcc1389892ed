#!/usr/bin/env python3
"""N real processes share the box's one GPU and run ONE ensemble over peer mailboxes: each rank maps the others'
mailboxes through HIP IPC, stores its shard's rows straight into them and polls its own -- the native multi-rank run
with no collective (what xGMI peer writes do on a node; here the peers' memory is the same device).  Every rank must
end with the chain of the single-GPU run, bit for bit.

    python tools/peer_ranks_check.py [n_ranks=2] [n_walkers=64] [n_steps=12] [peers|rows|auto]

`rows`: the same check for the row boards (every rank moves its share of the walkers itself and posts their rows).
`auto`: EnsembleSampler(collective='auto') -- the sampler probes the three drivers on its first run (the ranks agree on the
fastest one that works) and runs with it; the line says which.

Started without a launcher it spawns the ranks itself (fresh processes; the parent never touches the GPU); rank 0
prints one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    n_ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_walkers = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    driver = sys.argv[4] if len(sys.argv) > 4 else 'peers'
    if 'WORLD_SIZE' not in os.environ:
        import bench
        port = bench.free_port()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR='127.0.0.1',
                                           MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0'))
                 for r in range(n_ranks)]
        sys.exit(max(abs(p.wait()) for p in procs))
    import numpy as np
    import torch
    import torch.distributed as dist
    from helpers import lc_dict
    from lightcurve_fitting_amd import models as M
    from lightcurve_fitting_amd.sampler import EnsembleSampler
    from oracle import lcf_oracle as O
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=n_ranks)
    rng = np.random.default_rng(77)
    epochs = np.sort(rng.uniform(0.4, 9., 110))
    t, names = np.repeat(epochs, 6), list(np.tile(list('UBVgri'), 110))
    bands = [O.band(n) for n in names]
    truth = np.array([1.2, 0.5, 3.0, 2.0, 0.1])
    ytrue = O.evaluate(('ShockCooling', O.ShockCoolingOracle(0.004)), t, bands, truth)
    lc = lc_dict(t, names, ytrue * (1 + 0.05 * rng.standard_normal(len(t))), 0.05 * ytrue)
    priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
    eng = M.ShockCooling(redshift=0.004).engine_for(lc, priors=priors)
    x0 = truth * (1 + 0.05 * rng.standard_normal((n_walkers, 5)))
    ref = EnsembleSampler(n_walkers, 5, eng, seed=2024, group=None, collective='allgather')
    ref._distributed = lambda: False          # the single-GPU run of the same ensemble
    ref.run_mcmc(x0, n_steps)
    s = EnsembleSampler(n_walkers, 5, eng, seed=2024, collective=driver)
    s.run_mcmc(x0, n_steps // 2)
    probe = s.collective_probe
    if driver == 'auto':
        assert s.collective in ('rows', 'peers', 'allgather') and probe is not None and probe['selected'] == s.collective
        driver = s.collective
        t = EnsembleSampler(n_walkers, 5, eng, seed=1, collective='auto')   # a second sampler of the group: no new probe
        t.run_mcmc(x0, 2)
        assert t.collective == s.collective and t.collective_probe is None
    s.run_mcmc(None, n_steps - n_steps // 2)   # a second run: generations continue, the barrier between runs
    same = bool(np.array_equal(s.get_chain(), ref.get_chain()) and np.array_equal(s.get_log_prob(), ref.get_log_prob()) and
                np.array_equal(s.acceptance_fraction, ref.acceptance_fraction) and
                np.array_equal(s._state[0], ref._state[0]))
    flags = [None] * n_ranks
    connected = s._boards if driver == 'rows' else s._peers if driver == 'peers' else True
    dist.all_gather_object(flags, (same, bool(connected), float(s.last_run_ms), s._native.last_run_kernel(),
                                   s._native.last_run_launches()))
    if rank == 0:
        print(json.dumps({'ranks': n_ranks, 'walkers': n_walkers, 'steps': n_steps,
                          'every_rank_equals_the_single_gpu_chain': all(f[0] for f in flags),
                          'driver': driver, 'requested': sys.argv[4] if len(sys.argv) > 4 else 'peers',
                          'probe': probe, 'connected_on_every_rank': all(f[1] for f in flags),
                          'peer_mailboxes_connected_on_every_rank': driver == 'peers' and all(f[1] for f in flags),
                          'device_ms_last_run': [f[2] for f in flags],
                          'half_step_kernel_of_the_last_run': [f[3] for f in flags], 'its_launches': [f[4] for f in flags],
                          'acceptance': float(s.acceptance_fraction.mean())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if same else 1)


if __name__ == '__main__':
    main()
