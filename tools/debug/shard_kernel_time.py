"""Device time of ONE rank's launches of a sharded run (k_fused for its own slots + one wave per foreign slot) against
the world size, weak scaling (1024 walkers per rank): what the replicated bookkeeping of the other ranks' proposals
costs.  Nobody else runs: the rows of the other shards are filled once with a huge chi^2, so every foreign proposal is
rejected and this rank's own proposals stay ordinary ones."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd.engine import NativeSampler


def alias(ptr, shape):
    class _Alias:
        __cuda_array_interface__ = {'shape': shape, 'typestr': '<f8', 'data': (ptr, False), 'version': 2, 'strides': None}
    return torch.as_tensor(_Alias(), device='cuda:0')


model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
for world in (1, 2, 4, 8):
    nw = 1024 * world
    s = NativeSampler(eng, nw, 3)
    s.set_state(bench.initial_walkers(nw))
    steps = 200
    s.begin(0, steps, 'random', False)
    st = torch.cuda.Stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_half = nw // 2
    w = n_half // world
    with torch.cuda.stream(st):
        for k in range(2):   # both parity buffers exist after two half-steps: poison the foreign rows of each
            s.half_step_rows(k // 2, k % 2, 0, w, st.cuda_stream)
            ptr, rows = s.rows_ptr()
            if world > 1:
                t = alias(ptr, (n_half, rows))
                t[w:, :rows - 1] = 1e30
                t[w:, rows - 1] = 0.
        for k in range(2, 20):
            s.half_step_rows(k // 2, k % 2, 0, w, st.cuda_stream)
        a.record(st)
        for k in range(20, 2 * steps):
            s.half_step_rows(k // 2, k % 2, 0, w, st.cuda_stream)
        b.record(st)
    st.synchronize()
    x, lp = s.get_state()
    print(f'world {world}: {a.elapsed_time(b) / (2 * steps - 20) * 1e3:.2f} us per half-step launch (own {w} slots of {n_half}); '
          f'finite log-posteriors: {np.isfinite(lp).mean():.2f}, acceptance so far {s.naccepted().sum() / (nw / world * steps):.2f}', flush=True)
    s.close()
