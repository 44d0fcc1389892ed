"""Peer-mailbox protocol with 1 emulated rank (protocol only), then 2 (concurrency of two streams)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from helpers import lc_dict, small_problem  # noqa: E402
from lightcurve_fitting_amd import models as M  # noqa: E402
from lightcurve_fitting_amd.engine import NativeSampler  # noqa: E402

pb = small_problem()
lc = lc_dict(pb['t'], pb['names'], pb['y'], pb['dy'])
priors = [M.UniformPrior(0., 10.)] * 4 + [M.UniformPrior(-1., 0.5)]
x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(1).standard_normal((32, 5)))
ref = NativeSampler(M.ShockCooling(redshift=0.).engine_for(lc, priors=priors), 32, 5)
ref.set_state(x0)
ref.run(0, 6, 'random', True)
for ranks in (1, 2):
    engines = [M.ShockCooling(redshift=0.).engine_for(lc, priors=priors) for _ in range(ranks)]
    samplers = [NativeSampler(e, 32, 5) for e in engines]
    ptrs = [s.mailbox_export()[1] for s in samplers]
    for r, s in enumerate(samplers):
        s.mailbox_connect(ranks, r, local_ptrs=ptrs)
        s.set_state(x0)
    try:
        for s in samplers:
            s.run_peers(0, 6, 'random', True, asynchronous=True)
        for s in samplers:
            s.wait()
        print(ranks, 'ranks: equal to the single-GPU chain:',
              [bool(np.array_equal(s.get_chain()[0], ref.get_chain()[0])) for s in samplers])
    except Exception as exc:  # noqa: BLE001
        print(ranks, 'ranks: FAILED', exc)

# diagnostics: the tags in every rank's mailbox after the (failed) 2-rank run
import torch  # noqa: E402


def view(ptr, n):
    class _A:
        __cuda_array_interface__ = {'shape': (n,), 'typestr': '<u8', 'data': (ptr, False), 'version': 2, 'strides': None}
    return torch.as_tensor(_A(), device='cuda:0')


torch.cuda.synchronize()
n_half, stride = 16, 2   # 32 walkers, one part (40 points) + log-prior
for r, p in enumerate(ptrs):
    m = view(p, 4 * n_half * stride * 2).cpu().numpy().reshape(4, n_half, stride, 2)
    tags = (m >> np.uint64(32)).astype(np.int64)
    print('rank', r, 'mailbox tags [generation ring][slot] of the log-prior column (granule 0):')
    print(tags[:, :, stride - 1, 0])
    print('   of the partial-sum column:')
    print(tags[:, :, 0, 0])
