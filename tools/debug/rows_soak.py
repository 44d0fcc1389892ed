"""Soak of the row-board run: `ranks` emulated ranks in one process, many steps, random colouring; final state, counts and
(thinned) chain against the single-GPU run.   python tools/debug/rows_soak.py [ranks=2] [walkers=512] [steps=2000]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd import models as M
from lightcurve_fitting_amd.engine import NativeSampler
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
x0 = bench.initial_walkers(nw)
engines = []
for _ in range(ranks + 1):
    model, lc, priors = bench.build_problem(0)
    engines.append(model.engine_for(lc, priors=priors))
ref = NativeSampler(engines[-1], nw, 99)
ref.set_state(x0)
ref.run(0, steps, 'random', True)
want_chain, want_lp = ref.get_chain()
want_x, _ = ref.get_state()
samplers = [NativeSampler(e, nw, 99) for e in engines[:ranks]]
ptrs = [s.board_export()[1] for s in samplers]
for r, s in enumerate(samplers):
    s.board_connect(ranks, r, local_ptrs=ptrs)
    s.set_state(x0)
    s.run(10 ** 6, steps, 'random', True)      # size every buffer first (host-thread allocations would stall the others)
    s.set_state(x0)
import threading
t0 = time.perf_counter()
# one host thread per rank: a single thread that enqueues thousands of launches for rank 0 fills its launch queue and
# blocks before it ever reaches rank 1 (whose rows rank 0's first launch is waiting for)
errors = []
def work(s):
    try:
        s.run_rows(0, steps, 'random', True)
    except Exception as exc:  # noqa: BLE001
        errors.append(str(exc))
threads = [threading.Thread(target=work, args=(s,)) for s in samplers]
for t in threads:
    t.start()
for t in threads:
    t.join()
dt = time.perf_counter() - t0
if errors:
    print(errors)
    sys.exit(1)
ok = True
for r, s in enumerate(samplers):
    chain, lp = s.get_chain()
    x, _ = s.get_state()
    same = np.array_equal(chain, want_chain) and np.array_equal(lp, want_lp) and np.array_equal(x, want_x) and \
        np.array_equal(s.naccepted(), ref.naccepted())
    ok &= same
    print(f'rank {r}: equal to the single-GPU run: {same}')
print(f'{ranks} emulated ranks, {nw} walkers, {steps} steps in {dt:.2f} s ({1e6 * dt / (2 * steps):.1f} us per half-step); all equal: {ok}')
sys.exit(0 if ok else 1)
