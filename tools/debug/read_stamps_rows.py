"""Stamps (tools/debug/make_stamp_build.py) of the row-board form of k_solo, one rank over its own board."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from lightcurve_fitting_amd import engine as E
from lightcurve_fitting_amd.engine import NativeSampler
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
nw = 1024
for rows in (False, True):
    s = NativeSampler(eng, nw, 1)
    if rows:
        s.board_connect(1, 0, local_ptrs=[s.board_export()[1]])
    s.set_state(bench.initial_walkers(nw))
    (s.run_rows if rows else s.run)(0, 40, 'random', False)
    lib = E.load_library()
    buf = (C.c_ulonglong * (64 * 16))()
    lib.lcf_debug_read_stamps.argtypes = [C.c_void_p]
    lib.lcf_debug_read_stamps(buf)
    a = np.array(buf[:], dtype=np.int64).reshape(64, 16)
    names = ['entry->draw record', 'rows + proposal', 'logarithms', 'coefficients', 'priors + publish', 'barrier', 'thermal',
             'points', 'wave sums + barrier', 'commit']
    d = np.diff(a[:, :11], axis=1)
    print('ROW BOARD' if rows else 'PLAIN', ' '.join(f'{n}={v:.0f}' for n, v in zip(names, np.median(d, axis=0))),
          'total', np.median(a[:, 10] - a[:, 0]), 'wave1 check done at', np.median(a[:, 11] - a[:, 0]),
          'device us/half-step', 1e3 * s.last_run_ms() / 80)
    s.close()
