"""Where an emulated multi-rank row-board run leaves the single-GPU chain: repeats the scenario of
tests/test_gpu_half_step_kernels.py::test_row_boards_emulated_ranks and prints (run, rank, step, walker) of every mismatch,
with the reference taken from k_solo_run ('auto') and from k_solo ('solo').   python tools/debug/rows_mismatch.py [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import lc_dict  # noqa: E402
from lightcurve_fitting_amd import models as M  # noqa: E402
from lightcurve_fitting_amd.engine import NativeSampler  # noqa: E402
import test_gpu_half_step_kernels as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
resident = os.environ.get('ROWS_RESIDENT') == '1'
pb, eng = T._multiband()
priors = [M.UniformPrior(0., 10.)] * 3 + [M.UniformPrior(0., 2.2)] + [M.UniformPrior(-1., 0.5)]
lc = lc_dict(pb['t'], [b.name for b in pb['bands']], pb['y'], pb['dy'])
bad = 0
for rep in range(reps):
    for ranks, nwalkers, split in [(2, 48, 'random'), (3, 54, 'random'), (2, 44, 'identity'), (3, 42, 'random')]:
        nsteps = 9
        x0 = pb['truth'] * (1 + 0.05 * np.random.default_rng(4 + rep).standard_normal((nwalkers, 5)))
        import ctypes
        from lightcurve_fitting_amd.engine import load_library
        lib = load_library()
        if hasattr(lib, 'lcf_debug_read_progress'):
            buf = (ctypes.c_uint * 1024)()
            lib.lcf_debug_read_progress(buf)
            before = np.array(buf).reshape(64, 16)
        refs = {k: T._run(eng, nwalkers, 321, x0, nsteps, k, split) for k in ('auto', 'solo')}
        if hasattr(lib, 'lcf_debug_read_progress') and rep == 3 and nwalkers == 54:
            lib.lcf_debug_read_progress(buf)
            print('stamps passed per workgroup during the two reference runs (auto gave up; rows = workgroups, columns = stamps 0..12):')
            np.set_printoptions(linewidth=200)
            print(refs['auto'][4].last_run_kernel(), (np.array(buf).reshape(64, 16) - before)[:nwalkers // 2, :13])
        same = np.array_equal(refs['auto'][1], refs['solo'][1]) and np.array_equal(refs['auto'][3], refs['solo'][3])
        want_chain, want_lp, want_acc = refs['solo'][1], refs['solo'][2], refs['solo'][3]
        engines = [M.ShockCooling(redshift=0.004).engine_for(lc, priors=priors) for _ in range(ranks)]
        samplers = [NativeSampler(e, nwalkers, 321) for e in engines]
        ptrs = [s.board_export()[1] for s in samplers]
        for r, s in enumerate(samplers):
            s.board_connect(ranks, r, local_ptrs=ptrs)
            s.set_state(x0)
            if hasattr(lib, 'lcf_debug_read_progress'):
                lib.lcf_debug_read_progress(buf)
                before = np.array(buf).reshape(64, 16)
            s.run(100, nsteps, split, True)
            if s.last_run_kernel() != 'run':
                lib.lcf_debug_read_run_board.restype = ctypes.c_longlong
                lib.lcf_debug_read_run_board.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
                raw = np.zeros(128 * nwalkers * 8 * 2 + 64, dtype=np.uint64)
                got = lib.lcf_debug_read_run_board(s._h, raw.ctypes.data, raw.nbytes)
                rows = raw[:128 * nwalkers * 16].reshape(128, nwalkers, 8, 2)
                tags = (rows >> np.uint64(32)).astype(np.int64)
                print('run board of rank', r, ':', got, 'bytes; tags of column 0 (granule 0), rows = versions 0..23, columns = walkers')
                np.set_printoptions(linewidth=250, threshold=1000000)
                print(tags[:24, :, 0, 0])
                print('walkers with a version-21 row:', np.flatnonzero(tags[21, :, 0, 0] == 21).tolist())
                print('walkers with a version-22 row:', np.flatnonzero(tags[22, :, 0, 0] == 22).tolist())
                print('walker 29, tags of all 8 columns in versions 16..23:'); print(tags[16:24, 29, :, 0], flush=True)
            if hasattr(lib, 'lcf_debug_read_progress'):
                lib.lcf_debug_read_progress(buf)
                d = (np.array(buf).reshape(64, 16) - before)[:nwalkers // 2, :13]
                if s.last_run_kernel() != 'run' or d.max() != 18:
                    np.set_printoptions(linewidth=200)
                    print('warm-up of rank', r, 'ran as', s.last_run_kernel(), '; stamps passed per workgroup:')
                    print(d, flush=True)
            s.set_state(x0)
        for first, n in ((0, 4), (4, nsteps - 4)):
            for s in samplers:
                if resident:
                    s.run_rows(first, n, split, True, asynchronous=True, resident=True)
                else:
                    s.run_rows(first, n, split, True, asynchronous=True)
            for s in samplers:
                s.wait()
        msg = []
        for r, s in enumerate(samplers):
            chain, lp = s.get_chain()
            d = np.argwhere(np.any(chain != want_chain[4:], axis=2) | (lp != want_lp[4:]))
            if len(d):
                msg.append(f'rank {r}: {len(d)} (step, walker) rows differ, first {d[:4].tolist()}')
            if not np.array_equal(s.naccepted(), want_acc):
                msg.append(f'rank {r}: acceptance counts differ at {np.flatnonzero(s.naccepted() != want_acc)[:8].tolist()}')
        bad += bool(msg) or not same
        print(rep, ranks, nwalkers, split, 'run==solo' if same else 'RUN != SOLO', 'ok' if not msg else msg, flush=True)
        for s in samplers:
            s.close()
print('scenarios with a mismatch:', bad)
