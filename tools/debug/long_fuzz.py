"""The seeded parity sweeps of tests/test_gpu_fuzz.py over many more seeds (ad hoc soak; not part of the suite)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import test_gpu_fuzz as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = []
for seed in range(lo, hi):
    for fn in (T.test_shock_cooling_family_fuzz, T.test_companion_family_fuzz):
        try:
            fn(seed)
        except Exception as exc:  # noqa: BLE001
            bad.append((fn.__name__, seed, repr(exc)[:200]))
    if seed % 20 == 0:
        print('seed', seed, 'failures so far', len(bad), f'{time.time() - t0:.0f}s', flush=True)
print('done', hi - lo, 'seeds; failures:', bad)
