"""Twenty 20-step runs back to back, for a kernel trace of what a short run puts on the device
(rocprofv3 --kernel-trace --output-format csv -d gpurun_out/short -- python3 tools/debug/short_run_trace.py; then
python3 tools/debug/short_run_trace.py --read gpurun_out/short)."""
import os, sys, glob, csv
if len(sys.argv) > 2 and sys.argv[1] == '--read':
    rows = []
    for p in glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True):
        rows += list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # the last complete run: from the last k_board_init (or first half-step kernel) back 1 run
    names = [r['Kernel_Name'] for r in rows]
    starts = [i for i, n in enumerate(names) if 'k_board_init' in n or 'k_solo_run' in n and (i == 0 or 'k_board_init' not in names[i - 1])]
    inits = [i for i, n in enumerate(names) if 'k_board_init' in n]
    a, b = (inits[-3], inits[-2]) if len(inits) >= 3 else (0, len(rows))
    t0 = int(rows[a]['Start_Timestamp'])
    prev_end = t0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        m = r['Kernel_Name']
        m = m[m.find('k_'):][:40] if 'k_' in m else m[:40]
        print(f'{(s - t0) / 1e3:9.2f} us  +{(s - prev_end) / 1e3:6.2f} gap  {(e - s) / 1e3:8.2f} us  {m}')
        prev_end = e
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from lightcurve_fitting_amd.sampler import EnsembleSampler
model, lc, priors = bench.build_problem(0)
eng = model.engine_for(lc, priors=priors)
x0 = bench.initial_walkers(1024)
s = EnsembleSampler(1024, 5, eng, seed=1)
s.reserve_chain(20)
s.run_mcmc(x0, 5, store=True)
for i in range(20):
    torch.cuda.synchronize()
    s.run_mcmc(None, 20, store=True)
    torch.cuda.synchronize()
