"""Per-launch view of the LAST 40 k_solo launches of a rocprofv3 kernel trace of `bench.py --steps 20 --warmup 5
--no-cpu-baseline` (= the timed steps): durations, gaps, what ran in between."""
import csv, re, sys


def short(name):
    m = re.search(r'(k_\w+|__amd_\w+)', name)
    return m.group(1) if m else name[:40]

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows))
idx = [i for i, e in enumerate(ev) if 'k_solo' in e[2]]
last = idx[-40:]
t0 = ev[last[0]][0]
prev_end = None
for i in range(last[0] - 3, len(ev)):
    a, b, n = ev[i]
    gap = '' if prev_end is None else f'gap {1e-3 * (a - prev_end):6.2f}'
    print(f'{1e-3 * (a - t0):9.2f} us  {1e-3 * (b - a):7.2f} us  {gap:12s} {n}')
    prev_end = b
print('span of the 40 launches', 1e-3 * (ev[last[-1]][1] - t0))
