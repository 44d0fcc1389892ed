"""Summarise a rocprofv3 kernel trace CSV of tools/debug/trace_run.py: k_solo durations and the gaps between
consecutive launches, around the generation kernels of the draw-record blocks."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
solo = [(a, b) for a, b, n in ev if 'k_solo' in n]
gen = [(a, b, n.split('(')[0][-30:]) for a, b, n in ev if 'k_make_perm' in n or 'k_draws' in n or 'k_slots' in n]
dur = [b - a for a, b in solo]
gap = [solo[i + 1][0] - solo[i][1] for i in range(len(solo) - 1)]
import statistics as st
print('k_solo launches', len(solo), 'median duration us', st.median(dur) / 1e3, 'median gap us', st.median(gap) / 1e3)
t0 = solo[0][0]
for a, b, n in gen:
    print(f'  gen {n:32s} start {1e-3 * (a - t0):10.1f} us  duration {1e-3 * (b - a):8.1f} us')
# launches whose duration or following gap is far from the median
md, mg = st.median(dur), st.median(gap)
for i, (a, b) in enumerate(solo[:-1]):
    if dur[i] > 1.3 * md or gap[i] > mg + 3000:
        print(f'  launch {i:5d} at {1e-3 * (a - t0):10.1f} us: duration {1e-3 * dur[i]:7.1f} us, gap after {1e-3 * gap[i]:7.1f} us')
print('total span us', 1e-3 * (solo[-1][1] - solo[0][0]), ' sum of durations', 1e-3 * sum(dur), ' sum of gaps', 1e-3 * sum(gap))
seg = max(1, len(solo) // 10)
print('median duration us per tenth of the run:', [round(st.median(dur[i:i + seg]) / 1e3, 2) for i in range(0, len(dur), seg)])
print('sum of gaps us per tenth of the run:   ', [round(sum(gap[i:i + seg]) / 1e3, 1) for i in range(0, len(gap), seg)])
