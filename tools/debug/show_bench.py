"""Print the headline fields of bench.py JSON lines: python tools/debug/show_bench.py file.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    t = open(f).read().strip()
    if not t:
        print(f, 'EMPTY')
        continue
    d = json.loads(t.splitlines()[-1])
    r = d.get('roofline') or {}
    print(f"{f:40s} value={d['value']:.4e} ms/step={d.get('ms_per_step', 0):.4f} "
          f"dev={d.get('device_ms_per_step') or 0:.4f} kernel_ms={r.get('kernel_ms') or 0:.4f}")
