#!/usr/bin/env python3
"""Build and prove, once, the table levels of every bandpass at z = 0 -- the Gauss-compressed "cool" and "hot" tables and
the interpolant of ln S(ln T) -- and ship them with the package (``lightcurve_fitting_amd/data/table_levels.npz``).

``PackedTables`` takes them from there for any redshift (``filters.shipped_levels``: the band sum at redshift z is the
z = 0 one at T / (1 + z), times (1 + z)^3), so that creating an engine costs milliseconds instead of the 30 ms per filter
the proofs take.  Tables with a cut-off frequency or reddening are still built when they are needed.

Build-host tool; needs nothing but this package:   python tools/pack_table_levels.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightcurve_fitting_amd import filters as F  # noqa: E402


def main():
    out, n_levels = {}, [0, 0, 0]
    seen = set()
    for f in F.all_filters:
        if not f.filename or f.filename in seen:
            continue
        seen.add(f.filename)
        a, w = f.planck_table(0., np.inf, True)
        if not len(a):
            continue
        lv = F.table_levels(a, w)
        key = 'levels/' + f.filename
        out[key + '/n'] = np.int64(len(a))
        out[key + '/asum'] = np.float64(a.sum())
        out[key + '/wsum'] = np.float64(w.sum())
        for k, (name, tag) in enumerate((('cool', 'c'), ('hot', 'h'))):
            if lv[name] is not None:
                n_levels[k] += 1
                out[f'{key}/{tag}a'], out[f'{key}/{tag}w'] = lv[name][0], lv[name][1]
                out[f'{key}/{tag}tmin'], out[f'{key}/{tag}bound'] = np.float64(lv[name][2]), np.float64(lv[name][3])
        if lv['interp'] is not None:
            n_levels[2] += 1
            out[key + '/icoef'] = lv['interp'][0]
            out[key + '/itmin'], out[key + '/ibound'] = np.float64(lv['interp'][1]), np.float64(lv['interp'][2])
    dest = os.path.join(ROOT, 'lightcurve_fitting_amd', 'data', 'table_levels.npz')
    np.savez_compressed(dest, **out)
    print(f'wrote {dest}: {len(seen)} tables; cool / hot / interpolant levels: {n_levels}; '
          f'{os.path.getsize(dest) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
