#!/usr/bin/env python3
"""Pack the public bandpass tables and the SiFTO template into one .npz shipped with the package.

Build-host tool.  Reads the two-column (wavelength, transmission) ASCII/CSV tables under
``<reference>/lightcurve_fitting/filters`` and ``models/sifto.dat`` with its own tolerant parser and stores
the numbers *as they appear in the files* (no unit conversion, no sorting, no normalisation): all processing
happens in ``lightcurve_fitting_amd.filters`` at run time, so it can be checked against golden vectors.

Usage: python tools/pack_bandpass_data.py [/root/reference]
"""
import os
import sys

import numpy as np


def read_numeric_rows(path):
    rows = []
    with open(path) as fh:
        for line in fh:
            toks = line.replace(',', ' ').split()
            if not toks or toks[0].startswith('#'):
                continue
            try:
                rows.append([float(tk) for tk in toks])
            except ValueError:
                continue  # header row
    return np.array(rows, dtype=np.float64)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    pkg = os.path.join(ref, 'lightcurve_fitting')
    out = {}
    fdir = os.path.join(pkg, 'filters')
    for fn in sorted(os.listdir(fdir)):
        if not fn.endswith(('.dat', '.txt', '.csv', '.asci')):
            continue
        arr = read_numeric_rows(os.path.join(fdir, fn))
        assert arr.ndim == 2 and arr.shape[1] == 2, (fn, arr.shape)
        out['bandpass/' + fn] = arr
    sifto = read_numeric_rows(os.path.join(pkg, 'models', 'sifto.dat'))
    assert sifto.shape[1] == 7
    out['template/sifto'] = sifto  # columns: Epoch U B V g r i
    dest = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lightcurve_fitting_amd', 'data',
                        'bandpasses.npz')
    np.savez_compressed(dest, **out)
    print(f'wrote {dest}: {len(out)} tables, {os.path.getsize(dest) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
