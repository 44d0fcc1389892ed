"""Stand-in for the third-party ``extinction`` package.

Only the A_V = 0 case is supported (A_lambda is proportional to A_V under Fitzpatrick 1999,
so the result is identically zero).  Any other request raises: that arithmetic is third-party
and its parity is unpinned.
"""
import numpy as np


def fitzpatrick99(wave, a_v, r_v=3.1, unit='aa'):
    if np.any(np.asarray(a_v) != 0.):
        raise NotImplementedError('extinction stand-in supports a_v == 0 only')
    return np.zeros_like(np.asarray(wave, dtype=float))
