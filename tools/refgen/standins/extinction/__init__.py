"""Stand-in for the third-party ``extinction`` package (absent from this container, no network).

``fitzpatrick99`` restates the Fitzpatrick (1999) law as that package publishes it: Fitzpatrick & Massa (1990)
parametrisation below 2700 A, natural cubic spline through nine anchors in 1/lambda above.  It reproduces the
package's README example, ``fitzpatrick99([2000, 4000, 8000], 1.0, 3.1) = [2.76225609, 1.42325373, 0.55333671]``.
With ``a_v == 0`` the result is identically zero whatever the law, so fixtures generated with E(B-V) = 0 contain the
reference's own arithmetic only; fixtures with E(B-V) != 0 pin the reference's plumbing around this restatement.
"""
import numpy as np
from scipy.interpolate import CubicSpline

_XK = 1e4 / np.array([np.inf, 26500., 12200., 6000., 5470., 4670., 4110., 2700., 2600.])


def _uv(x, c1, c2):
    d = x * x / ((x * x - 4.596 ** 2) ** 2 + x * x * 0.99 ** 2)
    y = np.clip(x - 5.9, 0., None)
    return c1 + c2 * x + 3.23 * d + 0.41 * (0.5392 * y * y + 0.05644 * y * y * y)


def fitzpatrick99(wave, a_v, r_v=3.1, unit='aa'):
    if unit != 'aa':
        raise NotImplementedError('stand-in: angstrom only')
    x = 1e4 / np.asarray(wave, dtype=float)
    if np.all(np.asarray(a_v) == 0.):
        return np.zeros_like(x)
    c2 = -0.824 + 4.717 / r_v
    c1 = 2.030 - 3.007 * c2
    r2 = r_v * r_v
    yk = np.array([-r_v, 0.26469 * r_v / 3.1 - r_v, 0.82925 * r_v / 3.1 - r_v,
                   -0.422809 + 1.00270 * r_v + 2.13572e-04 * r2 - r_v,
                   -5.13540e-02 + 1.00216 * r_v - 7.35778e-05 * r2 - r_v,
                   0.700127 + 1.00184 * r_v - 3.32598e-05 * r2 - r_v,
                   1.19456 + 1.01707 * r_v - 5.46959e-03 * r2 + 7.97809e-04 * r2 * r_v - 4.45636e-05 * r2 * r2 - r_v,
                   _uv(_XK[7], c1, c2), _uv(_XK[8], c1, c2)])
    spline = CubicSpline(_XK, yk, bc_type='natural')
    k = np.where(x >= _XK[7], _uv(x, c1, c2), spline(np.minimum(x, _XK[8])))
    return a_v * (1. + k / r_v)
