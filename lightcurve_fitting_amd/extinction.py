"""Fitzpatrick (1999) dust extinction law, restated from the published algorithm.

The reference calls the third-party ``extinction`` package (``filters.py:14-33, 267-286``; no version pinned in its
``requirements.txt``), which is not available here, so this module cannot be compared with it by execution.  It is
pinned by the one known answer that package publishes (its README example, reproduced to all nine printed digits
in tests/test_host.py):  ``fitzpatrick99([2000, 4000, 8000] A, a_v=1.0, r_v=3.1) = [2.76225609, 1.42325373,
0.55333671]`` -- which also settles the spline's end conditions (natural; not-a-knot gives 1.42339, 0.55308).
What is restated is the law as published (Fitzpatrick 1999, PASP 111, 63, with the ultraviolet parametrisation of
Fitzpatrick & Massa 1990) in the form the IDL ``FM_UNRED`` routine and that package code it:

* ``x = 1e4 / wavelength[A]`` in inverse microns; ``k(x) = E(lambda - V) / E(B - V)``;
* ultraviolet, ``x >= 1e4 / 2700``: ``k = c1 + c2 x + c3 D(x) + c4 F(x)`` with ``c2 = -0.824 + 4.717 / R_V``,
  ``c1 = 2.030 - 3.007 c2``, ``c3 = 3.23``, ``c4 = 0.41``, Drude profile ``D = x^2 / ((x^2 - x0^2)^2 + x^2 gamma^2)``,
  ``x0 = 4.596``, ``gamma = 0.99``, and ``F = 0.5392 (x - 5.9)^2 + 0.05644 (x - 5.9)^3`` for ``x >= 5.9``;
* optical / infrared: cubic spline through nine anchor points at ``1e4 / (inf, 26500, 12200, 6000, 5470, 4670,
  4110, 2700, 2600) A`` (the last two taken from the ultraviolet formula), the optical anchors being polynomials
  in ``R_V``.  End conditions: natural (second derivative zero at both ends);
* ``A(lambda) = A_V (1 + k / R_V)``.

Further checks from the definition (tests/test_host.py): ``A(5470 A) = A_V (1 + k_5 / R_V)`` with the anchor value
``k_5``, ``A -> 0`` as ``x -> 0``, continuity at 2700 A, ``A_B - A_V ~ E(B-V)``.
"""
import numpy as np

from .spline import natural_coefficients

_X0 = 4.596
_GAMMA = 0.99
_C3 = 3.23
_C4 = 0.41
_C5 = 5.9
_KNOT_WAVE = np.array([np.inf, 26500., 12200., 6000., 5470., 4670., 4110., 2700., 2600.])
_XKNOTS = 1e4 / _KNOT_WAVE
_X_UV = 1e4 / 2700.


def _k_uv(x, r_v):
    """k(x) of the ultraviolet parametrisation."""
    c2 = -0.824 + 4.717 / r_v
    c1 = 2.030 - 3.007 * c2
    x2 = x * x
    d = x2 / ((x2 - _X0 * _X0) ** 2 + x2 * _GAMMA * _GAMMA)
    k = c1 + c2 * x + _C3 * d
    y = np.where(x >= _C5, x - _C5, 0.)
    return k + _C4 * (0.5392 * y * y + 0.05644 * y * y * y)


def _knot_values(r_v):
    rv2 = r_v * r_v
    k = np.empty(9)
    k[0] = -r_v
    k[1] = 0.26469 * r_v / 3.1 - r_v
    k[2] = 0.82925 * r_v / 3.1 - r_v
    k[3] = -0.422809 + 1.00270 * r_v + 2.13572e-04 * rv2 - r_v
    k[4] = -5.13540e-02 + 1.00216 * r_v - 7.35778e-05 * rv2 - r_v
    k[5] = 0.700127 + 1.00184 * r_v - 3.32598e-05 * rv2 - r_v
    k[6] = 1.19456 + 1.01707 * r_v - 5.46959e-03 * rv2 + 7.97809e-04 * rv2 * r_v - 4.45636e-05 * rv2 * rv2 - r_v
    k[7:] = _k_uv(_XKNOTS[7:], r_v)
    return k


_spline_cache = {}


def k_lambda(wave, r_v=3.1):
    """``E(lambda - V) / E(B - V)`` at wavelengths ``wave`` [angstrom]."""
    x = 1e4 / np.asarray(wave, dtype=np.float64)
    coef = _spline_cache.get(r_v)
    if coef is None:
        coef = _spline_cache[r_v] = natural_coefficients(_XKNOTS, _knot_values(r_v))
    seg = np.clip(np.searchsorted(_XKNOTS, x, side='right') - 1, 0, len(_XKNOTS) - 2)
    dx = x - _XKNOTS[seg]
    c = coef[seg]
    k_spline = ((c[..., 0] * dx + c[..., 1]) * dx + c[..., 2]) * dx + c[..., 3]
    return np.where(x >= _X_UV, _k_uv(x, r_v), k_spline)


def fitzpatrick99(wave, a_v, r_v=3.1):
    """Extinction ``A(lambda)`` [mag] at ``wave`` [angstrom] for total V-band extinction ``a_v`` (same argument
    order as ``extinction.fitzpatrick99``)."""
    return a_v * (1. + k_lambda(wave, r_v) / r_v)


def a_lambda_per_ebv(wave, r_v=3.1):
    """``A(lambda) / E(B-V) = R_V + k(lambda)``: the per-sample exponent the engine multiplies by E(B-V)."""
    return r_v + k_lambda(wave, r_v)


def extinction_law(freq, ebv, rv=3.1):
    """Extinction factor ``10 ** (A / -2.5)`` at frequencies ``freq`` [THz] in the frame of the dust, one row per
    element of ``ebv`` (squeezed) -- the reference's ``filters.extinction_law`` (filters.py:14-33)."""
    from .filters import c
    e = a_lambda_per_ebv(c / np.asarray(freq, dtype=np.float64), rv)
    A = np.squeeze([e * x for x in np.atleast_1d(ebv)])
    return 10. ** (A / -2.5)
