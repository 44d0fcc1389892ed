"""Cubic-spline interpolation on the host (knot tables are tiny; the device only evaluates the pieces)."""
import numpy as np


def not_a_knot_coefficients(x, y):
    """Piecewise-cubic coefficients ``c[i] = (c3, c2, c1, c0)`` of the C2 interpolant with not-a-knot ends:
    ``S(u) = c3 (u-x_i)^3 + c2 (u-x_i)^2 + c1 (u-x_i) + c0`` on ``[x_i, x_{i+1}]``.

    This is the interpolant SciPy's ``CubicSpline`` builds by default, which the reference uses for the SiFTO
    template (models.py:717).  Solved here as a dense (n x n) system for the knot derivatives."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(x)
    if n < 4:
        raise ValueError('need at least 4 knots')
    dx = np.diff(x)
    slope = np.diff(y) / dx
    A = np.zeros((n, n))
    b = np.zeros(n)
    for i in range(1, n - 1):  # continuity of the second derivative at interior knots
        A[i, i - 1] = dx[i]
        A[i, i] = 2. * (dx[i - 1] + dx[i])
        A[i, i + 1] = dx[i - 1]
        b[i] = 3. * (dx[i] * slope[i - 1] + dx[i - 1] * slope[i])
    d = x[2] - x[0]  # third derivative continuous across x_1
    A[0, 0], A[0, 1] = dx[1], d
    b[0] = ((dx[0] + 2. * d) * dx[1] * slope[0] + dx[0] ** 2 * slope[1]) / d
    d = x[-1] - x[-3]  # ... and across x_{n-2}
    A[-1, -1], A[-1, -2] = dx[-2], d
    b[-1] = (dx[-1] ** 2 * slope[-2] + (2. * d + dx[-1]) * dx[-2] * slope[-1]) / d
    s = np.linalg.solve(A, b)
    tq = (s[:-1] + s[1:] - 2. * slope) / dx
    return np.column_stack([tq / dx, (slope - s[:-1]) / dx - tq, s[:-1], y[:-1]])


def natural_coefficients(x, y):
    """Same layout as :func:`not_a_knot_coefficients` for the *natural* cubic spline (second derivative zero at
    both ends)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(x)
    if n < 3:
        raise ValueError('need at least 3 knots')
    dx = np.diff(x)
    slope = np.diff(y) / dx
    A = np.zeros((n, n))
    b = np.zeros(n)
    A[0, 0] = A[-1, -1] = 1.  # M_0 = M_{n-1} = 0
    for i in range(1, n - 1):  # continuity of the first derivative, in terms of the second derivatives M_i
        A[i, i - 1] = dx[i - 1]
        A[i, i] = 2. * (dx[i - 1] + dx[i])
        A[i, i + 1] = dx[i]
        b[i] = 6. * (slope[i] - slope[i - 1])
    m = np.linalg.solve(A, b)
    c3 = (m[1:] - m[:-1]) / (6. * dx)
    c2 = m[:-1] / 2.
    c1 = slope - dx * (2. * m[:-1] + m[1:]) / 6.
    return np.column_stack([c3, c2, c1, y[:-1]])
