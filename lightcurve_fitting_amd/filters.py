"""Bandpass registry and the filter-table packer for the MI355X likelihood engine.

Host-side, one-off work: load a (wavelength, transmission) table, normalise it the
way the reference does, and fold everything that does not depend on the walker into two per-sample constants
``(a_k, W_k)`` so that the device evaluates the band-averaged blackbody luminosity density as

    L_nu(filter; T, R) = R**2 * sum_k W_k / (exp(a_k / T) - 1)

Reference behaviour mirrored here (``/root/reference/lightcurve_fitting``):

* ``filters.py:117-168``  -- ``Filter`` identity: ``name``, ``names`` (aliases), ``char``, ``fnu``, ``m0``, ``M0``
* ``filters.py:170-230``  -- ``read_curve``: A->nm, stable sort by wavelength, T/max(T), nu = c/lambda in THz,
  photon-weighted normalisation ``T_norm_per_freq``; ``freq_eff``; ``dfreq`` (sign-flipped)
* ``filters.py:288-310``  -- ``synthesize``: trapezoid over the (descending) frequency grid of
  ``spectrum(nu*(1+z)) * T_norm_per_freq``
* ``filters.py:369-445``  -- the registry ``all_filters`` / ``filtdict`` (names and aliases are API surface)
* ``models.py:1101-1128`` -- ``c1``, ``c2`` and the Planck function with its optional cut-off frequency

Nothing in this module evaluates a likelihood; the arithmetic on walkers happens only in the HIP kernels.
"""
import os
from functools import total_ordering

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'bandpasses.npz')

# physical constants, CODATA 2018 / IAU 2015 (exact SI definitions where they exist)
_H = 6.62607015e-34  # J s
_KB = 1.380649e-23  # J / K
_C = 299792458.  # m / s
_RSUN = 6.957e8  # m

#: speed of light in Angstrom THz (``filters.py:11``)
c = _C * 1e10 / 1e12
#: h/k_B in kK / THz (``models.py:1101``)
c1 = 0.0479924307336622
#: 8 pi^2 h / c^2 in W Hz^-1 (1000 Rsun)^-2 THz^-3 (``models.py:1102``)
c2 = 281739904251.4432

_bank = None


def _tables():
    global _bank
    if _bank is None:
        _bank = np.load(_DATA)
    return _bank


def trapezoid_weights(x):
    """Weights ``w`` with ``sum(w * y) == trapz(y, x)`` for any ``y`` (x may be descending or have repeats)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.zeros_like(x)
    if len(x) > 1:
        dx = np.diff(x)
        w[:-1] += 0.5 * dx
        w[1:] += 0.5 * dx
    return w


def gauss_rule(x, w, m):
    """``m``-point Gauss quadrature of the discrete measure ``sum_k w_k delta(x - x_k)`` (all ``w_k > 0``): nodes
    ``xg`` and positive weights ``wg`` with ``sum_j wg_j p(xg_j) == sum_k w_k p(x_k)`` for every polynomial ``p`` of
    degree < 2m.  Lanczos tridiagonalisation of ``diag(x)`` started from ``sqrt(w)`` with full reorthogonalisation
    in extended precision, then the Golub-Welsch eigenproblem."""
    ld = np.longdouble
    x = np.asarray(x, dtype=ld)
    w = np.asarray(w, dtype=ld)
    mu0 = w.sum()
    basis = [np.sqrt(w / mu0)]
    alpha, beta = [], []
    for j in range(m):
        v = x * basis[j]
        alpha.append((basis[j] * v).sum())
        v = v - alpha[j] * basis[j] - (beta[j - 1] * basis[j - 1] if j else 0)
        for _ in range(2):
            for q in basis:
                v = v - (q * v).sum() * q
        beta.append(np.sqrt((v * v).sum()))
        basis.append(v / beta[j])
    off = np.array(beta[:-1], dtype=np.float64)
    jac = np.diag(np.array(alpha, dtype=np.float64)) + np.diag(off, 1) + np.diag(off, -1)
    nodes, vecs = np.linalg.eigh(jac)
    return nodes, float(mu0) * vecs[0] ** 2


#: relative accuracy the compressed band sum must reach against the full sum wherever it is used
COMPRESSION_TOL = 2e-14
#: lowest temperatures [kK] down to which the two compressed levels must hold (see PackedTables)
COOL_TMIN = 1.0
HOT_TMIN = 8.0
_T_GRID = np.geomspace(0.2, 2e4, 101)      # coarse grid: finds the order and a first t_min
_DENSE_N, _DENSE_TMAX = 2048, 1e5          # dense proof grid: 2048 temperatures from t_min to 1e5 kK


def band_sum_exact(a, w, temps):
    """``sum_k W_k / (e^{a_k / T} - 1)`` in extended precision for an array of temperatures: the yardstick the
    compressed tables are proved against (and nothing the engine computes with)."""
    ld = np.longdouble
    x = np.multiply.outer(1. / np.asarray(temps, dtype=ld), np.asarray(a, dtype=ld))
    with np.errstate(over='ignore'):
        return (np.asarray(w, dtype=ld) / np.expm1(x)).sum(axis=1)


def compression_error(a, w, ag, wg, t_min, n=_DENSE_N, t_max=_DENSE_TMAX):
    """Largest relative difference between the compressed and the full band sum on ``n`` log-spaced temperatures from
    ``t_min`` to ``t_max`` (extended precision on both sides), and the temperature at which it occurs.  Beyond
    ``t_max`` the summand tends to ``T / a_k``: the error there is the one at ``t_max`` (a_k / T < 2e-3)."""
    temps = np.geomspace(max(t_min, _T_GRID[0]), t_max, n)
    full, comp = band_sum_exact(a, w, temps), band_sum_exact(ag, wg, temps)
    err = np.abs(comp - full) / np.abs(full)
    k = int(np.argmax(err))
    return float(err[k]), float(temps[k]), temps, err


_compressed = {}   # proved levels by content: engines of one process share filters, redshift and cut-off


def compress_planck_table(a, w, tol=COMPRESSION_TOL, orders=(12, 16, 24, 32), max_tmin=2.0, min_ratio=2.):
    """:func:`prove_compressed_table`, remembered per table content and arguments."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    key = (a.tobytes(), w.tobytes(), float(tol), tuple(orders), float(max_tmin), float(min_ratio))
    if key not in _compressed:
        if len(_compressed) > 512:
            _compressed.clear()
        _compressed[key] = prove_compressed_table(a, w, tol, orders, max_tmin, min_ratio)
    return _compressed[key]


def prove_compressed_table(a, w, tol=COMPRESSION_TOL, orders=(12, 16, 24, 32), max_tmin=2.0, min_ratio=2.):
    """Shorter table ``(a', W')`` with ``sum W'/(e^{a'/T} - 1) == sum W/(e^{a/T} - 1)`` to ``tol`` for every
    temperature ``T >= t_min``: the Gauss rule of the table's own discrete measure (1/(e^{a/T} - 1) is analytic in a,
    so a rule exact to polynomial degree 2m-1 converges geometrically in m; it degrades only when the band spans many
    e-folds, i.e. at low T).

    A rule is first located on a coarse temperature grid and then PROVED on a dense one -- 2048 temperatures from its
    ``t_min`` to 1e5 kK, extended precision, for exactly the samples given (redshift and cut-off included): a level
    whose dense error exceeds ``tol`` has its ``t_min`` raised above the failing temperatures and is proved again, or is
    dropped (the engine then walks the next longer table: fail closed).  Returns ``(a', W', t_min, bound)`` with
    ``bound`` the largest relative error found, or ``None``."""
    a = np.asarray(a, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    with np.errstate(over='ignore'):
        full = np.array([np.sum(w / np.expm1(a / t)) for t in _T_GRID])
    step = _T_GRID[1] / _T_GRID[0]
    for m in orders:
        if min_ratio * m > len(a):
            break
        ag, wg = gauss_rule(a, w, m)
        if not (np.all(wg > 0.) and np.all(ag > 0.)):
            continue
        with np.errstate(over='ignore'):
            comp = np.array([np.sum(wg / np.expm1(ag / t)) for t in _T_GRID])
        ok = np.abs(comp - full) <= tol * np.abs(full)
        bad = np.nonzero(~ok)[0]
        first_good = 0 if len(bad) == 0 else bad[-1] + 1
        if not (first_good < len(_T_GRID) and _T_GRID[first_good] <= max_tmin):
            continue
        # one coarse grid step of margin above the last failing temperature (never below the grid's lower end)
        t_min = float(_T_GRID[min(first_good + 1, len(_T_GRID) - 1)] if len(bad) else _T_GRID[0])
        ag, wg = np.ascontiguousarray(ag[::-1]), np.ascontiguousarray(wg[::-1])
        for _ in range(3):  # the proof; a failure between coarse points raises t_min and is proved again
            bound, _, temps, err = compression_error(a, w, ag, wg, t_min)
            if bound <= tol:
                return ag, wg, t_min, bound
            t_min = float(temps[np.nonzero(err > tol)[0][-1]] * step)
            if t_min > max_tmin:
                break
    return None


# ---------------------------------------------------------------------------------------------------------------
# Third level: the band sum as a FUNCTION of temperature, tabulated once per (filter, redshift, cut-off)
# ---------------------------------------------------------------------------------------------------------------
#: the band sum of a filter depends on the walker through ONE number, the temperature: g(u) = ln S(e^u), u = ln T, is
#: analytic in the strip |Im u| < pi/2, so piecewise polynomials of degree 7 on 64 equal intervals of u reproduce it to
#: rounding between 2 and 256 kK.  A model whose temperature and radius are known in log space (power laws in time)
#: then costs one table lookup, 7 fused multiply-adds and one exponential per data point instead of a sum over samples.
INTERP_TMIN, INTERP_TMAX, INTERP_M, INTERP_DEGREE = 2.0, 256.0, 64, 7
#: relative accuracy of exp(g) against the full sum: the evaluation floor of a log-space result (|ln L_nu| ~ 46 times
#: a few roundings of 1.1e-16), a decade below it nothing is gained
INTERP_TOL = 3e-13
_interpolants = {}


def interp_planck_table(a, w, t_lo=INTERP_TMIN, t_hi=INTERP_TMAX, m=INTERP_M, degree=INTERP_DEGREE, tol=INTERP_TOL):
    """Piecewise-polynomial table of ``g(u) = ln sum_k W_k / (e^{a_k e^-u} - 1)`` on ``m`` equal intervals of
    ``u = ln T`` between ``t_lo`` and ``t_hi``: per interval the polynomial of degree ``degree`` through its Chebyshev
    points, stored as monomial coefficients in ``s in [-1, 1]`` (highest degree first, for Horner's rule).

    Proved like the compressed tables: on 2048 temperatures across the range (the interpolant evaluated in float64
    exactly as the device does, against the full float64 sum) the error of ``exp(g)`` must stay below ``tol`` from
    some temperature ``t_min`` on -- hot bands pass everywhere, far-ultraviolet ones only above a few kK.  Returns
    ``(coef[m, degree + 1], t_min, bound)`` or ``None`` if no part of the range can be proved (the engine then walks
    the sample tables for that filter)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    key = (a.tobytes(), w.tobytes(), t_lo, t_hi, m, degree, tol)
    if key in _interpolants:
        return _interpolants[key]
    def full(temps):  # float64 is enough here: its error (~1e-15 K^0.5 + x 1e-16) is two decades below tol
        with np.errstate(over='ignore'):
            return (w / np.expm1(np.multiply.outer(1. / temps, a))).sum(axis=1)
    u0, u1 = np.log(t_lo), np.log(t_hi)
    h = (u1 - u0) / m
    k = np.arange(degree + 1)
    xc = np.cos(np.pi * (2 * k + 1) / (2 * (degree + 1)))
    centres = u0 + (np.arange(m) + 0.5) * h
    nodes = centres[:, None] + 0.5 * h * xc[None, :]
    with np.errstate(divide='ignore'):
        gv = np.log(full(np.exp(nodes.ravel()))).reshape(m, degree + 1)
    # interpolation through the same nodes in every interval: one (degree + 1)^2 system, highest power first
    with np.errstate(invalid='ignore'):
        coef = np.linalg.solve(np.vander(xc, degree + 1), gv.T).T
    coef[~np.all(np.isfinite(gv), axis=1)] = np.nan
    # the proof: the device's own evaluation (interval from u, Horner in float64, one exponential) against the full sum
    temps = np.geomspace(t_lo, t_hi, 2048)
    u = np.log(temps)
    j = np.clip(((u - u0) * (1. / h)).astype(np.int64), 0, m - 1)
    sloc = (u - (u0 + (j + 0.5) * h)) * (2. / h)
    g = coef[j, 0]
    for d in range(1, degree + 1):
        g = g * sloc + coef[j, d]
    with np.errstate(invalid='ignore', divide='ignore'):
        err = np.abs(np.expm1(g - np.log(full(temps))))
    bad = np.nonzero(~(err <= tol))[0]
    out = None
    if len(bad) == 0:
        out = (coef, float(t_lo), float(err.max()))
    elif bad[-1] + 1 < len(temps) and temps[bad[-1]] < 0.25 * t_hi:
        first = bad[-1] + 1   # valid from the next interval boundary above the last failing temperature
        jb = min(m - 1, int(j[first]) + 1)
        t_min = float(np.exp(u0 + jb * h))
        good = temps >= t_min
        if good.any() and np.all(err[good] <= tol):
            out = (coef, t_min, float(err[good].max()))
    if len(_interpolants) > 256:
        _interpolants.clear()
    _interpolants[key] = out
    return out


@total_ordering
class Filter:
    """A broadband filter: identity, zero point and (lazily loaded) normalised transmission curve.

    Attributes mirror the reference's ``Filter`` (``filters.py:36-168``) as far as the likelihood path and its
    callers need them: ``name``, ``names``, ``char``, ``system``, ``offset``, ``fnu``, ``m0``, ``M0``,
    ``filename``, ``angstrom``; curve properties ``wl`` (nm, ascending), ``T`` (peak 1), ``freq`` (THz, descending),
    ``T_norm_per_freq``, ``freq_eff``, ``dfreq``, ``wl_eff``, ``dwl``.
    """

    order = None

    def __init__(self, names, offset=0, system=None, fnu=3.631e-23, filename='', angstrom=False, italics=True):
        if isinstance(names, list):
            self.name = names[0]
            self.names = names
        else:
            self.name = names
            self.names = [names]
        if len(self.name) == 1:
            self.char = self.name
        else:
            shortest = min(self.names, key=len)  # first of the shortest aliases, like a stable sort by length
            self.char = shortest if len(shortest) == 1 else 'x'
        self.offset = offset
        self.system = system
        self.italics = italics
        self.fnu = fnu
        if fnu is None:
            self.m0 = self.M0 = float('nan')
        else:
            self.m0 = 2.5 * np.log10(fnu)
            self.M0 = self.m0 + 90.19
        self.filename = filename
        self.angstrom = angstrom
        self._curve = None

    # --- curve ------------------------------------------------------------
    def read_curve(self, force=False):
        """Load and normalise the transmission curve (``filters.py:170-230``).  No-op for table-less filters."""
        if (self._curve is not None and not force) or not self.filename:
            return
        raw = _tables()['bandpass/' + self.filename]
        wl = raw[:, 0] / 10. if self.angstrom else raw[:, 0].copy()  # -> nm
        order = np.argsort(wl, kind='stable')
        wl = wl[order]
        trans = raw[order, 1] / np.max(raw[:, 1])
        freq = (_C * 1e-3) / wl  # c / lambda with lambda in nm -> THz
        tw = trapezoid_weights(freq)
        t_per_freq = trans / freq
        t_norm = t_per_freq / np.sum(tw * t_per_freq)
        dfreq = np.sum(tw * trans)
        freq_eff = np.sum(tw * trans * freq) / dfreq
        tw_wl = trapezoid_weights(wl)
        dwl = np.sum(tw_wl * trans)
        wl_eff = np.sum(tw_wl * trans * wl) / dwl
        self._curve = dict(wl=wl, T=trans, freq=freq, T_norm_per_freq=t_norm, tw=tw, freq_eff=freq_eff,
                           dfreq=-dfreq, wl_eff=wl_eff, dwl=dwl)

    def _get(self, key):
        self.read_curve()
        return None if self._curve is None else self._curve[key]

    wl = property(lambda self: self._get('wl'))
    T = property(lambda self: self._get('T'))
    freq = property(lambda self: self._get('freq'))
    T_norm_per_freq = property(lambda self: self._get('T_norm_per_freq'))
    freq_eff = property(lambda self: self._get('freq_eff'))
    dfreq = property(lambda self: self._get('dfreq'))
    wl_eff = property(lambda self: self._get('wl_eff'))
    dwl = property(lambda self: self._get('dwl'))

    @property
    def nsamples(self):
        fr = self.freq
        return 0 if fr is None else len(fr)

    def planck_table(self, z=0., cutoff_freq=np.inf, drop_zeros=True):
        """Per-sample constants ``(a_k, W_k)`` of the band integral for source redshift ``z``.

        ``a_k = c1 nu_k (1+z)`` [kK] and ``W_k = c2 nu'_k^3 min(1, nu_cut/nu'_k) tw_k Tnorm_k`` with the trapezoid
        weights ``tw_k`` of the descending frequency grid (``filters.py:308-310`` + ``models.py:1127-1128``).
        Samples with ``W_k == 0`` (zero-transmission rows, zero-width steps) contribute exactly 0 to the sum and
        are dropped unless ``drop_zeros`` is false.
        """
        if not self.filename:
            raise ValueError(f'filter {self.name!r} has no transmission table')
        nu = self.freq * (1. + z)
        a = c1 * nu
        w = c2 * nu ** 3 * np.minimum(1., cutoff_freq / nu) * self._get('tw') * self.T_norm_per_freq
        if drop_zeros:
            keep = w != 0.
            a, w = a[keep], w[keep]
        return np.ascontiguousarray(a), np.ascontiguousarray(w)

    def extinction_table(self, z=0., cutoff_freq=np.inf, drop_zeros=True, rv=3.1):
        """``A(lambda_k) / E(B-V)`` of the Fitzpatrick (1999) law at the emitted-frame wavelength of every sample
        :meth:`planck_table` returns (same order, same dropped rows): ``Filter.synthesize`` multiplies the spectrum
        by ``10 ** (-0.4 E(B-V) e_k)`` sample by sample (``filters.py:32-33, 308-310``)."""
        from .extinction import a_lambda_per_ebv
        nu = self.freq * (1. + z)
        e = a_lambda_per_ebv(c / nu, rv)
        if drop_zeros:
            w = nu ** 3 * np.minimum(1., cutoff_freq / nu) * self._get('tw') * self.T_norm_per_freq
            e = e[w != 0.]
        return np.ascontiguousarray(e)

    def extinction(self, ebv, rv=3.1, z=0.):
        """Extinction ``A_lambda`` [mag] at this filter's effective wavelength (``filters.py:267-286``); ``None``
        for a filter without a transmission table."""
        from .extinction import fitzpatrick99
        if self.wl_eff is None:
            return None
        return fitzpatrick99(np.array([self.wl_eff * 10. / (1. + z)]), ebv * rv, rv)[0]

    # --- identity ---------------------------------------------------------
    def __str__(self):
        return self.name

    def __repr__(self):
        return '<filter ' + self.name + '>'

    def __eq__(self, other):
        return isinstance(other, Filter) and self.name == other.name

    def __lt__(self, other):
        return isinstance(other, Filter) and Filter.order.index(self.name) < Filter.order.index(other.name)

    def __hash__(self):
        return hash(self.name)


def _mk(names, offset=0, system=None, fnu=3.631e-23, filename='', nm=False, italics=True):
    return Filter(names, offset, system, fnu, filename, angstrom=bool(filename) and not nm, italics=italics)


#: every recognised filter, bluest first (same names, aliases, zero points and tables as ``filters.py:369-440``)
all_filters = [
    _mk('FUV', 8, 'GALEX', filename='GALEX_GALEX.FUV.dat'),
    _mk('NUV', 8, 'GALEX', filename='GALEX_GALEX.NUV.dat'),
    _mk(['UVW2', 'uvw2', 'W2', '2', 'uw2'], 8, 'Swift', 7.379e-24, 'Swift_UVOT.UVW2.dat'),
    _mk(['UVM2', 'uvm2', 'M2', 'M', 'um2'], 8, 'Swift', 7.656e-24, 'Swift_UVOT.UVM2.dat'),
    _mk(['UVW1', 'uvw1', 'W1', '1', 'uw1'], 4, 'Swift', 9.036e-24, 'Swift_UVOT.UVW1.dat'),
    _mk(['u', "u'", 'up', 'uprime'], 3, 'Gunn', filename='SLOAN_SDSS.u.dat'),
    _mk(['U_S', 's', 'us'], 3, 'Swift', 1.419e-23, 'Swift_UVOT.U.dat'),
    _mk('U', 3, 'Johnson', 1.790e-23, 'Generic_Johnson.U.dat'),
    _mk('B', 2, 'Johnson', 4.063e-23, 'Generic_Johnson.B.dat'),
    _mk(['B_S', 'b', 'bs'], 2, 'Swift', 4.093e-23, 'Swift_UVOT.B.dat'),
    _mk(['g', "g'", 'gp', 'gprime', 'F475W'], 1, 'Gunn', filename='SLOAN_SDSS.g.dat'),
    _mk('g-DECam', 1, 'DECam', filename='CTIO_DECam.g.dat'),
    _mk(['c', 'cyan'], 1, 'ATLAS', filename='ATLAS_cyan.txt', nm=True),
    _mk('V', 1, 'Johnson', 3.636e-23, 'Generic_Johnson.V.dat'),
    _mk(['V_S', 'v', 'vs'], 1, 'Swift', 3.664e-23, 'Swift_UVOT.V.dat'),
    _mk('Itagaki', 0, 'Itagaki', filename='KAF-1001E.asci', nm=True, italics=False),
    _mk('white', 0, 'MOSFiT', filename='white.txt', nm=True, italics=False),
    _mk(['unfilt.', '0', 'C', 'clear', 'pseudobolometric', 'griz', 'RGB', 'LRGB'], 0, 'MOSFiT',
        filename='pseudobolometric.txt', nm=True, italics=False),
    _mk('G', 0, 'Gaia', filename='GAIA_GAIA0.G.dat'),
    _mk('Kepler', 0, 'Kepler', filename='Kepler_Kepler.K.dat', italics=False),
    _mk('TESS', 0, 'TESS', filename='TESS_TESS.Red.dat', italics=False),
    _mk(['DLT40', 'Open', 'Clear'], 0, 'DLT40', filename='QE_E2V_MBBBUV_Broadband.csv', nm=True, italics=False),
    _mk('w', 0, 'Gunn', filename='PAN-STARRS_PS1.w.dat'),
    _mk(['o', 'orange'], 0, 'ATLAS', filename='ATLAS_orange.txt', nm=True),
    _mk(['r', "r'", 'rp', 'rprime', 'F625W'], 0, 'Gunn', filename='SLOAN_SDSS.r.dat'),
    _mk('r-DECam', 0, 'DECam', filename='CTIO_DECam.r.dat'),
    _mk(['R', 'Rc', 'R_s'], 0, 'Johnson', 3.064e-23, 'Generic_Cousins.R.dat'),
    _mk(['i', "i'", 'ip', 'iprime', 'F775W'], -1, 'Gunn', filename='SLOAN_SDSS.i.dat'),
    _mk('i-DECam', -1, 'DECam', filename='CTIO_DECam.i.dat'),
    _mk(['I', 'Ic'], -1, 'Johnson', 2.416e-23, 'Generic_Cousins.I.dat'),
    _mk(['z_s', 'zs'], -2, 'Gunn', filename='PAN-STARRS_PS1.z.dat'),
    _mk(['z', "z'", 'zp', 'zprime'], -2, 'Gunn', filename='SLOAN_SDSS.z.dat'),
    _mk('z-DECam', -2, 'DECam', filename='CTIO_DECam.z.dat'),
    _mk('y', -3, 'Gunn', filename='PAN-STARRS_PS1.y.dat'),
    _mk('y-DECam', -3, 'DECam', filename='CTIO_DECam.Y.dat'),
    _mk('J', -2, 'UKIRT', 1.589e-23, 'Gemini_Flamingos2.J.dat'),
    _mk('H', -3, 'UKIRT', 1.021e-23, 'Gemini_Flamingos2.H.dat'),
    _mk(['K', 'Ks'], -4, 'UKIRT', 0.640e-23, 'Gemini_Flamingos2.Ks.dat'),
    _mk('L', -4, 'UKIRT', 0.285e-23),
] + [
    _mk(name, 0, 'JWST NIRCam', filename=f'JWST_NIRCam.{name}.dat', italics=False)
    for name in ('F070W', 'F090W', 'F115W', 'F150W', 'F182M', 'F200W', 'F250M', 'F277W', 'F300M', 'F335M', 'F356W',
                 'F360M', 'F444W')
] + [
    _mk(name, 0, 'JWST MIRI', filename=f'JWST_MIRI.{name}.dat', italics=False)
    for name in ('F560W', 'F770W', 'F1000W', 'F1130W', 'F1280W', 'F1500W', 'F1800W', 'F2100W', 'F2550W')
] + [
    _mk('pseudobolometric, curve_fit', italics=False),
    _mk('pseudobolometric, MCMC', italics=False),
    _mk('pseudobolometric, integration', italics=False),
    _mk('bolometric, curve_fit', italics=False),
    _mk('bolometric, MCMC', italics=False),
    _mk(['unknown', '?'], 0, 'unknown', italics=False),
]

Filter.order = [f.name for f in all_filters]

#: alias -> Filter (``filters.py:442-445``)
filtdict = {}
for _f in all_filters:
    for _n in _f.names:
        filtdict[_n] = _f


def as_filter(f):
    """Accept a ``Filter`` or any alias string."""
    if isinstance(f, Filter):
        return f
    try:
        return filtdict[str(f)]
    except KeyError:
        raise KeyError(f'unrecognised filter {f!r}') from None


_LEVELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'table_levels.npz')
_levels_bank = False


def table_levels(a, w, compress=True, interp=True, t_lo=INTERP_TMIN, t_hi=INTERP_TMAX):
    """The levels above one filter's full table ``(a, w)``, built and proved here: ``{'cool': (a', W', t_min, bound) or
    None, 'hot': ..., 'interp': (coef, t_min, bound) or None}`` (see :class:`PackedTables`)."""
    def quads(n):
        return (n + 3) // 4

    cool = compress_planck_table(a, w, orders=(8, 12, 16, 20, 24, 28, 32), max_tmin=COOL_TMIN,
                                 min_ratio=1.) if compress else None
    if cool is not None and quads(len(cool[0])) >= quads(len(a)):
        cool = None
    longer = len(a) if cool is None else len(cool[0])
    hot = compress_planck_table(a, w, orders=(8, 12, 16), max_tmin=HOT_TMIN, min_ratio=1.) if compress else None
    if hot is not None and (quads(len(hot[0])) >= quads(longer) or (cool is not None and hot[2] <= cool[2])):
        hot = None
    return {'cool': cool, 'hot': hot, 'interp': interp_planck_table(a, w, t_lo=t_lo, t_hi=t_hi) if interp else None}


def shipped_levels(filt, a, w, z):
    """The levels of ``filt``'s plain table at redshift ``z`` from the ones shipped for z = 0, or None (no shipped entry,
    or the table at hand is not the one the entry was made from).  With ``x = 1 + z``: ``a_k(z) = x a_k(0)`` and
    ``W_k(z) = x^3 W_k(0)``, so ``S(T; z) = x^3 S(T / x; 0)`` -- a Gauss rule of the z = 0 table is one of this table
    with nodes ``x a'`` and weights ``x^3 W'`` from ``x t_min`` on, and the interpolant of ``ln S(ln T; 0)`` is the one of
    this table on the grid shifted by ``ln x`` (``PackedTables.iu0``) plus ``3 ln x``: every proof carries over, to the
    rounding of the scaling."""
    global _levels_bank
    if _levels_bank is False:
        _levels_bank = np.load(_LEVELS) if os.path.exists(_LEVELS) else None
    bank = _levels_bank
    key = 'levels/' + (filt.filename or '')
    if bank is None or not filt.filename or key + '/n' not in bank.files:
        return None
    x = 1. + z
    # the entry's fingerprint: sample count and the two sums of the z = 0 table
    if int(bank[key + '/n']) != len(a) or not len(a):
        return None
    asum, wsum = float(bank[key + '/asum']), float(bank[key + '/wsum'])
    if abs(a.sum() - x * asum) > 1e-12 * abs(x * asum) or abs(w.sum() - x ** 3 * wsum) > 1e-12 * abs(x ** 3 * wsum):
        return None
    out = {}
    for name, tag in (('cool', 'c'), ('hot', 'h')):
        if f'{key}/{tag}a' in bank.files:
            out[name] = (x * bank[f'{key}/{tag}a'], x ** 3 * bank[f'{key}/{tag}w'], x * float(bank[f'{key}/{tag}tmin']),
                         float(bank[f'{key}/{tag}bound']))
        else:
            out[name] = None
    if key + '/icoef' in bank.files:
        coef = bank[key + '/icoef'].copy()
        coef[:, -1] += 3. * np.log1p(z)
        t0 = float(bank[key + '/itmin'])
        # (a filter proved from the first interval stays so: exactly the shifted grid's lower end)
        tmin = float(np.exp(np.log(INTERP_TMIN) + np.log1p(z))) if t0 == INTERP_TMIN else x * t0
        out['interp'] = (coef, tmin, float(bank[key + '/ibound']))
    else:
        out['interp'] = None
    return out


class PackedTables:
    """Concatenated ``(a_k, W_k)`` tables for a list of distinct filters (CSR layout: ``off[i]:off[i+1]``)."""

    def __init__(self, filters, z=0., cutoff_freq=np.inf, drop_zeros=True, compress=True, reddening=False, rv=3.1):
        self.filters = [as_filter(f) for f in filters]
        a_parts, w_parts, e_parts, off = [], [], [], [0]
        for f in self.filters:
            a, w = f.planck_table(z, cutoff_freq, drop_zeros)
            a_parts.append(a)
            w_parts.append(w)
            if reddening:
                e_parts.append(f.extinction_table(z, cutoff_freq, drop_zeros, rv))
            off.append(off[-1] + len(a))
        self.a = np.concatenate(a_parts) if a_parts else np.zeros(0)
        self.w = np.concatenate(w_parts) if w_parts else np.zeros(0)
        #: A_lambda / E(B-V) per sample (models with E(B-V) as a parameter), else None
        self.ext = (np.concatenate(e_parts) if e_parts else np.zeros(0)) if reddening else None
        self.off = np.asarray(off, dtype=np.int32)
        self.z = z
        self.cutoff_freq = cutoff_freq
        # Gauss-compressed companions, two levels (empty slice + t_min = inf where a level gains nothing):
        #   "cool": the shortest rule good down to COOL_TMIN (1 kK) -- cold photospheres rarely need the full table;
        #   "hot":  a still shorter one (8, 12 or 16 nodes) good down to HOT_TMIN (8 kK) at most -- where most of a
        #           fit's points are.
        # A level is kept only if it saves at least one quad of samples against the next longer table.
        # Third level: piecewise polynomials of ln S(ln T) per filter (None where it cannot be proved).
        # All three come from the levels SHIPPED for the filter's table at z = 0 (data/table_levels.npz, proved when it
        # was packed: tools/pack_table_levels.py) wherever the table is the plain one -- no cut-off, no reddening --,
        # scaled to this redshift (shipped_levels: S(T; z) = (1 + z)^3 S(T / (1 + z); 0), exactly); else they are
        # built and proved here (table_levels: ~30 ms per filter).
        self.levels_from = []
        levels = {'c': ([], [], [0], [], []), 'h': ([], [], [0], [], [])}
        self.iu0, self.ih, self.im = float(np.log(INTERP_TMIN)), float(np.log(INTERP_TMAX / INTERP_TMIN) / INTERP_M), INTERP_M
        plain = compress and not reddening and drop_zeros and not np.isfinite(cutoff_freq) and z > -1. and \
            os.environ.get('LCF_PACK_LEVELS') != '1'
        if plain:   # (the interpolants' origin moves with the redshift: the z = 0 blocks then serve as they are)
            self.iu0 = float(np.log(INTERP_TMIN) + np.log1p(z))
        self.icoef = np.zeros((len(self.filters), INTERP_M, INTERP_DEGREE + 1))
        self.itmin = np.full(len(self.filters), np.inf)
        self.ibound = np.full(len(self.filters), np.nan)
        for i in range(len(self.filters)):
            a, w = self.a[self.off[i]:self.off[i + 1]], self.w[self.off[i]:self.off[i + 1]]
            lv = shipped_levels(self.filters[i], a, w, z) if plain else None
            self.levels_from.append('shipped' if lv is not None else 'built')
            if lv is None:
                lo_hi = (float(np.exp(self.iu0)), float(np.exp(self.iu0 + self.im * self.ih))) if plain else \
                    (INTERP_TMIN, INTERP_TMAX)
                lv = table_levels(a, w, compress, interp=compress and not reddening, t_lo=lo_hi[0], t_hi=lo_hi[1])
            for key, comp in (('c', lv['cool']), ('h', lv['hot'])):
                aa, ww, oo, tt, bb = levels[key]
                if comp is None:
                    tt.append(np.inf)
                    bb.append(np.nan)
                else:
                    aa.append(comp[0])
                    ww.append(comp[1])
                    tt.append(comp[2])
                    bb.append(comp[3])
                oo.append(oo[-1] + (0 if comp is None else len(comp[0])))
            if lv['interp'] is not None:
                self.icoef[i], self.itmin[i], self.ibound[i] = lv['interp']
        #: per filter: largest relative error of each compressed level against the full sum, proved on the dense
        #: temperature grid from the level's t_min to 1e5 kK (NaN: the level does not exist)
        for key, names in (('c', ('ca', 'cw', 'coff', 'ctmin', 'cbound')), ('h', ('ha', 'hw', 'hoff', 'htmin', 'hbound'))):
            aa, ww, oo, tt, bb = levels[key]
            setattr(self, names[0], np.concatenate(aa) if aa else np.zeros(0))
            setattr(self, names[1], np.concatenate(ww) if ww else np.zeros(0))
            setattr(self, names[2], np.asarray(oo, dtype=np.int32))
            setattr(self, names[3], np.asarray(tt, dtype=np.float64))
            setattr(self, names[4], np.asarray(bb, dtype=np.float64))

    @property
    def itmax(self):
        """Upper end of the interpolants' range [kK]."""
        return float(np.exp(self.iu0 + self.im * self.ih))

    def interpolants(self, below=0):
        """``(coef[n_filters, m, 8], t_min[n_filters], u0, h)`` as the engines take them.  ``below`` > 0: the table
        starts that many intervals BELOW ``INTERP_TMIN`` (same interval width, so the intervals above are the engine's
        own; 10 intervals reach down to 0.94 kK: the per-epoch SED engine's candidates come from priors that start at
        1 kK, bolometric.py:729).  Every filter's part below is proved like the rest, or its ``t_min`` says where it holds."""
        if not below:
            return self.icoef, self.itmin, self.iu0, self.ih
        t_lo = float(np.exp(self.iu0 - below * self.ih))
        m = self.im + below
        coef = np.zeros((len(self.filters), m, INTERP_DEGREE + 1))
        tmin = np.full(len(self.filters), np.inf)
        if np.isfinite(self.itmin).any():
            for i in range(len(self.filters)):
                res = interp_planck_table(self.a[self.off[i]:self.off[i + 1]], self.w[self.off[i]:self.off[i + 1]],
                                          t_lo=t_lo, t_hi=float(np.exp(self.iu0 + self.im * self.ih)), m=m)
                if res is not None:
                    coef[i], tmin[i] = res[0], res[1]
        return coef, tmin, float(np.log(t_lo)), self.ih

    def index(self, f):
        return self.filters.index(as_filter(f))

    def samples_at(self, filt_idx, T, compressed=True):
        """Samples the device band sum walks for points of filter ``filt_idx`` at temperature ``T`` [kK]: the shortest
        table valid at that temperature (hot, else cool, else full), padded to quads as on the device."""
        f = np.asarray(filt_idx)
        T = np.broadcast_to(np.asarray(T, dtype=float), f.shape)
        n = np.diff(self.off)[f]
        if compressed:
            nc, nh = np.diff(self.coff)[f], np.diff(self.hoff)[f]
            n = np.where((nc > 0) & (T >= self.ctmin[f]), nc, n)
            n = np.where((nh > 0) & (T >= self.htmin[f]), nh, n)
        return np.where(T > 0., (n + 3) // 4 * 4, 0)
