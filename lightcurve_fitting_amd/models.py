"""Host-side mirror of the reference's model / prior interface, backed by the MI355X engine.

Same class names, constructor arguments, ``input_names``, call signatures and error behaviour as
``/root/reference/lightcurve_fitting/models.py`` for the likelihood path:

* ``Model.log_likelihood(lc, p, use_sigma=False, sigma_type='relative')``   (reference models.py:93-136)
* ``Model.__call__(t_in, f, *params)`` / ``evaluate``                        (models.py:86-87, per-model evaluate)
* ``temperature_radius``                                                     (models.py:231-269, 583-597, 727-755)
* ``UniformPrior`` / ``LogUniformPrior`` / ``GaussianPrior``                 (models.py:1048-1098)
* ``blackbody_to_filters``                                                   (models.py:1131-1165, ebv = 0)

All arithmetic on walkers runs in the HIP kernels (``csrc/``) through the C ABI (``include/lcf.h``).  This module
only marshals: it extracts the photometry columns, packs band tables, bakes the model's construction-time
constants, computes the SiFTO spline coefficients once, and caches one engine per (light curve, sigma mode).
There is no NumPy fallback for the model mathematics: without the native library and a GPU these calls raise.

Extension over the reference: ``p`` may be a C-contiguous ``(n, ndim)`` block (what emcee passes with
``vectorize=True``); the result then has shape ``(n,)``.
"""
import numpy as np

from . import engine as _eng
from .spline import not_a_knot_coefficients  # noqa: F401  (re-exported)
from .filters import Filter, PackedTables, as_filter, filtdict, _tables

# module-level constants with the reference's names (models.py:10-12, 1101-1102)
k_B = 0.08617333262145178
c3 = 5.38477047522316e-19
c4 = 8.357743635931361e-47
c1 = 0.0479924307336622
c2 = 281739904251.4432


# ---------------------------------------------------------------------------------------------------------------
# priors (models.py:1048-1098)
# ---------------------------------------------------------------------------------------------------------------
class Prior:
    """Base prior: ``__call__`` returns ``logp(p)`` strictly inside ``(p_min, p_max)`` and ``-inf`` otherwise."""
    kind = _eng.PRIOR_UNIFORM

    def __init__(self, p_min=-np.inf, p_max=np.inf):
        self.p_min = p_min
        self.p_max = p_max

    def __call__(self, p):
        if self.p_min < p < self.p_max:
            return self.logp(p)
        return -np.inf

    def logp(self, p):
        raise NotImplementedError

    def descriptor(self):
        """(kind, p_min, p_max, mean, stddev) as the C ABI wants it."""
        return (self.kind, float(self.p_min), float(self.p_max), float(getattr(self, 'mean', 0.)),
                float(getattr(self, 'stddev', 1.)))


class UniformPrior(Prior):
    """dP/dp proportional to 1."""
    kind = _eng.PRIOR_UNIFORM

    def logp(self, p):
        return np.zeros_like(p)


class LogUniformPrior(Prior):
    """dP/dp proportional to 1/p."""
    kind = _eng.PRIOR_LOG_UNIFORM

    def __init__(self, p_min=0., p_max=np.inf):
        if p_min < 0.:
            raise ValueError('a log-uniform prior cannot have negative limits')
        super().__init__(p_min, p_max)

    def logp(self, p):
        return -np.log(p)


class GaussianPrior(Prior):
    """Gaussian centred on ``mean`` with standard deviation ``stddev`` (unnormalised), truncated to the bounds."""
    kind = _eng.PRIOR_GAUSSIAN

    def __init__(self, p_min=-np.inf, p_max=np.inf, mean=0., stddev=1.):
        super().__init__(p_min, p_max)
        self.mean = mean
        self.stddev = stddev

    def logp(self, p):
        return -0.5 * ((p - self.mean) / self.stddev) ** 2.


# ---------------------------------------------------------------------------------------------------------------
# light-curve access (duck typed: astropy Table, lightcurve_fitting_amd.lightcurve.LC, dict of arrays)
# ---------------------------------------------------------------------------------------------------------------
def _column(lc, name):
    col = lc[name]
    return np.asarray(getattr(col, 'data', col))


def _photometry(lc, quantity):
    t = np.asarray(_column(lc, 'MJD'), dtype=np.float64)
    y = np.asarray(_column(lc, quantity), dtype=np.float64)
    dy = np.asarray(_column(lc, 'd' + quantity), dtype=np.float64)
    filts = [as_filter(f) for f in _column(lc, 'filter')]
    return t, filts, y, dy


class _BoundEngine:
    """An engine together with the photometry it was created from (host copies, for the content check)."""
    __slots__ = ('engine', 'mode', 't', 'filters', 'y', 'dy')

    def __init__(self, engine, mode, t, filters, y, dy):
        self.engine, self.mode = engine, mode
        self.t, self.y, self.dy = t.copy(), y.copy(), dy.copy()
        self.filters = filters.copy()

    def holds(self, t, filters, y, dy):
        """True if the engine's photometry equals these columns (NaNs compare equal to NaNs)."""
        if t.shape != self.t.shape:
            return False
        for mine, theirs in ((self.y, y), (self.dy, dy), (self.t, t)):
            if mine.shape != theirs.shape or not np.array_equal(mine, theirs, equal_nan=True):
                return False
        return filters.shape == self.filters.shape and bool(np.all(filters == self.filters))


def _index_filters(filts):
    """Distinct filters (first-seen order) and the per-point index into them."""
    uniq = list(dict.fromkeys(filts))
    lookup = {f: i for i, f in enumerate(uniq)}
    return uniq, np.array([lookup[f] for f in filts], dtype=np.int32)


# ---------------------------------------------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------------------------------------------
class Model:
    """An analytical model, defined by its parameters and evaluated on the GPU."""

    input_names = []
    units = []
    output_quantity = 'lum'
    model_id = None
    #: E(B-V) is a model parameter: band-table weights are reddened per walker (full tables only)
    reddened = False
    #: photometry-bound engines one model keeps (each holds a copy of a light curve on the device)
    max_bound_engines = 8

    def __init__(self, lc=None, redshift=0.):
        if redshift:
            self.z = redshift
        elif lc is not None and 'redshift' in getattr(lc, 'meta', {}):
            self.z = lc.meta['redshift']
        else:
            self.z = 0.
        # instance-level copies: lightcurve_mcmc(use_sigma=True) appends '\\sigma' (fitting.py:74-76)
        self.input_names = list(type(self).input_names)
        self.units = list(type(self).units)
        self._engines = {}   # evaluation engines (model grids), keyed by the (times, filters) they were built for
        self._bound = []     # engines bound to photometry, most recently used first (see engine_for)
        self.device = 0

    @property
    def nparams(self):
        return len(self.input_names)

    @property
    def n_model_params(self):
        return len(type(self).input_names)

    @property
    def axis_labels(self):
        return ['${}$ ({})'.format(var, unit) if unit else '${}$'.format(var)
                for var, unit in zip(self.input_names, self.units)]

    def __repr__(self):
        return f'<{self.__class__.__name__}: z={self.z:.3f}>'

    def __call__(self, *args, **kwargs):
        return self.evaluate(*args, **kwargs)

    # --- hooks for subclasses ----------------------------------------------------------------------------------
    def _consts(self):
        return []

    def _companion_tables(self, uniq_filters):
        return None

    # --- engine management -------------------------------------------------------------------------------------
    def make_engine(self, t, filts, y, dy, use_sigma=False, sigma_type='relative', priors=None, device=None):
        """Build a device engine for explicit photometry arrays (``filts``: Filter objects or aliases)."""
        if sigma_type == 'relative':
            st = _eng.SIGMA_RELATIVE
        elif sigma_type == 'absolute':
            st = _eng.SIGMA_ABSOLUTE
        else:
            raise Exception('sigma_type must either be "relative" or "absolute"')
        import time
        t0 = time.perf_counter()
        filts = [as_filter(f) for f in filts]
        uniq, idx = _index_filters(filts)
        tabs = PackedTables(uniq, z=self.z, compress=not self.reddened, reddening=self.reddened)
        t1 = time.perf_counter()
        pri = None if priors is None else [p.descriptor() for p in priors]
        eng = _eng.Engine(self.model_id, self.n_model_params, self._consts(), t, y, dy, idx, tabs.off, tabs.a, tabs.w,
                          use_sigma=use_sigma, sigma_type=st, priors=pri,
                          companion=self._companion_tables(uniq), device=self.device if device is None else device,
                          ctab=None if self.reddened else (tabs.coff, tabs.ca, tabs.cw, tabs.ctmin),
                          htab=None if self.reddened else (tabs.hoff, tabs.ha, tabs.hw, tabs.htmin),
                          itab=None if self.reddened else (tabs.icoef, tabs.itmin, tabs.iu0, tabs.ih),
                          tab_ext=tabs.ext)
        eng.tables, eng.filt_idx = tabs, idx   # what was packed (measurement tools count the samples a fit executes)
        #: seconds: band tables packed on the host (their levels shipped or built: tabs.levels_from) / engine created
        eng.timings = {'tables_s': t1 - t0, 'create_s': time.perf_counter() - t1, 'levels': sorted(set(tabs.levels_from))}
        return eng

    def engine_for(self, lc, use_sigma=False, sigma_type='relative', priors=None):
        """Engine bound to the photometry ``lc`` holds NOW (sigma mode and prior set as given).

        The reference reads the light-curve columns on every ``log_likelihood`` call (models.py:116-119); an engine
        copies them to the device once.  Engines are therefore cached by the CONTENT of the four columns, compared on
        every call: editing ``lc['lum']`` in place, or a new light-curve object at a recycled address, gets a new
        engine.  An engine that leaves the cache is only dropped, never destroyed here -- samplers and closures that
        were handed it keep it alive."""
        if sigma_type not in ('relative', 'absolute'):
            raise Exception('sigma_type must either be "relative" or "absolute"')
        t = np.asarray(_column(lc, 'MJD'), dtype=np.float64)
        y = np.asarray(_column(lc, self.output_quantity), dtype=np.float64)
        dy = np.asarray(_column(lc, 'd' + self.output_quantity), dtype=np.float64)
        filt_col = np.asarray(_column(lc, 'filter'), dtype=object)
        mode = (bool(use_sigma), sigma_type, None if priors is None else tuple(p.descriptor() for p in priors))
        for k, bound in enumerate(self._bound):
            if bound.mode == mode and bound.holds(t, filt_col, y, dy):
                if k:  # most recently used first
                    self._bound.insert(0, self._bound.pop(k))
                return bound.engine
        filts = [as_filter(f) for f in filt_col]
        eng = self.make_engine(t, filts, y, dy, use_sigma, sigma_type, priors)
        self._bound.insert(0, _BoundEngine(eng, mode, t, filt_col, y, dy))
        del self._bound[self.max_bound_engines:]
        return eng

    # --- the reference's public surface --------------------------------------------------------------------------
    def log_likelihood(self, lc, p, use_sigma=False, sigma_type='relative'):
        """Gaussian log-likelihood of the photometry in ``lc`` given parameters ``p`` (models.py:93-136).

        ``p`` 1-D -> float; ``p`` of shape (n, ndim) -> array (n,)."""
        eng = self.engine_for(lc, use_sigma, sigma_type)
        p = np.asarray(p, dtype=np.float64)
        out = eng.log_likelihood(p)
        return float(out[0]) if p.ndim == 1 else out

    def _eval_engine(self, t_in, f, scalar_params=True):
        """Engine for model evaluation at (t, f): pointwise when the lengths agree and the parameters are scalars
        (``T.ndim == 1 and len(T) == len(filters)``, models.py:1161), else the dense filters x times grid
        (models.py:1163-1164).  Returns (engine, grid_shape or None)."""
        t_in = np.atleast_1d(np.asarray(t_in, dtype=np.float64))
        single = isinstance(f, (Filter, str))
        filts = [as_filter(f)] if single else [as_filter(x) for x in f]
        if not single and scalar_params and t_in.ndim == 1 and len(t_in) == len(filts):
            t, fl, shape = t_in, filts, None
        else:
            t = np.tile(t_in.ravel(), len(filts))
            fl = [x for x in filts for _ in range(t_in.size)]
            shape = (len(filts), t_in.size)
        key = ('eval', t.tobytes(), tuple(x.name for x in fl))
        eng = self._engines.get(key)
        if eng is None:
            eng = self.make_engine(t, fl, np.zeros(len(t)), np.ones(len(t)))
            self._engines = {key: eng}  # one evaluation grid at a time; the old engine dies with its last reference
        return eng, shape

    def evaluate(self, t_in, f, *params):
        """Model light curve(s).  Scalar parameters -> (npoints,) [pointwise, when ``len(t_in) == len(f)``] or
        (nfilters, ntimes) [grid]; array parameters of length n -> the grid with a trailing axis of length n, as in
        the reference (whose pointwise branch needs a one-dimensional temperature)."""
        if len(params) != self.n_model_params:
            raise TypeError(f'{type(self).__name__} takes {self.n_model_params} parameters, got {len(params)}')
        cols = np.broadcast_arrays(*[np.asarray(x, dtype=np.float64) for x in params])
        scalar = cols[0].ndim == 0
        eng, shape = self._eval_engine(t_in, f, scalar)
        P = np.column_stack([np.atleast_1d(c).ravel() for c in cols])
        y = eng.evaluate(P)  # (n, npoints)
        y = y[0] if scalar else y.T
        if shape is not None:
            y = y.reshape(shape if scalar else shape + (len(P),))
        return y

    def temperature_radius(self, t_in, *params):
        """Blackbody temperature [kK] and radius [1000 Rsun] at times ``t_in`` (same broadcasting as evaluate)."""
        t_in = np.atleast_1d(np.asarray(t_in, dtype=np.float64))
        eng, _ = self._eval_engine(t_in, [self._any_filter()] * len(t_in))
        n_tr = self._n_tr_params()
        cols = np.broadcast_arrays(*[np.asarray(x, dtype=np.float64) for x in params[:n_tr]])
        scalar = cols[0].ndim == 0
        P = np.column_stack([np.atleast_1d(c).ravel() for c in cols])
        if P.shape[1] < self.n_model_params:  # parameters that do not enter T, R
            P = np.column_stack([P, np.ones((len(P), self.n_model_params - P.shape[1]))])
        T, R = eng.temperature_radius(P)
        return (T[0], R[0]) if scalar else (T.T, R.T)

    def _n_tr_params(self):
        return self.n_model_params

    @staticmethod
    def _any_filter():
        return filtdict['r']


class BaseShockCooling(Model):
    """Sapir & Waxman (2017) / Rabinak & Waxman (2011) shock cooling; constants per models.py:192-226."""

    def __init__(self, lc=None, redshift=0., n=1.5, RW=False):
        super().__init__(lc, redshift=redshift)
        if n == 1.5:
            self.n, self.A, self.a, self.alpha = 1.5, 0.94, 1.67, 0.8
            self.epsilon_1, self.epsilon_2, self.L_0, self.T_0, self.Tph_to_Tcol = 0.027, 0.086, 2.0e42, 1.61, 1.1
        elif n == 3.:
            self.n, self.A, self.a, self.alpha = 3., 0.79, 4.57, 0.73
            self.epsilon_1, self.epsilon_2, self.L_0, self.T_0, self.Tph_to_Tcol = 0.016, 0.175, 2.1e42, 1.69, 1.0
        else:
            raise ValueError('n can only be 1.5 or 3')
        self.epsilon_T = 2 * self.epsilon_1 - 0.5
        self.epsilon_L = -2 * self.epsilon_2
        self.RW = bool(RW)
        if RW:
            self.a = 0.
            self.Tph_to_Tcol = 1.2

    def __repr__(self):
        return f'<{self.__class__.__name__}: z={self.z:.3f}, n={self.n:.1f}, RW={self.RW}>'

    def _consts(self):
        return [self.A, self.a, self.alpha, self.epsilon_1, self.epsilon_2, self.L_0, self.T_0, self.Tph_to_Tcol]

    @staticmethod
    def t_min(p, kappa=1.):
        """Minimum validity time (models.py:276-287)."""
        v_s, f_rho_M, R = p[0], p[2], p[3]
        t_exp = p[4] if len(p) > 4 else 0.
        return 0.2 * R / v_s * np.maximum(0.5, R ** 0.4 * (f_rho_M * kappa) ** -0.2 * v_s ** -0.7) + t_exp

    @staticmethod
    def t_max(p, kappa=1.):
        """Maximum validity time (models.py:290-298)."""
        R = p[3]
        t_exp = p[4] if len(p) > 4 else 0.
        return 7.4 * (R / kappa) ** 0.55 + t_exp


class ShockCooling(BaseShockCooling):
    """Physical parameters v_s, M_env, f_rho M, R, t_0 (models.py:301-353)."""
    model_id = _eng.MODEL_SHOCK_COOLING
    input_names = ['v_\\mathrm{s*}', 'M_\\mathrm{env}', 'f_\\rho M', 'R', 't_0']
    units = ['10^8.5 cm/s', 'Msun', 'Msun', '10^13 cm', 'd']


class ShockCooling2(BaseShockCooling):
    """Scaling parameters T_1, L_1, t_tr, t_0 (models.py:356-430)."""
    model_id = _eng.MODEL_SHOCK_COOLING2
    input_names = ['T_1', 'L_1', 't_\\mathrm{tr}', 't_0']
    units = ['kK', '10^42 erg/s', 'd', 'd']

    @staticmethod
    def t_min(p, kappa=1.):
        return NotImplemented

    def t_max(self, p, kappa=1.):
        T_1 = p[0]
        t_exp = p[3] if len(p) > 3 else 0.
        return (8.12 / T_1) ** (self.epsilon_T ** -1) + t_exp


class ShockCooling3(BaseShockCooling):
    """``ShockCooling`` with the luminosity distance d_L [Mpc] and the reddening E(B-V) as free parameters; fits
    ``'flux'`` instead of ``'lum'`` (models.py:433-504): ``flux = c4 * Lnu(E(B-V)) / d_L ** 2``, the blackbody being
    reddened sample by sample inside the band integral with the Fitzpatrick (1999) law, R_V = 3.1
    (``extinction.py``; that law is third-party arithmetic for the reference -- see its header for how it is pinned)."""
    model_id = _eng.MODEL_SHOCK_COOLING3
    input_names = ['v_\\mathrm{s*}', 'M_\\mathrm{env}', 'f_\\rho M', 'R', 'd_L', 'E(B-V)', 't_0']
    units = ['10^8.5 cm/s', 'Msun', 'Msun', '10^13 cm', 'Mpc', 'mag', 'd']
    output_quantity = 'flux'
    reddened = True

    def temperature_radius(self, t_in, v_s, M_env, f_rho_M, R, t_exp=0.):
        """Same (T, R) as ``ShockCooling`` (``BaseShockCooling.temperature_radius``, models.py:231-269)."""
        one = np.ones_like(np.asarray(v_s, dtype=np.float64))
        return super().temperature_radius(t_in, v_s, M_env, f_rho_M, R, one, 0. * one, t_exp * one)

    @staticmethod
    def t_min(p, kappa=1.):
        return BaseShockCooling.t_min([p[0], p[1], p[2], p[3], p[6] if len(p) > 6 else 0.], kappa=kappa)

    @staticmethod
    def t_max(p, kappa=1.):
        return BaseShockCooling.t_max([p[0], p[1], p[2], p[3], p[6] if len(p) > 6 else 0.], kappa=kappa)


class ShockCooling4(Model):
    """Morag, Sapir & Waxman (2023) form (models.py:507-657), quirks of the reference included."""
    model_id = _eng.MODEL_SHOCK_COOLING4
    input_names = ['v_\\mathrm{s*}', 'M_\\mathrm{env}', 'f_\\rho M', 'R', 't_0']
    units = ['10^8.5 cm/s', 'Msun', 'Msun', '10^13 cm', 'd']

    def __init__(self, lc=None, redshift=0.):
        super().__init__(lc, redshift=redshift)
        self.A, self.a, self.alpha = 0.9, 2., 0.5
        self.L_br_0, self.T_col_br_0 = 3.69e42, 8.19
        self.t_min_0, self.t_br_0, self.t_07eV_0, self.t_tr_0 = 0.012, 0.036, 6.86, 19.5

    def _consts(self):
        return [self.A, self.a, self.alpha, self.L_br_0, self.T_col_br_0, self.t_br_0, self.t_tr_0]

    def t_min(self, p, kappa=1.):
        t_exp = p[4] if len(p) > 4 else 0.
        return self.t_min_0 * p[3] + t_exp

    def t_max(self, p, kappa=1.):
        v_s, M_env, f_rho_M, R, t_exp, *_ = p
        t_07eV = self.t_07eV_0 * R ** 0.56 * v_s ** 0.16 * kappa ** -0.61 * f_rho_M ** -0.06
        t_tr = self.t_tr_0 ** np.sqrt(kappa * M_env / v_s)  # sic: ** as in models.py:656
        return np.minimum(t_07eV, t_tr / self.a) + t_exp


_SIFTO_COLUMNS = ('Epoch', 'U', 'B', 'V', 'g', 'r', 'i')


def sifto_template():
    """SiFTO SN Ia template without its first three (~0) rows (models.py:660-661): columns Epoch,U,B,V,g,r,i."""
    return _tables()['template/sifto'][3:]


class BaseCompanionShocking(Model):
    """Kasen (2010) companion-shocking component + SiFTO template scaled to the observed peak (models.py:665-845)."""
    _kasen_factor = {}
    _sifto_factor = {}
    _dt_param = {}

    def __init__(self, lc, redshift=0.):
        super().__init__(lc, redshift=redshift)
        if 'lum' not in getattr(lc, 'colnames', list(getattr(lc, 'keys', lambda: [])())):
            if hasattr(lc, 'calcAbsMag'):
                if 'absmag' not in lc.colnames:
                    lc.calcAbsMag()
                lc.calcLum()
        tab = sifto_template()
        self._knots = np.ascontiguousarray(tab[:, 0])
        filts = [as_filter(f) for f in _column(lc, 'filter')]
        lum = np.asarray(_column(lc, 'lum'), dtype=np.float64)
        have_dlt40 = filtdict['DLT40'] in filts
        self.sifto = {}
        for filt in dict.fromkeys(filts):  # models.py:701-717
            if filt.name == 'unfilt.' and have_dlt40:
                column, scale_by = 'r', filtdict['DLT40']
            elif filt.name == 'DLT40':
                column, scale_by = 'r', filt
            elif filt.char in _SIFTO_COLUMNS[1:]:
                column, scale_by = filt.char, filt
            else:
                raise Exception('No SiFTO template for filter ' + filt.name)
            col = tab[:, _SIFTO_COLUMNS.index(column)]
            peak = np.max(lum[[f == scale_by for f in filts]])
            self.sifto[filt] = not_a_knot_coefficients(self._knots, col * peak / np.max(col))

    def _companion_tables(self, uniq_filters):
        nk = len(self._knots)
        coef = np.zeros((len(uniq_filters), nk - 1, 4))
        kp, sp, dtp = [], [], []
        for i, f in enumerate(uniq_filters):
            if f not in self.sifto:
                raise Exception('No SiFTO template for filter ' + f.name)
            coef[i] = self.sifto[f]
            kp.append(self._kasen_factor.get(f.char, -1))
            sp.append(self._sifto_factor.get(f.char, -1))
            dtp.append(self._dt_param.get(f.name, -1))
        return kp, sp, dtp, self._knots, coef

    def _n_tr_params(self):
        return 3

    @staticmethod
    def t_min(p):
        return p[3] + p[4] * sifto_template()[:, 0].min()

    @staticmethod
    def t_max(p):
        return p[3] + p[4] * sifto_template()[:, 0].max()


class CompanionShocking(BaseCompanionShocking):
    """Factors on the r and i templates and on the U shock component (models.py:848-918)."""
    model_id = _eng.MODEL_COMPANION_SHOCKING
    input_names = ['t_0', 'a', 'M v^7', 't_\\mathrm{max}', 's', 'r_r', 'r_i', 'r_U']
    units = ['d', '10^13 cm', 'M_Ch (10^9 cm/s)^7', 'd', '', '', '', '']
    _kasen_factor = {'U': 7}      # keyed on filt.char (models.py:913-916)
    _sifto_factor = {'r': 5, 'i': 6}


class CompanionShocking2(BaseCompanionShocking):
    """Time offsets for the U and i templates (models.py:921-980)."""
    model_id = _eng.MODEL_COMPANION_SHOCKING2
    input_names = ['t_0', 'a', 'M v^7', 't_\\mathrm{max}', 's', '\\Delta t_U', '\\Delta t_i']
    units = ['d', '10^13 cm', 'M_Ch (10^9 cm/s)^7', 'd', '', 'd', 'd']
    _dt_param = {'U': 5, 'i': 6}   # keyed on the filter itself (models.py:804-807)


class CompanionShocking3(BaseCompanionShocking):
    """Time offsets + Brown et al. (2012) viewing-angle dependence (models.py:983-1045)."""
    model_id = _eng.MODEL_COMPANION_SHOCKING3
    input_names = ['t_0', 'a', '\\theta', 't_\\mathrm{max}', 's', '\\Delta t_U', '\\Delta t_i']
    units = ['d', '10^13 cm', 'deg', 'd', '', 'd', 'd']
    _dt_param = {'U': 5, 'i': 6}


class Blackbody(Model):
    """Direct (T, R) blackbody -- the spectrum ``spectrum_mcmc`` fits per epoch (bolometric.py:154-164)."""
    model_id = _eng.MODEL_BLACKBODY
    input_names = ['T', 'R']
    units = ['kK', '1000 Rsun']

    def _eval_engine(self, t_in, f, scalar_params=True):
        # not a reference Model: one (filter, time) point per entry even for arrays of candidates, result (npoints, n)
        return super()._eval_engine(t_in, f, True)


def blackbody_to_filters(filters, T, R, z=0., cutoff_freq=np.inf, ebv=0., variant=None):
    """Band-averaged L_nu of blackbodies through filters (models.py:1131-1165).

    Pointwise when ``T`` is 1-D with one entry per filter, else every filter for every (T, R): result shape
    ``(nfilters,) + T.shape``.  ``ebv``: one E(B-V) for the whole call (the reddening goes into the table weights at
    pack time); per-walker reddening inside a fit is ``ShockCooling3``'s job.  ``variant``: the band-sum level
    (``Engine.set_variant``; None = the engine's default, the interpolants of ln S(ln T))."""
    if np.ndim(ebv) != 0:
        raise NotImplementedError('blackbody_to_filters takes one scalar E(B-V) per call')
    T = np.array(T, dtype=np.float64)
    R = np.array(R, dtype=np.float64)
    if T.shape != R.shape:
        raise Exception('T & R must have the same shape')
    filts = [as_filter(f) for f in (filters if not isinstance(filters, (Filter, str)) else [filters])]
    uniq, idx = _index_filters(filts)
    if ebv:
        tabs = PackedTables(uniq, z=z, cutoff_freq=cutoff_freq, compress=False, reddening=True)
        tabs.w = tabs.w * 10. ** (-0.4 * float(ebv) * tabs.ext)
        ctab = None
    else:
        tabs = PackedTables(uniq, z=z, cutoff_freq=cutoff_freq)
        ctab = (tabs.coff, tabs.ca, tabs.cw, tabs.ctmin)
    itab = None if ebv else (tabs.icoef, tabs.itmin, tabs.iu0, tabs.ih)
    eng = _eng.Engine(_eng.MODEL_BLACKBODY, 2, [], np.zeros(1), np.zeros(1), np.ones(1), np.zeros(1, dtype=np.int32),
                      tabs.off, tabs.a, tabs.w, ctab=ctab, itab=itab)  # (pointwise kernel: interpolants + cool level)
    if variant is not None:
        eng.set_variant(variant)
    try:
        if T.ndim == 1 and len(T) == len(filts):
            return eng.blackbody_to_filters(idx, T, R)
        flat_T, flat_R = T.ravel(), R.ravel()
        out = eng.blackbody_to_filters(np.repeat(idx, flat_T.size), np.tile(flat_T, len(filts)),
                                       np.tile(flat_R, len(filts)))
        return out.reshape((len(filts),) + T.shape)
    finally:
        eng.close()
