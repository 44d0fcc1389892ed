"""ctypes binding of the C ABI in ``include/lcf.h`` (``csrc/liblcf_hip.so``, built for gfx950).

There is NO CPU fallback: if the shared library is missing or no MI355X is visible, constructing an
:class:`Engine` raises.  Host-side work here is limited to marshalling (contiguous float64 blocks in, arrays out).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LCF_HIP_LIB') or os.path.join(_HERE, 'csrc', 'liblcf_hip.so')

LCF_ABI_VERSION = 8
N_CONSTS = 12

MODEL_SHOCK_COOLING = 1
MODEL_SHOCK_COOLING2 = 2
MODEL_SHOCK_COOLING3 = 3
MODEL_SHOCK_COOLING4 = 4
MODEL_COMPANION_SHOCKING = 5
MODEL_COMPANION_SHOCKING2 = 6
MODEL_COMPANION_SHOCKING3 = 7
MODEL_BLACKBODY = 8

PRIOR_UNIFORM, PRIOR_LOG_UNIFORM, PRIOR_GAUSSIAN = 0, 1, 2
SIGMA_RELATIVE, SIGMA_ABSOLUTE = 0, 1
SPLIT_IDENTITY, SPLIT_RANDOM, SPLIT_HOST = 0, 1, 2

STATUS_NAMES = {0: 'LCF_OK', 1: 'LCF_ERR_INVALID_ARGUMENT', 2: 'LCF_ERR_HIP', 3: 'LCF_ERR_NO_DEVICE',
                4: 'LCF_ERR_OUT_OF_MEMORY', 5: 'LCF_ERR_UNSUPPORTED', 6: 'LCF_ERR_NAN_LOGPROB', 7: 'LCF_ERR_STATE'}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class LcfPrior(C.Structure):
    _fields_ = [('kind', C.c_int32), ('reserved', C.c_int32), ('p_min', C.c_double), ('p_max', C.c_double),
                ('mean', C.c_double), ('stddev', C.c_double)]


class LcfProblem(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('model', C.c_int32), ('n_par', C.c_int32), ('use_sigma', C.c_int32),
                ('sigma_type', C.c_int32), ('n_filters', C.c_int32), ('n_points', C.c_int64),
                ('consts', C.c_double * N_CONSTS),
                ('t', _dp), ('y', _dp), ('dy', _dp), ('filt_idx', _ip), ('tab_off', _ip), ('tab_a', _dp),
                ('tab_w', _dp), ('tab_ext', _dp), ('ctab_off', _ip), ('ctab_a', _dp), ('ctab_w', _dp), ('ctab_tmin', _dp),
                ('htab_off', _ip), ('htab_a', _dp), ('htab_w', _dp), ('htab_tmin', _dp),
                ('itab_coef', _dp), ('itab_tmin', _dp), ('itab_m', C.c_int32), ('reserved2', C.c_int32),
                ('itab_u0', C.c_double), ('itab_h', C.c_double),
                ('filt_kasen_par', _ip), ('filt_sifto_par', _ip), ('filt_dt_par', _ip),
                ('n_knots', C.c_int32), ('reserved', C.c_int32), ('spline_knots', _dp), ('spline_coef', _dp),
                ('priors', C.POINTER(LcfPrior))]


class LcfError(RuntimeError):
    """A non-zero ``lcf_status`` from the native library."""

    def __init__(self, status, message):
        self.status = status
        super().__init__(f'{STATUS_NAMES.get(status, status)}: {message}')


_lib = None

#: every symbol include/lcf.h declares: (name, restype, argtypes)
SIGNATURES = [
    ('lcf_abi_version', C.c_int32, []),
    ('lcf_last_error', C.c_char_p, []),
    ('lcf_device_count', C.c_int32, []),
    ('lcf_engine_create', C.c_int, [C.POINTER(LcfProblem), C.c_int32, C.POINTER(C.c_void_p)]),
    ('lcf_engine_destroy', None, [C.c_void_p]),
    ('lcf_engine_ndim', C.c_int32, [C.c_void_p]),
    ('lcf_engine_npoints', C.c_int64, [C.c_void_p]),
    ('lcf_engine_samples_per_eval', C.c_int64, [C.c_void_p]),
    ('lcf_engine_set_variant', C.c_int, [C.c_void_p, C.c_int32]),
    ('lcf_log_likelihood', C.c_int, [C.c_void_p, C.c_int64, _dp, _dp]),
    ('lcf_log_posterior', C.c_int, [C.c_void_p, C.c_int64, _dp, _dp]),
    ('lcf_log_likelihood_dev', C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('lcf_log_posterior_dev', C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('lcf_model_evaluate', C.c_int, [C.c_void_p, C.c_int64, _dp, _dp]),
    ('lcf_temperature_radius', C.c_int, [C.c_void_p, C.c_int64, _dp, _dp, _dp]),
    ('lcf_blackbody_to_filters', C.c_int, [C.c_void_p, C.c_int64, _ip, _dp, _dp, _dp]),
    ('lcf_profile_loglike_kernel', C.c_int, [C.c_void_p, C.c_int64, _dp, C.c_int32, _dp]),
    ('lcf_sampler_create', C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_double, C.POINTER(C.c_void_p)]),
    ('lcf_sampler_destroy', None, [C.c_void_p]),
    ('lcf_sampler_set_state', C.c_int, [C.c_void_p, _dp]),
    ('lcf_sampler_get_state', C.c_int, [C.c_void_p, _dp, _dp]),
    ('lcf_sampler_run', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _ip, C.c_int32]),
    ('lcf_sampler_run_async', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _ip, C.c_int32]),
    ('lcf_sampler_wait', C.c_int, [C.c_void_p]),
    ('lcf_population_run', C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                     _dp]),
    ('lcf_sampler_get_chain', C.c_int, [C.c_void_p, _dp, _dp]),
    ('lcf_sampler_reserve_chain', C.c_int, [C.c_void_p, C.c_int64]),
    ('lcf_sampler_get_naccepted', C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    ('lcf_sampler_get_snapshot', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('lcf_sampler_last_run_ms', C.c_double, [C.c_void_p]),
    ('lcf_sampler_begin', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _ip, C.c_int32]),
    ('lcf_sampler_propose', C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    ('lcf_sampler_evaluate', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ('lcf_sampler_accept', C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    ('lcf_sampler_half_step', C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ('lcf_sampler_newlp_ptr', C.c_void_p, [C.c_void_p]),
    ('lcf_sampler_one_launch', C.c_int32, [C.c_void_p]),
    ('lcf_sampler_set_half_step_kernel', C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    ('lcf_sampler_last_run_kernel', C.c_int32, [C.c_void_p]),
    ('lcf_sampler_last_run_launches', C.c_int64, [C.c_void_p]),
    ('lcf_sampler_half_step_rows', C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ('lcf_sampler_rows_ptr', C.c_void_p, [C.c_void_p, C.POINTER(C.c_int32)]),
    ('lcf_sampler_check', C.c_int, [C.c_void_p]),
    ('lcf_comm_probe', C.c_int, [C.c_char_p]),
    ('lcf_comm_unique_id', C.c_int, [C.c_char_p, C.c_void_p]),
    ('lcf_comm_create', C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ('lcf_comm_destroy', None, [C.c_void_p]),
    ('lcf_comm_count', C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ('lcf_comm_time_allgather', C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, _dp]),
    ('lcf_sampler_run_sharded', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _ip, C.c_int32]),
    ('lcf_sampler_mailbox_export', C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    ('lcf_sampler_mailbox_connect', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    ('lcf_sampler_run_peers', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _ip, C.c_int32]),
    ('lcf_sampler_run_peers_async', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _ip, C.c_int32]),
    ('lcf_sampler_board_export', C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    ('lcf_sampler_board_connect', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    ('lcf_sampler_run_rows', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _ip, C.c_int32]),
    ('lcf_sampler_run_rows_async', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _ip, C.c_int32]),
    ('lcf_sed_create', C.c_int, [C.c_int32, _ip, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp, C.c_int32, C.c_double,
                                 C.c_double, C.c_int32, C.POINTER(C.c_void_p)]),
    ('lcf_sed_destroy', None, [C.c_void_p]),
    ('lcf_sed_set_observations', C.c_int, [C.c_void_p, C.c_int64, _ip, _ip, _dp, _dp]),
    ('lcf_sed_log_likelihood', C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, _dp, C.c_int32, C.c_int32, _dp,
                                         _dp]),
]


def load_library(path=None):
    """dlopen the native library and attach prototypes.  Raises ``OSError`` with build instructions if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    try:
        # PyTorch ships its own HIP/HSA runtime.  Two runtimes in one process do not coexist, so when torch is
        # installed it is loaded FIRST and this library binds to the runtime already in the process (same soname).
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise OSError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` or '
                      f'`make -C {os.path.dirname(path)}` (hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = C.CDLL(path)
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.lcf_abi_version() != LCF_ABI_VERSION:
        raise OSError(f'{path}: ABI version {lib.lcf_abi_version()} != {LCF_ABI_VERSION}')
    _lib = lib
    return lib


def _check(status):
    if status != 0:
        raise LcfError(status, load_library().lcf_last_error().decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a, typ=_dp):
    return a.ctypes.data_as(typ)


class Engine:
    """One light curve + one model instance resident on one MI355X.

    Parameters are the fields of ``lcf_problem`` (see ``include/lcf.h``): ``priors`` is a sequence of
    ``(kind, p_min, p_max, mean, stddev)`` or ``None``."""

    def __init__(self, model_id, n_par, consts, t, y, dy, filt_idx, tab_off, tab_a, tab_w, use_sigma=False,
                 sigma_type=SIGMA_RELATIVE, priors=None, companion=None, device=0, ctab=None, tab_ext=None,
                 htab=None, itab=None):
        lib = load_library()
        self._lib = lib
        self._h = C.c_void_p()
        keep = [_f64(t), _f64(y), _f64(dy), _i32(filt_idx), _i32(tab_off), _f64(tab_a), _f64(tab_w)]
        pr = LcfProblem()
        pr.abi_version = LCF_ABI_VERSION
        pr.model = int(model_id)
        pr.n_par = int(n_par)
        pr.use_sigma = int(bool(use_sigma))
        pr.sigma_type = int(sigma_type)
        pr.n_filters = len(keep[4]) - 1
        pr.n_points = len(keep[0])
        if not (len(keep[1]) == len(keep[2]) == len(keep[3]) == pr.n_points):
            raise ValueError('t, y, dy, filt_idx must have the same length')
        cs = list(consts) + [0.] * (N_CONSTS - len(consts))
        pr.consts = (C.c_double * N_CONSTS)(*cs)
        pr.t, pr.y, pr.dy = _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2])
        pr.filt_idx, pr.tab_off = _ptr(keep[3], _ip), _ptr(keep[4], _ip)
        pr.tab_a, pr.tab_w = _ptr(keep[5]), _ptr(keep[6])
        if tab_ext is not None:  # A_lambda / E(B-V) per table sample (ShockCooling3)
            ext = _f64(tab_ext)
            if len(ext) != len(keep[5]):
                raise ValueError('tab_ext must have one entry per table sample')
            keep.append(ext)
            pr.tab_ext = _ptr(ext)
        if ctab is not None:  # (coff, ca, cw, ctmin): Gauss-compressed companions of the band tables
            cx = [_i32(ctab[0]), _f64(ctab[1]), _f64(ctab[2]), _f64(ctab[3])]
            if len(cx[0]) != pr.n_filters + 1 or len(cx[3]) != pr.n_filters or len(cx[1]) != len(cx[2]):
                raise ValueError('inconsistent compressed tables')
            keep += cx
            pr.ctab_off, pr.ctab_a, pr.ctab_w, pr.ctab_tmin = _ptr(cx[0], _ip), _ptr(cx[1]), _ptr(cx[2]), _ptr(cx[3])
        if htab is not None:  # (hoff, ha, hw, htmin): the shorter "hot" level (needs ctab)
            hx = [_i32(htab[0]), _f64(htab[1]), _f64(htab[2]), _f64(htab[3])]
            if len(hx[0]) != pr.n_filters + 1 or len(hx[3]) != pr.n_filters or len(hx[1]) != len(hx[2]):
                raise ValueError('inconsistent hot-level tables')
            keep += hx
            pr.htab_off, pr.htab_a, pr.htab_w, pr.htab_tmin = _ptr(hx[0], _ip), _ptr(hx[1]), _ptr(hx[2]), _ptr(hx[3])
        if itab is not None:  # (coef[n_filters, m, 8], tmin[n_filters], u0, h): interpolants of ln S(ln T)
            ic, it = _f64(itab[0]), _f64(itab[1])
            if ic.ndim != 3 or ic.shape[0] != pr.n_filters or ic.shape[2] != 8 or it.shape != (pr.n_filters,):
                raise ValueError('inconsistent interpolant tables')
            keep += [ic, it]
            pr.itab_coef, pr.itab_tmin, pr.itab_m = _ptr(ic), _ptr(it), ic.shape[1]
            pr.itab_u0, pr.itab_h = float(itab[2]), float(itab[3])
        if companion is not None:
            kp, sp, dtp, knots, coef = companion
            extra = [_i32(kp), _i32(sp), _i32(dtp), _f64(knots), _f64(coef)]
            if extra[4].shape != (pr.n_filters, len(extra[3]) - 1, 4):
                raise ValueError('spline_coef must have shape (n_filters, n_knots - 1, 4)')
            keep += extra
            pr.filt_kasen_par, pr.filt_sifto_par, pr.filt_dt_par = (_ptr(x, _ip) for x in extra[:3])
            pr.n_knots = len(extra[3])
            pr.spline_knots, pr.spline_coef = _ptr(extra[3]), _ptr(extra[4])
        n_dim = pr.n_par + pr.use_sigma
        if priors is not None:
            if len(priors) != n_dim:
                raise ValueError(f'priors must have length {n_dim}')
            arr = (LcfPrior * n_dim)()
            for i, (kind, lo, hi, mean, std) in enumerate(priors):
                arr[i].kind, arr[i].p_min, arr[i].p_max, arr[i].mean, arr[i].stddev = int(kind), lo, hi, mean, std
            keep.append(arr)
            pr.priors = arr
        _check(lib.lcf_engine_create(C.byref(pr), int(device), C.byref(self._h)))
        self.ndim = n_dim
        self.npoints = pr.n_points
        self.device = int(device)
        self.samples_per_eval = lib.lcf_engine_samples_per_eval(self._h)

    def close(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self._h = None   # (no module global is touched here: this also runs at interpreter shutdown)
            self._lib.lcf_engine_destroy(h)

    __del__ = close

    @property
    def handle(self):
        return self._h

    def set_variant(self, variant):
        _check(self._lib.lcf_engine_set_variant(self._h, int(variant)))

    def _block(self, P):
        P = _f64(P)
        if P.ndim == 1:
            P = P[None, :]
        if P.ndim != 2 or P.shape[1] != self.ndim:
            raise ValueError(f'parameter block must have shape (n, {self.ndim}), got {P.shape}')
        return P

    def log_likelihood(self, P):
        P = self._block(P)
        out = np.empty(len(P))
        _check(self._lib.lcf_log_likelihood(self._h, len(P), _ptr(P), _ptr(out)))
        return out

    def log_posterior(self, P):
        P = self._block(P)
        out = np.empty(len(P))
        _check(self._lib.lcf_log_posterior(self._h, len(P), _ptr(P), _ptr(out)))
        return out

    def log_likelihood_dev(self, n, dP, dout, stream=0, posterior=False):
        """Device pointers (ints), enqueue only."""
        fn = self._lib.lcf_log_posterior_dev if posterior else self._lib.lcf_log_likelihood_dev
        _check(fn(self._h, int(n), C.c_void_p(dP), C.c_void_p(dout), C.c_void_p(stream)))

    def evaluate(self, P):
        P = self._block(P)
        out = np.empty((len(P), self.npoints))
        _check(self._lib.lcf_model_evaluate(self._h, len(P), _ptr(P), _ptr(out)))
        return out

    def temperature_radius(self, P):
        P = self._block(P)
        T = np.empty((len(P), self.npoints))
        R = np.empty((len(P), self.npoints))
        _check(self._lib.lcf_temperature_radius(self._h, len(P), _ptr(P), _ptr(T), _ptr(R)))
        return T, R

    def profile_loglike_kernel(self, P, reps=20):
        """Average duration [ms] of the per-point likelihood kernel alone (HIP events, engine stream)."""
        P = self._block(P)
        ms = C.c_double()
        _check(self._lib.lcf_profile_loglike_kernel(self._h, len(P), _ptr(P), int(reps), C.byref(ms)))
        return ms.value

    def blackbody_to_filters(self, filt_idx, T, R):
        f, T, R = _i32(filt_idx), _f64(T), _f64(R)
        if not (f.shape == T.shape == R.shape and f.ndim == 1):
            raise ValueError('filt_idx, T and R must be 1-D and of equal length')
        out = np.empty(len(f))
        _check(self._lib.lcf_blackbody_to_filters(self._h, len(f), _ptr(f, _ip), _ptr(T), _ptr(R), _ptr(out)))
        return out


#: `lcf_sampler_last_run_kernel` values (include/lcf.h: LCF_KERNEL_*) -> the names `last_run_kernel()` returns
KERNEL_NAMES = {0: 'phases', 1: 'fused', 2: 'solo', 3: 'population', 4: 'population-phases', 5: 'run', 6: 'population-run'}


class NativeSampler:
    """Thin handle on ``lcf_sampler`` (device-resident stretch move)."""

    def __init__(self, engine, nwalkers, seed=0, a=2.0):
        self._lib = engine._lib
        self.engine = engine
        self.nwalkers = int(nwalkers)
        self.ndim = engine.ndim
        self._h = C.c_void_p()
        _check(self._lib.lcf_sampler_create(engine.handle, self.nwalkers, C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                            float(a), C.byref(self._h)))

    def close(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self._h = None   # (no module global is touched here: this also runs at interpreter shutdown)
            self._lib.lcf_sampler_destroy(h)

    __del__ = close

    def set_state(self, coords):
        coords = _f64(coords)
        if coords.shape != (self.nwalkers, self.ndim):
            raise ValueError(f'coords must have shape ({self.nwalkers}, {self.ndim})')
        _check(self._lib.lcf_sampler_set_state(self._h, _ptr(coords)))

    def get_state(self):
        x = np.empty((self.nwalkers, self.ndim))
        lp = np.empty(self.nwalkers)
        _check(self._lib.lcf_sampler_get_state(self._h, _ptr(x), _ptr(lp)))
        return x, lp

    @staticmethod
    def _split(split, nsteps, nwalkers):
        """``split``: 'random' (device-generated, emcee's randomize_split), 'identity', or an int32 array
        (nsteps, nwalkers) of host-provided permutations."""
        if isinstance(split, str):
            return {'identity': SPLIT_IDENTITY, 'random': SPLIT_RANDOM}[split], None, None
        perm = _i32(split)
        if perm.shape != (nsteps, nwalkers):
            raise ValueError(f'perm must have shape ({nsteps}, {nwalkers})')
        return SPLIT_HOST, perm, _ptr(perm, _ip)

    def run(self, first_step, nsteps, split='random', store=True):
        mode, keep, pp = self._split(split, nsteps, self.nwalkers)
        _check(self._lib.lcf_sampler_run(self._h, int(first_step), int(nsteps), mode, pp, int(bool(store))))
        self._last = (int(nsteps), bool(store))

    def run_async(self, first_step, nsteps, split='random', store=True):
        """Enqueue the run and return; :meth:`wait` completes it (population mode: many samplers in flight)."""
        mode, keep, pp = self._split(split, nsteps, self.nwalkers)
        _check(self._lib.lcf_sampler_run_async(self._h, int(first_step), int(nsteps), mode, pp, int(bool(store))))
        self._last = (int(nsteps), bool(store))

    def wait(self):
        _check(self._lib.lcf_sampler_wait(self._h))

    def run_sharded(self, comm, first_step, nsteps, split='random', store=True):
        """Collective: the whole run natively over ``comm`` (a :class:`NativeComm`)."""
        mode, keep, pp = self._split(split, nsteps, self.nwalkers)
        _check(self._lib.lcf_sampler_run_sharded(self._h, comm._h, int(first_step), int(nsteps), mode, pp,
                                                 int(bool(store))))
        self._last = (int(nsteps), bool(store))

    def mailbox_export(self):
        """(64-byte IPC handle, local device pointer) of this rank's peer mailbox."""
        h = C.create_string_buffer(64)
        ptr = C.c_void_p()
        _check(self._lib.lcf_sampler_mailbox_export(self._h, h, C.byref(ptr)))
        return h.raw, ptr.value

    def mailbox_connect(self, n_ranks, rank, handles=None, local_ptrs=None):
        """Map every rank's mailbox: ``handles`` = the exported IPC handles of all ranks (other processes), or
        ``local_ptrs`` = their device pointers (ranks emulated inside this process)."""
        hb = C.create_string_buffer(b''.join(handles), 64 * n_ranks) if handles is not None else None
        lp = (C.c_void_p * n_ranks)(*local_ptrs) if local_ptrs is not None else None
        _check(self._lib.lcf_sampler_mailbox_connect(self._h, int(n_ranks), int(rank), hb, lp))

    def board_export(self):
        """(64-byte IPC handle, local device pointer) of this rank's row board (see ``lcf_sampler_run_rows``)."""
        h = C.create_string_buffer(64)
        ptr = C.c_void_p()
        _check(self._lib.lcf_sampler_board_export(self._h, h, C.byref(ptr)))
        return h.raw, ptr.value

    def board_connect(self, n_ranks, rank, handles=None, local_ptrs=None):
        """Map every rank's row board: IPC handles of all ranks, or device pointers of ranks emulated in this process."""
        hb = C.create_string_buffer(b''.join(handles), 64 * n_ranks) if handles is not None else None
        lp = (C.c_void_p * n_ranks)(*local_ptrs) if local_ptrs is not None else None
        _check(self._lib.lcf_sampler_board_connect(self._h, int(n_ranks), int(rank), hb, lp))

    def run_rows(self, first_step, nsteps, split='random', store=True, asynchronous=False):
        """Collective in effect: the sharded run in which nothing is replicated -- every rank moves its share of the
        walkers with the one-workgroup-per-proposal kernel and posts their rows on all ranks' boards (see
        ``lcf_sampler_run_rows``); ``asynchronous``: enqueue only, :meth:`wait` completes it."""
        mode, keep, pp = self._split(split, nsteps, self.nwalkers)
        fn = self._lib.lcf_sampler_run_rows_async if asynchronous else self._lib.lcf_sampler_run_rows
        _check(fn(self._h, int(first_step), int(nsteps), mode, pp, int(bool(store))))
        self._last = (int(nsteps), bool(store))

    def run_peers(self, first_step, nsteps, split='random', store=True, asynchronous=False):
        """Collective in effect: the sharded run over peer mailboxes (see ``lcf_sampler_run_peers``);
        ``asynchronous``: enqueue only, :meth:`wait` completes it."""
        mode, keep, pp = self._split(split, nsteps, self.nwalkers)
        fn = self._lib.lcf_sampler_run_peers_async if asynchronous else self._lib.lcf_sampler_run_peers
        _check(fn(self._h, int(first_step), int(nsteps), mode, pp, int(bool(store))))
        self._last = (int(nsteps), bool(store))

    def begin(self, first_step, nsteps, split='random', store=True):
        mode, keep, pp = self._split(split, nsteps, self.nwalkers)
        _check(self._lib.lcf_sampler_begin(self._h, int(first_step), int(nsteps), mode, pp, int(bool(store))))
        self._last = (int(nsteps), bool(store))

    def propose(self, step, half, stream=0):
        _check(self._lib.lcf_sampler_propose(self._h, int(step), int(half), C.c_void_p(stream)))

    def evaluate(self, lo, hi, stream=0):
        _check(self._lib.lcf_sampler_evaluate(self._h, int(lo), int(hi), C.c_void_p(stream)))

    def accept(self, step, half, stream=0):
        _check(self._lib.lcf_sampler_accept(self._h, int(step), int(half), C.c_void_p(stream)))

    def half_step(self, step, half, lo, hi, stream=0):
        _check(self._lib.lcf_sampler_half_step(self._h, int(step), int(half), int(lo), int(hi), C.c_void_p(stream)))

    def newlp_ptr(self):
        return self._lib.lcf_sampler_newlp_ptr(self._h)

    @property
    def one_launch(self):
        """True if a half-step of this sampler is a single kernel launch (see ``lcf_sampler_one_launch``)."""
        return bool(self._lib.lcf_sampler_one_launch(self._h))

    def last_run_kernel(self):
        """What executed the half-steps of the last run: 'phases' | 'fused' | 'solo' | 'run' (k_solo_run: resident
        workgroups, one launch per block of half-steps) | 'population' (one launch per half-step for all transients of
        a population) | 'population-run' (k_pop_run: resident workgroups for all transients) | 'population-phases'
        (None: no run yet)."""
        return KERNEL_NAMES.get(self._lib.lcf_sampler_last_run_kernel(self._h))

    def last_run_launches(self):
        """Launches of the half-step kernel in the last single-GPU run (two per step; 'run': one per block of steps)."""
        return int(self._lib.lcf_sampler_last_run_launches(self._h))

    def set_half_step_kernel(self, choice='auto'):
        """Restrict the kernels a single-GPU run uses for a half-step ('auto' | 'solo' | 'fused' | 'phases'; same
        chain bit for bit).  Returns what a run uses now: 'run' (one workgroup per proposal, accept test included,
        resident for a block of half-steps), 'solo' (the same with a launch per half-step), 'fused' (one workgroup per
        proposal and part) or 'phases' (proposal + likelihood launches)."""
        used = C.c_int32()
        _check(self._lib.lcf_sampler_set_half_step_kernel(self._h, {'auto': 0, 'fused': 1, 'phases': 2, 'solo': 3}[choice],
                                                          C.byref(used)))
        return {3: 'run', 2: 'solo', 1: 'fused', 0: 'phases'}[used.value]

    def half_step_rows(self, step, half, lo, hi, stream=0):
        _check(self._lib.lcf_sampler_half_step_rows(self._h, int(step), int(half), int(lo), int(hi),
                                                    C.c_void_p(stream)))

    def rows_ptr(self):
        """(device pointer, doubles per row) of the per-proposal rows of the half-step drawn last."""
        n = C.c_int32()
        return self._lib.lcf_sampler_rows_ptr(self._h, C.byref(n)), int(n.value)

    def check(self):
        _check(self._lib.lcf_sampler_check(self._h))

    def reserve_chain(self, n_steps):
        """Allocate the device memory a stored run of n_steps steps needs now, instead of inside that run."""
        _check(self._lib.lcf_sampler_reserve_chain(self._h, int(n_steps)))

    def get_chain(self):
        nsteps, store = self._last
        chain = np.empty((nsteps, self.nwalkers, self.ndim))
        lp = np.empty((nsteps, self.nwalkers))
        _check(self._lib.lcf_sampler_get_chain(self._h, _ptr(chain), _ptr(lp)))
        return chain, lp

    def snapshot(self):
        """``(coords, log_prob, n_accepted)`` of the sampler's present state in one native call."""
        x = np.empty((self.nwalkers, self.ndim))
        lp = np.empty(self.nwalkers)
        acc = np.empty(self.nwalkers, dtype=np.int64)
        _check(self._lib.lcf_sampler_get_snapshot(self._h, x.ctypes.data, lp.ctypes.data, acc.ctypes.data))
        return x, lp, acc

    def naccepted(self):
        out = np.empty(self.nwalkers, dtype=np.int64)
        _check(self._lib.lcf_sampler_get_naccepted(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def last_run_ms(self):
        return float(self._lib.lcf_sampler_last_run_ms(self._h))


def population_run(native_samplers, first_step, nsteps, split='random', store=True):
    """One batched native run over several :class:`NativeSampler` objects (population mode).  Returns device ms."""
    lib = load_library()
    n = len(native_samplers)
    arr = (C.c_void_p * n)(*[s._h for s in native_samplers])
    mode = {'identity': SPLIT_IDENTITY, 'random': SPLIT_RANDOM}[split]
    ms = C.c_double()
    _check(lib.lcf_population_run(arr, n, int(first_step), int(nsteps), mode, int(bool(store)), C.byref(ms)))
    for s in native_samplers:
        s._last = (int(nsteps), bool(store))
    return ms.value


def rccl_library_path():
    """The librccl.so PyTorch ships (so that one RCCL serves torch.distributed and the native loop), or ''."""
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        return cand if os.path.exists(cand) else ''
    except ImportError:
        return ''


class NativeComm:
    """An RCCL communicator owned by the native library, bootstrapped over an initialised torch.distributed group
    (the 128-byte unique id is broadcast from rank 0)."""

    @staticmethod
    def probe():
        """True if RCCL can be bound in this process (local check, no communication)."""
        return load_library().lcf_comm_probe(rccl_library_path().encode()) == 0

    def __init__(self, device, group=None):
        import torch.distributed as dist
        self._lib = load_library()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        path = rccl_library_path().encode()
        uid = C.create_string_buffer(128)
        if self.rank == 0:
            _check(self._lib.lcf_comm_unique_id(path, uid))
        box = [uid.raw]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        uid = C.create_string_buffer(box[0], 128)
        self._h = C.c_void_p()
        _check(self._lib.lcf_comm_create(path, uid, self.world, self.rank, int(device), C.byref(self._h)))

    def count(self):
        """(number of ranks, this rank) as RCCL's communicator reports them."""
        n, r = C.c_int32(), C.c_int32()
        _check(self._lib.lcf_comm_count(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def time_allgather(self, native_sampler, reps=200):
        """Average ms of one per-half-step all-gather of ``native_sampler``'s rows (collective call)."""
        ms = C.c_double()
        _check(self._lib.lcf_comm_time_allgather(self._h, native_sampler._h, int(reps), C.byref(ms)))
        return ms.value

    def close(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self._h = None   # (no module global is touched here: this also runs at interpreter shutdown)
            self._lib.lcf_comm_destroy(h)

    __del__ = close


class SedEngine:
    """Per-epoch blackbody SED likelihood on the device (``lcf_sed_*``): band tables are fixed at creation,
    observations are set per batch of epochs, candidates are evaluated per call."""

    def __init__(self, tab_off, tab_a, tab_w, device=0, ctab=None, itab=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        off, a, w = _i32(tab_off), _f64(tab_a), _f64(tab_w)
        if ctab is None:
            cargs = (None, None, None, None)
        else:
            cx = [_i32(ctab[0]), _f64(ctab[1]), _f64(ctab[2]), _f64(ctab[3])]
            cargs = (_ptr(cx[0], _ip), _ptr(cx[1]), _ptr(cx[2]), _ptr(cx[3]))
        if itab is None:   # (coef[n_filters, m, 8], tmin[n_filters], u0, h): interpolants of ln S(ln T)
            iargs = (None, None, 0, 0., 0.)
        else:
            ic, it = _f64(itab[0]), _f64(itab[1])
            if ic.ndim != 3 or ic.shape[0] != len(off) - 1 or ic.shape[2] != 8 or it.shape != (len(off) - 1,):
                raise ValueError('inconsistent interpolants')
            iargs = (_ptr(ic), _ptr(it), ic.shape[1], float(itab[2]), float(itab[3]))
        self.has_interpolants = itab is not None
        _check(self._lib.lcf_sed_create(len(off) - 1, _ptr(off, _ip), _ptr(a), _ptr(w), *cargs, *iargs, int(device),
                                        C.byref(self._h)))
        self.n_epochs = 0
        self.last_kernel_ms = 0.

    def close(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self._h = None   # (no module global is touched here: this also runs at interpreter shutdown)
            self._lib.lcf_sed_destroy(h)

    __del__ = close

    def set_observations(self, ep_off, filt_idx, y, dy):
        off, f, y, dy = _i32(ep_off), _i32(filt_idx), _f64(y), _f64(dy)
        if not (len(f) == len(y) == len(dy) == off[-1]):
            raise ValueError('filt_idx, y, dy must have ep_off[-1] entries')
        _check(self._lib.lcf_sed_set_observations(self._h, len(off) - 1, _ptr(off, _ip), _ptr(f, _ip), _ptr(y), _ptr(dy)))
        self.n_epochs = len(off) - 1

    def log_likelihood(self, cand, sigma_type=SIGMA_RELATIVE, precision=0, compressed=True):
        """``cand``: (n_epochs, n_cand, 2 or 3) -> (n_epochs, n_cand)."""
        cand = _f64(cand)
        if cand.ndim != 3 or cand.shape[0] != self.n_epochs or cand.shape[2] not in (2, 3):
            raise ValueError(f'candidates must have shape ({self.n_epochs}, n_cand, 2|3), got {cand.shape}')
        out = np.empty(cand.shape[:2])
        ms = C.c_double()
        _check(self._lib.lcf_sed_log_likelihood(self._h, cand.shape[1], cand.shape[2], int(sigma_type), _ptr(cand),
                                                int(precision), int(bool(compressed)), _ptr(out), C.byref(ms)))
        self.last_kernel_ms = ms.value
        return out
