"""numpy / device-pointer wrappers of the libccgp C ABI (include/ccgp.h).

Every function here is a 1:1 call into the shared library; nothing is computed in
Python.  Matrices cross the boundary column-major (R's layout): inputs are copied with
`np.asfortranarray`, outputs are allocated Fortran-ordered.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_double, c_int, c_size_t, c_void_p

import numpy as np

from ._lib import load_library

MEAN_PROFILE_BETA = 0
MEAN_ZERO_PLUS_TAU2 = 1
PRIOR_INVGAMMA, PRIOR_GV, PRIOR_ISO, PRIOR_ANI = 0, 1, 2, 3
T_COV, T_UPDATE, T_DIAG, T_TRSM, T_SOLVE, T_FUSED = range(6)
KERNEL_GAUSS, KERNEL_MATERN, KERNEL_MATERN_SPLINE = 0, 1, 2
OPT_FUSE_DIAG, OPT_TAIL_STRIPS, OPT_WIDE_OFFSETS, OPT_SMALL_GRID16 = 2, 3, 4, 5
OPT_SCHED, OPT_SCHED_POLICY, OPT_PREDICT_FACTOR = 7, 8, 9
TIMING_NAMES = ("cov", "update", "diag", "trsm", "solve", "fused", "sweep")

_dp = POINTER(c_double)
_ip = POINTER(c_int)

# name -> (restype, argtypes); also the list tests check against include/ccgp.h
SIGNATURES = {
    "ccgp_create": (c_int, [c_int, POINTER(c_void_p)]),
    "ccgp_destroy": (c_int, [c_void_p]),
    "ccgp_last_error": (c_char_p, [c_void_p]),
    "ccgp_version": (c_char_p, []),
    "ccgp_set_stream": (c_int, [c_void_p, c_void_p]),
    "ccgp_set_kernel": (c_int, [c_void_p, c_int, c_double]),
    "ccgp_set_workspace_limit": (c_int, [c_void_p, c_size_t]),
    "ccgp_set_option": (c_int, [c_void_p, c_int, c_int]),
    "ccgp_reserve": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "ccgp_synchronize": (c_int, [c_void_p]),
    "ccgp_corr_matrix": (c_int, [c_void_p, _dp, c_int, c_int, _dp, _dp]),
    "ccgp_corr_cross": (c_int, [c_void_p, _dp, c_int, _dp, c_int, c_int, _dp, _dp]),
    "ccgp_mixed_corr_matrix": (c_int, [c_void_p, _dp, c_int, c_int, c_int, _dp, _dp]),
    "ccgp_mixed_corr_cross": (c_int, [c_void_p, _dp, c_int, _dp, c_int, c_int, c_int, _dp, _dp]),
    "ccgp_beta_mle": (c_int, [c_void_p, _dp, _dp, c_int, _dp]),
    "ccgp_sigma2_mle": (c_int, [c_void_p, _dp, _dp, c_int, c_double, _dp]),
    "ccgp_loglik_batch": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_int, _dp, c_int, c_double,
                                  c_int, c_double, _dp, _dp, _ip]),
    "ccgp_loglik_batch_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                      c_int, c_double, c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "ccgp_loglik_grad_batch": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_int, _dp, c_int, c_double,
                                       _dp, _dp, _dp, _ip]),
    "ccgp_logpost": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_double, c_int, _dp, _dp, _dp, _dp,
                             _dp, _dp, _ip]),
    "ccgp_logpost_batch": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_double, c_int, _dp, c_int, _dp, _dp, _dp, _dp, _ip]),
    "ccgp_grid_marginal": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_double, _dp, c_int, c_int,
                                   c_double, c_int, c_double, _dp, _ip, _dp]),
    "ccgp_mixed_logdet_designs": (c_int, [c_void_p, _dp, c_int, c_int, c_int, c_int, _dp, _dp, _ip]),
    "ccgp_halton_base2": (c_int, [c_int, _dp]),
    "ccgp_qigamma": (c_int, [_dp, c_int, c_double, c_double, _dp]),
    "ccgp_predict_batch": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_int, _dp, c_int, _dp, c_int,
                                   c_double, _dp, _dp, _dp, _ip]),
    "ccgp_predict_batch_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                       c_int, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "ccgp_factor_batch": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_int, _dp, c_int, c_double, POINTER(c_void_p),
                                  _dp, _dp, _ip]),
    "ccgp_predict_from_factorset": (c_int, [c_void_p, c_void_p, _dp, c_int, _dp, _dp]),
    "ccgp_factorset_bytes": (c_size_t, [c_void_p]),
    "ccgp_factorset_free": (c_int, [c_void_p, c_void_p]),
    "ccgp_factors": (c_int, [c_void_p, _dp, c_double, _dp, c_int, _dp]),
    "ccgp_predict_from_factors": (c_int, [c_void_p, _dp, c_int, c_int, c_double, _dp, _dp, c_double,
                                          _dp, c_double, _dp, _dp]),
    "ccgp_predict_post": (c_int, [c_void_p, _dp, c_int, _dp, c_int, c_int, c_int, _dp, c_double, _dp, _dp, c_double,
                                  _dp, c_double, _dp, _dp]),
    "ccgp_multi_create": (c_int, [c_int, _ip, POINTER(c_void_p)]),
    "ccgp_multi_destroy": (c_int, [c_void_p]),
    "ccgp_multi_count": (c_int, [c_void_p]),
    "ccgp_multi_handle": (c_void_p, [c_void_p, c_int]),
    "ccgp_multi_last_error": (c_char_p, [c_void_p]),
    "ccgp_multi_set_kernel": (c_int, [c_void_p, c_int, c_double]),
    "ccgp_multi_loglik_batch": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_int, _dp, c_int, c_double,
                                        c_int, c_double, _dp, _dp, _ip]),
    "ccgp_multi_grid_marginal": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_double, _dp, c_int, c_int,
                                         c_double, c_int, c_double, _dp, _ip, _dp]),
    "ccgp_multi_predict_batch": (c_int, [c_void_p, _dp, c_int, c_int, _dp, c_int, _dp, c_int, _dp, c_int,
                                         c_double, _dp, _dp, _dp, _ip]),
    "ccgp_enable_timing": (c_int, [c_void_p, c_int]),
    "ccgp_get_timing": (c_int, [c_void_p, c_int, _dp, _ip]),
    "ccgp_last_sched_profile": (c_int, [c_void_p, c_void_p, c_int, _ip]),
}

_bound = None


def lib():
    global _bound
    if _bound is None:
        L = load_library()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = L
    return _bound


class CcgpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libccgp error %d: %s" % (code, msg))
        self.code = code


def _f(a, shape=None):
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _ipt(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def halton_base2(N):
    out = np.empty(int(N), dtype=np.float64)
    rc = lib().ccgp_halton_base2(int(N), _p(out))
    if rc:
        raise CcgpError(rc, "ccgp_halton_base2")
    return out


def qigamma(p, alpha, beta):
    p = np.ascontiguousarray(np.atleast_1d(np.asarray(p, dtype=np.float64)))
    out = np.empty_like(p)
    rc = lib().ccgp_qigamma(_p(p), p.size, float(alpha), float(beta), _p(out))
    if rc:
        raise CcgpError(rc, "ccgp_qigamma")
    return out


class FactorSet:
    """S Cholesky factors kept in HBM (ccgp_factor_batch); predict(Xtest) -> (mean[S, m], var[S, m])."""

    def __init__(self, handle, ptr, S, loglik, beta, status):
        self._handle, self._fs, self.S = handle, ptr, S
        self.loglik, self.beta, self.status = loglik, beta, status

    @property
    def nbytes(self):
        return int(lib().ccgp_factorset_bytes(self._fs))

    def predict(self, Xtest):
        Xtest = _f(np.atleast_2d(Xtest))
        m = Xtest.shape[0]
        mean = np.empty((self.S, m), dtype=np.float64, order="F")
        var = np.empty((self.S, m), dtype=np.float64, order="F")
        self._handle._chk(lib().ccgp_predict_from_factorset(self._handle._h, self._fs, _p(Xtest), m, _p(mean), _p(var)))
        return mean, var

    def free(self):
        if self._fs:
            lib().ccgp_factorset_free(self._handle._h, self._fs)
            self._fs = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MultiHandle:
    """Several devices behind one host process (ccgp_multi_*): the batched calls are cut into contiguous shards,
    one per listed device, and gathered in host memory.  devices: a count (0 .. k-1) or an explicit list; a device
    may be listed more than once (the shards then share it)."""

    def __init__(self, devices):
        devs = list(range(devices)) if isinstance(devices, int) else [int(v) for v in devices]
        arr = (c_int * len(devs))(*devs)
        self._m = c_void_p()
        rc = lib().ccgp_multi_create(len(devs), arr, ctypes.byref(self._m))
        if rc:
            self._m = None
            raise CcgpError(rc, "ccgp_multi_create(%s) failed -- is every device visible?" % devs)
        self.devices = devs

    def close(self):
        if getattr(self, "_m", None):
            lib().ccgp_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise CcgpError(rc, lib().ccgp_multi_last_error(self._m).decode())
        return rc

    def count(self):
        return lib().ccgp_multi_count(self._m)

    def set_kernel(self, family=0, nu=0.0):
        self._chk(lib().ccgp_multi_set_kernel(self._m, int(family), float(nu)))

    def set_workspace_limit(self, nbytes):
        for i in range(self.count()):
            rc = lib().ccgp_set_workspace_limit(c_void_p(lib().ccgp_multi_handle(self._m, i)), int(nbytes))
            if rc:
                raise CcgpError(rc, "ccgp_set_workspace_limit")

    def loglik_batch(self, X, y, K, params, sigma2, mean_mode=MEAN_PROFILE_BETA, tau2=0.0):
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        params = _f(np.atleast_2d(params))
        B, P = params.shape
        if P != K + K * d:
            raise ValueError("params must have K + K*d = %d columns" % (K + K * d))
        ll, beta = np.empty(B), np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        self._chk(lib().ccgp_multi_loglik_batch(self._m, _p(X), n, d, _p(y), K, _p(params), B, float(sigma2),
                                                int(mean_mode), float(tau2), _p(ll), _p(beta), _ipt(st)))
        return ll, beta, st

    def grid_marginal(self, X, y, sigma2, hyper, N, tau, take_log, aniso_lambda=-1.0, want_logs=False):
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        hyper = _f(hyper)
        G = hyper.shape[0]
        out = np.empty(G)
        arg = c_int()
        logs = np.empty((G, N), dtype=np.float64) if want_logs else None
        self._chk(lib().ccgp_multi_grid_marginal(self._m, _p(X), n, d, _p(y), float(sigma2), _p(hyper), G, int(N),
                                                 float(tau), 1 if take_log else 0, float(aniso_lambda), _p(out),
                                                 ctypes.byref(arg), _p(logs)))
        return (out, arg.value, logs) if want_logs else (out, arg.value)

    def predict_batch(self, X, y, K, params, Xtest, sigma2):
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        params = _f(np.atleast_2d(params))
        S = params.shape[0]
        Xtest = _f(np.atleast_2d(Xtest))
        m = Xtest.shape[0]
        mean = np.empty((S, m), dtype=np.float64, order="F")
        var = np.empty((S, m), dtype=np.float64, order="F")
        beta = np.empty(S)
        st = np.zeros(S, dtype=np.int32)
        self._chk(lib().ccgp_multi_predict_batch(self._m, _p(X), n, d, _p(y), K, _p(params), S, _p(Xtest), m,
                                                 float(sigma2), _p(mean), _p(var), _p(beta), _ipt(st)))
        return mean, var, beta, st


class Handle:
    """One libccgp handle bound to one HIP device (one per process / per GPU)."""

    def __init__(self, device=0):
        self._h = c_void_p()
        rc = lib().ccgp_create(int(device), ctypes.byref(self._h))
        if rc:
            self._h = None
            raise CcgpError(rc, "ccgp_create(device=%d) failed -- is a HIP device visible?" % device)
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            lib().ccgp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise CcgpError(rc, lib().ccgp_last_error(self._h).decode())
        return rc

    # -- plumbing ----------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr):
        self._chk(lib().ccgp_set_stream(self._h, c_void_p(hip_stream_ptr or 0)))

    def set_kernel(self, family=0, nu=0.0):
        """Correlation family for the calls that follow: KERNEL_GAUSS (default) or KERNEL_MATERN with
        smoothness nu (the 1-D scripts, D1:348-351; d must then be 1)."""
        self._chk(lib().ccgp_set_kernel(self._h, int(family), float(nu)))

    def set_workspace_limit(self, nbytes):
        self._chk(lib().ccgp_set_workspace_limit(self._h, int(nbytes)))

    def set_option(self, option, value):
        """Measurement switches of include/ccgp.h (OPT_*)."""
        self._chk(lib().ccgp_set_option(self._h, int(option), int(value)))

    def reserve(self, n, d, K, B, m=0):
        self._chk(lib().ccgp_reserve(self._h, n, d, K, B, m))

    def synchronize(self):
        self._chk(lib().ccgp_synchronize(self._h))

    def enable_timing(self, on=True, only=None):
        """Bracket launch groups with HIP events on their stream.  only = names from TIMING_NAMES to
        restrict the events to (each timed launch costs two event records)."""
        mask = 1 if on else 0
        if on and only is not None:
            mask = 0
            for name in only:
                mask |= 1 << (1 + TIMING_NAMES.index(name))
        self._chk(lib().ccgp_enable_timing(self._h, mask))

    def get_timing(self):
        out = {}
        for i, name in enumerate(TIMING_NAMES):
            ms, cnt = c_double(), c_int()
            self._chk(lib().ccgp_get_timing(self._h, i, ctypes.byref(ms), ctypes.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    def last_sched_profile(self):
        """Per-workgroup time account of the last scheduled sweep (OPT_SCHED_POLICY bit 2): array (workgroups, 8) --
        microseconds waiting for a task, in D / U / T tiles, applying arrivals; tasks run; XCD served; second-on-its-CU."""
        buf = np.zeros((1024, 8), dtype=np.uint64)
        nw = c_int()
        self._chk(lib().ccgp_last_sched_profile(self._h, buf.ctypes.data_as(c_void_p), 1024, ctypes.byref(nw)))
        out = buf[:nw.value].astype(np.float64)
        out[:, :5] *= 0.01
        return out

    def corr_matrix(self, X, theta):
        X = _f(X)
        n, d = X.shape
        theta = _f(np.broadcast_to(np.asarray(theta, dtype=np.float64), (d,)))
        out = np.empty((n, n), dtype=np.float64, order="F")
        self._chk(lib().ccgp_corr_matrix(self._h, _p(X), n, d, _p(theta), _p(out)))
        return out

    def corr_cross(self, Xnew, X, theta):
        X = _f(X)
        n, d = X.shape
        Xnew = _f(np.atleast_2d(Xnew))
        m = Xnew.shape[0]
        theta = _f(np.broadcast_to(np.asarray(theta, dtype=np.float64), (d,)))
        out = np.empty((m, n), dtype=np.float64, order="F")
        self._chk(lib().ccgp_corr_cross(self._h, _p(Xnew), m, _p(X), n, d, _p(theta), _p(out)))
        return out

    def mixed_corr_matrix(self, X, K, params_row):
        X = _f(X)
        n, d = X.shape
        row = _f(params_row, (K + K * d,))
        out = np.empty((n, n), dtype=np.float64, order="F")
        self._chk(lib().ccgp_mixed_corr_matrix(self._h, _p(X), n, d, K, _p(row), _p(out)))
        return out

    def mixed_corr_cross(self, Xnew, X, K, params_row):
        X = _f(X)
        n, d = X.shape
        Xnew = _f(np.atleast_2d(Xnew))
        m = Xnew.shape[0]
        row = _f(params_row, (K + K * d,))
        out = np.empty((m, n), dtype=np.float64, order="F")
        self._chk(lib().ccgp_mixed_corr_cross(self._h, _p(Xnew), m, _p(X), n, d, K, _p(row), _p(out)))
        return out

    # -- a6, a7, a10, a11 (explicit R.Inv forms) ---------------------------------------
    def beta_mle(self, R_inv, y):
        R_inv, y = _f(R_inv), _f(np.ravel(y))
        out = c_double()
        self._chk(lib().ccgp_beta_mle(self._h, _p(R_inv), _p(y), y.size, ctypes.byref(out)))
        return out.value

    def sigma2_mle(self, R_inv, y, beta):
        R_inv, y = _f(R_inv), _f(np.ravel(y))
        out = c_double()
        self._chk(lib().ccgp_sigma2_mle(self._h, _p(R_inv), _p(y), y.size, float(beta), ctypes.byref(out)))
        return out.value

    def factors(self, R_inv, beta, y):
        R_inv, y = _f(R_inv), _f(np.ravel(y))
        n = y.size
        out = np.empty(2 * n + 1, dtype=np.float64)
        self._chk(lib().ccgp_factors(self._h, _p(R_inv), float(beta), _p(y), n, _p(out)))
        return out

    def predict_from_factors(self, r, beta, mean_factor, var_factor1, var_factor2, R_inv, sigma2):
        r = _f(np.atleast_2d(r))
        m, n = r.shape
        mf, v1, R_inv = _f(mean_factor, (n,)), _f(var_factor1, (n,)), _f(R_inv, (n, n))
        mean, var = np.empty(m), np.empty(m)
        self._chk(lib().ccgp_predict_from_factors(self._h, _p(r), m, n, float(beta), _p(mf), _p(v1),
                                                  float(var_factor2), _p(R_inv), float(sigma2),
                                                  _p(mean), _p(var)))
        return mean, var

    def predict_post(self, Xnew, X, K, params_row, beta, mean_factor, var_factor1, var_factor2, R_inv, sigma2):
        """Literal predict.post (HX:655-673) in one device round trip: Mixed.corr.vec for the rows of Xnew, then the
        quadratic forms with the caller's cached terms.  Returns (mean[m], var[m])."""
        X = _f(X)
        n, d = X.shape
        Xnew = _f(np.atleast_2d(Xnew))
        m = Xnew.shape[0]
        row = _f(params_row, (K + K * d,))
        mf, v1, R_inv = _f(mean_factor, (n,)), _f(var_factor1, (n,)), _f(R_inv, (n, n))
        mean, var = np.empty(m), np.empty(m)
        self._chk(lib().ccgp_predict_post(self._h, _p(Xnew), m, _p(X), n, d, K, _p(row), float(beta), _p(mf), _p(v1),
                                          float(var_factor2), _p(R_inv), float(sigma2), _p(mean), _p(var)))
        return mean, var

    # -- a8, a9, a12 --------------------------------------------------------------------
    def loglik_batch(self, X, y, K, params, sigma2, mean_mode=MEAN_PROFILE_BETA, tau2=0.0):
        """params: [B, K + K*d].  Returns (loglik[B], beta[B], status[B])."""
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        params = _f(np.atleast_2d(params))
        B, P = params.shape
        if P != K + K * d:
            raise ValueError("params must have K + K*d = %d columns" % (K + K * d))
        ll, beta = np.empty(B), np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        self._chk(lib().ccgp_loglik_batch(self._h, _p(X), n, d, _p(y), K, _p(params), B, float(sigma2),
                                          int(mean_mode), float(tau2), _p(ll), _p(beta), _ipt(st)))
        return ll, beta, st

    def loglik_grad_batch(self, X, y, K, params, sigma2):
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        params = _f(np.atleast_2d(params))
        B, P = params.shape
        ll, beta = np.empty(B), np.empty(B)
        grad = np.empty((B, P), dtype=np.float64, order="F")
        st = np.zeros(B, dtype=np.int32)
        self._chk(lib().ccgp_loglik_grad_batch(self._h, _p(X), n, d, _p(y), K, _p(params), B,
                                               float(sigma2), _p(ll), _p(beta), _p(grad), _ipt(st)))
        return ll, beta, grad, st

    def logpost(self, X, y, sigma2, prior_id, theta_t, prior_pars=None, want_Rinv=True):
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        theta_t = _f(np.ravel(theta_t))
        pp = _f(np.ravel(prior_pars)) if prior_pars is not None else None
        val, beta, ll = c_double(), c_double(), c_double()
        st = c_int()
        Rinv = np.empty((n, n), dtype=np.float64, order="F") if want_Rinv else None
        self._chk(lib().ccgp_logpost(self._h, _p(X), n, d, _p(y), float(sigma2), int(prior_id),
                                     _p(theta_t), _p(pp), ctypes.byref(val), ctypes.byref(beta),
                                     ctypes.byref(ll), _p(Rinv), ctypes.byref(st)))
        return dict(val=val.value, beta=beta.value, loglik=ll.value, R_inv=Rinv, status=st.value)

    def logpost_batch(self, X, y, sigma2, prior_id, theta_t, prior_pars=None):
        """ccgp_logpost_batch: logpost of B transformed parameter vectors (rows of theta_t) in one call ->
        (val, beta, loglik, status), each value as Handle.logpost returns it, bit for bit."""
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        theta_t = _f(np.atleast_2d(theta_t))
        B = theta_t.shape[0]
        pp = _f(np.ravel(prior_pars)) if prior_pars is not None else None
        val, beta, ll = np.empty(B), np.empty(B), np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        self._chk(lib().ccgp_logpost_batch(self._h, _p(X), n, d, _p(y), float(sigma2), int(prior_id), _p(theta_t), B,
                                           _p(pp), _p(val), _p(beta), _p(ll), _ipt(st)))
        return val, beta, ll, st

    def grid_marginal(self, X, y, sigma2, hyper, N, tau, take_log, aniso_lambda=-1.0, want_logs=False):
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        hyper = _f(hyper)
        G = hyper.shape[0]
        if hyper.shape[1] != 4:
            raise ValueError("hyper must be G x 4")
        out = np.empty(G)
        arg = c_int()
        logs = np.empty((G, N), dtype=np.float64) if want_logs else None
        self._chk(lib().ccgp_grid_marginal(self._h, _p(X), n, d, _p(y), float(sigma2), _p(hyper), G,
                                           int(N), float(tau), 1 if take_log else 0,
                                           float(aniso_lambda), _p(out), ctypes.byref(arg), _p(logs)))
        return (out, arg.value, logs) if want_logs else (out, arg.value)

    # -- 8(f)-4: entropy criteria ------------------------------------------------------------
    def mixed_logdet_designs(self, designs, K, params_row):
        """designs: [B, n, d] candidate designs sharing one parameter row -> (logdet[B], status[B])."""
        designs = np.asarray(designs, dtype=np.float64)
        B, n, d = designs.shape
        Xs = np.ascontiguousarray(np.stack([np.asfortranarray(D).ravel(order="F") for D in designs]))
        row = _f(params_row, (K + K * d,))
        out = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        self._chk(lib().ccgp_mixed_logdet_designs(self._h, _p(Xs), n, d, B, K, _p(row), _p(out), _ipt(st)))
        return out, st

    # -- a10 + a11 -------------------------------------------------------------------------
    def predict_batch(self, X, y, K, params, Xtest, sigma2):
        """Returns (mean[S, m], var[S, m], beta[S], status[S])."""
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        params = _f(np.atleast_2d(params))
        S = params.shape[0]
        Xtest = _f(np.atleast_2d(Xtest))
        m = Xtest.shape[0]
        mean = np.empty((S, m), dtype=np.float64, order="F")
        var = np.empty((S, m), dtype=np.float64, order="F")
        beta = np.empty(S)
        st = np.zeros(S, dtype=np.int32)
        self._chk(lib().ccgp_predict_batch(self._h, _p(X), n, d, _p(y), K, _p(params), S, _p(Xtest), m,
                                           float(sigma2), _p(mean), _p(var), _p(beta), _ipt(st)))
        return mean, var, beta, st

    # -- 8(f)-2: device-resident factor set -------------------------------------------------
    def factor_batch(self, X, y, K, params, sigma2):
        """Factorise the S draws once and keep the factors on the device -> FactorSet (use as a context manager or
        call .free()).  .loglik / .beta / .status hold what ccgp_loglik_batch would have returned."""
        X, y = _f(X), _f(np.ravel(y))
        n, d = X.shape
        params = _f(np.atleast_2d(params))
        S = params.shape[0]
        ll, beta = np.empty(S), np.empty(S)
        st = np.zeros(S, dtype=np.int32)
        fs = c_void_p()
        self._chk(lib().ccgp_factor_batch(self._h, _p(X), n, d, _p(y), K, _p(params), S, float(sigma2),
                                          ctypes.byref(fs), _p(ll), _p(beta), _ipt(st)))
        return FactorSet(self, fs, S, ll, beta, st)

    # -- device-resident forms (torch tensors or raw pointers) ----------------------------
    @staticmethod
    def _dptr(t):
        if t is None:
            return c_void_p(0)
        if isinstance(t, int):
            return c_void_p(t)
        return c_void_p(t.data_ptr())

    def loglik_batch_dev(self, dX, n, d, dy, K, dparams, B, sigma2, mean_mode, tau2, d_loglik, d_beta,
                         d_status):
        """All d* are device buffers (torch CUDA tensors or integer addresses): dX n*d
        column-major, dparams B x P column-major, d_status int32[B].  Asynchronous."""
        self._chk(lib().ccgp_loglik_batch_dev(self._h, self._dptr(dX), n, d, self._dptr(dy), K,
                                              self._dptr(dparams), B, float(sigma2), int(mean_mode),
                                              float(tau2), self._dptr(d_loglik), self._dptr(d_beta),
                                              self._dptr(d_status)))

    def predict_batch_dev(self, dX, n, d, dy, K, dparams, S, dXtest, m, sigma2, d_mean, d_var, d_beta,
                          d_status):
        self._chk(lib().ccgp_predict_batch_dev(self._h, self._dptr(dX), n, d, self._dptr(dy), K,
                                               self._dptr(dparams), S, self._dptr(dXtest), m,
                                               float(sigma2), self._dptr(d_mean), self._dptr(d_var),
                                               self._dptr(d_beta), self._dptr(d_status)))
