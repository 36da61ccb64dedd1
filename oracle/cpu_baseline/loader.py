"""ctypes loader of the CPU baseline evaluator (oracle/cpu_baseline/ccgp_cpu.c).  Test / bench infrastructure:
imported only by bench.py's cpu_baseline leg and tests/test_cpu_baseline.py."""
import ctypes
import glob
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_lib = None
_blas = None


def openblas_path():
    """The OpenBLAS that scipy bundles (LP64 build: scipy_dpotrf_ / scipy_dtrsv_)."""
    try:
        import scipy
        hits = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
        return os.path.abspath(hits[0]) if hits else ""
    except Exception:
        return ""


def load(build=True):
    global _lib, _blas
    if _lib is not None:
        return _lib
    so = os.path.join(HERE, "libccgp_cpu.so")
    if not os.path.exists(so) and build:
        subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(so)
    L.ccgp_cpu_init.restype = ctypes.c_int
    L.ccgp_cpu_init.argtypes = [ctypes.c_char_p]
    L.ccgp_cpu_max_threads.restype = ctypes.c_int
    L.ccgp_cpu_loglik_batch.restype = ctypes.c_int
    L.ccgp_cpu_loglik_batch.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double, _dp, _dp, _ip,
                                        ctypes.c_int]
    L.ccgp_cpu_predict_batch.restype = ctypes.c_int
    L.ccgp_cpu_predict_batch.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                         ctypes.c_int, _dp, ctypes.c_int, ctypes.c_double, _dp, _dp, ctypes.c_int]
    L.ccgp_cpu_loglik_seq.restype = ctypes.c_int
    L.ccgp_cpu_loglik_seq.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double, _dp, _dp, _ip,
                                      ctypes.c_int]
    L.ccgp_cpu_predict_post.restype = None
    L.ccgp_cpu_predict_post.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_double, _dp, _dp,
                                        ctypes.c_double, _dp, ctypes.c_double, _dp, _dp]
    _blas = bool(L.ccgp_cpu_init(openblas_path().encode()))
    _lib = L
    return L


def lapack_bound():
    load()
    return _blas


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp)


def loglik_batch(X, y, K, params, sigma2, mode=0, tau2=0.0, threads=0):
    L = load()
    X, y, params = _f(X), _f(np.ravel(y)), _f(np.atleast_2d(params))
    n, d = X.shape
    B = params.shape[0]
    ll, beta = np.empty(B), np.empty(B)
    st = np.zeros(B, dtype=np.int32)
    L.ccgp_cpu_loglik_batch(_p(X), n, d, _p(y), K, _p(params), B, B, float(sigma2), int(mode), float(tau2), _p(ll),
                            _p(beta), st.ctypes.data_as(_ip), int(threads))
    return ll, beta, st


def loglik_seq(X, y, K, params, sigma2, mode=0, tau2=0.0, inner=1):
    """One evaluation at a time, each with `inner` threads to itself (cpu_worker.py: the concurrency x threads sweep)."""
    L = load()
    X, y, params = _f(X), _f(np.ravel(y)), _f(np.atleast_2d(params))
    n, d = X.shape
    B = params.shape[0]
    ll, beta = np.empty(B), np.empty(B)
    st = np.zeros(B, dtype=np.int32)
    L.ccgp_cpu_loglik_seq(_p(X), n, d, _p(y), K, _p(params), B, B, float(sigma2), int(mode), float(tau2), _p(ll),
                          _p(beta), st.ctypes.data_as(_ip), int(inner))
    return ll, beta, st


class PredictPost:
    """One literal predict.post per call with the cached terms bound once (buffers prepared outside the timed call, as an R
    caller's frame row already holds them)."""

    def __init__(self, X, K, row, beta, mean_factor, v1, v2, R_inv, sigma2):
        self.L = load()
        self.X, self.row = _f(X), _f(np.ravel(row))
        self.mf, self.v1, self.Ri = _f(np.ravel(mean_factor)), _f(np.ravel(v1)), _f(R_inv)
        self.n, self.d = self.X.shape
        self.K, self.beta, self.v2, self.s2 = int(K), float(beta), float(v2), float(sigma2)
        self.r, self.out = np.empty(self.n), np.empty(2)
        self.args = (_p(self.X), self.n, self.d, self.K, _p(self.row), self.beta, _p(self.mf), _p(self.v1), self.v2,
                     _p(self.Ri), self.s2, _p(self.r), _p(self.out))

    def __call__(self, x):
        x = _f(np.ravel(x))
        self.L.ccgp_cpu_predict_post(_p(x), *self.args)
        return self.out


def predict_batch(X, y, K, params, Xtest, sigma2, threads=0):
    L = load()
    X, y, params, Xtest = _f(X), _f(np.ravel(y)), _f(np.atleast_2d(params)), _f(np.atleast_2d(Xtest))
    n, d = X.shape
    S, m = params.shape[0], Xtest.shape[0]
    mean = np.empty((S, m), order="F")
    var = np.empty((S, m), order="F")
    L.ccgp_cpu_predict_batch(_p(X), n, d, _p(y), K, _p(params), S, S, _p(Xtest), m, float(sigma2), _p(mean), _p(var),
                             int(threads))
    return mean, var


def max_threads():
    return int(load().ccgp_cpu_max_threads())
