"""Python side of the functional R-API mock (rmock.c): builds R objects from numpy, issues `.Call`s through the
REGISTERED routine table of r/ccgp_shim.c, converts results back, and exposes the mock's bookkeeping (PROTECT
balance, GC hazards, type errors, captured warnings).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "ccgpR_mock.so")
NILSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 0, 9, 10, 13, 14, 16, 19
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


class RError(RuntimeError):
    """Rf_error() inside a .Call (R would stop() with this message)."""


class NAType:
    def __repr__(self):
        return "NA"


NA = NAType()


def build():
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)


class MockR:
    def __init__(self):
        if not os.path.exists(SO):
            build()
        L = ctypes.CDLL(SO)
        vp = ctypes.c_void_p
        L.rmock_load.restype = ctypes.c_int
        L.rmock_routine_name.restype = ctypes.c_char_p
        L.rmock_routine_name.argtypes = [ctypes.c_int]
        L.rmock_routine_nargs.restype = ctypes.c_int
        L.rmock_routine_nargs.argtypes = [ctypes.c_int]
        L.rmock_dot_call.restype = vp
        L.rmock_dot_call.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(vp)]
        L.rmock_real.restype = vp
        L.rmock_real.argtypes = [_dp, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int]
        L.rmock_int.restype = vp
        L.rmock_int.argtypes = [_ip, ctypes.c_ssize_t]
        L.rmock_nil.restype = vp
        L.rmock_list.restype = vp
        L.rmock_list.argtypes = [ctypes.c_ssize_t]
        L.rmock_list_set.restype = None
        L.rmock_list_set.argtypes = [vp, ctypes.c_ssize_t, vp]
        L.rmock_list_names.restype = None
        L.rmock_list_names.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
        for name, res, args in (("rmock_typeof", ctypes.c_int, [vp]), ("rmock_length", ctypes.c_long, [vp]),
                                ("rmock_dim", ctypes.c_int, [vp, ctypes.c_int]), ("rmock_data", vp, [vp]),
                                ("rmock_elt", vp, [vp, ctypes.c_long]), ("rmock_names", vp, [vp]),
                                ("rmock_char", ctypes.c_char_p, [vp]), ("rmock_warning", ctypes.c_char_p, [ctypes.c_int]),
                                ("rmock_last_error", ctypes.c_char_p, []), ("rmock_counter", ctypes.c_int, [ctypes.c_int]),
                                ("rmock_n_warnings", ctypes.c_int, []), ("rmock_na_real", ctypes.c_double, []),
                                ("rmock_is_na_real", ctypes.c_int, [ctypes.c_double]),
                                ("rmock_dynamic_symbols", ctypes.c_int, [])):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        self.L = L
        self.n_routines = L.rmock_load()          # dyn.load(): runs R_init_ccgpR
        self.routines = {L.rmock_routine_name(i).decode(): L.rmock_routine_nargs(i) for i in range(self.n_routines)}

    # ---- R objects -----------------------------------------------------------------------------------------
    def real(self, a):
        """numpy -> REALSXP: 2-D arrays become matrices (column-major, with `dim`), everything else a plain vector."""
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 2:
            f = np.asfortranarray(a)
            return self.L.rmock_real(f.ctypes.data_as(_dp), f.size, a.shape[0], a.shape[1])
        f = np.ascontiguousarray(a.ravel())
        return self.L.rmock_real(f.ctypes.data_as(_dp), f.size, -1, 0)

    def integer(self, a):
        f = np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.int32)))
        return self.L.rmock_int(f.ctypes.data_as(_ip), f.size)

    def frame(self, table, names=None, integer_columns=()):
        """2-D numpy table -> data.frame as .Call sees one: a VECSXP of REALSXP columns (INTSXP for the listed column
        indices: read.table gives integer columns for whole numbers) with a `names` attribute."""
        table = np.asarray(table, dtype=np.float64)
        lst = self.L.rmock_list(table.shape[1])
        for j in range(table.shape[1]):
            col = self.integer(table[:, j]) if j in integer_columns else self.real(table[:, j])
            self.L.rmock_list_set(lst, j, col)
        if names is not None:
            arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
            self.L.rmock_list_names(lst, arr, len(names))
        return lst

    def null(self):
        return self.L.rmock_nil()

    def to_python(self, sx):
        """SEXP -> python: REALSXP -> float array (matrix if `dim`), INTSXP -> int array, LGLSXP -> int array, VECSXP ->
        list or dict (when named)."""
        L = self.L
        t, n = L.rmock_typeof(sx), L.rmock_length(sx)
        if t == NILSXP:
            return None
        if t in (REALSXP, INTSXP, LGLSXP):
            ct = ctypes.c_double if t == REALSXP else ctypes.c_int
            buf = (ct * n).from_address(L.rmock_data(sx)) if n else []
            arr = np.array(buf, dtype=np.float64 if t == REALSXP else np.int32)
            nr = L.rmock_dim(sx, 0)
            if nr >= 0:
                arr = arr.reshape((nr, L.rmock_dim(sx, 1)), order="F")
            return arr
        if t == CHARSXP:
            return L.rmock_char(sx).decode()
        if t == STRSXP:
            return [L.rmock_char(L.rmock_elt(sx, i)).decode() for i in range(n)]
        if t == VECSXP:
            items = [self.to_python(L.rmock_elt(sx, i)) for i in range(n)]
            names = L.rmock_names(sx)
            if L.rmock_typeof(names) == STRSXP:
                return dict(zip(self.to_python(names), items))
            return items
        raise TypeError("unsupported SEXP type %d" % t)

    def is_na(self, x):
        """R's NA_real_ (NaN with payload 1954), elementwise."""
        a = np.atleast_1d(np.asarray(x, dtype=np.float64))
        return np.array([bool(self.L.rmock_is_na_real(float(v))) for v in a.ravel()]).reshape(a.shape)

    # ---- .Call ----------------------------------------------------------------------------------------------
    def dot_call(self, name, *args):
        arr = (ctypes.c_void_p * max(len(args), 1))(*args)
        r = self.L.rmock_dot_call(name.encode(), len(args), arr)
        if not r:
            raise RError(self.L.rmock_last_error().decode())
        return self.to_python(r)

    # ---- bookkeeping -----------------------------------------------------------------------------------------
    def warnings(self):
        return [self.L.rmock_warning(i).decode() for i in range(min(self.L.rmock_n_warnings(), 16))]

    def counters(self):
        names = ("unbalanced_protect", "gc_hazards", "type_errors", "unprotect_underflows", "protect_depth")
        return {k: self.L.rmock_counter(i) for i, k in enumerate(names)}

    def assert_clean(self):
        c = self.counters()
        assert not any(c.values()), c

    def reset(self):
        self.L.rmock_reset()

    def unload(self):
        self.L.rmock_unload()
