"""The R shim's host-side behaviour that needs no GPU: it builds against the functional R-API mock (tests/r_mock/),
registers its routines, dispatches by the registered argument count, and -- on a machine without a HIP device --
every device routine ends in Rf_error("... no HIP device ... no CPU fallback") with the PROTECT stack unwound.
The GPU twin (tests/test_gpu_r_shim.py) executes every routine for real."""
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "r_mock"))
import rmock  # noqa: E402


@pytest.fixture(scope="module")
def R():
    rmock.build()
    r = rmock.MockR()
    yield r


def test_registration_table_matches_the_source(R):
    src = open(os.path.join(ROOT, "r", "ccgp_shim.c")).read()
    table = dict((n, int(k)) for n, k in re.findall(r'\{"(ccgp_R_\w+)", \(DL_FUNC\)&\w+, (\d+)\}', src))
    assert len(table) >= 15 and R.routines == table
    assert R.L.rmock_dynamic_symbols() == 0          # R_useDynamicSymbols(dll, FALSE)


def test_argument_count_and_unknown_routine_are_errors(R):
    R.reset()
    with pytest.raises(rmock.RError, match="Incorrect number of arguments"):
        R.dot_call("ccgp_R_beta_mle", R.real(np.eye(2)))
    with pytest.raises(rmock.RError, match="not available"):
        R.dot_call("ccgp_loglik_batch")
    R.assert_clean()


def test_mock_semantics_the_shim_relies_on(R):
    """nrows / ncols on matrices and plain vectors, NA payload, coercions -- as in R."""
    R.reset()
    L = R.L
    import ctypes
    L.Rf_nrows.argtypes = L.Rf_ncols.argtypes = L.Rf_asInteger.argtypes = [ctypes.c_void_p]
    L.Rf_asReal.argtypes = [ctypes.c_void_p]
    L.Rf_asReal.restype = ctypes.c_double
    m = R.real(np.zeros((3, 5)))
    v = R.real(np.zeros(7))
    assert (L.Rf_nrows(m), L.Rf_ncols(m)) == (3, 5) and (L.Rf_nrows(v), L.Rf_ncols(v)) == (7, 1)
    assert L.Rf_asInteger(R.real([2.9])) == 2 and L.Rf_asInteger(R.real([np.nan])) == -2 ** 31
    assert L.Rf_asReal(R.integer(4)) == 4.0
    na = L.rmock_na_real()
    assert na != na and R.is_na(na).all() and not R.is_na(float("nan")).any()
    assert R.to_python(R.real(np.arange(6.0).reshape(2, 3))).tolist() == [[0, 1, 2], [3, 4, 5]]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a HIP device is present: the no-device path cannot be reached")
def test_without_a_device_every_routine_stops_with_an_r_error(R):
    R.reset()
    D = np.random.default_rng(0).random((6, 2))
    with pytest.raises(rmock.RError, match="no HIP device 0 .there is no CPU fallback."):
        R.dot_call("ccgp_R_corr_matrix", R.real(D), R.real([1.0, 2.0]))
    c = R.counters()
    assert c["protect_depth"] == 0 and c["type_errors"] == 0      # Rf_error unwound the PROTECTed result matrix


def test_r_overrides_coerce_every_call_argument():
    """REAL() on an integer vector is an error in R (a design or a hyperparameter table read from a file of whole
    numbers is integer): r/ccgp.R must hand the shim doubles / integers explicitly."""
    rsrc = open(os.path.join(ROOT, "r", "ccgp.R")).read()
    assert ".ccgp.mat <- function" in rsrc
    assert "as.matrix(" not in rsrc.replace(".ccgp.mat <- function(x) { x <- as.matrix(x)", "")
