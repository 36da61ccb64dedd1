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


def _strip_r_comments_and_strings(src):
    """R source with comments removed and string literals blanked (their quotes kept), for bracket / call scanning."""
    out, i, n, quote = [], 0, len(src), None
    while i < n:
        c = src[i]
        if quote:
            if c == "\\":
                out.append("  ")
                i += 2
                continue
            out.append(c if c == quote else " ")
            if c == quote:
                quote = None
        elif c == "#":
            while i < n and src[i] != "\n":
                i += 1
            continue
        else:
            if c in "\"'":
                quote = c
            out.append(c)
        i += 1
    return "".join(out), quote


def test_r_overrides_call_registered_routines_with_the_registered_arity():
    """r/ccgp.R cannot run here (no R).  What can be checked statically: its brackets balance, and every
    .Call("ccgp_R_*", ...) names a routine that r/ccgp_shim.c registers and passes exactly the registered number of
    arguments (R would stop with "Incorrect number of arguments" at run time)."""
    raw = open(os.path.join(ROOT, "r", "ccgp.R")).read()
    shim = open(os.path.join(ROOT, "r", "ccgp_shim.c")).read()
    table = dict((n, int(k)) for n, k in re.findall(r'\{"(ccgp_R_\w+)", \(DL_FUNC\)&\w+, (\d+)\}', shim))
    src, open_quote = _strip_r_comments_and_strings(raw)
    assert open_quote is None
    stack, pairs = [], {")": "(", "}": "{", "]": "["}
    for k, c in enumerate(src):
        if c in "({[":
            stack.append(c)
        elif c in ")}]":
            assert stack and stack.pop() == pairs[c], "unbalanced %r near line %d" % (c, src[:k].count("\n") + 1)
    assert not stack
    sites = 0
    for m in re.finditer(r'\.Call\("(ccgp_R_\w+)"', raw):
        name = m.group(1)
        line = raw[:m.start()].count("\n") + 1
        assert name in table, "%s (line %d) is not registered" % (name, line)
        # the same offsets hold in `src` (strings are blanked in place, comments removed only AFTER this call's line
        # start in the cases below, so re-find the call in the stripped text)
        sm = [x for x in re.finditer(r'\.Call\("', src) if src[:x.start()].count("\n") + 1 == line]
        assert sm, line
        p, depth, commas = src.index("(", sm[0].start()) + 1, 1, 0
        while depth:
            ch = src[p]
            depth += ch in "([{"
            depth -= ch in ")]}"
            commas += ch == "," and depth == 1
            p += 1
        assert commas == table[name], "%s at line %d passes %d arguments, %d registered" % (name, line, commas, table[name])
        sites += 1
    assert sites >= 30
