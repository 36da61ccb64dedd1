"""Prototype-drift guard for r/ccgp_shim.c (CPU).  The container has no R, so the shim cannot be built for real;
this compiles it (front end only) against the REAL include/ccgp.h and a minimal MOCK of the R headers
(tests/r_mock/): a changed libccgp signature, a missing symbol or a wrong argument count in the shim is then a
compile error here.  It does not show that the shim works under R."""
import os
import re
import subprocess

from conftest import ROOT


def test_shim_compiles_against_the_header():
    cmd = ["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration",
           "-Werror=incompatible-pointer-types", "-Werror=int-conversion", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "r_mock"), os.path.join(ROOT, "r", "ccgp_shim.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_registered_routine_is_defined_with_that_many_arguments():
    src = open(os.path.join(ROOT, "r", "ccgp_shim.c")).read()
    table = re.findall(r'\{"(ccgp_R_\w+)", \(DL_FUNC\)&(\w+), (\d+)\}', src)
    assert len(table) >= 14
    for name, sym, nargs in table:
        assert name == sym
        m = re.search(r"SEXP %s\(([^)]*)\)" % name, src)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == int(nargs), name
    # and every .Call in r/ccgp.R names a registered routine
    rsrc = open(os.path.join(ROOT, "r", "ccgp.R")).read()
    for called in set(re.findall(r'\.Call\("(\w+)"', rsrc)):
        assert called in {t[0] for t in table}, called
