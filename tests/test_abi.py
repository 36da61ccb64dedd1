"""The C-ABI library loads on a CPU-only host and exports every symbol include/ccgp.h declares.
No compute entry point is called here (no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from ccgp_amd import api, library_path


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ccgp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccgp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(library_path()), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    L = ctypes.CDLL(library_path())
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "missing export " + n


def test_python_binding_covers_header():
    assert sorted(api.SIGNATURES) == declared_symbols()


def test_version_string():
    assert b"gfx950" in api.lib().ccgp_version()


def test_bad_arguments_do_not_crash():
    L = api.lib()
    assert L.ccgp_destroy(None) == 0
    assert L.ccgp_halton_base2(-1, None) == -1
    out = np.empty(1)
    assert L.ccgp_qigamma(None, 1, 1.0, 1.0, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == -1


def test_no_cpu_fallback_without_device():
    """On a host without a HIP device the product path must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.CcgpError):
        api.Handle(0)
