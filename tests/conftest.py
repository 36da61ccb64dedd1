import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ccgp_amd  # noqa: E402,F401  (registers the package alias)
from ccgp_amd.tables import read_table  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def load_qian():
    _, tr = read_table(os.path.join(DATA, "qian_train.txt"))
    _, te = read_table(os.path.join(DATA, "qian_test.txt"))
    return tr[:, :4], tr[:, 4], te[:, :4], te[:, 4]


def load_gv(size, sample=1):
    _, tr = read_table(os.path.join(DATA, "gv", "train_%d_%d.txt" % (size, sample)))
    _, te = read_table(os.path.join(DATA, "gv", "test_%d_%d.txt" % (size, sample)))
    return tr[:, :9], tr[:, 9], te[:, :9], te[:, 9]


def load_maximin(npts):
    _, D = read_table(os.path.join(DATA, "maximin_%d.txt" % npts))
    return D


def load_hyper(which):
    _, H = read_table(os.path.join(DATA, "%s_hyperpars_matrix.txt" % which))
    return H


@pytest.fixture(scope="session")
def handle():
    """A libccgp handle on GPU 0 -- only requested by gpu-marked tests."""
    try:
        import torch  # noqa: F401  (a test that also uses torch needs it loaded BEFORE libccgp: INTEGRATION.md section 5)
    except ImportError:
        pass
    from ccgp_amd import api
    h = api.Handle(0)
    yield h
    h.close()


def synthetic_design(n, d, seed):
    """Seeded random Latin hypercube in [0,1]^d (the bench's cfg4 generator lives in bench.py;
    this is the small-size stand-in for parity tests)."""
    rng = np.random.default_rng(seed)
    X = np.empty((n, d))
    for k in range(d):
        X[:, k] = (rng.permutation(n) + rng.random(n)) / n
    y = np.sin(2 * np.pi * X).sum(axis=1)
    return X, y
