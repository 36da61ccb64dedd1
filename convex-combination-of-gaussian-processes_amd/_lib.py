"""ctypes loader for libccgp.so.  There is deliberately NO fallback: if the HIP
library has not been built (`make -C convex-combination-of-gaussian-processes_amd/csrc`
or `python -c "import __graft_entry__ as g; g.build()"`) every product call raises."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LibraryMissing(RuntimeError):
    pass


def library_path() -> str:
    """In-tree build product; CCGP_LIB overrides it (A/B runs of differently built libraries)."""
    return os.environ.get("CCGP_LIB") or os.path.join(_HERE, "csrc", "libccgp.so")


def load_library() -> ctypes.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise LibraryMissing(
            "libccgp.so not built at %s -- run `make -C %s` (needs hipcc); there is no CPU fallback"
            % (path, os.path.join(_HERE, "csrc")))
    _LIB = ctypes.CDLL(path)
    return _LIB
