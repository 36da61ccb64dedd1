"""MI355X-native Combined-GP evaluation path (hand-written HIP behind a C ABI).

Layout
  csrc/        HIP kernels + the extern "C" library (libccgp.so, include/ccgp.h)
  _lib.py      ctypes loader -- raises if the library is missing (no CPU fallback)
  api.py       thin numpy / device-pointer wrappers of every C-ABI entry point
  rsurface.py  the reference's R function surface (same names, argument order and
               return shapes) on top of api.py
  tables.py    reader/writer for the reference's text tables
  shard.py     one-process-per-GPU sharding of the independent evaluations
  fit.py       host-side counterpart of laplace / Metro / prediction / Combined.GP.fit (seeded RNG)
"""
from . import tables  # noqa: F401  (pure Python, usable without the library)
from ._lib import LibraryMissing, library_path, load_library  # noqa: F401

__all__ = ["tables", "api", "rsurface", "shard", "fit", "load_library", "library_path", "LibraryMissing"]


def __getattr__(name):
    if name in ("api", "rsurface", "shard", "fit"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
