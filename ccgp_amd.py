"""Import alias: the package directory is named after the reference repository
(`convex-combination-of-gaussian-processes_amd/`), which is not a valid Python
identifier.  `import ccgp_amd` loads that directory as the package `ccgp_amd`."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "convex-combination-of-gaussian-processes_amd")
_spec = importlib.util.spec_from_file_location(
    "ccgp_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ccgp_amd"] = _mod
_spec.loader.exec_module(_mod)
