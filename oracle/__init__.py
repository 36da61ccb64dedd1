"""CPU oracle for the Combined-GP hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product path (the ``ccgp_amd`` package
and ``libccgp.so``) never does and fails loudly without its HIP library.

PARITY UNPINNED: the reference (oharari/Convex-Combination-of-Gaussian-Processes)
is eight R scripts with no tests, no golden vectors and no recorded deterministic
output, and R is not installed in the build container, so this restatement cannot
be checked against the reference running.  It is pinned instead by (i) an
independent 50-digit mpmath re-evaluation (``oracle/mp_check.py``) and (ii)
analytic properties (tests/test_oracle.py).  See DESIGN.md section (c).
"""
