"""CPU oracle for the Combined-GP hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product path (the ``ccgp_amd`` package
and ``libccgp.so``) never does and fails loudly without its HIP library.

The reference (oharari/Convex-Combination-of-Gaussian-Processes) is eight R scripts with no
tests and no golden vectors, and R is not installed in the build container, so this
restatement cannot be checked against the reference running.  What pins it (DESIGN.md
section (c)): the ONE output the reference records, ``Ground Vibrations Emulator/Results/
Size 50 Results 1.txt`` -- its single-GP columns are reproduced to 1e-8 on all 450 numbers,
its Combined-GP columns up to Monte-Carlo noise -- and the hyperprior pair hard-coded at
HX:774-775, which is the argmax of the restated grid (tests/test_reference_pins.py).
PARITY UNPINNED for what no reference-held number covers: the log-likelihood scalar itself
(arbitrated by the independent 50-digit mpmath evaluation ``oracle/mp_check.py`` and by
LAPACK), the Halton start index, the Matern / spline / ADV results, base R's solve()
tolerance rule (restated from its documentation).
"""
