"""SURVEY.md section 5, "race detection / sanitizers: build-side ASan on the host shim".  csrc/multi.cpp (one host
thread per shard, results gathered into the caller's buffers at the shard's offset) and r/ccgp_shim.c (with the
functional R-API mock) are built on top of a CPU stub of the device entry points and run under AddressSanitizer +
UBSan and under ThreadSanitizer: 1 / 2 / 3 / 8 shards, ragged shards, fewer items than shards, a failing evaluation,
a failing shard, CCGP_DEVICES parsing.  CPU only: sanitizers are never run on the GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT

HERE = os.path.join(ROOT, "tests", "host_san")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", HERE, "asan", "tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind,marker", [("asan", "AddressSanitizer"), ("tsan", "ThreadSanitizer")])
def test_host_side_is_clean_under(built, kind, marker):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    env.pop("CCGP_DEVICES", None)
    r = subprocess.run([os.path.join(HERE, "driver_" + kind)], capture_output=True, text=True, env=env, timeout=240)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "host-sanitizer driver: OK" in r.stdout
    assert marker not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-3000:]
