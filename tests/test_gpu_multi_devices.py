"""Multi-GPU tests that switch themselves on: everything here needs at least TWO distinct devices and is skipped on the
one-GPU boxes of the pool (the skip count is visible in the GPU test record).  On the first multi-GPU box the suite meets,
RCCL itself (bench.py's one all-gather over xGMI, `shard.RowGatherer`) and one host thread per DISTINCT device
(`ccgp_multi_*`, csrc/multi.cpp) execute for the first time -- until then they are rehearsed with ranks / shards that
share device 0 (tests/test_gpu_bench_ranks.py, tests/test_gpu_multi.py, scripts/rehearse_n2.sh).
SURVEY 8(e); BASELINE configs 4 and 5 ("sharded over 8 GPUs")."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_gv, load_hyper, load_qian, synthetic_design


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()      # counting devices does not initialise the GPU on this image
    except Exception:
        return 0


NDEV = _device_count()
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(NDEV < 2, reason="needs two distinct GPUs (this box has %d)" % NDEV)]


def iso_row(p, t1, t2, d):
    return np.concatenate([[p, 1 - p], np.full(d, t1), np.full(d, t2)])


# ---- (a) one process per GPU, RCCL all-gather -----------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_bench_two_ranks_over_rccl_cfg4_slice():
    from test_gpu_bench_ranks import run_bench
    r = run_bench("--gpus", "2", "--backend", "nccl", "--workload", "cfg4", "--evals-total", "16", "--steps", "1",
                  "--warmup", "0", "--no-cpu-baseline", "--no-secondary")
    # bench.py asserts the gathered vector against every rank's local slice ("all-gather mismatch") and the timed
    # log-likelihoods against the CPU potrf digest; reaching the JSON line means both held on both ranks
    assert r["n_gpus"] == 2 and r["config"]["evals_per_gpu"] == 8 and r["config"]["failed_evals"] == 0
    assert r["config"]["matches_cpu_potrf_digest"] is True and "gloo" not in r["config"]["parallelism"]


@pytest.mark.timeout(900)
def test_bench_two_ranks_over_rccl_prediction_tables():
    from test_gpu_bench_ranks import run_bench
    one = run_bench("--gpus", "1", "--workload", "cfg5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    two = run_bench("--gpus", "2", "--backend", "nccl", "--workload", "cfg5", "--steps", "1", "--warmup", "0",
                    "--no-cpu-baseline")
    assert two["n_gpus"] == 2 and two["config"]["draws_per_gpu"] == 500 and two["config"]["failed_draws"] == 0
    assert two["config"]["gathered_bytes"] == one["config"]["gathered_bytes"] and two["config"]["all_finite"]


@pytest.mark.timeout(900)
@pytest.mark.skipif(NDEV < 3, reason="needs three distinct GPUs")
def test_bench_three_ranks_over_rccl_grid_by_row_ragged():
    from test_gpu_bench_ranks import run_bench
    r = run_bench("--gpus", "3", "--backend", "nccl", "--workload", "cfg3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r["n_gpus"] == 3 and r["config"]["evals_per_gpu"] == 20 * 1728 and r["config"]["failed_evals"] == 0


# ---- (b) one process, one host thread per DISTINCT device ------------------------------------------------------------
@pytest.fixture(scope="module")
def multi():
    from ccgp_amd import api
    m = api.MultiHandle(list(range(min(NDEV, 4))))
    yield m
    m.close()


def test_distinct_devices_loglik_small_and_blocked(handle, multi):
    D, y, _, _ = load_qian()
    rng = np.random.default_rng(2)
    P = np.array([iso_row(rng.uniform(0.5, 0.9), rng.uniform(0.2, 1), rng.uniform(5, 30), 4) for _ in range(37)])
    for mode, tau2 in ((0, 0.0), (1, 2500.0)):
        for u, v in zip(handle.loglik_batch(D, y, 2, P, 37.0, mode, tau2), multi.loglik_batch(D, y, 2, P, 37.0, mode, tau2)):
            np.testing.assert_array_equal(u, v)
    X, yy = synthetic_design(300, 3, 1)                                                 # blocked path on every device
    rows = np.array([np.concatenate([[0.5, 0.5], [1.0, 2.0, 3.0 + k], [30.0, 40.0, 50.0]]) for k in range(7)])
    for u, v in zip(handle.loglik_batch(X, yy, 2, rows, 1.0), multi.loglik_batch(X, yy, 2, rows, 1.0)):
        np.testing.assert_array_equal(u, v)


def test_distinct_devices_grid_by_row_and_predict_by_draw(handle, multi):
    D, y, _, _ = load_qian()
    H = load_hyper("hx")[:21]
    va, aa, la = handle.grid_marginal(D, y, 37.0, H, 200, 50.0, True, want_logs=True)
    vb, ab, lb = multi.grid_marginal(D, y, 37.0, H, 200, 50.0, True, want_logs=True)
    np.testing.assert_array_equal(va, vb)
    np.testing.assert_array_equal(la, lb)
    assert aa == ab
    Dg, yg, Dt, _ = load_gv(90)
    rng = np.random.default_rng(5)
    P = np.array([iso_row(rng.uniform(0.5, 0.9), rng.uniform(0.2, 0.5), rng.uniform(10, 20), 9) for _ in range(23)])
    for u, v in zip(handle.predict_batch(Dg, yg, 2, P, Dt, 10.0), multi.predict_batch(Dg, yg, 2, P, Dt, 10.0)):
        np.testing.assert_array_equal(u, v)


def test_distinct_devices_a_failing_evaluation_stays_in_its_shard(handle, multi):
    D, y, _, _ = load_qian()
    P = np.array([iso_row(0.7, 0.3 + 0.01 * k, 15.0, 4) for k in range(9)])
    P[7, 2:] = 0.0                        # R = 11': exactly singular, lands in the last shard
    a = handle.loglik_batch(D, y, 2, P, 37.0)
    b = multi.loglik_batch(D, y, 2, P, 37.0)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    assert np.isnan(b[0][7]) and b[2][7] != 0 and np.isfinite(np.delete(b[0], 7)).all()


# ---- (c) the R shim with CCGP_DEVICES naming distinct devices --------------------------------------------------------
def test_r_shim_ccgp_devices_on_distinct_gpus(handle):
    sys.path.insert(0, os.path.join(ROOT, "tests", "r_mock"))
    import rmock
    from ccgp_amd import api
    D, y, Dt, _ = load_gv(90)
    rng = np.random.default_rng(5)
    P = np.array([iso_row(rng.uniform(0.5, 0.9), rng.uniform(0.2, 0.5), rng.uniform(10, 20), 9) for _ in range(11)])
    os.environ["CCGP_DEVICES"] = "0,1"
    R = rmock.MockR()
    try:
        R.reset()
        assert R.dot_call("ccgp_R_devices")[0] == 2
        got = R.dot_call("ccgp_R_loglik_batch", R.real(D), R.real(y), R.integer(2), R.real(P), R.real(10.0),
                         R.integer(api.MEAN_PROFILE_BETA), R.real(0.0))
        ll, beta, _ = handle.loglik_batch(D, y, 2, P, 10.0)
        assert np.array_equal(got[0], ll) and np.array_equal(got[1], beta)
        fr = np.column_stack([P[:, 0], P[:, 2], P[:, 11]])
        got = R.dot_call("ccgp_R_prediction_table", R.real(fr), R.real(D), R.real(Dt[:20]), R.real(10.0), R.real(y),
                         R.integer(0), R.real(0.0))
        mean, var, b2, _ = handle.predict_batch(D, y, 2, P, Dt[:20], 10.0)
        assert np.array_equal(got[0], mean) and np.array_equal(got[1], var) and np.array_equal(got[2], b2)
        assert R.warnings() == []
        R.assert_clean()
    finally:
        R.unload()
        os.environ.pop("CCGP_DEVICES", None)
