#!/usr/bin/env python3
"""Benchmark of the GP log-likelihood hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload cfg4|cfg2|cfg3|cfg5] [--evals-per-gpu E]

metric   : GP log-lik evals/sec (n x n fp64, batched over the hyperparameter grid)
workload : cfg4 (default) = BASELINE config 4, the configuration the north_star's MFMA
           target is quoted on: synthetic 5-D design, n = 4096, K = 3 anisotropic
           components, the 512-point grid sharded over the GPUs (strong scaling: 64 per GPU at N = 8).
           cfg2 = Heat-Exchanger grid (Qian n = 64, 624 x 1000 evaluations, sharded by grid row);
           cfg3 = 2-D anisotropic grid on maximin-100 (60 x 1728, by grid row);
           cfg5 = Ground-Vibrations predictive tables (17 sets x 1000 draws, sharded by draw).
step     : one pass of the hot path over this rank's batch -- covariance build, Cholesky,
           solves, log-likelihood for every draw -- followed by the single all-gather of
           the log-likelihoods (RCCL).  Inputs (X, y, parameter matrix) are resident in HBM
           before the timed region.
One rank per GPU.  Under torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE set) this process IS a rank; a plain
`python bench.py --gpus N` with N > 1 starts the N ranks itself as fresh child processes before touching the GPU.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD datasheet; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_FILES = ("r05/pmc_traffic.json", "r05/pmc_traffic64.json", "r05/pmc_traffics1.json", "r04/pmc_traffic.json", "r04/pmc_traffic64.json", "r03/pmc_traffic.json", "r03/pmc_traffic64.json", "r02n/pmc_traffic.json", "r02n/pmc_traffic64.json")   # latest committed rocprofv3 --pmc passes of this bench command


# ----------------------------------------------------------------------------- synthetic inputs
def maximin_lhs(n, d, seed, sweeps=2000):
    """Seeded random Latin hypercube in [0,1]^d improved by maximin column swaps."""
    rng = np.random.default_rng(seed)
    X = np.empty((n, d))
    for k in range(d):
        X[:, k] = (rng.permutation(n) + 0.5) / n

    def near(i):
        dd = ((X - X[i]) ** 2).sum(axis=1)
        dd[i] = np.inf
        return dd.min()

    for _ in range(sweeps):
        a, b = rng.integers(0, n, size=2)
        k = rng.integers(0, d)
        if a == b:
            continue
        before = min(near(a), near(b))
        X[a, k], X[b, k] = X[b, k], X[a, k]
        if min(near(a), near(b)) < before:
            X[a, k], X[b, k] = X[b, k], X[a, k]
    return X


def cfg4_inputs(total_evals, n=4096, d=5, K=3, seed=20140101):
    X = maximin_lhs(n, d, seed)
    y = np.sin(2.0 * np.pi * X).sum(axis=1)
    rng = np.random.default_rng(seed + 1)
    P = np.empty((total_evals, K + K * d))
    for b in range(total_evals):
        w = 0.15 + 0.55 * rng.dirichlet(np.ones(K))           # weights on the simplex, none negligible
        th = np.exp(rng.uniform(math.log(0.5), math.log(50.0), size=(K, d)))
        th[K - 1] = np.maximum(th[K - 1], 20.0)                # roughest component keeps R PD (no nugget)
        P[b] = np.concatenate([w, th.ravel()])
    return X, y, P, K


def cfg2_inputs():
    from ccgp_amd.tables import read_table
    from ccgp_amd import api
    data = os.path.join(ROOT, "tests", "golden", "data")
    _, tr = read_table(os.path.join(data, "qian_train.txt"))
    _, H = read_table(os.path.join(data, "hx_hyperpars_matrix.txt"))
    X, y = tr[:, :4], tr[:, 4]
    N = 1000
    u = api.halton_base2(N)
    G = H.shape[0]
    P = np.empty((G * N, 2 + 2 * 4))
    for g in range(G):
        th1 = api.qigamma(u, H[g, 0], H[g, 1])
        th2 = api.qigamma(u, H[g, 2], H[g, 3])
        blk = P[g * N:(g + 1) * N]
        blk[:, 0], blk[:, 1] = u, 1.0 - u
        blk[:, 2:6] = th1[:, None]
        blk[:, 6:10] = th2[:, None]
    return X, y, P, 2, float(np.var(y, ddof=1))


def cfg3_inputs():
    """BASELINE config 3: maximin-100 design, anisotropic kernel (ANI:399-406), ADV grid semantics
    (60 rows x 1728 Halton nodes, tau = 100), lambda = 4, inverse-gamma scales x16."""
    from ccgp_amd.tables import read_table
    from ccgp_amd import api
    data = os.path.join(ROOT, "tests", "golden", "data")
    _, X = read_table(os.path.join(data, "maximin_100.txt"))
    _, H = read_table(os.path.join(data, "adv_hyperpars_matrix.txt"))
    H = H * np.array([1.0, 16.0, 1.0, 16.0])
    y = (np.sin(2 * X[:, 0]) + np.cos(4 * X[:, 0])) * (np.sin(8 * X[:, 1]) + np.cos(4 * X[:, 1]))   # ANI:338
    N, lam = 1728, 4.0
    u = api.halton_base2(N)
    P = np.empty((H.shape[0] * N, 6))
    for g in range(H.shape[0]):
        th1, th2 = api.qigamma(u, H[g, 0], H[g, 1]), api.qigamma(u, H[g, 2], H[g, 3])
        blk = P[g * N:(g + 1) * N]
        blk[:, 0], blk[:, 1], blk[:, 2], blk[:, 3] = u, 1.0 - u, th1, th2
        blk[:, 4], blk[:, 5] = (1 + lam) * th1, (1 + lam) * th2
    return X, y, P, 2, float(np.var(y, ddof=1))


def cfg5_inputs(S=1000, seed=20140105):
    """BASELINE config 5: every Ground-Vibrations train/test pair (9 of size 50, 8 of size 90),
    S posterior draws synthesised around (p, theta1, theta2) = (0.7, 0.3, 15) with log-normal jitter."""
    from ccgp_amd.tables import read_table
    data = os.path.join(ROOT, "tests", "golden", "data", "gv")
    rng = np.random.default_rng(seed)
    p = 1.0 / (1.0 + np.exp(-(math.log(0.7 / 0.3) + 0.3 * rng.normal(size=S))))
    th1 = 0.3 * np.exp(0.25 * rng.normal(size=S))
    th2 = 15.0 * np.exp(0.25 * rng.normal(size=S))
    P = np.empty((S, 2 + 18))
    P[:, 0], P[:, 1] = p, 1.0 - p
    P[:, 2:11] = th1[:, None]
    P[:, 11:20] = th2[:, None]
    sets = []
    for size, count in ((50, 9), (90, 8)):
        for i in range(1, count + 1):
            _, tr = read_table(os.path.join(data, "train_%d_%d.txt" % (size, i)))
            _, te = read_table(os.path.join(data, "test_%d_%d.txt" % (size, i)))
            sets.append((tr[:, :9], tr[:, 9], te[:, :9]))
    return sets, P


def update_kernel_flops(n, diag_tiles=True):
    """Algorithmic flops of the trailing-update launches for ONE matrix: tile (i,j), i > j,
    needs 2*128^3*j; a diagonal tile needs only its lower half.  diag_tiles=False: only the
    rows below the diagonal (the diagonal tiles are then a separate launch on the second stream)."""
    nt = (n + 127) // 128
    t3 = 2.0 * 128 ** 3
    return sum(j * t3 * ((nt - 1 - j) + (0.5 if diag_tiles else 0.0)) for j in range(1, nt))


def pmc_traffic(kernel, matrices_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (counters cannot
    be read from inside the process; collected separately exactly as the MI355X guide prescribes).
    Only quoted when a pass was collected with the same number of matrices per launch."""
    for name in PMC_TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                rec = json.load(fh)
            if rec.get("matrices_per_launch", 64) == matrices_per_launch:
                return rec["kernels"][kernel]["hbm_bytes_per_launch"], "profiles/" + name
        except Exception:
            continue
    return None, None


# ----------------------------------------------------------------------------- CPU baseline
def effective_cpus():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (a GPU box of the pool
    shows all 256 hardware threads of its host to os.cpu_count() but grants a one-GPU job 16 CPUs' worth of time; rounds 2 - 4
    started one OpenMP thread per VISIBLE thread and read the quota as "6 % parallel efficiency")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        n = min(n, max(1, int(math.ceil(q / int(fh2.read())))))
            break
        except Exception:
            continue
    return n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_compiled_loglik(X, y, P, K, sigma2, mode, tau2, per_core, one_core_evals, in_process_all=True):
    """SURVEY 8(d)'s CPU baseline: the compiled evaluator of oracle/cpu_baseline (covariance build, LAPACK dpotrf,
    two dtrsv, log-likelihood: the same algorithmic work as the GPU path), OpenMP over evaluations with one
    evaluation per core, on every core of this host and on one core.  Bounded sample of the workload's draws.
    in_process_all=False (large n): only the one-core figure; the all-cores figure then comes from cpu_process_sweep."""
    from oracle.cpu_baseline import loader as cpu
    cores = min(cpu.max_threads(), effective_cpus())
    B, t_all = 0, float("nan")
    if in_process_all:
        B = min(P.shape[0], per_core * cores)
        # warm-up: the first second of a fresh OpenMP team runs several times slower (thread start, core wake-up)
        t0 = time.perf_counter()
        cpu.loglik_batch(X, y, K, P[:B], sigma2, mode, tau2, threads=cores)
        if time.perf_counter() - t0 < 1.0:
            while time.perf_counter() - t0 < 1.0:
                cpu.loglik_batch(X, y, K, P[:B], sigma2, mode, tau2, threads=cores)
        t0 = time.perf_counter()
        cpu.loglik_batch(X, y, K, P[:B], sigma2, mode, tau2, threads=cores)
        t_all = time.perf_counter() - t0
    b1 = min(P.shape[0], one_core_evals)
    if not in_process_all:
        cpu.loglik_batch(X, y, K, P[:1], sigma2, mode, tau2, threads=1)     # warm-up of the one core
    t0 = time.perf_counter()
    cpu.loglik_batch(X, y, K, P[:b1], sigma2, mode, tau2, threads=1)
    t_one = time.perf_counter() - t0
    return {"all_cores": B / t_all if in_process_all else None, "one_core": b1 / t_one, "cores": cores, "unit": "evals/s",
            "sample": ("%d evaluations on %d cores in %.2f s, " % (B, cores, t_all) if in_process_all else "") +
                      "%d on one core in %.2f s" % (b1, t_one),
            "lapack": ("built-in C Cholesky (n <= 128: concurrent tiny LAPACK calls serialise inside OpenBLAS)"
                       if X.shape[0] <= 128 else
                       "scipy OpenBLAS dpotrf/dtrsv" if cpu.lapack_bound()
                       else "built-in C Cholesky (no LAPACK found)")}


def cpu_gradient_fd(X, y, P, K, sigma2, per_core=100):
    """The stated CPU baseline of the gradient: the reference has no analytic gradient -- LearnBayes::laplace differences
    logpost numerically (HX:493) -- so the baseline is the compiled evaluator with CENTRAL DIFFERENCES in every parameter
    (2 P likelihood evaluations per gradient), OpenMP over evaluations on every core."""
    from oracle.cpu_baseline import loader as cpu
    cores = min(cpu.max_threads(), effective_cpus())
    B = min(P.shape[0], per_core * cores)
    Pn = P.shape[1]
    h = 1e-4     # relative step: the likelihood itself carries cond(R) eps of rounding, so smaller steps only add noise
    big = np.repeat(P[:B], 2 * Pn, axis=0)
    for j in range(Pn):
        big[2 * j::2 * Pn, j] += h * np.abs(P[:B, j])
        big[2 * j + 1::2 * Pn, j] -= h * np.abs(P[:B, j])
    cpu.loglik_batch(X, y, K, big[:2 * Pn * cores], sigma2, 0, 0.0, threads=cores)      # warm the OpenMP team
    t0 = time.perf_counter()
    ll = cpu.loglik_batch(X, y, K, big, sigma2, 0, 0.0, threads=cores)[0]
    t_all = time.perf_counter() - t0
    ll = np.asarray(ll).reshape(B, Pn, 2)
    g = (ll[:, :, 0] - ll[:, :, 1]) / (2 * h * np.abs(P[:B]))
    return {"all_cores": B / t_all, "cores": cores, "unit": "gradients/s",
            "sample": "%d gradients = %d likelihood evaluations (central differences in %d parameters) on %d cores in %.2f s"
                      % (B, B * 2 * Pn, Pn, cores, t_all)}, g


def cpu_reference_opcount(workload, X, y, P, K, sigma2, mode, tau2, budget_s=8.0):
    """The oracle (numpy restatement of the reference's R operation sequence: materialised U + t(U) + V
    temporaries, LU inverse via solve(), then dmnorm's chol + chol2inv) timed with numpy's BLAS threads: an
    emulation of what R + LAPACK would do per evaluation, not a measurement of R (R is not installed anywhere)."""
    from oracle import ccgp_oracle as orc
    d = X.shape[1]
    done, t0 = 0, time.perf_counter()
    while True:
        w, Th = orc.unpack_params(P[done % P.shape[0]], K, d)
        orc.loglik_general(X, y, w, Th, sigma2, mode, tau2)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s or done >= 100000:
            break
    return {"value": done / el, "unit": "evals/s", "sample": "%d evaluations of %s through oracle.loglik_general in %.1f s"
                                                              % (done, workload, el)}


def cpu_predict_sample(sets, P5, draws_per_core=8):
    """cfg5 on the CPU: the compiled evaluator's predict.post tables for a bounded number of draws of every set."""
    from oracle.cpu_baseline import loader as cpu
    cores = min(cpu.max_threads(), effective_cpus())
    S = min(P5.shape[0], draws_per_core * cores)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:                 # warm-up of the OpenMP team (see cpu_compiled_loglik)
        cpu.predict_batch(sets[0][0], sets[0][1], 2, P5[:S], sets[0][2], 1.0, threads=cores)
    pairs, t_all = 0, 0.0
    for (Xs, ys, Xt) in sets:
        t0 = time.perf_counter()
        cpu.predict_batch(Xs, ys, 2, P5[:S], Xt, 1.0, threads=cores)
        t_all += time.perf_counter() - t0
        pairs += S * Xt.shape[0]
    Xs, ys, Xt = sets[0]
    s1 = min(S, 8)
    t0 = time.perf_counter()
    cpu.predict_batch(Xs, ys, 2, P5[:s1], Xt, 1.0, threads=1)
    t_one = time.perf_counter() - t0
    return {"all_cores": pairs / t_all, "one_core": s1 * Xt.shape[0] / t_one, "cores": cores,
            "unit": "(draw, test point) predictions/s",
            "sample": "%d draws x 17 sets on %d cores in %.2f s; %d draws of set 1 on one core in %.2f s" % (S, cores, t_all, s1, t_one)}


def cpu_logpost_latency(X, y, sigma2, calls=200):
    """What ONE sequential logpost costs on one CPU core, two ways: the oracle (the reference's operation sequence in numpy:
    solve(R) + beta.MLE + dmnorm -- an emulation of an R-level call, not a measurement of R) and the compiled evaluator
    (covariance + Cholesky + two solves, one evaluation per call through ctypes).  For Metro's sequential caller the device
    call is SLOWER than the compiled CPU call: a single n = 64 evaluation is a latency problem; batching is the remedy."""
    from oracle import ccgp_oracle as orc
    from oracle.cpu_baseline import loader as cpu
    theta_t = [math.log(0.3), math.log(15.0), math.log(4.0)]
    orc.logpost(X, theta_t, y, sigma2, "HX", (7, 3, 3, 28))
    t0 = time.perf_counter()
    for _ in range(calls):
        orc.logpost(X, theta_t, y, sigma2, "HX", (7, 3, 3, 28))
    el = (time.perf_counter() - t0) / calls
    d = X.shape[1]
    row = np.concatenate([[0.8, 0.2], np.full(d, 0.3), np.full(d, 15.0)])[None, :]
    cpu.loglik_batch(X, y, 2, row, sigma2, 0, 0.0, threads=1)
    t0 = time.perf_counter()
    for _ in range(10 * calls):
        cpu.loglik_batch(X, y, 2, row, sigma2, 0, 0.0, threads=1)
    el_c = (time.perf_counter() - t0) / (10 * calls)
    return {"us_per_call": 1e6 * el, "unit": "us per oracle.logpost call (numpy, reference operation sequence)", "calls": calls,
            "compiled_us_per_call": 1e6 * el_c,
            "compiled_what": "oracle/cpu_baseline evaluator, ONE evaluation per call on one core (value + beta, no R.Inv; "
                             "ctypes binding included, as for the device figure)"}


def cpu_predict_post_latency(X, y, x_new, sigma2, calls=200):
    """One literal predict.post call on the CPU with the cached terms given (HX:655-673: r = Mixed.corr.vec, then the
    arithmetic with the frame row's R.Inv): the oracle's numpy restatement, one call at a time as `apply` makes them, and
    the same arithmetic compiled (oracle/cpu_baseline: ccgp_cpu_predict_post) on one core."""
    from oracle import ccgp_oracle as orc
    from oracle.cpu_baseline import loader as cpu
    n, d = X.shape
    R = orc.mixed_corr_matrix_iso(X, 0.7, 0.3, 15.0)
    R_inv = orc.solve_inverse(R)
    beta = orc.beta_mle(R_inv, y)
    mf, v1, v2 = orc.factors(R_inv, beta, y)
    t0 = time.perf_counter()
    for _ in range(calls):
        r = orc.mixed_corr_vec_iso(x_new, X, 0.7, 0.3, 15.0)
        want = orc.predict_post_from_factors(r, beta, mf, v1, v2, R_inv, sigma2)
    el = (time.perf_counter() - t0) / calls
    pp = cpu.PredictPost(X, 2, np.concatenate([[0.7, 0.3], np.full(d, 0.3), np.full(d, 15.0)]), beta, mf, v1, v2, R_inv, sigma2)
    got = pp(x_new).copy()
    t0 = time.perf_counter()
    for _ in range(20 * calls):
        pp(x_new)
    el_c = (time.perf_counter() - t0) / (20 * calls)
    return {"us_per_call": 1e6 * el, "cores": 1, "calls": calls,
            "what": "numpy restatement of one R-level predict.post call (oracle/ccgp_oracle.py), cached terms given",
            "compiled_us_per_call": 1e6 * el_c,
            "compiled_what": "the same call compiled (ccgp_cpu_predict_post, one core, ctypes binding included)",
            "compiled_matches_numpy": bool(np.allclose(got, np.ravel(want), rtol=1e-9, atol=1e-12))}


def cpu_process_sweep(X, y, P, K, sigma2, mode, tau2, evals_per_worker=6, budget_s=60.0):
    """All-cores figure for LARGE n as a sweep over (concurrent evaluations c) x (threads per evaluation t), c t = the host's
    hardware threads: c worker processes (oracle/cpu_baseline/cpu_worker.py), each evaluating one draw at a time with t
    threads (OpenMP over the covariance columns, OpenBLAS dpotrf / dtrsv with t threads).  Processes, because scipy's
    pthread OpenBLAS serialises multi-threaded calls coming from several threads of one process; 128 single-threaded
    dpotrf of 134 MB matrices side by side are memory-bound (round 4 measured 6 % parallel efficiency that way).
    Wall time from "go" to the last worker's "done"; workers load and warm up before."""
    import subprocess
    import tempfile
    H = effective_cpus()
    ts = [t for t in (64, 32, 16, 8, 4, 2, 1) if t <= H]      # fewest processes first: the cheap configurations before the budget runs out
    worker = os.path.join(ROOT, "oracle", "cpu_baseline", "cpu_worker.py")
    rows, t_start = [], time.perf_counter()
    with tempfile.TemporaryDirectory() as td:
        npz = os.path.join(td, "in.npz")
        np.savez(npz, X=X, y=y, P=P[:max(2 * H, 8)], K=K, sigma2=sigma2, mode=mode, tau2=tau2)
        for t in ts:
            c = max(H // t, 1)
            if time.perf_counter() - t_start > budget_s:
                break
            procs = [subprocess.Popen([sys.executable, worker, npz, str(i * evals_per_worker), str(evals_per_worker), str(t)],
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                     for i in range(c)]
            ok = all(p.stdout.readline().strip() == "ready" for p in procs)
            t0 = time.perf_counter()
            for p in procs:
                try:
                    p.stdin.write("go\n")
                    p.stdin.flush()
                except Exception:
                    ok = False
            outs = [p.stdout.readline().split() for p in procs]
            wall = time.perf_counter() - t0
            for p in procs:
                p.wait()
            ok = ok and all(len(o) == 3 and o[0] == "done" for o in outs)
            if ok:
                rows.append({"concurrent": c, "threads_each": t, "evals": c * evals_per_worker, "seconds": wall,
                             "evals_per_s": c * evals_per_worker / wall})
    return rows


def cpu_baseline(workload, X, y, P, K, sigma2, mode, tau2):
    big = X.shape[0] > 1000
    c = cpu_compiled_loglik(X, y, P, K, sigma2, mode, tau2, per_core=2 if big else 2000, one_core_evals=2 if big else 4000,
                            in_process_all=not big)
    ref = cpu_reference_opcount(workload, X, y, P, K, sigma2, mode, tau2)
    sweep = []
    if big:
        # large n: the host's threads split between concurrent evaluations and threads per evaluation, best split reported
        sweep = cpu_process_sweep(X, y, P, K, sigma2, mode, tau2)
        best = dict(max(sweep, key=lambda r: r["evals_per_s"]), how="worker processes") if sweep else \
            {"concurrent": 1, "threads_each": 1, "evals_per_s": c["one_core"], "how": "one core (the sweep produced nothing)"}
    else:
        best = {"concurrent": c["cores"], "threads_each": 1, "evals_per_s": c["all_cores"], "how": "OpenMP threads of one process"}
    return {"value": best["evals_per_s"], "unit": "evals/s", "cores": c["cores"], "kind": "port",
            "sample": "%s workload (n=%d): all cores = %d concurrent evaluations x %d threads each (%s)%s; %s" % (
                workload, X.shape[0], best["concurrent"], best["threads_each"], best["how"],
                ", best of the sweep " + ", ".join("%dx%d: %.1f/s" % (r["concurrent"], r["threads_each"], r["evals_per_s"]) for r in sweep) if sweep else "",
                c["sample"]),
            "all_cores": best["evals_per_s"], "all_cores_config": best, "sweep": sweep, "one_core": c["one_core"],
            "parallel_efficiency": best["evals_per_s"] / (c["one_core"] * c["cores"]) if c["one_core"] > 0 else None,
            "model": cpu_model(), "lapack": c["lapack"],
            "what": "compiled evaluator oracle/cpu_baseline (covariance build + dpotrf + 2 dtrsv per evaluation); all_cores = the "
                    "best split of the host's hardware threads between concurrent evaluations and threads per evaluation",
            "reference_opcount": ref}


# ----------------------------------------------------------------------------- launcher
def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as FRESH child processes
    (torch.distributed.run, one per GPU) and relay rank 0's JSON line and the children's return code.  Runs before
    anything in this process has touched the GPU (no torch import yet), and never exec()s -- a process that has
    initialised the GPU must not be replaced on this pool."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ----------------------------------------------------------------------------- main
def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg4", choices=["cfg4", "cfg2", "cfg3", "cfg5"],
                    help="cfg4 = BASELINE config 4 (the headline); cfg2 / cfg3 = the hyperprior grids sharded by grid row; "
                         "cfg5 = Ground-Vibrations predictive tables sharded by posterior draw")
    ap.add_argument("--evals-total", type=int, default=512,
                    help="cfg4: size of the hyperparameter grid, sharded over the GPUs (BASELINE config 4: 512)")
    ap.add_argument("--evals-per-gpu", type=int, default=0,
                    help="cfg4: fixed per-GPU slice instead of a fixed grid (weak scaling; experiments only)")
    ap.add_argument("--no-tail-strips", action="store_true", help="every update tile whole (ccgp_set_option; measurements)")
    ap.add_argument("--half-tail-strips", action="store_true", help="half-width tail strips only, as in rounds 2 - 3 (ccgp_set_option; measurements)")
    ap.add_argument("--no-fuse-diag", action="store_true", help="separate diag_kernel launches (ccgp_set_option; measurements)")
    ap.add_argument("--sched", type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help="blocked sweep: 0 = one launch per phase and block column (rounds 1 - 4); 1 = dataflow tile scheduler, two "
                         "workgroups per CU; 2 = one per CU; 3 = the library's choice by chunk size (default) (ccgp_set_option)")
    ap.add_argument("--sched-policy", type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help="scheduler bit mask (default 3): bit 0 = a CU's second workgroup only takes tiles while a backlog exists; "
                         "bit 1 = XCD-local synchronisation (0: agent-scope fences)")
    ap.add_argument("--small-grid16", action="store_true",
                    help="64 < n <= 104 on the 16 x 16 thread grid of rounds 1 - 3 instead of one wave per matrix (ccgp_set_option; measurements)")
    ap.add_argument("--ws-limit-gib", type=float, default=0.0,
                    help="cap the device scratch (ccgp_set_workspace_limit) to force multi-chunk batches")
    ap.add_argument("--n", type=int, default=4096, help="cfg4 matrix order (parity/debug runs only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path).  gloo: rehearsal of the N > 1 code path on "
                         "a box with fewer GPUs than ranks (ranks share devices, results gathered through host memory)")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import ccgp_amd  # noqa: F401
    from ccgp_amd import api, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    host_gather = world > 1 and args.backend == "gloo"
    gdev = torch.device("cpu") if host_gather else dev
    f64 = dict(dtype=torch.float64, device=dev)
    ctx = dict(args=args, torch=torch, dist=dist, api=api, shard=shard, world=world, rank=rank, local=local, dev=dev,
               host_gather=host_gather, gdev=gdev, f64=f64)
    if args.workload == "cfg5":
        run_predict_workload(ctx)
    else:
        run_loglik_workload(ctx)
    if world > 1:
        dist.destroy_process_group()


def col_major(torch, a, **kw):
    return torch.tensor(np.asfortranarray(a).ravel(order="F"), **kw)


def fence(torch, dist, world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def warm_up(torch, dist, world, step, warmup, local_compute=None):
    """The W untimed warm-up steps of the contract.  The small-n workloads (a step is 4 - 9 ms) first run their LOCAL
    compute (no collective, so the ranks need not agree on a count) for a quarter of a second: after W = 1 or 2 such
    steps the clocks have not ramped yet and the first timed steps come out up to 20 % slow (profiles/r03: cfg3 7.70 ms
    against 6.32 ms for the same ten steps measured right after)."""
    if local_compute is not None:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            local_compute()
            torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    fence(torch, dist, world)


def max_over_ranks(torch, dist, world, gdev, seconds):
    if world == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=gdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_loglik_workload(c):
    """cfg4 (headline), cfg2, cfg3: a grid of likelihood evaluations, sharded over the ranks (the hyperprior grids by
    grid ROW, so that a row's mean over its Halton nodes stays rank-local, HX:574), ONE all-gather per step."""
    args, torch, dist, api, shard = c["args"], c["torch"], c["dist"], c["api"], c["shard"]
    world, rank, local, dev, f64 = c["world"], c["rank"], c["local"], c["dev"], c["f64"]
    host_gather, gdev = c["host_gather"], c["gdev"]

    mode, tau2 = api.MEAN_PROFILE_BETA, 0.0
    if args.workload == "cfg4":
        weak = args.evals_per_gpu > 0
        total = args.evals_per_gpu * world if weak else args.evals_total
        X, y, P, K = cfg4_inputs(total, n=args.n)
        sigma2 = 1.0
        lo, hi = shard.shard_bounds(total, rank, world)
        unit_rows, units = 1, total
        wl_name = "cfg4: synthetic maximin-LHS 5-D design n=%d, K=3 anisotropic components, %d-point hyperparameter grid%s" % (
            args.n, total, " (%d per GPU)" % args.evals_per_gpu if weak else " sharded over the GPUs")
    else:
        if args.workload == "cfg2":
            X, y, P, K, sigma2 = cfg2_inputs()
            tau2, unit_rows = 50.0 ** 2, 1000
            wl_name = "cfg2: Heat-Exchanger grid, Qian n=64, 624 rows x 1000 Halton nodes, sharded by grid row"
        else:
            X, y, P, K, sigma2 = cfg3_inputs()
            tau2, unit_rows = 100.0 ** 2, 1728
            wl_name = "cfg3: 2-D anisotropic grid on maximin-100, 60 rows x 1728 Halton nodes, sharded by grid row"
        mode = api.MEAN_ZERO_PLUS_TAU2
        total = P.shape[0]
        units = total // unit_rows
        glo, ghi = shard.shard_bounds(units, rank, world)     # strong scaling: the grid is fixed
        lo, hi = glo * unit_rows, ghi * unit_rows
    n, d = X.shape
    B = hi - lo

    # inputs resident in HBM (column-major, as the C ABI takes them)
    dX, dy = col_major(torch, X, **f64), torch.tensor(y, **f64)
    dP = col_major(torch, P[lo:hi], **f64)
    d_ll = torch.empty(B, **f64)
    d_beta = torch.empty(B, **f64)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    # the all-gather's buffers, allocated once: slots of whole grid rows (unit_rows evaluations each)
    gat = shard.RowGatherer(units, tail=(unit_rows,), dtype=torch.float64, device=gdev) if world > 1 else None
    h_ll = torch.empty(B, dtype=torch.float64).pin_memory() if host_gather else None

    # CPU legs first (rank 0, N = 1 only), so that the GPU legs that follow are one contiguous stretch of device work
    cpu_main, cpu_sec, sec_in = None, {}, None
    if rank == 0 and world == 1 and args.workload == "cfg4" and not args.no_secondary:
        sec_in = {"cfg2": cfg2_inputs(), "cfg3": cfg3_inputs(), "cfg5": cfg5_inputs()}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_main = cpu_baseline(args.workload, X, y, P, K, sigma2, mode, tau2)
        if sec_in is not None:
            X2, y2, P2, K2, s22 = sec_in["cfg2"]
            cpu_sec["cfg2"] = cpu_compiled_loglik(X2, y2, P2, K2, s22, api.MEAN_ZERO_PLUS_TAU2, 2500.0, 2000, 4000)
            X3, y3, P3, K3, s23 = sec_in["cfg3"]
            cpu_sec["cfg3"] = cpu_compiled_loglik(X3, y3, P3, K3, s23, api.MEAN_ZERO_PLUS_TAU2, 1e4, 800, 2000)
            cpu_sec["cfg5"] = cpu_predict_sample(*sec_in["cfg5"])
            cpu_sec["logpost"] = cpu_logpost_latency(X2, y2, s22)
            cpu_sec["gradient"], cpu_sec["gradient_fd"] = cpu_gradient_fd(X2, y2, P2, K2, s22)
            gvX, gvy, gvXt = sec_in["cfg5"][0][-1]
            cpu_sec["predict_post"] = {"Qian n=64": cpu_predict_post_latency(X2, y2, X2[0] * 0.97, 10.0),
                                       "GV n=%d" % gvX.shape[0]: cpu_predict_post_latency(gvX, gvy, gvXt[0], 10.0)}
            if args.n == 4096:
                cpu_sec["cfg4_predict"] = cpu_predict_n4096(X, y, K, P[:CFG4_PREDICT_DRAWS], sigma2)
                cpu_sec["cfg4_gradient"], cpu_sec["cfg4_gradient_fd"] = cpu_gradient_n4096(X, y, K, P[0], sigma2)

    h = api.Handle(local)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.half_tail_strips:
        h.set_option(api.OPT_TAIL_STRIPS, 2)
    if args.no_tail_strips:
        h.set_option(api.OPT_TAIL_STRIPS, 0)
    if args.no_fuse_diag:
        h.set_option(api.OPT_FUSE_DIAG, 0)
    if args.small_grid16:
        h.set_option(api.OPT_SMALL_GRID16, 1)
    if args.sched >= 0:
        h.set_option(api.OPT_SCHED, args.sched)
    if args.sched_policy >= 0:
        h.set_option(api.OPT_SCHED_POLICY, args.sched_policy)
    if args.ws_limit_gib > 0:
        h.set_workspace_limit(int(args.ws_limit_gib * 2 ** 30))
    h.reserve(n, d, K, max(B, 1), 0)

    gathered = [None]

    def step():
        h.loglik_batch_dev(dX, n, d, dy, K, dP, B, sigma2, mode, tau2, d_ll, d_beta, d_st)
        if world > 1:
            # the one collective of the path: RCCL all-gather of the per-rank slices into buffers allocated once
            # (shard.RowGatherer); in the gloo rehearsal the slice goes through pinned host memory, which synchronises
            if host_gather:
                h_ll.copy_(d_ll)
                gathered[0] = gat.gather(h_ll.view(-1, unit_rows))
            else:
                gathered[0] = gat.gather(d_ll.view(-1, unit_rows))

    warm_up(torch, dist, world, step, args.warmup,
            (lambda: h.loglik_batch_dev(dX, n, d, dy, K, dP, B, sigma2, mode, tau2, d_ll, d_beta, d_st)) if n <= 128 else None)
    # HIP events inside the timed region only around the launches of the roofline kernel (two event records
    # per launch are not free: 96 launch groups per step); the per-kernel breakdown comes from one extra,
    # untimed step afterwards, and a second region of the same K steps WITHOUT any event gives the cost of the
    # in-region events (notiming_ms_per_step).
    if n <= 128:
        main_id = "fused"
    else:   # which kind of sweep the library runs for this chunk: one untimed probe step with every group timed
        h.enable_timing(True)
        step()
        fence(torch, dist, world)
        main_id = "sweep" if h.get_timing()["sweep"][1] else "update"
        h.enable_timing(False)
    h.enable_timing(True, only=[main_id])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence(torch, dist, world)
    elapsed = time.perf_counter() - t0
    timing = h.get_timing()
    h.enable_timing(False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence(torch, dist, world)
    elapsed_notiming = time.perf_counter() - t0
    h.enable_timing(True)
    step()
    fence(torch, dist, world)
    breakdown = h.get_timing()
    h.enable_timing(False)
    elapsed = max_over_ranks(torch, dist, world, gdev, elapsed)
    elapsed_notiming = max_over_ranks(torch, dist, world, gdev, elapsed_notiming)
    if world > 1:
        # every rank now holds every shard's log-likelihoods: check the gather against the local slice
        full = gathered[0].reshape(-1)
        assert full.shape[0] == total and torch.equal(full[lo:hi].to(dev), d_ll), "all-gather mismatch"
    bad = int((d_st != 0).sum().item())
    finite = bool(torch.isfinite(d_ll).all().item())
    # what was just timed, against CPU potrf + forward solves of the same draws (tests/golden/make_cfg4_digest.py)
    digest_ok = None
    if args.workload == "cfg4" and n == 4096 and total <= 512:
        try:
            with open(os.path.join(ROOT, "tests", "golden", "cfg4_loglik_512.json")) as fh:
                ref = json.load(fh)
            got_ll, got_b = d_ll.cpu().numpy(), d_beta.cpu().numpy()
            digest_ok = bool(np.allclose(got_ll, np.array(ref["loglik"])[lo:hi], rtol=1e-9, atol=0.0) and
                             np.allclose(got_b, np.array(ref["beta"])[lo:hi], rtol=1e-7, atol=1e-10))
        except FileNotFoundError:
            digest_ok = None
        if digest_ok is False and not os.environ.get("CCGP_BENCH_ALLOW_MISMATCH"):   # (timing ablations set it)
            raise SystemExit("bench.py: rank %d log-likelihoods differ from tests/golden/cfg4_loglik_512.json" % rank)

    if rank == 0:
        value = total * args.steps / elapsed
        out = {
            "metric": "GP log-lik evals/sec (n x n fp64, batched over hyperpar grid)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak" if (args.workload == "cfg4" and args.evals_per_gpu > 0) else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_name, "n": n, "d": d, "K": K, "evals_total": total,
                       "evals_per_gpu": B, "parallelism": "grid sharded over %d GPU(s), one all-gather%s" % (
                           world, " (gloo rehearsal, ranks share devices)" if host_gather else ""),
                       "failed_evals": bad, "all_finite": finite,
                       "matches_cpu_potrf_digest": digest_ok},
            "notiming_ms_per_step": 1e3 * elapsed_notiming / args.steps,
            "notiming_note": "the same K steps again with no HIP event in the region (value / ms_per_step are the "
                             "region WITH the events around the roofline kernel's launches)",
            "kernel_ms_per_step": {k: v[0] for k, v in breakdown.items() if v[1]},
            "kernel_ms_per_step_source": "one extra step with every launch group timed (outside the timed region)",
        }
        if n > 128 and main_id == "sweep":
            # the scheduled sweep: ONE launch per chunk holds every update, diagonal and panel-solve tile, so the kernel is
            # priced with the factorisation's whole algorithmic count n^3 / 3 per matrix ("only the n^3/3 potrf
            # flops count") -- NOT comparable with the launch-per-phase rows of earlier rounds, whose roofline kernel was the
            # update alone (0.95 n^3 / 3 in 0.86 of the time)
            sw_ms, sw_launches = timing["sweep"]
            sw_ms = sw_ms or float("nan")
            flops = (n ** 3 / 3.0) * B * args.steps
            ach = flops / (sw_ms * 1e-3) / 1e12 if sw_ms > 0 else 0.0
            traffic, traffic_src = pmc_traffic("chol_sched", B) if n == 4096 else (None, None)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)",
                               "traffic_source": traffic_src,
                               "kernel": "chol_sched_kernel (blocked Cholesky sweep of the chunk as one persistent launch: update, "
                                         "diagonal and panel-solve tiles from dependency-driven queues, f64 MFMA)",
                               "launches": sw_launches,
                               "avg_launch_ms": sw_ms / max(sw_launches, 1),
                               "flops_per_launch": flops / max(sw_launches, 1),
                               "flops_note": "n^3/3 per matrix (whole factorisation); the update tiles alone are %.3e per matrix" % update_kernel_flops(n)}
        elif n > 128:
            upd_ms, upd_launches = timing["update"]
            upd_ms = upd_ms or float("nan")
            flops = update_kernel_flops(n) * B * args.steps   # this rank's launches
            ach = flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
            traffic, traffic_src = pmc_traffic("chol_update", B) if n == 4096 else (None, None)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)",
                               "traffic_source": traffic_src,
                               "kernel": "chol_update*_kernel (blocked-Cholesky trailing update, f64 MFMA)",
                               "launches": upd_launches,
                               "avg_launch_ms": upd_ms / max(upd_launches, 1),
                               "flops_per_launch": flops / max(upd_launches, 1)}
            out["whole_job_tflops"] = (n ** 3 / 3.0) * total * args.steps / elapsed / 1e12
        else:
            fused_ms, fl = timing["fused"]
            flops = (n ** 3 / 3.0) * B * args.steps
            ach = flops / (fused_ms * 1e-3) / 1e12 if fused_ms > 0 else 0.0
            out["roofline"] = small_roofline(ach, fl, fused_ms)
        if cpu_main is not None:
            out["cpu_baseline"] = cpu_main
        if sec_in is not None:
            out["secondary"] = secondary_items(c, h, sec_in, cpu_sec, X, y, P, K, sigma2)
        print(json.dumps(out), flush=True)
    h.close()


def small_roofline(ach, launches, fused_ms):
    return {"bound": "valu-issue", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
            "kernel": "small_reg_kernel (register-resident fused evaluator: 8 P + 20 bytes of HBM per evaluation, so "
                      "neither HBM nor -- for the VALU form -- MFMA bounds it; it is fp64 VALU-issue bound, and n^3/3 flop "
                      "per evaluation is priced against the fp64 peak, which is the same 78.6 TFLOP/s for vector and "
                      "matrix instructions on this chip; see DESIGN.md K2)",
            "launches": launches, "avg_launch_ms": fused_ms / max(launches, 1)}


def timed_passes(torch, fn, passes=5):
    """Untimed passes for a tenth of a second (clocks), then `passes` timed ones (wall clock around a device
    synchronise): seconds per pass."""
    t0 = time.perf_counter()
    while True:
        fn()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 0.1:
            break
    t1 = time.perf_counter()
    for _ in range(passes):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t1) / passes


CFG4_PREDICT_DRAWS, CFG4_PREDICT_SITES = 16, 128


def secondary_items(c, h, sec_in, cpu_sec, X, y, P, K, sigma2):
    """BASELINE configs 2, 3, 5 and the factor-set prediction on the same GPU (N = 1 only), each as the kernel-level
    figure (inputs resident in HBM) AND -- round 3 -- end to end through the host-pointer entry point the reference's
    caller would bind (`choose.hyperpars` -> ccgp_grid_marginal, `prediction` -> ccgp_predict_batch)."""
    torch, api, dev, f64 = c["torch"], c["api"], c["dev"], c["f64"]
    args = c["args"]
    items = []
    from ccgp_amd.tables import read_table
    data = os.path.join(ROOT, "tests", "golden", "data")
    # config 2: the Heat-Exchanger grid
    X2, y2, P2, K2, s22 = sec_in["cfg2"]
    dX2, dy2, dP2 = col_major(torch, X2, **f64), torch.tensor(y2, **f64), col_major(torch, P2, **f64)
    B2 = P2.shape[0]
    o1, o2 = torch.empty(B2, **f64), torch.empty(B2, **f64)
    o3 = torch.zeros(B2, dtype=torch.int32, device=dev)
    el2 = timed_passes(torch, lambda: h.loglik_batch_dev(dX2, 64, 4, dy2, K2, dP2, B2, s22, api.MEAN_ZERO_PLUS_TAU2,
                                                         2500.0, o1, o2, o3))
    _, H2 = read_table(os.path.join(data, "hx_hyperpars_matrix.txt"))
    res2 = [None]

    def grid2():
        res2[0] = h.grid_marginal(X2, y2, s22, H2, 1000, 50.0, True)
    e2e2 = timed_passes(torch, grid2)
    items.append({"workload": "cfg2: Heat-Exchanger grid, Qian n=64, 624 x 1000 evals",
                  "value": B2 / el2, "unit": "evals/s", "ms_per_pass": 1e3 * el2,
                  "failed_evals": int((o3 != 0).sum().item()),
                  "end_to_end": {"call": "Handle.grid_marginal = ccgp_grid_marginal (host pointers; what choose.hyperpars "
                                         "HX:584-595 binds): G x 4 hyperparameters in, G values + argmax out, the G x N "
                                         "draws built on the device",
                                 "ms_per_call": 1e3 * e2e2, "evals_per_s": B2 / e2e2,
                                 "over_kernel": e2e2 / el2, "argmax_row_1based": int(res2[0][1]) + 1}})
    # config 3: 2-D anisotropic grid on maximin-100 (60 x 1728 evaluations at n = 100)
    X3, y3, P3, K3, s23 = sec_in["cfg3"]
    dX3, dy3, dP3 = col_major(torch, X3, **f64), torch.tensor(y3, **f64), col_major(torch, P3, **f64)
    B3 = P3.shape[0]
    q1, q2 = torch.empty(B3, **f64), torch.empty(B3, **f64)
    q3 = torch.zeros(B3, dtype=torch.int32, device=dev)
    el3 = timed_passes(torch, lambda: h.loglik_batch_dev(dX3, 100, 2, dy3, K3, dP3, B3, s23, api.MEAN_ZERO_PLUS_TAU2,
                                                         1e4, q1, q2, q3))
    _, H3 = read_table(os.path.join(data, "adv_hyperpars_matrix.txt"))
    H3 = H3 * np.array([1.0, 16.0, 1.0, 16.0])
    e2e3 = timed_passes(torch, lambda: h.grid_marginal(X3, y3, s23, H3, 1728, 100.0, False, aniso_lambda=4.0))
    items.append({"workload": "cfg3: 2-D anisotropic grid, maximin-100, 60 x 1728 evals",
                  "value": B3 / el3, "unit": "evals/s", "ms_per_pass": 1e3 * el3,
                  "failed_evals": int((q3 != 0).sum().item()),
                  "end_to_end": {"call": "Handle.grid_marginal (host pointers)", "ms_per_call": 1e3 * e2e3,
                                 "evals_per_s": B3 / e2e3, "over_kernel": e2e3 / el3}})
    # config 5: Ground-Vibrations predictive mean/variance tables, all 17 train/test pairs
    sets, P5 = sec_in["cfg5"]
    S5 = P5.shape[0]
    dP5 = col_major(torch, P5, **f64)
    dsets, pairs = [], 0
    for (Xs, ys, Xt) in sets:
        m5 = Xt.shape[0]
        pairs += S5 * m5
        dsets.append((col_major(torch, Xs, **f64), torch.tensor(ys, **f64), col_major(torch, Xt, **f64), Xs.shape[0], m5,
                      torch.empty(S5 * m5, **f64), torch.empty(S5 * m5, **f64), torch.empty(S5, **f64),
                      torch.zeros(S5, dtype=torch.int32, device=dev)))

    def pass5():
        for (a1, a2, a3, n5, m5, o_m, o_v, o_b, o_s) in dsets:
            h.predict_batch_dev(a1, n5, 9, a2, 2, dP5, S5, a3, m5, float(1.0), o_m, o_v, o_b, o_s)

    # column-major on the host already, as R hands its matrices over (no layout conversion inside the timed call)
    sets_f, P5f = [(np.asfortranarray(a), b, np.asfortranarray(c_)) for (a, b, c_) in sets], np.asfortranarray(P5)

    def pass5_host():
        for (Xs, ys, Xt) in sets_f:
            h.predict_batch(Xs, ys, 2, P5f, Xt, 1.0)
    el5 = timed_passes(torch, pass5)
    e2e5 = timed_passes(torch, pass5_host)
    items.append({"workload": "cfg5: Ground-Vibrations predictive tables, 17 sets x 1000 draws x (150|110) test points",
                  "value": pairs / el5, "unit": "(draw, test point) predictions/s",
                  "ms_per_pass": 1e3 * el5,
                  "failed_draws": int(sum(int((t[8] != 0).sum().item()) for t in dsets)),
                  "end_to_end": {"call": "Handle.predict_batch = ccgp_predict_batch (host pointers; the table prediction() "
                                         "HX:686-693 averages), one call per train/test pair",
                                 "ms_per_pass": 1e3 * e2e5, "predictions_per_s": pairs / e2e5, "over_kernel": e2e5 / el5}})
    for entry, key in zip(items, ("cfg2", "cfg3", "cfg5")):
        entry["cpu"] = cpu_sec.get(key)
    # the SEQUENTIAL caller: Metro evaluates one proposal per logpost call (HX:512) and keeps R.Inv (HX:520) -- latency,
    # not throughput.  Host-pointer ccgp_logpost on the Qian set, with and without the n x n inverse coming back.
    theta_t = np.array([math.log(0.3), math.log(15.0), math.log(4.0)])
    pars = np.array([7.0, 3.0, 3.0, 28.0])
    calls = 200
    lat = {}
    for want in (True, False):
        h.logpost(X2, y2, s22, api.PRIOR_INVGAMMA, theta_t, pars, want_Rinv=want)
        t1 = time.perf_counter()
        for _ in range(calls):
            h.logpost(X2, y2, s22, api.PRIOR_INVGAMMA, theta_t, pars, want_Rinv=want)
        lat[want] = (time.perf_counter() - t1) / calls
    items.append({"workload": "logpost latency: Qian n=64, ONE draw per call through the host-pointer ccgp_logpost (what Metro "
                              "calls once per proposal, HX:505-512)",
                  "value": 1.0 / lat[True], "unit": "logpost calls/s (sequential, with R.Inv)",
                  "us_per_call_with_Rinv": 1e6 * lat[True], "us_per_call_value_only": 1e6 * lat[False], "calls": calls,
                  "cpu": cpu_sec.get("logpost")})
    # The LITERAL prediction path of an unchanged script: predict.post once per (draw, test site) (HX:688: S x m calls per
    # test set), each on one frame row with its n x n R.Inv (HX:655-663).  Round 3 bound it as two host-pointer calls
    # (Mixed.corr.vec, then the arithmetic of HX:667-670: six pageable copies and two synchronisations); round 4 as ONE
    # (ccgp_predict_post: one pinned copy each way), and r/ccgp.R's compare.GP / prediction wrappers bypass it altogether
    # with one ccgp_predict_batch per test set (the cfg5 line above).  us per call, next to the batched cost per
    # (draw, site) pair and to the other per-draw helpers factors.frame calls (factors HX:641, beta.MLE HX:458).
    from ccgp_amd.rsurface import CombinedGP, pack_iso
    lit = []
    gv_X, gv_y, gv_Xt = sets[-1] if sets[-1][0].shape[0] == 90 else max(sets, key=lambda t: t[0].shape[0])
    for name, (Xl, yl, xl, dl) in (("Qian n=64", (X2, y2, X2[:1] * 0.97, 4)), ("GV n=%d" % gv_X.shape[0], (gv_X, gv_y, gv_Xt[:1], 9))):
        nl = Xl.shape[0]
        gp = CombinedGP("GV", handle=h)
        row = gp.factors_frame_from_draws([(0.7, 0.3, 15.0)], Xl, 10.0, yl)[0]
        prow = pack_iso(0.7, 0.3, 15.0, dl)
        beta_l, mf_l, v1_l, v2_l = row[3], row[4:4 + nl], row[4 + nl:4 + 2 * nl], row[4 + 2 * nl]
        Rinv_l = np.asfortranarray(row[5 + 2 * nl:].reshape(nl, nl, order="F"))

        def pair():
            r = h.mixed_corr_cross(xl, Xl, 2, prow)
            return h.predict_from_factors(r, beta_l, mf_l, v1_l, v2_l, Rinv_l, 10.0)

        def fused():
            return h.predict_post(xl, Xl, 2, prow, beta_l, mf_l, v1_l, v2_l, Rinv_l, 10.0)
        timings = {}
        for key, fn in (("two_calls", pair), ("one_call", fused), ("factors", lambda: h.factors(Rinv_l, beta_l, yl)),
                        ("beta_mle", lambda: h.beta_mle(Rinv_l, yl))):
            fn()
            t1 = time.perf_counter()
            for _ in range(calls):
                fn()
            timings[key] = 1e6 * (time.perf_counter() - t1) / calls
        same = bool(np.array_equal(np.array(pair()), np.array(fused())))
        lit.append({"design": name, "us_per_predict_post_two_calls": timings["two_calls"],
                    "us_per_predict_post_one_call": timings["one_call"], "us_per_factors": timings["factors"],
                    "us_per_beta_mle": timings["beta_mle"], "identical": same})
    items.append({"workload": "literal predict.post latency (host pointers, one frame row with its R.Inv per call; HX:655-673)",
                  "value": 1e6 / lit[-1]["us_per_predict_post_one_call"], "unit": "predict.post calls/s (sequential, GV n=90)",
                  "per_design": lit, "calls": calls,
                  "batched_ns_per_prediction_kernel": 1e9 * el5 / pairs, "batched_ns_per_prediction_end_to_end": 1e9 * e2e5 / pairs,
                  "test_set_of_1000x150_literal_s": 1e-6 * lit[-1]["us_per_predict_post_one_call"] * 150000,
                  "test_set_of_1000x150_batched_ms": 1e3 * e2e5 / len(sets), "cpu": cpu_sec.get("predict_post")})
    # the north_star's "+ gradient": ccgp_loglik_grad_batch (host pointers) on the Heat-Exchanger design, next to the
    # same call without the gradient
    Bg = 65536
    Pg = np.asfortranarray(P2[:Bg])      # column-major on the host already, as R hands a matrix over
    h.loglik_grad_batch(X2, y2, K2, Pg, s22)
    t1 = time.perf_counter()
    _, _, gg, stg = h.loglik_grad_batch(X2, y2, K2, Pg, s22)
    t_g = time.perf_counter() - t1
    h.loglik_batch(X2, y2, K2, Pg, s22)
    t1 = time.perf_counter()
    h.loglik_batch(X2, y2, K2, Pg, s22)
    t_l = time.perf_counter() - t1
    items.append({"workload": "gradient: Qian n=64, %d draws, log-likelihood + d/d(w, theta) through ccgp_loglik_grad_batch "
                              "(host pointers, PCIe-inclusive)" % Bg,
                  "value": Bg / t_g, "unit": "gradients/s", "ms_per_call": 1e3 * t_g,
                  "ms_same_call_without_gradient": 1e3 * t_l, "failed": int((stg != 0).sum()),
                  "all_finite": bool(np.isfinite(gg).all()), "cpu": cpu_sec.get("gradient"),
                  "rel_dev_from_cpu_central_differences_median_p99": (
                      [float(v) for v in np.percentile(
                          np.abs(gg[:cpu_sec["gradient_fd"].shape[0]] - cpu_sec["gradient_fd"]) /
                          (np.abs(cpu_sec["gradient_fd"]) + 1e-3 * np.abs(cpu_sec["gradient_fd"]).max(axis=1, keepdims=True)), [50, 99])]
                      if "gradient_fd" in cpu_sec else None)})
    # prediction at a second test set from factors kept in HBM, against re-factorising
    if args.workload == "cfg4" and X.shape[0] == 4096:
        Sf, mf_ = CFG4_PREDICT_DRAWS, CFG4_PREDICT_SITES
        Xt = cfg4_predict_sites(X.shape[1])
        t1 = time.perf_counter()
        fs = h.factor_batch(X, y, K, P[:Sf], sigma2)
        t_fac = time.perf_counter() - t1
        fs.predict(Xt)
        t1 = time.perf_counter()
        m_a, v_a = fs.predict(Xt)
        t_keep = time.perf_counter() - t1
        h.predict_batch(X, y, K, P[:Sf], Xt, sigma2)
        t1 = time.perf_counter()
        m_b, v_b, _, _ = h.predict_batch(X, y, K, P[:Sf], Xt, sigma2)
        t_full = time.perf_counter() - t1
        items.append({
            "workload": "cfg4 prediction: n=4096, %d draws, %d test sites" % (Sf, mf_),
            "value": Sf * mf_ / t_keep, "unit": "(draw, test point) predictions/s from a kept factor set",
            "ms_from_factorset": 1e3 * t_keep, "ms_refactorising": 1e3 * t_full, "ms_factor_batch": 1e3 * t_fac,
            "factorset_bytes": fs.nbytes, "identical": bool(np.array_equal(m_a, m_b) and np.array_equal(v_a, v_b)),
            "cpu": cpu_sec.get("cfg4_predict")})
        fs.free()
        # analytic gradient at the headline size (the reference differences logpost numerically: LearnBayes::laplace, HX:493)
        Pg = np.asfortranarray(P[:Sf])
        h.loglik_grad_batch(X, y, K, Pg, sigma2)
        t1 = time.perf_counter()
        gll, _, gg, gst = h.loglik_grad_batch(X, y, K, Pg, sigma2)
        t_grad = time.perf_counter() - t1
        h.loglik_batch(X, y, K, Pg, sigma2)
        t1 = time.perf_counter()
        pll, _, _ = h.loglik_batch(X, y, K, Pg, sigma2)
        t_ll = time.perf_counter() - t1
        item = {"workload": "cfg4 gradient: n=4096, %d draws, log-likelihood + d/d(w, theta) (%d parameters) through "
                            "ccgp_loglik_grad_batch (host pointers)" % (Sf, Pg.shape[1]),
                "value": Sf / t_grad, "unit": "gradients/s", "ms_per_call": 1e3 * t_grad,
                "ms_same_call_without_gradient": 1e3 * t_ll, "failed": int(np.count_nonzero(gst)),
                "loglik_identical_to_plain_call": bool(np.array_equal(gll, pll)),
                "cpu": cpu_sec.get("cfg4_gradient")}
        fd = cpu_sec.get("cfg4_gradient_fd")
        if fd is not None:
            item["rel_dev_from_cpu_central_differences_max"] = float(
                np.max(np.abs(gg[0] - fd) / (np.abs(fd) + 1e-3 * np.abs(fd).max())))
        items.append(item)
    items.append(end_to_end_fit_item(h))
    return items


def end_to_end_fit_item(h):
    """The whole fit of the Ground-Vibrations script on training sample 1 (GV driver block GV:688-695: start c(1,1,0), N.max
    5000, samp.size 1000, alpha.geweke 0.5, batch 20; sigma2 given) -- laplace + Metro + the 1000-draw x 150-site prediction
    tables -- with the sampler one logpost per proposal (the script's loop) and with blocks of m = 4 proposals per device call
    (fit.Metro(speculate = 4) = what r/ccgp.R's Metro does through ccgp_R_metro_steps).  Same seed: the two chains must be the
    same chain."""
    import ccgp_amd  # noqa: F401
    from ccgp_amd import fit
    from ccgp_amd.rsurface import CombinedGP
    from ccgp_amd.tables import read_table
    data = os.path.join(ROOT, "tests", "golden", "data", "gv")
    _, tr = read_table(os.path.join(data, "train_50_1.txt"))
    _, te = read_table(os.path.join(data, "test_50_1.txt"))
    D, y, Dt, yt = tr[:, :9], tr[:, 9], te[:, :9], te[:, 9]
    gp = CombinedGP("GV", handle=h)
    out = {"workload": "end-to-end fit: Ground Vibrations sample 1 (n = 50, d = 9): laplace + Metro (1000 retained draws, Geweke "
                       "stopping rule) + prediction at 150 sites, sigma2 = 10.2494 (the value behind the reference's recorded table)"}
    runs = {}
    for name, m in (("sequential", 0), ("blocks_of_4", 4), ("blocks_of_6", 6), ("blocks_of_8", 8)):
        calls = {"logpost": 0, "rows": 0, "s": 0.0}
        inner = fit.logpost_batch

        def counted(gp_, D_, rows, y_, s2_, pars_=None):
            t0 = time.perf_counter()
            r = inner(gp_, D_, rows, y_, s2_, pars_)
            calls["s"] += time.perf_counter() - t0
            calls["logpost"] += 1
            calls["rows"] += np.atleast_2d(rows).shape[0]
            return r
        fit.logpost_batch = counted
        try:
            t0 = time.perf_counter()
            table = fit.Combined_GP_fit(gp, D, y, Dt, [1.0, 1.0, 0.0], 5000, 1000, 0.5, 20, alpha=0.05, y_new=yt, sigma2=10.2494,
                                        rng=20140101, speculate=m)
            el = time.perf_counter() - t0
        finally:
            fit.logpost_batch = inner
        ch = table["chain"]
        runs[name] = {"seconds_per_fit": el, "device_calls": calls["logpost"], "candidates_evaluated": calls["rows"],
                      "seconds_in_device_calls": calls["s"], "host_share": 1.0 - calls["s"] / el,
                      "proposals": ch["proposals"], "accepted": ch["accepted"], "rmspe": fit.comparison_summary(table)["rmspe"]}
        runs[name + "_draws"] = table["draws"]
    seq = runs.pop("sequential_draws")
    out["same_chain"] = bool(all(np.array_equal(seq, runs.pop(k)) for k in [k for k in runs if k.endswith("_draws")]))
    out.update(runs)
    out["note"] = ("device_calls counts laplace's (one candidate each: Nelder-Mead and the Hessian stencil) and the sampler's; "
                   "host_share = Python host layer (Geweke test, accept / reject walk, prediction summaries) and is what an R host "
                   "would spend in R")
    return out


def cfg4_predict_sites(d):
    return np.random.default_rng(5).random((CFG4_PREDICT_SITES, d))


def cpu_predict_n4096(X, y, K, P, sigma2):
    """The compiled CPU evaluator on the factor-set workload (n = 4096, 16 draws x 128 sites: one draw per core)."""
    from oracle.cpu_baseline import loader as cpu
    cores = min(cpu.max_threads(), effective_cpus())
    Xt = cfg4_predict_sites(X.shape[1])
    t0 = time.perf_counter()
    cpu.predict_batch(X, y, K, P, Xt, sigma2, threads=cores)
    el = time.perf_counter() - t0
    return {"all_cores": P.shape[0] * Xt.shape[0] / el, "cores": cores, "unit": "(draw, test point) predictions/s",
            "ms": 1e3 * el, "sample": "%d draws x %d sites, re-factorising (covariance + dpotrf + %d dtrsv per draw), one draw "
                                      "per core on %d cores in %.2f s" % (P.shape[0], Xt.shape[0], Xt.shape[0] + 2, cores, el)}


def cpu_gradient_n4096(X, y, K, row, sigma2):
    """ONE gradient at n = 4096 by central differences of the compiled evaluator: 2 P evaluations, one per core."""
    from oracle.cpu_baseline import loader as cpu
    cores = min(cpu.max_threads(), effective_cpus())
    Pn = row.size
    hstep = 1e-5
    big = np.repeat(row[None], 2 * Pn, axis=0)
    for j in range(Pn):
        big[2 * j, j] += hstep * abs(row[j])
        big[2 * j + 1, j] -= hstep * abs(row[j])
    t0 = time.perf_counter()
    ll = np.asarray(cpu.loglik_batch(X, y, K, big, sigma2, 0, 0.0, threads=cores)[0]).reshape(Pn, 2)
    el = time.perf_counter() - t0
    g = (ll[:, 0] - ll[:, 1]) / (2 * hstep * np.abs(row))
    return {"all_cores": 1.0 / el, "cores": cores, "unit": "gradients/s",
            "sample": "1 gradient = %d likelihood evaluations (central differences in %d parameters), one per core on %d "
                      "cores in %.2f s" % (2 * Pn, Pn, cores, el)}, g


def run_predict_workload(c):
    """cfg5 (BASELINE config 5): the (posterior draw x test point) predictive mean / variance tables of all 17
    Ground-Vibrations train/test pairs, sharded over the ranks BY DRAW; ONE all-gather per step collects the
    S x (sum of m) x 2 table (every rank then holds what prediction() averages, HX:688-693)."""
    args, torch, dist, api, shard = c["args"], c["torch"], c["dist"], c["api"], c["shard"]
    world, rank, local, dev, f64 = c["world"], c["rank"], c["local"], c["dev"], c["f64"]
    host_gather, gdev = c["host_gather"], c["gdev"]
    sets, P5 = cfg5_inputs()
    S = P5.shape[0]
    lo, hi = shard.shard_bounds(S, rank, world)
    Sl = hi - lo
    M = sum(t[2].shape[0] for t in sets)
    dP = col_major(torch, P5[lo:hi], **f64)
    # a device table S_local x m column-major is a row-major [m, S_local] block: all 17 sets' mean and variance tables
    # stacked give [2 M, S_local]; the draws are the gathered (last) dimension
    gat = shard.RowGatherer(S, lead=(2 * M,), dtype=torch.float64, device=gdev)
    direct = (not host_gather) and Sl == gat.slot      # compute straight into the send buffer
    table = gat.send if direct else torch.empty((2 * M, Sl), **f64)
    h_table = torch.empty((2 * M, Sl), dtype=torch.float64).pin_memory() if host_gather else None
    d_beta = torch.empty(Sl, **f64)
    d_st = torch.zeros(Sl, dtype=torch.int32, device=dev)
    dsets, off = [], 0
    for (Xs, ys, Xt) in sets:
        m = Xt.shape[0]
        dsets.append((col_major(torch, Xs, **f64), torch.tensor(ys, **f64), col_major(torch, Xt, **f64), Xs.shape[0], m,
                      table[off:off + m], table[M + off:M + off + m]))
        off += m
    cpu5 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu5 = cpu_predict_sample(sets, P5)
    h = api.Handle(local)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    full = [None]
    bad = [0]

    def compute():
        for (a1, a2, a3, n5, m5, o_m, o_v) in dsets:
            h.predict_batch_dev(a1, n5, 9, a2, 2, dP, Sl, a3, m5, 1.0, o_m, o_v, d_beta, d_st)

    def step():
        compute()
        if host_gather:
            h_table.copy_(table)
            full[0] = gat.gather(h_table)
        else:
            full[0] = gat.gather(None if direct else table)

    warm_up(torch, dist, world, step, args.warmup, compute)
    h.enable_timing(True, only=["fused"])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence(torch, dist, world)
    elapsed = time.perf_counter() - t0
    timing = h.get_timing()
    h.enable_timing(False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence(torch, dist, world)
    elapsed_notiming = time.perf_counter() - t0
    elapsed = max_over_ranks(torch, dist, world, gdev, elapsed)
    elapsed_notiming = max_over_ranks(torch, dist, world, gdev, elapsed_notiming)
    bad[0] = int((d_st != 0).sum().item())
    got = full[0]
    assert tuple(got.shape) == (2 * M, S), got.shape
    assert torch.equal(got[:, lo:hi].to(dev), table), "all-gather mismatch"
    finite = bool(torch.isfinite(got).all().item())
    if rank == 0:
        pairs = S * M
        fused_ms, fl = timing["fused"]
        # useful flop per draw and set: n^3/3 (factor) + m n^2 (forward substitution of the m cross-correlation rows)
        flops = sum((t[0].shape[0] ** 3 / 3.0 + t[2].shape[0] * t[0].shape[0] ** 2) for t in sets) * Sl * args.steps
        ach = flops / (fused_ms * 1e-3) / 1e12 if fused_ms > 0 else 0.0
        out = {"metric": "GP predictive mean/variance, (draw x test point) predictions/sec (BASELINE config 5)",
               "value": pairs * args.steps / elapsed, "unit": "(draw, test point) predictions/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "cfg5: Ground-Vibrations predictive tables, 17 train/test pairs (n = 50 | 90, d = 9), "
                                      "%d synthetic posterior draws x %d test points in total, sharded by draw" % (S, M),
                          "draws_total": S, "draws_per_gpu": Sl, "test_points_total": M,
                          "parallelism": "draws sharded over %d GPU(s), one all-gather of the S x M x 2 table%s" % (
                              world, " (gloo rehearsal, ranks share devices)" if host_gather else ""),
                          "gathered_bytes": int(2 * M * S * 8), "failed_draws": bad[0], "all_finite": finite},
               "notiming_ms_per_step": 1e3 * elapsed_notiming / args.steps,
               "roofline": small_roofline(ach, fl, fused_ms)}
        if cpu5 is not None:
            out["cpu_baseline"] = {"value": cpu5["all_cores"], "unit": cpu5["unit"], "cores": cpu5["cores"], "kind": "port",
                                   "sample": cpu5["sample"], "one_core": cpu5["one_core"], "model": cpu_model()}
        print(json.dumps(out), flush=True)
    h.close()


if __name__ == "__main__":
    main()
