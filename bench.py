#!/usr/bin/env python3
"""Benchmark of the GP log-likelihood hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload cfg4|cfg2] [--evals-per-gpu E]

metric   : GP log-lik evals/sec (n x n fp64, batched over the hyperparameter grid)
workload : cfg4 (default) = BASELINE config 4, the configuration the north_star's MFMA
           target is quoted on: synthetic 5-D design, n = 4096, K = 3 anisotropic
           components; 64 evaluations per GPU (weak scaling: 8 GPUs = the 512-point grid).
           cfg2 = Heat-Exchanger grid (Qian n = 64, 624 x 1000 evaluations, sharded by row).
step     : one pass of the hot path over this rank's batch -- covariance build, Cholesky,
           solves, log-likelihood for every draw -- followed by the single all-gather of
           the log-likelihoods (RCCL).  Inputs (X, y, parameter matrix) are resident in HBM
           before the timed region.
One rank per GPU; for N > 1 launch through torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE).
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
PMC_TRAFFIC_FILES = ("r02n/pmc_traffic.json", "r02n/pmc_traffic64.json")   # latest committed rocprofv3 --pmc passes of this bench command


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
def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_compiled_loglik(X, y, P, K, sigma2, mode, tau2, per_core, one_core_evals):
    """SURVEY 8(d)'s CPU baseline: the compiled evaluator of oracle/cpu_baseline (covariance build, LAPACK dpotrf,
    two dtrsv, log-likelihood: the same algorithmic work as the GPU path), OpenMP over evaluations with one
    evaluation per core, on every core of this host and on one core.  Bounded sample of the workload's draws."""
    from oracle.cpu_baseline import loader as cpu
    cores = cpu.max_threads()
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
    t0 = time.perf_counter()
    cpu.loglik_batch(X, y, K, P[:b1], sigma2, mode, tau2, threads=1)
    t_one = time.perf_counter() - t0
    return {"all_cores": B / t_all, "one_core": b1 / t_one, "cores": cores, "unit": "evals/s",
            "sample": "%d evaluations on %d cores in %.2f s, %d on one core in %.2f s" % (B, cores, t_all, b1, t_one),
            "lapack": ("built-in C Cholesky (n <= 128: concurrent tiny LAPACK calls serialise inside OpenBLAS)"
                       if X.shape[0] <= 128 else
                       "scipy OpenBLAS dpotrf/dtrsv, single-threaded per evaluation" if cpu.lapack_bound()
                       else "built-in C Cholesky (no LAPACK found)")}


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
    cores = cpu.max_threads()
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


def cpu_baseline(workload, X, y, P, K, sigma2, mode, tau2):
    big = X.shape[0] > 1000
    c = cpu_compiled_loglik(X, y, P, K, sigma2, mode, tau2, per_core=2 if big else 2000, one_core_evals=2 if big else 4000)
    ref = cpu_reference_opcount(workload, X, y, P, K, sigma2, mode, tau2)
    return {"value": c["all_cores"], "unit": "evals/s", "cores": c["cores"], "kind": "port",
            "sample": "%s workload (n=%d): %s" % (workload, X.shape[0], c["sample"]),
            "all_cores": c["all_cores"], "one_core": c["one_core"], "model": cpu_model(), "lapack": c["lapack"],
            "what": "compiled evaluator oracle/cpu_baseline (covariance build + dpotrf + 2 dtrsv per evaluation, OpenMP "
                    "over evaluations, one evaluation per core)",
            "reference_opcount": ref}


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg4", choices=["cfg4", "cfg2"])
    ap.add_argument("--evals-total", type=int, default=512,
                    help="cfg4: size of the hyperparameter grid, sharded over the GPUs (BASELINE config 4: 512)")
    ap.add_argument("--evals-per-gpu", type=int, default=0,
                    help="cfg4: fixed per-GPU slice instead of a fixed grid (weak scaling; experiments only)")
    ap.add_argument("--strips", type=int, default=0, choices=[0, 1, 2],
                    help="pin the update kernel's column-strip count (ccgp_set_option; 0 = per-launch choice)")
    ap.add_argument("--no-tail-strips", action="store_true", help="every update tile whole (ccgp_set_option; measurements)")
    ap.add_argument("--no-fuse-diag", action="store_true", help="separate diag_kernel launches (ccgp_set_option; measurements)")
    ap.add_argument("--ws-limit-gib", type=float, default=0.0,
                    help="cap the device scratch (ccgp_set_workspace_limit) to force multi-chunk batches")
    ap.add_argument("--n", type=int, default=4096, help="cfg4 matrix order (parity/debug runs only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path).  gloo: rehearsal of the N > 1 code path on "
                         "a box with fewer GPUs than ranks (ranks share devices, results gathered through host memory)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ccgp_amd  # noqa: F401
    from ccgp_amd import api, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)"
                         % (args.gpus, world))
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

    mode, tau2 = api.MEAN_PROFILE_BETA, 0.0
    if args.workload == "cfg4":
        weak = args.evals_per_gpu > 0
        total = args.evals_per_gpu * world if weak else args.evals_total
        X, y, P, K = cfg4_inputs(total, n=args.n)
        sigma2 = 1.0
        lo, hi = shard.shard_bounds(total, rank, world)
        wl_name = "cfg4: synthetic maximin-LHS 5-D design n=%d, K=3 anisotropic components, %d-point hyperparameter grid%s" % (
            args.n, total, " (%d per GPU)" % args.evals_per_gpu if weak else " sharded over the GPUs")
    else:
        X, y, P, K, sigma2 = cfg2_inputs()
        mode, tau2 = api.MEAN_ZERO_PLUS_TAU2, 50.0 ** 2
        total = P.shape[0]
        G = total // 1000
        glo, ghi = shard.shard_bounds(G, rank, world)     # shard by grid ROW (strong scaling: fixed grid)
        lo, hi = glo * 1000, ghi * 1000
        wl_name = "cfg2: Heat-Exchanger grid, Qian n=64, 624 rows x 1000 Halton nodes"
    n, d = X.shape
    B = hi - lo

    # inputs resident in HBM (column-major, as the C ABI takes them)
    f64 = dict(dtype=torch.float64, device=dev)
    dX = torch.tensor(np.asfortranarray(X).ravel(order="F"), **f64)
    dy = torch.tensor(y, **f64)
    dP = torch.tensor(np.asfortranarray(P[lo:hi]).ravel(order="F"), **f64)
    d_ll = torch.empty(B, **f64)
    d_beta = torch.empty(B, **f64)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    sizes = shard.shard_sizes(total if args.workload == "cfg4" else total, world)
    if args.workload == "cfg2":
        sizes = [1000 * s for s in shard.shard_sizes(total // 1000, world)]
    gdev = dict(dtype=torch.float64, device="cpu" if host_gather else dev)

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

    h = api.Handle(local)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.strips:
        h.set_option(api.OPT_UPDATE_STRIPS, args.strips)
    if args.no_tail_strips:
        h.set_option(api.OPT_TAIL_STRIPS, 0)
    if args.no_fuse_diag:
        h.set_option(api.OPT_FUSE_DIAG, 0)
    if args.ws_limit_gib > 0:
        h.set_workspace_limit(int(args.ws_limit_gib * 2 ** 30))
    h.reserve(n, d, K, max(B, 1), 0)

    gathered = [None]

    def step():
        h.loglik_batch_dev(dX, n, d, dy, K, dP, B, sigma2, mode, tau2, d_ll, d_beta, d_st)
        if world > 1:
            # the one collective of the path (shard.all_gather_rows: RCCL all-gather of the per-rank slices; in the
            # gloo rehearsal the slice goes through host memory, which synchronises)
            gathered[0] = shard.all_gather_rows(d_ll.cpu() if host_gather else d_ll, total)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # HIP events inside the timed region only around the launches of the roofline kernel (two event records
    # per launch are not free: 96 launch groups per step); the per-kernel breakdown comes from one extra,
    # untimed step afterwards.
    main_id = "update" if args.workload == "cfg4" else "fused"
    if os.environ.get("CCGP_BENCH_NOTIMING") is None:
        h.enable_timing(True, only=[main_id])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    timing = h.get_timing()
    h.enable_timing(True)
    step()
    fence()
    breakdown = h.get_timing()
    h.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], **gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank now holds every shard's log-likelihoods: check the gather against the local slice
        full = gathered[0]
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
        if digest_ok is False:
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
            "kernel_ms_per_step": {k: v[0] for k, v in breakdown.items() if v[1]},
            "kernel_ms_per_step_source": "one extra step with every launch group timed (outside the timed region)",
        }
        if args.workload == "cfg4":
            upd_ms, upd_launches = timing["update"]
            upd_ms = upd_ms or float("nan")
            split = timing.get("update_diag", (0.0, 0))[1] > 0        # diagonal tiles launched separately
            flops = update_kernel_flops(n, diag_tiles=not split) * B * args.steps   # this rank's launches
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
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                               "kernel": "small_reg_kernel (register-resident fused evaluator: no MFMA and 8 P + 20 bytes of "
                                         "HBM per evaluation; fp64 VALU-issue bound -- PMC: VALU busy 72 % of SIMD time -- so "
                                         "n^3/3 flop per evaluation is priced against the fp64 peak, which is the same "
                                         "78.6 TFLOP/s for vector and matrix instructions on this chip; see DESIGN.md)",
                               "launches": fl, "avg_launch_ms": fused_ms / max(fl, 1)}
        if cpu_main is not None:
            out["cpu_baseline"] = cpu_main
        if sec_in is not None:
            # secondary line item: the Heat-Exchanger grid (BASELINE config 2) on the same GPU
            X2, y2, P2, K2, s22 = sec_in["cfg2"]
            dX2 = torch.tensor(np.asfortranarray(X2).ravel(order="F"), **f64)
            dy2 = torch.tensor(y2, **f64)
            dP2 = torch.tensor(np.asfortranarray(P2).ravel(order="F"), **f64)
            B2 = P2.shape[0]
            o1, o2 = torch.empty(B2, **f64), torch.empty(B2, **f64)
            o3 = torch.zeros(B2, dtype=torch.int32, device=dev)
            for it in range(3):
                if it == 1:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                h.loglik_batch_dev(dX2, 64, 4, dy2, K2, dP2, B2, s22, api.MEAN_ZERO_PLUS_TAU2, 2500.0, o1, o2, o3)
            torch.cuda.synchronize()
            el2 = (time.perf_counter() - t1) / 2
            out["secondary"] = [{"workload": "cfg2: Heat-Exchanger grid, Qian n=64, 624 x 1000 evals",
                                 "value": B2 / el2, "unit": "evals/s", "ms_per_pass": 1e3 * el2,
                                 "failed_evals": int((o3 != 0).sum().item())}]
            # config 3: 2-D anisotropic grid on maximin-100 (60 x 1728 evaluations at n = 100)
            X3, y3, P3, K3, s23 = sec_in["cfg3"]
            dX3 = torch.tensor(np.asfortranarray(X3).ravel(order="F"), **f64)
            dy3 = torch.tensor(y3, **f64)
            dP3 = torch.tensor(np.asfortranarray(P3).ravel(order="F"), **f64)
            B3 = P3.shape[0]
            q1, q2 = torch.empty(B3, **f64), torch.empty(B3, **f64)
            q3 = torch.zeros(B3, dtype=torch.int32, device=dev)
            for it in range(3):
                if it == 1:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                h.loglik_batch_dev(dX3, 100, 2, dy3, K3, dP3, B3, s23, api.MEAN_ZERO_PLUS_TAU2, 1e4, q1, q2, q3)
            torch.cuda.synchronize()
            el3 = (time.perf_counter() - t1) / 2
            out["secondary"].append({"workload": "cfg3: 2-D anisotropic grid, maximin-100, 60 x 1728 evals",
                                     "value": B3 / el3, "unit": "evals/s", "ms_per_pass": 1e3 * el3,
                                     "failed_evals": int((q3 != 0).sum().item())})
            # config 5: Ground-Vibrations predictive mean/variance tables, all 17 train/test pairs
            sets, P5 = sec_in["cfg5"]
            S5 = P5.shape[0]
            dP5 = torch.tensor(np.asfortranarray(P5).ravel(order="F"), **f64)
            dsets, pairs = [], 0
            for (Xs, ys, Xt) in sets:
                m5 = Xt.shape[0]
                pairs += S5 * m5
                dsets.append((torch.tensor(np.asfortranarray(Xs).ravel(order="F"), **f64), torch.tensor(ys, **f64),
                              torch.tensor(np.asfortranarray(Xt).ravel(order="F"), **f64), Xs.shape[0], m5,
                              torch.empty(S5 * m5, **f64), torch.empty(S5 * m5, **f64), torch.empty(S5, **f64),
                              torch.zeros(S5, dtype=torch.int32, device=dev)))
            for it in range(3):
                if it == 1:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                for (a1, a2, a3, n5, m5, o_m, o_v, o_b, o_s) in dsets:
                    h.predict_batch_dev(a1, n5, 9, a2, 2, dP5, S5, a3, m5, float(1.0), o_m, o_v, o_b, o_s)
            torch.cuda.synchronize()
            el5 = (time.perf_counter() - t1) / 2
            out["secondary"].append({"workload": "cfg5: Ground-Vibrations predictive tables, 17 sets x 1000 draws x (150|110) test points",
                                     "value": pairs / el5, "unit": "(draw, test point) predictions/s",
                                     "ms_per_pass": 1e3 * el5,
                                     "failed_draws": int(sum(int((t[8] != 0).sum().item()) for t in dsets))})
            for entry, key in zip(out["secondary"], ("cfg2", "cfg3", "cfg5")):
                entry["cpu"] = cpu_sec.get(key)
            # prediction at a second test set from factors kept in HBM, against re-factorising
            if args.workload == "cfg4" and n == 4096:
                Sf, mf_ = 16, 128
                Xt = np.random.default_rng(5).random((mf_, d))
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
                out["secondary"].append({
                    "workload": "cfg4 prediction: n=4096, %d draws, %d test sites" % (Sf, mf_),
                    "value": Sf * mf_ / t_keep, "unit": "(draw, test point) predictions/s from a kept factor set",
                    "ms_from_factorset": 1e3 * t_keep, "ms_refactorising": 1e3 * t_full, "ms_factor_batch": 1e3 * t_fac,
                    "factorset_bytes": fs.nbytes, "identical": bool(np.array_equal(m_a, m_b) and np.array_equal(v_a, v_b)),
                    "cpu": None})
                fs.free()
        print(json.dumps(out))
    h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
