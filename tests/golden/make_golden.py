#!/usr/bin/env python3
"""Generate tests/golden/*.json from the oracle on the reference's bundled designs.

The reference ships no golden vectors and R is not installed here, so
these fixtures are produced by oracle/ccgp_oracle.py -- the numpy restatement of the R
code -- on fixed inputs: the data tables copied into tests/golden/data/ plus the
explicit hyperparameter draws below.  They pin (a) the oracle against regressions and
(b) the HIP path on the GPU box, where neither /root/reference nor R exists.

Run from the repo root:  python tests/golden/make_golden.py   (about 5 minutes, CPU).
"""
import json
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import ccgp_amd  # noqa: E402
from ccgp_amd.tables import read_table  # noqa: E402
from oracle import ccgp_oracle as orc  # noqa: E402

DATA = os.path.join(HERE, "data")


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as fh:
        json.dump(obj, fh, indent=0, separators=(",", ":"))
    print("wrote", name)


def arr(a):
    return np.asarray(a, dtype=np.float64).tolist()


def matrix_digest(M):
    M = np.asarray(M)
    n = M.shape[0]
    idx = [(0, 0), (1, 0), (n - 1, 0), (n // 2, n // 3), (n - 1, n - 2), (n - 1, n - 1)]
    return dict(sum=float(M.sum()), trace=float(np.trace(M)), fro=float(np.linalg.norm(M)),
                entries=[[i, j, float(M[i, j])] for i, j in idx])


# ----------------------------------------------------------------------------- Heat exchanger
def make_hx():
    _, tr = read_table(os.path.join(DATA, "qian_train.txt"))
    _, te = read_table(os.path.join(DATA, "qian_test.txt"))
    D, y = tr[:, :4], tr[:, 4]
    Dt = te[:, :4]
    sigma2s = [float(np.var(y, ddof=1)), 10.0]
    draws = [(0.80, 0.30, 15.0), (0.70, 0.50, 25.0), (0.90, 0.20, 40.0), (0.65, 1.00, 10.0),
             (0.75, 0.45, 28.0), (0.85, 0.15, 60.0), (0.60, 0.80, 12.0), (0.95, 0.35, 20.0)]
    t1p, t2p = (7.0, 3.0), (3.0, 28.0)  # HX:774-775
    out = dict(sigma2=sigma2s, draws=arr(draws), theta1_pars=t1p, theta2_pars=t2p, cases=[])
    for s2 in sigma2s:
        for (p, t1, t2) in draws:
            theta_t = [math.log(t1), math.log(t2), math.log(p / (1 - p))]
            lp = orc.logpost(D, theta_t, y, s2, "HX", (*t1p, *t2p))
            R = orc.mixed_corr_matrix_iso(D, p, t1, t2)
            ll1 = orc.cond_like_log(D, y, p, t1, t2, s2, 50.0)
            out["cases"].append(dict(sigma2=s2, draw=[p, t1, t2], theta_t=theta_t, val=lp["val"],
                                     beta=lp["beta"], loglik=lp["log_like"], cond_like_log=ll1,
                                     R=matrix_digest(R), R_inv=matrix_digest(lp["R_inv"])))
    s2 = sigma2s[0]
    mean, var, betas = orc.predict_table(D, y, draws, Dt, s2)
    out["predict"] = dict(sigma2=s2, mean=arr(mean), var=arr(var), beta=arr(betas))
    # corr.vec / corr.matrix known answers
    out["corr_vec_iso"] = dict(theta=0.45, x=arr(Dt[0]), r=arr(orc.corr_vec_iso(Dt[0], D, 0.45)))
    th = [0.3, 1.1, 2.0, 0.7]
    out["corr_matrix_general"] = dict(theta=th, R=matrix_digest(orc.corr_matrix(D, th)))
    # hyperprior grid, all 624 rows (HX:584-595), N = 1000, tau = 50
    _, hyper = read_table(os.path.join(DATA, "hx_hyperpars_matrix.txt"))
    t0 = time.time()
    arg, vals = orc.choose_hyperpars(D, y, hyper, s2, N=1000, tau=50.0, take_log=True)
    print("HX grid: %d rows in %.1fs, argmax row %d" % (hyper.shape[0], time.time() - t0, arg))
    out["grid"] = dict(sigma2=s2, N=1000, tau=50.0, take_log=True, values=arr(vals), which_max=arg,
                       row0_logs=arr(orc.likeli_hyperpars(D, y, hyper[0, :2], hyper[0, 2:], s2, 1000, 50.0,
                                                          return_logs=True)))
    dump("hx_golden.json", out)


# ----------------------------------------------------------------------------- 2-D anisotropic
def make_ani():
    _, D = read_table(os.path.join(DATA, "maximin_100.txt"))
    y = np.array([orc.test_function_2d(a, b, 4) for a, b in D])
    sigma2s = [float(np.var(y, ddof=1)), 1.0]
    draws = [(0.80, 0.8, 1.5, 6.0), (0.70, 1.2, 0.9, 3.0), (0.90, 0.5, 2.0, 10.0), (0.60, 2.0, 2.5, 2.0)]
    u = np.linspace(-1.0, 1.0, 5)
    Dt = np.array([[a, b] for b in u for a in u])
    out = dict(function_code=4, y=arr(y), sigma2=sigma2s, draws=arr(draws), Xtest=arr(Dt), cases=[])
    for s2 in sigma2s:
        for (p, t1, t2, lam) in draws:
            theta_t = [math.log(t1), math.log(t2), math.log(p / (1 - p)), math.log(lam)]
            lp = orc.logpost(D, theta_t, y, s2, "ANI")
            R = orc.mixed_corr_matrix_aniso(D, p, t1, t2, lam)
            out["cases"].append(dict(sigma2=s2, draw=[p, t1, t2, lam], theta_t=theta_t, val=lp["val"],
                                     beta=lp["beta"], loglik=lp["log_like"], R=matrix_digest(R),
                                     R_inv=matrix_digest(lp["R_inv"])))
    s2 = sigma2s[0]
    mean, var, betas = orc.predict_table(D, y, draws, Dt, s2, aniso=True)
    out["predict"] = dict(sigma2=s2, mean=arr(mean), var=arr(var), beta=arr(betas))
    # BASELINE config 3: ADV grid semantics (60 x 1728, tau 100, no log) on the anisotropic kernel,
    # lambda fixed at 4 -- a composition no reference script runs as is.
    # The bundled scale parameters were tuned for ADV's 14 points: on these 100 points they give
    # cond(Sigma) up to 7e15 (R's chol would stop or return noise), so config 3 multiplies the two
    # inverse-gamma SCALE columns by 16, which keeps cond(Sigma) <= ~1e8 over the whole grid.
    _, hyper = read_table(os.path.join(DATA, "adv_hyperpars_matrix.txt"))
    b_scale = 16.0
    hyper = hyper * np.array([1.0, b_scale, 1.0, b_scale])
    N, tau, lam = 1728, 100.0, 4.0
    uq = orc.runif_halton(N)
    vals = np.empty(hyper.shape[0])
    logs0 = None
    t0 = time.time()
    for g in range(hyper.shape[0]):
        th1 = orc.qigamma(uq, hyper[g, 0], hyper[g, 1])
        th2 = orc.qigamma(uq, hyper[g, 2], hyper[g, 3])
        logs = np.empty(N)
        for j in range(N):
            row = orc.params_from_aniso(uq[j], th1[j], th2[j], lam)
            w, Th = orc.unpack_params(row, 2, 2)
            logs[j] = orc.loglik_general(D, y, w, Th, s2, orc.MEAN_ZERO_PLUS_TAU2, tau * tau)[0]
        mx = logs.max()
        vals[g] = math.exp(mx + math.log(np.mean(np.exp(logs - mx))))
        if g == 0:
            logs0 = logs
    print("ANI grid: %.1fs" % (time.time() - t0))
    out["grid"] = dict(sigma2=s2, N=N, tau=tau, take_log=False, aniso_lambda=lam, b_scale=b_scale, values=arr(vals),
                       which_max=int(np.argmax(vals)), row0_logs=arr(logs0))
    dump("ani_golden.json", out)


# ----------------------------------------------------------------------------- ADV (14 points)
def make_adv():
    _, D = read_table(os.path.join(DATA, "maximin_14.txt"))
    y = np.array([orc.test_function_2d(a, b, 3) for a, b in D])   # ADV:934-936, func = 3
    s2 = float(np.var(y, ddof=1))
    _, hyper = read_table(os.path.join(DATA, "adv_hyperpars_matrix.txt"))
    arg, vals = orc.choose_hyperpars(D, y, hyper, s2, N=1728, tau=100.0, take_log=False)
    out = dict(function_code=3, y=arr(y), sigma2=s2,
               grid=dict(N=1728, tau=100.0, take_log=False, values=arr(vals), which_max=arg), cases=[])
    for theta_t in ([0.0, 0.5, 1.0], [0.3, 1.2, 0.4], [-0.5, 2.0, 2.0]):   # ADV:926 start first
        lp = orc.logpost(D, theta_t, y, s2, "ADV", tuple(hyper[arg]))
        out["cases"].append(dict(theta_t=theta_t, prior_pars=arr(hyper[arg]), val=lp["val"], beta=lp["beta"],
                                 loglik=lp["log_like"], like=lp["like"], R_inv=matrix_digest(lp["R_inv"])))
    dump("adv_golden.json", out)


# ----------------------------------------------------------------------------- Ground vibrations
def make_gv():
    out = dict(sets=[])
    draws = [(0.70, 0.30, 15.0), (0.75, 0.25, 20.0), (0.65, 0.40, 12.0), (0.80, 0.20, 30.0),
             (0.72, 0.35, 18.0), (0.68, 0.28, 25.0), (0.85, 0.15, 40.0), (0.60, 0.50, 10.0)]
    for size in (50, 90):
        _, tr = read_table(os.path.join(DATA, "gv", "train_%d_1.txt" % size))
        _, te = read_table(os.path.join(DATA, "gv", "test_%d_1.txt" % size))
        D, y, Dt = tr[:, :9], tr[:, 9], te[:, :9]
        s2 = float(np.var(y, ddof=1))
        mean, var, betas = orc.predict_table(D, y, draws, Dt, s2)
        cases = []
        for (p, t1, t2) in draws[:4]:
            theta_t = [math.log(t1), math.log(t2), math.log(p / (1 - p))]
            lp = orc.logpost(D, theta_t, y, s2, "GV")
            cases.append(dict(theta_t=theta_t, val=lp["val"], beta=lp["beta"], loglik=lp["log_like"],
                              R_inv=matrix_digest(lp["R_inv"])))
        out["sets"].append(dict(size=size, sigma2=s2, draws=arr(draws), mean=arr(mean), var=arr(var),
                                beta=arr(betas), cases=cases))
    dump("gv_golden.json", out)


def make_gv_all():
    """Every Ground-Vibrations train/test pair the reference ships (9 of size 50, 8 of size 90 -- BASELINE config 5
    runs all of them; gv_golden.json above holds sample 1 of each size in full): 4 draws x every 5th test site."""
    draws = [(0.70, 0.30, 15.0), (0.80, 0.20, 30.0), (0.62, 0.45, 11.0), (0.93, 0.24, 19.0)]
    out = dict(draws=arr(draws), site_step=5, sets=[])
    for size, count in ((50, 9), (90, 8)):
        for i in range(1, count + 1):
            _, tr = read_table(os.path.join(DATA, "gv", "train_%d_%d.txt" % (size, i)))
            _, te = read_table(os.path.join(DATA, "gv", "test_%d_%d.txt" % (size, i)))
            D, y, Dt = tr[:, :9], tr[:, 9], te[::5, :9]
            s2 = float(np.var(y, ddof=1))
            mean, var, betas = orc.predict_table(D, y, draws, Dt, s2)
            out["sets"].append(dict(size=size, sample=i, sigma2=s2, mean=arr(mean), var=arr(var), beta=arr(betas)))
    dump("gv_all_golden.json", out)


def make_cfg3_bundled():
    """BASELINE config 3 on the grid AS BUNDLED (2D Codes and Designs/hyperpars.matrix.txt was tuned for 14 points;
    on maximin-100 most of its covariance matrices are numerically singular): for three grid rows, every one of
    the 1728 Halton nodes -- the conditional log-likelihood where the reference arithmetic succeeds, null where
    its Cholesky (mnormt::dmnorm -> chol, ADV:573) stops, and the condition number, so that the test can tell
    a clear failure from a borderline one."""
    _, X = read_table(os.path.join(DATA, "maximin_100.txt"))
    _, H = read_table(os.path.join(DATA, "adv_hyperpars_matrix.txt"))
    y = np.array([orc.test_function_2d(a, b, 4) for a, b in X])
    s2, N, tau, lam = float(np.var(y, ddof=1)), 1728, 100.0, 4.0
    u = orc.runif_halton(N)
    rows = []
    for g in (0, 29, 59):
        th1, th2 = orc.qigamma(u, H[g, 0], H[g, 1]), orc.qigamma(u, H[g, 2], H[g, 3])
        vals, conds = [], []
        for j in range(N):
            w = np.array([u[j], 1.0 - u[j]])
            Th = np.array([[th1[j], th2[j]], [(1 + lam) * th1[j], (1 + lam) * th2[j]]])
            S = s2 * np.sum(w ** 2) * orc.mixed_corr_matrix_general(X, w, Th) + tau ** 2
            conds.append(float(np.linalg.cond(S)))
            try:
                vals.append(float(orc.loglik_general(X, y, w, Th, s2, orc.MEAN_ZERO_PLUS_TAU2, tau ** 2)[0]))
            except np.linalg.LinAlgError:
                vals.append(None)
        rows.append(dict(row=g, values=vals, cond=conds))
    dump("cfg3_bundled_golden.json", dict(sigma2=s2, N=N, tau=tau, aniso_lambda=lam, y=arr(y), rows=rows))


# ----------------------------------------------------------------------------- 1-D (config 1)
def make_d1():
    with open(os.path.join(DATA, "d1_designs_head.txt")) as fh:
        lines = [ln.strip() for ln in fh if ln.strip()]
    row = np.array([float(t) for t in lines[1].replace('"', " ").split()[1:]])   # first design
    X = row.reshape(-1, 1)
    # D1:331-339's default simulator is not needed for a likelihood known-answer: any y works
    y = np.sin(6.0 * row) + 0.3 * row
    nu = 5.0                                                                    # D1:1080
    out = dict(X=arr(row), y=arr(y), nu=nu, cases=[])
    for theta_t, s2 in (([0.0, 1.0, 0.5], 1.0), ([-0.7, 0.3, 1.5], 0.4)):
        lp = orc.logpost_1d(X, theta_t, y, s2, nu)
        out["cases"].append(dict(theta_t=theta_t, sigma2=s2, val=lp["val"], beta=lp["beta"],
                                 R_inv=matrix_digest(lp["R_inv"])))
    dump("d1_golden.json", out)


def make_d1f():
    """Two-family 1-D script (D1F): Matern(nu = 5, theta1) + cubic spline(theta2) on the same first design."""
    with open(os.path.join(DATA, "d1_designs_head.txt")) as fh:
        lines = [ln.strip() for ln in fh if ln.strip()]
    row = np.array([float(t) for t in lines[1].replace('"', " ").split()[1:]])
    X = row.reshape(-1, 1)
    y = np.sin(6.0 * row) + 0.3 * row
    nu = 5.0
    out = dict(X=arr(row), y=arr(y), nu=nu, cases=[], xt=[0.07, 0.41, 0.88])
    out["R_combined"] = matrix_digest(orc.corr_matrix_combined(X, 0.7, 0.5, 0.6, nu))
    out["r_combined"] = arr(orc.corr_vec_combined(0.41, X, 0.7, 0.5, 0.6, nu))       # un-normalised (D1F:479)
    out["R_spline"] = matrix_digest(orc.corr_matrix_spline(X, 0.45))
    for theta_t, s2 in (([-0.5, -0.4, 0.5], 1.0), ([-0.9, -0.2, 1.5], 0.4)):
        lp = orc.logpost_2f(X, theta_t, y, s2, nu)
        th1, th2, p = math.exp(theta_t[0]), math.exp(theta_t[1]), 1.0 / (1.0 + math.exp(-theta_t[2]))
        pred = [orc.predict_post_2f(x, X, y, p, th1, th2, s2, nu) for x in out["xt"]]
        out["cases"].append(dict(theta_t=theta_t, sigma2=s2, val=lp["val"], beta=lp["beta"],
                                 R_inv=matrix_digest(lp["R_inv"]), pred_mean=[float(m) for m, _ in pred],
                                 pred_var=[float(v) for _, v in pred]))
    dump("d1f_golden.json", out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["hx", "ani", "adv", "gv", "gv_all", "cfg3_bundled", "d1", "d1f"]
    for w in which:
        globals()["make_" + w]()
