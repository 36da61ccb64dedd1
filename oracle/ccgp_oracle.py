"""numpy/scipy restatement of the reference's per-draw GP evaluation.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  What pins it: the reference holds no tests and no
golden vectors, but it holds ONE recorded output, `Ground Vibrations Emulator/Results/Size 50 Results 1.txt`; its
single-GP columns are reproduced by corr_matrix / corr_vec / solve / beta_mle / the predictive formulas to 1e-8 on all
450 numbers (tests/test_reference_pins.py), the hyperprior pair hard-coded at HX:774-775 is the argmax of
choose_hyperpars, and the Combined-GP columns agree up to Monte-Carlo noise.  PARITY UNPINNED for what no reference-held
number covers: the log-likelihood scalar itself (arbitrated by oracle/mp_check.py at 50 digits and by LAPACK), the Halton
start index, every Matern / spline / ADV result, and base R's solve() tolerance rule (restated from its documentation).

Every function names the reference lines it follows.  Abbreviations:
  HX  = Heat Exchanger Emulator/Combined GP Heat Exchanger.R
  GV  = Ground Vibrations Emulator/Combined GP Ground Vibrations.R
  ISO = 2D Codes and Designs/2D Combined GP Isotropic Public.R
  ADV = 2D Codes and Designs/2D Combined GP Isotropic Advanced.R
  ANI = 2D Codes and Designs/2D Combined GP Anisotropic Public.R
  D1  = 1D Codes and Designs/1D Combined GP Public.R
  D1F = 1D Codes and Designs/1D Combined GP Two Families Public.R
  BSQ = Batch Sequential ME Designs/Batch Sequential ME Design.R

The operation ORDER of the R code is kept (expanded-distance form, LU inverse for
R.Inv, Cholesky + chol2inv inside dmnorm) because that order is what a drop-in has
to agree with; third-party pieces that are not under /root/reference
(mnormt::dmnorm, fOptions::runif.halton, pscl::qigamma, base besselK) are
restated from their published definitions.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla
import scipy.special as sps
import scipy.stats as sst

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- a1/a2
def corr_matrix(X, theta):
    """Gaussian Gram matrix, one scale per input dimension.

    HX:328-337 (general d), ANI:351-360 (d = 2, theta = c(theta1, theta2)).
    R = exp(-(U + t(U) + V)), U[i, .] = sum_k theta_k x_ik^2, V = -2 X Theta X'.
    """
    X = np.asarray(X, dtype=np.float64)
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    n = X.shape[0]
    Theta = np.diag(theta)
    u = ((X ** 2) @ Theta).sum(axis=1)          # apply(X^2 %*% Theta, 1, sum)
    U = np.repeat(u[:, None], n, axis=1)        # matrix(u, n, n, byrow = F)
    V = -2.0 * ((X @ Theta) @ X.T)
    dist = (U + U.T) + V
    return np.exp(-dist)


def corr_matrix_iso(X, theta):
    """HX:347-356 (= GV:346, ISO:350, ADV:353-362, BSQ:347): Theta = theta * I_d."""
    X = np.asarray(X, dtype=np.float64)
    return corr_matrix(X, np.full(X.shape[1], float(theta)))


# --------------------------------------------------------------------------- a3
def corr_vec(x, X, theta):
    """Correlations between one new site and the design. ANI:369-377 / HX:367-375.

    r_i = exp(-((theta' x^2) - 2 (X Theta x)_i + sum_k theta_k x_ik^2)).
    """
    X = np.asarray(X, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    Theta = np.diag(theta)
    a = float(theta @ (x ** 2))
    b = 2.0 * ((X @ Theta) @ x)
    c = ((X ** 2) @ Theta).sum(axis=1)
    return np.exp(-((a - b) + c))


def corr_vec_iso(x, X, theta):
    """HX:367-375."""
    X = np.asarray(X, dtype=np.float64)
    return corr_vec(x, X, np.full(X.shape[1], float(theta)))


# --------------------------------------------------------------------------- a4/a5
def _mix(p, A1, A2):
    return (p ** 2 * A1 + (1.0 - p) ** 2 * A2) / (p ** 2 + (1.0 - p) ** 2)


def mixed_corr_matrix_iso(D, p, theta1, theta2):
    """HX:408-415 (GV, ISO, BSQ identical; ADV:414-421 passes lambda as theta2)."""
    return _mix(p, corr_matrix_iso(D, theta1), corr_matrix_iso(D, theta2))


def mixed_corr_vec_iso(x, D, p, theta1, theta2):
    """HX:425-431."""
    return _mix(p, corr_vec_iso(x, D, theta1), corr_vec_iso(x, D, theta2))


def mixed_corr_matrix_aniso(D, p, theta1, theta2, lam):
    """ANI:399-406: R2 uses (1+lambda)*(theta1, theta2)."""
    t = np.array([theta1, theta2], dtype=np.float64)
    return _mix(p, corr_matrix(D, t), corr_matrix(D, (1.0 + lam) * t))


def mixed_corr_vec_aniso(x, D, p, theta1, theta2, lam):
    """ANI:416-422."""
    t = np.array([theta1, theta2], dtype=np.float64)
    return _mix(p, corr_vec(x, D, t), corr_vec(x, D, (1.0 + lam) * t))


# --------------------------------------------------------------------------- a6/a7
def beta_mle(R_inv, y):
    """HX:384-388: 1' R.Inv y / sum(R.Inv)."""
    R_inv = np.asarray(R_inv, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    one = np.ones(y.shape[0])
    return float(((one @ R_inv) @ y) / R_inv.sum())


def sigma2_mle(R_inv, y, beta):
    """HX:394-399 / D1:411-415: (y - beta 1)' R.Inv (y - beta 1) / n."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    u = y - beta * np.ones(y.shape[0])
    return float(((u @ np.asarray(R_inv, dtype=np.float64)) @ u) / y.shape[0])


# --------------------------------------------------------------------------- a12/a13
def solve_inverse(R, tol=None):
    """base R solve(R) (HX:454) = solve.default(a, tol = .Machine$double.eps) -> La_solve: LAPACK dgesv(R, I), an error
    for an exactly singular U ("system is exactly singular"), THEN rcond = dgecon("1", LU, ||R||_1) and an error when
    rcond < tol ("system is computationally singular: reciprocal condition number = ...").  The reference wraps the
    call in try() and maps either error to R.Inv <- NA (HX:454-455).  Base R is not in the reference tree; restated
    from its documented behaviour (?solve: "tol: the tolerance for detecting linear dependencies in the columns of a").
    tol: the 1-D scripts pass tol = 1e-16 (D1:440)."""
    A = np.asarray(R, dtype=np.float64)
    x = np.linalg.inv(A)                      # dgesv(A, I); raises LinAlgError on an exactly singular U
    lu, _, info = sla.lapack.dgetrf(A)        # the same factorisation again, for the condition estimate
    anorm = np.abs(A).sum(axis=0).max()
    rcond, info = sla.lapack.dgecon(lu, anorm, norm="1")
    if not rcond >= (np.finfo(np.float64).eps if tol is None else tol):
        raise np.linalg.LinAlgError("system is computationally singular: reciprocal condition number = %g" % rcond)
    return x


def dmnorm_log(x, mean, varcov):
    """mnormt::dmnorm(x, mean, varcov, log = TRUE) -- package source not in the
    reference tree; restated from its definition: pd.solve symmetrises, takes the
    upper Cholesky factor, chol2inv, log.det = 2 sum log diag(U);
    logPDF = -(Q + d log 2pi + log.det)/2 with Q = (x-mean)' Sigma^-1 (x-mean).
    Call sites: HX:460, HX:570, GV:448, ISO:451, ADV:465, ADV:573, ANI:455, BSQ:448.
    Raises numpy.linalg.LinAlgError when varcov is not positive definite (R: error).
    """
    S = np.asarray(varcov, dtype=np.float64)
    S = (S + S.T) / 2.0
    d = S.shape[0]
    xc = np.asarray(x, dtype=np.float64).reshape(-1) - mean
    U = sla.cholesky(S, lower=False)
    inv, info = sla.lapack.dpotri(U, lower=0)
    if info != 0:
        raise np.linalg.LinAlgError("dpotri failed")
    inv = np.triu(inv) + np.triu(inv, 1).T
    log_det = 2.0 * np.log(np.diag(U)).sum()
    Q = float((inv @ xc) @ xc)
    return -(Q + d * LOG_2PI + log_det) / 2.0


# --------------------------------------------------------------------------- a8
PRIOR_VARIANTS = ("HX", "ADV", "GV", "ISO", "BSQ", "D1", "ANI")


def log_jacobian(theta_t):
    """-phi - 2 log(1+e^-phi) + psi1 + psi2 [+ zeta]  (HX:461, ANI:459)."""
    t = np.asarray(theta_t, dtype=np.float64)
    val = -t[2] - 2.0 * math.log(1.0 + math.exp(-t[2])) + t[0] + t[1]
    if t.shape[0] == 4:
        val += t[3]
    return float(val)


def log_prior(theta_t, variant, prior_pars=None):
    """The script-specific log-prior on the transformed scale.

    HX:462 / ADV:467 inverse-gamma with passed (a1,b1,a2,b2); GV:450; ISO:453 =
    BSQ:450 = D1:636; ANI:462.
    """
    t = np.asarray(theta_t, dtype=np.float64)
    psi1, psi2 = t[0], t[1]
    th1, th2 = math.exp(psi1), math.exp(psi2)
    if variant in ("HX", "ADV"):
        a1, b1, a2, b2 = prior_pars
        return float(-(a1 + 1.0) * psi1 - b1 / th1 - (a2 + 1.0) * psi2 - b2 / th2)
    if variant == "GV":
        return float(-4.0 * psi1 - 1.0 / th1 - 6.0 * psi2 - 75.0 / th2)
    if variant in ("ISO", "BSQ", "D1"):
        return float(-4.0 * psi1 - 2.0 / th1 - 6.0 * psi2 - 16.0 / th2)
    if variant == "ANI":
        zeta = t[3]
        lam = math.exp(zeta)
        return float(-psi1 - psi1 ** 2 / 2.0 - psi2 - psi2 ** 2 / 2.0 - 4.0 * zeta - 4.0 / lam)
    raise ValueError(variant)


def logpost(D, theta_t, y, sigma2, variant="HX", prior_pars=None):
    """Joint log-posterior of one transformed draw -> dict(val, beta, R_inv).

    HX:441-466, GV:429-454, ISO:433-457, ADV:447-471, ANI:433-467, BSQ:430-454.
    Steps kept in the reference's order: unpack, Mixed.corr.matrix, solve(R) (NA on
    failure), beta.MLE, dmnorm(y, sum(beta), (p^2+(1-p)^2) sigma2 R), + log-Jacobian
    + log-prior.  ADV also returns like = exp(log.like) (ADV:470).
    """
    t = np.asarray(theta_t, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    theta1, theta2 = math.exp(t[0]), math.exp(t[1])
    p = 1.0 / (1.0 + math.exp(-t[2]))
    if variant == "ANI":
        R = mixed_corr_matrix_aniso(D, p, theta1, theta2, math.exp(t[3]))
    else:
        R = mixed_corr_matrix_iso(D, p, theta1, theta2)
    try:
        R_inv = solve_inverse(R)
    except np.linalg.LinAlgError:
        return dict(val=float("nan"), beta=float("nan"), R_inv=None, like=float("nan"))
    beta = beta_mle(R_inv, y)
    log_like = dmnorm_log(y, beta, (p ** 2 + (1.0 - p) ** 2) * sigma2 * R)
    val = log_like + log_jacobian(t) + log_prior(t, variant, prior_pars)
    return dict(val=float(val), beta=beta, R_inv=R_inv, like=math.exp(log_like),
                log_like=float(log_like))


# --------------------------------------------------------------------------- a9
def runif_halton(N):
    """fOptions::runif.halton(N, 1): base-2 radical inverse of 1..N (1/2, 1/4, 3/4, ...).
    Package source not in the reference tree; restated from the van der Corput
    definition (HX:554).  If the package used another start index the grid means
    move at O(1/N); flagged in DESIGN.md."""
    out = np.empty(N, dtype=np.float64)
    for i in range(1, N + 1):
        f, r, k = 0.5, 0.0, i
        while k:
            if k & 1:
                r += f
            k >>= 1
            f *= 0.5
        out[i - 1] = r
    return out


def qigamma(p, alpha, beta):
    """pscl::qigamma(p, alpha, beta) = 1 / qgamma(1 - p, shape alpha, rate beta)
    (HX:555-556).  Equals scipy.stats.invgamma.ppf(p, alpha, scale=beta)."""
    p = np.asarray(p, dtype=np.float64)
    return 1.0 / (sst.gamma.ppf(1.0 - p, alpha) / beta)


def cond_like_log(D, y, p, theta1, theta2, sigma2, tau):
    """log of cond.like, HX:561-572: dmnorm(y, 0, sigma2 (p^2+(1-p)^2) R + tau^2 11')."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n = y.shape[0]
    sigma2_t = sigma2 * (p ** 2 + (1.0 - p) ** 2)
    R = mixed_corr_matrix_iso(D, p, theta1, theta2)
    return dmnorm_log(y, 0.0, sigma2_t * R + tau ** 2 * np.ones((n, n)))


def likeli_hyperpars(D, y, theta1_pars, theta2_pars, sigma2, N=1000, tau=50.0,
                     return_logs=False):
    """HX:549-575 (N=1000, tau=50) / ADV:552-578 (N=1728, tau=100): mean over N Halton
    points of exp(cond.like); p, theta1 and theta2 all come from the SAME quantile."""
    u = runif_halton(N)
    th1 = qigamma(u, theta1_pars[0], theta1_pars[1])
    th2 = qigamma(u, theta2_pars[0], theta2_pars[1])
    logs = np.array([cond_like_log(D, y, u[j], th1[j], th2[j], sigma2, tau)
                     for j in range(N)])
    if return_logs:
        return logs
    return float(np.mean(np.exp(logs)))


def choose_hyperpars(D, y, hyper, sigma2, N=1000, tau=50.0, take_log=True):
    """HX:584-595 (log of the mean, HX:591) / ADV:588-599 (no log, ADV:595).
    Returns (argmax row index 0-based, per-row values)."""
    hyper = np.asarray(hyper, dtype=np.float64)
    vals = np.empty(hyper.shape[0])
    for i in range(hyper.shape[0]):
        m = likeli_hyperpars(D, y, hyper[i, 0:2], hyper[i, 2:4], sigma2, N, tau)
        vals[i] = math.log(m) if take_log else m
    return int(np.argmax(vals)), vals


# --------------------------------------------------------------------------- a10/a11
def factors(R_inv, beta, y):
    """HX:604-613: mean.factor = R.Inv (y - beta), var.factor1 = colSums(R.Inv),
    var.factor2 = sum(var.factor1)."""
    R_inv = np.asarray(R_inv, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    mean_factor = R_inv @ (y - beta)
    v1 = R_inv.sum(axis=0)
    return mean_factor, v1, float(v1.sum())


def predict_post_from_factors(r, beta, mean_factor, v1, v2, R_inv, sigma2):
    """The arithmetic of predict.post once r is known (HX:667-670)."""
    r = np.asarray(r, dtype=np.float64).reshape(-1)
    var = sigma2 * (1.0 - (r @ R_inv) @ r + (1.0 - v1 @ r) ** 2 / v2)
    mean = beta + mean_factor @ r
    return float(mean), float(var)


def predict_post_iso(x, D, y, p, theta1, theta2, sigma2):
    """HX:655-673 for one (draw, test point), recomputing the cached per-draw terms
    exactly as Metro/factors.frame would have stored them (HX:454-458, HX:604-613)."""
    R = mixed_corr_matrix_iso(D, p, theta1, theta2)
    R_inv = solve_inverse(R)
    beta = beta_mle(R_inv, y)
    mf, v1, v2 = factors(R_inv, beta, y)
    r = mixed_corr_vec_iso(x, D, p, theta1, theta2)
    return predict_post_from_factors(r, beta, mf, v1, v2, R_inv, sigma2)


def predict_post_aniso(x, D, y, p, theta1, theta2, lam, sigma2):
    """ANI:604-623."""
    R = mixed_corr_matrix_aniso(D, p, theta1, theta2, lam)
    R_inv = solve_inverse(R)
    beta = beta_mle(R_inv, y)
    mf, v1, v2 = factors(R_inv, beta, y)
    r = mixed_corr_vec_aniso(x, D, p, theta1, theta2, lam)
    return predict_post_from_factors(r, beta, mf, v1, v2, R_inv, sigma2)


def predict_table(D, y, draws, Xtest, sigma2, aniso=False):
    """(S x m) mean and variance tables: what prediction() consumes before its
    rnorm/quantile step (HX:686-693).  draws rows: (p, theta1, theta2[, lambda])."""
    draws = np.asarray(draws, dtype=np.float64)
    Xtest = np.asarray(Xtest, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    S, m = draws.shape[0], Xtest.shape[0]
    mean = np.empty((S, m))
    var = np.empty((S, m))
    betas = np.empty(S)
    for s in range(S):
        if aniso:
            p, t1, t2, lam = draws[s]
            R = mixed_corr_matrix_aniso(D, p, t1, t2, lam)
        else:
            p, t1, t2 = draws[s]
            R = mixed_corr_matrix_iso(D, p, t1, t2)
        R_inv = solve_inverse(R)
        beta = beta_mle(R_inv, y)
        betas[s] = beta
        mf, v1, v2 = factors(R_inv, beta, y)
        for j in range(m):
            if aniso:
                r = mixed_corr_vec_aniso(Xtest[j], D, p, t1, t2, lam)
            else:
                r = mixed_corr_vec_iso(Xtest[j], D, p, t1, t2)
            mean[s, j], var[s, j] = predict_post_from_factors(r, beta, mf, v1, v2, R_inv, sigma2)
    return mean, var, betas


# --------------------------------------------------------------------------- general K
def params_from_iso(p, theta1, theta2, d):
    """Pack an isotropic 2-component draw as the C-ABI row (w_1..w_K, theta_c,k)."""
    return np.concatenate([[p, 1.0 - p], np.full(d, theta1), np.full(d, theta2)])


def params_from_aniso(p, theta1, theta2, lam):
    return np.array([p, 1.0 - p, theta1, theta2, (1.0 + lam) * theta1, (1.0 + lam) * theta2])


def unpack_params(row, K, d):
    row = np.asarray(row, dtype=np.float64)
    return row[:K], row[K:].reshape(K, d)


def mixed_corr_matrix_general(X, w, Theta):
    """K-component generalisation of HX:408-415: sum_c w_c^2 R_c / sum_c w_c^2."""
    w = np.asarray(w, dtype=np.float64)
    acc = None
    for c in range(w.shape[0]):
        Rc = w[c] ** 2 * corr_matrix(X, Theta[c])
        acc = Rc if acc is None else acc + Rc
    return acc / np.sum(w ** 2)


def mixed_corr_vec_general(x, X, w, Theta):
    w = np.asarray(w, dtype=np.float64)
    acc = None
    for c in range(w.shape[0]):
        rc = w[c] ** 2 * corr_vec(x, X, Theta[c])
        acc = rc if acc is None else acc + rc
    return acc / np.sum(w ** 2)


MEAN_PROFILE_BETA = 0
MEAN_ZERO_PLUS_TAU2 = 1


def loglik_general(X, y, w, Theta, sigma2, mean_mode=MEAN_PROFILE_BETA, tau2=0.0):
    """The likelihood term of logpost (mode 0: HX:454-460) or of cond.like (mode 1:
    HX:567-570) for K components.  Returns (loglik, beta).  Raises LinAlgError on a
    non-PD matrix (the C-ABI reports that through status[])."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n = y.shape[0]
    w = np.asarray(w, dtype=np.float64)
    R = mixed_corr_matrix_general(X, w, Theta)
    c = sigma2 * np.sum(w ** 2)
    if mean_mode == MEAN_PROFILE_BETA:
        R_inv = solve_inverse(R)
        beta = beta_mle(R_inv, y)
        return dmnorm_log(y, beta, c * R), beta
    return dmnorm_log(y, 0.0, c * R + tau2 * np.ones((n, n))), 0.0


def loglik_grad_fd(X, y, row, K, d, sigma2, h=1e-6):
    """Central finite differences of the profiled log-likelihood with respect to the
    raw C-ABI parameter row -- the check for the build-defined gradient extension
    (the reference has no analytic gradient: LearnBayes::laplace differences
    numerically, HX:493)."""
    row = np.asarray(row, dtype=np.float64)
    g = np.empty_like(row)
    for j in range(row.shape[0]):
        e = np.zeros_like(row)
        e[j] = h * max(1.0, abs(row[j]))
        wp, Tp = unpack_params(row + e, K, d)
        wm, Tm = unpack_params(row - e, K, d)
        g[j] = (loglik_general(X, y, wp, Tp, sigma2)[0]
                - loglik_general(X, y, wm, Tm, sigma2)[0]) / (2.0 * e[j])
    return g


# --------------------------------------------------------------------------- entropy criteria (8(f)-4)
def cross_corr_matrix(D_old, D_new, theta):
    """BSQ:835-848: exp(-(U + V + W)), n.new x n.old, isotropic."""
    D_old, D_new = np.asarray(D_old, dtype=np.float64), np.asarray(D_new, dtype=np.float64)
    d = D_new.shape[1]
    Theta = np.diag(np.full(d, float(theta)))
    U = ((D_new ** 2) @ Theta).sum(axis=1)[:, None]
    V = -2.0 * ((D_new @ Theta) @ D_old.T)
    W = ((D_old ** 2) @ Theta).sum(axis=1)[None, :]
    return np.exp(-((U + V) + W))


def entropy(D, p, theta1, theta2):
    """BSQ:856-861: -det(R.mixed)."""
    return -float(np.linalg.det(mixed_corr_matrix_iso(D, p, theta1, theta2)))


def augmented_mixed_entropy(D_old, D_new, p, theta1, theta2):
    """BSQ:869-877 with R.old.Inv = solve(R.old) as in Batch.Entropy.optim (BSQ:924-925)."""
    R_old_inv = solve_inverse(mixed_corr_matrix_iso(D_old, p, theta1, theta2))
    R_cross = _mix(p, cross_corr_matrix(D_old, D_new, theta1), cross_corr_matrix(D_old, D_new, theta2))
    R_new = mixed_corr_matrix_iso(D_new, p, theta1, theta2)
    return -float(np.linalg.det(R_new - (R_cross @ R_old_inv) @ R_cross.T))


# --------------------------------------------------------------------------- config 1 (CPU plumbing)
def matern_corr(nu, h, theta):
    """D1:348-351: (2 sqrt(nu)|h|/theta)^nu K_nu(2 sqrt(nu)|h|/theta) / (Gamma(nu) 2^(nu-1)),
    1 at h = 0.  base besselK -> scipy.special.kv."""
    h = np.abs(np.asarray(h, dtype=np.float64))
    z = 2.0 * math.sqrt(nu) * h / theta
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        val = z ** nu * sps.kv(nu, z) / (sps.gamma(nu) * 2.0 ** (nu - 1.0))
    return np.where(h == 0.0, 1.0, val)


def corr_matrix_matern(nu, X, theta):
    """D1:368-374 (X is n x 1)."""
    x = np.asarray(X, dtype=np.float64).reshape(-1)
    return matern_corr(nu, x[:, None] - x[None, :], theta)


def corr_vec_matern(x, X, theta, nu):
    """D1:383-389."""
    return matern_corr(nu, float(x) - np.asarray(X, dtype=np.float64).reshape(-1), theta)


def log_likeli_1d(nu, theta, D, y):
    """D1:437-444 log.likeli(nu, theta, D.train, y.train) = log(det(R)) + n log(sigma2.MLE): the ordinary-kriging
    profile objective that MLEs() minimises with nlminb (D1:455-471); = D1F:527-537 with corr.matrix.Matern."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    R = corr_matrix_matern(nu, D, theta)
    R_inv = solve_inverse(R, tol=1e-16)
    beta = beta_mle(R_inv, y)
    s2 = sigma2_mle(R_inv, y, beta)
    return math.log(np.linalg.det(R)) + y.shape[0] * math.log(s2)


def logpost_1d(D, theta_t, y, sigma2, nu):
    """D1:609-641 (Matern nu, prior D1:636)."""
    t = np.asarray(theta_t, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    theta1, theta2 = math.exp(t[0]), math.exp(t[1])
    p = 1.0 / (1.0 + math.exp(-t[2]))
    R = _mix(p, corr_matrix_matern(nu, D, theta1), corr_matrix_matern(nu, D, theta2))
    R_inv = solve_inverse(R)
    beta = beta_mle(R_inv, y)
    log_like = dmnorm_log(y, beta, (p ** 2 + (1.0 - p) ** 2) * sigma2 * R)
    val = log_like + log_jacobian(t) + log_prior(t, "D1")
    return dict(val=float(val), beta=beta, R_inv=R_inv)


def predict_post_1d(x, D, y, p, theta1, theta2, sigma2, nu):
    """D1:794-812 for one (draw, test point), recomputing the cached per-draw terms as
    factors.frame would have stored them (D1:760-781)."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    R = _mix(p, corr_matrix_matern(nu, D, theta1), corr_matrix_matern(nu, D, theta2))
    R_inv = solve_inverse(R)
    beta = beta_mle(R_inv, y)
    mf, v1, v2 = factors(R_inv, beta, y)
    r = _mix(p, corr_vec_matern(x, D, theta1, nu), corr_vec_matern(x, D, theta2, nu))
    return predict_post_from_factors(r, beta, mf, v1, v2, R_inv, sigma2)


# --------------------------------------------------------------------------- two-family 1-D script (D1F)
def spline_corr(theta, h):
    """D1F:346-357 spline.corr.func(theta, h): the non-negative cubic spline correlation."""
    u = np.abs(np.asarray(h, dtype=np.float64)) / theta
    return np.where(u <= 0.5, 1.0 - 6.0 * u ** 2 + 6.0 * u ** 3, np.where(u <= 1.0, 2.0 * (1.0 - u) ** 3, 0.0))


def corr_matrix_spline(X, theta):
    """D1F:398-404."""
    x = np.asarray(X, dtype=np.float64).reshape(-1)
    return spline_corr(theta, x[:, None] - x[None, :])


def corr_vec_spline(x, X, theta):
    """D1F:412-418."""
    return spline_corr(theta, float(x) - np.asarray(X, dtype=np.float64).reshape(-1))


def corr_matrix_combined(X, p, theta1, theta2, nu):
    """D1F:453-462: (p^2 Matern(nu, theta1) + (1-p)^2 spline(theta2)) / (p^2 + (1-p)^2)."""
    return _mix(p, corr_matrix_matern(nu, X, theta1), corr_matrix_spline(X, theta2))


def corr_vec_combined(x, X, p, theta1, theta2, nu):
    """D1F:470-480 AS WRITTEN: `return(p^2*r1 + (1-p)^2*r2)/(p^2 + (1-p)^2)` returns before the
    division, so the vector is NOT normalised ("the D1F:479 quirk")."""
    return p ** 2 * corr_vec_matern(x, X, theta1, nu) + (1.0 - p) ** 2 * corr_vec_spline(x, X, theta2)


def logpost_2f(D, theta_t, y, sigma2, nu):
    """D1F:576-601 (prior D1F:596 = D1:636)."""
    t = np.asarray(theta_t, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    theta1, theta2 = math.exp(t[0]), math.exp(t[1])
    p = 1.0 / (1.0 + math.exp(-t[2]))
    R = corr_matrix_combined(D, p, theta1, theta2, nu)
    R_inv = solve_inverse(R)
    beta = beta_mle(R_inv, y)
    log_like = dmnorm_log(y, beta, (p ** 2 + (1.0 - p) ** 2) * sigma2 * R)
    val = log_like + log_jacobian(t) + log_prior(t, "D1")
    return dict(val=float(val), beta=beta, R_inv=R_inv)


def predict_post_2f(x, D, y, p, theta1, theta2, sigma2, nu):
    """D1F:737-754 for one (draw, test point) -- with the un-normalised corr.vec.combined."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    R_inv = solve_inverse(corr_matrix_combined(D, p, theta1, theta2, nu))
    beta = beta_mle(R_inv, y)
    mf, v1, v2 = factors(R_inv, beta, y)
    r = corr_vec_combined(x, D, p, theta1, theta2, nu)
    return predict_post_from_factors(r, beta, mf, v1, v2, R_inv, sigma2)


def test_function_2d(x, y, code):
    """ANI:330-341: the five bivariate test simulators (needed to make y.train)."""
    if code == 1:
        return math.exp(-1.4 * x) * math.cos(7 * math.pi * x * y / 2) + math.log(x + y + 0.1)
    if code == 2:
        return (((x - 0.2) ** 2 - (y - 0.7) ** 2) * math.exp(-5 * ((x - 0.8) ** 2 + (y - 0.1) ** 2))
                * math.cos(10 * (x - 0.5) * y))
    if code == 3:
        return ((x - 0.5) ** 2 + 4 * (y - 0.8) ** 2) * (math.cos(math.pi * (x - 0.1)) + math.cos(math.pi * (y - 0.5)))
    if code == 4:
        return (math.sin(2 * x) + math.cos(4 * x)) * (math.sin(8 * y) + math.cos(4 * y))
    if code == 5:
        return math.sin(9 * x - 4.5) / (9 * x - 4.5) * math.sin(12 * y - 6) / (12 * y - 6)
    raise ValueError(code)
