"""50-digit mpmath evaluation of the same quantities as oracle/ccgp_oracle.py.

TEST INFRASTRUCTURE ONLY.  This is the arbiter when the fp64 oracle and the HIP
path disagree, and the only independent pin the oracle has (the reference ships no
golden vectors; see oracle/__init__.py).  It deliberately uses a DIFFERENT
formulation from the fp64 restatement -- direct squared differences instead of the
expanded U + t(U) + V form (HX:352-355), a hand-rolled Cholesky instead of LAPACK --
so that a shared mistake is unlikely.  Small n only (pure-Python loops).
"""
from __future__ import annotations

import mpmath as mp

mp.mp.dps = 50


def _mpf_matrix(A):
    return [[mp.mpf(float(v)) for v in row] for row in A]


def mixed_corr(X, w, Theta):
    """sum_c w_c^2 exp(-sum_k theta_ck (x_ik - x_jk)^2) / sum_c w_c^2  (HX:408-415)."""
    Xm = _mpf_matrix(X)
    n, d = len(Xm), len(Xm[0])
    wm = [mp.mpf(float(v)) for v in w]
    Tm = _mpf_matrix(Theta)
    sw = sum(v * v for v in wm)
    R = [[mp.mpf(0)] * n for _ in range(n)]
    for i in range(n):
        for j in range(i + 1):
            acc = mp.mpf(0)
            for c in range(len(wm)):
                dist = sum(Tm[c][k] * (Xm[i][k] - Xm[j][k]) ** 2 for k in range(d))
                acc += wm[c] ** 2 * mp.exp(-dist)
            R[i][j] = R[j][i] = acc / sw
    return R


def mixed_corr_vec(x, X, w, Theta):
    Xm = _mpf_matrix(X)
    xm = [mp.mpf(float(v)) for v in x]
    wm = [mp.mpf(float(v)) for v in w]
    Tm = _mpf_matrix(Theta)
    sw = sum(v * v for v in wm)
    out = []
    for i in range(len(Xm)):
        acc = mp.mpf(0)
        for c in range(len(wm)):
            dist = sum(Tm[c][k] * (xm[k] - Xm[i][k]) ** 2 for k in range(len(xm)))
            acc += wm[c] ** 2 * mp.exp(-dist)
        out.append(acc / sw)
    return out


def cholesky(A):
    n = len(A)
    L = [[mp.mpf(0)] * n for _ in range(n)]
    for j in range(n):
        s = A[j][j] - sum(L[j][k] ** 2 for k in range(j))
        if s <= 0:
            raise ArithmeticError("not positive definite at pivot %d" % j)
        L[j][j] = mp.sqrt(s)
        for i in range(j + 1, n):
            L[i][j] = (A[i][j] - sum(L[i][k] * L[j][k] for k in range(j))) / L[j][j]
    return L


def forward(L, b):
    n = len(L)
    z = [mp.mpf(0)] * n
    for i in range(n):
        z[i] = (b[i] - sum(L[i][k] * z[k] for k in range(i))) / L[i][i]
    return z


def loglik(X, y, w, Theta, sigma2, mean_mode=0, tau2=0.0):
    """Same contract as ccgp_oracle.loglik_general; returns (loglik, beta) as mpf."""
    n = len(y)
    ym = [mp.mpf(float(v)) for v in y]
    R = mixed_corr(X, w, Theta)
    c = mp.mpf(float(sigma2)) * sum(mp.mpf(float(v)) ** 2 for v in w)
    if mean_mode == 0:
        L = cholesky(R)
        zy = forward(L, ym)
        z1 = forward(L, [mp.mpf(1)] * n)
        beta = sum(a * b for a, b in zip(z1, zy)) / sum(a * a for a in z1)
        q = sum((a - beta * b) ** 2 for a, b in zip(zy, z1)) / c
        logdet = n * mp.log(c) + 2 * sum(mp.log(L[i][i]) for i in range(n))
    else:
        t2 = mp.mpf(float(tau2))
        S = [[c * R[i][j] + t2 for j in range(n)] for i in range(n)]
        L = cholesky(S)
        zy = forward(L, ym)
        beta = mp.mpf(0)
        q = sum(a * a for a in zy)
        logdet = 2 * sum(mp.log(L[i][i]) for i in range(n))
    ll = -(n * mp.log(2 * mp.pi) + logdet + q) / 2
    return ll, beta


def predict(X, y, w, Theta, sigma2, Xtest):
    """Per-draw predictive mean / variance (HX:655-673) for every row of Xtest."""
    n = len(y)
    ym = [mp.mpf(float(v)) for v in y]
    R = mixed_corr(X, w, Theta)
    L = cholesky(R)
    zy = forward(L, ym)
    z1 = forward(L, [mp.mpf(1)] * n)
    v2 = sum(a * a for a in z1)
    beta = sum(a * b for a, b in zip(z1, zy)) / v2
    zc = [a - beta * b for a, b in zip(zy, z1)]
    s2 = mp.mpf(float(sigma2))
    means, variances = [], []
    for x in Xtest:
        r = mixed_corr_vec(x, X, w, Theta)
        wv = forward(L, r)
        means.append(beta + sum(a * b for a, b in zip(zc, wv)))
        t = 1 - sum(a * b for a, b in zip(z1, wv))
        variances.append(s2 * (1 - sum(a * a for a in wv) + t * t / v2))
    return means, variances, beta


def halton2(i):
    """Radical inverse base 2 of the positive integer i."""
    f, r = mp.mpf(1) / 2, mp.mpf(0)
    while i:
        if i & 1:
            r += f
        i >>= 1
        f /= 2
    return r
