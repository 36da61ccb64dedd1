"""The reference's R function surface on top of libccgp (drop-in boundary).

Same function names (dots become underscores), argument order, argument meaning and
return shapes as the R scripts, so that parity tests read like the reference's own
calls.  Each script re-defines the same ~12 functions with a different kernel / prior,
so the surface is a class parameterised by the script it mirrors:

    gp = CombinedGP("HX")            # Heat Exchanger Emulator/Combined GP Heat Exchanger.R
    gp.logpost(D_train, theta, y, sigma2, theta1_pars, theta2_pars)  -> dict(val, beta, R_Inv)

Nothing numerical happens here: every method packs arguments and calls the C ABI
(api.Handle).  Error behaviour follows the reference: a covariance that cannot be
factorised gives NaN / None where R gives NA (HX:454-455) instead of raising.
"""
from __future__ import annotations

import numpy as np

from . import api

_SCRIPTS = {
    # kernel, prior id, number of transformed parameters, grid (N, tau, take_log)
    "HX": dict(aniso=False, prior=api.PRIOR_INVGAMMA, npar=3, N=1000, tau=50.0, take_log=True),
    "ADV": dict(aniso=False, prior=api.PRIOR_INVGAMMA, npar=3, N=1728, tau=100.0, take_log=False),
    "GV": dict(aniso=False, prior=api.PRIOR_GV, npar=3),
    "ISO": dict(aniso=False, prior=api.PRIOR_ISO, npar=3),
    "BSQ": dict(aniso=False, prior=api.PRIOR_ISO, npar=3),
    "ANI": dict(aniso=True, prior=api.PRIOR_ANI, npar=4),
}


def pack_iso(p, theta1, theta2, d):
    """(p, theta1, theta2) -> C-ABI row (w1, w2, theta_1k.., theta_2k..), HX:408-415."""
    return np.concatenate([[p, 1.0 - p], np.full(d, float(theta1)), np.full(d, float(theta2))])


def pack_aniso(p, theta1, theta2, lam):
    """ANI:399-406: component 2 uses (1+lambda)(theta1, theta2)."""
    return np.array([p, 1.0 - p, theta1, theta2, (1.0 + lam) * theta1, (1.0 + lam) * theta2], dtype=np.float64)


class _FamilyHandle:
    """api.Handle proxy that (re)selects the correlation family before every call, so that objects
    mirroring different scripts can share one device handle."""

    def __init__(self, handle, family, nu):
        self._handle, self._family, self._nu = handle, family, nu

    def __getattr__(self, name):
        attr = getattr(self._handle, name)
        if not callable(attr) or name in ("close", "set_kernel"):
            return attr

        def call(*args, **kwargs):
            self._handle.set_kernel(self._family, self._nu)
            try:
                return attr(*args, **kwargs)
            finally:
                self._handle.set_kernel(api.KERNEL_GAUSS, 0.0)
        return call


class CombinedGP:
    def __init__(self, script="HX", handle=None, device=0):
        if script not in _SCRIPTS:
            raise ValueError("script must be one of %s" % sorted(_SCRIPTS))
        self.script = script
        self.cfg = _SCRIPTS[script]
        self.h = handle if handle is not None else api.Handle(device)

    # ------------------------------------------------------------------ a1 / a2 / a3
    def corr_matrix(self, X, theta1, theta2=None):
        """HX:328-337 corr.matrix(X, theta) / ANI:351-360 corr.matrix(X, theta1, theta2)."""
        theta = np.atleast_1d(theta1) if theta2 is None else np.array([theta1, theta2], dtype=np.float64)
        return self.h.corr_matrix(X, theta)

    def corr_matrix_ISO(self, X, theta):
        """HX:347-356."""
        return self.h.corr_matrix(X, float(theta))

    def corr_vec(self, x, X, theta1, theta2=None):
        """ANI:369-377 corr.vec(x, X, theta1, theta2); general theta vector accepted."""
        theta = np.atleast_1d(theta1) if theta2 is None else np.array([theta1, theta2], dtype=np.float64)
        return self.h.corr_cross(np.asarray(x, dtype=np.float64).reshape(1, -1), X, theta)[0]

    def corr_vec_ISO(self, x, X, theta):
        """HX:367-375."""
        return self.h.corr_cross(np.asarray(x, dtype=np.float64).reshape(1, -1), X, float(theta))[0]

    # ------------------------------------------------------------------ a4 / a5
    def _row(self, D_train, p, theta1, theta2, lam=None):
        d = np.asarray(D_train).shape[1]
        if self.cfg["aniso"]:
            if lam is None:
                raise TypeError("the anisotropic script needs lambda")
            return pack_aniso(p, theta1, theta2, lam)
        return pack_iso(p, theta1, theta2, d)

    def Mixed_corr_matrix(self, D_train, p, theta1, theta2, lam=None):
        """HX:408-415 / ANI:399-406 / ADV:414-421 (theta2 plays 'lambda' there)."""
        return self.h.mixed_corr_matrix(D_train, 2, self._row(D_train, p, theta1, theta2, lam))

    def Mixed_corr_vec(self, x_new, D_train, p, theta1, theta2, lam=None):
        """HX:425-431 / ANI:416-422."""
        x = np.asarray(x_new, dtype=np.float64).reshape(1, -1)
        return self.h.mixed_corr_cross(x, D_train, 2, self._row(D_train, p, theta1, theta2, lam))[0]

    # ------------------------------------------------------------------ a6 / a7
    def beta_MLE(self, R_Inv, y):
        """HX:384-388."""
        return self.h.beta_mle(R_Inv, y)

    def sigma2_MLE(self, R_Inv, y_train, beta):
        """HX:394-399."""
        return self.h.sigma2_mle(R_Inv, y_train, beta)

    # ------------------------------------------------------------------ a8
    def logpost(self, D_train, theta, y, sigma2, theta1_pars=None, theta2_pars=None, want_R_Inv=True):
        """HX:441-466 (pars passed) / GV:429-454 / ISO:433-457 / ADV:447-471 / ANI:433-467.
        Returns dict(val, beta, R_Inv[, like]); NaN / None when R is not factorisable."""
        theta = np.asarray(theta, dtype=np.float64).ravel()
        if theta.size != self.cfg["npar"]:
            raise ValueError("%s logpost takes %d transformed parameters" % (self.script, self.cfg["npar"]))
        pars = None
        if self.cfg["prior"] == api.PRIOR_INVGAMMA:
            if theta1_pars is None or theta2_pars is None:
                raise TypeError("%s logpost needs theta1.pars and theta2.pars" % self.script)
            pars = np.concatenate([np.ravel(theta1_pars)[:2], np.ravel(theta2_pars)[:2]])
        r = self.h.logpost(D_train, y, sigma2, self.cfg["prior"], theta, pars, want_Rinv=want_R_Inv)
        out = dict(val=r["val"], beta=r["beta"], R_Inv=r["R_inv"] if r["status"] == 0 else None)
        if self.script == "ADV":
            out["like"] = float(np.exp(r["loglik"]))  # ADV:470
        return out

    # ------------------------------------------------------------------ a9
    def likeli_hyperpars(self, D_train, y_train, theta1_pars, theta2_pars, sigma2, aniso_lambda=None):
        """HX:549-575 / ADV:552-578: mean over the Halton nodes of exp(cond.like)."""
        hyper = np.array([[theta1_pars[0], theta1_pars[1], theta2_pars[0], theta2_pars[1]]], dtype=np.float64)
        out, _ = self.h.grid_marginal(D_train, y_train, sigma2, hyper, self.cfg["N"], self.cfg["tau"],
                                      take_log=False,
                                      aniso_lambda=-1.0 if aniso_lambda is None else aniso_lambda)
        return float(out[0])

    def choose_hyperpars(self, D_train, y_train, hyperpars_matrix, sigma2, aniso_lambda=None):
        """HX:584-595 (log of the mean) / ADV:588-599 (the mean itself).
        Returns dict(pars = winning row, likelihoods = per-row values)."""
        hyper = np.asarray(hyperpars_matrix, dtype=np.float64)
        out, arg = self.h.grid_marginal(D_train, y_train, sigma2, hyper, self.cfg["N"], self.cfg["tau"],
                                        take_log=self.cfg["take_log"],
                                        aniso_lambda=-1.0 if aniso_lambda is None else aniso_lambda)
        return dict(pars=hyper[arg], likelihoods=out, which_max=arg)

    # ------------------------------------------------------------------ a10 / a11
    def factors(self, MCMC_data, n_train, y_train):
        """HX:604-613: MCMC.data = (R.Inv flattened column-major, beta) ->
        c(mean.factor, var.factor1, var.factor2)."""
        data = np.asarray(MCMC_data, dtype=np.float64).ravel()
        R_Inv = data[: n_train * n_train].reshape(n_train, n_train, order="F")
        return self.h.factors(R_Inv, data[n_train * n_train], y_train)

    def predict_post(self, x_new, D_train, pars, sigma2):
        """HX:655-673 / ANI:604-623 / ADV:660-678: one frame row -> (mean, var).
        pars = (p, theta1, theta2[, lambda], beta, mean.factor[n], var.factor1[n],
        var.factor2, R.Inv[n*n])."""
        D_train = np.asarray(D_train, dtype=np.float64)
        n = D_train.shape[0]
        pars = np.asarray(pars, dtype=np.float64).ravel()
        o = 4 if self.cfg["aniso"] else 3
        p, theta1, theta2 = pars[0], pars[1], pars[2]
        beta = pars[o]
        mf = pars[o + 1: o + 1 + n]
        v1 = pars[o + 1 + n: o + 1 + 2 * n]
        v2 = pars[o + 1 + 2 * n]
        R_Inv = pars[o + 2 + 2 * n: o + 2 + 2 * n + n * n].reshape(n, n, order="F")
        # r = Mixed.corr.vec(x.new, ...) (HX:665) and the arithmetic of HX:667-670 in ONE device round trip
        if self.cfg["aniso"]:
            row = pack_aniso(p, theta1, theta2, pars[3])
        elif self.script == "ADV":
            row = pack_iso(p, theta1, theta1 * (1.0 + theta2), D_train.shape[1])           # ADV:672 as written
        else:
            row = pack_iso(p, theta1, theta2, D_train.shape[1])
        mean, var = self.h.predict_post(np.asarray(x_new, dtype=np.float64).reshape(1, -1), D_train, 2, row, beta, mf,
                                        v1, v2, R_Inv, sigma2)
        return np.array([[mean[0], var[0]]])

    def cross_corr_matrix(self, D_old, D_new, theta):
        """BSQ:835-848 cross.corr.matrix(D.old, D.new, theta): n.new x n.old, row t = corr.vec.ISO(D.new[t,], D.old, theta)."""
        return self.h.corr_cross(np.atleast_2d(np.asarray(D_new, dtype=np.float64)), D_old, float(theta))

    # ------------------------------------------------------------------ entropy criteria (BSQ)
    def Entropy(self, D, p, theta1, theta2):
        """Batch Sequential ME Design.R:856-861: -det(Mixed.corr.matrix(D, p, theta1, theta2))."""
        return float(self.Entropy_batch(np.asarray(D, dtype=np.float64)[None], p, theta1, theta2)[0])

    def Entropy_batch(self, designs, p, theta1, theta2):
        """Entropy for many candidate designs [B, n, d] in one device call (what the multi-start
        L-BFGS-B of Entropy.optim, BSQ:886-912, evaluates one at a time)."""
        designs = np.asarray(designs, dtype=np.float64)
        ld, st = self.h.mixed_logdet_designs(designs, 2, pack_iso(p, theta1, theta2, designs.shape[2]))
        return np.where(st == 0, -np.exp(ld), np.nan)

    def Augmented_Mixed_Entropy(self, D_old, D_new, p, theta1, theta2, R_old_Inv=None):
        """BSQ:869-877: -det(R.new - R.cross R.old^-1 R.cross').  Computed as
        -det(R(D.old U D.new)) / det(R(D.old)) (Schur complement), so R.old.Inv is not needed."""
        D_old, D_new = np.asarray(D_old, dtype=np.float64), np.asarray(D_new, dtype=np.float64)
        row = pack_iso(p, theta1, theta2, D_old.shape[1])
        ld_all, s1 = self.h.mixed_logdet_designs(np.vstack([D_old, D_new])[None], 2, row)
        ld_old, s2 = self.h.mixed_logdet_designs(D_old[None], 2, row)
        if s1[0] or s2[0]:
            return float("nan")
        return -float(np.exp(ld_all[0] - ld_old[0]))

    # ------------------------------------------------------------------ batched forms
    def draws_to_params(self, D_train, draws):
        """draws rows (p, theta1, theta2[, lambda]) -> C-ABI parameter matrix."""
        draws = np.atleast_2d(np.asarray(draws, dtype=np.float64))
        return np.stack([self._row(D_train, *row) for row in draws])

    def prediction_table(self, D_test, draws, D_train, sigma2, y_train, as_written=False):
        """The deterministic part of prediction()/compare.GP (HX:686-693, HX:713-725): the
        (draw x test point) mean and variance tables and y.hat = colMeans(mean).  The
        rnorm/quantile interval step (HX:696-699) needs R's RNG and is out of scope.

        script "ADV": the reference trains with R2 = corr.matrix.ISO(D, lambda) (ADV:417, ADV:456) but its
        predict.post builds r with theta1 * (1 + lambda) (ADV:672) -- two different second components.  The
        batched device path uses ONE kernel for R and r (the training one: what the posterior draws were
        fitted under).  as_written=True reproduces ADV:672 instead, through the literal per-draw path
        (factors.frame row + predict.post per test site): exact to the script, S * m small device calls."""
        if as_written and self.script == "ADV":
            D_test = np.atleast_2d(np.asarray(D_test, dtype=np.float64))
            frame = self.factors_frame_from_draws(draws, D_train, sigma2, y_train)
            S, m = frame.shape[0], D_test.shape[0]
            mean, var = np.empty((S, m)), np.empty((S, m))
            for s in range(S):
                for t in range(m):
                    mean[s, t], var[s, t] = self.predict_post(D_test[t], D_train, frame[s], sigma2)[0]
            return dict(mean=mean, var=var, beta=frame[:, 3], status=np.zeros(S, dtype=np.int32),
                        y_hat=mean.mean(axis=0))
        params = self.draws_to_params(D_train, draws)
        mean, var, beta, status = self.h.predict_batch(D_train, y_train, 2, params, D_test, sigma2)
        return dict(mean=mean, var=var, beta=beta, status=status, y_hat=mean.mean(axis=0))

    def factors_frame_from_draws(self, draws, D_train, sigma2, y_train):
        """Materialise the data frame factors.frame() returns (HX:625-644) for drop-in
        callers, given the posterior draws (the Metropolis chain itself is out of scope):
        columns p, theta1, theta2[, lambda], beta, mean.factor, var.factor1, var.factor2, R.Inv."""
        D_train = np.asarray(D_train, dtype=np.float64)
        n = D_train.shape[0]
        draws = np.atleast_2d(np.asarray(draws, dtype=np.float64))
        rows = []
        for row in draws:
            p, t1, t2 = row[0], row[1], row[2]
            theta_t = [np.log(t1), np.log(t2), np.log(p / (1.0 - p))]
            if self.cfg["aniso"]:
                theta_t.append(np.log(row[3]))
            pars = (1.0, 1.0, 1.0, 1.0)
            lp = self.h.logpost(D_train, y_train, sigma2, self.cfg["prior"], np.array(theta_t), pars)
            f = self.h.factors(lp["R_inv"], lp["beta"], y_train)
            rows.append(np.concatenate([row, [lp["beta"]], f, lp["R_inv"].ravel(order="F")]))
        return np.stack(rows)


class CombinedGP1D(CombinedGP):
    """The 1-D script's surface (1D Codes and Designs/1D Combined GP Public.R = D1): Matern(nu)
    components, every function takes nu the way the script does.

        gp = CombinedGP1D(nu=5)                                   # D1:1080
        gp.logpost(D_train, theta, y, sigma2, nu)  -> dict(val, beta, R_Inv)     # D1:609-641

    The design is n x 1.  Same device path as the other scripts (ccgp_set_kernel selects the family);
    prior and Jacobian are D1:636 = ISO:453."""

    def __init__(self, nu=5.0, handle=None, device=0):
        self.script = "D1"
        self.cfg = dict(aniso=False, prior=api.PRIOR_ISO, npar=3)
        self.nu = float(nu)
        base = handle if handle is not None else api.Handle(device)
        self.h = _FamilyHandle(base, api.KERNEL_MATERN, self.nu)

    def _with(self, nu):
        if nu is None or float(nu) == self.nu:
            return self
        return CombinedGP1D(nu, handle=self.h._handle)

    @staticmethod
    def _col(X):
        X = np.asarray(X, dtype=np.float64)
        return X.reshape(-1, 1)

    def corr_matrix(self, nu, X, theta):
        """D1:368-374 corr.matrix(nu, X, theta)."""
        return self._with(nu).h.corr_matrix(self._col(X), float(theta))

    def corr_vec(self, x, X, theta, nu=None):
        """D1:383-389 corr.vec(x, X, theta, nu)."""
        return self._with(nu).h.corr_cross(np.array([[float(x)]]), self._col(X), float(theta))[0]

    def Mixed_corr_matrix(self, D_train, p, theta1, theta2, nu=None):
        """D1:575-584."""
        D = self._col(D_train)
        return self._with(nu).h.mixed_corr_matrix(D, 2, pack_iso(p, theta1, theta2, 1))

    def Mixed_corr_vec(self, x_new, D_train, p, theta1, theta2, nu=None):
        """D1:591-599."""
        D = self._col(D_train)
        return self._with(nu).h.mixed_corr_cross(np.array([[float(x_new)]]), D, 2, pack_iso(p, theta1, theta2, 1))[0]

    def logpost(self, D_train, theta, y, sigma2, nu=None, want_R_Inv=True):
        """D1:609-641 logpost(D.train, theta, y, sigma2, nu) -> list(val, beta, R.Inv)."""
        g = self._with(nu)
        theta = np.asarray(theta, dtype=np.float64).ravel()
        if theta.size != 3:
            raise ValueError("D1 logpost takes 3 transformed parameters")
        r = g.h.logpost(self._col(D_train), y, sigma2, api.PRIOR_ISO, theta, None, want_Rinv=want_R_Inv)
        return dict(val=r["val"], beta=r["beta"], R_Inv=r["R_inv"] if r["status"] == 0 else None)

    def predict_post(self, x_new, D_train, pars, sigma2, nu=None):
        """D1:794-812: one frame row (p, theta1, theta2, beta, mean.factor, var.factor1, var.factor2, R.Inv)."""
        g = self._with(nu)
        D = self._col(D_train)
        n = D.shape[0]
        pars = np.asarray(pars, dtype=np.float64).ravel()
        r = g.Mixed_corr_vec(x_new, D, pars[0], pars[1], pars[2])
        R_Inv = pars[5 + 2 * n: 5 + 2 * n + n * n].reshape(n, n, order="F")
        mean, var = g.h.predict_from_factors(r.reshape(1, -1), pars[3], pars[4:4 + n], pars[4 + n:4 + 2 * n],
                                             pars[4 + 2 * n], R_Inv, sigma2)
        return np.array([[mean[0], var[0]]])

    def draws_to_params(self, D_train, draws):
        draws = np.atleast_2d(np.asarray(draws, dtype=np.float64))
        return np.stack([pack_iso(r[0], r[1], r[2], 1) for r in draws])

    def prediction_table(self, D_test, draws, D_train, sigma2, y_train):
        return super().prediction_table(self._col(D_test), draws, self._col(D_train), sigma2, y_train)

    def factors_frame_from_draws(self, draws, D_train, sigma2, y_train):
        return super().factors_frame_from_draws(draws, self._col(D_train), sigma2, y_train)

    # grids and entropy criteria exist only in the 2-D / emulator scripts
    def likeli_hyperpars(self, *a, **k):
        raise NotImplementedError("the 1-D script has no hyperprior grid")

    choose_hyperpars = Entropy = Entropy_batch = Augmented_Mixed_Entropy = likeli_hyperpars


class CombinedGP1DTwoFamilies(CombinedGP1D):
    """The two-family 1-D script's surface (1D Codes and Designs/1D Combined GP Two Families Public.R =
    D1F): component 1 Matern(nu, theta1), component 2 the non-negative cubic spline(theta2).

        gp = CombinedGP1DTwoFamilies(nu=5)
        gp.corr_matrix_combined(X, p, theta1, theta2, nu)          # D1F:453-462
        gp.corr_vec_combined(x, X, p, theta1, theta2, nu)          # D1F:470-480 -- NOT normalised, as written
        gp.logpost(D_train, theta, y, sigma2, nu)                  # D1F:576-601
    """

    def __init__(self, nu=5.0, handle=None, device=0):
        super().__init__(nu, handle, device)
        self.script = "D1"       # prior D1F:596 = D1:636
        self._base = self.h._handle
        self._matern = CombinedGP1D(nu, handle=self._base)
        self.h = _FamilyHandle(self._base, api.KERNEL_MATERN_SPLINE, self.nu)

    def _with(self, nu):
        if nu is None or float(nu) == self.nu:
            return self
        return CombinedGP1DTwoFamilies(nu, handle=self._base)

    # single-family pieces
    def corr_matrix_Matern(self, nu, X, theta):
        """D1F:426-431."""
        return self._matern.corr_matrix(nu, X, theta)

    def corr_vec_Matern(self, x, X, theta, nu=None):
        """D1F:439-444."""
        return self._matern.corr_vec(x, X, theta, nu)

    def corr_matrix_spline(self, X, theta):
        """D1F:398-404: the pair with weights (0, 1) is the spline alone."""
        return self.h.mixed_corr_matrix(self._col(X), 2, np.array([0.0, 1.0, 1.0, float(theta)]))

    def corr_vec_spline(self, x, X, theta):
        """D1F:412-418."""
        return self.h.mixed_corr_cross(np.array([[float(x)]]), self._col(X), 2, np.array([0.0, 1.0, 1.0, float(theta)]))[0]

    # the combined model
    def corr_matrix_combined(self, X, p, theta1, theta2, nu=None):
        """D1F:453-462."""
        return self._with(nu).h.mixed_corr_matrix(self._col(X), 2, pack_iso(p, theta1, theta2, 1))

    def corr_vec_combined(self, x, X, p, theta1, theta2, nu=None):
        """D1F:470-480: p^2 r1 + (1-p)^2 r2 -- the division by p^2 + (1-p)^2 is dead code in the script."""
        return self._with(nu).h.mixed_corr_cross(np.array([[float(x)]]), self._col(X), 2, pack_iso(p, theta1, theta2, 1))[0]

    Mixed_corr_matrix = corr_matrix_combined
    Mixed_corr_vec = corr_vec_combined

    def corr_matrix(self, *a, **k):
        raise AttributeError("the two-family script has corr.matrix.Matern / corr.matrix.spline / corr.matrix.combined")

    corr_vec = corr_matrix
