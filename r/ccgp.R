# R-side drop-in for the Combined-GP hot path: re-defines the reference scripts' hot-path
# functions (same names, argument order and return shapes) on top of libccgp.
#
# Usage inside any of the reference scripts, AFTER its own function definitions and before
# its "Simulation starts here" driver block:
#
#     ccgp.script <- "HX"          # which script's variant: HX, GV, ISO, ADV, ANI, BSQ, D1 (1-D, Matern),
#                                  #   D1F (1-D, Matern + cubic spline)
#     Sys.setenv(CCGP_DEVICES = "8")   # optional: shard the batched calls over 8 GPUs of this node
#     source("r/ccgp.R")
#
# Everything that calls these functions (Metro, laplace via logpost.val, factors.frame,
# prediction, compare.GP, choose.hyperpars' callers) keeps working unchanged.
# Not executable in this repository's build container (no R); see INTEGRATION.md.

dyn.load(Sys.getenv("CCGP_R_SHIM", "ccgpR.so"))

if (!exists("ccgp.script")) ccgp.script <- "HX"
.ccgp.aniso <- ccgp.script == "ANI"
.ccgp.prior <- switch(ccgp.script, HX = 0L, ADV = 0L, GV = 1L, ISO = 2L, BSQ = 2L, ANI = 3L, D1 = 2L, D1F = 2L)

# numeric matrix with DOUBLE storage: the shim reads arguments with REAL(), which is an error on an integer
# matrix (a design or a hyperparameter table whose file holds whole numbers is read as integer)
.ccgp.mat <- function(x) { x <- as.matrix(x); storage.mode(x) <- "double"; x }

# (p, theta1, theta2[, lambda]) -> the C-ABI parameter row (w_1, w_2, theta_1k.., theta_2k..)
.ccgp.row <- function(d, p, theta1, theta2, lambda = NULL) {
  if (.ccgp.aniso) c(p, 1 - p, theta1, theta2, (1 + lambda) * theta1, (1 + lambda) * theta2)
  else c(p, 1 - p, rep(theta1, d), rep(theta2, d))
}

corr.matrix.ISO <- function(X, theta)
  .Call("ccgp_R_corr_matrix", .ccgp.mat(X), as.double(rep(theta, ncol(X))))

corr.vec.ISO <- function(x, X, theta)
  as.vector(.Call("ccgp_R_corr_cross", matrix(as.double(x), nrow = 1), .ccgp.mat(X),
                  as.double(rep(theta, ncol(X)))))

if (.ccgp.aniso) {
  corr.matrix <- function(X, theta1, theta2)
    .Call("ccgp_R_corr_matrix", .ccgp.mat(X), as.double(c(theta1, theta2)))
  corr.vec <- function(x, X, theta1, theta2)
    as.vector(.Call("ccgp_R_corr_cross", matrix(as.double(x), nrow = 1), .ccgp.mat(X),
                    as.double(c(theta1, theta2))))
  Mixed.corr.matrix <- function(D.train, p, theta1, theta2, lambda)
    .Call("ccgp_R_mixed_corr_matrix", .ccgp.mat(D.train), 2L, .ccgp.row(2, p, theta1, theta2, lambda))
  Mixed.corr.vec <- function(x.new, D.train, p, theta1, theta2, lambda)
    as.vector(.Call("ccgp_R_mixed_corr_cross", matrix(as.double(x.new), nrow = 1), .ccgp.mat(D.train),
                    2L, .ccgp.row(2, p, theta1, theta2, lambda)))
} else {
  corr.matrix <- function(X, theta)
    .Call("ccgp_R_corr_matrix", .ccgp.mat(X), as.double(theta))
  Mixed.corr.matrix <- function(D.train, p, theta1, theta2)
    .Call("ccgp_R_mixed_corr_matrix", .ccgp.mat(D.train), 2L, .ccgp.row(ncol(D.train), p, theta1, theta2))
  Mixed.corr.vec <- function(x.new, D.train, p, theta1, theta2)
    as.vector(.Call("ccgp_R_mixed_corr_cross", matrix(as.double(x.new), nrow = 1), .ccgp.mat(D.train),
                    2L, .ccgp.row(ncol(D.train), p, theta1, theta2)))
}

beta.MLE <- function(R.Inv, y) .Call("ccgp_R_beta_mle", .ccgp.mat(R.Inv), as.double(y))
sigma2.MLE <- function(R.Inv, y.train, beta) .Call("ccgp_R_sigma2_mle", .ccgp.mat(R.Inv), as.double(y.train), as.double(beta))

# logpost: HX / ADV pass the inverse-gamma hyperparameters, the other scripts do not
.ccgp.logpost <- function(D.train, theta, y, sigma2, pars) {
  r <- .Call("ccgp_R_logpost", .ccgp.mat(D.train), as.double(theta), as.double(y), as.double(sigma2),
             .ccgp.prior, pars)
  out <- list(val = r$val, beta = r$beta, R.Inv = r$R.Inv)
  if (ccgp.script == "ADV") out$like <- exp(r$loglik)
  out
}
if (ccgp.script %in% c("HX", "ADV")) {
  logpost <- function(D.train, theta, y, sigma2, theta1.pars, theta2.pars)
    .ccgp.logpost(D.train, theta, y, sigma2, as.double(c(theta1.pars[1:2], theta2.pars[1:2])))
} else {
  logpost <- function(D.train, theta, y, sigma2) .ccgp.logpost(D.train, theta, y, sigma2, NULL)
}

if (ccgp.script %in% c("HX", "ADV")) {
  .ccgp.N <- if (ccgp.script == "HX") 1000L else 1728L
  .ccgp.tau <- if (ccgp.script == "HX") 50 else 100
  likeli.hyperpars <- function(D.train, y.train, theta1.pars, theta2.pars, sigma2) {
    h <- matrix(as.double(c(theta1.pars[1:2], theta2.pars[1:2])), nrow = 1)
    .Call("ccgp_R_grid_marginal", .ccgp.mat(D.train), as.double(y.train), as.double(sigma2), h,
          .ccgp.N, .ccgp.tau, 0L, -1)[[1]]
  }
  choose.hyperpars <- function(D.train, y.train, hyperpars.matrix, sigma2) {
    r <- .Call("ccgp_R_grid_marginal", .ccgp.mat(D.train), as.double(y.train), as.double(sigma2),
               .ccgp.mat(hyperpars.matrix), .ccgp.N, .ccgp.tau,
               as.integer(ccgp.script == "HX"), -1)       # HX logs the mean (HX:591), ADV does not (ADV:595)
    list(pars = hyperpars.matrix[r[[2]], ], likelihoods = r[[1]])
  }
}

factors <- function(MCMC.data, n.train, y.train) {
  R.Inv <- matrix(as.numeric(MCMC.data[1:n.train^2]), nrow = n.train)
  .Call("ccgp_R_factors", R.Inv, as.numeric(MCMC.data[n.train^2 + 1]), as.double(y.train))
}

predict.post <- function(x.new, D.train, pars, sigma2) {
  n <- dim(D.train)[1]
  o <- if (.ccgp.aniso) 4 else 3
  pars <- as.numeric(pars)
  r <- if (.ccgp.aniso) Mixed.corr.vec(x.new, D.train, pars[1], pars[2], pars[3], pars[4])
       else if (ccgp.script == "ADV") Mixed.corr.vec(x.new, D.train, pars[1], pars[2], pars[2] * (1 + pars[3]))
       else Mixed.corr.vec(x.new, D.train, pars[1], pars[2], pars[3])
  out <- .Call("ccgp_R_predict_from_factors", matrix(r, nrow = 1), pars[o + 1],
               pars[(o + 2):(o + 1 + n)], pars[(o + 2 + n):(o + 1 + 2 * n)], pars[o + 2 + 2 * n],
               matrix(pars[(o + 3 + 2 * n):(o + 2 + 2 * n + n^2)], nrow = n), as.double(sigma2))
  colnames(out) <- c("mean", "var")
  out
}

# Batched replacement for the apply(pars.frame, 1, predict.post) inside prediction(): the
# whole (draw x test point) mean / variance table in one device call.
ccgp.prediction.table <- function(D.test, draws, D.train, sigma2, y.train) {
  d <- ncol(D.train)
  P <- t(apply(.ccgp.mat(draws), 1, function(r)
    if (.ccgp.aniso) .ccgp.row(d, r[1], r[2], r[3], r[4]) else .ccgp.row(d, r[1], r[2], r[3])))
  r <- .Call("ccgp_R_predict_batch", .ccgp.mat(D.train), as.double(y.train), 2L, P,
             .ccgp.mat(D.test), as.double(sigma2))
  list(mean = r[[1]], var = r[[2]], beta = r[[3]])
}

# Entropy criteria of the batch-sequential design script (Batch Sequential ME Design.R:856-877).
# det() of the Schur complement = det(R(D.old U D.new)) / det(R(D.old)), so R.old.Inv is not needed.
if (ccgp.script == "BSQ") {
  cross.corr.matrix <- function(D.old, D.new, theta)                                      # BSQ:835-848: n.new x n.old
    .Call("ccgp_R_corr_cross", .ccgp.mat(D.new), .ccgp.mat(D.old), as.double(rep(theta, ncol(D.new))))
  .ccgp.logdet <- function(D, p, theta1, theta2)
    .Call("ccgp_R_mixed_logdet_designs", matrix(as.double(D), ncol = 1), nrow(D), ncol(D), 2L,
          .ccgp.row(ncol(D), p, theta1, theta2))
  Entropy <- function(D, p, theta1, theta2) -exp(.ccgp.logdet(.ccgp.mat(D), p, theta1, theta2))
  Augmented.Mixed.Entropy <- function(D.old, D.new, p, theta1, theta2, R.old.Inv = NULL)
    -exp(.ccgp.logdet(rbind(.ccgp.mat(D.old), .ccgp.mat(D.new)), p, theta1, theta2) -
         .ccgp.logdet(.ccgp.mat(D.old), p, theta1, theta2))
}


# ---- 1-D script (1D Combined GP Public.R): Matern(nu) components, every function carries nu ---------
if (ccgp.script == "D1") {
  .ccgp.matern <- function(nu, expr) {          # select the family for one call, then back to Gaussian
    .Call("ccgp_R_set_kernel", 1L, as.double(nu))
    on.exit(.Call("ccgp_R_set_kernel", 0L, 0))
    force(expr)
  }
  corr.matrix <- function(nu, X, theta)                                                   # D1:368-374
    .ccgp.matern(nu, .Call("ccgp_R_corr_matrix", .ccgp.mat(X), as.double(theta)))
  corr.vec <- function(x, X, theta, nu)                                                   # D1:383-389
    .ccgp.matern(nu, as.vector(.Call("ccgp_R_corr_cross", matrix(as.double(x), nrow = 1), .ccgp.mat(X),
                                     as.double(theta))))
  Mixed.corr.matrix <- function(D.train, p, theta1, theta2, nu)                           # D1:575-584
    .ccgp.matern(nu, .Call("ccgp_R_mixed_corr_matrix", .ccgp.mat(D.train), 2L, c(p, 1 - p, theta1, theta2)))
  Mixed.corr.vec <- function(x.new, D.train, p, theta1, theta2, nu)                       # D1:591-599
    .ccgp.matern(nu, as.vector(.Call("ccgp_R_mixed_corr_cross", matrix(as.double(x.new), nrow = 1),
                                     .ccgp.mat(D.train), 2L, c(p, 1 - p, theta1, theta2))))
  logpost <- function(D.train, theta, y, sigma2, nu)                                      # D1:609-641
    .ccgp.matern(nu, .ccgp.logpost(D.train, theta, y, sigma2, NULL))
  predict.post <- function(x.new, D.train, pars, sigma2, nu) {                            # D1:794-812
    n <- dim(D.train)[1]
    pars <- as.numeric(pars)
    r <- Mixed.corr.vec(x.new, D.train, pars[1], pars[2], pars[3], nu = nu)
    out <- .Call("ccgp_R_predict_from_factors", matrix(r, nrow = 1), pars[4], pars[5:(4 + n)],
                 pars[(5 + n):(4 + 2 * n)], pars[5 + 2 * n],
                 matrix(pars[(6 + 2 * n):(5 + 2 * n + n^2)], nrow = n), as.double(sigma2))
    colnames(out) <- c("mean", "var")
    out
  }
}


# ---- two-family 1-D script (1D Combined GP Two Families Public.R): Matern(nu) + non-negative cubic spline ----
if (ccgp.script == "D1F") {
  .ccgp.two <- function(nu, expr) {             # family 2 = (Matern(nu, theta1), spline(theta2)) for one call
    .Call("ccgp_R_set_kernel", 2L, as.double(nu))
    on.exit(.Call("ccgp_R_set_kernel", 0L, 0))
    force(expr)
  }
  corr.matrix.spline <- function(X, theta)                                                # D1F:394-400
    .ccgp.two(5, .Call("ccgp_R_mixed_corr_matrix", .ccgp.mat(X), 2L, c(0, 1, 1, theta)))   # w = (0, 1): spline only
  corr.vec.spline <- function(x, X, theta)                                                # D1F:407-413
    .ccgp.two(5, as.vector(.Call("ccgp_R_mixed_corr_cross", matrix(as.double(x), nrow = 1), .ccgp.mat(X), 2L,
                                 c(0, 1, 1, theta))))
  corr.matrix.combined <- function(X, p, theta1, theta2, nu)                              # D1F:453-462
    .ccgp.two(nu, .Call("ccgp_R_mixed_corr_matrix", .ccgp.mat(X), 2L, c(p, 1 - p, theta1, theta2)))
  # D1F:470-480 as written: the division by p^2 + (1-p)^2 sits after the return, so r is NOT normalised;
  # the device keeps that (include/ccgp.h, CCGP_KERNEL_MATERN_SPLINE)
  corr.vec.combined <- function(x, X, p, theta1, theta2, nu)
    .ccgp.two(nu, as.vector(.Call("ccgp_R_mixed_corr_cross", matrix(as.double(x), nrow = 1), .ccgp.mat(X), 2L,
                                  c(p, 1 - p, theta1, theta2))))
  logpost <- function(D.train, theta, y, sigma2, nu)                                      # D1F:576-602
    .ccgp.two(nu, .ccgp.logpost(D.train, theta, y, sigma2, NULL))
  predict.post <- function(x.new, D.train, pars, sigma2, nu) {                            # D1F:737-754
    n <- dim(D.train)[1]
    pars <- as.numeric(pars)
    r <- corr.vec.combined(x.new, D.train, pars[1], pars[2], pars[3], nu)
    out <- .Call("ccgp_R_predict_from_factors", matrix(r, nrow = 1), pars[4], pars[5:(4 + n)],
                 pars[(5 + n):(4 + 2 * n)], pars[5 + 2 * n],
                 matrix(pars[(6 + 2 * n):(5 + 2 * n + n^2)], nrow = n), as.double(sigma2))
    colnames(out) <- c("mean", "var")
    out
  }
  # batched (draw x site) tables with this family: the device uses the same un-normalised r
  ccgp.prediction.table <- function(D.test, draws, D.train, sigma2, y.train, nu) {
    P <- t(apply(.ccgp.mat(draws), 1, function(r) c(r[1], 1 - r[1], r[2], r[3])))
    r <- .ccgp.two(nu, .Call("ccgp_R_predict_batch", .ccgp.mat(D.train), as.double(y.train), 2L, P,
                             .ccgp.mat(D.test), as.double(sigma2)))
    list(mean = r[[1]], var = r[[2]], beta = r[[3]])
  }
}
