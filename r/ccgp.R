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
#
# The prediction phase (compare.GP -> prediction -> apply(pars.frame, 1, predict.post): S x m calls per test set,
# HX:686-725, GV:620-676) runs as ONE device call: compare.GP and prediction are WRAPPED (the script's own
# definitions are kept and still do everything that is not the (draw x site) mean / variance table: rnorm, quantile,
# the CGP / mlegp comparators, plots).  Options, set before source():
#     ccgp.slim.frame <- TRUE      # default: logpost does not form / ship R.Inv, factors.frame's result is S x 7
#                                  #   instead of S x (5 + 2n + n^2); FALSE: the frame exactly as the script builds it
#     ccgp.adv.as.written <- FALSE # ADV only: TRUE keeps predict.post's theta1 * (1 + lambda) second scale (ADV:672)
#                                  #   through the literal per-draw path instead of the training kernel (ADV:417)
#     ccgp.metro.block <- 6        # Metro evaluates the next 6 iterations' 2^6 - 1 candidates in ONE device call
#                                  #   (same chain, same RNG stream: see Metro below); 1 = the script's own Metro, one logpost
#                                  #   per proposal
# Every index into a frame or a frame row is computed in r/ccgp_shim.c (executed by the test-suite), not here.
# Not executable in this repository's build container (no R); see INTEGRATION.md.

dyn.load(Sys.getenv("CCGP_R_SHIM", "ccgpR.so"))

if (!exists("ccgp.script")) ccgp.script <- "HX"
if (!exists("ccgp.slim.frame")) ccgp.slim.frame <- TRUE
if (!exists("ccgp.adv.as.written")) ccgp.adv.as.written <- FALSE
if (!exists("ccgp.metro.block")) ccgp.metro.block <- 6L
# ADV as written goes through the literal per-draw predict.post, which reads R.Inv and the factors from every frame row:
# a slim frame (7 numbers per row) would make every prediction NA, so that option implies the full frame
if (ccgp.script == "ADV" && ccgp.adv.as.written) ccgp.slim.frame <- FALSE
# how a draw's leading frame columns map to the parameter row (r/ccgp_shim.c: LAYOUT_*)
.ccgp.layout <- switch(ccgp.script, ANI = 2L, D1 = 3L, D1F = 4L, 0L)
.ccgp.layout.written <- if (ccgp.script == "ADV") 1L else .ccgp.layout       # predict.post as the script writes it
.ccgp.env <- new.env()               # y.train of the fit in progress (factors / compare.GP put it there)
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
             .ccgp.prior, pars, as.integer(!ccgp.slim.frame))      # slim: R.Inv is the 1 x 1 placeholder 0
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
  assign("y.train", as.double(y.train), envir = .ccgp.env)
  # slim frame: MCMC.data is (placeholder, beta); two placeholders keep factors.frame's t(apply(...)) an S x 2 block
  if (length(MCMC.data) < n.train^2 + 1) return(c(0, 0))
  R.Inv <- matrix(as.numeric(MCMC.data[1:n.train^2]), nrow = n.train)
  .Call("ccgp_R_factors", R.Inv, as.numeric(MCMC.data[n.train^2 + 1]), as.double(y.train))
}

# one frame row -> cbind(mean, var); the row is parsed in C (ccgp_R_predict_post), one device round trip per call
predict.post <- function(x.new, D.train, pars, sigma2, nu = 0) {
  out <- .Call("ccgp_R_predict_post", as.double(x.new), .ccgp.mat(D.train), as.numeric(pars), as.double(sigma2),
               .ccgp.layout.written, as.double(nu))
  colnames(out) <- c("mean", "var")
  out
}

# ---- the sampling phase on the batched path -----------------------------------------------------------------
# Metro (HX:483-540 and its per-script copies) asks for ONE logpost per proposal.  An iteration's random numbers --
# u <- runif(1) and the innovation rnorm(q) %*% chol(sqrt(2) V) that mnormt::rmnorm(1, theta.old, sqrt(2) V) adds to theta.old
# -- do not depend on whether the previous proposal was accepted, so a block of m iterations is pre-drawn in the script's
# order (runif(1), rnorm(q), runif(1), rnorm(q), ...), the 2^m - 1 candidates the chain can reach are evaluated in one device
# call and the accept / reject decisions are read off (r/ccgp_shim.c: ccgp_R_metro_steps).  The chain, the Geweke tests between
# iterations (coda, R code below as in the script) and the state of R's generator afterwards are those of the script's own
# loop: where the loop ends inside a block, .Random.seed is put back to the block's start and exactly the consumed pairs are
# drawn again.  Needs the slim frame (R.Inv per accepted draw is not formed); otherwise the script's Metro stays in place.
.ccgp.Metro <- function(start, N, samp.size, batch.size, alpha, D.train, sigma2, y, prior, logpost.one, ...) {
  pb <- txtProgressBar(min = 0, max = N, style = 3)
  est <- laplace(function(theta) logpost.one(theta)$val, start, ...)        # GV / ISO / ANI / BSQ hand `...` on to laplace
  U <- chol(sqrt(2) * est$var)                       # rmnorm's own factor: candidate = mean + rnorm(q) %*% chol(varcov)
  q <- length(start)
  samp <- matrix(0, ncol = q, nrow = N)
  beta <- list(); R.Inv <- list(); log.posterior <- list()
  theta.old <- as.vector(est$mode)
  l.old <- logpost.one(theta.old)
  state <- c(l.old$val, l.old$beta)
  k <- 1; pv <- 0
  X <- .ccgp.mat(D.train); yy <- as.double(y)
  while (k <= N & pv < alpha) {
    m <- as.integer(ccgp.metro.block)
    if (!exists(".Random.seed", envir = .GlobalEnv)) runif(1)
    seed0 <- get(".Random.seed", envir = .GlobalEnv)
    u <- numeric(m); E <- matrix(0, m, q)
    for (t in 1:m) { u[t] <- runif(1); E[t, ] <- drop(matrix(rnorm(q), 1, q) %*% U) }
    r <- .Call("ccgp_R_metro_steps", X, yy, as.double(sigma2), as.double(prior), as.double(theta.old), as.double(state), u, E)
    used <- 0L
    for (t in 1:m) {
      if (!(k <= N & pv < alpha)) break
      used <- used + 1L
      if (r$accepted[t]) {
        theta.old <- r$theta[t, ]
        samp[k, ] <- theta.old
        beta <- c(beta, r$beta[t]); R.Inv <- c(R.Inv, list(0))      # slim frame: the 1 x 1 placeholder (HX:520 stores R.Inv)
        log.posterior <- c(log.posterior, r$val[t])                 # BSQ:510 keeps the accepted values as well
        state <- c(r$val[t], r$beta[t])
        k <- k + 1
        setTxtProgressBar(pb, k)
      }
      if ((k - 1) >= samp.size & (k - 1) %% batch.size == 0) {      # HX:526-533, unchanged
        options(warn = 2)
        pv <- try(min(2 * (1 - pnorm(abs(geweke.diag(mcmc(samp[(k - samp.size):(k - 1)]))$z)))), silent = TRUE)
        if (inherits(pv, "try-error")) pv <- 0
      }
    }
    if (used < m) {                                  # the loop ended inside the block: R's generator as the script leaves it
      assign(".Random.seed", seed0, envir = .GlobalEnv)
      for (t in seq_len(used)) { runif(1); rnorm(q) }
    }
  }
  close(pb)
  list(sample = data.frame(samp[(k - samp.size):(k - 1), ]), beta = beta[(k - samp.size):(k - 1)],
       R.Inv = R.Inv[(k - samp.size):(k - 1)], logpost = log.posterior[(k - samp.size):(k - 1)])    # `logpost`: BSQ:529 only
}
if (ccgp.slim.frame && ccgp.metro.block > 1 && !(ccgp.script %in% c("D1", "D1F"))) {
  if (ccgp.script %in% c("HX", "ADV")) {
    Metro <- function(start, N, samp.size, batch.size, alpha, D.train, sigma2, y, theta1.pars, theta2.pars)
      .ccgp.Metro(start, N, samp.size, batch.size, alpha, D.train, sigma2, y,
                  c(.ccgp.prior, theta1.pars[1:2], theta2.pars[1:2]),
                  function(theta) logpost(D.train, theta, y, sigma2, theta1.pars, theta2.pars))
  } else {
    Metro <- function(start, N, samp.size, batch.size, alpha, D.train, sigma2, y, ...)
      .ccgp.Metro(start, N, samp.size, batch.size, alpha, D.train, sigma2, y, .ccgp.prior,
                  function(theta) logpost(D.train, theta, y, sigma2), ...)
  }
}

# ---- the prediction phase as one device call -----------------------------------------------------------------
# arguments of a call, matched to the formals of the script's own definition (the argument lists differ per script:
# HX:686 prediction(x.new, alpha, pars.frame, D.train, sigma2), ANI:637 adds code, D1:825 adds nu, ...)
.ccgp.bind <- function(f, ...) as.list(match.call(f, as.call(c(list(as.name("f")), list(...)))))[-1]
.ccgp.nu <- function(a) if (is.null(a$nu)) 0 else as.double(a$nu)

# the frame carries the y.train it was built from: prediction() outside compare.GP refactorises a slim frame against THAT
# vector, not against whatever fit ran last in this session
if (exists("factors.frame") && !exists(".ccgp.ref.factors.frame")) .ccgp.ref.factors.frame <- factors.frame
if (exists(".ccgp.ref.factors.frame")) {
  factors.frame <- function(...) {
    out <- .ccgp.ref.factors.frame(...)
    attr(out, "ccgp.y.train") <- get("y.train", envir = .ccgp.env)      # factors() put it there, row by row
    out
  }
}
.ccgp.frame.y <- function(frame, n) {
  y <- attr(frame, "ccgp.y.train")
  if (is.null(y) && exists("y.train", envir = .ccgp.env)) {
    y <- get("y.train", envir = .ccgp.env)
    warning("ccgp: the frame does not carry its y.train (built before source(\"r/ccgp.R\")?); using the last fit's")
  }
  if (is.null(y) || length(y) != n) NULL else y
}

if (exists("compare.GP") && !exists(".ccgp.ref.compare.GP")) .ccgp.ref.compare.GP <- compare.GP
if (exists("prediction") && !exists(".ccgp.ref.prediction")) .ccgp.ref.prediction <- prediction

if (exists(".ccgp.ref.compare.GP")) {
  # the whole (draw x test site) table once, kept on the C side while the script's compare.GP walks over D.test
  compare.GP <- function(...) {
    a <- .ccgp.bind(.ccgp.ref.compare.GP, ...)
    D.new <- if (is.null(a$D.test)) a$D.new else a$D.test
    assign("y.train", as.double(a$y.train), envir = .ccgp.env)
    if (!(ccgp.script == "ADV" && ccgp.adv.as.written)) {
      .Call("ccgp_R_table_cache", a$params, .ccgp.mat(a$D.train), .ccgp.mat(D.new), as.double(a$sigma2),
            as.double(a$y.train), .ccgp.layout, .ccgp.nu(a))
      on.exit(.Call("ccgp_R_table_clear"))
    }
    .ccgp.ref.compare.GP(...)
  }
}

if (exists(".ccgp.ref.prediction")) {
  # the script's own prediction() with its apply(pars.frame, 1, predict.post, ...) answered from the table: the
  # 2 x S block (rows mean, var) that apply would have returned; everything after that line (rnorm, quantile, ...)
  # is the script's code, untouched
  prediction <- function(...) {
    a <- .ccgp.bind(.ccgp.ref.prediction, ...)
    S <- nrow(a$pars.frame)
    tab <- .Call("ccgp_R_table_lookup", as.double(a$x.new), S)
    yt <- if (is.null(tab) && ncol(a$pars.frame) < nrow(a$D.train)^2) .ccgp.frame.y(a$pars.frame, nrow(a$D.train)) else NULL
    if (is.null(tab) && !(ccgp.script == "ADV" && ccgp.adv.as.written) && !is.null(yt)) {
      # outside compare.GP with a slim frame: the table of this one site
      r <- .Call("ccgp_R_prediction_table", a$pars.frame, .ccgp.mat(a$D.train), matrix(as.double(a$x.new), nrow = 1),
                 as.double(a$sigma2), as.double(yt), .ccgp.layout, .ccgp.nu(a))
      tab <- rbind(as.vector(r[[1]]), as.vector(r[[2]]))
    }
    if (is.null(tab)) return(.ccgp.ref.prediction(...))          # full frame, unknown site: the literal path
    f <- .ccgp.ref.prediction
    environment(f) <- list2env(list(apply = function(X, MARGIN, FUN, ...) tab), parent = environment(.ccgp.ref.prediction))
    f(...)
  }
}

# The (draw x site) tables directly: draws = a factors.frame() result or any matrix / data frame whose leading
# columns are p, theta1, theta2[, lambda]
ccgp.prediction.table <- function(D.test, draws, D.train, sigma2, y.train, nu = 0) {
  r <- .Call("ccgp_R_prediction_table", draws, .ccgp.mat(D.train), .ccgp.mat(D.test), as.double(sigma2),
             as.double(y.train), .ccgp.layout, as.double(nu))
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
  # predict.post (D1:794-812), prediction and compare.GP: the generic definitions above (layout 3, nu passed on)
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
  # predict.post (D1F:737-754, with corr.vec.combined's un-normalised r), prediction, compare.GP and
  # ccgp.prediction.table: the generic definitions above (layout 4, nu passed on)
}
