"""float64 numpy restatement of the SVI inner loop (oracle; test infrastructure).

PARITY UNPINNED: the reference contains no ELBO, sampler, gradient or update
code -- only the prose plan ``README.md:24-80`` and the exponential-family
decomposition contract ``bayesic/distribution/base.py:47-69`` (log-lik =
data term + interaction term - log-normaliser).  Each function cites the line
of that plan / the paper it restates, and is validated by tests against exact
conjugate posteriors, scipy.stats and finite differences.

Conventions shared with the HIP path (bayesic_amd/csrc):

* data arrays are float32 (``bayesic/algebra.py:109`` default dtype); every sum
  over rows is accumulated in float64 here;
* variational parameters are float64; Monte-Carlo weight samples are rounded to
  float32 before the data pass (the device streams float32 operands);
* ``scale = N_total / B`` multiplies every mini-batch sum (``README.md:69-79``,
  SVI ref [4]).
"""
import math

import numpy as np

from . import philox

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------
# Config 1: Gaussian with unknown mean and precision, Normal-Gamma node.
# Sufficient statistics of Normal (bayesic/distribution/core.py:16-17) summed
# over iid draws (bayesic/distribution/base.py:329-332).
# --------------------------------------------------------------------------

def normal_suffstats(x):
    """(count, sum x, sum x^2) in float64 from float32 data."""
    x64 = np.asarray(x, dtype=np.float32).astype(np.float64)
    return np.array([x64.size, x64.sum(), (x64 * x64).sum()], dtype=np.float64)


def normal_gamma_to_natural(mu0, kappa0, alpha0, beta0):
    """Natural parameters of NormalGamma(mu, tau) w.r.t. statistics
    (tau*mu, -tau*mu^2/2 (paired with kappa), log tau, -tau):

        eta = (kappa*mu, kappa, 2*alpha - 1, 2*beta + kappa*mu^2)

    chosen so that the conjugate update is a plain addition of
    (sum x, n, n, sum x^2) -- i.e. the VMP message from N iid Normal children
    (README.md:36, VIBES ref [1]).
    """
    return np.array([kappa0 * mu0, kappa0, 2.0 * alpha0 - 1.0,
                     2.0 * beta0 + kappa0 * mu0 * mu0], dtype=np.float64)


def normal_gamma_from_natural(eta):
    e1, e2, e3, e4 = (float(v) for v in eta)
    kappa = e2
    mu = e1 / kappa
    alpha = 0.5 * (e3 + 1.0)
    beta = 0.5 * (e4 - kappa * mu * mu)
    return mu, kappa, alpha, beta


def normal_gamma_message(stats):
    """Map summed Normal statistics (n, sx, sxx) to the natural-parameter
    increment of the Normal-Gamma parent."""
    n, sx, sxx = (float(v) for v in stats)
    return np.array([sx, n, n, sxx], dtype=np.float64)


def natgrad_update(eta, eta0, message, scale, rho):
    """SVI natural-gradient step (Hoffman et al. ref [4], README.md:36,75-77):

        eta <- (1 - rho) * eta + rho * (eta0 + scale * message)

    With rho = 1 and the full data set this is the exact VMP / conjugate update.
    """
    eta = np.asarray(eta, dtype=np.float64)
    return (1.0 - rho) * eta + rho * (np.asarray(eta0, np.float64) +
                                      scale * np.asarray(message, np.float64))


def normal_gamma_posterior_closed_form(x, mu0, kappa0, alpha0, beta0):
    """Textbook posterior (SURVEY.md 8(d) cfg 1) -- independent check."""
    x = np.asarray(x, dtype=np.float32).astype(np.float64)
    n = x.size
    xbar = x.mean()
    kappa = kappa0 + n
    mu = (kappa0 * mu0 + n * xbar) / kappa
    alpha = alpha0 + 0.5 * n
    beta = beta0 + 0.5 * ((x - xbar) ** 2).sum() + \
        kappa0 * n * (xbar - mu0) ** 2 / (2.0 * kappa)
    return mu, kappa, alpha, beta


# --------------------------------------------------------------------------
# Config 2: Bayesian linear regression, mean-field Gaussian q, reparam ELBO.
#
#   y_n ~ N(x_n . w, sigma^2),  w | sigma^2 ~ N(0, sigma^2 I),
#   sigma^2 ~ InvGamma(alpha0, beta0);   xi = log sigma^2
#   q(w) = N(m, diag exp(2 rho)),  q(xi) = N(a, exp(2 b))
#
# lam = [m (D), rho (D), a, b]  float64.
# --------------------------------------------------------------------------

def blr_sample(lam, D, S, seed, step=0):
    """Reparameterised draws (README.md:51 -> refs [10][11][12]).

    Returns eps [S, D+1] f64, W [S, D] float32 (rounded), xi [S] f64.
    eps[:, :D] from Philox stream 0, eps[:, D] from stream 1.
    """
    lam = np.asarray(lam, dtype=np.float64)
    m, rho, a, b = lam[:D], lam[D:2 * D], lam[2 * D], lam[2 * D + 1]
    eps_w = philox.normal_draws(seed, S, D, stream=0, step=step)
    eps_x = philox.normal_draws(seed, S, 1, stream=1, step=step)
    eps = np.concatenate([eps_w, eps_x], axis=1)
    W = (m[None, :] + np.exp(rho)[None, :] * eps_w).astype(np.float32)
    xi = a + math.exp(b) * eps_x[:, 0]
    return eps, W, xi


def blr_data_pass(X, y, W):
    """The one pass over the mini-batch.

    r[n, s] = y[n] - x[n] . W[s];  Q[s] = sum_n r^2;  G[s, :] = sum_n r[n, s] x[n].
    float32 operands, float64 arithmetic.  Returns (Q [S], G [S, D]).
    """
    X64 = np.asarray(X, dtype=np.float32).astype(np.float64)
    y64 = np.asarray(y, dtype=np.float32).astype(np.float64)
    W64 = np.asarray(W, dtype=np.float32).astype(np.float64)
    R = y64[:, None] - X64 @ W64.T
    return (R * R).sum(axis=0), R.T @ X64


def blr_data_pass_chunked(X, y, W, chunk=65536):
    """Same as ``blr_data_pass`` but bounded memory for 1M-row inputs."""
    S, D = W.shape
    Q = np.zeros(S)
    G = np.zeros((S, D))
    W64 = np.asarray(W, dtype=np.float32).astype(np.float64)
    for i in range(0, X.shape[0], chunk):
        Xc = np.asarray(X[i:i + chunk], dtype=np.float32).astype(np.float64)
        yc = np.asarray(y[i:i + chunk], dtype=np.float32).astype(np.float64)
        R = yc[:, None] - Xc @ W64.T
        Q += (R * R).sum(axis=0)
        G += R.T @ Xc
    return Q, G


def blr_log_joint(w, xi, Q, B, scale, alpha0=1.0, beta0=1.0):
    """f(w, xi) = scale * log p(y_B | w, xi) + log p(w | xi) + log p(xi)
    with Q = sum_n (y_n - x_n.w)^2.  Three-term decomposition per
    bayesic/distribution/base.py:47-69; Normal normaliser corrected."""
    D = w.shape[-1]
    e = np.exp(-xi)
    loglik = scale * (-0.5 * B * (LOG_2PI + xi) - 0.5 * e * Q)
    logpw = -0.5 * D * (LOG_2PI + xi) - 0.5 * e * (w * w).sum(axis=-1)
    logpxi = alpha0 * math.log(beta0) - math.lgamma(alpha0) - alpha0 * xi - beta0 * e
    return loglik + logpw + logpxi


def blr_elbo_and_grad(lam, eps, W, xi, Q, G, B, scale, alpha0=1.0, beta0=1.0):
    """Monte-Carlo ELBO estimate and its pathwise gradient w.r.t. lam.

    ELBO = mean_s f(w_s, xi_s) + H[q];  H = sum_d rho_d + b + (D+1)/2 log(2 pi e).
    d f/d w  = exp(-xi) (scale*G - w)
    d f/d xi = -(scale*B + D)/2 - alpha0 + exp(-xi) (scale*Q/2 + |w|^2/2 + beta0)
    (Kucukelbir et al. ref [11], eq. for mean-field Gaussian ADVI.)
    """
    lam = np.asarray(lam, dtype=np.float64)
    S, D = W.shape
    rho, b = lam[D:2 * D], lam[2 * D + 1]
    W64 = W.astype(np.float64)
    e = np.exp(-xi)
    f = blr_log_joint(W64, xi, Q, B, scale, alpha0, beta0)
    entropy = rho.sum() + b + 0.5 * (D + 1) * (1.0 + LOG_2PI)
    elbo = f.mean() + entropy
    dw = e[:, None] * (scale * G - W64)                       # [S, D]
    dxi = -0.5 * (scale * B + D) - alpha0 + \
        e * (0.5 * scale * Q + 0.5 * (W64 * W64).sum(axis=1) + beta0)
    grad = np.empty_like(lam)
    grad[:D] = dw.mean(axis=0)
    grad[D:2 * D] = (dw * eps[:, :D]).mean(axis=0) * np.exp(rho) + 1.0
    grad[2 * D] = dxi.mean()
    grad[2 * D + 1] = (dxi * eps[:, D]).mean() * math.exp(b) + 1.0
    return elbo, grad


def adam_ascent(lam, grad, m1, m2, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One Adam *ascent* step; t is the 1-based step count.  Returns new
    (lam, m1, m2)."""
    m1 = beta1 * m1 + (1.0 - beta1) * grad
    m2 = beta2 * m2 + (1.0 - beta2) * grad * grad
    mhat = m1 / (1.0 - beta1 ** t)
    vhat = m2 / (1.0 - beta2 ** t)
    return lam + lr * mhat / (np.sqrt(vhat) + eps), m1, m2


def blr_step(lam, m1, m2, t, X, y, S, seed, n_total, lr, alpha0=1.0, beta0=1.0,
             chunked=False, data_pass=None):
    """One full ELBO-gradient update on one mini-batch (sample -> pass ->
    gradient -> Adam).  ``t`` is the 1-based step index and also the Philox
    step counter (t - 1).  ``data_pass`` swaps in another restatement of the pass
    (oracle.cbuild.blr_data_pass).  Returns (lam, m1, m2, elbo, grad)."""
    B, D = X.shape
    eps, W, xi = blr_sample(lam, D, S, seed, step=t - 1)
    if data_pass is None:
        data_pass = blr_data_pass_chunked if chunked else blr_data_pass
    Q, G = data_pass(X, y, W)
    elbo, grad = blr_elbo_and_grad(lam, eps, W, xi, Q, G, B, n_total / B, alpha0, beta0)
    lam, m1, m2 = adam_ascent(lam, grad, m1, m2, t, lr)
    return lam, m1, m2, elbo, grad


def blr_exact_posterior(X, y, alpha0=1.0, beta0=1.0):
    """Exact Normal-Inverse-Gamma posterior for the prior w|s2 ~ N(0, s2 I),
    s2 ~ IG(alpha0, beta0): returns (mean_w, Lambda_n, a_n, b_n).  Uses the
    conjugate statistics XtX, Xty, yty (SURVEY 8(d) cfg 2')."""
    X64 = np.asarray(X, np.float32).astype(np.float64)
    y64 = np.asarray(y, np.float32).astype(np.float64)
    D = X64.shape[1]
    Lam = X64.T @ X64 + np.eye(D)
    mu = np.linalg.solve(Lam, X64.T @ y64)
    a_n = alpha0 + 0.5 * X64.shape[0]
    b_n = beta0 + 0.5 * (y64 @ y64 - mu @ Lam @ mu)
    return mu, Lam, a_n, b_n


def blr_conjugate_stats(X, y):
    """(XtX [D,D], Xty [D], yty) in float64 -- the lowered forms
    _tensordot(_dimshuffle(X,1,0), X,[1],[0]) etc.."""
    X64 = np.asarray(X, np.float32).astype(np.float64)
    y64 = np.asarray(y, np.float32).astype(np.float64)
    return X64.T @ X64, X64.T @ y64, float(y64 @ y64)


# --------------------------------------------------------------------------
# Synthetic inputs fixed by SURVEY.md 8(d).
# --------------------------------------------------------------------------

def make_cfg1(n=10000):
    return (2.0 + 1.5 * np.random.RandomState(1234).standard_normal(n)).astype(np.float32)


def make_cfg2(n=1000000, d=256):
    X = np.random.RandomState(1234).standard_normal((n, d)).astype(np.float32)
    w_true = (np.random.RandomState(1).standard_normal(d) / 16.0)
    y = (X.astype(np.float64) @ w_true +
         0.5 * np.random.RandomState(2).standard_normal(n)).astype(np.float32)
    return X, y, w_true


def blr_init_lam(D):
    """m = 0, rho = log 0.1, a = 0, b = log 0.1."""
    lam = np.zeros(2 * D + 2)
    lam[D:2 * D] = math.log(0.1)
    lam[2 * D + 1] = math.log(0.1)
    return lam


# --------------------------------------------------------------------------
# Config 3: mixture of Gaussians, diagonal precisions, discrete latent
# marginalised by summation (README.md:43,72), natural-gradient SVI on the
# global parameters (README.md:36,75-77; Hoffman et al. ref [4]).
#
#   z_n ~ Cat(pi), x_nd | z_n=k ~ N(mu_kd, 1/tau_kd)
#   pi ~ Dir(alpha0), (mu_kd, tau_kd) ~ NormalGamma(m0, kappa0, a0, b0)
#   q(pi) = Dir(alpha), q(mu_kd, tau_kd) = NormalGamma(m_kd, kappa_kd, a_kd, b_kd)
#
# Natural parameters (so that the conjugate update is additive):
#   eta_pi[k]      = alpha_k - 1                          <- + sum_n r_nk
#   eta1[k,d]      = kappa m                               <- + sum_n r_nk x_nd
#   eta2[k,d]      = kappa                                 <- + sum_n r_nk
#   eta3[k,d]      = 2a - 1                                <- + sum_n r_nk
#   eta4[k,d]      = 2b + kappa m^2                        <- + sum_n r_nk x_nd^2
# flat layout: [eta_pi (K) | eta1 (K*D) | eta2 (K*D) | eta3 (K*D) | eta4 (K*D)]
# --------------------------------------------------------------------------

def mog_prior_eta(K, D, alpha0=1.0, m0=0.0, kappa0=0.01, a0=1.0, b0=1.0):
    return np.concatenate([np.full(K, alpha0 - 1.0), np.full(K * D, kappa0 * m0),
                           np.full(K * D, kappa0), np.full(K * D, 2.0 * a0 - 1.0),
                           np.full(K * D, 2.0 * b0 + kappa0 * m0 * m0)])


def mog_unpack(eta, K, D):
    eta = np.asarray(eta, dtype=np.float64)
    alpha = eta[:K] + 1.0
    e1, e2, e3, e4 = (eta[K + i * K * D: K + (i + 1) * K * D].reshape(K, D) for i in range(4))
    kappa = e2
    m = e1 / kappa
    a = 0.5 * (e3 + 1.0)
    b = 0.5 * (e4 - kappa * m * m)
    return alpha, m, kappa, a, b


def mog_expected_params(eta, K, D, dtype=np.float32):
    """Coefficients of the per-row logits: logit_nk = c_k + sum_d (A_kd x_nd^2 + B_kd x_nd).
    Returns float32 Wmat [K, 2D] = [B | A] (x features first, then x^2) and c [K] -- rounded to
    ``dtype`` (float32: what the device streams)."""
    from scipy.special import digamma
    alpha, m, kappa, a, b = mog_unpack(eta, K, D)
    T = a / b                                            # E[tau]
    elog_pi = digamma(alpha) - digamma(alpha.sum())
    c = elog_pi + (0.5 * (digamma(a) - np.log(b)) - 0.5 * LOG_2PI
                   - 0.5 * T * m * m - 0.5 / kappa).sum(axis=1)
    Wmat = np.concatenate([T * m, -0.5 * T], axis=1)
    return Wmat.astype(dtype), c.astype(dtype)


def mog_local_step(X, Wmat, c, chunk=65536):
    """Marginalise z row by row.  float32 operands, float64 arithmetic.
    Returns stats [K, 1 + 2D] = [sum r | sum r x | sum r x^2] and
    sum_n logsumexp_k(logit_nk) (the local part of the bound)."""
    K, twoD = Wmat.shape
    D = twoD // 2
    W64, c64 = Wmat.astype(np.float64), c.astype(np.float64)
    stats = np.zeros((K, 1 + 2 * D))
    lse_total = 0.0
    for i in range(0, X.shape[0], chunk):
        Xc = np.asarray(X[i:i + chunk], dtype=np.float32).astype(np.float64)
        F = np.concatenate([Xc, Xc * Xc], axis=1)          # [n, 2D]
        logit = F @ W64.T + c64[None, :]
        mx = logit.max(axis=1, keepdims=True)
        e = np.exp(logit - mx)
        s = e.sum(axis=1, keepdims=True)
        R = e / s
        lse_total += float((mx[:, 0] + np.log(s[:, 0])).sum())
        stats[:, 0] += R.sum(axis=0)
        stats[:, 1:] += R.T @ F
    return stats, lse_total


def mog_message(stats, K, D):
    """Map summed statistics to the natural-parameter increment (flat layout)."""
    Rk, Sx, Sxx = stats[:, 0], stats[:, 1:1 + D], stats[:, 1 + D:]
    Rkd = np.repeat(Rk[:, None], D, axis=1)
    return np.concatenate([Rk, Sx.ravel(), Rkd.ravel(), Rkd.ravel(), Sxx.ravel()])


def mog_svi_step(eta, eta0, X, n_total, rho, K, D):
    """One SVI update on one mini-batch: local step (marginalise z), then
    eta <- (1-rho) eta + rho (eta0 + N/B * message)."""
    Wmat, c = mog_expected_params(eta, K, D)
    stats, lse = mog_local_step(X, Wmat, c)
    new = natgrad_update(eta, eta0, mog_message(stats, K, D), n_total / X.shape[0], rho)
    return new, stats, lse


def make_cfg3(n=10_000_000, d=16, k=64):
    centres = np.random.RandomState(3).standard_normal((k, d)) * 4.0
    labels = np.random.RandomState(4).randint(k, size=n)
    X = centres[labels] + np.random.RandomState(5).standard_normal((n, d))
    return X.astype(np.float32), centres, labels


def mog_init_eta(X_sample, K, D, seed=0, alpha0=1.0, kappa0=0.01, a0=1.0, b0=1.0):
    """Prior plus pseudo-observations at K randomly chosen rows (breaks symmetry)."""
    rs = np.random.RandomState(seed)
    picks = np.asarray(X_sample, np.float64)[rs.choice(len(X_sample), K, replace=False)]
    eta = mog_prior_eta(K, D, alpha0, 0.0, kappa0, a0, b0).reshape(-1)
    stats = np.zeros((K, 1 + 2 * D))
    stats[:, 0] = 1.0
    stats[:, 1:1 + D] = picks
    stats[:, 1 + D:] = picks * picks + 1.0
    return eta + mog_message(stats, K, D)


# --------------------------------------------------------------------------
# Config 5: hierarchical logistic regression, black-box VI with the
# score-function gradient and a control variate (README.md:52 -> ref [3]).
#
#   y_n ~ Bernoulli(sigmoid(x_n.w + b_{g_n})),  w_d ~ N(0,1),
#   b_g | tau ~ N(0, 1/tau),  tau ~ Gamma(a0, b0),  zeta = log tau
#   z = [w (D) | b (G) | zeta],  q(z) = N(mu, diag exp(2 rho)),  lam = [mu (P) | rho (P)]
# --------------------------------------------------------------------------

def bbvi_sample(lam, P, S, seed, step=0):
    """z_s = mu + exp(rho) * eps_s, eps from Philox stream 2.  Returns eps [S,P], z [S,P]."""
    lam = np.asarray(lam, dtype=np.float64)
    eps = philox.normal_draws(seed, S, P, stream=2, step=step)
    return eps, lam[:P][None, :] + np.exp(lam[P:])[None, :] * eps


def logreg_loglik(X, y, g, Wz, Bz, chunk=65536):
    """ell[s] = sum_n [ y_n l_ns - softplus(l_ns) ],  l_ns = x_n . Wz[s] + Bz[g_n, s].
    float32 operands, float64 arithmetic."""
    S = Wz.shape[0]
    W64, B64 = Wz.astype(np.float64), Bz.astype(np.float64)
    ell = np.zeros(S)
    for i in range(0, X.shape[0], chunk):
        Xc = np.asarray(X[i:i + chunk], np.float32).astype(np.float64)
        yc = np.asarray(y[i:i + chunk], np.float32).astype(np.float64)
        L = Xc @ W64.T + B64[np.asarray(g[i:i + chunk])]
        ell += (yc[:, None] * L - np.logaddexp(0.0, L)).sum(axis=0)
    return ell


def bbvi_log_prior(z, D, G, a0=1.0, b0=1.0):
    w, b, zeta = z[:, :D], z[:, D:D + G], z[:, D + G]
    lp_w = (-0.5 * LOG_2PI - 0.5 * w * w).sum(axis=1)
    lp_b = (-0.5 * LOG_2PI + 0.5 * zeta[:, None] - 0.5 * np.exp(zeta)[:, None] * b * b).sum(axis=1)
    lp_zeta = a0 * math.log(b0) - math.lgamma(a0) + a0 * zeta - b0 * np.exp(zeta)  # incl. Jacobian
    return lp_w + lp_b + lp_zeta


def bbvi_elbo_and_grad(lam, eps, ell, D, G, scale, a0=1.0, b0=1.0):
    """Score-function estimator with the scalar control variate of SURVEY 8(a) A14:
        f_s = scale*ell_s + log p(z_s) - log q(z_s)
        h_s = grad_lam log q(z_s) = [eps/sigma | eps^2 - 1]
        a   = sum_i Cov_s(f h_i, h_i) / sum_i Var_s(h_i)
        g   = mean_s (f_s - a) h_s ;   ELBO estimate = mean_s f_s."""
    lam = np.asarray(lam, dtype=np.float64)
    P = D + G + 1
    S = eps.shape[0]
    rho = lam[P:]
    z = lam[:P][None, :] + np.exp(rho)[None, :] * eps
    log_q = (-0.5 * LOG_2PI - rho[None, :] - 0.5 * eps * eps).sum(axis=1)
    f = scale * ell + bbvi_log_prior(z, D, G, a0, b0) - log_q
    h = np.concatenate([eps / np.exp(rho)[None, :], eps * eps - 1.0], axis=1)   # [S, 2P]
    fh = f[:, None] * h
    cov = ((fh - fh.mean(0)) * (h - h.mean(0))).sum(0) / (S - 1)
    var = ((h - h.mean(0)) ** 2).sum(0) / (S - 1)
    a = cov.sum() / var.sum()
    grad = ((f - a)[:, None] * h).mean(axis=0)
    return f.mean(), grad, a, f


def make_cfg5(n=1_000_000, d=256, n_groups=1000):
    X = np.random.RandomState(1234).standard_normal((n, d)).astype(np.float32)
    w_true = np.random.RandomState(1).standard_normal(d) / 16.0
    b_true = np.random.RandomState(7).standard_normal(n_groups) * 0.5
    g = np.random.RandomState(6).randint(n_groups, size=n).astype(np.int32)
    logits = X.astype(np.float64) @ w_true + b_true[g]
    y = (np.random.RandomState(8).uniform(size=n) < 1.0 / (1.0 + np.exp(-logits))).astype(np.float32)
    return X, y, g, w_true, b_true


def bbvi_init_lam(P):
    lam = np.zeros(2 * P)
    lam[P:] = math.log(0.05)
    return lam


def bbvi_step(lam, m1, m2, t, X, y, g, D, G, S, seed, n_total, lr):
    P = D + G + 1
    eps, z = bbvi_sample(lam, P, S, seed, step=t - 1)
    Wz = z[:, :D].astype(np.float32)
    Bz = np.ascontiguousarray(z[:, D:D + G].T).astype(np.float32)     # [G, S]
    ell = logreg_loglik(X, y, g, Wz, Bz)
    # the likelihood sees float32-rounded w, b (what the device streams); priors and log q use float64 z
    elbo, grad, a, f = bbvi_elbo_and_grad(lam, eps, ell, D, G, n_total / X.shape[0])
    lam, m1, m2 = adam_ascent(lam, grad, m1, m2, t, lr)
    return lam, m1, m2, elbo, grad, ell


# --------------------------------------------------------------------------
# Config 4: LDA-style Dirichlet-Multinomial, fixed-gamma local step (one
# iteration), natural-gradient SVI on lambda [K, V] (Hoffman, Blei, Bach 2010;
# README.md:69-79 -> ref [4]).  C [docs, V] are word counts.
#   Th = exp(E[log theta]) from gamma [docs, K];  Bt = exp(E[log beta]) from lambda
#   phinorm = Th Bt ;  sstats = Bt * (Th^T (C / phinorm))
#   lambda <- (1-rho) lambda + rho (eta + (docs_total / docs) * sstats)
# --------------------------------------------------------------------------

def dirichlet_expectation(alpha):
    from scipy.special import digamma
    a = np.asarray(alpha, dtype=np.float64)
    return np.exp(digamma(a) - digamma(a.sum(axis=1, keepdims=True)))


def lda_sstats(C, Th, Bt):
    C64, Th64, Bt64 = (np.asarray(v, np.float32).astype(np.float64) for v in (C, Th, Bt))
    phinorm = Th64 @ Bt64
    return Bt64 * (Th64.T @ (C64 / phinorm))


def lda_svi_step(lam, gamma, C, eta, docs_total, rho):
    Th = dirichlet_expectation(gamma).astype(np.float32)
    Bt = dirichlet_expectation(lam).astype(np.float32)
    ss = lda_sstats(C, Th, Bt)
    new = (1.0 - rho) * np.asarray(lam, np.float64) + rho * (eta + docs_total / C.shape[0] * ss)
    return new, ss


# --------------------------------------------------------------------------
# The evidence lower bound of configs 3 and 4 (README.md:30-37: "maximise a lower bound on
# the model evidence"; README.md:69-79: the mini-batch estimate of it; decomposition of every
# factor per bayesic/distribution/base.py:47-69 -- log p = <T, eta> - A(eta) + base).
#
# For a conjugate factor with prior natural parameters eta0 and variational eta (same family,
# same statistics T, the base measure cancels):
#
#     E_q[log p(theta)] - E_q[log q(theta)] = <eta0 - eta, E_q[T]> - A(eta0) + A(eta) = -KL(q || p)
#
# and for a finite discrete local latent held at its optimum (marginalised by summation,
# README.md:43,72) the local part is  sum_n logsumexp_k E_q[log p(x_n, z_n = k | theta)].
# --------------------------------------------------------------------------

def dirichlet_log_normalizer(alpha):
    """A = sum lnGamma(alpha_k) - lnGamma(sum alpha) along the last axis (statistics log theta_k,
    natural parameters alpha_k - 1)."""
    from scipy.special import gammaln
    a = np.asarray(alpha, np.float64)
    return gammaln(a).sum(axis=-1) - gammaln(a.sum(axis=-1))


def dirichlet_neg_kl(alpha, alpha0):
    """-KL(Dir(alpha) || Dir(alpha0)) per leading index; alpha0 broadcasts."""
    from scipy.special import digamma
    a = np.asarray(alpha, np.float64)
    a0 = np.broadcast_to(np.asarray(alpha0, np.float64), a.shape)
    elog = digamma(a) - digamma(a.sum(axis=-1, keepdims=True))
    return ((a0 - a) * elog).sum(axis=-1) - dirichlet_log_normalizer(a0) + dirichlet_log_normalizer(a)


def normal_gamma_log_normalizer(kappa, a, b):
    """A(eta) of NormalGamma(mu, tau | m, kappa, a, b) w.r.t. the statistics
    T = (tau mu, -tau mu^2 / 2, log(tau) / 2, -tau / 2) whose natural parameters are
    (kappa m, kappa, 2a - 1, 2b + kappa m^2) -- the layout of ``normal_gamma_to_natural``:
    A = lnGamma(a) - a log b - log(kappa) / 2 + log(2 pi) / 2."""
    from scipy.special import gammaln
    return gammaln(a) - a * np.log(b) - 0.5 * np.log(kappa) + 0.5 * LOG_2PI


def normal_gamma_expected_statistics(m, kappa, a, b):
    """E[T] for the statistics above: (m a/b, -(1/kappa + m^2 a/b)/2, (psi(a) - log b)/2, -a/(2b))."""
    from scipy.special import digamma
    T = a / b
    return T * m, -0.5 * (1.0 / kappa + m * m * T), 0.5 * (digamma(a) - np.log(b)), -0.5 * T


def mog_global_bound(eta, eta0, K, D):
    """E_q[log p(pi, mu, tau)] - E_q[log q(pi, mu, tau)] of config 3's global factors (a Dirichlet and
    K*D Normal-Gammas), natural parameters in the flat layout above."""
    eta, eta0 = np.asarray(eta, np.float64), np.asarray(eta0, np.float64)
    alpha, m, kappa, a, b = mog_unpack(eta, K, D)
    alpha0, _, kappa0, a0, b0 = mog_unpack(eta0, K, D)
    total = float(dirichlet_neg_kl(alpha, alpha0))
    ET = normal_gamma_expected_statistics(m, kappa, a, b)
    KD = K * D
    for j in range(4):
        e = eta[K + j * KD: K + (j + 1) * KD].reshape(K, D)
        e0 = eta0[K + j * KD: K + (j + 1) * KD].reshape(K, D)
        total += float(((e0 - e) * ET[j]).sum())
    total += float((normal_gamma_log_normalizer(kappa, a, b) - normal_gamma_log_normalizer(kappa0, a0, b0)).sum())
    return total


def mog_elbo(eta, eta0, lse_total, scale, K, D):
    """Mini-batch estimate of config 3's bound at q(theta) = eta with q(z) at its optimum:
    scale * sum_n logsumexp_k(logit_nk) + mog_global_bound.  ``lse_total`` is what
    ``mog_local_step`` returns for the same eta."""
    return scale * float(lse_total) + mog_global_bound(eta, eta0, K, D)


def lda_local_bound(C, Th, Bt):
    """sum_dv C_dv log(phinorm_dv), phinorm = Th Bt: the words' part of the bound with the
    per-word topic assignments at their optimum phi_dvk ~ Th_dk Bt_kv (Hoffman, Blei, Bach 2010,
    eq. 7 with the phi terms collapsed).  float32 operands, float64 arithmetic."""
    C64, Th64, Bt64 = (np.asarray(v, np.float32).astype(np.float64) for v in (C, Th, Bt))
    phinorm = Th64 @ Bt64
    nz = C64 != 0
    return float((C64[nz] * np.log(phinorm[nz])).sum())


def lda_elbo(lam, gamma, C, eta, alpha, docs_total):
    """Config 4's bound for one mini-batch of documents with fixed gamma:
    (docs_total / docs) * [ sum_dv C log phinorm - sum_d KL(Dir(gamma_d) || Dir(alpha)) ]
      - sum_k KL(Dir(lambda_k) || Dir(eta))."""
    Th = dirichlet_expectation(gamma).astype(np.float32)
    Bt = dirichlet_expectation(lam).astype(np.float32)
    docs, K = np.shape(gamma)
    scale = docs_total / docs
    local = lda_local_bound(C, Th, Bt) + float(dirichlet_neg_kl(gamma, alpha).sum())
    return scale * local + float(dirichlet_neg_kl(lam, eta).sum())
