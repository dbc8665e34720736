"""CPU checks that pin the float64 oracle (no GPU, no HIP library).

The ELBO path has no reference implementation ("parity unpinned"), so the oracle
is validated against known answers: Random123 Philox vectors, exact conjugate
posteriors, scipy.stats densities and finite differences.
"""
import math

import numpy as np
import pytest
import scipy.stats as st

from oracle import philox, svi


def _hex(words):
    return ["%08x" % w for w in words]


def test_philox4x32_10_random123_known_answers():
    # Random123 kat_vectors for philox4x32 10 rounds
    z4, z2 = np.zeros(4, np.uint32), np.zeros(2, np.uint32)
    assert _hex(philox.philox4x32_10(z4, z2)) == ["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]
    f4, f2 = np.full(4, 0xFFFFFFFF, np.uint32), np.full(2, 0xFFFFFFFF, np.uint32)
    assert _hex(philox.philox4x32_10(f4, f2)) == ["408f276d", "41c83b0e", "a20bc7c6", "6d5451fd"]
    c = np.array([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], np.uint32)
    k = np.array([0xA4093822, 0x299F31D0], np.uint32)
    assert _hex(philox.philox4x32_10(c, k)) == ["d16cfe09", "94fdcceb", "5001e420", "24126ea1"]


def test_normal_draws_are_standard_normal_and_keyed():
    z = philox.normal_draws(1234, 64, 4096)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3
    assert st.kstest(z.ravel()[:20000], "norm").pvalue > 1e-3
    # keyed by (sample, parameter): a prefix of a bigger draw is the same draw
    z2 = philox.normal_draws(1234, 8, 257)
    np.testing.assert_array_equal(z2, z[:8, :257])
    assert not np.allclose(philox.normal_draws(1234, 8, 8, step=1), z[:8, :8])
    assert not np.allclose(philox.normal_draws(1234, 8, 8, stream=1), z[:8, :8])


def test_gaussian_gamma_unit_step_natural_gradient_is_exact_posterior():
    # README.md:36 -- VMP == unit-step natural gradient; SURVEY 8(d) cfg 1
    x = svi.make_cfg1()
    eta0 = svi.normal_gamma_to_natural(0.0, 1.0, 1.0, 1.0)
    msg = svi.normal_gamma_message(svi.normal_suffstats(x))
    eta = svi.natgrad_update(eta0, eta0, msg, scale=1.0, rho=1.0)
    got = svi.normal_gamma_from_natural(eta)
    want = svi.normal_gamma_posterior_closed_form(x, 0.0, 1.0, 1.0, 1.0)
    np.testing.assert_allclose(got, want, rtol=1e-12)
    # round trip of the parameterisation
    np.testing.assert_allclose(svi.normal_gamma_from_natural(eta0), (0.0, 1.0, 1.0, 1.0))


def test_minibatch_natural_gradient_converges_to_posterior():
    x = svi.make_cfg1()
    eta0 = svi.normal_gamma_to_natural(0.0, 1.0, 1.0, 1.0)
    eta = eta0.copy()
    rs = np.random.RandomState(0)
    for t in range(1, 400):
        batch = x[rs.randint(0, x.size, 500)]
        msg = svi.normal_gamma_message(svi.normal_suffstats(batch))
        eta = svi.natgrad_update(eta, eta0, msg, scale=x.size / 500.0, rho=(t + 1.0) ** -0.7)
    mu, kappa, alpha, beta = svi.normal_gamma_from_natural(eta)
    wmu, wk, wa, wb = svi.normal_gamma_posterior_closed_form(x, 0.0, 1.0, 1.0, 1.0)
    # stochastic: mini-batch noise leaves O(rho_t) jitter around the exact posterior
    assert abs(mu - wmu) < 0.05 and abs(kappa / wk - 1) < 1e-4 and abs(beta / alpha - wb / wa) < 0.05


def test_blr_log_joint_matches_scipy():
    X, y, _ = svi.make_cfg2(300, 4)
    rs = np.random.RandomState(3)
    w, xi = rs.standard_normal(4) * 0.1, 0.3
    s2 = math.exp(xi)
    Q = float(((y.astype(float) - X.astype(float) @ w) ** 2).sum())
    got = svi.blr_log_joint(w, np.array(xi), Q, 300, 1.0, alpha0=1.5, beta0=0.7)
    want = st.norm.logpdf(y.astype(float), X.astype(float) @ w, math.sqrt(s2)).sum() \
        + st.norm.logpdf(w, 0.0, math.sqrt(s2)).sum() \
        + st.invgamma.logpdf(s2, 1.5, scale=0.7) + xi     # + log|d s2 / d xi|
    assert abs(got - want) < 1e-8 * abs(want)


def test_blr_pathwise_gradient_matches_finite_differences():
    X, y, _ = svi.make_cfg2(500, 8)
    D, S, B, scale = 8, 4, 500, 3.0
    lam = svi.blr_init_lam(D) + 0.01 * np.arange(2 * D + 2)

    def elbo_fixed_noise(lam):
        eps = np.concatenate([philox.normal_draws(7, S, D, 0), philox.normal_draws(7, S, 1, 1)], 1)
        W = lam[:D][None] + np.exp(lam[D:2 * D])[None] * eps[:, :D]
        xi = lam[2 * D] + math.exp(lam[2 * D + 1]) * eps[:, D]
        R = y.astype(float)[:, None] - X.astype(float) @ W.T
        Q, G = (R * R).sum(0), R.T @ X.astype(float)
        f = svi.blr_log_joint(W, xi, Q, B, scale)
        ent = lam[D:2 * D].sum() + lam[2 * D + 1] + 0.5 * (D + 1) * (1 + svi.LOG_2PI)
        return f.mean() + ent, (eps, W, xi, Q, G)

    e0, (eps, W, xi, Q, G) = elbo_fixed_noise(lam)
    elbo, grad = svi.blr_elbo_and_grad(lam, eps, W.astype(np.float64), xi, Q, G, B, scale)
    assert abs(elbo - e0) < 1e-9 * abs(e0)
    h = 1e-6
    fd = np.array([(elbo_fixed_noise(lam + h * e)[0] - elbo_fixed_noise(lam - h * e)[0]) / (2 * h)
                   for e in np.eye(2 * D + 2)])
    np.testing.assert_allclose(grad, fd, rtol=1e-6, atol=1e-6 * np.abs(grad).max())


def test_blr_svi_approaches_exact_posterior_mean():
    X, y, w_true = svi.make_cfg2(4000, 8)
    D = 8
    lam, m1, m2 = svi.blr_init_lam(D), np.zeros(2 * D + 2), np.zeros(2 * D + 2)
    for t in range(1, 801):
        lam, m1, m2, elbo, _ = svi.blr_step(lam, m1, m2, t, X, y, 8, 1234, 4000, 0.02)
    mu, Lam, a_n, b_n = svi.blr_exact_posterior(X, y)
    np.testing.assert_allclose(lam[:D], mu, atol=0.02)
    # E_q[log s2] should sit near the posterior's E[log s2] = log b_n - digamma(a_n)
    from scipy.special import digamma
    assert abs(lam[2 * D] - (math.log(b_n) - digamma(a_n))) < 0.1


def test_chunked_pass_equals_plain_pass():
    X, y, _ = svi.make_cfg2(1000, 16)
    W = np.random.RandomState(0).standard_normal((8, 16)).astype(np.float32)
    Q1, G1 = svi.blr_data_pass(X, y, W)
    Q2, G2 = svi.blr_data_pass_chunked(X, y, W, chunk=128)
    np.testing.assert_allclose(Q1, Q2, rtol=1e-12)
    np.testing.assert_allclose(G1, G2, rtol=1e-10, atol=1e-9)


def test_product_side_mog_helpers_match_the_oracle():
    """bayesic_amd.svi.mog.prior_eta / init_eta / message / unpack (what a user builds the
    driver's inputs with) against the oracle's layout."""
    from bayesic_amd.svi import mog
    rs = np.random.RandomState(0)
    K, D = 5, 3
    np.testing.assert_array_equal(mog.prior_eta(K, D, alpha0=1.5, kappa0=0.1),
                                  svi.mog_prior_eta(K, D, alpha0=1.5, kappa0=0.1))
    X = rs.standard_normal((50, D))
    np.testing.assert_allclose(mog.init_eta(X, K, D, seed=4), svi.mog_init_eta(X, K, D, seed=4))
    stats = rs.rand(K, 1 + 2 * D)
    np.testing.assert_array_equal(mog.message(stats, K, D), svi.mog_message(stats, K, D))
    eta = mog.init_eta(X, K, D, seed=1)
    for a, b in zip(mog.unpack(eta, K, D), svi.mog_unpack(eta, K, D)):
        np.testing.assert_allclose(a, b)


# ---- the bound of configs 3 and 4 (oracle.svi.mog_elbo / lda_elbo) --------------------------------

def test_dirichlet_neg_kl_against_quadrature():
    # K = 2: a Dirichlet is a Beta; integrate q (log p - log q) with scipy's densities
    from scipy.integrate import quad
    for (a, b), (a0, b0) in (((2.5, 1.7), (1.0, 1.0)), ((0.8, 3.0), (0.5, 0.5)), ((40.0, 7.0), (2.0, 3.0))):
        q, p = st.beta(a, b), st.beta(a0, b0)
        want, _ = quad(lambda x: q.pdf(x) * (p.logpdf(x) - q.logpdf(x)), 0.0, 1.0, epsabs=1e-12, epsrel=1e-12,
                       limit=400)
        got = svi.dirichlet_neg_kl(np.array([a, b]), np.array([a0, b0]))
        np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-9)
    # the entropy term alone against scipy's closed form: -KL(q || uniform) = H[q] + log density of the uniform, ln Gamma(K)
    alpha = np.array([0.7, 2.0, 3.5, 1.1])
    np.testing.assert_allclose(svi.dirichlet_neg_kl(alpha, np.ones(4)),
                               st.dirichlet(alpha).entropy() + math.lgamma(4.0), rtol=1e-12)


def test_normal_gamma_bound_against_quadrature():
    m, kappa, a, b = 0.7, 2.5, 3.0, 1.5
    m0, kappa0, a0, b0 = -0.2, 0.4, 1.5, 0.8

    def logpdf(mu, tau, m_, k_, a_, b_):     # scipy's densities: N(mu | m, 1/(kappa tau)) Gamma(tau | a, rate b)
        return st.norm.logpdf(mu, m_, 1.0 / np.sqrt(k_ * tau)) + st.gamma.logpdf(tau, a_, scale=1.0 / b_)

    # E_q over mu | tau by Gauss-Hermite (the integrand is quadratic in mu: exact), over u = log tau by the
    # trapezoid rule on a smooth, doubly decaying integrand (spectrally convergent)
    x, w = np.polynomial.hermite.hermgauss(40)
    u = np.linspace(-40.0, 6.0, 20001)
    tau = np.exp(u)[:, None]
    mu = m + np.sqrt(2.0) * x[None, :] / np.sqrt(kappa * tau)
    inner = ((logpdf(mu, tau, m0, kappa0, a0, b0) - logpdf(mu, tau, m, kappa, a, b)) * w[None, :]).sum(axis=1) / math.sqrt(math.pi)
    dens = np.exp(st.gamma.logpdf(tau[:, 0], a, scale=1.0 / b) + u)          # q(tau) d tau = q(tau) tau du
    want = float(((dens * inner)[1:] + (dens * inner)[:-1]).sum() * 0.5 * (u[1] - u[0]))
    eta = svi.normal_gamma_to_natural(m, kappa, a, b)
    eta0 = svi.normal_gamma_to_natural(m0, kappa0, a0, b0)
    ET = svi.normal_gamma_expected_statistics(m, kappa, a, b)
    got = sum((eta0[j] - eta[j]) * ET[j] for j in range(4)) \
        + svi.normal_gamma_log_normalizer(kappa, a, b) - svi.normal_gamma_log_normalizer(kappa0, a0, b0)
    np.testing.assert_allclose(got, want, rtol=1e-7)
    # the expected statistics are the gradient of the log-normaliser (exponential family identity)
    def A_of(e):
        mm, kk, aa, bb = svi.normal_gamma_from_natural(e)
        return svi.normal_gamma_log_normalizer(kk, aa, bb)
    for j in range(4):
        h = 1e-6 * max(1.0, abs(eta[j]))
        ep, em = eta.copy(), eta.copy()
        ep[j] += h
        em[j] -= h
        np.testing.assert_allclose((A_of(ep) - A_of(em)) / (2 * h), ET[j], rtol=1e-6)


def _small_mixture(n=600, d=3, k=4, seed=0):
    rs = np.random.RandomState(seed)
    centres = rs.standard_normal((k, d)) * 3.0
    X = (centres[rs.randint(k, size=n)] + rs.standard_normal((n, d))).astype(np.float32)
    eta0 = svi.mog_prior_eta(k, d)
    return X, eta0, svi.mog_init_eta(X, k, d, seed=1)


def _mog_bound_at(eta, eta0, X, scale, K, D, dtype=np.float64):
    Wmat, c = svi.mog_expected_params(eta, K, D, dtype=dtype)
    _, lse = svi.mog_local_step(X, Wmat, c)
    return svi.mog_elbo(eta, eta0, lse, scale, K, D)


def test_mog_elbo_rises_under_full_batch_unit_steps():
    # rho = 1 on the full data set is coordinate ascent (README.md:36): q(z) at its optimum, then both
    # global factors at theirs -- the bound can only go up
    K, D = 4, 3
    X, eta0, eta = _small_mixture()
    bounds = []
    for _ in range(12):
        bounds.append(_mog_bound_at(eta, eta0, X, 1.0, K, D))
        eta, _, _ = svi.mog_svi_step(eta, eta0, X, float(len(X)), 1.0, K, D)
    bounds = np.array(bounds)
    assert np.all(np.diff(bounds) >= -1e-6 * np.abs(bounds[:-1])), bounds   # (float32 logit coefficients inside the step)
    assert bounds[-1] > bounds[0] + 10.0


def test_mog_elbo_gradient_is_fisher_times_natural_gradient():
    # conjugate-exponential identity (Hoffman et al. ref [4], eq. 14): d L / d eta = A''(eta) (eta* - eta) with
    # eta* = eta0 + scale * message at the optimal q(z); by the envelope theorem q(z) may follow eta
    K, D = 3, 2
    X, eta0, eta = _small_mixture(n=200, d=D, k=K, seed=3)
    scale = 2.5
    f = lambda e: _mog_bound_at(e, eta0, X, scale, K, D)

    def A_total(e):
        alpha, m, kappa, a, b = svi.mog_unpack(e, K, D)
        return float(svi.dirichlet_log_normalizer(alpha) + svi.normal_gamma_log_normalizer(kappa, a, b).sum())

    Wmat, c = svi.mog_expected_params(eta, K, D, dtype=np.float64)
    stats, _ = svi.mog_local_step(X, Wmat, c)
    direction = eta0 + scale * svi.mog_message(stats, K, D) - eta
    n = eta.size
    h = 1e-4
    grad = np.zeros(n)
    for i in range(n):
        ep, em = eta.copy(), eta.copy()
        ep[i] += h
        em[i] -= h
        grad[i] = (f(ep) - f(em)) / (2 * h)
    # Fisher . direction by differencing the log-normaliser's gradient along the direction
    def gradA(e):
        g = np.zeros(n)
        for i in range(n):
            ep, em = e.copy(), e.copy()
            ep[i] += h
            em[i] -= h
            g[i] = (A_total(ep) - A_total(em)) / (2 * h)
        return g
    t = 1e-4 * float(np.min(np.abs(eta) / np.maximum(np.abs(direction), 1e-300)))   # stay inside the domain
    fisher_dir = (gradA(eta + t * direction) - gradA(eta - t * direction)) / (2 * t)
    np.testing.assert_allclose(grad, fisher_dir, rtol=2e-4, atol=2e-4 * np.abs(fisher_dir).max())


def _small_lda(docs=40, V=60, K=5, seed=0):
    rs = np.random.RandomState(seed)
    C = rs.poisson(0.4, size=(docs, V)).astype(np.float32)
    gamma = rs.gamma(2.0, 1.0, size=(docs, K)).astype(np.float32) + 0.1
    lam = rs.gamma(1.0, 1.0, size=(K, V)) + 0.05
    return C, gamma, lam


def test_lda_elbo_equals_the_bound_with_explicit_assignments():
    # Hoffman, Blei, Bach (2010) eq. 7 written out with phi_dvk, against the collapsed form
    from scipy.special import digamma, gammaln
    C, gamma, lam = _small_lda()
    eta, alpha, docs_total = 0.02, 0.3, 400.0
    docs, K = gamma.shape
    V = lam.shape[1]
    Th = svi.dirichlet_expectation(gamma).astype(np.float32).astype(np.float64)
    Bt = svi.dirichlet_expectation(lam).astype(np.float32).astype(np.float64)
    elt, elb = np.log(Th), np.log(Bt)              # E[log theta], E[log beta] as the pass sees them
    phi = Th[:, None, :] * Bt.T[None, :, :]        # [docs, V, K]
    phi /= phi.sum(axis=2, keepdims=True)
    C64 = C.astype(np.float64)
    words = (C64[:, :, None] * phi * (elt[:, None, :] + elb.T[None, :, :] - np.log(phi))).sum()
    g64, l64 = gamma.astype(np.float64), lam
    elt_exact = digamma(g64) - digamma(g64.sum(1, keepdims=True))
    elb_exact = digamma(l64) - digamma(l64.sum(1, keepdims=True))
    theta = ((alpha - g64) * elt_exact).sum() + gammaln(g64).sum() - gammaln(g64.sum(1)).sum() \
        + docs * (gammaln(K * alpha) - K * gammaln(alpha))
    beta = ((eta - l64) * elb_exact).sum() + gammaln(l64).sum() - gammaln(l64.sum(1)).sum() \
        + K * (gammaln(V * eta) - V * gammaln(eta))
    want = docs_total / docs * (words + theta) + beta
    np.testing.assert_allclose(svi.lda_elbo(lam, gamma, C, eta, alpha, docs_total), want, rtol=1e-12)


def test_lda_elbo_rises_under_full_batch_unit_steps():
    C, gamma, lam = _small_lda(docs=60, V=80, K=6, seed=2)
    eta, alpha = 0.05, 0.2
    bounds = []
    for _ in range(10):
        bounds.append(svi.lda_elbo(lam, gamma, C, eta, alpha, float(len(C))))
        lam, _ = svi.lda_svi_step(lam, gamma, C, eta, float(len(C)), 1.0)
    bounds = np.array(bounds)
    assert np.all(np.diff(bounds) >= -1e-6 * np.abs(bounds[:-1])), bounds
    assert bounds[-1] > bounds[0]
