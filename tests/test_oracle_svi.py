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
