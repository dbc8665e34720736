"""Conjugacy detection with ``match`` and the mean-field / VMP updates synthesised from
it (intent: bayesic/algebra.py:1-6, README.md:30-37).  The
reference has no inference code, so the pins are closed-form posteriors and the
textbook coordinate-ascent updates.  CPU: float64 numpy backend of the oracle."""
import numpy as np
import numpy.testing as npt
import pytest

from bayesic_amd import algebra as A
from bayesic_amd.distribution import Dirichlet, Normal
from bayesic_amd.inference import (GammaNode, MeanFieldVMP, NormalNode, NotConjugate,
                                   conjugate_coefficients, expand_terms)
from oracle.einsum_eval import NumpyBackend

B64 = NumpyBackend(np.float64)
rs = np.random.RandomState(21)


def run(expr, **inputs):
    return expr.compile(B64)(**inputs)


def f64(name, ndim):
    return A.var(name, ndim, "float64")


def test_expand_terms_distributes_products_and_sums_over_add():
    X, Y, Z = f64("X", 2), f64("Y", 2), f64("Z", 2)
    vals = dict(X=rs.standard_normal((3, 4)), Y=rs.standard_normal((3, 4)), Z=rs.standard_normal((3, 4)))
    e = A.sum((X + Y) * (Z - 2 * X)) - 3 * A.sum(Y + 1)
    terms = expand_terms(e)
    assert len(terms) == 6
    assert all(not isinstance(f, A.add) for t in terms if isinstance(t, A.Einsum)
               for f, _ in t.factors_and_indices)
    total = sum(float(run(t, **{k: vals[k] for k in t.input_types})) for t in terms)
    npt.assert_allclose(total, run(e, **vals), rtol=1e-12)
    assert expand_terms(X) == [X]
    # a broadcast summand under a sum: sum_n (x_n + 1) mu  =  mu sum_n x_n  +  N mu
    x, mu = f64("x", 1), f64("mu", 0)
    xs = rs.standard_normal(7)
    pieces = expand_terms(A.sum((x + 1) * mu))
    assert len(pieces) == 2
    npt.assert_allclose(sum(float(run(t, x=xs, mu=0.4)) for t in pieces), 0.4 * (xs.sum() + 7),
                        rtol=1e-12)
    (c, _), _ = conjugate_coefficients(A.sum((x + 1) * mu), mu, (mu, mu ** 2))
    npt.assert_allclose(run(c, x=xs), xs.sum() + 7, rtol=1e-12)


def normal_gamma_log_joint(x, mu, tau, m0, v0, a0, b0):
    """x_n ~ N(mu, 1/tau), mu ~ N(m0, v0), tau ~ Gamma(a0, rate b0); constants dropped."""
    N = A.shape(x, 0)
    lik = A.sum(x * mu * tau) + A.sum(x * x * (-0.5 * tau)) \
        - N * (0.5 * (mu ** 2) * tau - 0.5 * A.log(tau))
    prior_mu = mu * (m0 / v0) + (mu ** 2) * (-0.5 / v0)
    prior_tau = (a0 - 1.0) * A.log(tau) - b0 * tau
    return lik + prior_mu + prior_tau


def test_coefficients_of_the_normal_gamma_model():
    x, mu, tau = f64("x", 1), f64("mu", 0), f64("tau", 0)
    lj = normal_gamma_log_joint(x, mu, tau, 0.5, 4.0, 2.0, 3.0)
    xs = rs.standard_normal(50) * 0.7 + 1.0
    (c1, c2), rest = conjugate_coefficients(lj, mu, (mu, mu ** 2))
    npt.assert_allclose(run(c1, x=xs, tau=1.7), 1.7 * xs.sum() + 0.5 / 4.0, rtol=1e-12)
    npt.assert_allclose(run(c2, x=xs, tau=1.7), -0.5 * 50 * 1.7 - 0.5 / 4.0, rtol=1e-12)
    assert all("mu" not in t.input_types for t in rest)
    (d1, d2), _ = conjugate_coefficients(lj, tau, (A.log(tau), tau))
    npt.assert_allclose(run(d1, x=xs), 0.5 * 50 + 2.0 - 1.0, rtol=1e-12)
    # E[...] is taken by binding mu and mu ** 2 separately; here a point value binds both
    npt.assert_allclose(run(d2, x=xs, mu=0.3), 0.3 * xs.sum() - 0.5 * (xs ** 2).sum()
                        - 0.5 * 50 * 0.09 - 3.0, rtol=1e-12)


def test_known_variance_normal_mean_lands_on_the_exact_posterior():
    """Built from the Distribution nodes: one unit-step update IS the posterior
    (README.md:36)."""
    x, mu = f64("x", 1), f64("mu", 0)
    v, m0, v0 = 2.5, -1.0, 9.0
    xs = rs.standard_normal(200) * np.sqrt(v) + 3.0
    lj = Normal().iid(1).log_likelihood(x, mean=mu, variance=v) \
        + Normal().log_likelihood(mu, mean=m0, variance=v0)
    node = NormalNode(mu)
    vmp = MeanFieldVMP(lj, [node], {"x": xs}, backend=B64)
    vmp.sweep()
    post_prec = 1.0 / v0 + len(xs) / v
    npt.assert_allclose(node.variance, 1.0 / post_prec, rtol=1e-12)
    npt.assert_allclose(node.mean, (m0 / v0 + xs.sum() / v) / post_prec, rtol=1e-12)
    # a damped step is the convex combination in natural parameters (SVI, README.md:69-79)
    node2 = NormalNode(mu, mean=0.0, variance=1.0)
    vmp2 = MeanFieldVMP(lj, [node2], {"x": xs}, backend=B64)
    vmp2.update("mu", rho=0.25)
    npt.assert_allclose(node2.eta[1], 0.75 * (-0.5) + 0.25 * (-0.5 * post_prec), rtol=1e-12)


def reference_mean_field(xs, m0, v0, a0, b0, sweeps):
    """Textbook coordinate ascent for q(mu) q(tau) (e.g. Bishop 10.1.3, independent priors)."""
    N, sx, sxx = len(xs), xs.sum(), (xs ** 2).sum()
    e_tau = 1.0
    for _ in range(sweeps):
        prec = 1.0 / v0 + N * e_tau
        m = (m0 / v0 + e_tau * sx) / prec
        e_mu, e_mu2 = m, m * m + 1.0 / prec
        a = a0 + 0.5 * N
        b = b0 + 0.5 * (sxx - 2.0 * e_mu * sx + N * e_mu2)
        e_tau = a / b
    return m, 1.0 / prec, a, b


def test_mean_field_normal_gamma_matches_textbook_updates():
    x, mu, tau = f64("x", 1), f64("mu", 0), f64("tau", 0)
    xs = 2.0 + 1.5 * rs.standard_normal(500)
    m0, v0, a0, b0 = 0.0, 100.0, 1.0, 1.0
    lj = normal_gamma_log_joint(x, mu, tau, m0, v0, a0, b0)
    q_mu, q_tau = NormalNode(mu), GammaNode(tau, shape=1.0, rate=1.0)
    vmp = MeanFieldVMP(lj, [q_mu, q_tau], {"x": xs}, backend=B64)
    for _ in range(15):
        vmp.sweep()
    m, v, a, b = reference_mean_field(xs, m0, v0, a0, b0, 15)
    npt.assert_allclose([q_mu.mean, q_mu.variance, q_tau.shape, q_tau.rate], [m, v, a, b], rtol=1e-10)
    # and it found the data: posterior mean ~ sample mean, E[tau] ~ 1 / sample variance
    npt.assert_allclose(q_mu.mean, xs.mean(), rtol=1e-2)
    npt.assert_allclose(q_tau.shape / q_tau.rate, 1.0 / xs.var(), rtol=2e-2)


def test_dirichlet_categorical_counts():
    X, theta = f64("X", 2), f64("theta", 1)
    alpha0 = np.array([0.5, 1.0, 2.0, 4.0])
    onehot = np.eye(4)[rs.randint(0, 4, 60)]
    lj = A.sum(X * A.dimshuffle(A.log(theta), "x", 0)) \
        + Dirichlet().log_likelihood(theta, concentration=A.constant(alpha0))
    (c,), rest = conjugate_coefficients(lj, theta, (A.log(theta),))
    npt.assert_allclose(run(c, X=onehot), onehot.sum(0) + alpha0 - 1.0, rtol=1e-12)


def test_non_conjugate_terms_are_reported():
    x, w = f64("x", 1), f64("w", 0)
    lj = A.sum(A.log(1 + A.exp(x * w))) + w * 0.3
    with pytest.raises(NotConjugate) as info:
        conjugate_coefficients(lj, w, (w, w ** 2))
    assert "w" in info.value.term.input_types
    # bilinear use of a latent (mu * mu instead of the declared statistic mu ** 2)
    mu = f64("mu", 0)
    with pytest.raises(NotConjugate):
        conjugate_coefficients(A.sum(x * mu * mu), mu, (mu, mu ** 2))
    with pytest.raises(TypeError):          # an input that is neither data nor latent
        MeanFieldVMP(A.sum(x * mu) + (mu ** 2) * w, [NormalNode(mu)], {"x": np.ones(3)}, backend=B64)


@pytest.mark.gpu
def test_mean_field_normal_gamma_on_device(ctx):
    """BASELINE config 1's data (N = 10 000) through the MI355X backend: the data-sized
    messages are fused map-reduce launches over the resident x; float32 data."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from oracle import svi
    xs = np.asarray(svi.make_cfg1(), np.float32)
    x, mu, tau = A.var("x", 1), A.var("mu", 0), A.var("tau", 0)
    lj = normal_gamma_log_joint(x, mu, tau, 0.0, 100.0, 1.0, 1.0)
    q_mu, q_tau = NormalNode(mu), GammaNode(tau)
    vmp = MeanFieldVMP(lj, [q_mu, q_tau], {"x": xs}, backend=DeviceBackend(ctx))
    for _ in range(10):
        vmp.sweep()
    m, v, a, b = reference_mean_field(xs.astype(np.float64), 0.0, 100.0, 1.0, 1.0, 10)
    npt.assert_allclose([q_mu.mean, q_mu.variance, q_tau.shape, q_tau.rate], [m, v, a, b], rtol=2e-5)


@pytest.mark.gpu
def test_generic_score_function_vi_matches_the_fused_config5_path(ctx):
    """The general BBVI engine (model = algebra expression, evaluated by the executor) and the
    hand-fused config-5 kernels see the same Philox draws for the same seed, so one update
    from the same lam must agree: f_s to float32 evaluation error, and the engine's gradient
    must be the oracle's estimator applied to its own f."""
    import math
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference import ScoreFunctionVI
    from bayesic_amd.svi.bbvi import LogRegBBVI
    from oracle import svi
    N, D, G, S = 3000, 8, 5, 64
    X_, y_, g_, _, _ = svi.make_cfg5(N, D, G)
    n_total, a0, b0 = 10.0 * N, 1.0, 1.0
    scale = n_total / N
    onehot = np.eye(G, dtype=np.float32)[g_]

    Xv, yv, Gm = A.var("X", 2), A.var("y", 1), A.var("Gm", 2)
    W, Bg, Z = A.var("W", 2), A.var("Bg", 2), A.var("Z", 2)          # [S,D], [S,G], [S,1]
    L = A.dot(Xv, W.T) + A.dot(Gm, Bg.T)                              # logits [N, S]
    loglik = A.sum(A.dimshuffle(yv, 0, "x") * L - A.log(1 + A.exp(L)), axis=0)
    zeta = A.sum(Z, axis=1)                                            # [S]
    lp_w = A.sum(-0.5 * (W * W), axis=1) - 0.5 * D * math.log(2 * math.pi)
    lp_b = (-0.5 * G * math.log(2 * math.pi)) + (0.5 * G) * zeta \
        - 0.5 * (A.exp(zeta) * A.sum(Bg * Bg, axis=1))
    lp_z = (a0 * math.log(b0) - math.lgamma(a0)) + a0 * zeta - b0 * A.exp(zeta)
    log_joint = scale * loglik + lp_w + lp_b + lp_z

    eng = ScoreFunctionVI(log_joint, [(W, D), (Bg, G), (Z, 1)],
                          {"X": X_, "y": y_, "Gm": onehot}, n_samples=S, seed=5, lr=0.05,
                          backend=DeviceBackend(ctx))
    fused = LogRegBBVI(X_, y_, g_, G, n_total=n_total, n_samples=S, seed=5, lr=0.05, a0=a0, b0=b0,
                       ctx=ctx)
    lam0 = eng.lam.copy()
    npt.assert_array_equal(lam0, fused.lam.cpu().numpy())
    eng.step()
    fused.step()
    ctx.sync()
    f_fused = fused.f.cpu().numpy()
    npt.assert_allclose(eng.f, f_fused, rtol=2e-5, atol=2e-2)          # |f| ~ 1e4, float32 sums
    P = D + G + 1
    eps, _ = svi.bbvi_sample(lam0, P, S, 5, step=0)
    elbo_ref, grad_ref, _, _ = svi.bbvi_elbo_and_grad(
        lam0, eps, (eng.f - (svi.bbvi_log_prior(lam0[:P][None, :] + np.exp(lam0[P:])[None, :] * eps,
                                                D, G, a0, b0)
                             - (-0.5 * math.log(2 * math.pi) - lam0[P:][None, :]
                                - 0.5 * eps * eps).sum(1))) / scale, D, G, scale, a0, b0)
    npt.assert_allclose(eng.elbo, elbo_ref, rtol=1e-9)
    npt.assert_allclose(eng.grad, grad_ref, rtol=1e-6, atol=1e-8 * np.abs(grad_ref).max())
    # both paths move lam the same way (same draws, same estimator)
    d_eng, d_fused = eng.lam - lam0, fused.lam.cpu().numpy() - lam0
    assert np.mean(np.sign(d_eng) == np.sign(d_fused)) > 0.95


# ---- Dirichlet / Categorical nodes: the discrete side of a mixture ---------------------------

def mixture_log_joint(x, Z, theta, mu, alpha0, v, m0, v0):
    """x_n ~ N(mu_{z_n}, v) with one-hot z_n ~ Categorical(theta), theta ~ Dirichlet(alpha0),
    mu_k ~ N(m0, v0); constants dropped.  Z is [N, K] one-hot, theta and mu are [K]."""
    lt = A.dimshuffle(A.log(theta), "x", 0)
    mu_row, mu2_row = A.dimshuffle(mu, "x", 0), A.dimshuffle(mu ** 2, "x", 0)
    x_col = A.dimshuffle(x, 0, "x")
    lik = A.sum(Z * x_col * mu_row) * (1.0 / v) + A.sum(Z * mu2_row) * (-0.5 / v) \
        + A.sum(Z * A.dimshuffle(x * x, 0, "x")) * (-0.5 / v)
    return lik + A.sum(Z * lt) + A.sum(A.log(theta)) * (alpha0 - 1.0) \
        + A.sum(mu) * (m0 / v0) + A.sum(mu ** 2) * (-0.5 / v0)


def test_dirichlet_weights_land_on_the_exact_posterior_when_assignments_are_observed():
    from bayesic_amd.inference import DirichletNode
    K, N = 4, 300
    x, Z, theta, mu = f64("x", 1), f64("Z", 2), f64("theta", 1), f64("mu", 1)
    z = rs.randint(K, size=N)
    Zs = np.eye(K)[z]
    xs = rs.standard_normal(N)
    lj = mixture_log_joint(x, Z, theta, mu, alpha0=1.5, v=1.0, m0=0.0, v0=10.0)
    (c,), _ = conjugate_coefficients(lj, theta, (A.log(theta),))
    npt.assert_allclose(run(c, Z=Zs), Zs.sum(0) + 0.5, rtol=1e-12)      # counts + alpha0 - 1
    node = DirichletNode(theta, alpha=np.ones(K))
    mus = NormalNode(mu, mean=np.zeros(K), variance=np.ones(K))
    vmp = MeanFieldVMP(lj, [node, mus], dict(x=xs, Z=Zs), backend=B64)
    vmp.update("theta")
    npt.assert_allclose(node.alpha, 1.5 + np.bincount(z, minlength=K), rtol=1e-12)
    from scipy.special import digamma as psi
    npt.assert_allclose(node.expectations()[0], psi(node.alpha) - psi(node.alpha.sum()), rtol=1e-12)


def test_mixture_coordinate_ascent_matches_the_textbook_updates():
    """Latent assignments: q(z_n) = softmax_k(E[log theta_k] + (x_n E[mu_k] - E[mu_k^2]/2)/v),
    then the component means and the weights from the responsibilities (Bishop 10.2 with known
    variance)."""
    from bayesic_amd.inference import CategoricalNode, DirichletNode
    from scipy.special import digamma as psi
    K, N, v, a0, m0, v0 = 3, 400, 0.25, 2.0, 0.0, 25.0
    x, Z, theta, mu = f64("x", 1), f64("Z", 2), f64("theta", 1), f64("mu", 1)
    centres = np.array([-3.0, 0.5, 4.0])
    z = rs.randint(K, size=N)
    xs = centres[z] + np.sqrt(v) * rs.standard_normal(N)
    lj = mixture_log_joint(x, Z, theta, mu, alpha0=a0, v=v, m0=m0, v0=v0)
    zn = CategoricalNode(Z, log_prob=np.zeros((N, K)))
    tn = DirichletNode(theta, alpha=np.full(K, a0))
    mn = NormalNode(mu, mean=np.array([-1.0, 0.0, 1.0]), variance=np.ones(K))
    vmp = MeanFieldVMP(lj, [zn, mn, tn], dict(x=xs), backend=B64)

    # one sweep by hand
    Elt = psi(tn.alpha) - psi(tn.alpha.sum())
    Em, Em2 = mn.mean, mn.mean ** 2 + mn.variance
    logit = Elt[None, :] + (xs[:, None] * Em[None, :] - 0.5 * Em2[None, :] - 0.5 * xs[:, None] ** 2) / v
    R = np.exp(logit - logit.max(1, keepdims=True))
    R /= R.sum(1, keepdims=True)
    prec = R.sum(0) / v + 1.0 / v0
    mean = (R.T @ xs / v + m0 / v0) / prec
    alpha = a0 + R.sum(0)

    vmp.sweep()
    npt.assert_allclose(zn.expectations()[0], R, rtol=1e-10, atol=1e-14)
    npt.assert_allclose(mn.mean, mean, rtol=1e-10)
    npt.assert_allclose(mn.variance, 1.0 / prec, rtol=1e-10)
    npt.assert_allclose(tn.alpha, alpha, rtol=1e-10)
    for _ in range(30):
        vmp.sweep()
    order = np.argsort(mn.mean)
    npt.assert_allclose(mn.mean[order], centres, atol=0.15)              # finds the three clusters
    npt.assert_allclose(tn.alpha[order] - a0, np.bincount(z, minlength=K), rtol=0.1)


@pytest.mark.gpu
def test_mixture_vmp_on_device_matches_the_float64_backend(ctx):
    """The same derived updates with the data-sized messages (responsibility-weighted sums over
    20 000 points) evaluated by the HIP executor; float32 data and expectations."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference import CategoricalNode, DirichletNode
    K, N, v, a0 = 3, 20000, 0.25, 2.0
    centres = np.array([-3.0, 0.5, 4.0])
    r = np.random.RandomState(5)
    z = r.randint(K, size=N)
    xs = (centres[z] + np.sqrt(v) * r.standard_normal(N)).astype(np.float32)

    def build(backend, dtype):
        x, Z = A.var("x", 1, dtype), A.var("Z", 2, dtype)
        theta, mu = A.var("theta", 1, dtype), A.var("mu", 1, dtype)
        lj = mixture_log_joint(x, Z, theta, mu, alpha0=a0, v=v, m0=0.0, v0=25.0)
        nodes = [CategoricalNode(Z, log_prob=np.zeros((N, K))),
                 NormalNode(mu, mean=np.array([-1.0, 0.0, 1.0]), variance=np.ones(K)),
                 DirichletNode(theta, alpha=np.full(K, a0))]
        return MeanFieldVMP(lj, nodes, dict(x=xs), backend=backend), nodes

    dev, dn = build(DeviceBackend(ctx), "float32")
    ref, rn = build(B64, "float64")
    for _ in range(5):
        dev.sweep()
        ref.sweep()
    npt.assert_allclose(dn[1].mean, rn[1].mean, rtol=2e-4, atol=2e-4)
    npt.assert_allclose(dn[1].variance, rn[1].variance, rtol=2e-4)
    npt.assert_allclose(dn[2].alpha, rn[2].alpha, rtol=2e-4)
    npt.assert_allclose(dn[0].expectations()[0], rn[0].expectations()[0], atol=5e-4)


# ---- vector latent with a full covariance: Bayesian linear regression (config 2's model) ------

def blr_log_joint(X, y, w, W2, tau, P0, a0, b0):
    """y_n ~ N(x_n.w, 1/tau), w ~ N(0, P0^-1), tau ~ Gamma(a0, rate b0) (tau may be a number);
    W2 stands for w w^T (MVNormalNode).  Constants dropped."""
    N = A.shape(y, 0)
    quad = A.sum(A.dot(X.T, X) * W2) - 2.0 * A.sum(A.dot(X.T, y) * w) + A.sum(y * y)
    lj = quad * (-0.5 * tau) + A.sum(W2 * P0) * (-0.5)
    if isinstance(tau, A.Expression):
        lj = lj + N * (0.5 * A.log(tau)) + (a0 - 1.0) * A.log(tau) - b0 * tau
    return lj


def test_linear_regression_with_known_noise_is_exact_in_one_step():
    from bayesic_amd.inference import MVNormalNode
    N, D, tau = 500, 6, 4.0
    Xs = rs.standard_normal((N, D))
    ys = Xs @ rs.standard_normal(D) + rs.standard_normal(N) / np.sqrt(tau)
    P0s = np.diag(rs.uniform(0.5, 2.0, D))
    X, y, w, W2, P0 = f64("X", 2), f64("y", 1), f64("w", 1), f64("W2", 2), f64("P0", 2)
    lj = blr_log_joint(X, y, w, W2, tau, P0, None, None)
    node = MVNormalNode(w, W2, mean=np.zeros(D), covariance=np.eye(D))
    vmp = MeanFieldVMP(lj, [node], dict(X=Xs, y=ys, P0=P0s), backend=B64)
    vmp.update("w")
    lam = tau * Xs.T @ Xs + P0s
    npt.assert_allclose(node.precision, lam, rtol=1e-11)      # (two float64 summation orders of 500 terms; the data depend on the tests drawn before)
    npt.assert_allclose(node.mean, np.linalg.solve(lam, tau * Xs.T @ ys), rtol=1e-10)


def blr_mean_field_by_hand(Xs, ys, P0s, a0, b0, sweeps):
    N, D = Xs.shape
    XtX, Xty, yty = Xs.T @ Xs, Xs.T @ ys, ys @ ys
    Etau = 1.0
    for _ in range(sweeps):
        lam = Etau * XtX + P0s
        m = np.linalg.solve(lam, Etau * Xty)
        S2 = np.linalg.inv(lam) + np.outer(m, m)
        a = a0 + 0.5 * N
        b = b0 + 0.5 * ((XtX * S2).sum() - 2.0 * Xty @ m + yty)
        Etau = a / b
    return m, lam, a, b


def test_linear_regression_with_unknown_noise_matches_hand_written_coordinate_ascent():
    from bayesic_amd.inference import MVNormalNode
    N, D, a0, b0 = 400, 5, 2.0, 1.5
    Xs = rs.standard_normal((N, D))
    ys = Xs @ rs.standard_normal(D) + 0.3 * rs.standard_normal(N)
    P0s = np.eye(D) * 0.7
    X, y, w, W2, P0, tau = f64("X", 2), f64("y", 1), f64("w", 1), f64("W2", 2), f64("P0", 2), f64("tau", 0)
    lj = blr_log_joint(X, y, w, W2, tau, P0, a0, b0)
    qw = MVNormalNode(w, W2, mean=np.zeros(D), covariance=np.eye(D))
    qt = GammaNode(tau, shape=1.0, rate=1.0)
    vmp = MeanFieldVMP(lj, [qw, qt], dict(X=Xs, y=ys, P0=P0s), backend=B64)
    for _ in range(8):
        vmp.sweep()
    m, lam, a, b = blr_mean_field_by_hand(Xs, ys, P0s, a0, b0, 8)
    npt.assert_allclose(qt.shape, a, rtol=1e-12)
    npt.assert_allclose(qt.rate, b, rtol=1e-9)
    # the last w update used the tau of the sweep before: redo it with the final E[tau]
    vmp.update("w")
    lam_f = (a / b) * Xs.T @ Xs + P0s
    npt.assert_allclose(qw.precision, lam_f, rtol=1e-9)
    npt.assert_allclose(qw.mean, np.linalg.solve(lam_f, (a / b) * Xs.T @ ys), rtol=1e-8)


@pytest.mark.gpu
def test_linear_regression_vmp_on_device(ctx):
    """Config 2's shapes in small: the D x D Gram message is one MFMA GEMM over the resident X."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference import MVNormalNode
    N, D, a0, b0 = 50_000, 256, 1.0, 1.0
    r = np.random.RandomState(11)
    Xs = r.standard_normal((N, D)).astype(np.float32)
    ys = (Xs @ (r.standard_normal(D) / 16) + 0.5 * r.standard_normal(N)).astype(np.float32)
    P0s = np.eye(D, dtype=np.float32)
    X, y, w, W2, P0, tau = A.var("X", 2), A.var("y", 1), A.var("w", 1), A.var("W2", 2), A.var("P0", 2), A.var("tau", 0)
    lj = blr_log_joint(X, y, w, W2, tau, P0, a0, b0)
    qw = MVNormalNode(w, W2, mean=np.zeros(D), covariance=np.eye(D))
    qt = GammaNode(tau, shape=1.0, rate=1.0)
    vmp = MeanFieldVMP(lj, [qw, qt], dict(X=Xs, y=ys, P0=P0s), backend=DeviceBackend(ctx))
    for _ in range(4):
        vmp.sweep()
    m, lam, a, b = blr_mean_field_by_hand(Xs.astype(np.float64), ys.astype(np.float64),
                                         P0s.astype(np.float64), a0, b0, 4)
    npt.assert_allclose(qt.shape, a, rtol=1e-6)
    npt.assert_allclose(qt.rate, b, rtol=1e-4)
    npt.assert_allclose(qw.mean, m, rtol=2e-3, atol=2e-5)


def test_config2_model_normal_inverse_gamma_by_derived_mean_field():
    """BASELINE config 2's model as stated: w | s2 ~ N(0, s2 I), s2 ~ InvGamma(1, 1),
    y_n ~ N(x_n.w, s2).  The derived coordinate ascent against the hand-written one, and its
    E[w] against the exact Normal-Inverse-Gamma posterior mean (which mean field shares)."""
    from bayesic_amd.inference import InverseGammaNode, MVNormalNode
    from oracle import svi
    N, D, a0, b0 = 600, 8, 1.0, 1.0
    Xs = rs.standard_normal((N, D))
    ys = Xs @ (rs.standard_normal(D) / 4) + 0.5 * rs.standard_normal(N)
    X, y, w, W2, Id, s2 = f64("X", 2), f64("y", 1), f64("w", 1), f64("W2", 2), f64("Id", 2), f64("s2", 0)
    quad = A.sum(A.dot(X.T, X) * W2) - 2.0 * A.sum(A.dot(X.T, y) * w) + A.sum(y * y)
    prec = s2 ** -1
    lj = quad * (-0.5 * prec) + A.sum(W2 * Id) * (-0.5 * prec) \
        - (0.5 * (N + D) + a0 + 1.0) * A.log(s2) - b0 * prec
    qw = MVNormalNode(w, W2, mean=np.zeros(D), covariance=np.eye(D))
    qs = InverseGammaNode(s2, shape=1.0, scale=1.0)
    vmp = MeanFieldVMP(lj, [qw, qs], dict(X=Xs, y=ys, Id=np.eye(D)), backend=B64)
    for _ in range(10):
        vmp.sweep()
    # by hand
    XtX, Xty, yty = Xs.T @ Xs, Xs.T @ ys, ys @ ys
    Eprec = 1.0
    for _ in range(10):
        lam = Eprec * (XtX + np.eye(D))
        m = np.linalg.solve(lam, Eprec * Xty)
        S2 = np.linalg.inv(lam) + np.outer(m, m)
        a = a0 + 0.5 * (N + D)
        b = b0 + 0.5 * ((XtX * S2).sum() - 2.0 * Xty @ m + yty + np.trace(S2))
        Eprec = a / b
    npt.assert_allclose(qs.shape, a, rtol=1e-12)
    npt.assert_allclose(qs.scale, b, rtol=1e-9)
    mu_exact, _, _, _ = svi.blr_exact_posterior(Xs, ys)      # (the oracle rounds X, y to float32 first)
    npt.assert_allclose(qw.mean, mu_exact, rtol=1e-6, atol=1e-8)


def test_multivariate_gaussian_with_normal_wishart_mean_field():
    """x_n ~ N(mu, Lambda^-1) with mu ~ N(0, (k0 I)^-1) and Lambda ~ Wishart(nu0, V0): the derived
    coordinate ascent (Bishop 10.1.3 without the coupling of the priors) against the hand-written
    one.  t(x) = (x, x x^T) of bayesic/distribution/core.py:41-44 in full-covariance form."""
    from bayesic_amd.distribution import logdet
    from bayesic_amd.inference import MVNormalNode, WishartNode
    N, D, k0, nu0 = 300, 3, 0.1, 5.0
    Ltrue = np.array([[2.0, 0.5, 0.0], [0.5, 1.0, 0.2], [0.0, 0.2, 1.5]])
    Xs = rs.multivariate_normal([1.0, -2.0, 0.5], np.linalg.inv(Ltrue), size=N)
    V0 = np.eye(D) / nu0
    X, mu, M2, Lam, Id = f64("X", 2), f64("mu", 1), f64("M2", 2), f64("Lam", 2), f64("Id", 2)
    Nn = A.shape(X, 0)
    sx = A.sum(X, axis=0)
    lj = Nn * (0.5 * logdet(Lam)) + A.sum(Lam * A.dot(X.T, X)) * (-0.5) \
        + A.sum(Lam * A.outer(sx, mu)) + Nn * (A.sum(Lam * M2) * (-0.5)) \
        + A.sum(M2 * Id) * (-0.5 * k0) \
        + (0.5 * (nu0 - D - 1.0)) * logdet(Lam) + A.sum(Lam * Id) * (-0.5 * nu0)    # V0^-1 = nu0 I
    qm = MVNormalNode(mu, M2, mean=np.zeros(D), covariance=np.eye(D))
    ql = WishartNode(Lam, dof=nu0, scale=V0)
    vmp = MeanFieldVMP(lj, [qm, ql], dict(X=Xs, Id=np.eye(D)), backend=B64)
    for _ in range(6):
        vmp.sweep()
    # by hand
    S, sxs = Xs.T @ Xs, Xs.sum(0)
    EL = nu0 * V0
    for _ in range(6):
        prec = N * EL + k0 * np.eye(D)
        m = np.linalg.solve(prec, EL @ sxs)
        M2s = np.linalg.inv(prec) + np.outer(m, m)
        nu = nu0 + N
        Vinv = nu0 * np.eye(D) + S - np.outer(sxs, m) - np.outer(m, sxs) + N * M2s
        EL = nu * np.linalg.inv(Vinv)
    npt.assert_allclose(ql.dof, nu, rtol=1e-12)
    npt.assert_allclose(np.linalg.inv(ql.scale), Vinv, rtol=1e-8)
    vmp.update("mu")
    prec = N * EL + k0 * np.eye(D)
    npt.assert_allclose(qm.mean, np.linalg.solve(prec, EL @ sxs), rtol=1e-8)
    npt.assert_allclose(ql.expectations()[1], Ltrue, rtol=0.35, atol=0.2)     # recovers the precision


# ---- full-covariance Gaussian mixture: every node type at once ----------------------------------

def full_mixture_log_joint(X, Z, theta, mu, M2, Lam, Id, alpha0, k0, nu0, D):
    """x_n ~ N(mu_{z_n}, Lambda_{z_n}^-1); theta ~ Dirichlet(alpha0); mu_k ~ N(0, (k0 I)^-1);
    Lambda_k ~ Wishart(nu0, I / nu0).  Z [N,K] one-hot, mu [K,D], M2 [K,D,D] stands for
    mu_k mu_k^T, Lam [K,D,D], Id the D x D identity.  Constants dropped."""
    from bayesic_amd.distribution import logdet
    ld = logdet(Lam)                                                     # [K]
    Zb = A.dimshuffle(Z, 0, 1, "x", "x")                                 # [N,K,1,1]
    Xd, Xe = A.dimshuffle(X, 0, "x", 1, "x"), A.dimshuffle(X, 0, "x", "x", 1)
    Lb = A.dimshuffle(Lam, "x", 0, 1, 2)
    Idb = A.dimshuffle(Id, "x", 0, 1)
    lik = A.sum(Z * A.dimshuffle(A.log(theta), "x", 0)) \
        + A.sum(Z * A.dimshuffle(ld, "x", 0)) * 0.5 \
        + A.sum(Zb * Xd * Xe * Lb) * (-0.5) \
        + A.sum(Zb * Xd * Lb * A.dimshuffle(mu, "x", 0, "x", 1)) \
        + A.sum(Zb * Lb * A.dimshuffle(M2, "x", 0, 1, 2)) * (-0.5)
    prior = A.sum(A.log(theta)) * (alpha0 - 1.0) + A.sum(M2 * Idb) * (-0.5 * k0) \
        + A.sum(ld) * (0.5 * (nu0 - D - 1.0)) + A.sum(Lam * Idb) * (-0.5 * nu0)
    return lik + prior


def full_mixture_by_hand(Xs, R0, m0, K, alpha0, k0, nu0, sweeps):
    from scipy.special import digamma as psi
    N, D = Xs.shape
    alpha, m, C = np.full(K, alpha0), m0.copy(), np.stack([np.eye(D)] * K)
    nu, V = np.full(K, nu0), np.stack([np.eye(D) / nu0] * K)
    R = R0
    for _ in range(sweeps):
        EL = nu[:, None, None] * V
        Eld = psi(0.5 * (nu[:, None] - np.arange(D))).sum(1) + D * np.log(2.0) + np.linalg.slogdet(V)[1]
        Elt = psi(alpha) - psi(alpha.sum())
        M2 = C + m[:, :, None] * m[:, None, :]
        quad = np.einsum("nd,kde,ne->nk", Xs, EL, Xs) - 2.0 * np.einsum("nd,kde,ke->nk", Xs, EL, m) \
            + np.einsum("kde,kde->k", EL, M2)[None, :]
        logit = Elt[None, :] + 0.5 * Eld[None, :] - 0.5 * quad
        R = np.exp(logit - logit.max(1, keepdims=True))
        R /= R.sum(1, keepdims=True)
        Nk, Sx = R.sum(0), R.T @ Xs
        prec = Nk[:, None, None] * EL + k0 * np.eye(D)
        m = np.linalg.solve(prec, np.einsum("kde,ke->kd", EL, Sx)[..., None])[..., 0]
        C = np.linalg.inv(prec)
        M2 = C + m[:, :, None] * m[:, None, :]
        Sxx = np.einsum("nk,nd,ne->kde", R, Xs, Xs)
        Vinv = nu0 * np.eye(D) + Sxx - Sx[:, :, None] * m[:, None, :] - m[:, :, None] * Sx[:, None, :] \
            + Nk[:, None, None] * M2
        nu, V = nu0 + Nk, np.linalg.inv(Vinv)
        alpha = alpha0 + Nk
    return R, m, C, nu, V, alpha


def _full_mixture(backend, dtype, Xs, K, m_init, alpha0=2.0, k0=0.05, nu0=6.0):
    from bayesic_amd.inference import CategoricalNode, DirichletNode, MVNormalNode, WishartNode
    N, D = Xs.shape
    X, Z, theta = A.var("X", 2, dtype), A.var("Z", 2, dtype), A.var("theta", 1, dtype)
    mu, M2, Lam, Id = A.var("mu", 2, dtype), A.var("M2", 3, dtype), A.var("Lam", 3, dtype), A.var("Id", 2, dtype)
    lj = full_mixture_log_joint(X, Z, theta, mu, M2, Lam, Id, alpha0, k0, nu0, D)
    nodes = [CategoricalNode(Z, log_prob=np.zeros((N, K))),
             MVNormalNode(mu, M2, mean=m_init, covariance=np.stack([np.eye(D)] * K)),
             WishartNode(Lam, dof=nu0, scale=np.stack([np.eye(D) / nu0] * K)),
             DirichletNode(theta, alpha=np.full(K, alpha0))]
    return MeanFieldVMP(lj, nodes, dict(X=Xs, Id=np.eye(D)), backend=backend), nodes


def _mixture_data(N, K, D, seed):
    r = np.random.RandomState(seed)
    centres = r.standard_normal((K, D)) * 4.0
    Ls = [np.linalg.cholesky(np.eye(D) * 0.5 + 0.3 * np.cov(r.standard_normal((D, 3 * D)))) for _ in range(K)]
    z = r.randint(K, size=N)
    Xs = np.stack([centres[k] + Ls[k] @ r.standard_normal(D) for k in z])
    return Xs, centres + 0.5 * r.standard_normal((K, D))


def test_full_covariance_mixture_matches_hand_written_updates():
    K, D, N = 3, 2, 500
    Xs, m_init = _mixture_data(N, K, D, 31)
    vmp, (qz, qm, ql, qt) = _full_mixture(B64, "float64", Xs, K, m_init)
    for sweeps in (1, 4):
        vmp2, (qz, qm, ql, qt) = _full_mixture(B64, "float64", Xs, K, m_init)
        for _ in range(sweeps):
            vmp2.sweep()
        R, m, C, nu, V, alpha = full_mixture_by_hand(Xs, None, m_init, K, 2.0, 0.05, 6.0, sweeps)
        npt.assert_allclose(qz.expectations()[0], R, rtol=1e-8, atol=1e-12)
        npt.assert_allclose(qm.mean, m, rtol=1e-8)
        npt.assert_allclose(qm.covariance, C, rtol=1e-8)
        npt.assert_allclose(ql.dof, nu, rtol=1e-10)
        npt.assert_allclose(ql.scale, V, rtol=1e-7)
        npt.assert_allclose(qt.alpha, alpha, rtol=1e-10)


@pytest.mark.gpu
def test_full_covariance_mixture_on_device_uses_the_one_pass_second_moment_kernel(ctx):
    """The same derived updates on the MI355X backend.  The Wishart message contains
    sum_n E[z_nk] x_n x_n^T: the executor must run it through bsc_weighted_outer (one pass over
    the data) and not materialise the K x D x N product."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from test_fusion_gpu import Counting
    K, D, N = 4, 4, 20000
    Xs, m_init = _mixture_data(N, K, D, 32)
    Xs = Xs.astype(np.float32)
    dev, (dz, dm, dl, dt) = _full_mixture(DeviceBackend(ctx), "float32", Xs, K, m_init)
    ref, (rz, rm, rl, rt) = _full_mixture(B64, "float64", Xs.astype(np.float64), K, m_init)
    with Counting(ctx) as c:
        dev.update("Lam")
    assert c.count("bsc_weighted_outer") >= 1, c.calls
    dev, (dz, dm, dl, dt) = _full_mixture(DeviceBackend(ctx), "float32", Xs, K, m_init)
    for _ in range(3):
        dev.sweep()
        ref.sweep()
    npt.assert_allclose(dm.mean, rm.mean, rtol=1e-3, atol=1e-3)
    npt.assert_allclose(dl.dof, rl.dof, rtol=1e-4)
    npt.assert_allclose(dl.expectations()[1], rl.expectations()[1], rtol=5e-3, atol=5e-3)
    npt.assert_allclose(dt.alpha, rt.alpha, rtol=1e-3)


# ---- the evidence lower bound: every coordinate update must raise it ---------------------------

def test_node_entropies_are_the_scipy_ones():
    import scipy.stats as st
    from bayesic_amd.inference import (CategoricalNode, DirichletNode, InverseGammaNode, MVNormalNode,
                                       WishartNode)
    v, t, M, Z = f64("v", 0), f64("t", 1), f64("M", 2), f64("Zc", 2)
    S = np.array([[2.0, 0.3], [0.3, 1.0]])
    npt.assert_allclose(NormalNode(v, 1.0, 2.5).entropy(), st.norm(1.0, np.sqrt(2.5)).entropy(), rtol=1e-12)
    npt.assert_allclose(GammaNode(v, 3.0, 2.0).entropy(), st.gamma(3.0, scale=0.5).entropy(), rtol=1e-12)
    npt.assert_allclose(InverseGammaNode(v, 3.0, 2.0).entropy(), st.invgamma(3.0, scale=2.0).entropy(), rtol=1e-12)
    npt.assert_allclose(DirichletNode(t, np.array([1.5, 2.0, 0.7])).entropy(),
                        st.dirichlet([1.5, 2.0, 0.7]).entropy(), rtol=1e-12)
    npt.assert_allclose(MVNormalNode(t, M, np.zeros(2), S).entropy(),
                        st.multivariate_normal(np.zeros(2), S).entropy(), rtol=1e-12)
    npt.assert_allclose(WishartNode(M, 5.0, S).entropy(), st.wishart(5.0, S).entropy(), rtol=1e-12)
    npt.assert_allclose(CategoricalNode(Z, np.log(np.array([[0.2, 0.3, 0.5]]))).entropy(),
                        st.entropy([0.2, 0.3, 0.5]), rtol=1e-12)


def _assert_bound_rises(vmp, sweeps):
    bound = [vmp.elbo()]
    for _ in range(sweeps):
        for node in vmp.nodes:
            vmp.update(node.var.name)
            bound.append(vmp.elbo())
    steps = np.diff(bound)
    assert (steps >= -1e-9 * np.abs(bound[:-1])).all(), (steps.min(), bound)
    assert bound[-1] > bound[0]
    return bound


def test_every_coordinate_update_raises_the_bound():
    """Mean field is coordinate ascent on the ELBO: with E[log p] taken by binding and the
    closed-form entropies, no update of any model here may lower it."""
    from bayesic_amd.inference import InverseGammaNode, MVNormalNode
    # Normal-Gamma (config 1's model)
    x, mu, tau = f64("x", 1), f64("mu", 0), f64("tau", 0)
    xs = rs.standard_normal(120) * 0.7 + 1.0
    vmp = MeanFieldVMP(normal_gamma_log_joint(x, mu, tau, 0.5, 4.0, 2.0, 3.0),
                       [NormalNode(mu), GammaNode(tau)], dict(x=xs), backend=B64)
    _assert_bound_rises(vmp, 5)
    # linear regression with Gamma noise precision
    N, D = 200, 4
    Xs = rs.standard_normal((N, D))
    ys = Xs @ rs.standard_normal(D) + 0.3 * rs.standard_normal(N)
    X, y, w, W2, P0, t2 = f64("X", 2), f64("y", 1), f64("w", 1), f64("W2", 2), f64("P0", 2), f64("tau", 0)
    vmp = MeanFieldVMP(blr_log_joint(X, y, w, W2, t2, P0, 2.0, 1.5),
                       [MVNormalNode(w, W2, np.zeros(D), np.eye(D)), GammaNode(t2)],
                       dict(X=Xs, y=ys, P0=np.eye(D)), backend=B64)
    _assert_bound_rises(vmp, 5)
    # scalar-mean mixture and the full-covariance mixture
    K = 3
    xs1 = np.array([-3.0, 0.5, 4.0])[rs.randint(K, size=300)] + 0.5 * rs.standard_normal(300)
    from bayesic_amd.inference import CategoricalNode, DirichletNode
    x1, Z, theta, m1 = f64("x", 1), f64("Z", 2), f64("theta", 1), f64("mu", 1)
    vmp = MeanFieldVMP(mixture_log_joint(x1, Z, theta, m1, alpha0=2.0, v=0.25, m0=0.0, v0=25.0),
                       [CategoricalNode(Z, np.zeros((300, K))), NormalNode(m1, np.array([-1.0, 0.0, 1.0]), np.ones(K)),
                        DirichletNode(theta, np.full(K, 2.0))], dict(x=xs1), backend=B64)
    _assert_bound_rises(vmp, 6)
    Xm, m_init = _mixture_data(400, 3, 2, 33)
    vmp, _ = _full_mixture(B64, "float64", Xm, 3, m_init)
    _assert_bound_rises(vmp, 5)


@pytest.mark.gpu
def test_bound_on_device_matches_the_float64_backend_and_rises(ctx):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference import MVNormalNode
    N, D = 30_000, 64
    r = np.random.RandomState(3)
    Xs = r.standard_normal((N, D)).astype(np.float32)
    ys = (Xs @ (r.standard_normal(D) / 8) + 0.5 * r.standard_normal(N)).astype(np.float32)

    def build(backend, dtype):
        X, y, w, W2 = A.var("X", 2, dtype), A.var("y", 1, dtype), A.var("w", 1, dtype), A.var("W2", 2, dtype)
        P0, tau = A.var("P0", 2, dtype), A.var("tau", 0, dtype)
        return MeanFieldVMP(blr_log_joint(X, y, w, W2, tau, P0, 1.0, 1.0),
                            [MVNormalNode(w, W2, np.zeros(D), np.eye(D)), GammaNode(tau)],
                            dict(X=Xs, y=ys, P0=np.eye(D, dtype=np.float32)), backend=backend)

    dev, ref = build(DeviceBackend(ctx), "float32"), build(B64, "float64")
    last = -np.inf
    for _ in range(3):
        for name in ("w", "tau"):
            dev.update(name)
            ref.update(name)
            b_dev, b_ref = dev.elbo(), ref.elbo()
            npt.assert_allclose(b_dev, b_ref, rtol=2e-5)
            assert b_dev >= last - 1e-4 * abs(b_dev)
            last = b_dev


def test_stochastic_updates_on_mini_batches_converge_to_the_full_batch_posterior():
    """README.md:69-79: mini-batches, the data term scaled by N / B, Robbins-Monro steps.  Known
    variance, unknown mean: the fixed point is the exact posterior of the whole data set."""
    N, B, v, m0, v0 = 4000, 100, 2.0, 0.0, 50.0
    xs = rs.standard_normal(N) * np.sqrt(v) + 1.7
    x, mu, scale = f64("x", 1), f64("mu", 0), f64("scale", 0)
    lj = (A.sum(x * mu) * (1.0 / v) + A.shape(x, 0) * ((mu ** 2) * (-0.5 / v))) * scale \
        + mu * (m0 / v0) + (mu ** 2) * (-0.5 / v0)
    q = NormalNode(mu)
    vmp = MeanFieldVMP(lj, [q], dict(x=xs[:B], scale=np.asarray(N / B)), backend=B64)
    order = rs.permutation(N)
    for t in range(400):
        idx = order[(t * B) % N:(t * B) % N + B]
        vmp.set_data(x=xs[idx])
        vmp.update("mu", rho=(t + 2.0) ** -0.7)
    post_v = 1.0 / (N / v + 1.0 / v0)
    post_m = post_v * (xs.sum() / v + m0 / v0)
    npt.assert_allclose(q.variance, post_v, rtol=1e-6)      # the precision message is the same for every batch
    npt.assert_allclose(q.mean, post_m, atol=4 * np.sqrt(post_v))
    with pytest.raises(TypeError):
        vmp.set_data(nope=xs)


# ---- general reparameterisation-trick engine ----------------------------------------------------

def _linear_model(dtype, s2):
    X, y, W = A.var("X", 2, dtype), A.var("y", 1, dtype), A.var("W", 2, dtype)     # W: [S, D]
    r = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)                                   # [S, N]
    return A.sum(r * r, axis=1) * (-0.5 / s2) + A.sum(W * W, axis=1) * (-0.5), W


def _mc_close(estimate, exact, samples):
    """|mean of the samples - exact| within five standard errors (plus float slack)."""
    se = samples.std(axis=0, ddof=1) / np.sqrt(samples.shape[0])
    assert (np.abs(estimate - exact) <= 5.0 * se + 1e-9 * (1.0 + np.abs(exact))).all(), \
        (estimate, exact, se)


def test_pathwise_gradient_is_the_gradient_of_the_gaussian_elbo():
    """Quadratic log-joint, Gaussian q: the ELBO and its gradient are known in closed form; the
    estimator must agree within Monte-Carlo error, and ascent must find the mean-field optimum."""
    from bayesic_amd.inference import ReparamVI
    r = np.random.RandomState(2718)
    N, D, s2, S = 300, 5, 0.5, 4096
    Xs = r.standard_normal((N, D))
    ys = Xs @ r.standard_normal(D) + np.sqrt(s2) * r.standard_normal(N)
    lj, W = _linear_model("float64", s2)
    eps0 = np.random.RandomState(77).standard_normal((S, D))
    eng = ReparamVI(lj, [(W, D)], dict(X=Xs, y=ys), n_samples=S, backend=B64, lr=0.05,
                    noise=lambda step: eps0)
    eng.lam[:D] = 0.3 * r.standard_normal(D)
    eng.lam[D:] = np.log(0.2)
    elbo, grad = eng.estimate(0)
    prec = Xs.T @ Xs / s2 + np.eye(D)
    mu, sig = eng.lam[:D], np.exp(eng.lam[D:])
    f, g = eng.log_joint_and_gradient(mu[None, :] + sig[None, :] * eps0)
    _mc_close(grad[:D], Xs.T @ (ys - Xs @ mu) / s2 - mu, g)
    _mc_close(grad[D:], -sig ** 2 * np.diag(prec) + 1.0, g * eps0 * sig[None, :] + 1.0)
    exact = -0.5 / s2 * ((ys - Xs @ mu) ** 2).sum() - 0.5 * mu @ mu - 0.5 * (sig ** 2 * np.diag(prec)).sum() \
        + eng.lam[D:].sum() + 0.5 * D * (1 + np.log(2 * np.pi))
    _mc_close(elbo, exact, f + eng.lam[D:].sum() + 0.5 * D * (1 + np.log(2 * np.pi)))
    small = ReparamVI(lj, [(W, D)], dict(X=Xs, y=ys), n_samples=64, backend=B64, lr=0.05,
                      noise=lambda step: np.random.RandomState(step).standard_normal((64, D)))
    for _ in range(600):
        small.step()
    post_mean = np.linalg.solve(prec, Xs.T @ ys / s2)
    npt.assert_allclose(small.lam[:D], post_mean, atol=0.05)
    npt.assert_allclose(np.exp(small.lam[D:]), 1.0 / np.sqrt(np.diag(prec)), rtol=0.3)   # 64 draws a step


@pytest.mark.gpu
def test_reparam_engine_on_device_shares_draws_with_the_score_function_engine(ctx):
    """Same model, same Philox noise: the two estimators' ELBO values agree, their gradients agree
    within Monte-Carlo error (the pathwise one with far less variance), and the device gradient of
    log p equals the float64 one."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference import ReparamVI, ScoreFunctionVI
    N, D, s2, S = 20_000, 8, 0.5, 512
    r = np.random.RandomState(9)
    Xs = r.standard_normal((N, D)).astype(np.float32)
    ys = (Xs @ r.standard_normal(D) + np.sqrt(s2) * r.standard_normal(N)).astype(np.float32)
    lj, W = _linear_model("float32", s2)
    lam0 = np.concatenate([0.1 * r.standard_normal(D), np.full(D, np.log(0.01))])
    dev = DeviceBackend(ctx)
    rp = ReparamVI(lj, [(W, D)], dict(X=Xs, y=ys), n_samples=S, seed=5, backend=dev, lam0=lam0)
    sf = ScoreFunctionVI(lj, [(W, D)], dict(X=Xs, y=ys), n_samples=S, seed=5, backend=dev, lam0=lam0)
    np.testing.assert_array_equal(rp.draw(3), sf.draw(3))
    e_rp, g_rp = rp.estimate(3)
    e_sf, g_sf, _ = sf.estimate(3)
    npt.assert_allclose(e_rp, e_sf, rtol=2e-3)      # same f values; the entropy enters exactly vs sampled
    # exact gradient
    X64, y64 = Xs.astype(np.float64), ys.astype(np.float64)
    prec = X64.T @ X64 / s2 + np.eye(D)
    mu, sig2 = lam0[:D], np.exp(2 * lam0[D:])
    g_mu = X64.T @ (y64 - X64 @ mu) / s2 - mu
    eps3, sig = rp.draw(3), np.exp(lam0[D:])
    z = mu[None, :] + sig[None, :] * eps3
    f_dev, g_dev = rp.log_joint_and_gradient(z)
    _mc_close(g_rp[:D], g_mu, g_dev)
    _mc_close(g_rp[D:], -sig2 * np.diag(prec) + 1.0, g_dev * eps3 * sig[None, :] + 1.0)
    assert np.abs(g_rp[:D] - g_mu).max() < np.abs(g_sf[:D] - g_mu).max()      # the point of the trick
    # the device derivative of log p itself against float64 on the same draws
    lj64, W64 = _linear_model("float64", s2)
    ref = ReparamVI(lj64, [(W64, D)], dict(X=X64, y=y64), n_samples=S, backend=B64, lam0=lam0,
                    noise=lambda step: rp.draw(step))
    f_ref, g_ref = ref.log_joint_and_gradient(z.astype(np.float32).astype(np.float64))
    npt.assert_allclose(f_dev, f_ref, rtol=2e-5)
    npt.assert_allclose(g_dev, g_ref, rtol=2e-3, atol=2e-3 * np.abs(g_ref).max())


def test_reparam_engine_on_mini_batches_reaches_the_full_data_optimum():
    """Stochastic optimisation proper: a fresh mini-batch every step, the data term scaled by
    N / B inside the log-joint (README.md:69-79)."""
    from bayesic_amd.inference import ReparamVI
    r = np.random.RandomState(99)
    N, B, D, s2 = 2000, 100, 4, 0.5
    Xs = r.standard_normal((N, D))
    ys = Xs @ r.standard_normal(D) + np.sqrt(s2) * r.standard_normal(N)
    X, y, W = f64("X", 2), f64("y", 1), f64("W", 2)
    res = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
    lj = A.sum(res * res, axis=1) * (-0.5 / s2 * (N / B)) + A.sum(W * W, axis=1) * (-0.5)
    eng = ReparamVI(lj, [(W, D)], dict(X=Xs[:B], y=ys[:B]), n_samples=16, backend=B64, lr=0.02,
                    noise=lambda step: np.random.RandomState(1000 + step).standard_normal((16, D)))
    for t in range(1500):
        idx = r.randint(N, size=B)
        eng.set_data(X=Xs[idx], y=ys[idx])
        eng.step()
    prec = Xs.T @ Xs / s2 + np.eye(D)
    npt.assert_allclose(eng.lam[:D], np.linalg.solve(prec, Xs.T @ ys / s2), atol=0.05)
    with pytest.raises(TypeError):
        eng.set_data(W=np.zeros((16, D)))


def test_elbo_sees_data_that_occurs_only_in_latent_free_terms_and_untouched_statistics_decay():
    """(1) An input of the log-joint that no message depends on (here the offsets z of a
    latent-free term) must still be uploaded: elbo() evaluates that term.  (2) A statistic no
    term touches receives the message 0, so a damped update shrinks its natural parameter
    instead of leaving the initial value in place."""
    x, z, mu = f64("x", 1), f64("z", 1), f64("mu", 0)
    xs, zs = rs.standard_normal(40) + 1.0, rs.standard_normal(40)
    # known unit variance, flat prior on mu: the log-joint is linear + quadratic in mu ...
    lj = A.sum(x * mu) - 0.5 * A.shape(x, 0) * (mu ** 2) - 0.5 * A.sum(z * z)
    node = NormalNode(mu)
    vmp = MeanFieldVMP(lj, [node], {"x": xs, "z": zs}, backend=B64)
    vmp.sweep()
    npt.assert_allclose(node.mean, xs.mean(), rtol=1e-12)
    e = vmp.elbo()
    want = xs.sum() * node.mean - 0.5 * 40 * (node.mean ** 2 + node.variance) - 0.5 * (zs ** 2).sum() \
        + node.entropy()
    npt.assert_allclose(e, want, rtol=1e-12)
    vmp.set_data(z=2.0 * zs)                       # ... and set_data refreshes the bound's copy
    npt.assert_allclose(vmp.elbo(), want - 1.5 * (zs ** 2).sum(), rtol=1e-12)
    # (2): only the linear statistic has a coefficient here
    lin = A.sum(x * mu)
    (c1, c2), _ = conjugate_coefficients(lin, mu, (mu, mu ** 2))
    assert c1 is not None and c2 is None
    node2 = NormalNode(mu, mean=0.0, variance=1.0)
    before = float(node2.eta[1])
    MeanFieldVMP(lin, [node2], {"x": xs}, backend=B64).update("mu", rho=0.25)
    npt.assert_allclose(node2.eta[1], 0.75 * before, rtol=1e-14)
    npt.assert_allclose(node2.eta[0], 0.25 * xs.sum(), rtol=1e-12)


# ---- config 3's model, updates derived: resident assignments, NormalGamma factors ---------------

def _cfg3_derived_and_oracle(backend, dtype, n, d, k, steps, resident=True, resident_globals=None):
    from bayesic_amd.inference.mixture import DiagonalMixtureVMP
    from oracle import svi
    X, _, _ = svi.make_cfg3(n, d, k)
    eta0 = svi.mog_prior_eta(k, d)
    eta = svi.mog_init_eta(X[:500], k, d, seed=2)
    alpha, m, kappa, a, b = svi.mog_unpack(eta, k, d)
    model = DiagonalMixtureVMP(X if dtype == "float32" else X.astype(np.float64), k, n_total=10.0 * n,
                               init=(alpha, m, kappa, a, b), backend=backend, dtype=dtype, resident=resident,
                               resident_globals=resident_globals, route="derived")
    for t in range(1, steps + 1):
        rho = (t + 1.0) ** -0.6
        model.step(rho)
        eta, _, _ = svi.mog_svi_step(eta, eta0, X, 10.0 * n, rho, k, d)
    return model, eta


def test_derived_diagonal_mixture_equals_the_config3_svi_step():
    """The mean-field engine's derived updates for config 3's model (Dirichlet weights, NormalGamma
    per component and column, resident N x K assignments) reproduce oracle.svi.mog_svi_step -- the
    update the fused kernels of svi/mog.py implement.  float64 backend; the oracle rounds the
    expected parameters to float32 (what the device streams), hence 1e-5."""
    model, eta = _cfg3_derived_and_oracle(B64, "float64", 3000, 5, 4, steps=3)
    npt.assert_allclose(model.eta_fused_layout(), eta, rtol=2e-5, atol=1e-6)
    # the bound is evaluable with resident assignments, and a further local update cannot lower it
    full = _cfg3_derived_and_oracle(B64, "float64", 3000, 5, 4, steps=1)[0]
    full.vmp.update("Z", 1.0, message_scale=1.0 / full.scale)
    assert np.isfinite(full.vmp.elbo())
    # resident and host-side assignments are the same update
    host, _ = _cfg3_derived_and_oracle(B64, "float64", 3000, 5, 4, steps=3, resident=False)
    npt.assert_allclose(model.eta_fused_layout(), host.eta_fused_layout(), rtol=1e-9)


def test_oracle_mog_elbo_equals_the_derived_engines_bound():
    """oracle.svi.mog_elbo (config 3's bound with q(z) collapsed to its optimum, what the fused path
    reports) against MeanFieldVMP.elbo() of the same model written as a symbolic log-joint, right after
    the local update: equal to 1e-9 once the constants the symbolic log-joint drops are added back
    (the -1/2 log 2 pi of every datum and column, and the priors' log-normalisers)."""
    from bayesic_amd.inference.mixture import DiagonalMixtureVMP
    from oracle import svi
    n, d, k = 800, 3, 4
    X = svi.make_cfg3(n, d, k)[0].astype(np.float64)
    eta0 = svi.mog_prior_eta(k, d)
    eta = svi.mog_init_eta(X[:300], k, d, seed=2)
    for rho in (1.0, 0.6):
        alpha, m, kappa, a, b = svi.mog_unpack(eta, k, d)
        model = DiagonalMixtureVMP(X, k, n_total=float(n), init=(alpha, m, kappa, a, b), backend=B64,
                                   dtype="float64", resident=False, resident_globals=False)
        model.vmp.update("Z", 1.0)
        Wmat, c = svi.mog_expected_params(eta, k, d, dtype=np.float64)
        _, lse = svi.mog_local_step(X, Wmat, c)
        alpha0, _, kappa0, a0, b0 = svi.mog_unpack(eta0, k, d)
        dropped = -0.5 * n * d * svi.LOG_2PI - float(svi.dirichlet_log_normalizer(alpha0)) \
            - float(svi.normal_gamma_log_normalizer(kappa0, a0, b0).sum())
        npt.assert_allclose(svi.mog_elbo(eta, eta0, lse, 1.0, k, d), model.vmp.elbo() + dropped, rtol=1e-9)
        eta = svi.natgrad_update(eta, eta0, svi.mog_message(svi.mog_local_step(X, Wmat, c)[0], k, d), 1.0, rho)


def test_derived_mixture_with_resident_global_factors():
    """resident_globals=True: the Dirichlet and NormalGamma factors keep their natural parameters and
    expectations on the backend (digamma, log, reciprocals as element-wise backend ops), so an update
    reads nothing back.  Same updates as the host-side factors (float64 backend) and as the oracle."""
    model, eta = _cfg3_derived_and_oracle(B64, "float64", 3000, 5, 4, steps=3, resident_globals=True)
    npt.assert_allclose(model.eta_fused_layout(), eta, rtol=2e-5, atol=1e-6)
    host, _ = _cfg3_derived_and_oracle(B64, "float64", 3000, 5, 4, steps=3, resident_globals=False)
    assert not host.resident_globals and model.resident_globals
    npt.assert_allclose(model.eta_fused_layout(), host.eta_fused_layout(), rtol=1e-6, atol=1e-9)
    npt.assert_allclose(model.vmp.elbo(), host.vmp.elbo(), rtol=1e-6)


@pytest.mark.gpu
def test_derived_mixture_with_resident_global_factors_on_device(ctx):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    model, eta = _cfg3_derived_and_oracle(DeviceBackend(ctx), "float32", 100_000, 16, 64, steps=2,
                                          resident_globals=True)
    got = model.eta_fused_layout()
    scale = np.maximum(np.abs(eta), 1.0)
    assert (np.abs(got - eta) <= 1e-3 * scale).all(), np.abs((got - eta) / scale).max()
    assert np.isfinite(model.vmp.elbo())


@pytest.mark.gpu
def test_derived_diagonal_mixture_on_device_100k_rows(ctx):
    """The same on the MI355X backend over a 100 000-row slice of config 3 (K = 64, D = 16): the
    assignments, their softmax and every responsibility-weighted message stay on the device."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    model, eta = _cfg3_derived_and_oracle(DeviceBackend(ctx), "float32", 100_000, 16, 64, steps=2)
    got = model.eta_fused_layout()
    scale = np.maximum(np.abs(eta), 1.0)
    # float32 data, expectations and GEMM partial sums against the float64 oracle
    assert (np.abs(got - eta) <= 1e-3 * scale).all(), np.abs((got - eta) / scale).max()
    import torch
    # never left the device: the responsibilities are a device tensor, and the logits were not even
    # stored (their softmax was taken inside the product, with the 1 / (N / B) of a local latent as
    # the kernel's multiplier -- the oracle comparison above is that kernel's parity check)
    # ... and, since round 3, not even written: the neighbours only ask for statistics of the responsibilities
    # against the features the logits were formed from (tests/test_softmax_stats_gpu.py)
    from bayesic_amd.algebra.device_backend import DeferredSoftmax
    r = model.z.expectations_backend()[0]
    assert isinstance(r, DeferredSoftmax) and model.z.eta[0] is model.z.FUSED
    assert r.tensor().is_cuda and tuple(r.tensor().shape) == (100_000, 64)


@pytest.mark.gpu
def test_softmax_inside_the_logits_product_is_the_same_update(ctx):
    """A resident Categorical node's update (rho = 1) takes its softmax INSIDE the product that
    forms the logits (DeviceBackend.evaluate_softmax_rows -> bsc_gemm_softmax_rows: the logits are
    never stored).  Same responsibilities, same global updates, same bound as the route that
    materialises them (MeanFieldVMP.fuse_softmax = False)."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference.mixture import DiagonalMixtureVMP
    from oracle import svi
    n, d, k = 30_011, 16, 64
    X, _, _ = svi.make_cfg3(n, d, k)
    eta = svi.mog_init_eta(X[:500], k, d, seed=2)
    alpha, m, kappa, a, b = svi.mog_unpack(eta, k, d)
    models = []
    for fuse in (True, False):
        model = DiagonalMixtureVMP(X, k, n_total=float(n), init=(alpha, m, kappa, a, b),
                                   backend=DeviceBackend(ctx), dtype="float32", route="derived")
        model.vmp.fuse_softmax = fuse
        model.vmp.defer_responsibilities = False      # this test is about WRITING the responsibilities in one pass
        calls = []
        real = ctx.call

        def spy(name, *args, _calls=calls, _real=real):
            _calls.append(name)
            return _real(name, *args)

        ctx.call = spy
        try:
            for t in range(1, 4):
                model.step((t + 1.0) ** -0.6)
        finally:
            ctx.call = real
        models.append((model, calls))
    (fused, fused_calls), (plain, plain_calls) = models
    assert fused_calls.count("bsc_gemm_softmax_rows") == 3 and fused_calls.count("bsc_softmax_rows") == 0
    assert plain_calls.count("bsc_gemm_softmax_rows") == 0 and plain_calls.count("bsc_softmax_rows") == 3
    assert fused.z.eta[0] is fused.z.FUSED
    # (float32 responsibilities from two different summation orders, three damped updates on)
    npt.assert_allclose(fused.eta_fused_layout(), plain.eta_fused_layout(), rtol=1e-3, atol=1e-3)
    rf = fused.vmp.backend.to_host(fused.z.expectations_backend()[0])
    rp = plain.vmp.backend.to_host(plain.z.expectations_backend()[0])
    npt.assert_allclose(rf, rp, rtol=1e-3, atol=1e-5)
    npt.assert_allclose(fused.z.entropy(), plain.z.entropy(), rtol=1e-4)
    npt.assert_allclose(fused.vmp.elbo(), plain.vmp.elbo(), rtol=1e-5)


@pytest.mark.gpu
def test_successive_models_on_one_backend_do_not_share_cached_constants(ctx):
    """the executor caches values of marked constants (X * X, the wide operand
    [X | X^2 | 1], R^T [X | X^2 | 1]) keyed by ADDRESS.  A second same-shaped model built on the same backend
    after the first was dropped used to receive the first model's block from the allocator -- and its cached
    values.  The marks now hold the tensors (an address in use cannot be handed out again) and a model takes
    its marks back when it is closed or collected.  Both orders: first model dropped, and both alive."""
    import gc
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference.mixture import DiagonalMixtureVMP
    from oracle import svi
    be = DeviceBackend(ctx)
    n, d, k = 20_000, 16, 8

    def build(seed):
        rs = np.random.RandomState(seed)
        centres = rs.standard_normal((k, d)) * 4.0
        X = (centres[rs.randint(k, size=n)] + rs.standard_normal((n, d))).astype(np.float32)
        eta = svi.mog_init_eta(X[:500], k, d, seed=2)
        alpha, m, kappa, a, b = svi.mog_unpack(eta, k, d)
        return X, eta, DiagonalMixtureVMP(X, k, n_total=float(n), init=(alpha, m, kappa, a, b), backend=be,
                                          route="derived")

    def check(model, X, eta):
        eta0 = svi.mog_prior_eta(k, d)
        for t in range(1, 3):
            rho = (t + 1.0) ** -0.6
            model.step(rho)
            eta, _, _ = svi.mog_svi_step(eta, eta0, X, float(n), rho, k, d)
        got = model.eta_fused_layout()
        scale = np.maximum(np.abs(eta), 1.0)
        assert (np.abs(got - eta) <= 1e-3 * scale).all(), np.abs((got - eta) / scale).max()

    X1, eta1, first = build(11)
    check(first, X1, eta1)
    cached = len(be._const_cache)
    assert cached > 0
    del first
    gc.collect()
    assert len(be._const_cache) == 0 and not be._const and not be._const_ptrs   # the dropped model took its marks and their values
    X2, eta2, second = build(12)                                 # (same shapes: the allocator reuses the blocks)
    check(second, X2, eta2)
    X3, eta3, third = build(13)                                  # two live models on one backend
    check(third, X3, eta3)
    second.close()
    assert be._const                                             # the third model's marks survive the second's close
    third.close()
    assert not be._const and not be._const_cache and not be._const_ptrs


# ---- resident (backend-side) variants of every factor: same updates, nothing read back ---------------------------

def _counting(backend):
    """to_host calls made through `backend` from now on."""
    calls = []
    real = backend.to_host

    def spy(value):
        calls.append(1)
        return real(value)
    backend.to_host = spy
    return calls


def _regression_models(Xs, ys, P0s, a0, b0, backend, resident, fl):
    from bayesic_amd.inference import MVNormalNode, ResidentGammaNode, ResidentMVNormalNode
    D = Xs.shape[1]
    X, y, w, W2, P0, tau = fl("X", 2), fl("y", 1), fl("w", 1), fl("W2", 2), fl("P0", 2), fl("tau", 0)
    lj = blr_log_joint(X, y, w, W2, tau, P0, a0, b0)
    MV, Ga = (ResidentMVNormalNode, ResidentGammaNode) if resident else (MVNormalNode, GammaNode)
    qw = MV(w, W2, mean=np.zeros(D), covariance=np.eye(D))
    qt = Ga(tau, shape=1.0, rate=1.0)
    return MeanFieldVMP(lj, [qw, qt], dict(X=Xs, y=ys, P0=P0s), backend=backend), qw, qt


def test_resident_factors_give_the_host_factors_updates_on_the_reference_backend():
    """ResidentMVNormalNode + ResidentGammaNode (a scalar factor) against MVNormalNode + GammaNode: the same
    coordinate ascent, the same bound, float64 both."""
    N, D, a0, b0 = 400, 5, 2.0, 1.5
    Xs = rs.standard_normal((N, D))
    ys = Xs @ rs.standard_normal(D) + 0.3 * rs.standard_normal(N)
    P0s = np.eye(D) * 0.7
    host, hw, ht = _regression_models(Xs, ys, P0s, a0, b0, B64, False, f64)
    res, rw, rt = _regression_models(Xs, ys, P0s, a0, b0, B64, True, f64)
    for _ in range(5):
        host.sweep()
        res.sweep()
        npt.assert_allclose(res.elbo(), host.elbo(), rtol=1e-10)
    npt.assert_allclose(rt.host_copy().shape, ht.shape, rtol=1e-12)
    npt.assert_allclose(rt.host_copy().rate, ht.rate, rtol=1e-10)
    npt.assert_allclose(rw.host_copy().mean, hw.mean, rtol=1e-9)
    npt.assert_allclose(rw.host_copy().precision, hw.precision, rtol=1e-9)


def _normal_wishart_model(Xs, k0, nu0, backend, resident, fl):
    from bayesic_amd.distribution import logdet
    from bayesic_amd.inference import MVNormalNode, ResidentMVNormalNode, ResidentWishartNode, WishartNode
    D = Xs.shape[1]
    X, mu, M2, Lam, Id = fl("X", 2), fl("mu", 1), fl("M2", 2), fl("Lam", 2), fl("Id", 2)
    Nn = A.shape(X, 0)
    sx = A.sum(X, axis=0)
    lj = Nn * (0.5 * logdet(Lam)) + A.sum(Lam * A.dot(X.T, X)) * (-0.5) \
        + A.sum(Lam * A.outer(sx, mu)) + Nn * (A.sum(Lam * M2) * (-0.5)) \
        + A.sum(M2 * Id) * (-0.5 * k0) \
        + (0.5 * (nu0 - D - 1.0)) * logdet(Lam) + A.sum(Lam * Id) * (-0.5 * nu0)
    MV, Wi = (ResidentMVNormalNode, ResidentWishartNode) if resident else (MVNormalNode, WishartNode)
    qm = MV(mu, M2, mean=np.zeros(D), covariance=np.eye(D))
    ql = Wi(Lam, dof=nu0, scale=np.eye(D) / nu0)
    return MeanFieldVMP(lj, [qm, ql], dict(X=Xs, Id=np.eye(D, dtype=Xs.dtype)), backend=backend), qm, ql


def test_resident_normal_wishart_factors_on_the_reference_backend():
    N, D, k0, nu0 = 300, 3, 0.1, 5.0
    Ltrue = np.array([[2.0, 0.5, 0.0], [0.5, 1.0, 0.2], [0.0, 0.2, 1.5]])
    Xs = rs.multivariate_normal([1.0, -2.0, 0.5], np.linalg.inv(Ltrue), size=N)
    host, hm, hl = _normal_wishart_model(Xs, k0, nu0, B64, False, f64)
    res, rm, rl = _normal_wishart_model(Xs, k0, nu0, B64, True, f64)
    for _ in range(5):
        host.sweep()
        res.sweep()
        npt.assert_allclose(res.elbo(), host.elbo(), rtol=1e-10)
    npt.assert_allclose(rl.host_copy().dof, hl.dof, rtol=1e-12)
    npt.assert_allclose(rl.host_copy().scale, hl.scale, rtol=1e-9)
    npt.assert_allclose(rm.host_copy().mean, hm.mean, rtol=1e-9)


def _scalar_family_model(xs, backend, resident, fl):
    """x_n ~ N(mu, 1 / tau), mu ~ N(0, 10), tau ~ Gamma(1, 1); and a scale s2 ~ InvGamma(2, 1) on a second sample."""
    from bayesic_amd.inference import (InverseGammaNode, ResidentGammaNode, ResidentInverseGammaNode,
                                       ResidentNormalNode)
    x, z, mu, tau, s2 = fl("x", 1), fl("z", 1), fl("mu", 0), fl("tau", 0), fl("s2", 0)
    n = A.shape(x, 0)
    lj = tau * (A.sum(x * x) - 2.0 * mu * A.sum(x) + n * (mu ** 2)) * (-0.5) + n * (0.5 * A.log(tau)) \
        + (mu ** 2) * (-0.05) - tau \
        + (s2 ** -1) * A.sum(z * z) * (-0.5) - (0.5 * A.shape(z, 0) + 3.0) * A.log(s2) - (s2 ** -1)
    No, Ga, IG = (ResidentNormalNode, ResidentGammaNode, ResidentInverseGammaNode) if resident else \
        (NormalNode, GammaNode, InverseGammaNode)
    nodes = [No(mu, mean=0.0, variance=1.0), Ga(tau, shape=1.0, rate=1.0), IG(s2, shape=2.0, scale=1.0)]
    return MeanFieldVMP(lj, nodes, dict(x=xs, z=(xs * 0.5 + 1.0).astype(xs.dtype)), backend=backend), nodes


def test_resident_scalar_normal_gamma_inverse_gamma_factors_on_the_reference_backend():
    xs = 2.0 + 0.5 * rs.standard_normal(500)
    host, hn = _scalar_family_model(xs, B64, False, f64)
    res, rn = _scalar_family_model(xs, B64, True, f64)
    for _ in range(5):
        host.sweep()
        res.sweep()
        npt.assert_allclose(res.elbo(), host.elbo(), rtol=1e-10)
    for h, r in zip(hn, rn):
        for a, b in zip(h.eta, r.host_copy().eta):
            npt.assert_allclose(b, a, rtol=1e-10)


@pytest.mark.gpu
def test_resident_factors_on_device_read_nothing_back(ctx):
    """The three models above on the device with every factor resident: the updates match the float64 host-side
    factors' and NOT ONE value crosses to the host during the sweeps (to_host is counted)."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    r = np.random.RandomState(11)
    # regression: D x D precision message, scalar Gamma factor
    N, D, a0, b0 = 20_000, 64, 1.0, 1.0
    Xs = r.standard_normal((N, D)).astype(np.float32)
    ys = (Xs @ (r.standard_normal(D) / 8) + 0.5 * r.standard_normal(N)).astype(np.float32)
    P0s = np.eye(D, dtype=np.float32)
    backend = DeviceBackend(ctx)
    dev, qw, qt = _regression_models(Xs, ys, P0s, a0, b0, backend, True, lambda n, k: A.var(n, k))
    calls = _counting(backend)
    for _ in range(4):
        dev.sweep()
    assert not calls
    m, lam, a, b = blr_mean_field_by_hand(Xs.astype(np.float64), ys.astype(np.float64), P0s.astype(np.float64), a0, b0, 4)
    npt.assert_allclose(qt.host_copy().shape, a, rtol=1e-6)
    npt.assert_allclose(qt.host_copy().rate, b, rtol=1e-4)
    npt.assert_allclose(qw.host_copy().mean, m, rtol=2e-3, atol=2e-5)
    dev.close()
    # mean and precision matrix of a Gaussian
    Ltrue = np.array([[2.0, 0.5, 0.0], [0.5, 1.0, 0.2], [0.0, 0.2, 1.5]])
    Xg = r.multivariate_normal([1.0, -2.0, 0.5], np.linalg.inv(Ltrue), size=3000).astype(np.float32)
    host, hm, hl = _normal_wishart_model(Xg.astype(np.float64), 0.1, 5.0, B64, False, f64)
    backend = DeviceBackend(ctx)
    dev, rm, rl = _normal_wishart_model(Xg, 0.1, 5.0, backend, True, lambda n, k: A.var(n, k))
    calls = _counting(backend)
    for _ in range(5):
        host.sweep()
        dev.sweep()
    assert not calls
    npt.assert_allclose(rl.host_copy().dof, hl.dof, rtol=1e-6)
    npt.assert_allclose(rl.host_copy().scale, hl.scale, rtol=2e-4, atol=1e-6)
    npt.assert_allclose(rm.host_copy().mean, hm.mean, rtol=2e-4)
    npt.assert_allclose(dev.elbo(), host.elbo(), rtol=2e-5)
    dev.close()
    # scalar Normal / Gamma / InverseGamma factors
    xs = (2.0 + 0.5 * r.standard_normal(5000)).astype(np.float32)
    host, hn = _scalar_family_model(xs.astype(np.float64), B64, False, f64)
    backend = DeviceBackend(ctx)
    dev, rn = _scalar_family_model(xs, backend, True, lambda n, k: A.var(n, k))
    calls = _counting(backend)
    for _ in range(5):
        host.sweep()
        dev.sweep()
    assert not calls
    for h, rnode in zip(hn, rn):
        for a_, b_ in zip(h.eta, rnode.host_copy().eta):
            npt.assert_allclose(b_, a_, rtol=2e-4)
    dev.close()


@pytest.mark.gpu
def test_inverse_spd_on_device(ctx):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    r = np.random.RandomState(5)
    b = DeviceBackend(ctx)
    for batch, n in [((), 1), ((), 7), ((3,), 16), ((2, 2), 33), ((), 256)]:
        M = r.standard_normal(batch + (n, n + 3))
        Aspd = (M @ np.swapaxes(M, -1, -2) + n * np.eye(n)).astype(np.float32)
        got = b.to_host(b.inverse_spd(b.from_host(Aspd, "float32", len(batch) + 2)))
        npt.assert_allclose(got, np.linalg.inv(Aspd.astype(np.float64)), rtol=2e-5, atol=1e-7)
