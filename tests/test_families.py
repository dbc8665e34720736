"""Exponential-family nodes beyond Normal / MVN against
scipy.stats known answers, evaluated in float64 by the oracle's numpy backend.
The reference defines none of these (bayesic/distribution/core.py has Normal and
MultivariateNormal only); the contract is bayesic/distribution/base.py:47-95,
271-314: log-likelihood = data term + <t(x), eta(theta)> - A(theta), one value per
observation for extra leading dimensions, statistics of iid draws summed."""
import numpy as np
import numpy.testing as npt
import pytest
import scipy.special as sp
import scipy.stats as st

from bayesic_amd.algebra import var
from bayesic_amd.distribution import (Bernoulli, Categorical, Dirichlet, Gamma, InverseGamma,
                                      Multinomial, Normal, Wishart, digamma, gammaln)
from oracle.einsum_eval import NumpyBackend

B64 = NumpyBackend(np.float64)
rs = np.random.RandomState(7)


def run(expr, **inputs):
    return expr.compile(B64)(**inputs)


def f64(name, ndim):
    return var(name, ndim, "float64")


def test_special_functions():
    x = f64("x", 1)
    xs = rs.uniform(0.05, 30.0, 50)
    npt.assert_allclose(run(gammaln(x), x=xs), sp.gammaln(xs), rtol=1e-13)
    npt.assert_allclose(run(digamma(x), x=xs), sp.digamma(xs), rtol=1e-13)


def test_gamma_and_inverse_gamma_match_scipy():
    x, a, b = f64("x", 1), f64("a", 1), f64("b", 1)
    xs, av, bv = rs.uniform(0.1, 5, 20), rs.uniform(0.5, 6, 20), rs.uniform(0.3, 4, 20)
    ll = Gamma().log_likelihood(x, shape=a, rate=b)
    assert ll.ndim == 1
    npt.assert_allclose(run(ll, x=xs, a=av, b=bv), st.gamma.logpdf(xs, av, scale=1 / bv), rtol=1e-12)
    ll = InverseGamma().log_likelihood(x, shape=a, scale=b)
    npt.assert_allclose(run(ll, x=xs, a=av, b=bv), st.invgamma.logpdf(xs, av, scale=bv), rtol=1e-12)
    # expectations of the sufficient statistics: E[log x], E[x] and E[log x], E[1/x]
    e1, e2 = Gamma().expected_sufficient_statistics(shape=a, rate=b)
    npt.assert_allclose(run(e1, a=av, b=bv), sp.digamma(av) - np.log(bv), rtol=1e-12)
    npt.assert_allclose(run(e2, a=av, b=bv), av / bv, rtol=1e-12)
    e1, e2 = InverseGamma().expected_sufficient_statistics(shape=a, scale=b)
    npt.assert_allclose(run(e1, a=av, b=bv), np.log(bv) - sp.digamma(av), rtol=1e-12)
    npt.assert_allclose(run(e2, a=av, b=bv), av / bv, rtol=1e-12)


def test_gamma_iid_statistics_and_loglik():
    g = Gamma().iid(1)
    x, a, b = f64("x", 1), f64("a", 0), f64("b", 0)
    xs = rs.gamma(3.0, 0.5, 400)
    s1, s2 = g.sufficient_statistics(x)
    npt.assert_allclose(run(s1, x=xs), np.log(xs).sum(), rtol=1e-12)
    npt.assert_allclose(run(s2, x=xs), xs.sum(), rtol=1e-12)
    npt.assert_allclose(run(g.log_likelihood(x, shape=a, rate=b), x=xs, a=2.7, b=1.9),
                        st.gamma.logpdf(xs, 2.7, scale=1 / 1.9).sum(), rtol=1e-12)


def test_bernoulli_and_categorical_match_scipy():
    x, p = f64("x", 1), f64("p", 1)
    xs = rs.randint(0, 2, 30).astype(np.float64)
    ps = rs.uniform(0.05, 0.95, 30)
    bern = Bernoulli()
    assert bern.is_discrete()
    npt.assert_allclose(run(bern.log_likelihood(x, probability=p), x=xs, p=ps),
                        st.bernoulli.logpmf(xs, ps), rtol=1e-12)
    X, P = f64("X", 2), f64("P", 2)
    probs = rs.dirichlet(np.ones(6), 12)
    onehot = np.eye(6)[rs.randint(0, 6, 12)]
    cat = Categorical()
    ll = cat.log_likelihood(X, probabilities=P)
    assert ll.ndim == 1
    npt.assert_allclose(run(ll, X=onehot, P=probs), np.log((probs * onehot).sum(1)), rtol=1e-12)
    # unnormalised weights are normalised by the log-normaliser
    npt.assert_allclose(run(ll, X=onehot, P=3.0 * probs), np.log((probs * onehot).sum(1)), rtol=1e-12)
    (e,) = cat.expected_sufficient_statistics(probabilities=P)
    npt.assert_allclose(run(e, P=3.0 * probs), probs, rtol=1e-12)


def test_multinomial_matches_scipy():
    X, P = f64("X", 2), f64("P", 2)
    n = 25
    probs = rs.dirichlet(np.ones(5), 9)
    counts = np.stack([rs.multinomial(n, pr) for pr in probs]).astype(np.float64)
    m = Multinomial(total_count=n)
    want = np.array([st.multinomial.logpmf(c, n, pr) for c, pr in zip(counts, probs)])
    npt.assert_allclose(run(m.log_likelihood(X, probabilities=P), X=counts, P=probs), want, rtol=1e-11)
    (e,) = m.expected_sufficient_statistics(probabilities=P)
    npt.assert_allclose(run(e, P=probs), n * probs, rtol=1e-12)


def test_dirichlet_matches_scipy():
    X, Al = f64("X", 2), f64("Al", 2)
    alpha = rs.uniform(0.5, 4.0, (8, 5))
    xs = np.stack([rs.dirichlet(a) for a in alpha])
    d = Dirichlet()
    want = np.array([st.dirichlet.logpdf(x, a) for x, a in zip(xs, alpha)])
    npt.assert_allclose(run(d.log_likelihood(X, concentration=Al), X=xs, Al=alpha), want, rtol=1e-11)
    (e,) = d.expected_sufficient_statistics(concentration=Al)
    npt.assert_allclose(run(e, Al=alpha), sp.digamma(alpha) - sp.digamma(alpha.sum(1))[:, None],
                        rtol=1e-12)
    # single observation, no leading dimension
    x1, a1 = f64("x1", 1), f64("a1", 1)
    npt.assert_allclose(run(d.log_likelihood(x1, concentration=a1), x1=xs[0], a1=alpha[0]), want[0],
                        rtol=1e-11)


def test_wishart_matches_scipy():
    D = 4
    w = Wishart(dim=D)
    X, W, nu = f64("X", 2), f64("W", 2), f64("nu", 0)
    A_ = rs.standard_normal((D, D))
    scale = A_ @ A_.T + D * np.eye(D)
    rate = np.linalg.inv(scale)
    for dof in (4.5, 7.0, 12.0):
        Xs = st.wishart.rvs(dof, scale, random_state=rs)
        got = run(w.log_likelihood(X, dof=nu, rate=W), X=Xs, W=rate, nu=dof)
        npt.assert_allclose(got, st.wishart.logpdf(Xs, dof, scale), rtol=1e-10)
    # per-observation leading dimension
    Xb, Wb, nub = f64("Xb", 3), f64("Wb", 3), f64("nub", 1)
    dofs = np.array([5.0, 6.5, 9.0])
    Xs = np.stack([st.wishart.rvs(d, scale, random_state=rs) for d in dofs])
    got = run(w.log_likelihood(Xb, dof=nub, rate=Wb), Xb=Xs, Wb=np.stack([rate] * 3), nub=dofs)
    want = [st.wishart.logpdf(x, d, scale) for x, d in zip(Xs, dofs)]
    npt.assert_allclose(got, want, rtol=1e-10)


def test_normal_expected_statistics():
    m, v = f64("m", 1), f64("v", 1)
    mv, vv = rs.standard_normal(5), rs.uniform(0.2, 2, 5)
    e1, e2 = Normal().expected_sufficient_statistics(mean=m, variance=v)
    npt.assert_allclose(run(e1, m=mv, v=vv), mv)
    npt.assert_allclose(run(e2, m=mv, v=vv), mv ** 2 + vv, rtol=1e-12)


@pytest.mark.gpu
def test_families_on_device(ctx):
    """The same nodes on the MI355X backend (float32 data; lgamma / digamma are unary ops
    of the fused map-reduce kernel).  Tolerance: float32 evaluation of sums of O(10)
    terms of magnitude O(10)."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    dev = DeviceBackend(ctx)
    x, a, b = var("x", 1), var("a", 1), var("b", 1)
    xs = rs.uniform(0.1, 5, 2000).astype(np.float32)
    av = rs.uniform(0.5, 6, 2000).astype(np.float32)
    bv = rs.uniform(0.3, 4, 2000).astype(np.float32)
    got = Gamma().log_likelihood(x, shape=a, rate=b).compile(dev)(x=xs, a=av, b=bv)
    npt.assert_allclose(got, st.gamma.logpdf(xs.astype(np.float64), av, scale=1 / bv.astype(np.float64)),
                        rtol=2e-5, atol=2e-5)
    npt.assert_allclose(gammaln(x).compile(dev)(x=xs), sp.gammaln(xs.astype(np.float64)), rtol=1e-6,
                        atol=1e-6)
    npt.assert_allclose(digamma(x).compile(dev)(x=xs), sp.digamma(xs.astype(np.float64)), rtol=1e-6,
                        atol=1e-6)
    X, Al = var("X", 2), var("Al", 2)
    alpha = rs.uniform(0.5, 4.0, (300, 7)).astype(np.float32)
    xd = np.stack([rs.dirichlet(a_) for a_ in alpha.astype(np.float64)]).astype(np.float32)
    d = Dirichlet()
    got = d.log_likelihood(X, concentration=Al).compile(dev)(X=xd, Al=alpha)
    want = np.array([st.dirichlet.logpdf(x_ / x_.sum(), a_) for x_, a_ in
                     zip(xd.astype(np.float64), alpha.astype(np.float64))])
    npt.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    (e,) = d.expected_sufficient_statistics(concentration=Al)
    npt.assert_allclose(e.compile(dev)(Al=alpha),
                        sp.digamma(alpha.astype(np.float64)) -
                        sp.digamma(alpha.astype(np.float64).sum(1))[:, None], rtol=1e-5, atol=1e-5)
    # iid Gamma draws: summed statistics in one fused pass each, then the log-likelihood
    g = Gamma().iid(1)
    a0, b0 = var("a0", 0), var("b0", 0)
    got = g.log_likelihood(x, shape=a0, rate=b0).compile(dev)(x=xs, a0=np.float32(2.7), b0=np.float32(1.9))
    npt.assert_allclose(got, st.gamma.logpdf(xs.astype(np.float64), 2.7, scale=1 / 1.9).sum(), rtol=1e-5)
    w = Wishart(dim=3)
    Xw, W, nu = var("Xw", 2), var("W", 2), var("nu", 0)
    scale = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, 0.2], [0.1, 0.2, 1.0]])
    Xs = st.wishart.rvs(6.0, scale, random_state=rs)
    got = w.log_likelihood(Xw, dof=nu, rate=W).compile(dev)(
        Xw=Xs.astype(np.float32), W=np.linalg.inv(scale).astype(np.float32), nu=np.float32(6.0))
    npt.assert_allclose(got, st.wishart.logpdf(Xs, 6.0, scale), rtol=1e-4)
