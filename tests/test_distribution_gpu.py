"""Distribution nodes evaluated on the MI355X backend vs scipy.stats (float32
data; rtol 1e-5 as for any contraction, bayesic/tests/test_algebra.py:82)."""
import math

import numpy as np
import numpy.testing as npt
import pytest
import scipy.stats as st

from bayesic_amd.algebra import var
from bayesic_amd.distribution import MultivariateNormal, Normal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(ctx):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    return DeviceBackend(ctx)


def test_normal_iid_loglik_and_statistics_on_device(dev):
    rs = np.random.RandomState(0)
    xs = (rs.standard_normal(200_000) * 1.5 + 2.0).astype(np.float32)
    n = Normal().iid(1)
    x, m, v = var("x", 1), var("m", 0), var("v", 0)
    got = n.log_likelihood(x, mean=m, variance=v).compile(dev)(x=xs, m=np.float32(1.9), v=np.float32(2.1))
    want = st.norm.logpdf(xs.astype(np.float64), np.float32(1.9), math.sqrt(np.float32(2.1))).sum()
    npt.assert_allclose(got, want, rtol=1e-5)
    s1, s2 = n.sufficient_statistics(x)
    npt.assert_allclose(s1.compile(dev)(x=xs), xs.astype(np.float64).sum(), rtol=1e-6)
    npt.assert_allclose(s2.compile(dev)(x=xs), (xs.astype(np.float64) ** 2).sum(), rtol=1e-6)


def test_mvn_iid_on_device_and_logdet(dev):
    rs = np.random.RandomState(1)
    D, N = 32, 20_000
    Araw = rs.standard_normal((D, D))
    Lm = (Araw @ Araw.T / D + np.eye(D)).astype(np.float32)
    mus = rs.standard_normal(D).astype(np.float32) * 0.1
    Xs = rs.standard_normal((N, D)).astype(np.float32)
    mv = MultivariateNormal().iid(1)
    X, mu, L = var("X", 2), var("mu", 1), var("L", 2)
    got = mv.log_likelihood(X, mean=mu, precision=L).compile(dev)(X=Xs, mu=mus, L=Lm)
    want = st.multivariate_normal.logpdf(Xs.astype(np.float64), mus.astype(np.float64),
                                         np.linalg.inv(Lm.astype(np.float64))).sum()
    npt.assert_allclose(got, want, rtol=2e-5)
    s1, s2 = mv.sufficient_statistics(X)
    npt.assert_allclose(s2.compile(dev)(X=Xs), Xs.astype(np.float64).T @ Xs.astype(np.float64),
                        rtol=1e-4, atol=0.05)
    from bayesic_amd.distribution import logdet
    for n in (1, 3, 64, 200):
        A = rs.standard_normal((n, n))
        S = A @ A.T + n * np.eye(n)
        M = var("M", 2, "float64")
        npt.assert_allclose(logdet(M).compile(dev)(M=S), np.linalg.slogdet(S)[1], rtol=1e-10)
