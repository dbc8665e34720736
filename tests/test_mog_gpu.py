"""GPU parity for BASELINE config 3 (mixture of Gaussians, discrete latent
marginalised by summation) vs the float64 oracle.

Tolerance: float32 operands, fp32 MFMA contractions (exact-f32 fma chains), fast
exp/log in the softmax (~2 ulp), float64 cross-tile finish: statistics within
2e-5 of the scale of the sums; the bound term rtol 2e-6; parameters derived in
float64 from identical inputs rtol 1e-10 (digamma: series vs scipy)."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

from oracle import svi

pytestmark = pytest.mark.gpu


def _estep(ctx, X, Wmat, c):
    from bayesic_amd._ffi import ptr
    K, twoD = Wmat.shape
    D = twoD // 2
    # keep every device buffer referenced until the (asynchronous) call has finished
    Xd = ctx.to_device(X) if X.shape[0] else ctx.zeros((1, D))
    Wd, cd = ctx.to_device(Wmat), ctx.to_device(c)
    stats, lse = ctx.zeros((K, 1 + 2 * D), torch.float64), ctx.zeros(1, torch.float64)
    ctx.call("bsc_mog_estep", ptr(Xd), D, X.shape[0], D, K, ptr(Wd), ptr(cd), ptr(stats), ptr(lse))
    ctx.sync()
    return stats.cpu().numpy(), lse.item()


@pytest.mark.parametrize("N,D,K", [(32, 16, 64), (1000, 16, 64), (1003, 16, 64), (7, 16, 64),
                                   (5000, 16, 33), (4097, 5, 64), (333, 1, 2), (20000, 12, 40),
                                   (0, 16, 64)])
def test_estep_matches_oracle(ctx, N, D, K):
    rs = np.random.RandomState(N + D + K)
    centres = rs.standard_normal((K, D)) * 3
    X = (centres[rs.randint(K, size=N)] + rs.standard_normal((N, D))).astype(np.float32)
    T = rs.uniform(0.5, 2.0, (K, D))
    Wmat = np.concatenate([T * centres, -0.5 * T], axis=1).astype(np.float32)
    c = (rs.standard_normal(K) - 0.5 * (T * centres ** 2).sum(1)).astype(np.float32)
    stats, lse = _estep(ctx, X, Wmat, c)
    want, lse_ref = svi.mog_local_step(X, Wmat, c) if N else (np.zeros((K, 1 + 2 * D)), 0.0)
    X64 = X.astype(np.float64)
    scale = np.concatenate([[max(N, 1)], np.abs(X64).sum(0) + 1e-9, (X64 ** 2).sum(0) + 1e-9])
    assert (np.abs(stats - want) <= 2e-5 * scale[None, :] + 1e-9).all(), \
        np.abs((stats - want) / scale[None, :]).max()
    npt.assert_allclose(stats[:, 0].sum(), N, rtol=1e-6, atol=1e-6)   # responsibilities sum to 1
    npt.assert_allclose(lse, lse_ref, rtol=2e-6, atol=1e-4)


def test_estep_exact_on_separated_integer_data(ctx):
    """Well-separated clusters: r is exactly one-hot in float32, so every statistic is
    an exact integer sum -- any row/component/feature mix-up in the MFMA operand
    layouts shows up as a wrong integer."""
    K, D, N = 64, 16, 64 * 40
    k = np.arange(K)
    centres = np.zeros((K, D))
    centres[k, k % D] = 100.0 * (1 + k // D)              # distinct, far apart
    labels = np.arange(N) % K
    offs = (np.arange(N)[:, None] * 7 + np.arange(D)[None, :] * 3) % 5 - 2
    X = (centres[labels] + offs).astype(np.float32)
    T = np.ones((K, D))
    Wmat = np.concatenate([T * centres, -0.5 * T], axis=1).astype(np.float32)
    c = (-0.5 * (centres ** 2).sum(1)).astype(np.float32)
    stats, _ = _estep(ctx, X, Wmat, c)
    want = np.zeros((K, 1 + 2 * D))
    for n in range(N):
        want[labels[n], 0] += 1
        want[labels[n], 1:1 + D] += X[n]
        want[labels[n], 1 + D:] += X[n].astype(np.float64) ** 2
    npt.assert_array_equal(stats, want)


def test_expected_params_and_natgrad_match_oracle(ctx):
    from bayesic_amd._ffi import ptr
    K, D = 37, 9
    rs = np.random.RandomState(2)
    X = rs.standard_normal((500, D)).astype(np.float32) * 2
    eta0 = svi.mog_prior_eta(K, D)
    eta = svi.mog_init_eta(X, K, D, seed=3) + rs.uniform(0, 5, K + 4 * K * D) * \
        np.concatenate([np.ones(K), np.zeros(K * D), np.ones(2 * K * D), np.zeros(K * D)])
    f64 = torch.float64
    etad = ctx.to_device(eta, f64)
    Wmat, c = ctx.zeros((K, 2 * D)), ctx.zeros(K)
    ctx.call("bsc_mog_expected_params", ptr(etad), K, D, ptr(Wmat), ptr(c))
    ctx.sync()
    Wr, cr = svi.mog_expected_params(eta, K, D)
    npt.assert_allclose(Wmat.cpu().numpy(), Wr, rtol=1e-6)
    npt.assert_allclose(c.cpu().numpy(), cr, rtol=1e-6, atol=1e-6)
    stats = rs.uniform(0, 10, (K, 1 + 2 * D))
    eta0d, statsd = ctx.to_device(eta0, f64), ctx.to_device(stats, f64)
    ctx.call("bsc_mog_natgrad", ptr(etad), ptr(eta0d), ptr(statsd), K, D, 3.5, 0.3)
    ctx.sync()
    want = svi.natgrad_update(eta, eta0, svi.mog_message(stats, K, D), 3.5, 0.3)
    npt.assert_allclose(etad.cpu().numpy(), want, rtol=1e-13)


def test_svi_steps_track_the_oracle_and_recover_clusters(ctx):
    from bayesic_amd.svi.mog import MoGNatGradSVI
    K, D = 8, 16
    X, centres, _ = svi.make_cfg3(40000, D, K)
    eta0 = svi.mog_prior_eta(K, D)
    eta = svi.mog_init_eta(X[:4000], K, D, seed=1)
    model = MoGNatGradSVI(X, K, eta0, eta, n_total=len(X), ctx=ctx)
    for t in range(1, 9):
        model.step()
        eta, stats, lse = svi.mog_svi_step(eta, eta0, X, len(X), (t + 1.0) ** -0.6, K, D)
        ctx.sync()
        npt.assert_allclose(model.lse.item(), lse, rtol=1e-5)
        npt.assert_allclose(model.eta.cpu().numpy(), eta, rtol=2e-4, atol=2e-3)
    _, m, _, _, _ = svi.mog_unpack(model.eta.cpu().numpy(), K, D)
    dist = np.linalg.norm(m[:, None, :] - centres[None], axis=2)
    assert (dist.min(axis=0) < 0.5).sum() >= K - 2       # (local optima may merge a pair)


def test_global_bound_kernel_matches_oracle(ctx):
    """bsc_mog_expected_params_bound: E_q[log p(theta)] - E_q[log q(theta)] of the Dirichlet and the K*D
    Normal-Gamma factors, and the same logit coefficients as bsc_mog_expected_params."""
    from bayesic_amd._ffi import ptr
    f64 = torch.float64
    for K, D, seed in ((37, 9, 2), (64, 16, 3), (1, 1, 4), (200, 40, 5)):
        rs = np.random.RandomState(seed)
        X = rs.standard_normal((max(500, 2 * K), D)).astype(np.float32) * 2
        eta0 = svi.mog_prior_eta(K, D, alpha0=0.7, m0=0.3, kappa0=0.05, a0=1.5, b0=0.8)
        eta = svi.mog_init_eta(X, K, D, seed=3) + rs.uniform(0, 5, K + 4 * K * D) * \
            np.concatenate([np.ones(K), np.zeros(K * D), np.ones(2 * K * D), np.zeros(K * D)])
        etad, eta0d = ctx.to_device(eta, f64), ctx.to_device(eta0, f64)
        Wmat, c, bound, prior_A = ctx.zeros((K, 2 * D)), ctx.zeros(K), ctx.zeros(1, f64), ctx.zeros(1, f64)
        ctx.call("bsc_mog_log_normalizer", ptr(eta0d), K, D, ptr(prior_A))
        ctx.sync()
        alpha0, _, kappa0, a0, b0 = svi.mog_unpack(eta0, K, D)
        npt.assert_allclose(prior_A.item(), float(svi.dirichlet_log_normalizer(alpha0)
                                                  + svi.normal_gamma_log_normalizer(kappa0, a0, b0).sum()), rtol=1e-12)
        ctx.call("bsc_mog_expected_params_bound", ptr(etad), ptr(eta0d), ptr(prior_A), K, D, ptr(Wmat), ptr(c),
                 ptr(bound))
        W2, c2 = ctx.zeros((K, 2 * D)), ctx.zeros(K)
        ctx.call("bsc_mog_expected_params", ptr(etad), K, D, ptr(W2), ptr(c2))
        ctx.sync()
        npt.assert_array_equal(Wmat.cpu().numpy(), W2.cpu().numpy())
        npt.assert_array_equal(c.cpu().numpy(), c2.cpu().numpy())
        Wr, cr = svi.mog_expected_params(eta, K, D)
        npt.assert_allclose(Wmat.cpu().numpy(), Wr, rtol=1e-6)
        npt.assert_allclose(c.cpu().numpy(), cr, rtol=1e-6, atol=1e-6)
        want = svi.mog_global_bound(eta, eta0, K, D)
        npt.assert_allclose(bound.item(), want, rtol=1e-11, atol=1e-9)
        # at the prior itself the bound vanishes (KL(p || p) = 0)
        ctx.call("bsc_mog_expected_params_bound", ptr(eta0d), ptr(eta0d), ptr(prior_A), K, D, ptr(Wmat), ptr(c),
                 ptr(bound))
        ctx.sync()
        assert abs(bound.item()) <= 1e-9 * K * D


@pytest.mark.parametrize("via", ["kernel", "executor"])
def test_elbo_tracks_the_oracle_and_rises_under_unit_steps(ctx, via):
    """model.elbo after step() = oracle.svi.mog_elbo at the parameters the step started from; with the full
    data set and rho = 1 (coordinate ascent, README.md:36) it can only rise."""
    from bayesic_amd.svi.mog import MoGNatGradSVI
    K, D = 8, 16
    X, _, _ = svi.make_cfg3(40000, D, K)
    eta0 = svi.mog_prior_eta(K, D)
    eta = svi.mog_init_eta(X[:4000], K, D, seed=1)
    model = MoGNatGradSVI(X, K, eta0, eta, n_total=3 * len(X), ctx=ctx, via=via)
    for t in range(1, 4):                       # damped mini-batch steps, scale = 3
        Wmat, c = svi.mog_expected_params(eta, K, D)
        _, lse = svi.mog_local_step(X, Wmat, c)
        want = svi.mog_elbo(eta, eta0, lse, 3.0, K, D)
        model.step()
        eta, _, _ = svi.mog_svi_step(eta, eta0, X, 3 * len(X), (t + 1.0) ** -0.6, K, D)
        ctx.sync()
        npt.assert_allclose(model.elbo.item(), want, rtol=2e-6)
        model.eta.copy_(torch.as_tensor(eta, dtype=torch.float64))
    full = MoGNatGradSVI(X, K, eta0, svi.mog_init_eta(X[:4000], K, D, seed=1), n_total=len(X), ctx=ctx, via=via)
    bounds = []
    for _ in range(8):
        full.step(rho=1.0)
        ctx.sync()
        bounds.append(full.elbo.item())
    bounds = np.array(bounds)
    assert np.all(np.diff(bounds) >= -1e-6 * np.abs(bounds[:-1])), bounds
    assert bounds[-1] > bounds[0]


def test_unsupported_sizes_fail_loudly(ctx):
    from bayesic_amd._ffi import BayesicHipError, ptr
    X, W, c = ctx.zeros((8, 17)), ctx.zeros((4, 34)), ctx.zeros(4)
    stats, lse = ctx.zeros(4 * 35, torch.float64), ctx.zeros(1, torch.float64)
    with pytest.raises(BayesicHipError, match="tile limits"):
        ctx.call("bsc_mog_estep", ptr(X), 17, 8, 17, 4, ptr(W), ptr(c), ptr(stats), ptr(lse))


@pytest.mark.parametrize("K,D", [(100, 24), (8, 16), (65, 4), (200, 40), (64, 24), (32, 20)])
def test_executor_route_beyond_one_tile_matches_oracle(ctx, K, D):
    """K > 64 or D > 16 (outside the fused kernel's MFMA tile): the same local step through the
    algebra executor -- GEMMs, one fused add, bsc_softmax_rows, three contractions -- against the
    oracle, and against the fused kernel where both apply."""
    from bayesic_amd.svi.mog import MoGNatGradSVI
    X, _, _ = svi.make_cfg3(20000, D, K)
    eta0 = svi.mog_prior_eta(K, D)
    eta = svi.mog_init_eta(X[:4000], K, D, seed=1)
    model = MoGNatGradSVI(X, K, eta0, eta, n_total=10 * len(X), ctx=ctx, via="executor")
    assert model.via == "executor"
    auto = MoGNatGradSVI(X, K, eta0, eta, n_total=10 * len(X), ctx=ctx)
    assert auto.via == ("kernel" if K <= 64 and D <= 16 else "executor")
    import torch
    for t in range(1, 4):
        # every update starts from the oracle's parameters: with K = 100 overlapping components a
        # float32-sized difference in a responsibility feeds back through the softmax
        for mdl in (model, auto):
            mdl.eta.copy_(torch.as_tensor(eta, dtype=torch.float64))
        model.step()
        auto.step()
        eta, stats, lse = svi.mog_svi_step(eta, eta0, X, 10 * len(X), (t + 1.0) ** -0.6, K, D)
        ctx.sync()
        npt.assert_allclose(model.lse.item(), lse, rtol=1e-5)
        scale = np.maximum(np.abs(eta), 1.0)
        assert (np.abs(model.eta.cpu().numpy() - eta) <= 2e-3 * scale).all()
        assert (np.abs(auto.eta.cpu().numpy() - eta) <= 2e-3 * scale).all()
