"""bsc_gemm_softmax_stats: the responsibility-weighted statistics R^T . A taken in the pass that takes the softmax
(VERDICT r2 #4 -- the backward half of the mixture's local step inside gemm_softmax_rows_kernel), and its use
by the derived mean-field engine: a resident Categorical node's responsibilities are NOT written when its
neighbours only ask for statistics against the features the logits were formed from."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(A, B, alpha):
    L = alpha * (A.astype(np.float64) @ B.astype(np.float64))
    m = L.max(axis=1, keepdims=True)
    e = np.exp(L - m)
    R = e / e.sum(axis=1, keepdims=True)
    lse = (m[:, 0] + np.log(e.sum(axis=1)))
    return R, R.T @ A.astype(np.float64), lse.sum()


@pytest.mark.parametrize("rows,K,N", [(1, 8, 4), (31, 8, 64), (33, 16, 12), (1000, 40, 64), (4097, 64, 64),
                                      (70001, 40, 64), (50000, 24, 36), (12345, 56, 8), (0, 16, 8)])
@pytest.mark.parametrize("write_r", [False, True])
def test_statistics_in_the_softmax_pass_match_float64(ctx, rows, K, N, write_r):
    rs = np.random.RandomState(rows + K + N)
    lda = K + 4
    Ah = np.zeros((max(rows, 1), lda), np.float32)
    Ah[:, :K] = rs.standard_normal((max(rows, 1), K)) * 0.7
    Bh = (rs.standard_normal((K, N)) * 0.8).astype(np.float32)
    alpha = 0.9
    A = ctx.to_device(Ah)[:rows]
    B = ctx.to_device(Bh)
    stats = torch.full((N, K + 3), float("nan"), dtype=torch.float32, device=ctx.device)
    lse = torch.full((1,), float("nan"), dtype=torch.float64, device=ctx.device)
    R = torch.full((max(rows, 1), N), float("nan"), dtype=torch.float32, device=ctx.device) if write_r else None
    ctx.call("bsc_gemm_softmax_stats", A if rows else ctx.to_device(Ah), lda, rows, K, B, N, 1, N, alpha, None, R, N,
             stats, K + 3, lse)
    ctx.sync()
    got = stats.cpu().numpy()
    assert np.isnan(got[:, K:]).all()                                  # padding untouched
    if rows == 0:
        assert (got[:, :K] == 0).all() and lse.item() == 0.0
        return
    Rr, Sr, lr = _reference(Ah[:rows, :K], Bh, alpha)
    bound = np.abs(Rr).T @ np.abs(Ah[:rows, :K].astype(np.float64))   # sums of |terms|
    assert (np.abs(got[:, :K] - Sr) <= 3e-6 * bound + 1e-6).all(), np.abs(got[:, :K] - Sr).max()
    npt.assert_allclose(lse.item(), lr, rtol=2e-6)
    if write_r:
        npt.assert_allclose(R.cpu().numpy()[:rows], Rr, rtol=2e-5, atol=2e-7)
    again = torch.empty_like(stats)
    ctx.call("bsc_gemm_softmax_stats", A, lda, rows, K, B, N, 1, N, alpha, None, R, N, again, K + 3, lse)
    ctx.sync()
    npt.assert_array_equal(again.cpu().numpy()[:, :K], got[:, :K])   # fixed order: run-to-run identical


@pytest.mark.parametrize("rows,K,N", [(1000, 32, 64), (70001, 32, 64), (33, 8, 12), (50000, 56, 36), (4097, 16, 4)])
@pytest.mark.parametrize("write_r", [False, True])
def test_bias_row_and_column_sums(ctx, rows, K, N, write_r):
    """The ones column out of the product: L = alpha (A . B + bias), stats[:, K] = column sums of R."""
    rs = np.random.RandomState(rows + K + N + 7)
    lda = K + 8
    Ah = np.zeros((rows, lda), np.float32)
    Ah[:, :K] = rs.standard_normal((rows, K)) * 0.7
    Ah[:, K] = 1.0                                              # (what the wide operand holds there; not read)
    Bh = (rs.standard_normal((K + 1, N)) * 0.8).astype(np.float32)
    alpha = 1.1
    A, B = ctx.to_device(Ah), ctx.to_device(Bh)
    stats = torch.full((N, K + 2), float("nan"), dtype=torch.float32, device=ctx.device)
    lse = torch.full((1,), float("nan"), dtype=torch.float64, device=ctx.device)
    R = torch.full((rows, N), float("nan"), dtype=torch.float32, device=ctx.device) if write_r else None
    ctx.call("bsc_gemm_softmax_stats", A, lda, rows, K, B, N, 1, N, alpha, B[K], R, N, stats, K + 2, lse)
    ctx.sync()
    Rr, Sr, lr = _reference(Ah[:, :K + 1], Bh, alpha)          # the same logits with the ones column IN the product
    got = stats.cpu().numpy()
    assert np.isnan(got[:, K + 1]).all()
    bound = np.abs(Rr).T @ np.abs(Ah[:, :K + 1].astype(np.float64))
    assert (np.abs(got[:, :K + 1] - Sr) <= 3e-6 * bound + 1e-6).all(), np.abs(got[:, :K + 1] - Sr).max()
    npt.assert_allclose(lse.item(), lr, rtol=2e-6)
    if write_r:
        npt.assert_allclose(R.cpu().numpy(), Rr, rtol=2e-5, atol=2e-7)


def test_limits_fail_loudly(ctx):
    from bayesic_amd._ffi import BayesicHipError
    A, B = ctx.zeros((64, 72)), ctx.zeros((72, 8))
    stats, lse = ctx.zeros((8, 72)), ctx.zeros(1, torch.float64)
    with pytest.raises(BayesicHipError, match="K=72"):
        ctx.call("bsc_gemm_softmax_stats", A, 72, 64, 72, B, 8, 1, 8, 1.0, None, None, 8, stats, 72, lse)
    with pytest.raises(BayesicHipError, match="bias row"):
        ctx.call("bsc_gemm_softmax_stats", A, 72, 64, 64, B, 8, 1, 8, 1.0, B, None, 8, stats, 72, lse)


def test_derived_mixture_does_not_write_its_responsibilities(ctx):
    """The derived engine's update of config 3's model with deferred responsibilities: the same parameters as
    with the responsibilities written (and as the oracle), one data-sized launch per update, and no [rows, K]
    tensor on the device."""
    from bayesic_amd.algebra.device_backend import DeferredSoftmax, DeviceBackend
    from bayesic_amd.inference.mixture import DiagonalMixtureVMP
    from oracle import svi
    n, d, k = 100_000, 16, 64
    X, _, _ = svi.make_cfg3(n, d, k)
    eta0 = svi.mog_prior_eta(k, d)
    eta = svi.mog_init_eta(X[:500], k, d, seed=2)
    alpha, m, kappa, a, b = svi.mog_unpack(eta, k, d)
    models = []
    for defer in (True, False):
        model = DiagonalMixtureVMP(X, k, n_total=10.0 * n, init=(alpha, m, kappa, a, b), backend=DeviceBackend(ctx),
                                   route="derived")
        model.vmp.defer_responsibilities = defer
        models.append(model)
    calls = []
    real_call = ctx.call
    ctx.call = lambda name, *a: (calls.append(name), real_call(name, *a))[1]
    try:
        for t in range(1, 3):
            rho = (t + 1.0) ** -0.6
            calls.clear()
            models[0].step(rho)
            deferred_calls = list(calls)
            models[1].step(rho)
            eta, _, _ = svi.mog_svi_step(eta, eta0, X, 10.0 * n, rho, k, d)
    finally:
        ctx.call = real_call
    assert isinstance(models[0].z.expectations_backend()[0], DeferredSoftmax)
    assert deferred_calls.count("bsc_gemm_softmax_stats") == 1
    assert "bsc_gemm_softmax_rows" not in deferred_calls and "bsc_gemm_strided_batched" not in deferred_calls
    got, ref = models[0].eta_fused_layout(), models[1].eta_fused_layout()
    scale = np.maximum(np.abs(eta), 1.0)
    assert (np.abs(got - eta) <= 1e-3 * scale).all(), np.abs((got - eta) / scale).max()
    assert (np.abs(got - ref) <= 1e-3 * scale).all()        # (float32 statistics summed along different routes)
    # the bound is the same number whichever way the entropy of the assignments is obtained
    npt.assert_allclose(models[0].vmp.elbo(), models[1].vmp.elbo(), rtol=2e-6)
    # asking for the responsibilities themselves still works: they are written then
    r = models[0].z.expectations()[0]
    assert r.shape == (n, k)
    npt.assert_allclose(r.sum(axis=1), 1.0, rtol=1e-5)
