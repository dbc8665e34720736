"""bsc_softmax_rows: the expectation of a resident Categorical node (responsibilities) and the
bound's log-sum-exp, against float64 numpy."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(1, 1), (7, 3), (1000, 64), (4099, 17), (333, 2), (50, 65), (9, 1000),
                                       (20000, 16), (5, 1024), (0, 8),
                                       # the 16-bytes-per-lane kernel: every lane count, ragged row groups
                                       (100003, 64), (7, 8), (4099, 12), (333, 28), (1001, 100), (50, 256), (13, 132)])
def test_softmax_rows_matches_numpy(ctx, rows, cols):
    rs = np.random.RandomState(rows + cols)
    x = (rs.standard_normal((rows, cols)) * 6.0).astype(np.float32)
    if rows:
        x[0, 0] = 80.0          # a dominant logit and a very negative one
        x[-1, -1] = -90.0
    xd = ctx.to_device(x) if rows else ctx.zeros((1, cols))
    out, lse = ctx.zeros((max(rows, 1), cols)), ctx.zeros(max(rows, 1))
    ctx.call("bsc_softmax_rows", xd, rows, cols, cols, out, cols, lse)
    ctx.sync()
    if rows == 0:
        return
    x64 = x.astype(np.float64)
    m = x64.max(1, keepdims=True)
    w = np.exp(x64 - m)
    # exp2 of a float32 product x * log2(e): relative error ~ |x - max| * 2^-24
    npt.assert_allclose(out.cpu().numpy(), w / w.sum(1, keepdims=True), rtol=3e-5, atol=1e-30)
    npt.assert_allclose(lse.cpu().numpy(), (m + np.log(w.sum(1, keepdims=True)))[:, 0], rtol=1e-6, atol=1e-6)
    npt.assert_allclose(out.cpu().numpy().sum(1), 1.0, rtol=1e-6)


def test_softmax_rows_strided_and_limits(ctx):
    from bayesic_amd._ffi import BayesicHipError
    x = torch.randn((100, 40), device=ctx.device)
    out = torch.zeros((100, 48), device=ctx.device)
    ctx.call("bsc_softmax_rows", x[:, :20], 100, 20, 40, out, 48, 0)        # lse = NULL, padded rows
    ctx.sync()
    npt.assert_allclose(out[:, :20].cpu().numpy(), torch.softmax(x[:, :20].double(), 1).cpu().numpy(), rtol=3e-5)
    assert float(out[:, 20:].abs().max()) == 0.0
    with pytest.raises(BayesicHipError, match="1024"):
        ctx.call("bsc_softmax_rows", x, 1, 2000, 2000, out, 2000, 0)


# ---- bsc_gemm_softmax_rows: the softmax taken inside the tall-skinny product ---------------------
@pytest.mark.parametrize("rows,K,N,lda,transposed_b", [
    (1, 8, 4, 8, False), (33, 16, 64, 16, False), (4099, 40, 64, 40, True), (1000, 64, 60, 68, False),
    (70_001, 40, 64, 40, False), (257, 24, 12, 24, True),
])
def test_gemm_softmax_rows_matches_numpy(ctx, rows, K, N, lda, transposed_b):
    """R = softmax_rows(A . B), lse and cross = sum_c R * (A . B) in one pass; float32 products and
    sums against float64 numpy: rtol 2e-5 on the responsibilities' scale, 1e-5 * |logit| on lse / cross."""
    import torch
    rs = np.random.RandomState(rows + K + N)
    A_ = np.zeros((rows, lda), np.float32)
    A_[:, :K] = rs.standard_normal((rows, K))
    B_ = (rs.standard_normal((K, N)) * 1.5).astype(np.float32)
    Ad = ctx.to_device(A_)
    Bd = ctx.to_device(np.ascontiguousarray(B_.T)) if transposed_b else ctx.to_device(B_)
    ldbk, ldbn = (1, K) if transposed_b else (N, 1)
    R = ctx.zeros((rows, N), torch.float32)
    lse = ctx.zeros(rows, torch.float32)
    cross = ctx.zeros(rows, torch.float32)
    alpha = 1.0 if rows % 2 else 0.375           # (the multiplier of a local latent under mini-batch scaling)
    ctx.call("bsc_gemm_softmax_rows", Ad, lda, rows, K, Bd, ldbk, ldbn, N, alpha, R, N, lse, cross)
    ctx.sync()
    logits = alpha * (A_[:, :K].astype(np.float64) @ B_.astype(np.float64))
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    want = e / e.sum(axis=1, keepdims=True)
    scale = np.abs(logits).max() + 1.0
    np.testing.assert_allclose(R.cpu().numpy(), want, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(lse.cpu().numpy(), (m + np.log(e.sum(axis=1, keepdims=True)))[:, 0],
                               rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(cross.cpu().numpy(), (want * logits).sum(axis=1), rtol=0, atol=2e-5 * scale)
    # rows sum to one; a null `cross` is allowed
    np.testing.assert_allclose(R.cpu().numpy().sum(axis=1), 1.0, rtol=1e-5)
    ctx.call("bsc_gemm_softmax_rows", Ad, lda, rows, K, Bd, ldbk, ldbn, N, alpha, R, N, lse, None)
    ctx.sync()


def test_gemm_softmax_rows_envelope(ctx):
    import torch
    from bayesic_amd._ffi import BayesicHipError
    A = ctx.zeros((8, 72), torch.float32)
    B = ctx.zeros((72, 64), torch.float32)
    R = ctx.zeros((8, 64), torch.float32)
    lse = ctx.zeros(8, torch.float32)
    for K, N in ((72, 64), (12, 64), (16, 6)):
        with pytest.raises(BayesicHipError, match="bsc_gemm_softmax_rows"):
            ctx.call("bsc_gemm_softmax_rows", A, 72, 8, K, B, 64, 1, N, 1.0, R, 64, lse, None)
    ctx.call("bsc_gemm_softmax_rows", A, 72, 0, 16, B, 64, 1, 64, 1.0, R, 64, lse, None)     # no rows: nothing to do
