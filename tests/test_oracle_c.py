"""The plain-C second oracle (oracle/c/oracle_kernels.c) against the numpy oracle on small
inputs (CPU), and the HIP kernels against the C oracle at BASELINE's FULL sizes (GPU) -- the
numpy oracle takes minutes there, the OpenMP C one seconds."""
import math
import shutil

import numpy as np
import numpy.testing as npt
import pytest

from oracle import svi

needs_gcc = pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
# the MFMA-bound configs also on the operand-split bf16 route (bsc_ctx_set_mfma_split 2): the same
# comparisons at the same tolerances
both_routes = pytest.mark.parametrize("terms", [0, 2])


class _split(object):
    def __init__(self, ctx, terms):
        self.ctx, self.terms = ctx, terms

    def __enter__(self):
        self.ctx.call("bsc_ctx_set_mfma_split", self.terms)

    def __exit__(self, *exc):
        self.ctx.call("bsc_ctx_set_mfma_split", 0)
rs = np.random.RandomState(8)


@needs_gcc
def test_c_oracle_matches_numpy_oracle():
    from oracle import cbuild
    X = rs.standard_normal((777, 24)).astype(np.float32)
    y = rs.standard_normal(777).astype(np.float32)
    W = (rs.standard_normal((5, 24)) / 4).astype(np.float32)
    Q, G = cbuild.blr_data_pass(X, y, W)
    q_ref, g_ref = svi.blr_data_pass(X, y, W)
    npt.assert_allclose(Q, q_ref, rtol=1e-12)
    npt.assert_allclose(G, g_ref, rtol=1e-11, atol=1e-11)

    g = rs.randint(7, size=777).astype(np.int32)
    yb = (rs.uniform(size=777) < 0.4).astype(np.float32)
    Wz = (rs.standard_normal((16, 24)) / 4).astype(np.float32)
    Bz = rs.standard_normal((7, 16)).astype(np.float32)
    npt.assert_allclose(cbuild.logreg_loglik(X, yb, g, Wz, Bz), svi.logreg_loglik(X, yb, g, Wz, Bz),
                        rtol=1e-12)

    K, D = 6, 24
    T = rs.uniform(0.5, 2.0, (K, D))
    cen = rs.standard_normal((K, D))
    Wmat = np.concatenate([T * cen, -0.5 * T], axis=1).astype(np.float32)
    c = rs.standard_normal(K).astype(np.float32)
    stats, lse = cbuild.mog_estep(X, Wmat, c)
    s_ref, l_ref = svi.mog_local_step(X, Wmat, c)
    npt.assert_allclose(stats, s_ref, rtol=1e-10, atol=1e-10)
    npt.assert_allclose(lse, l_ref, rtol=1e-12)

    C = rs.poisson(0.3, (90, 40)).astype(np.float32)
    Th = rs.uniform(0.1, 1.0, (90, 32)).astype(np.float32)
    Bt = rs.uniform(0.1, 1.0, (32, 40)).astype(np.float32)
    npt.assert_allclose(cbuild.lda_sstats(C, Th, Bt), svi.lda_sstats(C, Th, Bt), rtol=1e-12)

    npt.assert_allclose(cbuild.lda_local_bound(C, Th, Bt), svi.lda_local_bound(C, Th, Bt), rtol=1e-12)

    Rr = rs.dirichlet(np.ones(6), 777).astype(np.float32)
    Yy = rs.standard_normal((777, 5)).astype(np.float32)
    npt.assert_allclose(cbuild.weighted_outer(Rr, X[:, :7], Yy),
                        np.einsum("nk,nd,ne->kde", Rr.astype(np.float64), X[:, :7].astype(np.float64),
                                  Yy.astype(np.float64)), rtol=1e-11, atol=1e-11)


@needs_gcc
@pytest.mark.gpu
def test_cfg2_full_size_against_c_oracle(ctx):
    """1M x 256, S = 8: the data pass against the C oracle on the very same data."""
    import torch
    from oracle import cbuild
    g = torch.Generator(device=ctx.device).manual_seed(21)
    B, D, S = 1_000_000, 256, 8
    X = torch.randn((B, D), generator=g, device=ctx.device)
    y = torch.randn(B, generator=g, device=ctx.device)
    W = torch.randn((S, D), generator=g, device=ctx.device) / 16
    Q, G = ctx.zeros(S, torch.float64), ctx.zeros((S, D), torch.float64)
    ctx.call("bsc_blr_data_pass", X, D, y, B, D, W, S, Q, G)
    ctx.sync()
    Xh, yh, Wh = X.cpu().numpy(), y.cpu().numpy(), W.cpu().numpy()
    q_ref, g_ref = cbuild.blr_data_pass(Xh, yh, Wh)
    npt.assert_allclose(Q.cpu().numpy(), q_ref, rtol=2e-6)
    # each G[s, d] is a sum of 1e6 terms of either sign: bound by the scale of the sum
    scale = np.sqrt(q_ref)[:, None] * np.sqrt((Xh.astype(np.float64) ** 2).sum(0))[None, :]
    assert (np.abs(G.cpu().numpy() - g_ref) <= 2e-5 * scale).all()


@both_routes
@needs_gcc
@pytest.mark.gpu
def test_cfg5_full_size_against_c_oracle(ctx, terms):
    """1M x 256, G = 1000, S = 64."""
    import torch
    from oracle import cbuild
    g = torch.Generator(device=ctx.device).manual_seed(22)
    N, D, G, S = 1_000_000, 256, 1000, 64
    X = torch.randn((N, D), generator=g, device=ctx.device)
    y = (torch.rand(N, generator=g, device=ctx.device) < 0.4).float()
    grp = torch.randint(G, (N,), generator=g, device=ctx.device).to(torch.int32)
    Wz = torch.randn((S, D), generator=g, device=ctx.device) / 16
    Bz = torch.randn((G, S), generator=g, device=ctx.device)
    ell = ctx.zeros(S, torch.float64)
    with _split(ctx, terms):
        ctx.call("bsc_logreg_bbvi_loglik", X, D, y, grp, N, D, G, Wz, Bz, S, ell)
        ctx.sync()
    want = cbuild.logreg_loglik(X.cpu().numpy(), y.cpu().numpy(), grp.cpu().numpy(),
                                Wz.cpu().numpy(), Bz.cpu().numpy())
    npt.assert_allclose(ell.cpu().numpy(), want, rtol=2e-6)


@both_routes
@needs_gcc
@pytest.mark.gpu
def test_cfg3_full_size_against_c_oracle(ctx, terms):
    """10M x 16, K = 64."""
    import torch
    from oracle import cbuild
    g = torch.Generator(device=ctx.device).manual_seed(23)
    N, D, K = 10_000_000, 16, 64
    cen = torch.randn((K, D), generator=g, device=ctx.device) * 2
    X = cen[torch.randint(K, (N,), generator=g, device=ctx.device)] + \
        torch.randn((N, D), generator=g, device=ctx.device)
    T = torch.rand((K, D), generator=g, device=ctx.device) + 0.5
    Wmat = torch.cat([T * cen, -0.5 * T], dim=1).contiguous()
    c = (-0.5 * (T * cen ** 2).sum(1)).contiguous()
    stats = ctx.zeros((K, 1 + 2 * D), torch.float64)
    lse = ctx.zeros(1, torch.float64)
    with _split(ctx, terms):
        ctx.call("bsc_mog_estep", X, D, N, D, K, Wmat, c, stats, lse)
        ctx.sync()
    Xh = X.cpu().numpy()
    s_ref, l_ref = cbuild.mog_estep(Xh, Wmat.cpu().numpy(), c.cpu().numpy())
    X64 = np.abs(Xh.astype(np.float64))
    scale = np.concatenate([[float(N)], X64.sum(0), (X64 ** 2).sum(0)])
    assert (np.abs(stats.cpu().numpy() - s_ref) <= 2e-5 * scale[None, :] / math.sqrt(K)).all()
    npt.assert_allclose(lse.item(), l_ref, rtol=2e-6)


@both_routes
@needs_gcc
@pytest.mark.gpu
def test_cfg4_full_size_against_c_oracle(ctx, terms):
    """6250 x 100 000 counts (one GPU's shard of config 4), K = 128: the dense MFMA kernel and
    the sparse (CSC) kernel against the C oracle."""
    import scipy.sparse as sp
    import torch
    from oracle import cbuild
    docs, V, K = 6250, 100_000, 128
    g = torch.Generator(device=ctx.device).manual_seed(24)
    C = torch.poisson(torch.full((docs, V), 0.05, device=ctx.device), generator=g)
    Th = torch.rand((docs, K), generator=g, device=ctx.device) + 0.1
    Bt = torch.rand((K, V), generator=g, device=ctx.device) + 0.1
    dense = ctx.zeros((K, V), torch.float32)
    with _split(ctx, terms):
        ctx.call("bsc_lda_sstats", C, V, docs, V, K, Th, K, Bt, V, dense, V)
        ctx.sync()
    Ch = C.cpu().numpy()
    want = cbuild.lda_sstats(Ch, Th.cpu().numpy(), Bt.cpu().numpy())
    # a sum of ~300 positive float32 terms per output
    npt.assert_allclose(dense.cpu().numpy(), want, rtol=2e-5)
    csc = sp.csc_matrix(Ch)
    colptr = torch.from_numpy(csc.indptr.astype(np.int64)).to(ctx.device)
    rowidx = torch.from_numpy(csc.indices.astype(np.int32)).to(ctx.device)
    vals = torch.from_numpy(csc.data.astype(np.float32)).to(ctx.device)
    sparse = ctx.zeros((K, V), torch.float32)
    ctx.call("bsc_lda_sstats_csc", colptr, rowidx, vals, docs, V, K, Th, K, Bt, V, sparse, V)
    ctx.sync()
    npt.assert_allclose(sparse.cpu().numpy(), want, rtol=2e-5)


@needs_gcc
@pytest.mark.gpu
def test_weighted_second_moment_full_size_against_c_oracle(ctx):
    """10M x 16, K = 64: sum_n r_nk x_n x_n^T in one pass (bsc_weighted_outer)."""
    import torch
    from oracle import cbuild
    g = torch.Generator(device=ctx.device).manual_seed(25)
    N, D, K = 10_000_000, 16, 64
    X = torch.randn((N, D), generator=g, device=ctx.device)
    R = torch.softmax(2 * torch.randn((N, K), generator=g, device=ctx.device), dim=1)
    out = ctx.zeros((K, D, D), torch.float32)
    ctx.call("bsc_weighted_outer", R, K, X, D, X, D, N, K, D, D, 1.0, out)
    ctx.sync()
    Xh, Rh = X.cpu().numpy(), R.cpu().numpy()
    want = cbuild.weighted_outer(Rh, Xh, Xh)
    # |sum| is bounded by sum_n r |x_d||x_e|; the diagonal entries are sums of positive terms
    bound = cbuild.weighted_outer(Rh, np.abs(Xh), np.abs(Xh))
    assert (np.abs(out.cpu().numpy() - want) <= 2e-5 * bound).all()


@both_routes
@needs_gcc
@pytest.mark.gpu
def test_cfg3_full_size_elbo_against_c_oracle(ctx, terms):
    """10M x 16, K = 64: model.elbo of the fused driver = oracle.svi.mog_elbo with the local term from the C oracle's
    pass over the very same data (the numpy oracle takes minutes at this size)."""
    import torch
    from bayesic_amd.svi.mog import MoGNatGradSVI
    from oracle import cbuild
    N, D, K = 10_000_000, 16, 64
    X, _, _ = svi.make_cfg3(N, D, K)
    eta0 = svi.mog_prior_eta(K, D)
    eta = svi.mog_init_eta(X[:4000], K, D, seed=1)
    model = MoGNatGradSVI(X, K, eta0, eta, n_total=4.0 * N, ctx=ctx)
    for t in range(1, 3):
        Wmat, c = svi.mog_expected_params(eta, K, D)
        stats, lse = cbuild.mog_estep(X, Wmat, c)
        want = svi.mog_elbo(eta, eta0, lse, 4.0, K, D)
        rho = (t + 1.0) ** -0.6
        with _split(ctx, terms):
            model.step(rho)
            ctx.sync()
        npt.assert_allclose(model.elbo.item(), want, rtol=2e-6)
        eta = svi.natgrad_update(eta, eta0, svi.mog_message(stats, K, D), 4.0, rho)
        model.eta.copy_(torch.as_tensor(eta, dtype=torch.float64))


@both_routes
@needs_gcc
@pytest.mark.gpu
def test_cfg4_full_size_elbo_against_c_oracle(ctx, terms):
    """6250 x 100 000 counts, K = 128 (one GPU's shard of config 4): the words' term taken inside the statistic kernel
    (dense persistent kernel and sparse kernel) against the C oracle; the topics' and documents' terms against the
    float64 numpy oracle (parameter-sized); model.elbo of the driver against their combination."""
    import scipy.sparse as sp
    import torch
    from bayesic_amd.svi.lda import LDAFixedGammaSVI
    from oracle import cbuild
    docs, V, K = 6250, 100_000, 128
    g = torch.Generator(device=ctx.device).manual_seed(24)
    C = torch.poisson(torch.full((docs, V), 0.05, device=ctx.device), generator=g)
    gamma = torch.rand((docs, K), generator=g, device=ctx.device) + 0.5
    lam = torch.rand((K, V), generator=g, device=ctx.device) + 0.5
    eta_prior, alpha, docs_total = 0.01, 1.0 / K, 50_000.0
    Ch, gh, lh = C.cpu().numpy(), gamma.cpu().numpy(), lam.cpu().numpy()
    Th = svi.dirichlet_expectation(gh).astype(np.float32)
    Bt = svi.dirichlet_expectation(lh).astype(np.float32)
    words = cbuild.lda_local_bound(Ch, Th, Bt)
    want = docs_total / docs * (words + float(svi.dirichlet_neg_kl(gh, alpha).sum())) \
        + float(svi.dirichlet_neg_kl(lh, eta_prior).sum())
    for counts in (C, sp.csr_matrix(Ch)):
        model = LDAFixedGammaSVI(counts, gamma, lam, eta=eta_prior, docs_total=docs_total, ctx=ctx, alpha=alpha)
        with _split(ctx, terms):
            model.step()
            ctx.sync()
        # (the device forms Th, Bt in float32 from float64 digammas exactly as the oracle does; the words' term sums
        # 31 M float32 products)
        npt.assert_allclose(model._ll.item(), words, rtol=3e-6)
        npt.assert_allclose(model.elbo.item(), want, rtol=3e-6)


def test_fused_f32_host_leg_agrees_with_the_numpy_oracle():
    """oracle_blr_data_pass_f32 (bench.py's cpu_baseline.fused_f32: one fused float32 OpenMP pass, AVX2 vectors, float64
    across threads) restates the same formula (README.md:51): held to the device pass's tolerance against oracle.svi."""
    from oracle import cbuild, svi
    for B, D, S in ((4099, 256, 8), (1003, 252, 5), (777, 36, 11), (5, 8, 1), (130_003, 256, 8)):
        rs = np.random.RandomState(B + D + S)
        X = rs.standard_normal((B, D)).astype(np.float32)
        y = rs.standard_normal(B).astype(np.float32)
        W = (rs.standard_normal((S, D)) / np.sqrt(D)).astype(np.float32)
        Q, G = cbuild.blr_data_pass_f32(X, y, W)
        Qr, Gr = svi.blr_data_pass_chunked(X, y, W)
        np.testing.assert_allclose(Q, Qr, rtol=2e-5)
        bound = np.sqrt(Qr)[:, None] * np.sqrt((X.astype(np.float64) ** 2).sum(axis=0))[None, :]
        assert (np.abs(G - Gr) <= 2e-5 * bound + 1e-12).all()
        Xc = cbuild.first_touch_copy(X)
        assert Xc is not X and np.array_equal(Xc, X)
    with pytest.raises(ValueError, match="envelope"):
        cbuild.blr_data_pass_f32(np.zeros((4, 8), np.float32), np.zeros(4, np.float32), np.zeros((17, 8), np.float32))
