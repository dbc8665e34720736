"""The operand-split bf16 route of the MFMA-bound contractions (bsc_ctx_set_mfma_split; csrc/bsc_bf16split.h):
off by default, and when on it has to pass the SAME comparisons with the float64 oracle, at the SAME tolerances,
as the f32 route (tests/test_lda_gpu.py) -- two terms as well as three."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

from oracle import svi

pytestmark = pytest.mark.gpu


@pytest.fixture
def split_ctx(ctx):
    def set_terms(n):
        ctx.call("bsc_ctx_set_mfma_split", n)
    yield set_terms
    ctx.call("bsc_ctx_set_mfma_split", 0)


def _rel(got, want):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)))


@pytest.mark.parametrize("terms", [2, 3])
@pytest.mark.parametrize("docs,V", [
    (32, 128),             # one full tile
    (1, 4),                # a single document
    (45, 132),             # ragged documents and vocabulary
    (257, 1300),
    (5000, 3000),          # column blocks split along the documents, partial statistics added by the fix-up
    (70, 70000),           # whole rounds of column blocks, then a split tail
])
def test_lda_statistics_and_bound_on_split_operands(ctx, split_ctx, terms, docs, V):
    K = 128
    rs = np.random.RandomState(docs + V + terms)
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    Th = (rs.rand(docs, K) + 0.05).astype(np.float32)
    Bt = (rs.rand(K, V) + 0.05).astype(np.float32)
    dC, dTh, dBt = ctx.to_device(C), ctx.to_device(Th), ctx.to_device(Bt)
    want = svi.lda_sstats(C, Th, Bt)
    want_ll = svi.lda_local_bound(C, Th, Bt)
    f32 = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, f32, V)
    split_ctx(terms)
    out = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, out, V)
    out_b = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
    ll = ctx.zeros(1, torch.float64)
    ctx.call("bsc_lda_sstats_bound", dC, V, docs, V, K, dTh, K, dBt, V, out_b, V, ll)
    ctx.sync()
    got = out.cpu().numpy()
    npt.assert_allclose(got, want, rtol=3e-5, atol=1e-6)               # tests/test_lda_gpu.py's tolerance
    npt.assert_array_equal(got, out_b.cpu().numpy())
    scale = float((C.astype(np.float64) * np.abs(np.log(Th.astype(np.float64) @ Bt.astype(np.float64)))).sum())
    assert abs(ll.item() - want_ll) <= 3e-6 * scale + 1e-9
    assert not np.array_equal(got, f32.cpu().numpy()) or docs * V < 1000     # it IS another arithmetic
    # three terms: the f32 class -- no further from the float64 oracle than twice the f32 route
    if terms == 3:
        assert _rel(got, want) <= 2.0 * _rel(f32.cpu().numpy(), want) + 1e-7
    # run-to-run identical
    out2 = torch.empty_like(out)
    ctx.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, out2, V)
    ctx.sync()
    npt.assert_array_equal(got, out2.cpu().numpy())


def test_split_is_off_by_default_and_rejects_other_term_counts(ctx):
    from bayesic_amd._ffi import BayesicHipError
    with pytest.raises(BayesicHipError):
        ctx.call("bsc_ctx_set_mfma_split", 4)
    with pytest.raises(BayesicHipError):
        ctx.call("bsc_ctx_set_mfma_split", 1)
    ctx.call("bsc_ctx_set_mfma_split", 0)


def _loglik(ctx, X, y, g, Wz, Bz):
    D, G, S = X.shape[1], Bz.shape[0], Wz.shape[0]
    Xd, yd = ctx.to_device(X), ctx.to_device(y)
    gd = torch.as_tensor(g, dtype=torch.int32).to(ctx.device)
    ell = ctx.zeros(S, torch.float64)
    ctx.call("bsc_logreg_bbvi_loglik", Xd, D, yd, gd, X.shape[0], D, G, ctx.to_device(Wz), ctx.to_device(Bz), S, ell)
    ctx.sync()
    return ell.cpu().numpy()


@pytest.mark.parametrize("N,D,G,S", [(32, 256, 7, 64), (1003, 256, 1000, 64), (5, 256, 3, 64), (4099, 64, 11, 64),
                                     (20000, 252, 100, 64), (3000, 256, 9, 48), (70001, 256, 33, 64)])
def test_logreg_loglik_on_split_operands(ctx, split_ctx, N, D, G, S):
    """Config 5's data-sized contraction with X and the draws as two bf16 terms: tests/test_bbvi_gpu.py's comparison
    and tolerance."""
    import math
    rs = np.random.RandomState(N + D + G)
    X = rs.standard_normal((N, D)).astype(np.float32)
    g = rs.randint(G, size=N).astype(np.int32)
    y = (rs.uniform(size=N) < 0.4).astype(np.float32)
    Wz = (rs.standard_normal((S, D)) / math.sqrt(D)).astype(np.float32)
    Bz = rs.standard_normal((G, S)).astype(np.float32)
    f32 = _loglik(ctx, X, y, g, Wz, Bz)
    split_ctx(2)
    ell = _loglik(ctx, X, y, g, Wz, Bz)
    want = svi.logreg_loglik(X, y, g, Wz, Bz)
    L = np.abs(X.astype(np.float64) @ Wz.astype(np.float64).T + Bz.astype(np.float64)[g])
    bound = (L + 1.0).sum(axis=0)
    assert (np.abs(ell - want) <= 2e-5 * bound + 1e-9).all(), np.abs((ell - want) / bound).max()
    assert not np.array_equal(ell, f32)
    npt.assert_array_equal(ell, _loglik(ctx, X, y, g, Wz, Bz))        # run-to-run identical
    split_ctx(3)                                                       # three terms: not offered here -> the f32 route
    npt.assert_array_equal(_loglik(ctx, X, y, g, Wz, Bz), f32)


def test_logreg_loglik_split_operand_layout_with_exact_integers(ctx, split_ctx):
    """tests/test_bbvi_gpu.py's layout test on the split route: integers are exact in one bf16 term, so any mix-up of
    k order, sample or row mapping changes which rows count."""
    N, D, G = 96, 256, 4
    X = np.zeros((N, D), np.float32)
    X[np.arange(N), (np.arange(N) * 7) % D] = 1.0
    Wz = np.zeros((64, D), np.float32)
    for s in range(64):
        Wz[s, (s * 5) % D] = 40.0
    Bz = np.full((G, 64), -20.0, np.float32)
    g = (np.arange(N) % G).astype(np.int32)
    y = (np.arange(N) % 3 == 0).astype(np.float32)
    split_ctx(2)
    ell = _loglik(ctx, X, y, g, Wz, Bz)
    want = svi.logreg_loglik(X, y, g, Wz, Bz)
    npt.assert_allclose(ell, want, rtol=1e-6, atol=1e-6)


def _estep(ctx, X, Wmat, c):
    K, twoD = Wmat.shape
    D = twoD // 2
    Xd = ctx.to_device(X) if X.shape[0] else ctx.zeros((1, D))
    Wd, cd = ctx.to_device(Wmat), ctx.to_device(c)
    stats, lse = ctx.zeros((K, 1 + 2 * D), torch.float64), ctx.zeros(1, torch.float64)
    ctx.call("bsc_mog_estep", Xd, D, X.shape[0], D, K, Wd, cd, stats, lse)
    ctx.sync()
    return stats.cpu().numpy(), lse.item()


@pytest.mark.parametrize("terms", [2, 3])
@pytest.mark.parametrize("N,D,K", [(32, 16, 64), (1003, 16, 64), (7, 16, 64), (5000, 16, 33), (4097, 5, 64),
                                   (333, 1, 2), (20000, 12, 40), (300000, 16, 64)])
def test_mog_estep_on_split_operands(ctx, split_ctx, terms, N, D, K):
    """Config 3's pass with the forward on three bf16 terms and the backward on two: tests/test_mog_gpu.py's
    comparison and tolerances."""
    rs = np.random.RandomState(N + D + K)
    centres = rs.standard_normal((K, D)) * 3
    X = (centres[rs.randint(K, size=N)] + rs.standard_normal((N, D))).astype(np.float32)
    T = rs.uniform(0.5, 2.0, (K, D))
    Wmat = np.concatenate([T * centres, -0.5 * T], axis=1).astype(np.float32)
    c = (rs.standard_normal(K) - 0.5 * (T * centres ** 2).sum(1)).astype(np.float32)
    f32_stats, _ = _estep(ctx, X, Wmat, c)
    split_ctx(terms)
    stats, lse = _estep(ctx, X, Wmat, c)
    want, lse_ref = svi.mog_local_step(X, Wmat, c)
    X64 = X.astype(np.float64)
    scale = np.concatenate([[max(N, 1)], np.abs(X64).sum(0) + 1e-9, (X64 ** 2).sum(0) + 1e-9])
    assert (np.abs(stats - want) <= 2e-5 * scale[None, :] + 1e-9).all(), \
        np.abs((stats - want) / scale[None, :]).max()
    npt.assert_allclose(stats[:, 0].sum(), N, rtol=1e-6, atol=1e-6)
    npt.assert_allclose(lse, lse_ref, rtol=2e-6, atol=1e-4)
    assert not np.array_equal(stats, f32_stats)
    again, lse2 = _estep(ctx, X, Wmat, c)
    npt.assert_array_equal(stats, again)
    assert lse == lse2


def test_mog_estep_split_operand_layout_with_separated_integer_data(ctx, split_ctx):
    """tests/test_mog_gpu.py's layout test: one-hot responsibilities, integer data.  Counts and first moments are exact
    (every operand fits its terms); squares of up to 18 bits do not fit two terms: 2^-17 of each."""
    K, D, N = 64, 16, 64 * 40
    k = np.arange(K)
    centres = np.zeros((K, D))
    centres[k, k % D] = 100.0 * (1 + k // D)
    labels = np.arange(N) % K
    offs = (np.arange(N)[:, None] * 7 + np.arange(D)[None, :] * 3) % 5 - 2
    X = (centres[labels] + offs).astype(np.float32)
    T = np.ones((K, D))
    Wmat = np.concatenate([T * centres, -0.5 * T], axis=1).astype(np.float32)
    c = (-0.5 * (centres ** 2).sum(1)).astype(np.float32)
    split_ctx(2)
    stats, _ = _estep(ctx, X, Wmat, c)
    want = np.zeros((K, 1 + 2 * D))
    for n in range(N):
        want[labels[n], 0] += 1
        want[labels[n], 1:1 + D] += X[n]
        want[labels[n], 1 + D:] += X[n].astype(np.float64) ** 2
    npt.assert_array_equal(stats[:, :1 + D], want[:, :1 + D])
    npt.assert_allclose(stats[:, 1 + D:], want[:, 1 + D:], rtol=1e-5)


def test_mog_driver_on_split_operands_tracks_the_oracle(ctx, split_ctx):
    from bayesic_amd.svi.mog import MoGNatGradSVI
    K, D = 8, 16
    X, _, _ = svi.make_cfg3(40000, D, K)
    eta0 = svi.mog_prior_eta(K, D)
    eta = svi.mog_init_eta(X[:4000], K, D, seed=1)
    split_ctx(2)
    model = MoGNatGradSVI(X, K, eta0, eta, n_total=3 * len(X), ctx=ctx)
    for t in range(1, 4):
        Wmat, c = svi.mog_expected_params(eta, K, D)
        _, lse = svi.mog_local_step(X, Wmat, c)
        want = svi.mog_elbo(eta, eta0, lse, 3.0, K, D)
        model.step()
        eta, _, _ = svi.mog_svi_step(eta, eta0, X, 3 * len(X), (t + 1.0) ** -0.6, K, D)
        ctx.sync()
        npt.assert_allclose(model.elbo.item(), want, rtol=2e-6)
        npt.assert_allclose(model.eta.cpu().numpy(), eta, rtol=2e-4, atol=1e-6)
        model.eta.copy_(torch.as_tensor(eta, dtype=torch.float64))


@pytest.mark.parametrize("N", [4096 + 17, 33_000, 262_144 + 5])
def test_gram_ping_pong_loop_gives_the_same_bits(N):
    """Option gram_pp = 1 (gram256_pp_kernel: the two waves of a SIMD take turns on the matrix pipe; measured no faster,
    off by default): the same products added in the same order -- bit for bit gram256_bx_kernel's result, ragged
    row counts and a padded leading dimension included."""
    from bayesic_amd.device import Context
    D, ld = 256, 260
    rs = np.random.RandomState(N)
    Xp = (rs.standard_normal((N, ld)) + 0.3).astype(np.float32)
    outs = []
    for pp in (0, 1):
        c = Context(0, options=dict(gram_pp=pp))
        c.call("bsc_ctx_set_mfma_split", 2)
        Xd = c.to_device(Xp)
        out = torch.full((D, D), float("nan"), dtype=torch.float32, device=c.device)
        c.call("bsc_gemm_strided_batched", 0, 1, D, D, N, Xd, 0, 1, ld, Xd, 0, ld, 1, out, 0, D, 1)
        c.sync()
        outs.append(out.cpu().numpy())
    X_ = Xp[:, :D].astype(np.float64)
    n2 = np.sqrt((X_ ** 2).sum(0))
    assert (np.abs(outs[1] - X_.T @ X_) <= 2e-5 * n2[:, None] * n2[None, :]).all()
    npt.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("N,D,ld", [(5000, 256, 256), (4096, 32, 32), (100000, 96, 100), (7001, 160, 160), (300000, 256, 256)])
def test_gram_statistic_on_split_operands(ctx, split_ctx, N, D, ld):
    """dot(X.T, X) -- the summed second-moment statistic -- with X as two bf16 terms (csrc/bsc_gram.hip): through the
    C ABI's product entry and through the executor, against float64; symmetric to the bit."""
    from bayesic_amd import algebra as A
    from bayesic_amd.algebra.device_backend import DeviceBackend
    rs = np.random.RandomState(N + D)
    Xp = (rs.standard_normal((N, ld)) + 0.3).astype(np.float32)
    X_ = Xp[:, :D]
    Xd = ctx.to_device(Xp)
    want = X_.astype(np.float64).T @ X_.astype(np.float64)
    n2 = np.sqrt((X_.astype(np.float64) ** 2).sum(0))
    scale = n2[:, None] * n2[None, :]
    f32 = ctx.zeros((D, D), torch.float32)
    ctx.call("bsc_gemm_strided_batched", 0, 1, D, D, N, Xd, 0, 1, ld, Xd, 0, ld, 1, f32, 0, D, 1)
    split_ctx(2)
    out = torch.full((D, D), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_gemm_strided_batched", 0, 1, D, D, N, Xd, 0, 1, ld, Xd, 0, ld, 1, out, 0, D, 1)
    ctx.sync()
    got = out.cpu().numpy()
    assert (np.abs(got - want) <= 2e-5 * scale).all(), (np.abs(got - want) / scale).max()
    npt.assert_array_equal(got, got.T)
    out2 = torch.full((D, D), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_gemm_strided_batched", 0, 1, D, D, N, Xd, 0, 1, ld, Xd, 0, ld, 1, out2, 0, D, 1)
    ctx.sync()
    npt.assert_array_equal(got, out2.cpu().numpy())            # run-to-run identical
    assert not np.array_equal(got, f32.cpu().numpy())
    if ld == D:
        Xv = A.var("X", 2)
        ex = A.dot(Xv.T, Xv).compile(DeviceBackend(ctx))(X=X_)
        npt.assert_array_equal(ex, got)
        ex2 = (A.dot(Xv.T, Xv) * 0.5).compile(DeviceBackend(ctx))(X=X_)
        npt.assert_allclose(ex2, 0.5 * got, rtol=1e-6)


def test_plugin_surface_honours_the_split_option(ctx, split_ctx):
    """Config 3's SYMBOLIC model (inference/mixture.py) is recognised and runs the fused kernels; with the context's
    split option on, those are the bf16 ones: the oracle's update and bound at the f32 route's tolerances."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference.mixture import DiagonalMixtureVMP
    n, d, k = 50_000, 16, 64
    X, _, _ = svi.make_cfg3(n, d, k)
    prior = dict(alpha0=1.5, m0=0.1, kappa0=0.05, a0=2.0, b0=0.7)
    eta0 = svi.mog_prior_eta(k, d, **prior)
    eta = svi.mog_init_eta(X[:500], k, d, seed=2)
    alpha, m, kappa, a, b = svi.mog_unpack(eta, k, d)
    plain = DiagonalMixtureVMP(X, k, n_total=10.0 * n, init=(alpha, m, kappa, a, b), backend=DeviceBackend(ctx), **prior)
    plain.step(0.5)
    plain_eta = plain.eta_fused_layout()
    split_ctx(2)
    model = DiagonalMixtureVMP(X, k, n_total=10.0 * n, init=(alpha, m, kappa, a, b), backend=DeviceBackend(ctx), **prior)
    assert model.route.startswith("fused"), model.route
    Wmat, c = svi.mog_expected_params(eta, k, d)
    _, lse = svi.mog_local_step(X, Wmat, c)
    want_elbo = svi.mog_elbo(eta, eta0, lse, 10.0, k, d)
    model.step(0.5)
    want, _, _ = svi.mog_svi_step(eta, eta0, X, 10.0 * n, 0.5, k, d)
    npt.assert_allclose(model.elbo(), want_elbo, rtol=2e-6)
    got = model.eta_fused_layout()
    scale = np.maximum(np.abs(want), 1.0)
    assert (np.abs(got - want) <= 1e-3 * scale).all()
    assert not np.array_equal(got, plain_eta)
