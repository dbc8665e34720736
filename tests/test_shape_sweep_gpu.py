"""Seeded sweeps of awkward extents and padded leading dimensions through the data-sized C-ABI
entry points, each against the float64 oracle.  The fixed cases elsewhere pin the BASELINE sizes
and hand-picked edges; these visit what nobody picked: row counts that end mid-tile, column counts
that are not a tile, operands that are views of wider buffers (so the uniform-base "fast" loaders
and the general ones both run, depending on where a tile lies)."""
import numpy as np
import pytest

from oracle import svi

pytestmark = pytest.mark.gpu


def padded(ctx, a, extra):
    """Device view of `a` inside a buffer whose rows are `extra` floats wider (filled with NaN)."""
    wide = np.full((a.shape[0], a.shape[1] + extra), np.nan, np.float32)
    wide[:, :a.shape[1]] = a
    return ctx.to_device(wide)[:, :a.shape[1]]


@pytest.mark.parametrize("seed", range(12))
def test_blr_pass_sweep(ctx, seed):
    import torch
    rs = np.random.RandomState(100 + seed)
    B = int(rs.choice([1, 15, 16, 17, 255, 4097, 33333]))
    D = int(rs.choice([4, 32, 124, 252, 256]))
    S = int(rs.choice([1, 3, 8]))
    pad = int(rs.choice([0, 4, 12]))
    X = rs.standard_normal((B, D)).astype(np.float32)
    y = rs.standard_normal(B).astype(np.float32)
    W = (rs.standard_normal((S, D)) / 8).astype(np.float32)
    Xd = padded(ctx, X, pad)
    Q = ctx.zeros(S, torch.float64)
    G = ctx.zeros((S, D), torch.float64)
    ctx.call("bsc_blr_data_pass", Xd, Xd.stride(0), ctx.to_device(y), B, D, ctx.to_device(W), S, Q, G)
    ctx.sync()
    q_ref, g_ref = svi.blr_data_pass(X, y, W)
    # a residual is a difference of float32 values of size |y| + |x|.|w|: bound Q's error by that
    size = (np.abs(y.astype(np.float64))[:, None] + np.abs(X.astype(np.float64)) @ np.abs(W.astype(np.float64)).T)
    assert (np.abs(Q.cpu().numpy() - q_ref) <= 3e-6 * (q_ref + 1e-1 * (size ** 2).sum(0))).all()
    scale = np.sqrt(q_ref)[:, None] * np.sqrt((X.astype(np.float64) ** 2).sum(0))[None, :] + 1e-12
    g_size = (size[:, :, None] * np.abs(X.astype(np.float64))[:, None, :]).sum(0)
    assert (np.abs(G.cpu().numpy() - g_ref) <= 2e-5 * scale + 3e-7 * g_size).all()


@pytest.mark.parametrize("seed", range(10))
def test_lda_dense_sweep(ctx, seed):
    import torch
    rs = np.random.RandomState(200 + seed)
    docs = int(rs.choice([1, 31, 32, 33, 100, 257]))
    V = int(rs.choice([1, 127, 128, 129, 640, 1000]))
    K = int(rs.choice([32, 64, 96, 128]))
    pc, pt = int(rs.choice([0, 4, 8])), int(rs.choice([0, 4]))
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    Th = rs.uniform(0.1, 1.0, (docs, K)).astype(np.float32)
    Bt = rs.uniform(0.1, 1.0, (K, V)).astype(np.float32)
    Cd, Thd = padded(ctx, C, pc), padded(ctx, Th, pt)
    out = torch.full((K, V), float("nan"), device=ctx.device)
    ctx.call("bsc_lda_sstats", Cd, Cd.stride(0), docs, V, K, Thd, Thd.stride(0), ctx.to_device(Bt), V, out, V)
    ctx.sync()
    np.testing.assert_allclose(out.cpu().numpy(), svi.lda_sstats(C, Th, Bt), rtol=3e-5, atol=1e-6)


@pytest.mark.parametrize("seed", range(10))
def test_bbvi_loglik_sweep(ctx, seed):
    import torch
    rs = np.random.RandomState(300 + seed)
    N = int(rs.choice([1, 31, 32, 33, 1000, 20001]))
    D = int(rs.choice([4, 64, 200, 256]))
    G = int(rs.choice([1, 7, 300]))
    pad = int(rs.choice([0, 4, 16]))
    X = rs.standard_normal((N, D)).astype(np.float32)
    y = (rs.uniform(size=N) < 0.4).astype(np.float32)
    g = rs.randint(G, size=N).astype(np.int32)
    Wz = (rs.standard_normal((64, D)) / 8).astype(np.float32)
    Bz = rs.standard_normal((G, 64)).astype(np.float32)
    Xd = padded(ctx, X, pad)
    ell = ctx.zeros(64, torch.float64)
    ctx.call("bsc_logreg_bbvi_loglik", Xd, Xd.stride(0), ctx.to_device(y), ctx.to_device(g), N, D, G,
             ctx.to_device(Wz), ctx.to_device(Bz), 64, ell)
    ctx.sync()
    want = svi.logreg_loglik(X, y, g, Wz, Bz)
    np.testing.assert_allclose(ell.cpu().numpy(), want, rtol=5e-6, atol=1e-4)


@pytest.mark.parametrize("seed", range(10))
def test_mog_estep_sweep(ctx, seed):
    import torch
    rs = np.random.RandomState(400 + seed)
    N = int(rs.choice([1, 31, 32, 33, 2049, 30000]))
    D = int(rs.choice([1, 3, 8, 16]))
    K = int(rs.choice([1, 5, 33, 64]))
    pad = int(rs.choice([0, 1, 4]))
    cen = rs.standard_normal((K, D)) * 2
    X = (cen[rs.randint(K, size=N)] + rs.standard_normal((N, D))).astype(np.float32)
    T = rs.uniform(0.5, 2.0, (K, D))
    Wmat = np.concatenate([T * cen, -0.5 * T], axis=1).astype(np.float32)
    c = (-0.5 * (T * cen ** 2).sum(1)).astype(np.float32)
    Xd = padded(ctx, X, pad)
    stats = ctx.zeros((K, 1 + 2 * D), torch.float64)
    lse = ctx.zeros(1, torch.float64)
    ctx.call("bsc_mog_estep", Xd, Xd.stride(0), N, D, K, ctx.to_device(Wmat), ctx.to_device(c), stats, lse)
    ctx.sync()
    s_ref, l_ref = svi.mog_local_step(X, Wmat, c)
    X64 = np.abs(X.astype(np.float64))
    scale = np.concatenate([[max(N, 1)], X64.sum(0) + 1e-9, (X64 ** 2).sum(0) + 1e-9])
    assert (np.abs(stats.cpu().numpy() - s_ref) <= 3e-5 * scale[None, :] + 1e-9).all()
    np.testing.assert_allclose(lse.item(), l_ref, rtol=3e-6, atol=1e-4)


@pytest.mark.parametrize("seed", range(10))
def test_gemm_strided_sweep(ctx, seed):
    """Padded, transposed and batched operands: M/N/K that end mid-tile and mid-k-step."""
    import torch
    rs = np.random.RandomState(500 + seed)
    M, N, K = (int(rs.choice([1, 127, 128, 129, 300])), int(rs.choice([1, 128, 200, 257])),
               int(rs.choice([1, 31, 32, 33, 1000, 2049])))
    batch = int(rs.choice([1, 3]))
    a_t = bool(rs.randint(2))
    pa, pb = int(rs.choice([0, 4])), int(rs.choice([0, 4, 12]))
    A = rs.standard_normal((batch, K, M) if a_t else (batch, M, K)).astype(np.float32)
    B = rs.standard_normal((batch, K, N)).astype(np.float32)

    def up(a, extra):
        wide = np.full(a.shape[:-1] + (a.shape[-1] + extra,), np.nan, np.float32)
        wide[..., :a.shape[-1]] = a
        return ctx.to_device(wide)[..., :a.shape[-1]]

    Ad, Bd = up(A, pa), up(B, pb)
    C = torch.full((batch, M, N), float("nan"), device=ctx.device)
    sa_m, sa_k = (Ad.stride(2), Ad.stride(1)) if a_t else (Ad.stride(1), Ad.stride(2))
    ctx.call("bsc_gemm_strided_batched", 0, batch, M, N, K, Ad, Ad.stride(0), sa_m, sa_k,
             Bd, Bd.stride(0), Bd.stride(1), Bd.stride(2), C, M * N, N, 1)
    ctx.sync()
    A64 = (A.transpose(0, 2, 1) if a_t else A).astype(np.float64)
    want = A64 @ B.astype(np.float64)
    bound = np.abs(A64) @ np.abs(B.astype(np.float64))
    assert (np.abs(C.cpu().numpy() - want) <= 1e-5 * bound + 1e-30).all()


@pytest.mark.parametrize("seed", range(8))
def test_sparse_lda_sweep(ctx, seed):
    import scipy.sparse as sp
    import torch
    rs = np.random.RandomState(600 + seed)
    docs = int(rs.choice([1, 40, 333]))
    V = int(rs.choice([1, 63, 64, 65, 500]))
    K = int(rs.choice([32, 64, 96, 128]))
    pt = int(rs.choice([0, 4]))
    density = float(rs.choice([0.0, 0.02, 0.5]))
    C = (rs.poisson(1.0, (docs, V)) * (rs.uniform(size=(docs, V)) < density)).astype(np.float32)
    Th = rs.uniform(0.1, 1.0, (docs, K)).astype(np.float32)
    Bt = rs.uniform(0.1, 1.0, (K, V)).astype(np.float32)
    csc = sp.csc_matrix(C)
    dev = ctx.device
    colptr = torch.from_numpy(csc.indptr.astype(np.int64)).to(dev)
    rowidx = torch.from_numpy(csc.indices.astype(np.int32)).to(dev) if csc.nnz else torch.zeros(1, dtype=torch.int32, device=dev)
    vals = torch.from_numpy(csc.data.astype(np.float32)).to(dev) if csc.nnz else torch.zeros(1, device=dev)
    Thd = padded(ctx, Th, pt)
    out = torch.full((K, V), float("nan"), device=dev)
    ctx.call("bsc_lda_sstats_csc", colptr, rowidx, vals, docs, V, K, Thd, Thd.stride(0), ctx.to_device(Bt), V, out, V)
    ctx.sync()
    np.testing.assert_allclose(out.cpu().numpy(), svi.lda_sstats(C, Th, Bt), rtol=3e-5, atol=1e-6)


@pytest.mark.parametrize("seed", range(8))
def test_weighted_outer_sweep(ctx, seed):
    import torch
    rs = np.random.RandomState(700 + seed)
    N = int(rs.choice([1, 63, 64, 65, 5000]))
    K = int(rs.choice([4, 28, 32, 36, 64]))
    D = int(rs.choice([4, 12, 16, 32]))
    E = int(rs.choice([4, 8, 32]))
    sym = bool(rs.randint(2))
    pr, px = int(rs.choice([0, 4])), int(rs.choice([0, 8]))
    R = rs.standard_normal((N, K)).astype(np.float32)
    X = rs.standard_normal((N, D)).astype(np.float32)
    Y = X if sym else rs.standard_normal((N, E)).astype(np.float32)
    Rd, Xd = padded(ctx, R, pr), padded(ctx, X, px)
    Yd = Xd if sym else padded(ctx, Y, 4)
    out = torch.full((K, D, Y.shape[1]), float("nan"), device=ctx.device)
    ctx.call("bsc_weighted_outer", Rd, Rd.stride(0), Xd, Xd.stride(0), Yd, Yd.stride(0), N, K, D,
             Y.shape[1], 1.0, out)
    ctx.sync()
    R64, X64, Y64 = (a.astype(np.float64) for a in (R, X, Y))
    want = np.einsum("nk,nd,ne->kde", R64, X64, Y64)
    bound = np.einsum("nk,nd,ne->kde", np.abs(R64), np.abs(X64), np.abs(Y64))
    assert (np.abs(out.cpu().numpy() - want) <= 2e-5 * bound + 1e-30).all()
