"""GPU parity of the algebra front end's device executor (HIP kernels through the
C ABI) against numpy -- the reference's own numeric oracle
(bayesic/tests/test_algebra.py:44-191) -- and against the einsum definition.

Tolerances are the reference's: rtol 1e-5 wherever a sum/contraction is involved
(test_algebra.py:82), tight for pure element-wise results (device libm vs numpy:
rtol 1e-6 for log/exp/pow, exact for + - * abs).
"""
import numpy as np
import numpy.testing as npt
import pytest

import test_algebra as T                     # the CPU suite's inputs and CHECKS
from bayesic_amd.algebra import *            # noqa: F401,F403
from oracle.einsum_eval import einsum_semantics

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(ctx):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    return DeviceBackend(ctx)


def run(dev, expr, **inputs):
    return expr.compile(dev)(**inputs)


def test_reference_numeric_tests_on_device(dev):
    X, Y, x, y, S, a = T.X, T.Y, T.x, T.y, T.S, T.a
    X_, Y_, x_, y_, S_ = T.X_, T.Y_, T.x_, T.y_, T.S_
    npt.assert_array_equal(run(dev, X + Y, X=X_, Y=Y_), X_ + Y_)
    npt.assert_array_equal(run(dev, X - Y, X=X_, Y=Y_), X_ - Y_)
    npt.assert_array_equal(run(dev, abs(X), X=X_), abs(X_))
    npt.assert_array_equal(run(dev, X + 1, X=X_), X_ + 1)
    npt.assert_array_equal(run(dev, 1 - X, X=X_), 1 - X_)
    npt.assert_array_equal(run(dev, 2 * X, X=X_), 2 * X_)
    assert run(dev, add(1, 1)) == 2
    npt.assert_array_equal(run(dev, X * Y_, X=X_), X_ * Y_)
    npt.assert_allclose(run(dev, dot(X, Y), X=X_, Y=Y_), np.dot(X_, Y_), rtol=1e-5)
    npt.assert_allclose(run(dev, X.dot(y), X=X_, y=y_), np.dot(X_, y_), rtol=1e-5)
    npt.assert_allclose(run(dev, dot(x, y), x=x_, y=y_), np.dot(x_, y_), rtol=1e-5)
    npt.assert_array_equal(run(dev, X * Y, X=X_, Y=Y_), X_ * Y_)
    npt.assert_allclose(run(dev, X / Y, X=X_, Y=Y_), X_ / Y_, rtol=1e-5)
    with np.errstate(all="ignore"):
        npt.assert_allclose(run(dev, X ** Y, X=X_, Y=Y_), X_ ** Y_, rtol=1e-6)   # NaN == NaN
        npt.assert_allclose(run(dev, 2 ** X, X=X_), 2 ** X_, rtol=1e-6)
        npt.assert_allclose(run(dev, X ** 2, X=X_), X_ ** 2, rtol=1e-6)
        npt.assert_allclose(run(dev, log(X), X=X_), np.log(X_), rtol=1e-6)
    npt.assert_allclose(run(dev, exp(X), X=X_), np.exp(X_), rtol=1e-6)
    npt.assert_array_equal(run(dev, X.T, X=X_), X_.T)
    npt.assert_array_equal(run(dev, dimshuffle(S, 2, 0, 1), S=S_), np.transpose(S_, (2, 0, 1)))
    npt.assert_array_equal(run(dev, X + x.dimshuffle(0, "x"), X=X_, x=x_), X_ + x_[:, None])
    npt.assert_array_equal(run(dev, X * dimshuffle(x, "x", 0), X=X_, x=x_), X_ * x_[None, :])
    npt.assert_allclose(run(dev, trace(X), X=X_), np.trace(X_), rtol=1e-5)
    npt.assert_array_equal(run(dev, diagonal(X), X=X_), np.diagonal(X_))
    npt.assert_array_equal(run(dev, outer(x, y), x=x_, y=y_), np.outer(x_, y_))
    npt.assert_allclose(run(dev, sum(S), S=S_), S_.sum(), rtol=1e-5)
    npt.assert_allclose(run(dev, sum(S, axis=0), S=S_), S_.sum(axis=0), rtol=1e-5)
    npt.assert_allclose(run(dev, S.sum(axis=(0, 2)), S=S_), S_.sum(axis=(0, 2)), rtol=1e-5)
    data = np.array([[1, 2], [3, 4], [5, 6]], dtype="float32")
    assert run(dev, X.shape[0], X=data) == 3 and run(dev, X.size, X=data) == 6
    npt.assert_equal(run(dev, eye(a), a=2), np.eye(2))
    npt.assert_equal(run(dev, eye(a), a=5), np.eye(5))
    expr = dot(diagonal(dot(X, outer(x, y))), Y)
    npt.assert_allclose(run(dev, expr, x=x_, y=y_, X=X_, Y=Y_),
                        np.dot(np.diagonal(np.dot(X_, np.outer(x_, y_))), Y_), rtol=1e-5)


@pytest.mark.parametrize("case", range(len(T.CHECKS)))
def test_corpus_on_device_equals_einsum_definition(dev, case):
    build, expected = T.CHECKS[case]
    e = build()
    got = run(dev, e, **{k: T.VALUES[k] for k in e.input_types})
    want = expected()
    if want is not None:
        npt.assert_allclose(got, want, rtol=2e-5, atol=1e-6)
    if isinstance(e, Einsum):
        ref64 = einsum_semantics(e, {k: T.VALUES[k].astype(np.float64) for k in e.input_types})
        assert got.shape == ref64.shape
        npt.assert_allclose(got, ref64, rtol=2e-5, atol=1e-5)


def test_batched_tensordot_on_device(dev):
    X, Y, S = T.X, T.Y, T.S
    e = (X * Y.T).sum(axis=1)
    npt.assert_allclose(run(dev, e, X=T.X_, Y=T.Y_), (T.X_ * T.Y_.T).sum(axis=1), rtol=1e-5)
    e2 = tensordot(S, S, [1], [1], [0], [0])
    npt.assert_allclose(run(dev, e2, S=T.S_), np.einsum("uiv,uiw->uvw", T.S_, T.S_), rtol=1e-5)


def test_batched_tensordot_more_batches_than_a_grid_dimension(dev):
    """70 000 batches of a 2x3 . 3x2 product (the batch index is a grid dimension
    of at most 65 535) -- also with fusion off, where batched dot products take the
    GEMM path as well."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    rs = np.random.RandomState(5)
    A_ = rs.standard_normal((70000, 2, 3)).astype(np.float32)
    B_ = rs.standard_normal((70000, 3, 2)).astype(np.float32)
    S1, S2 = var("S1", 3), var("S2", 3)
    e = tensordot(S1, S2, [2], [1], [0], [0])
    want = np.einsum("bik,bkj->bij", A_.astype(np.float64), B_)
    npt.assert_allclose(run(dev, e, S1=A_, S2=B_), want, rtol=1e-5, atol=1e-6)
    e2 = tensordot(S1, S1, [1, 2], [1, 2], [0], [0])          # batched dot products
    want2 = (A_.astype(np.float64) ** 2).sum(axis=(1, 2))
    npt.assert_allclose(run(dev, e2, S1=A_), want2, rtol=1e-5)
    unfused = DeviceBackend(dev.ctx, fuse=False)
    npt.assert_allclose(e2.compile(unfused)(S1=A_), want2, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (33, 17, 5), (128, 128, 16), (130, 257, 1000),
                                   (256, 256, 40000), (64, 16, 100003), (5, 300, 2), (300, 5, 777)])
def test_gemm_shapes_incl_split_k(dev, M, N, K):
    """dot(A.T, B) with K long: exercises m-contiguous operands, ragged tiles and the
    deterministic split-K (the X^T X / R^T X shapes of configs 2 and 3)."""
    rs = np.random.RandomState(M + N + K)
    A_ = rs.standard_normal((K, M)).astype(np.float32)
    B_ = rs.standard_normal((K, N)).astype(np.float32)
    A, Bv = var("A", 2), var("Bv", 2)
    got = run(dev, dot(A.T, Bv), A=A_, Bv=B_)
    want = A_.astype(np.float64).T @ B_.astype(np.float64)
    bound = np.sqrt((A_.astype(np.float64) ** 2).sum(0))[:, None] * \
        np.sqrt((B_.astype(np.float64) ** 2).sum(0))[None, :]
    assert (np.abs(got - want) <= 1e-5 * bound + 1e-12).all()
    assert (got == run(dev, dot(A.T, Bv), A=A_, Bv=B_)).all()       # deterministic
    # k-contiguous operands of the same product
    got2 = run(dev, dot(A, Bv.T), A=np.ascontiguousarray(A_.T), Bv=np.ascontiguousarray(B_.T))
    assert (np.abs(got2 - want) <= 1e-5 * bound + 1e-12).all()


def test_sum_access_patterns_and_determinism(dev):
    rs = np.random.RandomState(3)
    Xb = rs.standard_normal((20000, 96)).astype(np.float32)
    X = T.X
    for expr, want in [(sum(X, 0), Xb.astype(np.float64).sum(0)), (sum(X, 1), Xb.astype(np.float64).sum(1)),
                       (sum(X), Xb.astype(np.float64).sum()), (sum(X.T, 0), Xb.astype(np.float64).sum(1))]:
        got = run(dev, expr, X=Xb)
        npt.assert_allclose(got, want, rtol=1e-5, atol=1e-3)
        npt.assert_array_equal(got, run(dev, expr, X=Xb))
    S5 = rs.standard_normal((7, 5, 3, 4, 6)).astype(np.float32)
    V = var("V", 5)
    npt.assert_allclose(run(dev, sum(V, axis=(1, 3)), V=S5), S5.sum(axis=(1, 3)), rtol=1e-5, atol=1e-5)


def test_float64_inputs_stay_float64(dev):
    A, b = var("A", 2, "float64"), var("b", 1, "float64")
    rs = np.random.RandomState(0)
    A_, b_ = rs.standard_normal((40, 30)), rs.standard_normal(30)
    got = run(dev, dot(A, b) + exp(sum(A, 1)), A=A_, b=b_)
    assert got.dtype == np.float64
    npt.assert_allclose(got, A_ @ b_ + np.exp(A_.sum(1)), rtol=1e-12)


def test_config_shaped_expressions_on_device(dev):
    """The lowered forms of SURVEY 8(a) A7 at small sizes: conjugate BLR statistics,
    MoG responsibilities-weighted moments, LDA sufficient statistics."""
    rs = np.random.RandomState(7)
    n, d, k = 3000, 32, 8
    Xv = rs.standard_normal((n, d)).astype(np.float32)
    yv = rs.standard_normal(n).astype(np.float32)
    Rv = rs.dirichlet(np.ones(k), n).astype(np.float32)
    X, y, R = var("X", 2), var("y", 1), var("R", 2)
    X64, y64, R64 = Xv.astype(np.float64), yv.astype(np.float64), Rv.astype(np.float64)
    npt.assert_allclose(run(dev, dot(X.T, X), X=Xv), X64.T @ X64, rtol=1e-4, atol=1e-2)
    npt.assert_allclose(run(dev, dot(X.T, y), X=Xv, y=yv), X64.T @ y64, rtol=1e-4, atol=1e-2)
    npt.assert_allclose(run(dev, sum(y * y), y=yv), y64 @ y64, rtol=1e-5)
    npt.assert_allclose(run(dev, sum(R, 0), R=Rv), R64.sum(0), rtol=1e-5)
    npt.assert_allclose(run(dev, dot(R.T, X), R=Rv, X=Xv), R64.T @ X64, rtol=1e-4, atol=1e-3)
    npt.assert_allclose(run(dev, dot(R.T, X * X), R=Rv, X=Xv), R64.T @ (X64 * X64), rtol=1e-4, atol=1e-3)
    Th, C, Bm = var("Th", 2), var("C", 2), var("Bm", 2)
    docs, V, K = 60, 500, 16
    Thv = rs.gamma(1.0, 1.0, (docs, K)).astype(np.float32)
    Cv = rs.poisson(0.3, (docs, V)).astype(np.float32)
    Bv = rs.gamma(1.0, 1.0, (K, V)).astype(np.float32)
    npt.assert_allclose(run(dev, Bm * dot(Th.T, C), Bm=Bv, Th=Thv, C=Cv),
                        Bv.astype(np.float64) * (Thv.astype(np.float64).T @ Cv.astype(np.float64)),
                        rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,d", [(1000, 300), (70000, 256), (64, 64), (5000, 7)])
def test_matrix_vector_paths(dev, n, d):
    """X w, X^T y and the vector-matrix forms take the streaming GEMV kernels."""
    rs = np.random.RandomState(n + d)
    Xv = rs.standard_normal((n, d)).astype(np.float32)
    wv = rs.standard_normal(d).astype(np.float32)
    yv = rs.standard_normal(n).astype(np.float32)
    X, w, y = var("X", 2), var("w", 1), var("y", 1)
    X64 = Xv.astype(np.float64)
    tol = dict(rtol=1e-5, atol=1e-5 * np.sqrt(max(n, d)))
    npt.assert_allclose(run(dev, dot(X, w), X=Xv, w=wv), X64 @ wv, **tol)
    npt.assert_allclose(run(dev, dot(X.T, y), X=Xv, y=yv), X64.T @ yv, **tol)
    npt.assert_allclose(run(dev, dot(y, X), X=Xv, y=yv), yv @ X64, **tol)
    npt.assert_allclose(run(dev, dot(w, X.T), X=Xv, w=wv), wv @ X64.T, **tol)
    got = run(dev, dot(X.T, y), X=Xv, y=yv)
    assert (got == run(dev, dot(X.T, y), X=Xv, y=yv)).all()


@pytest.mark.parametrize("M,N,K", [(8, 100003, 256), (1, 4099, 256), (16, 70000, 64), (17, 50001, 252),
                                   (32, 9000, 128), (5, 20000, 4), (31, 33333, 200), (8, 4096, 16)])
def test_skinny_products_one_tiny_extent_nt(dev, M, N, K):
    """dot(W, X.T) -> _tensordot(W, _dimshuffle(X,1,0), [1],[0]) with M <= 32 draws, and the same
    product the other way round (dot(X, W.T)): the LDS-DMA kernel of csrc/bsc_skinny.hip, against
    float64 numpy and against the 128 x 128-tile GEMM on the same operands."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.device import Context
    import os
    rs = np.random.RandomState(M * 7 + K)
    W_ = rs.standard_normal((M, K)).astype(np.float32)
    X_ = rs.standard_normal((N, K)).astype(np.float32)
    W, X = var("W", 2), var("X", 2)
    want = W_.astype(np.float64) @ X_.astype(np.float64).T
    bound = np.sqrt((W_.astype(np.float64) ** 2).sum(1))[:, None] * \
        np.sqrt((X_.astype(np.float64) ** 2).sum(1))[None, :]
    got = run(dev, dot(W, X.T), W=W_, X=X_)
    assert got.shape == (M, N)
    assert (np.abs(got - want) <= 1e-5 * bound + 1e-12).all(), np.abs(got - want).max()
    got_t = run(dev, dot(X, W.T), W=W_, X=X_)                # C^T: strided stores
    assert (np.abs(got_t - want.T) <= 1e-5 * bound.T + 1e-12).all()
    plain = DeviceBackend(Context(0, options=dict(gemm_skinny=0)))
    ref = dot(W, X.T).compile(plain)(W=W_, X=X_)
    # both are k-ordered fp32 fma chains over different groupings: equal to rounding, not bitwise
    assert (np.abs(got - ref) <= 2e-5 * bound + 1e-12).all()
    assert (got == run(dev, dot(W, X.T), W=W_, X=X_)).all()          # deterministic


@pytest.mark.parametrize("M,D,K", [(8, 256, 100003), (1, 256, 20000), (16, 64, 70001), (3, 252, 16384),
                                   (8, 4, 50000), (13, 128, 1000003)])
def test_skinny_products_long_contraction_tn(dev, M, D, K):
    """dot(R, X) -> _tensordot(R, X, [1],[0]) with M <= 16 rows and a long contracted axis (the
    pathwise gradient G = R X of config 2's estimator in general form), and its transpose
    dot(X.T, R.T); deterministic float64 finish."""
    rs = np.random.RandomState(M * 11 + D)
    R_ = rs.standard_normal((M, K)).astype(np.float32)
    X_ = rs.standard_normal((K, D)).astype(np.float32)
    R, X = var("R", 2), var("X", 2)
    want = R_.astype(np.float64) @ X_.astype(np.float64)
    bound = np.sqrt((R_.astype(np.float64) ** 2).sum(1))[:, None] * \
        np.sqrt((X_.astype(np.float64) ** 2).sum(0))[None, :]
    got = run(dev, dot(R, X), R=R_, X=X_)
    assert got.shape == (M, D)
    assert (np.abs(got - want) <= 1e-5 * bound + 1e-12).all(), np.abs(got - want).max()
    got_t = run(dev, dot(X.T, R.T), R=R_, X=X_)
    assert (np.abs(got_t - want.T) <= 1e-5 * bound.T + 1e-12).all()
    assert (got == run(dev, dot(R, X), R=R_, X=X_)).all()


@pytest.mark.parametrize("M,N,K,batch", [(128, 128, 32, 1), (132, 260, 1004, 1), (130, 257, 1000, 1), (256, 384, 36, 1),
                                          (100, 516, 8, 3), (4, 8, 4, 2), (640, 132, 4100, 1), (257, 129, 20, 1),
                                          # one small tile, long contraction: the waves split the k-groups
                                          (64, 16, 96000, 1), (60, 64, 32768, 1), (8, 4, 40000, 1)])
def test_gemm_operands_by_lds_dma_all_layouts(ctx, M, N, K, batch):
    """bsc_gemm_strided_batched in the four operand layouts (each operand contiguous along its
    free axis or along k): the LDS-DMA kernel (gemm_f32_dma_kernel) wherever 16-byte pieces fall
    wholly inside or outside an operand, the register-staged kernel otherwise -- ragged tiles,
    a last k-tile shorter than 32, batches -- against float64, and the two kernels against each
    other on the same operands."""
    import os
    import torch
    from bayesic_amd.device import Context
    staged = Context(0, options=dict(gemm_dma=0))
    rs = np.random.RandomState(M + 3 * N + 7 * K)
    A_ = rs.standard_normal((batch, M, K)).astype(np.float32)
    B_ = rs.standard_normal((batch, K, N)).astype(np.float32)
    want = np.einsum("bmk,bkn->bmn", A_.astype(np.float64), B_.astype(np.float64))
    bound = np.sqrt((A_.astype(np.float64) ** 2).sum(2))[:, :, None] * np.sqrt((B_.astype(np.float64) ** 2).sum(1))[:, None, :]
    dev = ctx.device
    for a_m in (False, True):
        for b_n in (False, True):
            # A stored [b][m][k] (k contiguous) or [b][k][m] (m contiguous); B likewise
            At = torch.from_numpy(np.ascontiguousarray(A_.transpose(0, 2, 1)) if a_m else A_).to(dev)
            Bt = torch.from_numpy(B_ if b_n else np.ascontiguousarray(B_.transpose(0, 2, 1))).to(dev)
            sa = (M * K, 1, M) if a_m else (M * K, K, 1)
            sb = (K * N, N, 1) if b_n else (K * N, 1, K)
            outs = []
            for c in (ctx, staged):
                C = torch.full((batch, M, N), np.nan, dtype=torch.float32, device=dev)
                c.call("bsc_gemm_strided_batched", 0, batch, M, N, K, At, sa[0], sa[1], sa[2], Bt, sb[0], sb[1], sb[2],
                       C, M * N, N, 1)
                outs.append(C.cpu().numpy().astype(np.float64))
            assert (np.abs(outs[0] - want) <= 1e-5 * bound + 1e-12).all(), (a_m, b_n, np.abs(outs[0] - want).max())
            assert (np.abs(outs[0] - outs[1]) <= 2e-5 * bound + 1e-12).all()
            # the epilogue on the same product: C = 0.5 * E / acc
            E = torch.from_numpy(rs.standard_normal((batch, M, N)).astype(np.float32)).to(dev)
            C = torch.full((batch, M, N), np.nan, dtype=torch.float32, device=dev)
            ctx.call("bsc_gemm_epilogue", 0, batch, M, N, K, At, sa[0], sa[1], sa[2], Bt, sb[0], sb[1], sb[2],
                     C, M * N, N, 1, -1, 0.5, E, M * N, N, 1)
            ref = 0.5 * E.cpu().numpy().astype(np.float64) / outs[0]
            ok = np.abs(outs[0]) > 1e-2 * bound
            npt.assert_allclose(C.cpu().numpy()[ok], ref[ok], rtol=2e-5)


@pytest.mark.parametrize("rows,cols", [(100003, 64), (5000, 4), (70001, 128), (999, 36), (33, 8), (400000, 16)])
def test_sums_over_the_rows_of_a_narrow_matrix(dev, rows, cols):
    """sum(A, 0), sum(A * B, 0) and sum(A * w[:, None], 0) with fewer than 256 columns: the reduce
    kernel that packs several rows into one wave load (map_reduce_lane_narrow_f32_kernel), against
    float64 numpy; run-to-run identical."""
    rs = np.random.RandomState(rows + cols)
    A_ = rs.standard_normal((rows, cols)).astype(np.float32)
    B_ = rs.standard_normal((rows, cols)).astype(np.float32)
    w_ = rs.standard_normal(rows).astype(np.float32)
    A, Bv, w = var("A", 2), var("Bv", 2), var("w", 1)
    A64, B64, w64 = A_.astype(np.float64), B_.astype(np.float64), w_.astype(np.float64)
    scale = np.sqrt(rows)
    for expr, inputs, want in [
            (sum(A, 0), dict(A=A_), A64.sum(0)),
            (sum(A * Bv, 0), dict(A=A_, Bv=B_), (A64 * B64).sum(0)),
            (sum(A * dimshuffle(w, 0, "x"), 0), dict(A=A_, w=w_), (A64 * w64[:, None]).sum(0)),
            (sum(exp(A * 0.1), 0), dict(A=A_), np.exp(A64 * 0.1).sum(0))]:
        got = run(dev, expr, **inputs)
        assert got.shape == (cols,)
        npt.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * scale)
        npt.assert_array_equal(got, run(dev, expr, **inputs))


@pytest.mark.parametrize("rows,D,batch", [(50000, 384, 1), (4100, 260, 1), (20000, 256, 2), (1000, 640, 1), (333, 132, 3)])
def test_gram_matrix_computes_the_upper_triangle_of_tiles_once(ctx, rows, D, batch):
    """dot(X.T, X): the same matrix on both sides -- the GEMM computes the tiles on and above the
    diagonal and stores each off-diagonal one twice.  Against float64, against the full schedule
    (option gemm_sym = 0), exactly symmetric, run-to-run identical; both operand layouts."""
    import os
    import torch
    from bayesic_amd.device import Context
    full = Context(0, options=dict(gemm_sym=0))
    g = torch.Generator(device=ctx.device).manual_seed(rows + D)
    ld = D + 4
    X = torch.randn((batch, rows, ld), generator=g, device=ctx.device)
    X64 = X[:, :, :D].double()
    want = torch.matmul(X64.transpose(1, 2), X64).cpu().numpy()
    norm = X64.pow(2).sum(1).sqrt().cpu().numpy()
    bound = norm[:, :, None] * norm[:, None, :]
    # X^T X with X row-major (operands m- / n-contiguous), and X X^T with X stored transposed (k-contiguous)
    Xt = X[:, :, :D].transpose(1, 2).contiguous()                 # [batch, D, rows]
    for (A, sa, B, sb) in [(X, (rows * ld, 1, ld), X, (rows * ld, ld, 1)),
                           (Xt, (D * rows, rows, 1), Xt, (D * rows, 1, rows))]:
        outs = []
        for c in (ctx, full, ctx):
            C = torch.full((batch, D, D + 3), float("nan"), device=ctx.device)
            c.call("bsc_gemm_strided_batched", 0, batch, D, D, rows, A, *sa, B, *sb, C, D * (D + 3), D + 3, 1)
            c.sync()
            outs.append(C.cpu().numpy())
        got = outs[0][:, :, :D].astype(np.float64)
        assert (np.abs(got - want) <= 1e-5 * bound + 1e-12).all()
        assert (np.abs(got - outs[1][:, :, :D]) <= 2e-5 * bound + 1e-12).all()
        npt.assert_array_equal(outs[0][:, :, :D], outs[0][:, :, :D].transpose(0, 2, 1))      # exactly symmetric
        npt.assert_array_equal(outs[0][:, :, :D], outs[2][:, :, :D])
        assert np.isnan(outs[0][:, :, D:]).all()
