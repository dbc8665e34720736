"""bsc_weighted_outer (out[k,d,e] = scale * sum_n R[n,k] X[n,d] Y[n,e], the full-covariance
mixture statistic) against numpy float64 on the same float32 inputs, directly through the C ABI
and through the executor, which must recognise the lowered form
_tensordot(_mul(_dimshuffle(R,1,'x',0), _dimshuffle(X,'x',1,0)), X, [2], [0]) and not
materialise the K x D x N product.

Tolerance: float32 products summed in float32 per workgroup, float64 across workgroups:
2e-5 of sum_n |R||X||Y| per output."""
import numpy as np
import pytest

from bayesic_amd.algebra import *            # noqa: F401,F403
import builtins

pytestmark = pytest.mark.gpu
RNG = np.random.RandomState(77)


def reference(R, X, Y, scale=1.0):
    R, X, Y = (a.astype(np.float64) for a in (R, X, Y))
    want = scale * np.einsum("nk,nd,ne->kde", R, X, Y)
    bound = builtins.abs(scale) * np.einsum("nk,nd,ne->kde", np.abs(R), np.abs(X), np.abs(Y))
    return want, bound


def run(ctx, R, X, Y=None, scale=1.0, pad=(0, 0, 0)):
    """Uploads row-major operands (optionally as views of wider buffers) and calls the C ABI."""
    import torch

    def up(a, extra):
        wide = np.zeros((a.shape[0], a.shape[1] + extra), np.float32)
        wide[:, :a.shape[1]] = a
        wide[:, a.shape[1]:] = 7.0                 # must never be read as data
        return ctx.to_device(wide)[:, :a.shape[1]]

    r, x = up(R, pad[0]), up(X, pad[1])
    y = x if Y is None else up(Y, pad[2])
    K, D, E = R.shape[1], X.shape[1], (X if Y is None else Y).shape[1]
    out = torch.full((K, D, E), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_weighted_outer", r, R.shape[1] + pad[0], x, X.shape[1] + pad[1], y,
             (X if Y is None else Y).shape[1] + (pad[1] if Y is None else pad[2]), R.shape[0], K, D, E,
             float(scale), out)
    ctx.sync()
    return out.cpu().numpy()


@pytest.mark.parametrize("N,K,D", [(1000, 64, 16), (4097, 32, 8), (33, 20, 32), (1, 4, 4),
                                   (70000, 64, 16), (5000, 8, 28)])
def test_symmetric_second_moment(ctx, N, K, D):
    R = RNG.dirichlet(np.ones(K), N).astype(np.float32)
    X = RNG.standard_normal((N, D)).astype(np.float32)
    got = run(ctx, R, X, scale=0.5)
    want, bound = reference(R, X, X, 0.5)
    assert (np.abs(got - want) <= 2e-5 * bound + 1e-30).all()
    # mirrored halves are the same bits
    np.testing.assert_array_equal(got, got.transpose(0, 2, 1))


@pytest.mark.parametrize("N,K,D,E,pad", [(777, 64, 16, 12, (0, 0, 0)), (2000, 12, 32, 32, (4, 8, 12)),
                                         (100, 36, 4, 24, (28, 4, 0))])
def test_two_different_factors_and_padded_rows(ctx, N, K, D, E, pad):
    R = RNG.standard_normal((N, K)).astype(np.float32)
    X = RNG.standard_normal((N, D)).astype(np.float32)
    Y = RNG.standard_normal((N, E)).astype(np.float32)
    got = run(ctx, R, X, Y, scale=-2.0, pad=pad)
    want, bound = reference(R, X, Y, -2.0)
    assert (np.abs(got - want) <= 2e-5 * bound).all()


def test_empty_batch_is_zero_and_reruns_are_identical(ctx):
    R = RNG.standard_normal((0, 8)).astype(np.float32)
    X = RNG.standard_normal((0, 4)).astype(np.float32)
    assert (run(ctx, R, X) == 0).all()
    R = RNG.standard_normal((30000, 64)).astype(np.float32)
    X = RNG.standard_normal((30000, 16)).astype(np.float32)
    np.testing.assert_array_equal(run(ctx, R, X), run(ctx, R, X))


def test_limits_are_reported_not_guessed(ctx):
    from bayesic_amd._ffi import BayesicHipError
    R = RNG.standard_normal((10, 68)).astype(np.float32)       # K > 64
    X = RNG.standard_normal((10, 8)).astype(np.float32)
    with pytest.raises(BayesicHipError):
        run(ctx, R, X)
    with pytest.raises(BayesicHipError):
        run(ctx, R[:, :6], X)                                   # K % 4 != 0


def test_executor_takes_the_one_pass_route(ctx):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from test_fusion_gpu import Counting
    dev = DeviceBackend(ctx)
    R, X = var("R", ndim=2), var("X", ndim=2)
    second = sum(dimshuffle(R, 0, 1, "x", "x") * dimshuffle(X, 0, "x", 1, "x") *
                 dimshuffle(X, 0, "x", "x", 1), axis=0)
    assert repr(second) == "einsum(out_uvw = sum_i R_iu X_iv X_iw)"
    f = (3 * second).compile(dev)
    N, K, D = 20000, 64, 16
    R_ = RNG.dirichlet(np.ones(K), N).astype(np.float32)
    X_ = RNG.standard_normal((N, D)).astype(np.float32)
    with Counting(ctx) as c:
        got = f(R=R_, X=X_)
    # one pass for the contraction; the constant factor scales the small K x D x D result
    assert c.count("bsc_weighted_outer") == 1 and c.count("bsc_gemm_strided_batched") == 0 \
        and c.count("bsc_map_reduce") <= 1, c.calls
    want, bound = reference(R_, X_, X_, 3.0)
    assert got.shape == (K, D, D)
    assert (np.abs(got - want) <= 2e-5 * bound).all()

    # the wide factor on the other side: out[d, k, e] -- same kernel, permuted view
    g = sum(dimshuffle(X, 0, 1, "x", "x") * dimshuffle(R, 0, "x", 1, "x") *
            dimshuffle(X, 0, "x", "x", 1), axis=0).compile(dev)
    with Counting(ctx) as c:
        got = g(R=R_, X=X_)
    assert c.count("bsc_weighted_outer") == 1, c.calls
    np.testing.assert_allclose(got, reference(R_, X_, X_)[0].transpose(1, 0, 2), rtol=0,
                               atol=2e-5 * reference(R_, X_, X_)[1].max())

    # outside the kernel's limits the general route still answers (K = 6 is not a multiple of 4)
    h = second.compile(dev)
    with Counting(ctx) as c:
        got = h(R=R_[:500, :6].copy(), X=X_[:500])
    assert c.count("bsc_weighted_outer") == 0
    want, bound = reference(R_[:500, :6], X_[:500], X_[:500])
    assert (np.abs(got - want) <= 2e-5 * bound + 1e-6).all()
