"""Reverse-mode derivatives of algebra expressions (bayesic_amd/algebra/autodiff.py): against
central finite differences on the float64 oracle backend for seeded random trees (the generator
of test_fuzz_gpu.py), and the device backend's gradients against the float64 ones."""
import builtins

import numpy as np
import pytest

import test_fuzz_gpu as F
from bayesic_amd import algebra as A
from bayesic_amd.algebra.autodiff import value_and_grad
from oracle.einsum_eval import NumpyBackend

B64 = NumpyBackend(np.float64)


def _tree(seed):
    made = F.Grower(seed).grow(2 + seed % 9)
    if made is None:
        pytest.skip("generator produced nothing for this seed")
    expr = made[0]
    names = sorted(expr.input_types)
    vals = {n: F.inputs()[n].astype(np.float64) for n in names}
    return expr, names, vals


def _grad64(expr, names, vals):
    out, grads = value_and_grad(B64, expr, {n: B64.from_host(v, "float64", v.ndim) for n, v in vals.items()}, names)
    return np.asarray(out, np.float64), {n: np.asarray(g, np.float64) for n, g in grads.items()}


@pytest.mark.parametrize("seed", range(120))
def test_gradient_of_random_tree_matches_central_differences(seed):
    expr, names, vals = _tree(seed)
    _, grads = _grad64(expr, names, vals)
    f = expr.compile(B64)
    rs = np.random.RandomState(seed)
    for n in names:
        assert grads[n].shape == vals[n].shape
        for _ in range(3):
            idx = tuple(rs.randint(d) for d in vals[n].shape)
            h = 1e-6
            up = {k: v.copy() for k, v in vals.items()}
            dn = {k: v.copy() for k, v in vals.items()}
            up[n][idx] += h
            dn[n][idx] -= h
            fd = (np.sum(f(**up)) - np.sum(f(**dn))) / (2 * h)
            assert np.isclose(grads[n][idx], fd, rtol=2e-5, atol=2e-6 * builtins.max(1.0, abs(fd))), \
                (repr(expr), n, idx, grads[n][idx], fd)


def test_known_derivatives():
    X, w = A.var("X", 2, "float64"), A.var("w", 1, "float64")
    Xs, ws = np.arange(6.0).reshape(3, 2) / 3 + 0.1, np.array([0.5, -1.5])
    # d/dw sum(exp(X w)) = X^T exp(X w);  d/dX = outer(exp(X w), w)
    out, g = _grad64(A.sum(A.exp(A.dot(X, w))), ["X", "w"], dict(X=Xs, w=ws))
    e = np.exp(Xs @ ws)
    np.testing.assert_allclose(out, e.sum(), rtol=1e-12)
    np.testing.assert_allclose(g["w"], Xs.T @ e, rtol=1e-12)
    np.testing.assert_allclose(g["X"], np.outer(e, ws), rtol=1e-12)
    # trace(Q Q^T) -> 2 Q ; log-sum of a diagonal
    Q = A.var("Q", 2, "float64")
    Qs = np.arange(9.0).reshape(3, 3) / 4 + 0.2
    _, g = _grad64(A.trace(A.dot(Q, Q.T)), ["Q"], dict(Q=Qs))
    np.testing.assert_allclose(g["Q"], 2 * Qs, rtol=1e-12)
    _, g = _grad64(A.sum(A.log(A.diagonal(Q))), ["Q"], dict(Q=Qs))
    np.testing.assert_allclose(g["Q"], np.diag(1.0 / np.diag(Qs)), rtol=1e-12)
    with pytest.raises(KeyError):
        _grad64(A.sum(X), ["X", "w"], dict(X=Xs, w=ws))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(60))
def test_device_gradient_matches_float64(ctx, seed):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    dev = DeviceBackend(ctx)
    expr, names, vals = _tree(seed)
    _, want = _grad64(expr, names, vals)
    v32 = {n: v.astype(np.float32) for n, v in vals.items()}
    types = expr.input_types
    inputs = {n: dev.from_host(v32[n], *types[n]) for n in names}
    _, grads = value_and_grad(dev, expr, inputs, names)
    for n in names:
        got = np.asarray(dev.to_host(grads[n]), np.float64)
        assert got.shape == want[n].shape, (repr(expr), n)
        scale = builtins.max(float(np.abs(want[n]).max()), 1e-3)
        assert np.abs(got - want[n]).max() <= 3e-4 * scale, (repr(expr), n, np.abs(got - want[n]).max(), scale)


def test_sum_vjp_broadcasts_from_shape_not_from_the_forward_value():
    """d sum(exp(x)) / dx at x = [0, 800]: exp(800) is inf in float64, the gradient is [1, inf].
    Broadcasting the adjoint as g + 0 * value made it [1, nan] (0 * inf)."""
    import numpy as np
    from bayesic_amd import algebra as A
    from bayesic_amd.algebra.autodiff import value_and_grad
    from oracle.einsum_eval import NumpyBackend
    be = NumpyBackend(np.float64)
    x = A.var("x", 1)
    with np.errstate(over="ignore"):
        _, grads = value_and_grad(be, A.sum(A.exp(x)), {"x": np.array([0.0, 800.0])}, ["x"])
    g = np.asarray(grads["x"])
    assert g[0] == 1.0 and np.isposinf(g[1])


def test_no_adjoint_is_formed_towards_inputs_that_are_not_differentiated():
    """d/dW of sum((y - dot(W, X.T))^2): the vector-Jacobian product towards the DATA operand X
    (an N x D product) must never be computed -- counted on a backend that records tensordots."""
    import numpy as np
    from bayesic_amd import algebra as A
    from bayesic_amd.algebra.autodiff import value_and_grad
    from oracle.einsum_eval import NumpyBackend

    class Counting(NumpyBackend):
        shapes = []

        def tensordot(self, x, y, *a):
            out = NumpyBackend.tensordot(self, x, y, *a)
            Counting.shapes.append(np.shape(out))
            return out

    be = Counting(np.float64)
    rs = np.random.RandomState(0)
    S, N, D = 3, 50, 7
    Xv, yv, Wv = rs.standard_normal((N, D)), rs.standard_normal(N), rs.standard_normal((S, D))
    X, y, W = A.var("X", 2, "float64"), A.var("y", 1, "float64"), A.var("W", 2, "float64")
    r = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
    f = A.sum(r * r, axis=1)
    _, g = value_and_grad(be, f, {"X": Xv, "y": yv, "W": Wv}, ["W"])
    R = yv[None, :] - Wv @ Xv.T
    np.testing.assert_allclose(np.asarray(g["W"]), -2.0 * R @ Xv, rtol=1e-12)
    assert (N, D) not in Counting.shapes and (D, N) not in Counting.shapes, Counting.shapes
