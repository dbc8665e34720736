"""GPU parity of the fused map(+reduce) kernel (bsc_map_reduce) and of the
executor's deferred element-wise values against numpy,
plus the launch counts that show the fusion really happens: a chain
elemwise -> _mul -> _sum must be ONE launch that writes only its result.

Tolerances: rtol 1e-5 wherever a sum is involved (the reference's own,
bayesic/tests/test_algebra.py:82), 2e-6 for device libm vs numpy otherwise.
"""
import ctypes

import numpy as np
import numpy.testing as npt
import pytest

from bayesic_amd.algebra import *            # noqa: F401,F403
import builtins

pytestmark = pytest.mark.gpu

OPS = {"add": 0, "mul": 1, "log": 2, "exp": 3, "pow": 4, "abs_": 5, "copy": 6}
NP_UNARY = {"copy": lambda x, a: x, "log": lambda x, a: np.log(x), "exp": lambda x, a: np.exp(x),
            "abs_": lambda x, a: np.abs(x), "pow": lambda x, a: np.power(x, a)}


@pytest.fixture(scope="module")
def dev(ctx):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    return DeviceBackend(ctx)


def _i64(v):
    v = list(v)
    return (ctypes.c_int64 * builtins.max(len(v), 1))(*v)


def map_reduce(ctx, arrays, pre, combine, red_axes, scale=1.0, shift=0.0, post=("copy", 0.0),
               views=None):
    """Direct C-ABI call.  arrays: numpy arrays of one rank (extent 1 broadcasts);
    views[i]: optional function applied to the uploaded tensor (strided views)."""
    import torch
    dtype = arrays[0].dtype
    rank = arrays[0].ndim
    shape = [builtins.max(a.shape[ax] for a in arrays) for ax in range(rank)]
    tens = []
    for i, a in enumerate(arrays):
        t = ctx.to_device(np.ascontiguousarray(a) if views is None or views[i] is None else a)
        tens.append(t)
    red = sorted(red_axes)
    keep = [a for a in range(rank) if a not in red]
    out = torch.empty([shape[a] for a in keep], dtype=tens[0].dtype, device=ctx.device)

    def strides(t, axes):
        return [0 if (t.shape[ax] == 1 and shape[ax] != 1) else t.stride(ax) for ax in axes]

    ks, rs = [], []
    for t in tens:
        ks += strides(t, keep)
        rs += strides(t, red)
    n = len(tens)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tens])
    pre_ops = (ctypes.c_int32 * n)(*[OPS[p[0]] for p in pre])
    pre_args = (ctypes.c_double * n)(*[float(p[1]) for p in pre])
    ctx.call("bsc_map_reduce", 0 if dtype == np.float32 else 1, OPS[combine], len(keep),
             _i64(shape[a] for a in keep), len(red), _i64(shape[a] for a in red), n, ptrs,
             _i64(ks), _i64(rs), pre_ops, pre_args, float(scale), float(shift), OPS[post[0]],
             float(post[1]), out, _i64(out.stride()))
    ctx.sync()
    return out.cpu().numpy()


def expected(arrays, pre, combine, red_axes, scale=1.0, shift=0.0, post=("copy", 0.0)):
    vals = [NP_UNARY[p[0]](a.astype(np.float64), p[1]) for a, p in zip(arrays, pre)]
    v = vals[0]
    for u in vals[1:]:
        v = v * u if combine == "mul" else v + u
    v = NP_UNARY[post[0]](scale * v + shift, post[1])
    v = np.broadcast_to(v, np.broadcast_shapes(*[a.shape for a in arrays]))
    return v.sum(axis=tuple(red_axes)) if red_axes else v


RNG = np.random.RandomState(11)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,red", [
    ((1024,), ()),                # dense map, 16-byte path
    ((37, 53), ()),               # ragged map
    ((64, 4096), (1,)),           # reduce the fast axis (wave variant, dense)
    ((64, 4099), (1,)),           # ragged reduce length (wave variant, scalar loads)
    ((4096, 96), (0,)),           # reduce the slow axis (lane variant)
    ((8, 300000), (1,)),          # few outputs: split reduction
    ((300000, 8), (0,)),          # few outputs, lane variant (below 16 outputs -> wave)
    ((20, 30, 40), (0, 2)),       # two reduced axes around a kept one
    ((6, 5, 4, 3, 2, 7), (1, 3, 5)),
    ((0, 5), (0,)),               # empty reduction -> zeros
])
def test_map_reduce_matches_numpy(ctx, dtype, shape, red):
    a = (RNG.rand(*shape) + 0.5).astype(dtype)
    b = RNG.standard_normal(shape).astype(dtype)
    c = (RNG.rand(*[1 if i % 2 else s for i, s in enumerate(shape)]) + 0.5).astype(dtype)  # broadcasts
    pre = [("log", 0), ("exp", 0), ("pow", -1.0)]
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == np.float32 else dict(rtol=1e-12, atol=1e-12)
    for combine in ("mul", "add"):
        got = map_reduce(ctx, [a, b, c], pre, combine, red, scale=0.5, shift=0.25, post=("abs_", 0))
        want = expected([a, b, c], pre, combine, red, scale=0.5, shift=0.25, post=("abs_", 0))
        assert got.shape == want.shape
        npt.assert_allclose(got, want, **tol)


def test_map_reduce_strided_views_and_determinism(ctx):
    import torch
    a = RNG.standard_normal((300, 200)).astype(np.float32)
    ta = ctx.to_device(a)
    out1 = torch.empty(200, dtype=torch.float32, device=ctx.device)
    out2 = torch.empty(200, dtype=torch.float32, device=ctx.device)
    tt = ta.t()                                   # [200, 300] view, reduce axis 1 has stride 200
    for out in (out1, out2):
        ptrs = (ctypes.c_void_p * 2)(tt.data_ptr(), tt.data_ptr())
        ctx.call("bsc_map_reduce", 0, OPS["mul"], 1, _i64([200]), 1, _i64([300]), 2, ptrs,
                 _i64([tt.stride(0)] * 2), _i64([tt.stride(1)] * 2),
                 (ctypes.c_int32 * 2)(OPS["copy"], OPS["copy"]), (ctypes.c_double * 2)(0, 0),
                 1.0, 0.0, OPS["copy"], 0.0, out, _i64([1]))
    ctx.sync()
    npt.assert_array_equal(out1.cpu().numpy(), out2.cpu().numpy())        # fixed order
    npt.assert_allclose(out1.cpu().numpy(), (a.astype(np.float64) ** 2).sum(axis=0), rtol=1e-5)


def test_map_reduce_rejects_bad_arguments(ctx):
    from bayesic_amd._ffi import BayesicHipError
    import torch
    t = torch.zeros(8, device=ctx.device)
    args = lambda combine, pre, post: (
        "bsc_map_reduce", 0, combine, 1, _i64([8]), 0, _i64([]), 1,
        (ctypes.c_void_p * 1)(t.data_ptr()), _i64([1]), _i64([]), (ctypes.c_int32 * 1)(pre),
        (ctypes.c_double * 1)(0.0), 1.0, 0.0, post, 0.0, t, _i64([1]))
    with pytest.raises(BayesicHipError):
        ctx.call(*args(OPS["log"], OPS["copy"], OPS["copy"]))     # combine must be add / mul
    with pytest.raises(BayesicHipError):
        ctx.call(*args(OPS["mul"], OPS["add"], OPS["copy"]))      # pre op must be unary
    with pytest.raises(BayesicHipError):
        ctx.call(*args(OPS["mul"], OPS["copy"], OPS["mul"]))      # post op must be unary


# ---- executor level --------------------------------------------------------------

class Counting(object):
    """Counts C-ABI calls issued through a context while an expression runs."""

    def __init__(self, ctx):
        self.ctx, self.calls = ctx, []

    def __enter__(self):
        self._orig = self.ctx.call
        def call(name, *a):
            self.calls.append(name)
            return self._orig(name, *a)
        self.ctx.call = call
        return self

    def __exit__(self, *exc):
        self.ctx.call = self._orig

    def count(self, name):
        return self.calls.count(name)


def test_chain_into_sum_is_one_launch(dev):
    X, Y = var("X", ndim=2), var("Y", ndim=2)
    X_ = RNG.standard_normal((700, 300)).astype(np.float32)
    Y_ = RNG.standard_normal((700, 300)).astype(np.float32)
    f = sum(exp(X) * Y).compile(dev)              # lowered: _tensordot(exp(X), Y, [0,1], [0,1])
    with Counting(dev.ctx) as c:
        got = f(X=X_, Y=Y_)
    assert c.count("bsc_map_reduce") == 1
    assert c.count("bsc_elemwise") == 0 and c.count("bsc_gemm_strided_batched") == 0
    npt.assert_allclose(got, (np.exp(X_.astype(np.float64)) * Y_).sum(), rtol=1e-5)

    g = sum(exp(X) * Y, axis=0).compile(dev)      # -> a batched full contraction or _sum(_mul)
    with Counting(dev.ctx) as c:
        got = g(X=X_, Y=Y_)
    assert c.count("bsc_map_reduce") == 1 and c.count("bsc_elemwise") == 0
    npt.assert_allclose(got, (np.exp(X_.astype(np.float64)) * Y_).sum(axis=0), rtol=1e-5, atol=1e-4)


def test_elementwise_trees_fuse_and_match_numpy(dev):
    X, Y, Z, x = var("X", ndim=2), var("Y", ndim=2), var("Z", ndim=2), var("x", ndim=1)
    X_ = (RNG.rand(64, 48) + 0.5).astype(np.float32)
    Y_ = RNG.standard_normal((64, 48)).astype(np.float32)
    Z_ = (RNG.rand(64, 48) + 0.5).astype(np.float32)
    x_ = (RNG.rand(48) + 0.5).astype(np.float32)
    cases = [
        (X / Y, X_ / Y_, 1),                                        # mul(X, pow(Y,-1))
        (exp(X * Y) * Z, np.exp(X_ * Y_) * Z_, 2),                  # post op, then a second launch
        (log(abs(Y) + 1), np.log(np.abs(Y_) + 1), 1),
        (2 * exp(X) * log(Z) * 3, 2 * np.exp(X_) * np.log(Z_) * 3, 1),
        (X * dimshuffle(log(x), "x", 0), X_ * np.log(x_)[None, :], 1),   # a view of a deferred value
        ((X + Y) * Z, (X_ + Y_) * Z_, 2),
        (exp(X).T * Y.T, np.exp(X_).T * Y_.T, 1),
        (X + Y + Z + 1, X_ + Y_ + Z_ + 1, 1),
        (X - 2 * Y, X_ - 2 * Y_, 1),            # the coefficient rides on the operand (BSC_OP_SCALE)
    ]
    for expr, want, launches in cases:
        f = expr.compile(dev)
        with Counting(dev.ctx) as c:
            got = f(**{n: {"X": X_, "Y": Y_, "Z": Z_, "x": x_}[n] for n in expr.input_types})
        npt.assert_allclose(got, want, rtol=3e-6, atol=1e-6, err_msg=repr(expr))
        assert c.count("bsc_map_reduce") + c.count("bsc_elemwise") == launches, (repr(expr), c.calls)


def test_more_than_eight_factors(dev):
    vs = [var("v%d" % i, ndim=1) for i in range(11)]
    vals = {"v%d" % i: (RNG.rand(100) + 0.5).astype(np.float32) for i in range(11)}
    e = vs[0]
    for v in vs[1:]:
        e = e * v
    want = np.prod([vals["v%d" % i].astype(np.float64) for i in range(11)], axis=0)
    npt.assert_allclose(e.compile(dev)(**vals), want, rtol=1e-5)


def test_batched_dot_products_take_the_fused_path(dev):
    S1, S2 = var("S1", ndim=3), var("S2", ndim=3)
    A = RNG.standard_normal((7, 50, 30)).astype(np.float32)
    B = RNG.standard_normal((7, 50, 30)).astype(np.float32)
    idx = [("out", 0), ("sum", 0), ("sum", 1)]
    e = einsum([(S1, idx), (S2, idx)], 1)                  # out_b = sum_ij S1_bij S2_bij
    f = e.compile(dev)
    with Counting(dev.ctx) as c:
        got = f(S1=A, S2=B)
    assert c.count("bsc_map_reduce") == 1 and c.count("bsc_gemm_strided_batched") == 0
    npt.assert_allclose(got, np.einsum("bij,bij->b", A.astype(np.float64), B), rtol=1e-5, atol=1e-4)


def test_lda_statistic_has_no_data_sized_elementwise_intermediate(dev):
    Th, Bm, C = var("Th", ndim=2), var("Bm", ndim=2), var("C", ndim=2)
    docs, V, K = 96, 640, 16
    Th_ = (RNG.rand(docs, K) + 0.1).astype(np.float32)
    Bm_ = (RNG.rand(K, V) + 0.1).astype(np.float32)
    C_ = RNG.poisson(0.3, (docs, V)).astype(np.float32)
    e = Bm * dot(Th.T, C / dot(Th, Bm))
    f = e.compile(dev)
    with Counting(dev.ctx) as c:
        got = f(Th=Th_, Bm=Bm_, C=C_)
    # two launches in all: C / dot(Th, Bm) and Bm * dot(Th.T, .) each fold into their GEMM's store
    # (bsc_gemm_epilogue: power -1 with E = C, power 1 with E = Bm) -- no element-wise launch at all
    assert c.count("bsc_gemm_epilogue") == 2 and c.count("bsc_gemm_strided_batched") == 0
    assert c.count("bsc_map_reduce") == 0 and c.count("bsc_elemwise") == 0
    want = Bm_ * (Th_.T.astype(np.float64) @ (C_ / (Th_.astype(np.float64) @ Bm_)))
    npt.assert_allclose(got, want, rtol=2e-5)
    # with fusion off: the same value from four launches
    from bayesic_amd.algebra.device_backend import DeviceBackend
    npt.assert_allclose(e.compile(DeviceBackend(dev.ctx, fuse=False))(Th=Th_, Bm=Bm_, C=C_), want, rtol=2e-5)


@pytest.mark.parametrize("M,N,K", [(130, 257, 40), (64, 64, 3000), (5, 300, 700), (256, 256, 40000), (2, 2, 5)])
def test_gemm_consumers_fold_into_the_store(dev, M, N, K):
    """B * dot(X, Y), dot(X, Y) * 2.5, C / dot(X, Y), row / column broadcast factors, split-K
    shapes (the epilogue then runs in the split-K reduce) -- against float64 numpy."""
    X, Y, E, r, c = var("X", ndim=2), var("Y", ndim=2), var("E", ndim=2), var("r", ndim=1), var("c", ndim=1)
    X_ = (RNG.rand(M, K) + 0.1).astype(np.float32)
    Y_ = (RNG.rand(K, N) + 0.1).astype(np.float32)
    E_ = (RNG.rand(M, N) + 0.5).astype(np.float32)
    r_, c_ = (RNG.rand(N) + 0.5).astype(np.float32), (RNG.rand(M) + 0.5).astype(np.float32)
    P = X_.astype(np.float64) @ Y_.astype(np.float64)
    cases = [(E * dot(X, Y), E_ * P), (dot(X, Y) * 2.5, 2.5 * P), (E / dot(X, Y), E_ / P),
             (dot(X, Y) * dimshuffle(r, "x", 0), P * r_[None, :]),
             (dimshuffle(c, 0, "x") / dot(X, Y) * 3.0, 3.0 * c_[:, None] / P),
             (exp(E) * dot(X, Y), np.exp(E_.astype(np.float64)) * P),          # a deferred factor is computed, then folded
             (E * dot(X, Y) * E, E_ * P * E_)]                                    # two factors: not folded, still right
    vals = dict(X=X_, Y=Y_, E=E_, r=r_, c=c_)
    for expr, want in cases:
        got = expr.compile(dev)(**{n: vals[n] for n in expr.input_types})
        npt.assert_allclose(got, want, rtol=3e-5, err_msg=repr(expr))
    with Counting(dev.ctx) as cnt:
        (E / dot(X, Y)).compile(dev)(X=X_, Y=Y_, E=E_)
    assert cnt.count("bsc_gemm_epilogue") == 1 and cnt.count("bsc_map_reduce") == 0


def test_memory_plan_reuses_intermediates_but_never_the_result(dev):
    """Intermediates of a compiled expression are allocated once; the result always
    gets storage of its own (a second call must not overwrite the first result),
    and a change of input shapes re-plans."""
    X, Y = var("X", ndim=2), var("Y", ndim=2)
    e = dot(exp(X), Y.T) * 2 + 1                   # exp(X) is forced (GEMM operand), result is fused
    f = e.compile(dev).device_fn
    rs = np.random.RandomState(3)
    outs, wants = [], []
    for shape in [(40, 30), (40, 30), (17, 9), (40, 30)]:
        X_ = rs.standard_normal(shape).astype(np.float32)
        Y_ = rs.standard_normal(shape).astype(np.float32)
        outs.append(f(X=dev.from_host(X_, "float32", 2), Y=dev.from_host(Y_, "float32", 2)))
        wants.append(np.exp(X_.astype(np.float64)) @ Y_.T * 2 + 1)
    dev.ctx.sync()
    ptrs = {o.untyped_storage().data_ptr() for o in outs}
    assert len(ptrs) == len(outs)                  # four live results, four storages
    for o, w in zip(outs, wants):
        npt.assert_allclose(o.cpu().numpy(), w, rtol=2e-5, atol=1e-5)
    plan = dev._plans[id(e)][1]
    live = [b for b in plan if b is not None]
    assert len(live) >= 1                          # the forced exp(X) stays in the plan
    before = [b.data_ptr() for b in live]
    X_ = rs.standard_normal((40, 30)).astype(np.float32)
    f(X=dev.from_host(X_, "float32", 2), Y=dev.from_host(X_, "float32", 2))
    assert [b.data_ptr() for b in dev._plans[id(e)][1] if b is not None][:len(before)] == before


@pytest.mark.parametrize("R,C", [(1000, 256), (37, 64), (5, 4096), (4097, 128)])
def test_row_and_column_broadcasts_take_the_vector_paths(dev, R, C):
    """dimshuffle('x', ...) broadcasts -- column scalings, row weights, biases -- with and
    without a trailing sum, against numpy (these shapes run on the 16-byte kernels with one
    operand constant along the fast axis)."""
    X, v, u = var("X", ndim=2), var("v", ndim=1), var("u", ndim=1)
    X_ = RNG.standard_normal((R, C)).astype(np.float32)
    v_ = (RNG.rand(C) + 0.5).astype(np.float32)
    u_ = (RNG.rand(R) + 0.5).astype(np.float32)
    vals = {"X": X_, "v": v_, "u": u_}
    X64 = X_.astype(np.float64)
    cases = [
        (X * dimshuffle(v, "x", 0), X64 * v_[None, :]),
        (X + dimshuffle(u, 0, "x"), X64 + u_[:, None]),
        (exp(X) * dimshuffle(u, 0, "x") * dimshuffle(log(v), "x", 0), np.exp(X64) * u_[:, None] * np.log(v_)[None, :]),
        (sum(exp(X) * dimshuffle(u, 0, "x"), axis=0), (np.exp(X64) * u_[:, None]).sum(0)),
        (sum(X * dimshuffle(v, "x", 0), axis=0), (X64 * v_[None, :]).sum(0)),
        (sum(abs(X) * dimshuffle(u, 0, "x"), axis=1), (np.abs(X64) * u_[:, None]).sum(1)),
        (sum(X * X * dimshuffle(v, "x", 0), axis=1), (X64 * X64 * v_[None, :]).sum(1)),
    ]
    for expr, want in cases:
        got = expr.compile(dev)(**{n: vals[n] for n in expr.input_types})
        assert got.shape == want.shape, repr(expr)
        npt.assert_allclose(got, want, rtol=2e-5, atol=1e-4, err_msg=repr(expr))


def test_short_and_wide_matrices_are_split_along_the_columns(dev):
    """[8, 1M]-shaped per-sample values (the general VI engines make them): the row/column
    broadcast kernel must also spread the columns over the machine -- a wave per row walked
    32 MB with eight waves (10 ms).  Correctness of the chunked column ranges, ragged tail included."""
    Xv, v, u = var("X", ndim=2), var("v", ndim=1), var("u", ndim=1)
    for R, C in [(3, 100_004), (8, 262_144), (1, 70_000), (17, 9_996)]:
        X_ = RNG.standard_normal((R, C)).astype(np.float32)
        v_ = RNG.standard_normal(C).astype(np.float32)
        u_ = RNG.standard_normal(R).astype(np.float32)
        got = (dimshuffle(v, "x", 0) - Xv).compile(dev)(X=X_, v=v_)
        npt.assert_array_equal(got, v_[None, :] - X_)
        got = (Xv * dimshuffle(u, 0, "x") + 1).compile(dev)(X=X_, u=u_)
        npt.assert_allclose(got, X_ * u_[:, None] + 1, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("R,C", [(100_003, 64), (999, 16), (4097, 4), (50, 252), (3000, 1024)])
def test_narrow_rows_with_several_row_vector_operands(dev, R, C):
    """The logits of a mixture's assignments: two dense [N, K] products plus three K-vectors
    repeated down the rows and a scale -- one launch of the flat dense kernel with periodic
    operands (the row kernel gives a whole wave to one narrow row and takes at most three
    operands; this used to fall to the generic strided kernel at 0.27 TB/s)."""
    P, Q = var("P", ndim=2), var("Q", ndim=2)
    a, b, c = var("a", ndim=1), var("b", ndim=1), var("c", ndim=1)
    P_ = RNG.standard_normal((R, C)).astype(np.float32)
    Q_ = RNG.standard_normal((R, C)).astype(np.float32)
    a_, b_, c_ = (RNG.standard_normal(C).astype(np.float32) for _ in range(3))
    row = lambda v: dimshuffle(v, "x", 0)
    e = (P + Q * (-0.5) + row(a) * 0.5 + row(b) * (-0.5) + row(log(exp(c)))) * 0.1
    want = (P_.astype(np.float64) - 0.5 * Q_ + 0.5 * a_[None, :] - 0.5 * b_[None, :] + c_[None, :]) * 0.1
    got = e.compile(dev)(P=P_, Q=Q_, a=a_, b=b_, c=c_)
    npt.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    e2 = P * row(a) * row(b) * Q
    npt.assert_allclose(e2.compile(dev)(P=P_, Q=Q_, a=a_, b=b_), P_.astype(np.float64) * a_ * b_ * Q_, rtol=2e-5,
                        atol=1e-6)


def test_values_of_constant_data_are_computed_once(ctx):
    """DeviceBackend.mark_constant: an element-wise value of marked inputs only (x * x feeding a
    contraction) is launched once and reused by later evaluations -- of this and of other expressions --
    until forget_constants(); unmarked operands are recomputed as before."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    be = DeviceBackend(ctx)
    rs = np.random.RandomState(5)
    X_ = rs.standard_normal((3000, 12)).astype(np.float32)
    R_ = rs.rand(3000, 7).astype(np.float32)
    T_ = rs.rand(7, 12).astype(np.float32)
    X, R, Tm = var("X", 2), var("R", 2), var("Tm", 2)
    stat = dot(R.T, X * X)                      # sum_n r_nk x_nd^2
    logit = dot(X * X, Tm.T)                    # another expression with the same constant sub-value
    Xd, Rd, Td = (be.from_host(a, "float32", 2) for a in (X_, R_, T_))
    f, h = stat.compile(be).device_fn, logit.compile(be).device_fn
    with Counting(ctx) as c:
        a0 = be.to_host(f(R=Rd, X=Xd))
    assert c.count("bsc_map_reduce") == 1       # unmarked: x * x is a launch of every evaluation
    be.mark_constant(Xd)
    with Counting(ctx) as c:
        a1 = be.to_host(f(R=Rd, X=Xd))
        a2 = be.to_host(f(R=Rd, X=Xd))
        b1 = be.to_host(h(X=Xd, Tm=Td))
    assert c.count("bsc_map_reduce") == 1       # once for all three
    npt.assert_array_equal(a0, a1)
    npt.assert_array_equal(a1, a2)
    x64 = X_.astype(np.float64)
    npt.assert_allclose(a1, R_.astype(np.float64).T @ (x64 * x64), rtol=2e-5)
    npt.assert_allclose(b1, (x64 * x64) @ T_.astype(np.float64).T, rtol=2e-5)
    be.forget_constants()
    with Counting(ctx) as c:
        be.to_host(f(R=Rd, X=Xd))
    assert c.count("bsc_map_reduce") == 1


@pytest.mark.parametrize("n_rows,d,k", [(3000, 12, 7), (4097, 16, 64), (5, 3, 2)])
def test_sums_of_products_over_constant_rows_become_one_product(ctx, n_rows, d, k):
    """sum_i s_i dot(X_i, Y_i) + row vectors + a scalar with every X_i CONSTANT (a model's data and
    cached element-wise values of it: the logits of an exponential-family mixture) is ONE product over
    the concatenated contraction [X_1 | X_2 | 1] . [s_1 Y_1 ; s_2 Y_2 ; bias] -- the wide left operand
    built once, no [M, N] intermediate per product and no n-ary add over them.  With unmarked data the
    expression runs as before; both agree with float64."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    be = DeviceBackend(ctx)
    rs = np.random.RandomState(n_rows)
    X_ = rs.standard_normal((n_rows, d)).astype(np.float32)
    A_ = rs.standard_normal((k, d)).astype(np.float32)
    B_ = rs.rand(k, d).astype(np.float32)
    c_ = rs.standard_normal(k).astype(np.float32)
    e_ = rs.standard_normal(k).astype(np.float32)
    X, Am, Bm, cv, ev = var("X", 2), var("Am", 2), var("Bm", 2), var("cv", 1), var("ev", 1)
    logits = dot(X, Am.T) + dot(X * X, Bm.T) * (-0.5) + dimshuffle(cv, "x", 0) * 0.5 \
        + dimshuffle(ev, "x", 0) + 0.25
    f = logits.compile(be).device_fn
    Xd, Ad, Bd = (be.from_host(a, "float32", 2) for a in (X_, A_, B_))
    cd, ed = (be.from_host(a, "float32", 1) for a in (c_, e_))
    x64 = X_.astype(np.float64)
    want = x64 @ A_.astype(np.float64).T - 0.5 * (x64 * x64) @ B_.astype(np.float64).T \
        + 0.5 * c_[None, :] + e_[None, :] + 0.25
    tol = dict(rtol=2e-5, atol=2e-5 * np.abs(want).max())
    with Counting(ctx) as c:
        plain = be.to_host(f(X=Xd, Am=Ad, Bm=Bd, cv=cd, ev=ed))
    # (unmarked X: the X * X of the second product may be formed inside it -- bsc_gemm_fused)
    assert c.count("bsc_gemm_strided_batched") + c.count("bsc_gemm_epilogue") + c.count("bsc_gemm_fused") >= 2
    npt.assert_allclose(plain, want, **tol)
    be.mark_constant(Xd)
    be.to_host(f(X=Xd, Am=Ad, Bm=Bd, cv=cd, ev=ed))          # builds [X | X^2 | 1] once
    with Counting(ctx) as c:
        fused = be.to_host(f(X=Xd, Am=Ad, Bm=Bd, cv=cd, ev=ed))
    assert c.count("bsc_gemm_strided_batched") + c.count("bsc_gemm_epilogue") == 1
    npt.assert_allclose(fused, want, **tol)
    # other right-hand sides reuse the wide operand; the result follows them
    A2 = be.from_host((2.0 * A_).astype(np.float32), "float32", 2)
    again = be.to_host(f(X=Xd, Am=A2, Bm=Bd, cv=cd, ev=ed))
    npt.assert_allclose(again, want + x64 @ A_.astype(np.float64).T, **tol)
    be.forget_constants()
    with Counting(ctx) as c:
        back = be.to_host(f(X=Xd, Am=Ad, Bm=Bd, cv=cd, ev=ed))
    # (unmarked X: the X * X of the second product may be formed inside it -- bsc_gemm_fused)
    assert c.count("bsc_gemm_strided_batched") + c.count("bsc_gemm_epilogue") + c.count("bsc_gemm_fused") >= 2
    npt.assert_allclose(back, want, **tol)


def test_statistics_against_the_wide_operand_are_one_pass(ctx):
    """Once [X | X^2 | 1] exists (previous test), the products of a CONSTANT R with its constituents --
    dot(R.T, X), dot(R.T, X * X) -- and the column sums of R are column blocks of ONE product
    dot(R.T, [X | X^2 | 1]), computed once while R stays marked: the responsibility-weighted statistics
    of a mixture in one pass over R instead of three."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    be = DeviceBackend(ctx)
    rs = np.random.RandomState(11)
    n_rows, d, k = 5003, 16, 24
    X_ = rs.standard_normal((n_rows, d)).astype(np.float32)
    A_ = rs.standard_normal((k, d)).astype(np.float32)
    B_ = rs.rand(k, d).astype(np.float32)
    c_ = rs.standard_normal(k).astype(np.float32)
    R_ = rs.rand(n_rows, k).astype(np.float32)
    X, Am, Bm, cv, R = var("X", 2), var("Am", 2), var("Bm", 2), var("cv", 1), var("R", 2)
    logits = (dot(X, Am.T) + dot(X * X, Bm.T) * (-0.5) + dimshuffle(cv, "x", 0)).compile(be).device_fn
    s1 = dot(R.T, X).compile(be).device_fn
    s2 = (dot(R.T, X * X) * (-0.5)).compile(be).device_fn
    s0 = sum(R, axis=0).compile(be).device_fn
    Xd, Ad, Bd, Rd = (be.from_host(a, "float32", 2) for a in (X_, A_, B_, R_))
    cd = be.from_host(c_, "float32", 1)
    be.mark_constant(Xd)
    be.to_host(logits(X=Xd, Am=Ad, Bm=Bd, cv=cd))            # builds the wide operand
    x64, r64 = X_.astype(np.float64), R_.astype(np.float64)

    def check(r64):
        with Counting(ctx) as c:
            a = be.to_host(s1(R=Rd, X=Xd))
            b = be.to_host(s2(R=Rd, X=Xd))
            z = be.to_host(s0(R=Rd))
        npt.assert_allclose(a, r64.T @ x64, rtol=2e-5, atol=1e-3)
        npt.assert_allclose(b, -0.5 * (r64.T @ (x64 * x64)), rtol=2e-5, atol=1e-3)
        npt.assert_allclose(z, r64.sum(axis=0), rtol=2e-5)
        return c.count("bsc_gemm_strided_batched") + c.count("bsc_gemm_epilogue"), c.count("bsc_sum")

    gemms, sums = check(r64)
    assert gemms == 2 and sums == 1                          # R not marked: as before
    be.mark_constant_tensor(Rd)
    gemms, sums = check(r64)
    assert gemms == 1 and sums == 0                          # one product serves all three
    gemms, sums = check(r64)
    assert gemms == 0 and sums == 0                          # and is kept while R stands
    # R changes: un-mark first (what CategoricalNode.set_eta does), then the statistics follow
    be.unmark_constant(Rd)
    R2 = (R_ * 0.5 + 0.125).astype(np.float32)
    Rd.copy_(be.from_host(R2, "float32", 2))
    be.mark_constant_tensor(Rd)
    gemms, sums = check(R2.astype(np.float64))
    assert gemms == 1 and sums == 0


def test_dropping_a_constant_drops_what_was_built_from_it(ctx):
    """unmark_constant(X) removes the cached X * X, the wide operand [X | X^2 | 1] built from both and
    the products with it; the same expressions then run on the plain route and still agree."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    be = DeviceBackend(ctx)
    rs = np.random.RandomState(12)
    n_rows, d, k = 2000, 8, 5
    X_ = rs.standard_normal((n_rows, d)).astype(np.float32)
    A_ = rs.standard_normal((k, d)).astype(np.float32)
    R_ = rs.rand(n_rows, k).astype(np.float32)
    X, Am, R = var("X", 2), var("Am", 2), var("R", 2)
    logits = (dot(X, Am.T) + dot(X * X, Am.T) * 0.5).compile(be).device_fn
    stat = dot(R.T, X * X).compile(be).device_fn
    Xd, Ad, Rd = (be.from_host(a, "float32", 2) for a in (X_, A_, R_))
    be.mark_constant(Xd)
    be.mark_constant_tensor(Rd)
    x64 = X_.astype(np.float64)
    want_l = x64 @ A_.T.astype(np.float64) + 0.5 * (x64 * x64) @ A_.T.astype(np.float64)
    want_s = R_.astype(np.float64).T @ (x64 * x64)
    for _ in range(2):
        npt.assert_allclose(be.to_host(logits(X=Xd, Am=Ad)), want_l, rtol=2e-5, atol=1e-4)
        npt.assert_allclose(be.to_host(stat(R=Rd, X=Xd)), want_s, rtol=2e-5, atol=1e-3)
    assert be._wide and any(key[0] == "kcat" for key in be._const_cache)
    be.unmark_constant(Xd)
    assert not be._wide and not be._const_cache            # X * X, the wide operand, the products: gone
    X2 = (X_ * 2).astype(np.float32)
    Xd.copy_(be.from_host(X2, "float32", 2))                # X may change now
    npt.assert_allclose(be.to_host(logits(X=Xd, Am=Ad)), 2 * (x64 @ A_.T.astype(np.float64))
                        + 2.0 * (x64 * x64) @ A_.T.astype(np.float64), rtol=2e-5, atol=1e-4)
    npt.assert_allclose(be.to_host(stat(R=Rd, X=Xd)), 4 * want_s, rtol=2e-5, atol=1e-3)


# ---- an element-wise PRODUCER of a product's operand goes into the product (bsc_gemm_fused) -------

@pytest.mark.parametrize("M,N,K", [(300, 260, 4096), (128, 128, 64), (1000, 96, 777 * 4), (256, 256, 40000)])
def test_producers_fold_into_the_operand_reads(dev, M, N, K):
    """dot(exp(X), Y), dot(X * X, A.T), dot(abs(X), Y * Y) and a consumer folded into the same launch's store: ONE
    launch each, no element-wise launch, against float64 numpy."""
    X, Y, Am, E = var("X", ndim=2), var("Y", ndim=2), var("Am", ndim=2), var("E", ndim=2)
    X_ = (RNG.standard_normal((M, K)) * 0.5).astype(np.float32)
    Y_ = RNG.standard_normal((K, N)).astype(np.float32)
    A_ = RNG.standard_normal((N, K)).astype(np.float32)
    E_ = (RNG.rand(M, N) + 0.5).astype(np.float32)
    X64, Y64, A64 = X_.astype(np.float64), Y_.astype(np.float64), A_.astype(np.float64)
    cases = [
        (dot(exp(X), Y), np.exp(X64) @ Y64),
        (dot(X * X, Am.T), (X64 * X64) @ A64.T),
        (dot(abs_(X), Y * Y), np.abs(X64) @ (Y64 * Y64)),
        (dot(2.0 * (X * X), Y), 2.0 * (X64 * X64) @ Y64),
        (E * dot(X ** 2, Y), E_ * ((X64 ** 2) @ Y64)),
        (dot(exp(Am), Y), np.exp(A64) @ Y64),            # the producer on the LEFT operand of a [N, K] x [K, N] product
    ]
    for expr, want in cases:
        f = expr.compile(dev)
        with Counting(dev.ctx) as c:
            got = f(X=X_, Y=Y_, Am=A_, E=E_)
        assert c.count("bsc_gemm_fused") == 1, (repr(expr), c.calls)
        assert c.count("bsc_map_reduce") + c.count("bsc_elemwise") + c.count("bsc_gemm_strided_batched") \
            + c.count("bsc_gemm_epilogue") == 0, (repr(expr), c.calls)
        scale = np.abs(want).max()
        npt.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * scale, err_msg=repr(expr))


def test_producers_fall_back_when_the_shape_takes_another_kernel(dev):
    """A matrix-vector product (fused map-reduce), a skinny product (bsc_gemm_skinny: no prologue there -> handled = 0
    and the producer is launched), exp on both sides (padded zeros would not cancel): right values whatever the path."""
    X, Y, v = var("X", ndim=2), var("Y", ndim=2), var("v", ndim=1)
    X_ = (RNG.standard_normal((5000, 256)) * 0.5).astype(np.float32)
    Y_ = (RNG.standard_normal((256, 8)) * 0.5).astype(np.float32)
    v_ = RNG.standard_normal(256).astype(np.float32)
    X64, Y64 = X_.astype(np.float64), Y_.astype(np.float64)
    for expr, want in [(dot(exp(X), v), np.exp(X64) @ v_), (dot(exp(X), Y), np.exp(X64) @ Y64),
                       (dot(exp(X), exp(Y)), np.exp(X64) @ np.exp(Y64)), (dot(X * X, Y), (X64 * X64) @ Y64)]:
        got = expr.compile(dev)(X=X_, Y=Y_, v=v_)
        npt.assert_allclose(got, want, rtol=2e-4, atol=2e-4)


def test_constant_operands_keep_their_cached_producers(dev):
    """X marked constant (a model's data): X * X is computed once and cached, not recomputed inside every product."""
    X, Y = var("X", ndim=2), var("Y", ndim=2)
    X_ = RNG.standard_normal((512, 2048)).astype(np.float32)
    Y_ = RNG.standard_normal((2048, 256)).astype(np.float32)
    Xd = dev.from_host(X_, "float32", 2)
    dev.mark_constant(Xd)
    try:
        f = dot(X * X, Y).compile(dev)
        for _ in range(2):
            with Counting(dev.ctx) as c:
                got = f.device_fn(X=Xd, Y=dev.from_host(Y_, "float32", 2))
        assert c.count("bsc_gemm_fused") == 0
        npt.assert_allclose(dev.to_host(got), (X_.astype(np.float64) ** 2) @ Y_, rtol=2e-4, atol=1e-3)
    finally:
        dev.unmark_constant(Xd)
