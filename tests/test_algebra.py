"""Restatement of the reference's own test strategy for the algebra front end
(bayesic/tests/test_algebra.py), against bayesic_amd.algebra.

Same three kinds of assertion: (1) numeric vs numpy expressions on the
seed-1234 float32 inputs with the reference's tolerances (rtol 1e-7 element-wise,
1e-5 wherever a contraction is involved); (2) structural ==; (3) the shape of the
lowered five-op tree.  Numeric evaluation here goes through the oracle's numpy
backend (CPU); tests/test_algebra_gpu.py runs the same checks through the HIP
backend.  Also covers what the reference could not run: batched _tensordot
execution and expressions on which the reference's lowering crashes.
"""
import numpy as np
import numpy.random as npr
import numpy.testing as npt
import pytest

from bayesic_amd.algebra import *  # noqa: F401,F403
from bayesic_amd.algebra import _diagonal, _dimshuffle, _mul, _sum, _tensordot
from oracle.einsum_eval import NumpyBackend, einsum_semantics

BACKEND = NumpyBackend()

# same draw order as bayesic/tests/test_algebra.py:15-38
_rs = npr.RandomState(1234)


def randn(*shape_):
    return _rs.randn(*shape_).astype("float32")


X, Y, Z, W = var("X", 2), var("Y", 2), var("Z", 2), var("W", 2)
X_, Y_ = randn(5, 5), randn(5, 5)
x, y = var("x", 1), var("y", 1)
x_, y_ = randn(5), randn(5)
S = var("S", 3)
S_ = randn(3, 5, 7)
a = var("a", 0, "int32")
Z_, W_ = randn(5, 5), randn(5, 5)
VALUES = dict(X=X_, Y=Y_, Z=Z_, W=W_, x=x_, y=y_, S=S_)


def run(expr, **inputs):
    return expr.compile(BACKEND)(**inputs)


def test_star_import_surface():
    import bayesic_amd.algebra as alg
    for name in ["var", "constant", "shape", "eye", "elemwise", "add", "einsum", "Einsum", "match",
                 "dot", "tensordot", "mul", "outer", "sum", "trace", "diagonal", "transpose",
                 "dimshuffle", "div", "neg", "sub", "log", "exp", "pow", "abs_",
                 "find_injections", "find_injection", "find_bijections", "find_bijection",
                 "submultisets_of_size", "find_duplicate", "equivalence_classes",
                 "wrap_if_literal", "with_wrapped_literals", "autobroadcast_or_match",
                 "np", "it", "Counter", "defaultdict"]:
        assert name in alg.__all__ and hasattr(alg, name), name
    assert Counter([1, 1]) == {1: 2}      # the reference tests rely on the star-imported Counter


# ---- (1) numeric vs numpy ---------------------------------------------------

def test_add_sub_abs():
    npt.assert_allclose(run(X + Y, X=X_, Y=Y_), X_ + Y_)
    npt.assert_allclose(run(X - Y, X=X_, Y=Y_), X_ - Y_)
    npt.assert_allclose(run(abs(X), X=X_), abs(X_))


def test_scalar_autobroadcast():
    npt.assert_allclose(run(X + 1, X=X_), X_ + 1)
    npt.assert_allclose(run(1 - X, X=X_), 1 - X_)
    npt.assert_allclose(run(2 * X, X=X_), 2 * X_)
    assert run(add(1, 1)) == 2


def test_literal_wrapping():
    npt.assert_allclose(run(X * Y_, X=X_), X_ * Y_)


def test_dots():
    npt.assert_allclose(run(dot(X, Y), X=X_, Y=Y_), np.dot(X_, Y_), rtol=1e-5)
    npt.assert_allclose(run(X.dot(y), X=X_, y=y_), np.dot(X_, y_), rtol=1e-5)
    npt.assert_allclose(run(dot(x, y), x=x_, y=y_), np.dot(x_, y_), rtol=1e-5)
    npt.assert_allclose(run(X @ Y, X=X_, Y=Y_), X_ @ Y_, rtol=1e-5)


def test_mul_div_pow_log_exp():
    npt.assert_allclose(run(X * Y, X=X_, Y=Y_), X_ * Y_, rtol=1e-5)
    npt.assert_allclose(run(X / Y, X=X_, Y=Y_), X_ / Y_, rtol=1e-5)
    with np.errstate(all="ignore"):
        npt.assert_allclose(run(X ** Y, X=X_, Y=Y_), X_ ** Y_)      # NaN == NaN here
        npt.assert_allclose(run(2 ** X, X=X_), 2 ** X_)
        npt.assert_allclose(run(X ** 2, X=X_), X_ ** 2)
        npt.assert_allclose(run(log(X), X=X_), np.log(X_))
    npt.assert_allclose(run(exp(X), X=X_), np.exp(X_))


def test_transpose_dimshuffle_broadcasting():
    npt.assert_allclose(run(X.T, X=X_), X_.T)
    npt.assert_allclose(run(dimshuffle(S, 2, 0, 1), S=S_), np.transpose(S_, (2, 0, 1)))
    npt.assert_allclose(run(X + x.dimshuffle(0, "x"), X=X_, x=x_), X_ + x_[:, None])
    npt.assert_allclose(run(X * dimshuffle(x, "x", 0), X=X_, x=x_), X_ * x_[None, :])


def test_trace_diagonal_outer_sum():
    npt.assert_allclose(run(trace(X), X=X_), np.trace(X_), rtol=1e-6)
    npt.assert_allclose(run(diagonal(X), X=X_), np.diagonal(X_))
    npt.assert_allclose(run(outer(x, y), x=x_, y=y_), np.outer(x_, y_))
    npt.assert_allclose(run(sum(S), S=S_), S_.sum(), rtol=1e-5)
    npt.assert_allclose(run(sum(S, axis=0), S=S_), S_.sum(axis=0), rtol=1e-5)
    npt.assert_allclose(run(S.sum(axis=(0, 2)), S=S_), S_.sum(axis=(0, 2)), rtol=1e-5)


def test_shape_size_eye():
    data = np.array([[1, 2], [3, 4], [5, 6]], dtype="float32")
    assert run(X.shape[0], X=data) == 3
    assert run(X.shape[1], X=data) == 2
    assert run(X.size, X=data) == 6
    npt.assert_equal(run(eye(a), a=2), np.eye(2))
    npt.assert_equal(run(eye(a), a=5), np.eye(5))


def test_composition_of_einsums_collapses_to_single_einsum():
    expr = dot(diagonal(dot(X, outer(x, y))), Y)
    assert expr.parents == (X, x, y, Y)
    npt.assert_allclose(run(expr, x=x_, y=y_, X=X_, Y=Y_),
                        np.dot(np.diagonal(np.dot(X_, np.outer(x_, y_))), Y_), rtol=1e-5)


def test_two_equivalent_einsum_expressions_same_result():
    expr, expr2 = trace(dot(X.T, Y)), sum(X * Y)
    assert expr.parents == (X, Y) and expr2.parents == (X, Y)
    assert expr.factors_and_indices == expr2.factors_and_indices


def test_batched_tensordot_executes_by_einsum_semantics():
    # lowered to a batched _tensordot (pinned structurally by the reference, which
    # cannot execute it: bayesic/algebra.py:1358-1383)
    e = (X * Y.T).sum(axis=1)
    assert e._rewrite_as_special_case_ops() == _tensordot(X, Y, [1], [0], [0], [1])
    npt.assert_allclose(run(e, X=X_, Y=Y_), (X_ * Y_.T).sum(axis=1), rtol=1e-5)
    e2 = tensordot(S, S, [1], [1], [0], [0])        # batched matrix-matrix
    npt.assert_allclose(run(e2, S=S_), np.einsum("uiv,uiw->uvw", S_, S_), rtol=1e-5)


def test_expressions_the_reference_lowering_crashes_on():
    # dot(sum(X,0), y): TypeError (unhashable list) at bayesic/algebra.py:636 after :592
    e = dot(sum(X, 0), y)
    npt.assert_allclose(run(e, X=X_, y=y_), X_.sum(0) @ y_, rtol=1e-5)
    e = S.sum(axis=(0, 2)).dot(y)
    npt.assert_allclose(run(e, S=S_, y=y_), S_.sum(axis=(0, 2)) @ y_, rtol=1e-5)


# every corpus-style expression: lowered tree == einsum definition == (where given) numpy
CHECKS = [
    (lambda: dot(X, dot(Y, Z)), lambda: X_ @ (Y_ @ Z_)),
    (lambda: dot(dot(X, Y), dot(Z, W)), lambda: (X_ @ Y_) @ (Z_ @ W_)),
    (lambda: X * Y.T * x.dimshuffle(0, "x"), lambda: X_ * Y_.T * x_[:, None]),
    (lambda: X.sum(1) * y, lambda: X_.sum(1) * y_),
    (lambda: dot(Z, x * y), lambda: Z_ @ (x_ * y_)),
    (lambda: dot(X * Y, Z * W), lambda: (X_ * Y_) @ (Z_ * W_)),
    (lambda: sum(exp(X) * Y), lambda: (np.exp(X_) * Y_).sum()),
    (lambda: dot(dimshuffle(S, 0, 2, 1), X), lambda: np.tensordot(np.transpose(S_, (0, 2, 1)), X_, 1)),
    (lambda: dot(x, dot(X, x)), lambda: x_ @ X_ @ x_),
    (lambda: trace(dot(X, outer(x, x))), lambda: np.trace(X_ @ np.outer(x_, x_))),
    (lambda: sum(outer(x, y)), lambda: np.outer(x_, y_).sum()),
    (lambda: sum(outer(x, y), axis=1), lambda: np.outer(x_, y_).sum(1)),
    (lambda: dot(outer(x, y), x), lambda: np.outer(x_, y_) @ x_),
    (lambda: tensordot(S, S, [0, 2], [0, 2]), lambda: np.tensordot(S_, S_, ([0, 2], [0, 2]))),
    (lambda: sum(X * X.T), lambda: (X_ * X_.T).sum()),
    (lambda: dot(X.T, X), lambda: X_.T @ X_),
    (lambda: dot(X.T, y), lambda: X_.T @ y_),
    (lambda: dot(dot(X, X), dot(X, X)), lambda: (X_ @ X_) @ (X_ @ X_)),   # reference gets this wrong
    (lambda: einsum([(X, [("sum", 0), ("out", 0)]), (Y, [("sum", 0), ("out", 1)]),
                     (Y, [("sum", 0), ("out", 2)])], 3), lambda: np.einsum("iu,iv,iw->uvw", X_, Y_, Y_)),
    (lambda: 2 * (3 * X), lambda: 6 * X_),
    (lambda: -X, lambda: -X_),
    (lambda: (X + Y) * Z, lambda: (X_ + Y_) * Z_),
    (lambda: dot(X, eye(X.shape[1]) * 2), lambda: 2 * X_),
    (lambda: einsum([], 2), lambda: np.ones((1, 1))),
    (lambda: einsum([(x, [("out", 1)])], 3), lambda: x_[None, :, None]),
]


@pytest.mark.parametrize("case", range(len(CHECKS)))
def test_lowered_tree_equals_einsum_definition(case):
    build, expected = CHECKS[case]
    e = build()
    got = run(e, **{k: VALUES[k] for k in e.input_types})
    want = expected()
    if want is not None:
        npt.assert_allclose(got, want, rtol=2e-5, atol=1e-6)
    if isinstance(e, Einsum):
        ref64 = einsum_semantics(e, {k: VALUES[k].astype(np.float64) for k in e.input_types})
        assert got.shape == ref64.shape
        npt.assert_allclose(got, ref64, rtol=2e-5, atol=1e-5)


# ---- (2) structural ----------------------------------------------------------

def test_equality_of_expressions():
    assert X == X and X != Y
    assert constant(1) == constant(1) and constant(1) != constant(2)
    assert X + Y == X + Y and X + Y == Y + X and X + Y != X + Z
    assert X - Y == -Y + X
    assert X / Y == X * (Y ** -1)
    assert log(X) == log(X) and log(X) != exp(X) and log(X) != log(Y)
    assert X * Y == Y * X
    assert X * X.T == X.T * X and X * X.T != X * X
    assert dot(X, Y).T == dot(Y.T, X.T) and dot(X, Y) != dot(Y, X)
    assert sum(X * X.T) == sum(X.T * X)
    assert sum(X * X.T) != trace(X) * trace(X)
    assert trace(dot(X, Y.T)) == sum(Y * X)
    assert dot(dot(X, Y), Z) == dot(X, dot(Y, Z))


def test_match():
    assert match(X * Y, X * Z, Z) == Y
    assert match(X * X, X * Z, Z) == X
    assert match(X * X, Y * Z, Z) is None
    assert match(Y * X, X * Z, Z) == Y
    assert match(sum(Y * X), sum(X * Z), Z) == Y
    assert match(dot(X, Y), dot(X, Z), Z) == Y
    assert match(dot(X, X), dot(X, Z), Z) == X
    assert match(dot(X, X.T), dot(X, Z), Z) == X.T
    assert match(dot(X, X.T), dot(X.T, Z), Z) is None
    assert match(dot(X, X * X), dot(X, Z), Z) == X * X
    assert match(dot(X, X * X), dot(Z, X * X), Z) == X
    assert match(dot(X, X * X), dot(X * X, Z), Z) is None
    assert match(dot(X, X * X), dot(X, X * Z), Z) == X
    assert match(trace(dot(X, X)), sum(X * Z), Z) == X.T
    assert match(dot(X, Y).T, dot(X, Z), Z) is None
    assert match(dot(X, Y).T, dot(X.T, Z), Z) is None
    assert match(dot(X, Y).T, dot(Z, X.T), Z) == Y.T
    assert match(dot(X, dot(Y, X)), dot(X, Z), Z) == dot(Y, X)
    assert match(X, Z, Z) == X
    assert match(X * Y, Z, Z) == X * Y
    with pytest.raises(ValueError):
        match(X, Y, Z)                       # template must contain the slot


def test_identity_elimination_and_insertion():
    assert dot(X, eye(X.shape[1])) == X
    assert dot(eye(X.shape[0]), X) == X
    assert dot(Y, dot(eye(X.shape[0]), X)) == dot(Y, X)
    assert match(X * Y, dot(X * Y, Z), Z) == eye(X.shape[1])
    assert match(X, dot(X, Z), Z) == eye(X.shape[1])
    assert match(X * y.dimshuffle("x", 0), dot(X, Z), Z) == eye(X.shape[1]) * y.dimshuffle(0, "x")


def test_match_substitution_roundtrip_is_numerically_consistent():
    """Substituting the match back into the template reproduces the expression."""
    for expr, template in [(dot(X, dot(Y, X)), dot(X, Z)), (dot(X, Y).T, dot(Z, X.T)),
                           (X * y.dimshuffle("x", 0), dot(X, Z)), (trace(dot(X, X)), sum(X * Z))]:
        m = match(expr, template, Z)
        vals = dict(VALUES)
        vals["Z"] = run(m, **{k: VALUES[k] for k in m.input_types})
        npt.assert_allclose(run(template, **{k: vals[k] for k in template.input_types}),
                            run(expr, **{k: VALUES[k] for k in expr.input_types}), rtol=1e-5)


def test_construction_errors():
    with pytest.raises(ValueError):
        X + x                                   # rank mismatch needs explicit dimshuffle
    with pytest.raises(ValueError):
        dimshuffle(X, 0, 0)
    with pytest.raises(ValueError):
        dimshuffle(X, 0)
    with pytest.raises(ValueError):
        einsum([(X, [("out", 0)])])
    with pytest.raises(ValueError):
        einsum([(X, [("out", 0), ("out", 5)])], 2)
    with pytest.raises(ValueError):
        eye()
    with pytest.raises(ValueError):
        wrap_if_literal([1, 2])                 # neither scalar, ndarray nor Expression
    with pytest.raises(TypeError):
        (var("q", 1) + var("q", 1, "int32")).input_types


def test_input_types():
    assert (dot(X, y) + exp(var("z", 1))).input_types == \
        {"X": ("float32", 2), "y": ("float32", 1), "z": ("float32", 1)}


# ---- (3) lowering shapes -------------------------------------------------------

def assert_implemented_as(expr, impl):
    rewritten = expr._rewrite_as_special_case_ops()
    assert rewritten == impl, "%r != %r" % (rewritten, impl)


def test_basic_einsum_rewriting():
    assert_implemented_as(diagonal(X), _diagonal(X, 0, 1))
    assert_implemented_as(dot(X, Y), _tensordot(X, Y, [1], [0]))
    assert_implemented_as(sum(X, 1), _sum(X, 1))
    assert_implemented_as(mul(X, Y), _mul(X, Y))
    assert_implemented_as(dimshuffle(X, 1, 0), _dimshuffle(X, 1, 0))


def test_nested_tensordots_preserve_bracketing():
    assert_implemented_as(dot(X, dot(Y, Z)), _tensordot(X, _tensordot(Y, Z, [1], [0]), [1], [0]))
    assert_implemented_as(dot(dot(X, Y), Z), _tensordot(_tensordot(X, Y, [1], [0]), Z, [1], [0]))
    assert_implemented_as(dot(dot(X, Y), dot(Z, W)),
                          _tensordot(_tensordot(X, Y, [1], [0]), _tensordot(Z, W, [1], [0]), [1], [0]))
    assert_implemented_as(dot(dot(X, dot(Y, Z)), W),
                          _tensordot(_tensordot(X, _tensordot(Y, Z, [1], [0]), [1], [0]), W, [1], [0]))
    assert dot(X, dot(Y, Z)) == dot(dot(X, Y), Z)
    assert dot(dot(X, Y), dot(Z, W)) == dot(dot(X, dot(Y, Z)), W)


def test_rewriting_without_dot():
    assert_implemented_as(X * Y.T * x.dimshuffle(0, "x"),
                          _mul(X, _dimshuffle(Y, 1, 0), _dimshuffle(x, 0, "x")))
    assert_implemented_as(X.sum(1) * y, _mul(_sum(X, 1), y))


def test_rewriting_as_tensordot():
    assert_implemented_as(trace(dot(X.T, Y)), _tensordot(X, Y, [0, 1], [0, 1]))
    assert_implemented_as((X * Y.T).sum(axis=1),
                          _tensordot(X, Y, X_dot_axes=[1], Y_dot_axes=[0],
                                     X_batch_axes=[0], Y_batch_axes=[1]))


def test_lhs_rhs_grouping_heuristic():
    for e in (dot(Z, x * y), dot(Z * x.dimshuffle("x", 0), y), dot(Z * y.dimshuffle("x", 0), x)):
        assert_implemented_as(e, _tensordot(Z, _mul(x, y), [1], [0]))
    assert_implemented_as(dot(X * Y, Z * W), _tensordot(_mul(X, Y), _mul(Z, W), [1], [0]))
    assert_implemented_as(
        tensordot(X.dimshuffle(0, 1, "x") * Z.dimshuffle("x", 0, 1),
                  Y.dimshuffle(0, 1, "x") * W.dimshuffle("x", 0, 1),
                  X_sum_axes=[1], Y_sum_axes=[1], X_batch_axes=[0, 2], Y_batch_axes=[0, 2]),
        _tensordot(_mul(X, Y), _mul(Z, W), [1], [0]))


def test_config_shaped_lowerings():
    """The GEMM shapes the device sees for the BASELINE configs."""
    R, Th, C = var("R", 2), var("Th", 2), var("C", 2)
    assert_implemented_as(dot(X.T, X), _tensordot(_dimshuffle(X, 1, 0), X, [1], [0]))
    assert_implemented_as(dot(X.T, y), _tensordot(_dimshuffle(X, 1, 0), y, [1], [0]))
    assert_implemented_as(dot(X, x), _tensordot(X, x, [1], [0]))
    assert_implemented_as(sum(x * x), _tensordot(x, x, [0], [0]))
    assert_implemented_as(dot(R.T, X * X), _tensordot(_dimshuffle(R, 1, 0), _mul(X, X), [1], [0]))
    assert_implemented_as(mul(W, dot(Th.T, C)),
                          _mul(W, _tensordot(_dimshuffle(Th, 1, 0), C, [1], [0])))
