"""Expression corpus shared by the golden-fixture generator (which evaluates it
with the REFERENCE front end) and tests/test_algebra_golden.py (which evaluates
it with bayesic_amd.algebra).  Pure data: source strings only.

Every string is evaluated in a namespace holding the public algebra API plus
the symbolic inputs of the reference's own tests
(bayesic/tests/test_algebra.py:23-41): X Y Z W (ndim 2), x y z (ndim 1),
S (ndim 3), a (int32 scalar), R Th C P B (config-shaped operands).
"""

NAMESPACE_VARS = [
    ("X", 2, "float32"), ("Y", 2, "float32"), ("Z", 2, "float32"), ("W", 2, "float32"),
    ("x", 1, "float32"), ("y", 1, "float32"), ("z", 1, "float32"),
    ("S", 3, "float32"), ("a", 0, "int32"),
    ("R", 2, "float32"), ("Th", 2, "float32"), ("C", 2, "float32"), ("P", 3, "float32"),
    ("B", 2, "float32"), ("mu", 1, "float32"), ("tau", 1, "float32"), ("eta", 2, "float32"),
]

# canonical repr / lowered repr / input_types / ndim
EXPRESSIONS = [
    "X", "X + Y", "X - Y", "abs(X)", "X + 1", "1 - X", "2 * X", "add(1, 1)",
    "dot(X, Y)", "X.dot(y)", "dot(x, y)", "X * Y", "X / Y", "X ** Y", "2 ** X", "X ** 2",
    "log(X)", "exp(X)", "X.T", "X.T.T", "dimshuffle(S, 2, 0, 1)", "dimshuffle(X, 0, 1)",
    "X + x.dimshuffle(0, 'x')", "X * dimshuffle(x, 'x', 0)", "dimshuffle(x, 'x', 0)",
    "trace(X)", "diagonal(X)", "outer(x, y)", "sum(S)", "sum(S, axis=0)", "S.sum(axis=(0, 2))",
    "X.shape[0]", "X.size", "eye(a)", "eye(X.shape[0])",
    "dot(diagonal(dot(X, outer(x, y))), Y)", "trace(dot(X.T, Y))", "sum(X * Y)",
    "dot(X, dot(Y, Z))", "dot(dot(X, Y), Z)", "dot(dot(X, Y), dot(Z, W))",
    "dot(dot(X, dot(Y, Z)), W)", "X * Y.T * x.dimshuffle(0, 'x')", "X.sum(1) * y",
    "(X * Y.T).sum(axis=1)", "dot(Z, x * y)", "dot(Z * x.dimshuffle('x', 0), y)",
    "dot(Z * y.dimshuffle('x', 0), x)", "dot(X * Y, Z * W)",
    "tensordot(X.dimshuffle(0, 1, 'x') * Z.dimshuffle('x', 0, 1), "
    "Y.dimshuffle(0, 1, 'x') * W.dimshuffle('x', 0, 1), "
    "X_sum_axes=[1], Y_sum_axes=[1], X_batch_axes=[0, 2], Y_batch_axes=[0, 2])",
    "dot(X, eye(X.shape[1]))", "dot(eye(X.shape[0]), X)", "dot(Y, dot(eye(X.shape[0]), X))",
    "2 * (3 * X)", "-X", "(X + Y) * Z", "sum(exp(X) * Y)", "sum(x ** 2)", "sum(x * x)",
    "sum(x)", "sum(x * y * z)", "dot(diagonal(X), y)", "dot(S, X)", "dot(x, dot(X, x))",
    "trace(dot(X, outer(x, x)))", "dot(X, y) + exp(z)", "mul(x, y, z)", "outer(X, y)",
    "tensordot(S, X, [2], [0])", "tensordot(S, S, [0, 2], [0, 2])",
    "tensordot(S, S, [1], [1], [0], [0])", "sum(X, 1)", "sum(X, 0)", "sum(X * X.T)",
    "trace(X) * trace(X)", "X * X.T", "dot(X, Y).T", "dot(Y.T, X.T)", "trace(dot(X, Y.T))",
    # config-shaped
    "dot(X.T, X)", "dot(X.T, y)", "dot(X, x)", "sum(R, 0)", "dot(R.T, X)", "dot(R.T, X * X)",
    "einsum([(R, [('sum', 0), ('out', 0)]), (X, [('sum', 0), ('out', 1)]), "
    "(X, [('sum', 0), ('out', 2)])], 3)",
    "B * dot(Th.T, C)",
    "einsum([(C, [('sum', 0), ('out', 0)]), (P, [('sum', 0), ('out', 0), ('out', 1)])], 2)",
    "dot(W, X.T)", "sum(eta * X)", "sum(x * mu * tau)", "sum(x * x * tau)",
    "einsum([(X, [('out', 0), ('sum', 0)]), (Y, [('sum', 0), ('sum', 1)]), (x, [('sum', 1)])], 1)",
    "einsum([], 2)", "einsum([(x, [('out', 1)])], 3)",
    "sum(outer(x, y))", "sum(outer(x, y), axis=1)", "diagonal(outer(x, y))",
    "trace(outer(x, y))", "dot(outer(x, y), z)", "sum(S * S)", "sum(S, axis=1) * X.dimshuffle(0, 'x', 1).sum(1)",
    "sum(S.dimshuffle(1, 0, 2) * dimshuffle(X, 0, 'x', 1), axis=(0, 2))",
    "dot(X, X)", "dot(X, X.T)", "dot(X, X * X)", "trace(dot(X, X))",
    "X * y.dimshuffle('x', 0)", "eye(X.shape[1]) * y.dimshuffle(0, 'x')",
]

# (expression, template, slot-name) -> repr(match) or None
MATCHES = [
    ("X * Y", "X * Z", "Z"), ("X * X", "X * Z", "Z"), ("X * X", "Y * Z", "Z"),
    ("Y * X", "X * Z", "Z"), ("sum(Y * X)", "sum(X * Z)", "Z"), ("dot(X, Y)", "dot(X, Z)", "Z"),
    ("dot(X, X)", "dot(X, Z)", "Z"), ("dot(X, X.T)", "dot(X, Z)", "Z"),
    ("dot(X, X.T)", "dot(X.T, Z)", "Z"), ("dot(X, X*X)", "dot(X, Z)", "Z"),
    ("dot(X, X*X)", "dot(Z, X*X)", "Z"), ("dot(X, X*X)", "dot(X*X, Z)", "Z"),
    ("dot(X, X*X)", "dot(X, X*Z)", "Z"), ("trace(dot(X, X))", "sum(X*Z)", "Z"),
    ("dot(X, Y).T", "dot(X, Z)", "Z"), ("dot(X, Y).T", "dot(X.T, Z)", "Z"),
    ("dot(X, Y).T", "dot(Z, X.T)", "Z"), ("dot(X, dot(Y, X))", "dot(X, Z)", "Z"),
    ("X", "Z", "Z"), ("X * Y", "Z", "Z"), ("X * Y", "dot(X*Y, Z)", "Z"),
    ("X", "dot(X, Z)", "Z"), ("X * y.dimshuffle('x', 0)", "dot(X, Z)", "Z"),
    ("sum(x * mu * tau)", "sum(x * z)", "z"), ("sum(x * x * tau)", "sum(x * x * z)", "z"),
    ("dot(X.T, dot(X, y))", "dot(Z, y)", "Z"), ("sum(eta * X)", "sum(Z * X)", "Z"),
    ("dot(x, dot(X, x))", "sum(X * Z)", "Z"), ("2 * X", "X * a", "a"),
    ("sum(x * y)", "dot(x, z)", "z"),
]

# (lhs, rhs) -> bool(lhs == rhs), and hash agreement when equal
EQUALITIES = [
    ("X", "X"), ("X", "Y"), ("constant(1)", "constant(1)"), ("constant(1)", "constant(2)"),
    ("X + Y", "X + Y"), ("X + Y", "Y + X"), ("X + Y", "X + Z"), ("X - Y", "-Y + X"),
    ("X / Y", "X * (Y ** -1)"), ("log(X)", "log(X)"), ("log(X)", "exp(X)"), ("log(X)", "log(Y)"),
    ("X * Y", "Y * X"), ("X * X.T", "X.T * X"), ("X * X.T", "X * X"),
    ("dot(X, Y).T", "dot(Y.T, X.T)"), ("dot(X, Y)", "dot(Y, X)"),
    ("sum(X * X.T)", "sum(X.T * X)"), ("sum(X * X.T)", "trace(X) * trace(X)"),
    ("trace(dot(X, Y.T))", "sum(Y * X)"), ("dot(dot(X, Y), Z)", "dot(X, dot(Y, Z))"),
    ("dot(X, eye(X.shape[1]))", "X"), ("dot(eye(X.shape[0]), X)", "X"),
    ("dot(Y, dot(eye(X.shape[0]), X))", "dot(Y, X)"),
    ("dot(dot(X, Y), dot(Z, W))", "dot(dot(X, dot(Y, Z)), W)"),
    ("dot(x, dot(X, x))", "trace(dot(X, outer(x, x)))"),
    ("eye(X.shape[0])", "eye(X.shape[0], Y.shape[1])"), ("eye(X.shape[0])", "eye(X.shape[1])"),
    ("sum(x * y * z)", "dot(x * z, y)"), ("outer(x, y).T", "outer(y, x)"),
    ("trace(X)", "sum(diagonal(X))"), ("X.T.T", "X"),
]
