"""Public einsum-based constructors, the five executable primitives, the
element-wise sugar and the operator overloads.

Behavioural contract: bayesic/algebra.py:1149-1277 (dot, tensordot, mul, outer,
sum, trace, diagonal, transpose, dimshuffle), :1284-1414 (_sum, _mul,
_dimshuffle, _tensordot, _diagonal), :1416-1448 (div, neg, sub, log, exp, pow,
abs_), :1451-1478 (operators).  The primitives carry no numeric code: each one
asks the backend for the matching device operation.
"""
import builtins

from .einsum_form import OUT, SUM, einsum
from .expr import (OPS, Expression, add, autobroadcast_or_match, elemwise, wrap_if_literal,
                   with_wrapped_literals)

_builtin_sum = builtins.sum


# ---------------------------------------------------------------------------
# constructors: every one of these is just an einsum
# ---------------------------------------------------------------------------

@with_wrapped_literals
def dot(X, Y):
    """Contracts the last axis of X with the first axis of Y."""
    k = (SUM, 0)
    x_out = [(OUT, i) for i in range(X.ndim - 1)]
    y_out = [(OUT, i) for i in range(X.ndim - 1, X.ndim + Y.ndim - 2)]
    return einsum(((X, x_out + [k]), (Y, [k] + y_out)), X.ndim + Y.ndim - 2)


def tensordot(X, Y, X_sum_axes, Y_sum_axes, X_batch_axes=[], Y_batch_axes=[]):
    """tensordot / batched tensordot with explicit axes.  Result axes: batch axes,
    then X's remaining axes, then Y's remaining axes (bayesic/algebra.py:1161-1194)."""
    X, Y = wrap_if_literal(X), wrap_if_literal(Y)
    x_idx, y_idx = [None] * X.ndim, [None] * Y.ndim
    for n, axis in enumerate(X_sum_axes):
        x_idx[axis] = (SUM, n)
    for n, axis in enumerate(Y_sum_axes):
        y_idx[axis] = (SUM, n)
    for n, axis in enumerate(X_batch_axes):
        x_idx[axis] = (OUT, n)
    for n, axis in enumerate(Y_batch_axes):
        y_idx[axis] = (OUT, n)
    next_out = len(X_batch_axes)
    for idx in (x_idx, y_idx):
        for axis in range(len(idx)):
            if idx[axis] is None:
                idx[axis] = (OUT, next_out)
                next_out += 1
    return einsum(((X, x_idx), (Y, y_idx)), next_out)


@with_wrapped_literals
def mul(*args):
    """Element-wise (Hadamard) product; scalars broadcast, other ranks must match."""
    ndim = max(arg.ndim for arg in args)
    args = [autobroadcast_or_match(arg, ndim) for arg in args]
    return einsum([(arg, [(OUT, i) for i in range(arg.ndim)]) for arg in args], ndim)


@with_wrapped_literals
def outer(X, Y):
    """Tensor (outer) product of any two tensors."""
    return einsum([(X, [(OUT, i) for i in range(X.ndim)]),
                   (Y, [(OUT, i) for i in range(X.ndim, X.ndim + Y.ndim)])], X.ndim + Y.ndim)


def sum(X, axis=None):
    """Sum over all axes, or over the given axis / axes."""
    if isinstance(axis, int):
        axis = [axis]
    X = wrap_if_literal(X)
    if axis is None:
        axis = range(X.ndim)
    indices, n_sum, n_out = [], 0, 0
    for i in range(X.ndim):
        if i in axis:
            indices.append((SUM, n_sum))
            n_sum += 1
        else:
            indices.append((OUT, n_out))
            n_out += 1
    return einsum([(X, indices)], n_out)


@with_wrapped_literals
def trace(X):
    return einsum([(X, [(SUM, 0), (SUM, 0)])], 0)


@with_wrapped_literals
def diagonal(X):
    return einsum([(X, [(OUT, 0), (OUT, 0)])], 1)


@with_wrapped_literals
def transpose(X):
    return dimshuffle(X, *reversed(range(X.ndim)))


def dimshuffle(X, *axes):
    """Permute axes; 'x' inserts a broadcastable axis (as Theano's dimshuffle)."""
    X = wrap_if_literal(X)
    indices = [None] * X.ndim
    for position, axis in enumerate(axes):
        if axis != "x":
            if indices[axis] is not None:
                raise ValueError("dimshuffle: same input axis can't occur twice")
            indices[axis] = (OUT, position)
    if any(i is None for i in indices):
        raise ValueError("dimshuffle: can't drop an axis")
    return einsum([(X, indices)], len(axes))


# ---------------------------------------------------------------------------
# the executable IR (no argument checking: internal, but imported by tests)
# ---------------------------------------------------------------------------

class _sum(Expression):
    def __init__(self, X, *axes):
        self.axes = tuple(axes)
        self.ndim = X.ndim - len(axes)
        super(_sum, self).__init__([X])

    def _emit(self, backend, X):
        return backend.sum(X, self.axes)

    def _equality_by(self):
        return (self.parents[0], frozenset(self.axes))


class _mul(Expression):
    def __init__(self, *factors):
        self.ndim = factors[0].ndim
        super(_mul, self).__init__(factors)

    def _emit(self, backend, *factors):
        return factors[0] if len(factors) == 1 else backend.mul(*factors)

    def _equality_by(self):
        return frozenset(self.parents)


class _dimshuffle(Expression):
    def __init__(self, X, *axes):
        self.axes = tuple(axes)
        self.ndim = len(axes)
        super(_dimshuffle, self).__init__([X])

    def _emit(self, backend, X):
        return backend.dimshuffle(X, self.axes)

    def _equality_by(self):
        return (self.parents[0], self.axes)

    def __repr__(self):
        return "%s(%r, %s)" % (type(self).__name__, self.parents[0],
                               ", ".join(repr(a) for a in self.axes))


class _tensordot(Expression):
    """Contract X_dot_axes of X with Y_dot_axes of Y; X_batch_axes / Y_batch_axes
    are paired batch axes.  Result axes: batch, X others, Y others."""

    def __init__(self, X, Y, X_dot_axes, Y_dot_axes, X_batch_axes=[], Y_batch_axes=[]):
        self.X_dot_axes = X_dot_axes
        self.Y_dot_axes = Y_dot_axes
        self.X_batch_axes = X_batch_axes
        self.Y_batch_axes = Y_batch_axes
        self.X_other_axes = [n for n in range(X.ndim)
                             if n not in X_dot_axes and n not in X_batch_axes]
        self.Y_other_axes = [n for n in range(Y.ndim)
                             if n not in Y_dot_axes and n not in Y_batch_axes]
        self.ndim = X.ndim + Y.ndim - len(X_dot_axes) - len(Y_dot_axes) - len(Y_batch_axes)
        super(_tensordot, self).__init__([X, Y])

    def _emit(self, backend, X, Y):
        # Batched contractions are defined by the einsum semantics
        # (bayesic/algebra.py:334-338); the reference's own batched execution is
        # broken three ways (:1358-1383) and is not reproduced.
        return backend.tensordot(X, Y, list(self.X_dot_axes), list(self.Y_dot_axes),
                                 list(self.X_batch_axes), list(self.Y_batch_axes))

    def _equality_by(self):
        return (self.parents,
                frozenset(zip(self.X_dot_axes, self.Y_dot_axes)),
                frozenset(zip(self.X_batch_axes, self.Y_batch_axes)))

    def __repr__(self):
        head = "%s(%r, %r, %r, %r" % (type(self).__name__, self.parents[0], self.parents[1],
                                      self.X_dot_axes, self.Y_dot_axes)
        if self.X_batch_axes:
            return head + ", %r, %r)" % (self.X_batch_axes, self.Y_batch_axes)
        return head + ")"


class _diagonal(Expression):
    """Diagonal of two axes; the diagonal becomes the LAST axis of the result."""

    def __init__(self, X, axis1, axis2):
        self.axis1 = axis1
        self.axis2 = axis2
        self.ndim = X.ndim - 1
        super(_diagonal, self).__init__([X])

    def _emit(self, backend, X):
        return backend.diagonal(X, self.axis1, self.axis2)

    def __repr__(self):
        return "%s(%r, %s, %s)" % (type(self).__name__, self.parents[0], self.axis1, self.axis2)

    def _equality_by(self):
        return (self.parents[0], frozenset([self.axis1, self.axis2]))


# ---------------------------------------------------------------------------
# element-wise sugar
# ---------------------------------------------------------------------------

@with_wrapped_literals
def div(X, Y):
    """X * Y**-1, so that division takes part in einsum algebra."""
    return mul(X, Y ** -1)


@with_wrapped_literals
def neg(X):
    return -1 * X


@with_wrapped_literals
def sub(self, other):
    return add(self, -other)


def log(X):
    return elemwise(OPS["log"], X)


def exp(X):
    return elemwise(OPS["exp"], X)


def pow(X, Y):
    return elemwise(OPS["pow"], X, Y)


def abs_(X):
    return elemwise(OPS["abs_"], X)


# ---------------------------------------------------------------------------
# operators and methods on every Expression
# ---------------------------------------------------------------------------

def _reflected(fn):
    def swapped(x, y):
        return fn(y, x)
    return swapped


def _add(*args):   # `add` is a class and cannot be bound as a method directly
    return add(*args)


Expression.__add__ = _add
Expression.__radd__ = _reflected(add)
Expression.__sub__ = sub
Expression.__rsub__ = _reflected(sub)
Expression.__mul__ = mul
Expression.__rmul__ = _reflected(mul)
Expression.__truediv__ = div
Expression.__rtruediv__ = _reflected(div)
Expression.__pow__ = pow
Expression.__rpow__ = _reflected(pow)
Expression.__matmul__ = dot
Expression.__rmatmul__ = _reflected(dot)
Expression.__neg__ = neg
Expression.__abs__ = abs_
Expression.T = property(transpose)
Expression.dimshuffle = dimshuffle
Expression.sum = sum
Expression.dot = dot
