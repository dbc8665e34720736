"""Expression tree: symbolic tensors with value semantics.

Behavioural contract: bayesic/algebra.py:17-290 (Expression, var, constant,
shape, elemwise, add, eye and the literal/broadcast helpers).  Unlike the
reference, nodes never touch a numeric library: evaluation goes through an
explicit backend object (bayesic_amd/algebra/backend.py), which is where the
MI355X path plugs in (the reference inlines Theano calls at
bayesic/algebra.py:42-58,132-134,155,202,217,258).
"""
import numpy as np


class Expression(object):
    """Node of a symbolic tensor expression.  Subclasses define ``ndim`` and
    ``_emit(backend, *parent_values)``."""

    def __init__(self, parents):
        self.parents = tuple(parents)

    # -- typing ------------------------------------------------------------
    @property
    def input_types(self):
        """{input name: (dtype, ndim)} over the whole tree
        (bayesic/algebra.py:22-30)."""
        merged = {}
        for parent in self.parents:
            for name, type_ in parent.input_types.items():
                if name in merged and merged[name] != type_:
                    raise TypeError("same input %s occurs with different types %s, %s"
                                    % (name, type_, merged[name]))
                merged[name] = type_
        return merged

    @property
    def shape(self):
        return tuple(shape(self, axis) for axis in range(self.ndim))

    @property
    def size(self):
        from .ops import mul
        return mul(*self.shape)

    # -- evaluation ----------------------------------------------------------
    def apply(self, inputs, backend=None):
        """Evaluate with ``inputs`` = {name: backend value}
        (bayesic/algebra.py:34-40); post-order, parents first."""
        from .backend import resolve_backend
        backend = resolve_backend(backend)
        return backend.evaluate(self, inputs)

    def _emit(self, backend, *parent_values):
        raise NotImplementedError

    def compile(self, backend=None, **options):
        """``f(**{name: array}) -> ndarray`` (bayesic/algebra.py:50-58).  The
        lowered plan is built once here, not on every call; ``f.device_fn`` is
        the analogue of the reference's ``f.theano_fn`` handle.  ``options`` go to the
        backend's ``compile`` (the MI355X backend: ``graph=True`` records ``f.device_fn``'s
        launches as a hipGraph)."""
        from .backend import resolve_backend
        return resolve_backend(backend).compile(self, **options)

    # -- printing ----------------------------------------------------------
    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join(repr(p) for p in self.parents))

    def bracketed_repr(self):
        return repr(self)

    def terms(self):
        """Summands; ``self`` equals their sum."""
        return [self]

    # -- value semantics -------------------------------------------------------
    def _equality_by(self):
        return self.parents

    def __eq__(self, other):
        return isinstance(other, self.__class__) and self._equality_by() == other._equality_by()

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        # expressions are immutable once built; the executor looks every node of a tree up in the
        # bindings of each evaluation, so the structural hash is computed once per object
        h = self.__dict__.get("_hash_value")
        if h is None:
            h = self.__dict__["_hash_value"] = hash(self._equality_by())
        return h


class var(Expression):
    """Named symbolic input (bayesic/algebra.py:108-126)."""

    def __init__(self, name, ndim, dtype="float32"):
        self.name = name
        self.ndim = ndim
        self.dtype = dtype
        super(var, self).__init__([])

    @property
    def input_types(self):
        return {self.name: (self.dtype, self.ndim)}

    def _emit(self, backend):
        raise KeyError(self.name)  # inputs are bound by the backend, never emitted

    def __repr__(self):
        return self.name

    def _equality_by(self):
        return self.name


class constant(Expression):
    """Literal scalar or array (bayesic/algebra.py:129-144).  dtype follows numpy
    (the reference's came from Theano's constant typing, which no test pins)."""

    def __init__(self, value):
        self.value = value
        array = np.asarray(value)
        self.ndim = array.ndim
        self.dtype = str(array.dtype)
        super(constant, self).__init__([])

    def _emit(self, backend):
        return backend.constant(self.value)

    def __repr__(self):
        return repr(self.value)

    def _equality_by(self):
        return self.value

    def __eq__(self, other):
        if not isinstance(other, constant):
            return False
        if isinstance(self.value, np.ndarray) or isinstance(other.value, np.ndarray):
            return self.value is other.value or bool(np.array_equal(self.value, other.value))
        return self.value == other.value

    def __hash__(self):
        if isinstance(self.value, np.ndarray):
            return hash((self.value.shape, self.value.dtype.str, self.value.tobytes()))
        return hash(self.value)


class shape(Expression):
    """Extent of one axis of an expression, as a scalar (bayesic/algebra.py:147-161)."""
    ndim = 0

    def __init__(self, expression, axis):
        super(shape, self).__init__([expression])
        self.axis = axis

    def _emit(self, backend, value):
        return backend.shape(value, self.axis)

    def _equality_by(self):
        return (self.parents[0], self.axis)

    def __repr__(self):
        return "%s.shape[%d]" % (self.parents[0].bracketed_repr(), self.axis)


def wrap_if_literal(x):
    if np.isscalar(x) or isinstance(x, np.ndarray):
        return constant(x)
    if isinstance(x, Expression):
        return x
    raise ValueError("must be a scalar, numpy array or Expression")


def with_wrapped_literals(fn):
    def wrapped_fn(*args):
        return fn(*(wrap_if_literal(x) for x in args))
    wrapped_fn.__name__ = getattr(fn, "__name__", "wrapped_fn")
    wrapped_fn.__doc__ = fn.__doc__
    return wrapped_fn


def autobroadcast_or_match(X, ndim):
    """Scalars broadcast up to ``ndim``; anything else must already match
    (bayesic/algebra.py:179-192)."""
    if X.ndim == ndim:
        return X
    if X.ndim == 0:
        from .ops import dimshuffle
        return dimshuffle(X, *(["x"] * ndim))
    raise ValueError(
        "Dimension mismatch, was %d, should be %d. If you want broadcasting "
        "you need to do it explicitly via dimshuffle" % (X.ndim, ndim))


class ElementwiseOp(object):
    """Backend-neutral stand-in for the op object the reference passes to
    ``elemwise`` (it reads ``theano_op.scalar_op.name``, bayesic/algebra.py:202)."""

    class _ScalarOp(object):
        def __init__(self, name):
            self.name = name

    def __init__(self, name):
        self.name = name
        self.scalar_op = ElementwiseOp._ScalarOp(name)

    def __repr__(self):
        return "<elementwise %s>" % self.name


OPS = {name: ElementwiseOp(name) for name in ("add", "mul", "log", "exp", "pow", "abs_")}


class elemwise(Expression):
    """Opaque element-wise node (bayesic/algebra.py:195-209).  ``op`` is an
    ElementwiseOp (or any object with ``.scalar_op.name``); arguments are
    literal-wrapped and scalars auto-broadcast."""

    def __init__(self, op, *args, name=None):
        args = [wrap_if_literal(x) for x in args]
        self.ndim = max(arg.ndim for arg in args)
        args = [autobroadcast_or_match(arg, self.ndim) for arg in args]
        self.op = op
        self.name = name or op.scalar_op.name
        super(elemwise, self).__init__(args)

    def _emit(self, backend, *values):
        return backend.elemwise(self.op.scalar_op.name, *values)

    def __repr__(self):
        return "%s(%s)" % (self.name, ", ".join(repr(p) for p in self.parents))

    def _equality_by(self):
        return (self.op.scalar_op.name, self.parents)


class add(elemwise):
    """n-ary sum; nested sums are flattened and equality ignores term order
    (bayesic/algebra.py:212-233)."""

    def __init__(self, *summands):
        flat = [t for s in summands for t in wrap_if_literal(s).terms()]
        super(add, self).__init__(OPS["add"], *flat)

    def terms(self):
        return self.parents

    def _emit(self, backend, *values):
        return backend.elemwise("add", *values)

    def __repr__(self):
        return " + ".join(repr(p) for p in self.parents)

    def bracketed_repr(self):
        return "(%r)" % self

    def _equality_by(self):
        return frozenset(self.parents)


class eye(Expression):
    """Square identity matrix whose size is given by one or more shape
    expressions known to be equal at run time; two eyes are equal when their
    shape sets overlap (bayesic/algebra.py:236-290)."""
    ndim = 2

    def __init__(self, *shapes):
        if len(shapes) == 0:
            raise ValueError("need at least one shape for eye")
        super(eye, self).__init__([wrap_if_literal(s) for s in shapes])

    def _emit(self, backend, first, *_):
        return backend.eye(first)

    def __eq__(self, other):
        return isinstance(other, self.__class__) and \
            len(set(self.parents) & set(other.parents)) > 0

    def __hash__(self):
        return hash(self.__class__)  # must not depend on which shapes are listed
