"""Numeric backends for expression evaluation.

The reference evaluates an expression by asking every node for a Theano
variable (``Expression.apply`` / ``_apply_to_parents``, bayesic/algebra.py:34-61)
and compiling the graph with ``theano.function`` (:50-58).  Here a backend is an
object with one method per hook the reference's nodes call into Theano for
(SURVEY.md 8(b)): constant, shape, eye, elemwise, sum, mul, dimshuffle,
tensordot, diagonal -- plus ``from_host`` / ``to_host``.  ``evaluate`` walks the
(lowered, cached) tree once in post-order; there is no tracing compiler, so
``compile`` just binds the cached plan to a callable.

The only backend shipped with the package is the MI355X one
(bayesic_amd/algebra/device_backend.py).  There is NO CPU fallback: without a GPU
``resolve_backend(None)`` raises.  (The float64 numpy executor used by the CPU
tests lives in oracle/einsum_eval.py and is test infrastructure.)
"""


class Backend(object):
    name = "abstract"

    # -- hooks (one per Theano call site of the reference) --------------------
    def from_host(self, array, dtype, ndim):
        raise NotImplementedError

    def to_host(self, value):
        raise NotImplementedError

    def constant(self, value):
        raise NotImplementedError

    def shape(self, x, axis):
        raise NotImplementedError

    def eye(self, n):
        raise NotImplementedError

    def elemwise(self, op_name, *args):
        raise NotImplementedError

    def sum(self, x, axes):
        raise NotImplementedError

    def mul(self, *factors):
        raise NotImplementedError

    def dimshuffle(self, x, axes):
        raise NotImplementedError

    def tensordot(self, x, y, x_dot, y_dot, x_batch, y_batch):
        raise NotImplementedError

    def diagonal(self, x, axis1, axis2):
        raise NotImplementedError

    def broadcast_to(self, g, shape):
        """g (rank-less, or of len(shape) axes with extents 1 or full) as a value of `shape`,
        without arithmetic on any data-sized operand: a view with stride-0 axes where the
        backend has views.  Used by the reverse-mode derivative (algebra/autodiff.py)."""
        raise NotImplementedError

    def logdet(self, x):
        """log det over the trailing two axes (bayesic/distribution/core.py:50)."""
        raise NotImplementedError

    def inverse_spd(self, x):
        """Inverse of symmetric positive-definite matrices over the trailing two axes: a resident
        multivariate-normal / Wishart factor's natural parameters -> expectations (inference/vmp.py)."""
        raise NotImplementedError

    def softmax_rows(self, x):
        """(softmax over the LAST axis, log-sum-exp over the last axis): the expectation of a
        Categorical node whose natural parameters live on the backend (inference/vmp.py)."""
        raise NotImplementedError

    def materialize(self, value):
        """A backend value that can be kept (a deferred element-wise value is computed)."""
        return value

    # -- tree walk -------------------------------------------------------------
    def evaluate(self, expr, inputs, bindings=None):
        """Post-order walk.  ``bindings`` maps sub-expressions (by value) to input
        names: such a node is not computed but read from ``inputs`` -- how a mean-field
        update substitutes E[t(z)] for the statistic t(z) of another latent
        (bayesic_amd/inference/vmp.py)."""
        from .einsum_form import Einsum
        from .expr import var
        done = {}

        def visit(node):
            key = id(node)
            if key in done:
                return done[key]
            if bindings and not isinstance(node, var) and node in bindings:
                value = inputs[bindings[node]]
            elif isinstance(node, var):
                value = inputs[node.name]
            elif isinstance(node, Einsum):
                value = visit(node.lowered())
            else:
                value = node._emit(self, *[visit(p) for p in node.parents])
            done[key] = value
            return value

        try:
            return visit(expr)
        finally:
            # `visit` refers to itself, so `done` would otherwise keep every intermediate
            # value alive until the cyclic garbage collector runs
            done.clear()

    def compile(self, expr, bindings=None):
        """``bindings``: {sub-expression: input name}; the named inputs (float, of the
        sub-expression's ndim) replace those sub-expressions (see ``evaluate``)."""
        types = dict(expr.input_types)
        bindings = dict(bindings or {})
        for sub, name in bindings.items():
            types[name] = ("float32", sub.ndim)
        backend = self

        def device_fn(**device_inputs):
            return backend.evaluate(expr, device_inputs, bindings) if bindings \
                else backend.evaluate(expr, device_inputs)

        def f(**inputs):
            # inputs that only occur inside bound sub-expressions are not needed
            bound = {n: backend.from_host(inputs[n], *types[n]) for n in types if n in inputs}
            try:
                out = backend.to_host(device_fn(**bound))
            except KeyError as e:
                raise TypeError("missing input: %s" % e.args[0])
            if getattr(out, "ndim", 0) == 0 and expr.ndim > 0:
                # a broadcast scalar (e.g. einsum([], 2)): all axes are broadcastable
                out = out.reshape((1,) * expr.ndim)
            return out

        f.device_fn = device_fn
        f.backend = backend
        return f


_default = None


def set_default_backend(backend):
    global _default
    _default = backend


def resolve_backend(backend):
    """``backend`` itself, else the process default, else a new MI355X backend
    (raises when no GPU / library is available)."""
    global _default
    if backend is not None:
        return backend
    if _default is None:
        from .device_backend import DeviceBackend
        _default = DeviceBackend()
    return _default
