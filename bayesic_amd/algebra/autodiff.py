"""Reverse-mode derivatives of algebra expressions, computed with the backend's own primitives.

The reference plans a reparameterisation-trick estimator (README.md:51) but has no derivative
machinery of its own: it would have leaned on ``theano.grad``.  Here the lowered five-op tree
(bayesic/algebra.py:553-765) is walked once forwards, keeping every node's value, and once
backwards, and every vector-Jacobian product is itself one of the backend hooks -- ``mul``,
``sum``, ``dimshuffle``, ``tensordot``, ``eye``, ``elemwise`` -- so on the MI355X backend a gradient
is the same fused map-reduce launches and MFMA GEMMs as a forward evaluation, with nothing new
in the C ABI.

``value_and_grad(backend, expr, inputs, wrt)`` differentiates the SUM of ``expr``'s entries
(for a batch of independent samples along a leading axis that is the per-sample gradient).
"""
from .einsum_form import Einsum
from .expr import add, constant, elemwise, eye, shape, var
from .ops import _diagonal, _dimshuffle, _mul, _sum, _tensordot


def _shape_of(value):
    s = getattr(value, "shape", None)
    if s is None:
        return ()
    return tuple(int(d) for d in s)


class _Tape(object):
    def __init__(self, backend, needs=None):
        self.b = backend
        self.adjoint = {}
        self._needs = needs or {}      # id(node) -> does it depend on a variable we differentiate by?

    def needs(self, node):
        """Adjoints only flow to nodes that depend on a `wrt` input: the vector-Jacobian product
        towards a data operand (g W for dot(W, X.T): an N x D GEMM and a 1-GB store at config 2's
        size) is never formed."""
        return self._needs.get(id(node), True)

    # -- small helpers on backend values -------------------------------------------------
    def const(self, v):
        return self.b.constant(v)

    def plus(self, a, c):
        return self.b.elemwise("add", a, c)

    def times(self, *factors):
        return self.b.mul(*factors)

    def like(self, g, value):
        """g broadcast to the full shape of `value` (g is rank-less, or has the same rank with
        extents 1 or full).  From the SHAPE only: arithmetic with the forward value (g + 0*value)
        would re-read a data-sized operand per sum VJP and turn a non-finite forward value into a
        NaN gradient where the true gradient is finite or inf."""
        if _shape_of(g) == _shape_of(value):
            return g
        return self.b.broadcast_to(g, _shape_of(value))

    def fit(self, term, parent_value):
        """The adjoint contribution `term` in the form the parent needs: a backend may keep a
        broadcast scalar rank-less (the MI355X backend does), so ranks are not assumed."""
        ps = _shape_of(parent_value)
        if not ps:
            return _total(self, term)
        if not _shape_of(term):
            return term                      # a uniform adjoint; materialised where it is used
        return self.reduce_to(term, ps)

    def reduce_to(self, g, target_shape):
        """Sum g over the axes along which the operand it belongs to was broadcast."""
        gs = _shape_of(g)
        if len(gs) != len(target_shape):
            # a host scalar adjoint for an array operand (or the reverse) cannot happen below:
            # every VJP keeps the operand's rank
            raise AssertionError("rank mismatch in reduce_to: %r vs %r" % (gs, target_shape))
        axes = [i for i, (a, t) in enumerate(zip(gs, target_shape)) if t == 1 and a != 1]
        if not axes:
            return g
        reduced = self.b.sum(g, axes)
        pattern, k = [], 0
        for i in range(len(gs)):
            if i in axes:
                pattern.append("x")
            else:
                pattern.append(k)
                k += 1
        return self.b.dimshuffle(reduced, pattern)

    def accumulate(self, node, g):
        if not self.needs(node):
            return
        key = id(node)
        self.adjoint[key] = g if key not in self.adjoint else self.plus(self.adjoint[key], g)


def value_and_grad(backend, expr, inputs, wrt):
    """Returns (value, {name: d sum(value) / d input[name]}) as backend values.

    inputs: {name: backend value} (as for ``Backend.evaluate``); wrt: input names."""
    order, values, lowered_of, needs = [], {}, {}, {}
    wrt = list(wrt)

    def forward(node):
        key = id(node)
        if key in values:
            return values[key]
        if isinstance(node, var):
            value = inputs[node.name]
            needs[key] = node.name in wrt
        elif isinstance(node, Einsum):
            low = node.lowered()
            lowered_of[key] = low
            value = forward(low)
            needs[key] = needs[id(low)]
        else:
            value = node._emit(backend, *[forward(p) for p in node.parents])
            needs[key] = any(needs[id(p)] for p in node.parents)
        values[key] = value
        order.append(node)
        return value

    out = forward(expr)
    tape = _Tape(backend, needs)
    out_shape = _shape_of(out)
    seed = tape.const(1.0)
    if out_shape:
        seed = tape.like(seed, out)
    tape.adjoint[id(expr)] = seed

    grads = {}
    for node in reversed(order):
        g = tape.adjoint.pop(id(node), None)
        if g is None:
            continue
        if isinstance(node, var):
            if node.name in wrt:
                full = tape.like(g, values[id(node)]) if _shape_of(values[id(node)]) else g
                grads[node.name] = full if node.name not in grads else tape.plus(grads[node.name], full)
            continue
        if isinstance(node, (constant, shape, eye)):
            continue
        if isinstance(node, Einsum):
            tape.accumulate(lowered_of[id(node)], g)
            continue
        parents = node.parents
        pv = [values[id(p)] for p in parents]
        if _shape_of(values[id(node)]) and not _shape_of(g):
            g = tape.like(g, values[id(node)])           # a rank-less uniform adjoint: give it the node's shape
        if isinstance(node, add):
            for p, v in zip(parents, pv):
                if tape.needs(p):
                    tape.accumulate(p, tape.fit(g, v))
        elif isinstance(node, _mul):
            for i, (p, v) in enumerate(zip(parents, pv)):
                if not tape.needs(p):
                    continue
                rest = [w for j, w in enumerate(pv) if j != i]
                term = tape.times(g, *rest) if rest else g
                tape.accumulate(p, tape.fit(term, v))
        elif isinstance(node, _sum):
            (p,), (v,) = parents, pv
            pattern, k = [], 0
            for i in range(len(_shape_of(v))):
                if i in node.axes:
                    pattern.append("x")
                else:
                    pattern.append(k)
                    k += 1
            tape.accumulate(p, tape.like(backend.dimshuffle(g, pattern), v))
        elif isinstance(node, _dimshuffle):
            (p,), (v,) = parents, pv
            if not _shape_of(g) or not _shape_of(v):
                tape.accumulate(p, tape.fit(g, v))             # a broadcast scalar kept rank-less
                continue
            xs = [k for k, a in enumerate(node.axes) if a == "x"]
            g1 = backend.sum(g, xs) if xs else g
            kept = [a for a in node.axes if a != "x"]          # source axis of each remaining axis
            inverse = [kept.index(a) for a in range(len(kept))]
            g2 = backend.dimshuffle(g1, inverse) if inverse != list(range(len(kept))) else g1
            tape.accumulate(p, tape.reduce_to(g2, _shape_of(v)))
        elif isinstance(node, _tensordot):
            _tensordot_vjp(tape, backend, node, parents, pv, g)
        elif isinstance(node, _diagonal):
            (p,), (v,) = parents, pv
            nd = len(_shape_of(v))
            a1, a2 = node.axis1, node.axis2
            others = [a for a in range(nd) if a not in (a1, a2)]
            # g axes: others..., diagonal last
            g_pat = [None] * nd
            for k, a in enumerate(others):
                g_pat[a] = k
            g_pat[a1], g_pat[a2] = len(others), "x"
            e_pat = ["x"] * nd
            e_pat[a1], e_pat[a2] = 0, 1
            identity = backend.eye(backend.shape(v, a1))
            tape.accumulate(p, tape.times(backend.dimshuffle(g, g_pat),
                                          backend.dimshuffle(identity, e_pat)))
        elif isinstance(node, elemwise):
            _elemwise_vjp(tape, backend, node, parents, pv, values[id(node)], g)
        else:
            raise NotImplementedError("no derivative rule for %s" % type(node).__name__)
    for name in wrt:
        if name not in grads:
            raise KeyError("the expression does not depend on %r" % name)
    return out, grads


def _total(tape, g):
    gs = _shape_of(g)
    return tape.b.sum(g, list(range(len(gs)))) if gs else g


def _elemwise_vjp(tape, backend, node, parents, pv, y, g):
    name = node.op.scalar_op.name
    x = pv[0]
    if name == "log":
        term = tape.times(g, backend.elemwise("pow", x, tape.const(-1.0)))
    elif name == "exp":
        term = tape.times(g, y)
    elif name == "abs_":
        inv = backend.elemwise("pow", y, tape.const(-1.0))
        term = tape.times(g, x, inv)                       # sign(x); undefined at 0 like |x|' itself
    elif name == "pow":
        e = pv[1]
        if parents[1].input_types:
            raise NotImplementedError("pow with a non-constant exponent is not differentiated")
        em1 = tape.plus(e, tape.const(-1.0))
        term = tape.times(g, e, backend.elemwise("pow", x, em1))
    else:
        raise NotImplementedError("no derivative rule for element-wise %s" % name)
    tape.accumulate(parents[0], tape.fit(term, x))


def _tensordot_vjp(tape, backend, node, parents, pv, g):
    X, Y = pv
    xd, yd = list(node.X_dot_axes), list(node.Y_dot_axes)
    xb, yb = list(node.X_batch_axes), list(node.Y_batch_axes)
    xo, yo = list(node.X_other_axes), list(node.Y_other_axes)
    nb, nxo, nyo = len(xb), len(xo), len(yo)
    if not _shape_of(g):
        # a full contraction (scalar result): dX is g Y with Y's axes put in X's order, and
        # the other way round -- no tensordot with a rank-less operand
        px = [yd[xd.index(a)] for a in range(len(_shape_of(X)))]
        py = [xd[yd.index(a)] for a in range(len(_shape_of(Y)))]
        if tape.needs(parents[0]):
            rx = tape.times(g, Y)
            tape.accumulate(parents[0], backend.dimshuffle(rx, px) if px != sorted(px) else rx)
        if tape.needs(parents[1]):
            ry = tape.times(g, X)
            tape.accumulate(parents[1], backend.dimshuffle(ry, py) if py != sorted(py) else ry)
        return
    g_batch = list(range(nb))
    g_xo = list(range(nb, nb + nxo))
    g_yo = list(range(nb + nxo, nb + nxo + nyo))
    if tape.needs(parents[0]):
        # dX = g . Y over Y's free axes: result axes = batch, X others, Y's dot axes (ascending)
        r = backend.tensordot(g, Y, g_yo, yo, g_batch, yb)
        yd_sorted = sorted(yd)
        pattern = []
        for a in range(len(_shape_of(X))):
            if a in xb:
                pattern.append(xb.index(a))
            elif a in xo:
                pattern.append(nb + xo.index(a))
            else:
                pattern.append(nb + nxo + yd_sorted.index(yd[xd.index(a)]))
        tape.accumulate(parents[0], backend.dimshuffle(r, pattern)
                        if pattern != list(range(len(pattern))) else r)
    if tape.needs(parents[1]):
        # dY = X . g over X's free axes: result axes = batch, X's dot axes (ascending), Y others
        r = backend.tensordot(X, g, xo, g_xo, xb, g_batch)
        xd_sorted = sorted(xd)
        pattern = []
        for a in range(len(_shape_of(Y))):
            if a in yb:
                pattern.append(yb.index(a))
            elif a in yo:
                pattern.append(nb + len(xd) + yo.index(a))
            else:
                pattern.append(nb + xd_sorted.index(xd[yd.index(a)]))
        tape.accumulate(parents[1], backend.dimshuffle(r, pattern)
                        if pattern != list(range(len(pattern))) else r)
