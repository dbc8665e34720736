"""Recognising, in a symbolic log-joint, the data-sized structure one of the fused kernels computes.

The plugin surface of the path is the reference's: a model is written with Distribution nodes
(bayesic/distribution/base.py:9-172) and ``bayesic.algebra`` expressions, and evaluated through
``Expression.compile()`` (bayesic/algebra.py:42-58).  Evaluated literally, config 2's log-joint is two
skinny products over X and a few dozen [S, N] element-wise launches (0.85 ms at 1M x 256); the same
numbers come out of ONE pass over X (csrc/bsc_blr.hip) if one knows that the data enter only through

    Q_s = sum_n (y_n - x_n . w_s)^2      (and  G_s = sum_n (y_n - x_n . w_s) x_n  for the gradient).

``match`` (bayesic/algebra.py:1037-1063) is the tool the reference built for pulling the coefficient of a
statistic out of a multilinear term; here the "statistics" are the three data-sized contractions a
Gaussian-linear likelihood expands into,

    sum_nd y_n X_nd W_sd,     sum_nde X_nd W_sd X_ne W_se,     sum_n y_n^2,

and the coefficients c1, c2, c3 must stand in the ratio -2 : 1 : 1 for the three to be c2 * Q_s.  What is
left of the log-joint is parameter-sized.  Whether THAT belongs to the family the fused finish kernel
computes (``bsc_blr_fused_update_general``) is decided by fitting the family's five numbers at a few
probe points and checking the fit at random others, in host float64 (``_param_backend``) -- identity
testing by evaluation, which does not care in which of many equivalent ways the model was written
(``exp(-xi)``, ``1 / exp(xi)``, ``pow(var, -1)`` ...).
"""
import numpy as np

from .. import algebra as A
from ..algebra.einsum_form import OUT, SUM, Einsum
from ..algebra.expr import Expression, add, constant, elemwise, eye, shape, var
from ._param_backend import ParameterBackend, ShapeOnly
from .conjugacy import _carried_axes, expand_terms

_PROBE = ParameterBackend()

# What "this is not the model I compute" looks like when it surfaces as an exception: a term that is not linear in a
# statistic (conjugacy.NotConjugate, a ValueError), shapes that do not fit, an input the probe cannot evaluate, a fit
# that divides by zero.  ANYTHING ELSE raised inside a recogniser is a bug in the recogniser or in the derived-update
# code beneath it and propagates: the engines turn it into a RuntimeWarning under route="auto" (the model still runs,
# on the general route, and says why) and re-raise it under route="fused".
NOT_THIS_MODEL = (ValueError, KeyError, TypeError, IndexError, ZeroDivisionError, FloatingPointError, np.linalg.LinAlgError)


def guarded_route(try_route, strict):
    """Run an engine's ``_try_fused_route`` (returns None = routed, or the reason it was not).  A recogniser declines by
    returning; an exception out of it is a bug.  ``strict`` (route="fused"): it propagates.  Otherwise (route="auto")
    the model still runs on its general route, but not silently: a RuntimeWarning names the exception, and the
    engine's ``route_reason`` keeps it."""
    if strict:
        return try_route()
    try:
        return try_route()
    except Exception as e:      # noqa: BLE001 -- reported (warning + route_reason), not swallowed
        import warnings
        reason = "recognition FAILED with %s: %s -- a bug, not a verdict on the model" % (type(e).__name__, e)
        warnings.warn("bayesic_amd: route='auto' falls back to the general route because %s" % reason, RuntimeWarning,
                      stacklevel=3)
        return reason


def _say(why, message):
    """Record why a recogniser declined (``why``: a list the caller passed, or None)."""
    if why is not None:
        why.append(message)
    return None


# ---- small tree utilities ------------------------------------------------------------------------

def rewrite(expr, fn):
    """Bottom-up reconstruction: ``fn(node_with_rewritten_parents)`` may return a replacement or None."""
    expr = A.wrap_if_literal(expr)
    if isinstance(expr, Einsum):
        rebuilt = A.einsum([(rewrite(f, fn), idx) for f, idx in expr.factors_and_indices], expr.ndim)
    elif isinstance(expr, add):
        rebuilt = add(*[rewrite(p, fn) for p in expr.parents])
    elif isinstance(expr, elemwise):
        rebuilt = elemwise(expr.op, *[rewrite(p, fn) for p in expr.parents], name=expr.name)
    elif isinstance(expr, shape):
        rebuilt = shape(rewrite(expr.parents[0], fn), expr.axis)
    elif isinstance(expr, eye):
        rebuilt = eye(*[rewrite(p, fn) for p in expr.parents])
    else:
        rebuilt = expr
    replaced = fn(rebuilt)
    return rebuilt if replaced is None else replaced


def static_extent(expr, axis, shapes):
    """Extent of one axis from the shapes of the inputs alone (no evaluation); 1 for a broadcast axis."""
    if isinstance(expr, var):
        return int(shapes[expr.name][axis])
    if isinstance(expr, constant):
        return int(np.shape(expr.value)[axis])
    if isinstance(expr, Einsum):
        for factor, indices in expr.factors_and_indices:
            real = _carried_axes(factor)
            for ax, (kind, n) in enumerate(indices):
                if kind == OUT and n == axis and ax in real:
                    return static_extent(factor, ax, shapes)
        return 1
    if isinstance(expr, elemwise):
        return max(static_extent(p, axis, shapes) for p in expr.parents)
    if isinstance(expr, eye):
        return int(round(float(_PROBE.evaluate(_without_shapes(expr.parents[0], shapes), {}))))
    raise ValueError("cannot tell the extent of %r" % (expr,))


def _without_shapes(expr, shapes):
    """``shape(e, axis)`` nodes replaced by the number they stand for."""
    return rewrite(expr, lambda node: constant(float(static_extent(node.parents[0], node.axis, shapes)))
                   if isinstance(node, shape) else None)


def _constant_value(expr):
    """The value of an expression made of constants only (a broadcast literal such as the exponent of
    ``x ** 2``), or None."""
    if expr.input_types:
        return None
    try:
        value = np.asarray(_PROBE.evaluate(expr, {}), np.float64)
    except (ValueError, KeyError):
        return None
    return float(value.reshape(-1)[0]) if value.size and np.all(value == value.reshape(-1)[0]) else None


def normalise(expr, shapes):
    """Shapes resolved to numbers; ``pow(E, 2)`` written as the product ``E * E`` -- the front end keeps
    ``pow`` opaque (bayesic/algebra.py:1435-1448), so the square of a mean would hide the contraction
    inside it from the einsum form."""
    def step(node):
        if isinstance(node, shape):
            return constant(float(static_extent(node.parents[0], node.axis, shapes)))
        if isinstance(node, elemwise) and not isinstance(node, add) and node.op.scalar_op.name == "pow" \
                and _constant_value(node.parents[1]) == 2.0:
            return A.mul(node.parents[0], node.parents[0])
        return None
    return rewrite(expr, step)


def _names(expr):
    return set(expr.input_types)


# ---- the Gaussian-linear data term -------------------------------------------------------------

class GaussianLinear(object):
    """What ``gaussian_linear`` found: the data enter the log-joint only as ``coefficient_s * Q_s``.

    X, y, W      : names of the design matrix [N, D], the targets [N] and the latent weights [S, D]
    coefficient  : expression [S] over the other latents (c2 above; for a Normal likelihood
                   -scale / (2 variance_s))
    rest         : the parameter-sized terms, a list of [S] expressions (shapes resolved)
    surrogate    : rest + coefficient * Q with ``Q`` the input ``Q_NAME`` [S]: the whole log-joint per draw
                   as a parameter-sized expression
    family       : (c0, c_xi, s_q, k_w, beta, xi_name) when the surrogate is
                   c0 + c_xi xi + e^{-xi} (-s_q Q / 2 - k_w |w|^2 / 2 - beta) for the one other latent xi
                   [S, 1] -- what bsc_blr_fused_update_general computes --, else None
    """
    Q_NAME = "_gl_Q"

    def __init__(self, X, y, W, coefficient, rest, family=None):
        self.X, self.y, self.W = X, y, W
        self.coefficient, self.rest, self.family = coefficient, rest, family
        Q = var(self.Q_NAME, 1)
        self.surrogate = add(*(list(rest) + [coefficient * Q])) if rest else coefficient * Q


def _templates(X, y, W):
    s, n, d, e = (OUT, 0), (SUM, 0), (SUM, 1), (SUM, 2)
    out = []
    for slot_nd in (1, 2):
        Z = var("_gl_slot", slot_nd)
        z = [s] if slot_nd == 1 else [s, n]
        out.append((Z,
                    Einsum([(y, [n]), (X, [n, d]), (W, [s, d]), (Z, z)], 1),
                    Einsum([(X, [n, d]), (W, [s, d]), (X, [n, e]), (W, [s, e]), (Z, z)], 1),
                    Einsum([(y, [n]), (y, [n]), (Z, z)], 1)))
    return out


def _coefficient(term, templates, which, forbidden):
    """Coefficient [S] of data statistic ``which`` (0, 1, 2) in ``term``, or None."""
    for Z, *stats in templates:
        c = A.match(term, stats[which], Z)
        if c is None or (_names(c) & forbidden):
            continue
        if Z.ndim == 2:
            # the coefficient was broadcast along the rows (a variance written [S, 'x']): it must not
            # vary along them; summing the extent-1 axis takes it back to [S]
            c = A.wrap_if_literal(c)
            if 1 in _carried_axes(Einsum._wrap_if_not_einsum(c)):
                continue
            c = A.sum(c, axis=1)
        return c
    return None


def _evaluate(expr, latents, S, rng, extra=None):
    inputs = {v.name: rng.standard_normal((S, n)) * 0.7 for v, n in latents}
    if extra:
        inputs.update(extra)
    return np.asarray(_PROBE.evaluate(expr, inputs), np.float64), inputs


def gaussian_linear(log_joint, latents, data_shapes, n_samples, why=None):
    """``log_joint``: expression of ndim 1 (one value per draw); ``latents``: [(var [S, size], size)];
    ``data_shapes``: {data input name: shape}.  Returns a ``GaussianLinear`` or None -- None means "not
    this structure": the caller then evaluates the expression as written, and ``why`` (a list, optional)
    receives the reason.  Exceptions outside ``NOT_THIS_MODEL`` are bugs and propagate."""
    try:
        return _gaussian_linear(log_joint, latents, data_shapes, n_samples, why)
    except NOT_THIS_MODEL as e:
        return _say(why, "%s while reading the log-joint: %s" % (type(e).__name__, e))


def _gaussian_linear(log_joint, latents, data_shapes, n_samples, why):
    shapes = dict(data_shapes)
    shapes.update({v.name: (n_samples, n) for v, n in latents})
    # (distributing over an add can introduce extents of its own -- shape nodes -- for summands that
    # were broadcast inside it: resolved again afterwards)
    terms = [_without_shapes(t, shapes) for t in expand_terms(normalise(log_joint, shapes))]
    data_names = set(data_shapes)
    data_terms = [t for t in terms if _names(t) & data_names]
    rest = [t for t in terms if not (_names(t) & data_names)]
    if not data_terms:
        return _say(why, "no term of the log-joint mentions a data input")
    if any(t.ndim != 1 for t in terms):
        return _say(why, "a term of the expanded log-joint is not one value per draw (ndim 1)")
    types = log_joint.input_types
    rng = np.random.RandomState(20240)
    reason = "no (matrix [N, D], vector [N], latent [S, D]) triple among the inputs"
    for Xn in sorted(n for n in data_names if types.get(n, (None, 0))[1] == 2):
        for yn in sorted(n for n in data_names if types.get(n, (None, 0))[1] == 1
                         and data_shapes[n][0] == data_shapes[Xn][0]):
            for Wv, width in latents:
                if width != data_shapes[Xn][1]:
                    continue
                X, y = var(Xn, 2, types[Xn][0]), var(yn, 1, types[yn][0])
                templates = _templates(X, y, Wv)
                forbidden = {Xn, yn, Wv.name}
                found = [[], [], []]
                for term in data_terms:
                    for which in (1, 0, 2):                 # the quadratic form first: it contains the others' factors
                        c = _coefficient(term, templates, which, forbidden)
                        if c is not None:
                            found[which].append(c)
                            break
                    else:
                        reason = ("the data term %r is none of sum_nd y X W, sum_nde X W X W, sum_n y^2 times a "
                                  "parameter-sized coefficient (X = %s, y = %s, W = %s)" % (term, Xn, yn, Wv.name))
                        break
                else:
                    if not all(found):
                        names = ("sum_nd y_n X_nd W_sd", "sum_nde X_nd W_sd X_ne W_se", "sum_n y_n^2")
                        reason = "the log-joint has no term in %s (X = %s, y = %s, W = %s)" % (
                            ", ".join(n for n, f in zip(names, found) if not f), Xn, yn, Wv.name)
                        continue
                    c1, c2, c3 = (cs[0] if len(cs) == 1 else add(*cs) for cs in found)
                    others = [(v, n) for v, n in latents if v.name != Wv.name]
                    ok = True
                    for _ in range(3):          # c1 = -2 c2 and c3 = c2 as functions of the other latents
                        v2, inputs = _evaluate(c2, others, 3, rng)
                        v1 = np.asarray(_PROBE.evaluate(c1, inputs), np.float64)
                        v3 = np.asarray(_PROBE.evaluate(c3, inputs), np.float64)
                        tol = 1e-12 * np.abs(v2).max()
                        if not (np.all(np.abs(v1 + 2.0 * v2) <= tol) and np.all(np.abs(v3 - v2) <= tol)
                                and np.all(v2 <= 0.0)):
                            ok = False
                    if not ok:
                        reason = ("the coefficients of the three data contractions do not stand in the ratio -2 : 1 : 1 "
                                  "with a negative quadratic one: not a multiple of sum_n (y_n - x_n . w_s)^2")
                        continue
                    plan = GaussianLinear(Xn, yn, Wv.name, c2, rest)
                    plan.family = _fit_family(plan, Wv, width, others, rng)
                    return plan
    return _say(why, reason)


def _fit_family(plan, Wv, D, others, rng):
    """The five numbers of  f = c0 + c_xi xi + e^{-xi} (-s_q Q / 2 - k_w |w|^2 / 2 - beta)  from probe
    evaluations of the surrogate, verified at random points; None when the surrogate is not of that form."""
    if len(others) != 1 or others[0][1] != 1:
        return None
    xi_v = others[0][0]

    def f(w, xi, q):
        w = np.atleast_2d(np.asarray(w, np.float64))
        S = w.shape[0]
        inputs = {Wv.name: w, xi_v.name: np.full((S, 1), 0.0) + np.reshape(xi, (-1, 1)),
                  GaussianLinear.Q_NAME: np.zeros(S) + q}
        return np.asarray(_PROBE.evaluate(plan.surrogate, inputs), np.float64).reshape(-1)

    try:
        zero = np.zeros((1, D))
        t = 2.0
        g0, g1, g2 = f(zero, [0.0], 0.0)[0], f(zero, [t], 0.0)[0], f(zero, [-t], 0.0)[0]
        # g(xi) = c0 + c_xi xi - beta e^{-xi} at xi = 0, t, -t:
        #   g1 + g2 - 2 g0 = -beta (e^{-t} + e^{t} - 2),   g1 - g2 = 2 c_xi t + beta (e^{t} - e^{-t})
        beta = -((g1 + g2) - 2.0 * g0) / (np.exp(-t) + np.exp(t) - 2.0)
        c_xi = ((g1 - g2) - beta * (np.exp(t) - np.exp(-t))) / (2.0 * t)
        c0 = g0 + beta
        s_q = -2.0 * (f(zero, [0.0], 1.0)[0] - g0)
        unit = np.zeros((1, D))
        unit[0, 0] = 1.0
        k_w = -2.0 * (f(unit, [0.0], 0.0)[0] - g0)
        if not (np.isfinite([c0, c_xi, s_q, k_w, beta]).all() and s_q > 0.0 and k_w >= 0.0):
            return None
        for _ in range(4):
            S = 3
            w = rng.standard_normal((S, D))
            xi = rng.standard_normal(S) * 0.8
            q = rng.uniform(0.5, 50.0, S)
            e = np.exp(-xi)
            want = c0 + c_xi * xi + e * (-0.5 * s_q * q - 0.5 * k_w * (w * w).sum(axis=1) - beta)
            got = f(w, xi, q)
            scale = np.abs(c0) + np.abs(c_xi * xi) + e * (0.5 * s_q * q + 0.5 * k_w * (w * w).sum(axis=1) + abs(beta))
            if not np.all(np.abs(got - want) <= 1e-10 * scale):
                return None
    except (ValueError, KeyError, FloatingPointError):
        return None
    return (float(c0), float(c_xi), float(s_q), float(k_w), float(beta), xi_v.name)


# ---- the diagonal Gaussian mixture (config 3) ---------------------------------------------------

def _mog_message_fused(R, X, K, D):
    """The natural-parameter increment svi/mog.py's kernels apply, from responsibilities: fused layout
    [sum r (K) | R^T X (K D) | sum r (K D) | sum r (K D) | R^T X^2 (K D)]."""
    Rk = R.sum(axis=0)
    Rkd = np.repeat(Rk[:, None], D, axis=1)
    return np.concatenate([Rk, (R.T @ X).ravel(), Rkd.ravel(), Rkd.ravel(), (R.T @ (X * X)).ravel()])


def diagonal_mixture(log_joint, Z, pi, ng_vars, X_name, K, D, scale, dtype="float64", why=None):
    """Is the mean-field update DERIVED from ``log_joint`` (conjugacy.conjugate_coefficients: every message a
    coefficient ``match`` pulled out of a term) the update csrc/bsc_mog.hip computes -- a Categorical local
    latent ``Z`` [N, K] whose logits are E[log pi_k] + sum_d (E[tau mu] x - E[tau] x^2 / 2 + E[log tau] / 2 -
    E[tau mu^2] / 2), a Dirichlet ``pi`` and Normal-Gamma factors (``ng_vars`` = the four statistic variables)
    receiving [sum r | R^T X | R^T X^2] times ``scale`` on top of their priors?  Decided by running the derived
    update rules on a six-row instance in host float64 and comparing with that closed form at random
    parameters: identity testing by evaluation.  Returns the prior's natural parameters in the fused layout
    [alpha0 - 1 | kappa0 m0 | kappa0 | 2 a0 - 1 | 2 b0 + kappa0 m0^2], or None."""
    from scipy.special import digamma
    from .vmp import CategoricalNode, DirichletNode, MeanFieldVMP, NormalGammaNode
    rng = np.random.RandomState(77)
    n = 6
    eta0 = None
    try:
        for trial in range(2):
            X = rng.standard_normal((n, D)) * 1.3
            alpha = rng.uniform(0.5, 4.0, K)
            m, kappa = rng.standard_normal((K, D)), rng.uniform(0.5, 3.0, (K, D))
            a, b = rng.uniform(1.0, 4.0, (K, D)), rng.uniform(0.5, 3.0, (K, D))
            R = rng.dirichlet(np.ones(K), n)
            z = CategoricalNode(Z, log_prob=np.log(R))
            vmp = MeanFieldVMP(log_joint, [z, DirichletNode(pi, alpha=alpha),
                                           NormalGammaNode(*ng_vars, m=m, kappa=kappa, a=a, b=b)],
                               {X_name: X}, backend=_PROBE)
            # messages of the global factors, in the fused layout (NormalGammaNode keeps
            # (kappa m, -kappa / 2, a - 1/2, -b - kappa m^2 / 2): the fused layout is (e1, -2 e2, 2 e3, -2 e4))
            (m_pi,) = vmp.message(pi.name)
            g = [np.broadcast_to(np.asarray(v, np.float64), (K, D)) for v in vmp.message(ng_vars[0].name)]
            got = np.concatenate([np.broadcast_to(m_pi, (K,)), g[0].ravel(), (-2.0 * g[1]).ravel(),
                                  (2.0 * g[2]).ravel(), (-2.0 * g[3]).ravel()])
            prior = got - scale * _mog_message_fused(R, X, K, D)
            if eta0 is None:
                eta0 = prior
            elif not np.allclose(prior, eta0, rtol=1e-9, atol=1e-9):
                return _say(why, "the global factors' messages are not a fixed prior + %g * [sum r | R^T X | R^T X^2]" % scale)
            # the local latent's logits (up to a constant per row, which the softmax does not see)
            (logits,) = vmp.message(Z.name)
            T = a / b
            want = scale * ((digamma(alpha) - digamma(alpha.sum()))[None, :]
                            + X @ (T * m).T - 0.5 * (X * X) @ T.T
                            + (0.5 * (digamma(a) - np.log(b)) - 0.5 * (1.0 / kappa + m * m * T)).sum(axis=1)[None, :])
            d1 = logits - logits[:, :1]
            d2 = want - want[:, :1]
            if not np.allclose(d1, d2, rtol=1e-9, atol=1e-9 * np.abs(d2).max()):
                return _say(why, "the assignments' logits are not E[log pi_k] + sum_d (E[tau mu] x - E[tau] x^2 / 2 + "
                                 "E[log tau] / 2 - E[tau mu^2] / 2) times %g" % scale)
    except NOT_THIS_MODEL as e:       # not conjugate, other node types, shapes that do not fit ...: not this model
        return _say(why, "%s while deriving the update rules: %s" % (type(e).__name__, e))
    kappa0 = eta0[K + K * D:K + 2 * K * D]
    if not (np.all(kappa0 > 0.0) and np.all(eta0[:K] > -1.0)):
        return _say(why, "the recovered prior is improper (kappa0 <= 0 or alpha0 <= 0)")
    return eta0


# ---- hierarchical logistic regression (config 5) ----------------------------------------------------

class LogisticHierarchy(object):
    """What ``logistic_hierarchy`` found: log p(data, z_s) = scale * sum_n [y_n l_ns - softplus(l_ns)] +
    log N(w_s | 0, I) + sum_g log N(b_sg | 0, e^{-zeta_s}) + log Gamma(e^{zeta_s} | a0, b0) + zeta_s + offset,
    l_ns = x_n . w_s + b_{s, g_n} -- the model csrc/bsc_bbvi.hip's kernels compute, the group of row n given
    by a one-hot matrix in the symbolic form (``dot(Gm, B.T)``) and by an index vector on the device."""

    def __init__(self, X, y, onehot, W, B, zeta, scale, a0, b0, offset):
        self.X, self.y, self.onehot, self.W, self.B, self.zeta = X, y, onehot, W, B, zeta
        self.scale, self.a0, self.b0, self.offset = scale, a0, b0, offset


def logistic_hierarchy(log_joint, latents, data_shapes, n_samples, why=None):
    """Identity testing by evaluation on a seven-row instance (host float64): the log-joint must be the closed
    form above for SOME (scale, a0, b0, offset), whatever way it was written.  Returns a ``LogisticHierarchy``
    or None."""
    import math
    types = log_joint.input_types
    if len(latents) != 3:
        return _say(why, "config 5 has three latent blocks (weights, group intercepts, log precision); got %d" % len(latents))
    # explicit extents (a log-normaliser times the number of rows) keep the REAL data's value; the sums over the
    # rows are then taken over the instance's rows
    shapes = dict(data_shapes)
    shapes.update({v.name: (n_samples, size) for v, size in latents})
    try:
        log_joint = _without_shapes(log_joint, shapes)
    except NOT_THIS_MODEL as e:
        return _say(why, "%s while resolving the extents of the log-joint: %s" % (type(e).__name__, e))
    reason = "no (design matrix [N, D], one-hot group matrix [N, G], targets [N]) triple whose widths match the latents"
    two_d = [n for n in data_shapes if types.get(n, (None, 0))[1] == 2]
    one_d = [n for n in data_shapes if types.get(n, (None, 0))[1] == 1]
    rng = np.random.RandomState(31)
    n = 7
    for Xn in two_d:
        for Gn in two_d:
            if Gn == Xn or data_shapes[Gn][0] != data_shapes[Xn][0]:
                continue
            D, G = data_shapes[Xn][1], data_shapes[Gn][1]
            by_size = {}
            for v, size in latents:
                by_size.setdefault(size, []).append(v)
            if D == G or D == 1 or G == 1 or sorted(by_size) != sorted({D, G, 1}) \
                    or any(len(vs) != 1 for vs in by_size.values()):
                continue
            Wv, Bv, Zv = by_size[D][0], by_size[G][0], by_size[1][0]
            for yn in one_d:
                if data_shapes[yn][0] != data_shapes[Xn][0]:
                    continue
                try:
                    def draw_data():
                        g = rng.randint(G, size=n)
                        return rng.standard_normal((n, D)), (rng.uniform(size=n) < 0.5).astype(np.float64), g

                    def F(data, w, b, zeta):
                        X, y, g = data
                        return np.asarray(_PROBE.evaluate(log_joint, {
                            Xn: X, yn: y, Gn: np.eye(G)[g], Wv.name: w, Bv.name: b, Zv.name: zeta[:, None]}),
                            np.float64).reshape(-1)

                    def ell(data, w, b):
                        X, y, g = data
                        L = X @ w.T + b[:, g].T
                        return (y[:, None] * L - np.logaddexp(0.0, L)).sum(axis=0)

                    S = 3
                    # the shapes of the REAL data enter the expression only through resolved extents: none here
                    # (sums over rows are sums over the instance's rows), so the instance stands for any size
                    d1, d2 = draw_data(), draw_data()
                    w, b, zeta = rng.standard_normal((S, D)) * 0.5, rng.standard_normal((S, G)) * 0.5, rng.standard_normal(S) * 0.5
                    de = ell(d1, w, b) - ell(d2, w, b)
                    scale_s = (F(d1, w, b, zeta) - F(d2, w, b, zeta)) / de
                    scale = float(scale_s[0])
                    if not (np.isfinite(scale) and scale > 0 and np.allclose(scale_s, scale, rtol=1e-9)):
                        reason = ("the data do not enter as scale * sum_n [y_n l_ns - softplus(l_ns)], l = X w + b[group] "
                                  "(X = %s, groups = %s, y = %s)" % (Xn, Gn, yn))
                        continue
                    zero_w, zero_b = np.zeros((1, D)), np.zeros((1, G))
                    r = lambda zt: (F(d1, zero_w, zero_b, np.array([zt])) - scale * ell(d1, zero_w, zero_b))[0]
                    t = 1.0
                    r0, r1, r2 = r(0.0), r(t), r(-t)
                    # r(zeta) = c + (a0 + G / 2) zeta - b0 e^{zeta}
                    b0 = -((r1 + r2) - 2.0 * r0) / (math.exp(t) + math.exp(-t) - 2.0)
                    slope = ((r1 - r2) + b0 * (math.exp(t) - math.exp(-t))) / (2.0 * t)
                    a0 = slope - 0.5 * G
                    c = r0 + b0
                    if not (a0 > 0 and b0 > 0 and np.isfinite(c)):
                        reason = "the log-precision's terms are not those of a Gamma(a0, b0) prior with a0, b0 > 0"
                        continue
                    const = -0.5 * (D + G) * math.log(2 * math.pi) + a0 * math.log(b0) - math.lgamma(a0)
                    ok = True
                    for _ in range(3):
                        w, b, zeta = rng.standard_normal((S, D)), rng.standard_normal((S, G)), rng.standard_normal(S) * 0.7
                        d = draw_data()
                        want = scale * ell(d, w, b) - 0.5 * (w * w).sum(1) + 0.5 * G * zeta \
                            - 0.5 * np.exp(zeta) * (b * b).sum(1) + a0 * zeta - b0 * np.exp(zeta) + c
                        got = F(d, w, b, zeta)
                        ok = ok and np.allclose(got, want, rtol=1e-9, atol=1e-9 * np.abs(want).max())
                    if ok:
                        return LogisticHierarchy(Xn, yn, Gn, Wv.name, Bv.name, Zv.name, scale, float(a0), float(b0),
                                                 float(c - const))
                    reason = ("the parameter-sized part is not log N(w | 0, I) + sum_g log N(b_g | 0, e^{-zeta}) + "
                              "log Gamma(e^{zeta} | a0, b0) + zeta (checked at random points)")
                except NOT_THIS_MODEL as e:
                    reason = "%s while evaluating the log-joint on a seven-row instance: %s" % (type(e).__name__, e)
                    continue
    return _say(why, reason)
