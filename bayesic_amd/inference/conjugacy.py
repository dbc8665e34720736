"""Reading conjugacy off a symbolic log-joint with ``match``.

The reference's algebra module exists "to help with reasoning about conjugacy"
(bayesic/algebra.py:1-6) and its README plans to "identify nodes which have the
conjugate exponential family property in the context of their Markov blanket"
and to apply "a traditional variational message passing update (equivalently a
unit-step natural gradient update) ... derivable automatically / in closed form"
(README.md:30-37).  Neither exists in the reference; ``match``
(bayesic/algebra.py:1037-1063) is the tool it built for the job: it pulls the
coefficient of a factor out of a multilinear term.

A latent z with variational family q(z) proportional to exp(sum_j <t_j(z), eta_j>)
is conjugate in its Markov blanket iff every term of the log-joint that mentions z
is linear in one of the statistics t_j(z):

    log p = sum_j < t_j(z), c_j(everything else) > + terms without z.

``conjugate_coefficients`` finds the c_j term by term with ``match`` or reports the
first term that does not fit.  The VMP update is then eta_j <- E_q(others)[c_j].
"""
from .. import algebra as A
from ..algebra.einsum_form import Einsum
from ..algebra.expr import add


class NotConjugate(ValueError):
    """A term of the log-joint mentions the latent but is not linear in any of its
    sufficient statistics; ``.term`` is that term."""

    def __init__(self, latent, term):
        ValueError.__init__(self, "term %r is not linear in a sufficient statistic of %s"
                            % (term, latent))
        self.term = term


def depends_on(expr, variable):
    """Does the expression mention the ``var`` (by name)?"""
    return variable.name in expr.input_types


def _carried_axes(factor):
    """Axes of a factor that are real.  An Einsum whose out index occurs on none of its factors --
    or only on broadcast axes of them -- has a BROADCAST axis there (bayesic/algebra.py:340-344); an
    element-wise node (opaque to the einsum form: log, exp, pow, add) is broadcast along an axis
    exactly when all its arguments are: ``log(dimshuffle(v, 0, 'x'))`` has one real axis, and a sum
    over the other counts it extent-of-the-axis times once it is taken out of the ``add`` it was
    broadcast in (the value semantics the reference inherits from Theano's broadcastable axes)."""
    from ..algebra.expr import elemwise
    if isinstance(factor, Einsum):
        carried = set()
        for inner, indices in factor.factors_and_indices:
            real = _carried_axes(inner)
            carried.update(n for axis, (kind, n) in enumerate(indices) if kind == "out" and axis in real)
        return carried
    if isinstance(factor, elemwise):
        carried = set()
        for parent in factor.parents:
            carried |= _carried_axes(parent)
        return carried
    return set(range(factor.ndim))


def expand_terms(expr):
    """Summands of ``expr`` with products and sums distributed over ``add``:
    a flat list of multilinear terms whose sum equals ``expr``.  The front end
    flattens nested adds but never distributes (``(X + Y) * Z`` stays a product of
    a sum, bayesic/algebra.py:69-71); conjugacy is a statement about terms.

    A summand that was broadcast inside the add (``Y + 1``: the 1 has broadcast
    axes) no longer carries the summed indices once it stands alone, so the sum over
    such an index turns into a factor "extent of that index": sum_ij (Y + 1)_ij =
    sum_ij Y_ij + n_i n_j."""
    expr = A.wrap_if_literal(expr)
    if isinstance(expr, add):
        return [t for s in expr.terms() for t in expand_terms(s)]
    if isinstance(expr, Einsum):
        pairs = list(expr.factors_and_indices)
        for position, (factor, indices) in enumerate(pairs):
            if not isinstance(factor, add):
                continue
            sum_indices = {i for _, idx in pairs for i in idx if i[0] == "sum"}
            out = []
            for summand in factor.terms():
                replaced = pairs[:position] + [(summand, indices)] + pairs[position + 1:]
                carried = {idx[ax] for f, idx in replaced for ax in _carried_axes(f)}
                extents = []
                for lost in sorted(sum_indices - carried):
                    ax = list(indices).index(lost)
                    donor = next((s for s in factor.terms() if ax in _carried_axes(s)), None)
                    if donor is None:
                        # broadcast in EVERY summand: the axis has extent 1 in this add (where it was
                        # broadcast against something wider, that level has supplied the extent already)
                        continue
                    extents.append((A.shape(donor, ax), ()))
                out += expand_terms(A.einsum(replaced + extents, expr.ndim))
            return out
    return [expr]


def conjugate_coefficients(log_joint, latent, statistics):
    """Coefficients c_j with  log_joint = sum_j <t_j(latent), c_j> + rest.

    log_joint : scalar expression, or a list of scalar expressions (summands)
    latent    : the ``var`` of the latent variable
    statistics: expressions t_j(latent), e.g. ``(mu, mu ** 2)`` or ``(log(tau), tau)``

    Returns ``(coefficients, rest)``: a list with one expression (or None when no term
    touches that statistic) per statistic, of the statistic's shape, free of the latent;
    and the list of terms that do not mention the latent.  Raises NotConjugate when a
    term mentions the latent but is linear in none of the statistics.

    Statistics are tried in REVERSE order, so list the higher-order one last:
    ``sum(x * mu * mu * tau)`` would also match the template ``sum(mu * Z)`` with a
    coefficient that still contains mu, which is rejected.
    """
    pieces = log_joint if isinstance(log_joint, (list, tuple)) else [log_joint]
    terms = [t for piece in pieces for t in expand_terms(piece)]
    found = [[] for _ in statistics]
    rest = []
    # a statistic may be carried by a variable of its own (the second moment of a vector
    # latent, see MVNormalNode): a term then belongs to the latent when it mentions any of them
    carriers = {latent.name}
    for t in statistics:
        carriers.update(A.wrap_if_literal(t).input_types)

    def mentions(expr):
        return any(name in expr.input_types for name in carriers)

    for term in terms:
        if term.ndim != 0:
            raise ValueError("log-joint terms must be scalars, got ndim %d: %r" % (term.ndim, term))
        if not mentions(term):
            rest.append(term)
            continue
        for j in reversed(range(len(statistics))):
            t = A.wrap_if_literal(statistics[j])
            slot = A.var("_coefficient_%d" % j, t.ndim)
            template = A.sum(t * slot) if t.ndim else t * slot
            coefficient = A.match(term, template, slot)
            if coefficient is not None and not mentions(coefficient):
                found[j].append(coefficient)
                break
        else:
            raise NotConjugate(latent, term)
    coefficients = [None if not cs else (cs[0] if len(cs) == 1 else A.add(*cs)) for cs in found]
    return coefficients, rest
