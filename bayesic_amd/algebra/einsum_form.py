"""The Einsum canonical form: one flat ``sum over sum-indices of a product of
indexed factors`` for every multilinear expression.

Behavioural contract: bayesic/algebra.py:314-511 (einsum, Einsum, canonical
form), :770-805 (repr), :983-1034 (equality by isomorphism, order-insensitive
hash); pinned by bayesic/tests/test_algebra.py:182-205,282-358 and the golden
fixtures.  Indices are the public tuples ('sum', n) / ('out', n).
"""
from collections import Counter, defaultdict

from .expr import Expression, eye, var, wrap_if_literal
from .multiset import equivalence_classes

OUT, SUM = "out", "sum"


def einsum(factors_and_indices, ndim=None):
    """General multilinear form, like numpy.einsum with explicit index roles.

    ``factors_and_indices`` pairs each factor with one index per axis; an index is
    ('sum', n) (summed over) or ('out', n) (axis n of the result).  E.g.

        einsum([(X, (o2, s0, o1)), (Y, (s0, s1, o0, o1))])

    is T[o0,o1,o2] = sum_{s0,s1} X[o2,s0,o1] * Y[s0,s1,o0,o1].  ``ndim`` defaults
    to max out index + 1; an out index that never occurs is a broadcast axis
    (bayesic/algebra.py:314-347).
    """
    return Einsum(factors_and_indices, ndim)._canonicalize()


class Einsum(Expression):
    def __init__(self, factors_and_indices, ndim=None):
        pairs = [(factor, tuple(indices)) for factor, indices in factors_and_indices]
        for factor, indices in pairs:
            if factor.ndim != len(indices):
                raise ValueError(
                    "The indices for each factor must have same length as factor.ndim")
        out_numbers = [n for _, indices in pairs for kind, n in indices if kind == OUT]
        if ndim is None:
            ndim = max(out_numbers) + 1
        if any(n < 0 or n >= ndim for n in out_numbers):
            raise ValueError("some output indices are out of range")
        self.ndim = ndim
        self.factors_and_indices = tuple(pairs)
        super(Einsum, self).__init__([factor for factor, _ in pairs])

    @classmethod
    def _wrap_if_not_einsum(cls, expr):
        """Identity einsum around a non-einsum expression."""
        if isinstance(expr, cls):
            return expr
        return cls([(expr, tuple((OUT, i) for i in range(expr.ndim)))], expr.ndim)

    # -- index bookkeeping -------------------------------------------------------
    @property
    def out_indices(self):
        return [(OUT, i) for i in range(self.ndim)]

    @property
    def sum_indices(self):
        return sorted({i for _, indices in self.factors_and_indices for i in indices
                       if i[0] == SUM})

    def factors(self):
        return self.parents

    # -- canonical form ------------------------------------------------------------
    def _canonicalize(self):
        return self._absorb_nested()._drop_summed_eyes()._unwrap_identity()

    def _absorb_nested(self):
        """Inline factors that are themselves einsums.  Sum indices of the nested
        einsums are numbered first (in factor order), then this einsum's own, in
        order of first appearance -- which preserves the bracketing order that the
        lowering later honours (bayesic/algebra.py:398-461)."""
        numbering = {}
        for position, (factor, _) in enumerate(self.factors_and_indices):
            if isinstance(factor, Einsum):
                for inner in factor.sum_indices:
                    numbering[(position, inner)] = (SUM, len(numbering))
        for _, indices in self.factors_and_indices:
            for index in indices:
                if index[0] == SUM and index not in numbering:
                    numbering[index] = (SUM, len(numbering))

        flat = []
        for position, (factor, outer_indices) in enumerate(self.factors_and_indices):
            if not isinstance(factor, Einsum):
                flat.append((wrap_if_literal(factor),
                             tuple(numbering.get(i, i) for i in outer_indices)))
                continue
            for inner_factor, inner_indices in factor.factors_and_indices:
                translated = []
                for kind, n in inner_indices:
                    if kind == OUT:   # axis n of the nested result: what we call it out here
                        outer = outer_indices[n]
                        translated.append(numbering.get(outer, outer))
                    else:             # summed inside the nested einsum
                        translated.append(numbering[(position, (kind, n))])
                flat.append((inner_factor, tuple(translated)))
        return Einsum(flat, self.ndim)

    def _drop_summed_eyes(self):
        """delta_ij with a summed index just identifies i with j: remove the eye
        and merge the indices, preferring an out index as the survivor, then
        renumber the surviving sum indices densely (bayesic/algebra.py:463-500)."""
        identified = [(i, i) for i in self.sum_indices]
        kept = []
        for factor, indices in self.factors_and_indices:
            if isinstance(factor, eye) and SUM in (indices[0][0], indices[1][0]):
                identified.append(indices)
            else:
                kept.append((factor, indices))
        survivor = {}
        for group in equivalence_classes(identified):
            chosen = min(group)   # ('out', n) sorts before ('sum', n)
            for index in group:
                survivor[index] = chosen
        # One entry per merged index, so a survivor that absorbed others is listed
        # several times and takes the LAST of its positions.  The numbers can have
        # gaps; only their order matters downstream.  (Same numbering as the
        # reference, so `factors_and_indices` compares equal across the two.)
        live_sums = sorted(n for kind, n in survivor.values() if kind == SUM)
        dense = {(SUM, n): (SUM, k) for k, n in enumerate(live_sums)}

        def rename(index):
            index = survivor.get(index, index)
            return dense.get(index, index)

        return Einsum([(f, tuple(rename(i) for i in indices)) for f, indices in kept],
                      self.ndim)

    def _unwrap_identity(self):
        if len(self.factors_and_indices) == 1:
            factor, indices = self.factors_and_indices[0]
            if self.ndim == factor.ndim and \
                    all(index == (OUT, axis) for axis, index in enumerate(indices)):
                return factor
        return self

    # -- evaluation: lower, then let the backend walk the five-op tree -------------
    def _rewrite_as_special_case_ops(self):
        from .lowering import lower
        return lower(self)

    def lowered(self):
        """Cached five-op implementation tree (the reference re-lowers on every
        ``apply``, bayesic/algebra.py:767-768)."""
        cached = getattr(self, "_lowered_cache", None)
        if cached is None:
            cached = self._rewrite_as_special_case_ops()
            self._lowered_cache = cached
        return cached

    # -- printing --------------------------------------------------------------------
    def __repr__(self):
        sums = self.sum_indices

        def letter(index):
            kind, n = index
            if kind == OUT:
                return "uvwxyz"[n] if n < 6 else "o%d" % n
            rank = sums.index(index)
            return "ijklmn"[rank] if rank < 6 else "s%d" % rank

        def show(factor, indices):
            if not indices:
                return factor.bracketed_repr()
            return "%s_%s" % (factor.bracketed_repr(), "".join(letter(i) for i in indices))

        body = " ".join(show(f, i) for f, i in self.factors_and_indices) or "1"
        if sums:
            body = "sum_%s %s" % ("".join(letter(i) for i in sums), body)
        if self.ndim > 0:
            return "einsum(out_%s = %s)" % ("".join(letter(i) for i in self.out_indices), body)
        return "einsum(%s)" % body

    # -- pattern matching / equality ----------------------------------------------------
    def match(self, template, slot):
        from .matching import match_einsum
        return match_einsum(self, template, slot)

    def __eq__(self, other):
        """Equal iff isomorphic up to factor order and sum-index naming: match
        against ``other * scalar_slot`` must leave nothing over
        (bayesic/algebra.py:983-999)."""
        if self is other:
            return True
        if not isinstance(other, self.__class__):
            return False
        from .ops import mul
        slot = var("__slot__", ndim=0)
        leftover = self.match(mul(other, slot), slot)
        return leftover is not None and len(leftover.factors_and_indices) == 0

    def __hash__(self):
        """Insensitive to factor order and sum-index numbering: a sum index is
        described by where it occurs (bayesic/algebra.py:1001-1034).  Computed once per object."""
        cached = self.__dict__.get("_hash_value")
        if cached is not None:
            return cached
        occurrences = defaultdict(Counter)
        for factor, indices in self.factors_and_indices:
            for axis, index in enumerate(indices):
                if index[0] == SUM:
                    occurrences[index][(factor, axis)] += 1
        signature = {i: frozenset(c.items()) for i, c in occurrences.items()}
        described = Counter(
            (factor, tuple(i if i[0] == OUT else (SUM, signature[i]) for i in indices))
            for factor, indices in self.factors_and_indices)
        value = self.__dict__["_hash_value"] = hash(frozenset(described.items()))
        return value

    @staticmethod
    def _factor_axes_for_indices(indices_for_factors):
        """{index: frozenset of (factor number, axis number) where it occurs}."""
        where = defaultdict(set)
        for factor_no, indices in enumerate(indices_for_factors):
            for axis_no, index in enumerate(indices):
                where[index].add((factor_no, axis_no))
        return {index: frozenset(axes) for index, axes in where.items()}
