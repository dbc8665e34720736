"""Planner: Einsum -> tree of the five executable primitives
(_tensordot, _sum, _mul, _dimshuffle, _diagonal).

Behavioural contract: bayesic/algebra.py:513-765; the chosen trees are pinned
by bayesic/tests/test_algebra.py:376-505 and by tests/golden/algebra_golden.json.
This decides every GEMM's M/N/K and operand orientation that the device sees
(SURVEY.md 8(a) A7).

Rules reproduced on purpose:
  * sum indices are eliminated in numeric order, so the bracketing the user
    wrote is the bracketing that runs;
  * indices that always co-occur with the one being eliminated go with it
    (sum_ij X_ij Y_ij is ONE two-axis tensordot);
  * the lhs/rhs split of a contraction greedily minimises
        #factors(lhs) * #indices(lhs) + #factors(rhs) * #indices(rhs)
    where every candidate move is compared with the cost of the INITIAL split
    (the reference never refreshes `current_cost`, bayesic/algebra.py:648-654;
    the pinned outcomes depend on that);
  * indices on both sides that are not being summed become batch axes.
"""
from collections import Counter

from .einsum_form import OUT, SUM, Einsum
from .expr import constant
from .multiset import find_duplicate


def lower(e):
    return _eliminate_sums(_peel_diagonals(e))


def _peel_diagonals(e):
    """A repeated index on one factor becomes a _diagonal whose new axis goes
    last (bayesic/algebra.py:513-525)."""
    from .ops import _diagonal
    pairs = []
    for factor, indices in e.factors_and_indices:
        while True:
            repeat = find_duplicate(indices)
            if repeat is None:
                break
            later, earlier, index = repeat
            factor = _diagonal(factor, later, earlier)
            indices = tuple(i for axis, i in enumerate(indices)
                            if axis != later and axis != earlier) + (index,)
        pairs.append((factor, indices))
    return Einsum(pairs, e.ndim)


def _product_of_aligned_factors(e):
    """No sum indices left: broadcast every factor to the output axes and
    multiply (bayesic/algebra.py:741-765)."""
    from .ops import _dimshuffle, _mul, dimshuffle
    aligned = []
    for factor, indices in e.factors_and_indices:
        axes = [indices.index(o) if o in indices else "x" for o in e.out_indices]
        aligned.append(factor if axes == list(range(factor.ndim))
                       else _dimshuffle(factor, *axes))
    if not aligned:
        return dimshuffle(constant(1), *(["x"] * e.ndim))
    if len(aligned) == 1:
        return aligned[0]
    return _mul(*aligned)


def _group_cost(group):
    """#distinct (factor, indices) entries times #distinct indices among them."""
    return len(group) * len({i for _, indices in group for i in indices})


def _split_sides(holders):
    """Greedy lhs/rhs split of the factors carrying the contracted index.  Both
    sides are multisets that remember insertion order."""
    lhs = list(holders)
    rhs = Counter([lhs.pop()])
    lhs = Counter(lhs)
    baseline = _group_cost(lhs) + _group_cost(rhs)   # never refreshed, see module doc
    while len(lhs) > 1:
        best = None
        for candidate in lhs:
            moved_lhs = lhs - Counter([candidate])
            moved_rhs = rhs + Counter([candidate])
            cost = _group_cost(moved_lhs) + _group_cost(moved_rhs)
            if best is None or cost < best[0]:
                best = (cost, moved_lhs, moved_rhs)
        if best[0] >= baseline:
            break
        _, lhs, rhs = best
    return list(lhs.elements()), list(rhs.elements())


def _eliminate_sums(e):
    from .ops import _sum, _tensordot
    sums = e.sum_indices
    if not sums:
        return _product_of_aligned_factors(e)
    pairs = e.factors_and_indices

    def carriers(index):
        return [n for n, (_, indices) in enumerate(pairs) if index in indices]

    lead = sums[0]
    lead_carriers = carriers(lead)
    contracted = [i for i in sums if carriers(i) == lead_carriers]

    if len(lead_carriers) == 1:
        # the indices live on one factor only: sum that factor first
        n = lead_carriers[0]
        factor, indices = pairs[n]
        summed = _sum(factor, *[indices.index(i) for i in contracted])
        rest = tuple(i for i in indices if i not in contracted)
        replaced = list(pairs)
        replaced[n] = (summed, rest)
        return _eliminate_sums(Einsum(replaced, e.ndim))

    holders = [pairs[n] for n in lead_carriers]
    lhs, rhs = _split_sides(holders)
    others = [p for p in pairs if lead not in p[1]]
    needed_outside = {i for _, indices in others for i in indices}
    lhs_all = {i for _, indices in lhs for i in indices}
    rhs_all = {i for _, indices in rhs for i in indices}
    shared = lhs_all & rhs_all
    batch = sorted(shared - set(contracted))

    def side(group, contracted_first):
        """Sub-einsum for one operand, its contraction / batch axes and the
        indices of its remaining ("other") axes."""
        present = {i for _, indices in group for i in indices}
        exposed = sorted(i for i in present
                         if i[0] == OUT or i in needed_outside or i in batch)
        # usual dot convention: contracted axes last on the lhs, first on the rhs
        axes_order = (contracted + exposed) if contracted_first else (exposed + contracted)
        to_out = {index: (OUT, n) for n, index in enumerate(axes_order)}
        operand = _eliminate_sums(Einsum(
            [(f, tuple(to_out.get(i, i) for i in indices)) for f, indices in group],
            len(axes_order)))
        return (operand,
                [to_out[i][1] for i in contracted],
                [to_out[i][1] for i in batch],
                [i for i in axes_order if i not in shared])

    lhs_op, lhs_dot, lhs_batch, lhs_free = side(lhs, False)
    rhs_op, rhs_dot, rhs_batch, rhs_free = side(rhs, True)
    product = _tensordot(lhs_op, rhs_op, lhs_dot, rhs_dot, lhs_batch, rhs_batch)
    product_indices = tuple(batch + lhs_free + rhs_free)   # batch, lhs others, rhs others

    # Put the product roughly where its factors were, so that later contractions
    # keep the operand order the user wrote (bayesic/algebra.py:722-739).
    used = lhs + rhs
    where = sum(pairs.index(p) for p in used) / len(used)
    placed = [(p, pairs.index(p)) for p in others] + [((product, product_indices), where)]
    placed.sort(key=lambda item: item[1])
    return _eliminate_sums(Einsum([p for p, _ in placed], e.ndim))
