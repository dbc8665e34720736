"""Pattern matching of an einsum against a template with one slot.

``match(expression, template, slot)`` returns the expression which, substituted
for the var ``slot`` in ``template``, reproduces ``expression`` -- or None.  It
is how a coefficient of a sufficient statistic is pulled out of a log-joint
term (conjugacy reasoning), and it is also the engine of einsum equality.

Behavioural contract: bayesic/algebra.py:815-981 and :1037-1063; pinned by
bayesic/tests/test_algebra.py:313-358 and tests/golden/algebra_golden.json.

Method (own formulation): choose, by backtracking, a distinct factor of the
expression for every non-slot factor of the template (same factor value).  Such
a choice fixes the index correspondence axis by axis, so there is nothing left
to search: it only has to be a consistent one-to-one map in which
  * a sum index of the expression corresponds to a sum index of the template,
  * an out index corresponds to the same out index, or to a template sum index
    -- in which case a delta (eye) factor is added to the result so that the out
    index becomes summed.
The factors not chosen are re-indexed onto the slot's axes; if one of their
indices has no place on the slot the choice is abandoned and the next is tried.
"""
from .einsum_form import OUT, SUM, Einsum, einsum
from .expr import eye


def match(expression, template, slot):
    """See module docstring.  Can invent identity factors: matching ``A`` against
    ``dot(A, slot)`` gives ``eye(A.shape[1])`` (bayesic/algebra.py:1037-1063)."""
    return Einsum._wrap_if_not_einsum(expression).match(template, slot)


def match_einsum(subject, template, slot):
    template = Einsum._wrap_if_not_einsum(template)
    slot_indices = None
    for factor, indices in template.factors_and_indices:
        if factor is slot:
            slot_indices = indices
            break
    if slot_indices is None:
        raise ValueError("template must contain slot as a factor")
    if len(set(slot_indices)) != len(slot_indices):
        raise ValueError("Same index used on multiple slot axes is not currently supported")
    slot_axis = {index: axis for axis, index in enumerate(slot_indices)}

    fixed = [(f, i) for f, i in template.factors_and_indices if f is not slot]
    subject_pairs = list(subject.factors_and_indices)

    for chosen in _assignments(fixed, subject_pairs):
        result = _complete(subject_pairs, fixed, chosen, slot_axis, slot.ndim)
        if result is not None:
            return result
    return None


def _assignments(fixed, subject_pairs):
    """Yield tuples c with c[k] = position in subject_pairs matched to fixed[k]:
    distinct positions, equal factor values.  Plain backtracking -- einsums have a
    handful of factors."""
    chosen, used = [], set()

    def extend(k):
        if k == len(fixed):
            yield tuple(chosen)
            return
        wanted = fixed[k][0]
        for position, (factor, _) in enumerate(subject_pairs):
            if position in used or not (factor == wanted):
                continue
            used.add(position)
            chosen.append(position)
            yield from extend(k + 1)
            chosen.pop()
            used.discard(position)

    return extend(0)


def _complete(subject_pairs, fixed, chosen, slot_axis, slot_ndim):
    # index correspondence forced by the chosen factors, axis by axis
    to_template, to_subject = {}, {}
    for k, position in enumerate(chosen):
        t_indices = fixed[k][1]
        s_indices = subject_pairs[position][1]
        for t_index, s_index in zip(t_indices, s_indices):
            if to_template.setdefault(s_index, t_index) != t_index:
                return None
            if to_subject.setdefault(t_index, s_index) != s_index:
                return None
    for s_index, t_index in to_template.items():
        if s_index[0] == SUM:
            if t_index[0] != SUM:
                return None
        elif t_index[0] == OUT and t_index[1] != s_index[1]:
            return None

    def place(s_index):
        """Index of a left-over factor, expressed on the slot's axes."""
        if s_index[0] == SUM and s_index not in to_template:
            return s_index                      # private to the left-over factors
        t_index = to_template.get(s_index, s_index)
        axis = slot_axis.get(t_index)
        return None if axis is None else (OUT, axis)

    taken = set(chosen)
    result = []
    for position, (factor, indices) in enumerate(subject_pairs):
        if position in taken:
            continue
        placed = [place(i) for i in indices]
        if any(p is None for p in placed):
            return None
        result.append((factor, placed))

    # out index of the subject that the template sums over: tie the two slot axes
    # together with a delta whose size comes from every axis the index sits on
    matched_factors = [subject_pairs[p][0] for p in chosen]
    matched_indices = [subject_pairs[p][1] for p in chosen]
    where = Einsum._factor_axes_for_indices(matched_indices)
    for s_index, t_index in to_template.items():
        if s_index[0] == OUT and t_index[0] == SUM:
            a, b = slot_axis.get(t_index), slot_axis.get(s_index)
            if a is None or b is None:
                return None
            sizes = [matched_factors[f].shape[axis] for f, axis in where[s_index]]
            result.append((eye(*sizes), sorted([(OUT, a), (OUT, b)])))
    return einsum(result, ndim=slot_ndim)
