"""Random expression trees through the device executor against the float64 numpy oracle.

SURVEY.md 8(f) rank 1 claims that ANY bayesic.algebra expression runs on the GPU; the fixed
corpora (test_algebra_gpu.py, test_fusion_gpu.py) only visit what somebody thought of.  Here a
seeded generator grows 160 trees from the public operators -- add / sub / mul / div by positive
values, exp, log, abs, pow, dot, tensordot, outer, sum over axes, dimshuffle with broadcast axes,
transpose, trace, diagonal -- over float32 inputs of awkward extents, evaluates each with
oracle.einsum_eval.NumpyBackend in float64 and with the HIP executor, and compares.

Tolerance: float32 arithmetic over contractions of up to a few hundred terms: 2e-4 of the
result's largest magnitude (the reference's own contraction tolerance is rtol 1e-5 on much
smaller sums, bayesic/tests/test_algebra.py:82)."""
import builtins

import numpy as np
import pytest

from bayesic_amd import algebra as A
from oracle.einsum_eval import NumpyBackend

pytestmark = pytest.mark.gpu

SHAPES = {"X": (5, 7), "Y": (5, 7), "Z": (7, 4), "Q": (6, 6), "x": (7,), "y": (5,), "T": (3, 5, 7)}


def inputs():
    rs = np.random.RandomState(2024)
    return {n: rs.uniform(-1.0, 1.0, s).astype(np.float32) for n, s in SHAPES.items()}


class Grower(object):
    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.pool = [(A.var(n, ndim=len(s)), s) for n, s in SHAPES.items()]

    def pick(self, pred=lambda s: True):
        cands = [p for p in self.pool if pred(p[1])]
        return cands[self.rs.randint(len(cands))] if cands else None

    def step(self):
        r = self.rs
        op = r.randint(12)
        e, s = self.pick()
        if op == 0:                                   # same-shape add / sub / mul
            o = self.pick(lambda t: t == s)
            f = [lambda a, b: a + b, lambda a, b: a - b, lambda a, b: a * b][r.randint(3)]
            return f(e, o[0]), s
        if op == 1:                                   # scalar arithmetic
            c = float(np.round(r.uniform(0.5, 2.0), 2))
            return [lambda a: a * c, lambda a: a + c, lambda a: c - a, lambda a: a / c][r.randint(4)](e), s
        if op == 2:                                   # bounded unary chains
            return [lambda a: A.exp(a * 0.25), lambda a: A.log(abs(a) + 1.5), lambda a: abs(a),
                    lambda a: A.pow(abs(a) + 0.5, 1.5)][r.randint(4)](e), s
        if op == 3:                                   # division by something positive
            o = self.pick(lambda t: t == s)
            return e / (abs(o[0]) + 1.0), s
        if op == 4 and len(s) >= 1:                   # sum over one axis or all
            if r.randint(2) or len(s) == 1:
                return A.sum(e), ()
            ax = int(r.randint(len(s)))
            return A.sum(e, axis=ax), tuple(d for i, d in enumerate(s) if i != ax)
        if op == 5 and len(s) == 2:                   # transpose
            return e.T, (s[1], s[0])
        if op == 6 and len(s) >= 1:                   # dot with a fitting operand
            o = self.pick(lambda t: len(t) >= 1 and t[0] == s[-1])
            if o is not None and len(s) + len(o[1]) - 2 <= 3:
                return A.dot(e, o[0]), s[:-1] + o[1][1:]
        if op == 7 and len(s) == 1:                   # outer product
            o = self.pick(lambda t: len(t) == 1)
            return A.outer(e, o[0]), s + o[1]
        if op == 8 and len(s) == 2 and s[0] == s[1]:  # trace / diagonal
            return (A.trace(e), ()) if r.randint(2) else (A.diagonal(e), (s[0],))
        if op == 9 and 1 <= len(s) <= 2:              # broadcast axis in, then multiply it away
            pat = list(range(len(s)))
            pat.insert(int(r.randint(len(s) + 1)), "x")
            shp = tuple(1 if p == "x" else s[p] for p in pat)
            return A.dimshuffle(e, *pat), shp
        if op == 10 and len(s) == 2:                  # permuting dimshuffle
            return A.dimshuffle(e, 1, 0), (s[1], s[0])
        if op == 11 and len(s) == 3:                  # contraction of a 3-D factor
            o = self.pick(lambda t: len(t) == 2 and t == s[1:])
            if o is not None:
                return A.tensordot(e, o[0], [1, 2], [0, 1]), (s[0],)
        return None

    def grow(self, steps):
        made = None
        for _ in range(steps):
            out = self.step()
            if out is None:
                continue
            made = out
            if builtins.all(d != 1 for d in out[1]):     # keep broadcast-shaped values out of the pool
                self.pool.append(out)
        return made


@pytest.mark.parametrize("seed", range(160))
def test_random_tree_matches_the_float64_oracle(ctx, seed):
    from bayesic_amd.algebra.device_backend import DeviceBackend
    made = Grower(seed).grow(2 + seed % 7)
    if made is None:
        pytest.skip("generator produced nothing for this seed")
    expr, _ = made
    vals = {n: v for n, v in inputs().items() if n in expr.input_types}
    want = np.asarray(expr.compile(NumpyBackend(np.float64))(**vals), np.float64)
    got = np.asarray(expr.compile(DeviceBackend(ctx))(**vals), np.float64)
    assert got.shape == want.shape, repr(expr)
    scale = builtins.max(float(np.abs(want).max()) if want.size else 0.0, 1e-3)
    assert np.abs(got - want).max() <= 2e-4 * scale if want.size else True, \
        "%r: max err %g of scale %g" % (expr, np.abs(got - want).max(), scale)
