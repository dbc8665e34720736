"""Generates tests/golden/algebra_golden.json by evaluating algebra_corpus.py
with the REFERENCE front end (/root/reference/bayesic/algebra.py).

Only SYMBOLIC results are recorded -- canonical reprs, lowered five-op trees,
match() results, equality/hash outcomes, combinatorics -- all of which are
computed by the reference's own pure-Python code.  Theano (the reference's only
import-time dependency, absent from this image) never takes part in those
computations; the stub below exists only so that `import bayesic.algebra`
succeeds and `constant(v)` can report ndim/dtype.  No numeric output of the stub
is recorded: the numeric pin of the einsum path is numpy itself, exactly as in
the reference's own tests (bayesic/tests/test_algebra.py:44-191).

Run in the build container only (the reference is not on the GPU box):

    python tests/golden/make_algebra_golden.py
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"


def install_import_stub():
    class _Op:
        def __init__(self, name):
            self.scalar_op = types.SimpleNamespace(name=name)
            self.name = name

        def __repr__(self):
            return "<stub op %s>" % self.name

    class _Const:
        def __init__(self, value):
            arr = np.asarray(value)
            self.ndim = arr.ndim
            self.dtype = str(arr.dtype)

    theano = types.ModuleType("theano")
    tensor = types.ModuleType("theano.tensor")
    for n in ("add", "mul", "log", "exp", "pow", "abs_"):
        setattr(tensor, n, _Op(n))
    tensor.constant = _Const
    theano.tensor = tensor
    sys.modules["theano"] = theano
    sys.modules["theano.tensor"] = tensor


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("reference not present at %s (expected on the GPU box)" % REFERENCE)
    install_import_stub()
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, HERE)
    import bayesic.algebra as ref
    import algebra_corpus as corpus

    ns = {k: getattr(ref, k) for k in dir(ref) if not k.startswith("__")}
    for name, ndim, dtype in corpus.NAMESPACE_VARS:
        ns[name] = ref.var(name, ndim, dtype)
    ns["abs"] = abs

    def ev(src):
        return eval(src, dict(ns))

    out = {"expressions": [], "matches": [], "equalities": [], "injections": []}
    for src in corpus.EXPRESSIONS:
        e = ev(src)
        rec = {"src": src, "repr": repr(e), "ndim": e.ndim, "type": type(e).__name__,
               "input_types": {k: list(v) for k, v in sorted(e.input_types.items())}}
        if isinstance(e, ref.Einsum):
            rec["lowered"] = repr(e._rewrite_as_special_case_ops())
            rec["n_factors"] = len(e.factors_and_indices)
        out["expressions"].append(rec)
    for expr_src, tmpl_src, slot in corpus.MATCHES:
        m = ref.match(ev(expr_src), ev(tmpl_src), ns[slot])
        out["matches"].append({"expr": expr_src, "template": tmpl_src, "slot": slot,
                               "repr": None if m is None else repr(m)})
    for lhs, rhs in corpus.EQUALITIES:
        a, b = ev(lhs), ev(rhs)
        eq = bool(a == b)
        out["equalities"].append({"lhs": lhs, "rhs": rhs, "equal": eq,
                                  "hash_equal": bool(hash(a) == hash(b))})
    # combinatorics: the parametrised cases of bayesic/tests/test_algebra.py:208-279
    second = lambda p, q: p[1] == q[1]
    cases = [
        ([1], [], None), ([1, 1], [1], None), ([1, 1, 2], [1, 2], None), ([], [], None),
        ([1], [1], None), ([1], [1, 2], None), ([1, 1], [1, 1], None), ([1, 1], [1, 1, 2], None),
        ([1, 2], [5, 2, 1, 3], None),
        (["a1", "a1"], ["b1"], "second"), (["a1", "a1"], ["b1", "b2"], "second"),
        (["a1"], ["b1"], "second"), (["a1", "a3"], ["b3", "b1"], "second"),
        (["a1", "a1"], ["a2", "b1", "c1"], "second"), (["a1", "b1"], ["c1", "c1", "d2"], "second"),
        (["a1", "a1"], ["a1", "b1", "c1", "d1", "e1"], "second"),
        (["a1", "a1", "b1", "b1"], ["w1", "x1", "y1", "z1"], "second"),
        (["a1", "b1"], ["x1", "y1", "z2"], "second"),
        (["a1", "a1", "a1", "b2"], ["x1", "x1", "x1", "y1", "y1", "z2", "extra"], "second"),
    ]
    for A, B, m in cases:
        res = list(ref.find_injections(A, B, second) if m else ref.find_injections(A, B))
        canon = sorted(sorted([[repr(k[0]), repr(k[1]), c] for k, c in inj.items()])
                       for inj in res)
        out["injections"].append({"A": A, "B": B, "match": m, "result": canon})
    path = os.path.join(HERE, "algebra_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s: %d expressions, %d matches, %d equalities, %d injection cases"
          % (path, len(out["expressions"]), len(out["matches"]), len(out["equalities"]),
             len(out["injections"])))


if __name__ == "__main__":
    main()
