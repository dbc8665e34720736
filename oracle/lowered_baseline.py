"""CPU baseline of SURVEY.md 8(d): numpy executing the SAME lowered op tree the algebra
front end emits for each config's data-sized work, float32 data on BLAS.  TEST / BENCH
INFRASTRUCTURE ONLY (see oracle/__init__.py): called by bench.py's ``cpu_baseline`` leg and by
tests, never by bayesic_amd.

"numpy restatement, not Theano": the reference's backend is Theano (bayesic/algebra.py:8-9),
which is not installable here; its ``_tensordot._apply_to_parents`` (bayesic/algebra.py:1347-1351)
hands an un-batched contraction to ``tensordot`` = BLAS sgemm, which is what
``oracle.einsum_eval.NumpyBackend.tensordot`` does with ``np.tensordot``.

Config 2's pass as algebra expressions (lowered forms per SURVEY 8(a) A7):
    P = dot(W, X.T)                 -> _tensordot(W, _dimshuffle(X,1,0), [1],[0])
    R = y['x',:] - P                -> add(_dimshuffle(y,'x',0), _mul(-1, P))
    Q = sum(R*R, axis=1)            -> _tensordot over the row axis, batched over s / or _sum(_mul)
    G = dot(R, X)                   -> _tensordot(R, X, [1],[0])
"""
import os
import platform
import subprocess

import numpy as np


def host_facts():
    """What SURVEY 8(d) asks to be recorded beside a CPU number."""
    facts = {"cpu_count": os.cpu_count(), "machine": platform.machine(), "numpy": np.__version__}
    try:
        from . import cbuild
        facts["cpu_share"] = cbuild.cpu_share()       # affinity cut down to the cgroup's CPU quota: what the legs can use
    except Exception:
        pass
    try:
        out = subprocess.run(["lscpu"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                             timeout=10).stdout
        for line in out.splitlines():
            key, _, val = line.partition(":")
            if key.strip() in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core",
                               "CPU max MHz"):
                facts["lscpu " + key.strip()] = val.strip()
    except Exception as e:  # lscpu missing: say so rather than guess
        facts["lscpu"] = "unavailable (%s)" % type(e).__name__
    try:
        from threadpoolctl import threadpool_info
        facts["blas"] = [{k: p.get(k) for k in ("internal_api", "version", "num_threads", "threading_layer",
                                                "architecture")}
                         for p in threadpool_info() if p.get("user_api") == "blas"]
    except Exception:
        facts["blas"] = "threadpoolctl unavailable"
    return facts


class _Limit:
    """BLAS / OpenMP thread limit for a with-block (threadpoolctl), or a no-op."""

    def __init__(self, n):
        self.n, self.ctl = n, None

    def __enter__(self):
        if self.n:
            try:
                from threadpoolctl import threadpool_limits
                self.ctl = threadpool_limits(limits=int(self.n))
            except Exception:
                self.ctl = None
        return self

    def __exit__(self, *a):
        if self.ctl is not None:
            self.ctl.restore_original_limits() if hasattr(self.ctl, "restore_original_limits") \
                else self.ctl.unregister()
        return False


def threads(n):
    return _Limit(n)


_BLR = {}


def blr_pass_functions():
    """The config-2 pass as three compiled lowered trees on the float32 numpy backend."""
    if not _BLR:
        from bayesic_amd import algebra as A
        from oracle.einsum_eval import NumpyBackend
        be = NumpyBackend(np.float32)
        X, W, y, R = A.var("X", 2), A.var("W", 2), A.var("y", 1), A.var("R", 2)
        resid = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
        _BLR["resid"] = resid.compile(be)
        _BLR["Q"] = A.sum(R * R, axis=1).compile(be)
        _BLR["G"] = A.dot(R, X).compile(be)
        from bayesic_amd.algebra.lowering import lower
        _BLR["lowered"] = ("resid = add(" + ", ".join(repr(lower(t)) for t in resid.terms()) + "); Q = " +
                           repr(lower(A.sum(R * R, axis=1))) + "; G = " + repr(lower(A.dot(R, X))))
    return _BLR


def blr_data_pass_lowered(X, y, W):
    """(Q [S], G [S, D]) of oracle.svi.blr_data_pass through the lowered trees, float32."""
    f = blr_pass_functions()
    W = np.ascontiguousarray(W, np.float32)
    R = f["resid"](X=X, W=W, y=y)
    Q = f["Q"](R=R)
    G = f["G"](R=R, X=X)
    return np.asarray(Q, np.float64), np.asarray(G, np.float64)
