"""Builds and loads the plain-C second oracle (oracle/c/oracle_kernels.c).  TEST
INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Output goes to oracle/_build/ (git-ignored;
it travels to the GPU box with the snapshot, and is rebuilt there if gcc is present)."""
import ctypes
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "oracle_kernels.c")
OUT = os.path.join(HERE, "_build", "liboracle.so")


def build(force=False):
    gcc = shutil.which("gcc")
    if gcc is None:
        if os.path.exists(OUT):
            return OUT
        raise RuntimeError("gcc not found and no prebuilt oracle/_build/liboracle.so")
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call([gcc, "-O2", "-fopenmp", "-shared", "-fPIC", SRC, "-o", OUT, "-lm"])
    return OUT


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        vp, c_long, c_int = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
        lib.oracle_blr_data_pass.argtypes = [vp, c_long, vp, c_long, c_int, vp, c_int, vp, vp]
        lib.oracle_blr_data_pass_f32.argtypes = [vp, c_long, vp, c_long, c_int, vp, c_int, vp, vp]
        lib.oracle_blr_data_pass_f32.restype = c_int
        lib.oracle_parallel_copy.argtypes = [vp, vp, c_long, c_long]
        lib.oracle_parallel_copy.restype = None
        lib.oracle_logreg_loglik.argtypes = [vp, c_long, vp, vp, c_long, c_int, c_int, vp, vp, c_int, vp]
        lib.oracle_mog_estep.argtypes = [vp, c_long, c_long, c_int, c_int, vp, vp, vp, vp]
        lib.oracle_lda_sstats.argtypes = [vp, c_long, c_long, c_long, c_int, vp, vp, vp]
        lib.oracle_lda_local_bound.argtypes = [vp, c_long, c_long, c_long, c_int, vp, vp]
        lib.oracle_lda_local_bound.restype = ctypes.c_double
        lib.oracle_weighted_outer.argtypes = [vp, vp, vp, c_long, c_int, c_int, c_int, vp]
        lib.oracle_weighted_outer.restype = None
        lib.oracle_threads.restype = c_int
        lib.oracle_set_threads.argtypes = [c_int]
        lib.oracle_set_threads.restype = None
        for f in (lib.oracle_blr_data_pass, lib.oracle_logreg_loglik, lib.oracle_mog_estep,
                  lib.oracle_lda_sstats):
            f.restype = None
        _lib = lib
    return _lib


def cpu_share():
    """Logical CPUs this process may actually use: the scheduler affinity, cut down to the cgroup's CPU quota when there
    is one (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def use_cpu_share():
    """Run the OpenMP passes on cpu_share() threads from now on; returns that number."""
    n = min(int(load().oracle_threads()), cpu_share())
    load().oracle_set_threads(n)
    return n


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def blr_data_pass(X, y, W):
    import numpy as np
    X, y, W = (np.ascontiguousarray(a, np.float32) for a in (X, y, W))
    S, D = W.shape
    Q, G = np.zeros(S), np.zeros((S, D))
    load().oracle_blr_data_pass(_p(X), X.shape[1], _p(y), X.shape[0], D, _p(W), S, _p(Q), _p(G))
    return Q, G


def blr_data_pass_f32(X, y, W):
    """The fused single-pass float32 OpenMP leg (oracle_blr_data_pass_f32): bench.py's cpu_baseline.fused_f32."""
    import numpy as np
    X, y, W = (np.ascontiguousarray(a, np.float32) for a in (X, y, W))
    S, D = W.shape
    Q, G = np.zeros(S), np.zeros((S, D))
    rc = load().oracle_blr_data_pass_f32(_p(X), X.shape[1], _p(y), X.shape[0], D, _p(W), S, _p(Q), _p(G))
    if rc != 0:
        raise ValueError("oracle_blr_data_pass_f32: S = %d, D = %d outside its envelope (S <= 16, D <= 1024)" % (S, D))
    return Q, G


def first_touch_copy(a):
    """A copy of the row-major array `a` whose pages were first touched by the threads (row blocks) of the OpenMP passes."""
    import numpy as np
    a = np.ascontiguousarray(a)
    out = np.empty_like(a)
    rows = a.shape[0]
    load().oracle_parallel_copy(_p(out), _p(a), rows, a.nbytes // max(rows, 1))
    return out


def logreg_loglik(X, y, g, Wz, Bz):
    import numpy as np
    X, y, Wz, Bz = (np.ascontiguousarray(a, np.float32) for a in (X, y, Wz, Bz))
    g = np.ascontiguousarray(g, np.int32)
    S, D = Wz.shape
    ell = np.zeros(S)
    load().oracle_logreg_loglik(_p(X), X.shape[1], _p(y), _p(g), X.shape[0], D, Bz.shape[0], _p(Wz),
                                _p(Bz), S, _p(ell))
    return ell


def mog_estep(X, Wmat, c):
    import numpy as np
    X, Wmat, c = (np.ascontiguousarray(a, np.float32) for a in (X, Wmat, c))
    K, D = Wmat.shape[0], X.shape[1]
    stats, lse = np.zeros((K, 1 + 2 * D)), np.zeros(1)
    load().oracle_mog_estep(_p(X), X.shape[1], X.shape[0], D, K, _p(Wmat), _p(c), _p(stats), _p(lse))
    return stats, float(lse[0])


def lda_sstats(C, Th, Bt):
    import numpy as np
    C, Th, Bt = (np.ascontiguousarray(a, np.float32) for a in (C, Th, Bt))
    K, V = Bt.shape
    out = np.zeros((K, V))
    load().oracle_lda_sstats(_p(C), C.shape[1], C.shape[0], V, K, _p(Th), _p(Bt), _p(out))
    return out


def lda_local_bound(C, Th, Bt):
    import numpy as np
    C, Th, Bt = (np.ascontiguousarray(a, np.float32) for a in (C, Th, Bt))
    K, V = Bt.shape
    return float(load().oracle_lda_local_bound(_p(C), C.shape[1], C.shape[0], V, K, _p(Th), _p(Bt)))


def weighted_outer(R, X, Y):
    import numpy as np
    R, X, Y = (np.ascontiguousarray(a, np.float32) for a in (R, X, Y))
    K, D, E = R.shape[1], X.shape[1], Y.shape[1]
    out = np.zeros((K, D, E))
    load().oracle_weighted_outer(_p(R), _p(X), _p(Y), R.shape[0], K, D, E, _p(out))
    return out
