"""INTEGRATION.md section 1 executed literally: numpy + ctypes, NO torch in this process, null
stream, bsc_malloc / bsc_h2d / bsc_d2h / bsc_memset / bsc_free -- the binding a maintainer of
the reference would write in place of the Theano protocol of bayesic/algebra.py:42-58.

Run by tests/test_boundary_ctypes_gpu.py in a child process; prints "ok ..." lines and exits 0.
Checked against closed forms and float64 numpy on the same inputs (the oracle package is used for
its data generators and its exact posterior only).
"""
import ctypes
import os
import sys
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

lib = ctypes.CDLL(os.path.join(ROOT, "bayesic_amd", "_lib", "libbayesic_hip.so"))
lib.bsc_last_error.restype = c_char_p
lib.bsc_ctx_create.argtypes = [c_int, c_void_p, POINTER(c_void_p)]
lib.bsc_ctx_destroy.argtypes = [c_void_p]
lib.bsc_ctx_sync.argtypes = [c_void_p]
lib.bsc_malloc.argtypes = [c_void_p, c_size_t, POINTER(c_void_p)]
lib.bsc_free.argtypes = [c_void_p, c_void_p]
lib.bsc_h2d.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
lib.bsc_d2h.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
lib.bsc_memset.argtypes = [c_void_p, c_void_p, c_int, c_size_t]
lib.bsc_suffstats_normal.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
lib.bsc_natgrad_update.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double]
lib.bsc_gemm_strided_batched.argtypes = [c_void_p, c_int] + [c_int64] * 4 + \
    [c_void_p, c_int64, c_int64, c_int64] * 3
lib.bsc_allreduce_sum.argtypes = [c_void_p, c_void_p, c_int64, c_int]


def check(rc):
    if rc != 0:
        raise RuntimeError(lib.bsc_last_error().decode())


ctx = c_void_p()
check(lib.bsc_ctx_create(0, None, ctypes.byref(ctx)))          # device 0, null stream


def to_device(a):                     # caller-owned numpy array -> device buffer
    a = np.ascontiguousarray(a)
    p = c_void_p()
    check(lib.bsc_malloc(ctx, a.nbytes, ctypes.byref(p)))
    check(lib.bsc_h2d(ctx, p, a.ctypes.data_as(c_void_p), a.nbytes))
    return p


def device_zeros(nbytes):
    p = c_void_p()
    check(lib.bsc_malloc(ctx, nbytes, ctypes.byref(p)))
    check(lib.bsc_memset(ctx, p, 0, nbytes))
    return p


def to_host(p, shape, dtype):
    out = np.empty(shape, dtype)
    check(lib.bsc_d2h(ctx, out.ctypes.data_as(c_void_p), p, out.nbytes))   # synchronises
    return out


def config1_exact_posterior():
    """BASELINE config 1: Gaussian-Gamma conjugate node, N = 10 000.  A full-batch natural-gradient
    step with rho = 1 must land on the exact Normal-Gamma posterior (README.md:36)."""
    x = (2.0 + 1.5 * np.random.RandomState(1234).standard_normal(10000)).astype(np.float32)
    dx = to_device(x)
    dstats = device_zeros(3 * 8)
    check(lib.bsc_suffstats_normal(ctx, dx, x.size, dstats))
    stats = to_host(dstats, 3, np.float64)
    x64 = x.astype(np.float64)
    np.testing.assert_allclose(stats, [x.size, x64.sum(), (x64 * x64).sum()], rtol=1e-13)
    # Normal-Gamma(mu0, kappa0, alpha0, beta0) natural parameters of the prior:
    #   eta = [kappa*mu, kappa, 2*alpha - 1, 2*beta + kappa*mu^2]; message = [sum x, N, N, sum x^2]
    mu0, kappa0, alpha0, beta0 = 0.0, 1.0, 1.0, 1.0
    eta0 = np.array([kappa0 * mu0, kappa0, 2 * alpha0 - 1, 2 * beta0 + kappa0 * mu0 ** 2])
    n, sx, sxx = stats
    message = np.array([sx, n, n, sxx])
    deta, deta0, dmsg = to_device(np.zeros(4)), to_device(eta0), to_device(message)
    check(lib.bsc_natgrad_update(ctx, deta, deta0, dmsg, 4, 1.0, 1.0))
    eta = to_host(deta, 4, np.float64)
    kappa = eta[1]
    mu = eta[0] / kappa
    alpha = 0.5 * (eta[2] + 1.0)
    beta = 0.5 * (eta[3] - kappa * mu * mu)
    xbar = x64.mean()
    kappa_n = kappa0 + x.size
    np.testing.assert_allclose(kappa, kappa_n, rtol=1e-14)
    np.testing.assert_allclose(mu, (kappa0 * mu0 + x64.sum()) / kappa_n, rtol=1e-13)
    np.testing.assert_allclose(alpha, alpha0 + 0.5 * x.size, rtol=1e-14)
    np.testing.assert_allclose(beta, beta0 + 0.5 * ((x64 - xbar) ** 2).sum() +
                               kappa0 * x.size * (xbar - mu0) ** 2 / (2 * kappa_n), rtol=1e-10)
    for p in (dx, dstats, deta, deta0, dmsg):
        check(lib.bsc_free(ctx, p))
    print("ok config1: exact Normal-Gamma posterior through bsc_suffstats_normal + bsc_natgrad_update")


def gram_matrix():
    """dot(X.T, X) -> _tensordot(_dimshuffle(X,1,0), X, [1],[0]) as ONE
    bsc_gemm_strided_batched call on the caller's row-major X: the transpose is a stride swap."""
    n, d = 50_000, 64
    X = np.random.RandomState(7).standard_normal((n, d)).astype(np.float32)
    dX = to_device(X)
    dC = device_zeros(d * d * 4)
    # C[m, n'] = sum_k A[m, k] B[k, n'],  A = X.T: sa_m = 1, sa_k = d;  B = X: sb_k = d, sb_n = 1
    check(lib.bsc_gemm_strided_batched(ctx, 0, 1, d, d, n, dX, 0, 1, d, dX, 0, d, 1, dC, 0, d, 1))
    C = to_host(dC, (d, d), np.float32)
    ref = X.astype(np.float64).T @ X.astype(np.float64)
    # the reference's own tolerance for contractions: rtol 1e-5 (bayesic/tests/test_algebra.py:82)
    np.testing.assert_allclose(C, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    check(lib.bsc_free(ctx, dX))
    check(lib.bsc_free(ctx, dC))
    print("ok gram: dot(X.T, X) through bsc_gemm_strided_batched")


def world_of_one():
    """bsc_allreduce_sum on a context without a communicator is the identity (no librccl needed)."""
    v = np.arange(5, dtype=np.float64)
    dv = to_device(v)
    check(lib.bsc_allreduce_sum(ctx, dv, 5, 1))
    np.testing.assert_array_equal(to_host(dv, 5, np.float64), v)
    check(lib.bsc_free(ctx, dv))
    print("ok allreduce: identity on a world of one")


if __name__ == "__main__":
    config1_exact_posterior()
    gram_matrix()
    world_of_one()
    check(lib.bsc_ctx_sync(ctx))
    check(lib.bsc_ctx_destroy(ctx))
    assert "torch" not in sys.modules, "this binding must not import torch"
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert "libtorch" not in maps and "librccl" not in maps, "torch / rccl were mapped into the process"
    print("ok torch-free: neither torch nor librccl is loaded in this process")
