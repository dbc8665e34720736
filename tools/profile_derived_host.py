"""Host-side cost of one derived-mixture update: cProfile over many updates of a SMALL problem (the
kernels take microseconds, what is measured is the Python walk and the ctypes calls between them).

    python tools/profile_derived_host.py [updates]
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from bayesic_amd.algebra.device_backend import DeviceBackend
from bayesic_amd.device import Context
from bayesic_amd.inference.mixture import DiagonalMixtureVMP
from bayesic_amd.svi import mog as mog_mod


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n, D, K = 4096, 16, 64
    ctx = Context(0)
    rs = np.random.RandomState(3)
    X = rs.standard_normal((n, D)).astype(np.float32) * 3
    eta = mog_mod.init_eta(X[:2000], K, D, seed=2)
    alpha, m, kappa, a, b = mog_mod.unpack(eta, K, D)
    model = DiagonalMixtureVMP(X, K, n_total=float(n), init=(alpha, m, kappa, a, b), backend=DeviceBackend(ctx), route="derived")
    for _ in range(5):
        model.step()
    ctx.sync()
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        model.step()
    ctx.sync()
    pr.disable()
    print("%.3f ms per update (host-bound: %d rows)" % ((time.perf_counter() - t0) / reps * 1e3, n))
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
