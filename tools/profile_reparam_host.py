"""Host-side cost of one step of the GENERAL reparameterisation engine (config 2's model, known noise): cProfile over
many steps of a SMALL problem -- the kernels take microseconds, what is measured is the Python walk and the ctypes
calls between them.

    python tools/profile_reparam_host.py [steps]
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bayesic_amd import algebra as A
from bayesic_amd.algebra.device_backend import DeviceBackend
from bayesic_amd.device import Context
from bayesic_amd.inference import ReparamVI


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ctx = Context(0)
    N, D, S, s2 = 8192, 256, 8, 0.25
    g = torch.Generator(device=ctx.device).manual_seed(0)
    Xd = torch.randn((N, D), generator=g, device=ctx.device)
    yd = torch.randn(N, generator=g, device=ctx.device)
    X, y, W = A.var("X", 2), A.var("y", 1), A.var("W", 2)
    r = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
    lj = A.sum(r * r, axis=1) * (-0.5 / s2) + A.sum(W * W, axis=1) * (-0.5)
    eng = ReparamVI(lj, [(W, D)], dict(X=Xd, y=yd), n_samples=S, seed=1, backend=DeviceBackend(ctx), lr=1e-3, route="general", resident=True)
    for _ in range(10):
        eng.step()
    ctx.sync()
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        eng.step()
    ctx.sync()
    pr.disable()
    print("%.3f ms per step (host-bound: %d rows)" % ((time.perf_counter() - t0) / reps * 1e3, N))
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(40)
    st.sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
