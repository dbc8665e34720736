"""Config 2's model (known noise variance for brevity) through the GENERAL reparameterisation engine
at BASELINE size, beside the fused kernels: what hand fusion buys.
    python tools/bench_generic_reparam.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd import algebra as A                          # noqa: E402
from bayesic_amd.algebra.device_backend import DeviceBackend   # noqa: E402
from bayesic_amd.device import Context                         # noqa: E402
from bayesic_amd.inference import ReparamVI                    # noqa: E402


def main():
    ctx = Context(0)
    N, D, S, s2 = 1_000_000, 256, 8, 0.25
    g = torch.Generator(device=ctx.device).manual_seed(0)
    Xd = torch.randn((N, D), generator=g, device=ctx.device)
    yd = torch.randn(N, generator=g, device=ctx.device)
    X, y, W = A.var("X", 2), A.var("y", 1), A.var("W", 2)
    r = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
    lj = A.sum(r * r, axis=1) * (-0.5 / s2) + A.sum(W * W, axis=1) * (-0.5)
    ctx.set_stream(torch.cuda.Stream(ctx.device))          # a stream of its own: what a graph capture needs
    data = dict(X=Xd, y=yd)
    for graph, route, resident, replay in ((False, "general", False, False), (True, "general", False, False),
                                           (False, "general", True, False), (False, "general", True, True),
                                           (True, "general", True, False), (False, "auto", False, False),
                                           (False, "auto", True, False), (False, "auto", True, True)):
        eng = ReparamVI(lj, [(W, D)], data, n_samples=S, seed=1, backend=DeviceBackend(ctx), lr=1e-3, graph=graph,
                        route=route, resident=resident, replay=replay)
        for _ in range(5):
            eng.step()
        ctx.sync()
        t0 = time.perf_counter()
        steps = 40
        for _ in range(steps):
            eng.step()
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        how = " (walk recorded as a hipGraph)" if graph else \
            " (recorded C-ABI call list re-issued: the default of a resident engine)" if (replay and resident) else ""
        print("reparameterisation engine, route = %s%s, %dx%d, S=%d: %.2f ms per update (%.1f updates/s), elbo %.6e; "
              "the fused config-2 kernels: 0.17 ms" % (eng.route, how, N, D, S, dt * 1e3, 1.0 / dt, eng.elbo))
    # the same engine on the context's DEFAULT stream (no stream of its own: a graph capture is not available there)
    ctx2 = Context(0)
    for replay in (False, True):
        eng = ReparamVI(lj, [(W, D)], data, n_samples=S, seed=1, backend=DeviceBackend(ctx2), lr=1e-3, route="general",
                        resident=True, replay=replay)
        for _ in range(5):
            eng.step()
        ctx2.sync()
        t0 = time.perf_counter()
        for _ in range(40):
            eng.step()
        ctx2.sync()
        dt = (time.perf_counter() - t0) / 40
        print("    default stream, resident, replay=%s: %.2f ms per update" % (replay, dt * 1e3))


if __name__ == "__main__":
    main()
