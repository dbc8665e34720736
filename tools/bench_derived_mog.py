"""Config 3 (10M x 16, K = 64) through the DERIVED mean-field engine (inference/mixture.py: symbolic
log-joint -> conjugacy detection -> messages through the executor, assignments resident on the
device) against the hand-fused update of svi/mog.py.

    python tools/bench_derived_mog.py [rows]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd.algebra.device_backend import DeviceBackend
from bayesic_amd.device import Context
from bayesic_amd.inference.mixture import DiagonalMixtureVMP
from bayesic_amd.svi import mog as mog_mod


def main():
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    n = int(nums[0]) if nums else 10_000_000
    D, K = 16, 64
    ctx = Context(0)
    rs = np.random.RandomState(3)
    centres = rs.standard_normal((K, D)) * 4.0
    X = (centres[np.random.RandomState(4).randint(K, size=n)] +
         np.random.RandomState(5).standard_normal((n, D)).astype(np.float32)).astype(np.float32)
    eta0 = mog_mod.prior_eta(K, D)
    eta = mog_mod.init_eta(X[:2000], K, D, seed=2)
    alpha, m, kappa, a, b = mog_mod.unpack(eta, K, D)
    fused = mog_mod.MoGNatGradSVI(ctx.to_device(X), K, eta0, eta, n_total=float(n), ctx=ctx)
    derived = DiagonalMixtureVMP(X, K, n_total=float(n), init=(alpha, m, kappa, a, b), backend=DeviceBackend(ctx),
                                 resident_globals="--host-globals" not in sys.argv, route="derived")
    # the same symbolic model with route="auto": the derived update rules are recognised as the fused kernels' own
    auto = DiagonalMixtureVMP(X, K, n_total=float(n), init=(alpha, m, kappa, a, b), backend=DeviceBackend(ctx))
    print("route of the symbolic model: %s" % auto.route, flush=True)
    for name, step in (("hand-written driver (svi/mog.py)", fused.step),
                       ("symbolic model, route=auto", auto.step),
                       ("symbolic model, route=derived", derived.step)):
        for _ in range(3):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            step()
        ctx.sync()
        print("%-34s %9.3f ms per update (%d rows, K = %d, D = %d)" % (name, (time.perf_counter() - t0) / reps * 1e3, n, K, D),
              flush=True)
        if "--steps" in sys.argv:          # each update on its own: is a slow run slow in every update?
            each = []
            for _ in range(8):
                t1 = time.perf_counter()
                step()
                ctx.sync()
                each.append((time.perf_counter() - t1) * 1e3)
            print("    single updates (ms): " + " ".join("%.2f" % v for v in each), flush=True)
            import torch
            st = torch.cuda.memory_stats()
            print("    torch allocator: %d device mallocs, %d frees, %.1f GB reserved"
                  % (st["num_device_alloc"], st["num_device_free"], st["reserved_bytes.all.current"] / 1e9))
    want = fused.eta.cpu().numpy()
    scale = np.maximum(np.abs(want), 1.0)
    for name, model in (("derived", derived), ("auto", auto)):
        got = model.eta_fused_layout()
        print("max relative difference of the natural parameters after 13 updates (%s vs hand-written): %.2e"
              % (name, np.abs((got - want) / scale).max()))


if __name__ == "__main__":
    main()
