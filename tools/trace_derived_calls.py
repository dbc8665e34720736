"""Lists the C-ABI calls of one derived-mixture update (inference/mixture.py) with the extents and
strides the executor passed -- to see which launches take the generic paths.

    python tools/trace_derived_calls.py [rows]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from bayesic_amd.algebra.device_backend import DeviceBackend
from bayesic_amd.device import Context
from bayesic_amd.inference.mixture import DiagonalMixtureVMP
from bayesic_amd.svi import mog as mog_mod


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    D, K = 16, 64
    ctx = Context(0)
    rs = np.random.RandomState(3)
    X = rs.standard_normal((n, D)).astype(np.float32) * 3
    eta = mog_mod.init_eta(X[:2000], K, D, seed=2)
    alpha, m, kappa, a, b = mog_mod.unpack(eta, K, D)
    derived = DiagonalMixtureVMP(X, K, n_total=float(n), init=(alpha, m, kappa, a, b), backend=DeviceBackend(ctx), route="derived")
    for _ in range(2):
        derived.step()
    ctx.sync()
    real = ctx.call
    log = []

    def spy(name, *args):
        def show(v):
            if hasattr(v, "_length_") or isinstance(v, (list, tuple)):
                try:
                    return [int(x) for x in v]
                except Exception:
                    return "[...]"
            if hasattr(v, "shape"):
                return "T%s/%s" % (tuple(v.shape), tuple(v.stride()))
            return v
        log.append((name, [show(a) for a in args]))
        return real(name, *args)

    ctx.call = spy
    derived.step()
    ctx.sync()
    ctx.call = real
    for name, args in log:
        print(name, args)


if __name__ == "__main__":
    main()
