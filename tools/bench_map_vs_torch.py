"""The executor's element-wise launches beside torch's own kernels for the same expressions (a yardstick for what a
tuned read + write kernel reaches on this box, not a dependency).    python tools/bench_map_vs_torch.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd.algebra import dimshuffle, exp, var     # noqa: E402
from bayesic_amd.algebra.device_backend import DeviceBackend  # noqa: E402
from bayesic_amd.device import Context   # noqa: E402


def main():
    ctx = Context(0)
    be = DeviceBackend(ctx)
    dev = ctx.device
    g = torch.Generator(device=dev).manual_seed(0)
    rows = 1_000_000
    Xd = torch.randn((rows, 256), generator=g, device=dev)
    Yd = torch.randn((rows, 256), generator=g, device=dev)
    vd = torch.randn(256, generator=g, device=dev)
    ud = torch.randn(rows, generator=g, device=dev)
    X, Y, v, u = var("X", 2), var("Y", 2), var("v", 1), var("u", 1)
    n = rows * 256 * 4
    cases = [
        ("X * v[None,:]", X * dimshuffle(v, "x", 0), dict(X=Xd, v=vd), lambda: Xd * vd[None, :], 2 * n),
        ("X + u[:,None]", X + dimshuffle(u, 0, "x"), dict(X=Xd, u=ud), lambda: Xd + ud[:, None], 2 * n),
        ("X / Y", X / Y, dict(X=Xd, Y=Yd), lambda: Xd / Yd, 3 * n),
        ("X * Y", X * Y, dict(X=Xd, Y=Yd), lambda: Xd * Yd, 3 * n),
        ("exp(X)", exp(X), dict(X=Xd), lambda: torch.exp(Xd), 2 * n),
        ("X * 2.0", X * 2.0, dict(X=Xd), lambda: Xd * 2.0, 2 * n),
    ]
    for name, expr, inputs, tfn, nbytes in cases:
        f = expr.compile(be).device_fn
        out = {}
        for label, fn in (("executor", lambda: f(**inputs)), ("torch", tfn)):
            for _ in range(3):
                r = fn()
            del r
            torch.cuda.synchronize()
            ctx.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            if label == "torch":
                e0.record()
                for _ in range(reps):
                    r = fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
            else:
                a, b = ctx.event(), ctx.event()
                a.record()
                for _ in range(reps):
                    r = fn()
                b.record()
                us = a.elapsed_ms(b) / reps * 1e3
            out[label] = us
        print("%-16s executor %7.1f us = %.2f TB/s   torch %7.1f us = %.2f TB/s"
              % (name, out["executor"], nbytes / out["executor"] * 1e-6, out["torch"], nbytes / out["torch"] * 1e-6))


if __name__ == "__main__":
    main()
