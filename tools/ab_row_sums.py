"""Deletion builds of map_reduce_rows_f32_kernel (sum(X * Y, axis=1), 1M x 256): what keeps row sums of short rows
at 4.7-4.9 TB/s.  BSC_ROWS_DBG: 2 no sums across lanes, 3 also float32 lane sums, 4 plain loads (results WRONG by construction;
BSC_PROFILING_BUILDS=1), 8 the four ds_bpermute butterflies of round 3 (results right); BSC_ROWS_WG = workgroups per
CU in the grid, 0 = one step per wave (results unchanged).

    python tools/ab_row_sums.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BSC_PROFILING_BUILDS"] = "1"

import numpy as np
import torch

from bayesic_amd.algebra import var
from bayesic_amd.algebra import sum as asum
from bayesic_amd.algebra.device_backend import DeviceBackend
from bayesic_amd.device import Context


def make_ctx(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    n, d = 1_000_000, 256
    Xs = [torch.randn((n, d), generator=g, device=dev) for _ in range(2)]
    Ys = [torch.randn((n, d), generator=g, device=dev) for _ in range(2)]
    X, Y = var("X", 2), var("Y", 2)
    nbytes = 8.0 * n * d + 4.0 * n
    names = {0: "as shipped", 2: "no sums across lanes", 3: "float32 lane sums, none across lanes", 4: "plain loads",
             8: "four ds_bpermute butterflies (round 3)"}
    for wg in (64, 8, 0):
        for dbg in ((0, 8, 2, 3, 4) if wg == 64 else (0, 8)):
            b = DeviceBackend(make_ctx({"BSC_ROWS_DBG": str(dbg), "BSC_ROWS_WG": str(wg)}))
            f = asum(X * Y, axis=1).compile(b)
            k = [0]

            def fn():
                i = k[0] % 2
                k[0] += 1
                return f.device_fn(X=Xs[i], Y=Ys[i])
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            best = []
            for _ in range(3):
                e0, e1 = b.ctx.event(), b.ctx.event()
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                best.append(e0.elapsed_ms(e1) / 20 * 1e3)
            us = min(best)
            print("WG/CU %2d  dbg %d  %-40s %7.1f us  %5.2f TB/s" % (wg, dbg, names[dbg], us, nbytes / us / 1e6), flush=True)
    t = [(Xs[0] * Ys[0]).sum(1) for _ in range(3)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        torch.sum(Xs[i % 2] * Ys[i % 2], dim=1)
    e1.record()
    torch.cuda.synchronize()
    print("torch.sum(X * Y, dim=1) (two launches, X * Y materialised): %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))


if __name__ == "__main__":
    main()
