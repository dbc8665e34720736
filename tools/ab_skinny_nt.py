"""Deletion builds of gemm_skinny_nt_kernel (P = dot(W, X.T), 8 x 1M x 256): what the 4.7 TB/s are made of.
BSC_SKINNY_NT_DBG bits: 1 no MFMAs, 4 no stores
(results WRONG by construction; BSC_PROFILING_BUILDS=1).  Three distinct X in rotation: no Infinity-Cache reuse.

    python tools/ab_skinny_nt.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BSC_PROFILING_BUILDS"] = "1"

import torch

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
    N, D, S = 1_000_000, 256, 8
    Xs = [torch.randn((N, D), generator=g, device=dev) for _ in range(3)]
    W = torch.randn((S, D), generator=g, device=dev) / 16
    P = torch.empty((S, N), device=dev)
    nbytes = 4.0 * N * D + 4.0 * S * N
    names = {0: "as shipped", 1: "no MFMAs", 4: "no stores", 5: "DMAs and LDS reads only"}
    for wg in (1, 2):
        for dbg in (0, 1, 4, 5):
            c = make_ctx({"BSC_SKINNY_NT_DBG": str(dbg), "BSC_SKINNY_NT_WG": str(wg)})
            k = [0]

            def fn():
                X = Xs[k[0] % 3]
                k[0] += 1
                c.call("bsc_gemm_strided_batched", 0, 1, S, N, D, W, 0, D, 1, X, 0, 1, D, P, 0, N, 1)
            for _ in range(30):
                fn()
            torch.cuda.synchronize()
            best = []
            for _ in range(3):
                e0, e1 = c.event(), c.event()
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                best.append(e0.elapsed_ms(e1) / 30 * 1e3)
            us = min(best)
            print("WG/CU %d  dbg %d  %-42s %7.1f us  %5.2f TB/s" % (wg, dbg, names[dbg], us, nbytes / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
