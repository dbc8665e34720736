"""The two skinny products of the general reparameterisation step at config 2's size
(S = 8 draws, 1M x 256): P = dot(W, X.T) and G = dot(R, X), LDS-DMA kernels against the
128 x 128-tile GEMM (BSC_GEMM_SKINNY=0), in one process.

    python tools/bench_skinny.py
"""
import os
os.environ.setdefault("BSC_PROFILING_BUILDS", "1")   # the Context honours BSC_<OPTION> variables only in a process that opts in (device.py)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

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
    variants = {"128x128-tile GEMM": make_ctx({"BSC_GEMM_SKINNY": "0"}), "LDS-DMA skinny": make_ctx({})}
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    N, D = 1_000_000, 256
    X = torch.randn((N, D), generator=g, device=dev)
    for S in (8, 16, 32):
        W = torch.randn((S, D), generator=g, device=dev) / 16
        R = torch.randn((S, N), generator=g, device=dev)
        P = torch.empty((S, N), device=dev)
        G = torch.empty((S, D), device=dev)
        calls = {
            "P = dot(W, X.T)  [%d x 1M x 256]" % S:
                (lambda c: c.call("bsc_gemm_strided_batched", 0, 1, S, N, D, W, 0, D, 1, X, 0, 1, D, P, 0, N, 1),
                 4.0 * N * D + 4.0 * S * N),
        }
        if S <= 16:
            calls["G = dot(R, X)    [%d x 256 x 1M]" % S] = (
                lambda c: c.call("bsc_gemm_strided_batched", 0, 1, S, D, N, R, 0, N, 1, X, 0, D, 1, G, 0, D, 1),
                4.0 * N * D + 4.0 * S * N)
        for what, (fn, nbytes) in calls.items():
            for name, c in variants.items():
                for _ in range(60):
                    fn(c)
                torch.cuda.synchronize()
                e0, e1 = c.event(), c.event()
                e0.record()
                for _ in range(30):
                    fn(c)
                e1.record()
                us = e0.elapsed_ms(e1) / 30 * 1e3
                print("%-36s %-20s %8.1f us  %5.2f TB/s" % (what, name, us, nbytes / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
