"""X^T X at 1M x 256 (config 2's companion conjugate statistic) on the f32 MFMA and with X as two bf16 terms.
    python tools/bench_gram_split.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd.device import Context     # noqa: E402


def main():
    N, D = 1_000_000, 256
    ctx = Context(0)
    g = torch.Generator(device=ctx.device).manual_seed(1)
    X = torch.randn((N, D), generator=g, device=ctx.device)
    C = ctx.zeros((D, D), torch.float32)
    ref = torch.zeros((D, D), dtype=torch.float64, device=ctx.device)
    for i in range(0, N, 100_000):
        x = X[i:i + 100_000].double()
        ref += x.T @ x
    n2 = (X.double() ** 2).sum(0).sqrt()
    scale = n2[:, None] * n2[None, :]
    for terms in (0, 2):
        ctx.call("bsc_ctx_set_mfma_split", terms)
        run = lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, D, D, N, X, 0, 1, D, X, 0, D, 1, C, 0, D, 1)
        for _ in range(5):
            run()
        ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        n = 30
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        ms = e0.elapsed_ms(e1) / n
        err = ((C.double() - ref).abs() / scale).max().item()
        print("terms %d: %.1f us per call  %.1f TF f32-equivalent  %.2f TB/s of X;  max |C - float64| / (|x_d| |x_e|) %.2e"
              % (terms, ms * 1e3, 2.0 * N * D * D / ms * 1e-9, N * D * 4 / ms * 1e-9, err))
    ctx.call("bsc_ctx_set_mfma_split", 0)


if __name__ == "__main__":
    main()
