"""Config 3's pass (10M x 16 rows, K = 64) on the f32 MFMA and on the operand-split bf16 route.
    python tools/bench_mog_split.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd.device import Context     # noqa: E402


def main():
    N, D, K = 10_000_000, 16, 64
    ctx = Context(0)
    g = torch.Generator(device=ctx.device).manual_seed(23)
    cen = torch.randn((K, D), generator=g, device=ctx.device) * 4
    X = cen[torch.randint(K, (N,), generator=g, device=ctx.device)] + torch.randn((N, D), generator=g, device=ctx.device)
    T = torch.rand((K, D), generator=g, device=ctx.device) + 0.5
    Wmat = torch.cat([T * cen, -0.5 * T], dim=1).contiguous()
    c = (-0.5 * (T * cen ** 2).sum(1)).contiguous()
    stats, lse = ctx.zeros((K, 1 + 2 * D), torch.float64), ctx.zeros(1, torch.float64)
    # float64 reference on the device, in chunks
    ref = torch.zeros((K, 1 + 2 * D), dtype=torch.float64, device=ctx.device)
    ref_l = 0.0
    W64, c64 = Wmat.double(), c.double()
    for i in range(0, N, 1_000_000):
        x = X[i:i + 1_000_000].double()
        F = torch.cat([x, x * x], 1)
        L = F @ W64.T + c64
        l = torch.logsumexp(L, 1)
        R = torch.exp(L - l[:, None])
        ref += torch.cat([R.sum(0)[:, None], R.T @ F], 1)
        ref_l += l.sum().item()
    scale = torch.cat([torch.tensor([float(N)], device=ctx.device), X.double().abs().sum(0), (X.double() ** 2).sum(0)])
    for terms in (0, 2):
        ctx.call("bsc_ctx_set_mfma_split", terms)
        run = lambda: ctx.call("bsc_mog_estep", X, D, N, D, K, Wmat, c, stats, lse)
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
        err = ((stats - ref).abs() / scale[None, :]).max().item()
        print("terms %d: %.1f us per call  %.1f TF f32-equivalent  %.2f TB/s of X;  max |stats - float64| / scale %.2e;  "
              "lse rel. error %.2e" % (terms, ms * 1e3, 8.0 * K * D * N / ms * 1e-9, N * D * 4 / ms * 1e-9, err,
                                       abs(lse.item() - ref_l) / abs(ref_l)))
    ctx.call("bsc_ctx_set_mfma_split", 0)


if __name__ == "__main__":
    main()
