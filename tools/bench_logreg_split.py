"""Config 5's data pass (1M x 256 rows, 64 draws, 1000 groups) on the f32 MFMA and with X and the draws as two bf16
terms (bsc_ctx_set_mfma_split 2).    python tools/bench_logreg_split.py"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd.device import Context     # noqa: E402


def main():
    N, D, G, S = 1_000_000, 256, 1000, 64
    ctx = Context(0)
    g = torch.Generator(device=ctx.device).manual_seed(3)
    X = torch.randn((N, D), generator=g, device=ctx.device)
    y = (torch.rand(N, generator=g, device=ctx.device) < 0.4).float()
    ids = torch.randint(G, (N,), generator=g, device=ctx.device, dtype=torch.int32)
    Wz = torch.randn((S, D), generator=g, device=ctx.device) / math.sqrt(D)
    Bz = torch.randn((G, S), generator=g, device=ctx.device)
    ell = ctx.zeros(S, torch.float64)
    L = X.double() @ Wz.double().T + Bz.double()[ids.long()]
    ref = (y.double()[:, None] * L - torch.nn.functional.softplus(L)).sum(0).cpu().numpy()
    scale = (L.abs() + 1.0).sum(0).cpu().numpy()
    del L
    for terms in (0, 2):
        ctx.call("bsc_ctx_set_mfma_split", terms)
        run = lambda: ctx.call("bsc_logreg_bbvi_loglik", X, D, y, ids, N, D, G, Wz, Bz, S, ell)
        for _ in range(5):
            run()
        ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        n = 50
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        ms = e0.elapsed_ms(e1) / n
        got = ell.cpu().numpy()
        print("terms %d: %.1f us per call  %.2f TB/s of X  %.1f TF f32-equivalent;  |ell - float64| / sum(|l| + 1): max %.2e"
              % (terms, ms * 1e3, N * D * 4 / ms * 1e-9, 2.0 * N * D * S / ms * 1e-9, np.abs((got - ref) / scale).max()))
    ctx.call("bsc_ctx_set_mfma_split", 0)


if __name__ == "__main__":
    main()
