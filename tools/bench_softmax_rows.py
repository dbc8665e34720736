"""bsc_softmax_rows at 10M x 64 (the responsibilities of a resident Categorical node).   python tools/bench_softmax_rows.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd.device import Context     # noqa: E402

ctx = Context(0)
for N, K in ((10_000_000, 64), (4_000_000, 64), (10_000_000, 16), (1_000_000, 256)):
    g = torch.Generator(device=ctx.device).manual_seed(0)
    X = torch.randn((N, K), generator=g, device=ctx.device)
    R, lse = torch.empty_like(X), torch.empty(N, device=ctx.device)
    run = lambda: ctx.call("bsc_softmax_rows", X, N, K, K, R, K, lse)
    for _ in range(3):
        run()
    ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    ms = e0.elapsed_ms(e1) / 20
    ref = torch.softmax(X[:100000].double(), 1)
    print("%d x %d: %.1f us  %.2f TB/s (read + write);  max |R - float64| %.1e"
          % (N, K, ms * 1e3, 2 * N * K * 4 / ms * 1e-9, (R[:100000].double() - ref).abs().max().item()))
