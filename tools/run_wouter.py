"""A few launches of bsc_weighted_outer at config 3's size (for rocprofv3 passes).
    python tools/run_wouter.py [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd.device import Context  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ctx = Context(0)
    g = torch.Generator(device=ctx.device).manual_seed(0)
    N, K, D = 10_000_000, 64, 16
    X = torch.randn((N, D), generator=g, device=ctx.device)
    R = torch.softmax(torch.randn((N, K), generator=g, device=ctx.device), dim=1)
    out = torch.empty((K, D, D), device=ctx.device)
    for _ in range(reps):
        ctx.call("bsc_weighted_outer", R, K, X, D, X, D, N, K, D, D, 1.0, out)
    ctx.sync()
    print(float(out.sum().item()))


if __name__ == "__main__":
    main()
