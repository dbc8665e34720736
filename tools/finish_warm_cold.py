"""blr_fused_update_kernel behind the 155-us pass (instruction cache and L2 cold) against the same kernel launched again
right behind itself (warm): how much of its ~6.4 us is the cold start?  Timing only -- the second finish of an
iteration consumes the same slab with the next state (finite garbage); read the durations with
rocprofv3 --kernel-trace (tools/finish_warm_cold.py prints nothing itself but a marker).

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/finish_warm_cold.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bayesic_amd.device import Context
from bayesic_amd.svi.blr import BLRReparamSVI


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ctx = Context(0)
    g = torch.Generator(device=ctx.device).manual_seed(3)
    X = torch.randn((rows, 256), generator=g, device=ctx.device)
    y = torch.randn(rows, generator=g, device=ctx.device)
    m = BLRReparamSVI(X, y, n_samples=8, seed=1, lr=1e-3, ctx=ctx)
    for _ in range(10):
        m.step()
    for _ in range(60):
        m.ctx.call("bsc_blr_data_pass_partial_sweep", m._Xarg, m._ldx, m._yarg, m.B, m.D, m.W, m.S, 0)
        m._finish(None)         # cold: behind the pass
        m._finish(None)         # warm: behind the first finish, same slab (still pending in the context)
    ctx.sync()
    print("done")


if __name__ == "__main__":
    main()
