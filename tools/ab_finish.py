"""In-process A/B of the finish kernel's workgroup size (BSC_BLR_FINISH_BLOCK = 1024 | 512 | 256):
whole config-2 updates (pass + finish) in interleaved bursts, wall time per update from events on
the context stream and the finish kernel's own duration from timing slot 2.

    python tools/ab_finish.py [rounds]
"""
import os
os.environ.setdefault("BSC_PROFILING_BUILDS", "1")   # the Context honours BSC_<OPTION> variables only in a process that opts in (device.py)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bayesic_amd.device import Context
from bayesic_amd.svi.blr import BLRReparamSVI


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
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    N, D = 1_000_000, 256
    X = torch.randn((N, D), generator=g, device=dev)
    y = X @ (torch.randn(D, generator=g, device=dev) / 16) + 0.5 * torch.randn(N, generator=g, device=dev)
    models = {}
    for blk in (1024, 512, 256):
        c = make_ctx({"BSC_BLR_FINISH_BLOCK": str(blk)})
        models[blk] = (c, BLRReparamSVI(X, y, n_total=float(N), n_samples=8, seed=1234, lr=1e-3, ctx=c))
    for blk, (c, m) in models.items():
        for _ in range(300):
            m.step()
    torch.cuda.synchronize()
    for r in range(rounds):
        for blk, (c, m) in models.items():
            e0, e1 = c.event(), c.event()
            e0.record()
            for _ in range(200):
                m.step()
            e1.record()
            us = e0.elapsed_ms(e1) / 200 * 1e3
            c.profile(4)
            for _ in range(40):
                m.step()
            ms2, n2 = c.profile_read(2)
            ms0, n0 = c.profile_read(0)
            c.profile(0)
            print("round %d  finish block %4d: %7.2f us per update; pass %6.1f us, finish %5.2f us (event pairs)"
                  % (r, blk, us, ms0 / n0 * 1e3, ms2 / n2 * 1e3), flush=True)
    ref = models[1024][1].lam.cpu()
    for blk in (512, 256):
        print("max |lam(%d) - lam(1024)| after the same number of updates: %.2e" % (blk, (models[blk][1].lam.cpu() - ref).abs().max().item()))


if __name__ == "__main__":
    main()
