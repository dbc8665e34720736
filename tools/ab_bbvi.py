"""In-process A/B of the two bsc_logreg_bbvi_loglik kernels at config 5's size (1M x 256, G = 1000,
S = 64): interleaved bursts after a time-based warm-up, kernel time from hipEvent pairs.

    python tools/ab_bbvi.py [rounds]
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
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = int(pos[0]) if pos else 6
    S = int(pos[1]) if len(pos) > 1 else 64
    variants = {"lds-staged (r1)": make_ctx({"BSC_BBVI_KERNEL": "0"}),
                "X through VGPRs (r2a)": make_ctx({"BSC_BBVI_KERNEL": "2"}),
                "X by LDS-DMA (r2b, default)": make_ctx({"BSC_BBVI_KERNEL": "1"})}
    if S != 64:
        variants.pop("lds-staged (r1)")
    if "--deletion" in sys.argv:
        # profiling-only builds of the r2 kernel with parts removed (results wrong, time matters)
        for dbg, what in ((16, "no strip waits (reads race the DMAs)"), (1, "no X refill loads"), (2, "no epilogue"), (3, "no X loads, no epilogue"),
                          (11, "no X loads, epilogue, side loads"), (7, "MFMA + side loads only"),
                          (15, "MFMA only")):
            variants["r2b dbg=%d %s" % (dbg, what)] = make_ctx({"BSC_BBVI_KERNEL": "1", "BSC_BBVI_DBG": str(dbg), "BSC_PROFILING_BUILDS": "1"})
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    N, D, G = 1_000_000, 256, 1000
    X = torch.randn((N, D), generator=g, device=dev)
    y = (torch.rand(N, generator=g, device=dev) < 0.4).float()
    gi = torch.randint(0, G, (N,), generator=g, device=dev, dtype=torch.int32)
    Wz = torch.randn((S, D), generator=g, device=dev) / 16
    Bz = torch.randn((G, S), generator=g, device=dev)
    ells = {k: torch.zeros(S, dtype=torch.float64, device=dev) for k in variants}

    def run(name, n):
        c = variants[name]
        for _ in range(n):
            c.call("bsc_logreg_bbvi_loglik", X, D, y, gi, N, D, G, Wz, Bz, S, ells[name])

    for name in variants:            # warm-up by time: ~80 ms of the same call
        run(name, 250)
    torch.cuda.synchronize()
    flops = 2.0 * N * D * S
    for r in range(rounds):
        for name, c in variants.items():
            c.profile(1)
            run(name, 40)
            ms, n = c.profile_read()
            c.profile(0)
            us = ms / n * 1e3
            print("round %d  %-44s %7.1f us  %6.1f TF  %.3f of 157.3 TF  (%d launches)"
                  % (r, name, us, flops / us / 1e6, flops / us / 1e6 / 157.3, n), flush=True)
    names = [k for k in variants if "dbg" not in k]
    ref = ells[names[0]].cpu()
    for k in names[1:]:
        print("max relative difference %s vs %s: %.3e" % (k, names[0], ((ells[k].cpu() - ref).abs() / ref.abs()).max().item()))


if __name__ == "__main__":
    main()
