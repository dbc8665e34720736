"""Interleaved A/B timing of blr_pass_kernel variants in ONE process
(tile rows 4 vs 8), hipEvents around the kernel alone.

    python tools/ab_pass.py [rounds] [launches_per_round]
"""
import os
os.environ.setdefault("BSC_PROFILING_BUILDS", "1")   # the Context honours BSC_<OPTION> variables only in a process that opts in (device.py)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd._ffi import ptr
from bayesic_amd.device import Context


def main():
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    rounds = int(nums[0]) if len(nums) > 0 else 8
    per = int(nums[1]) if len(nums) > 1 else 20
    B, D, S = 1_000_000, 256, 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1234)
    X = torch.randn((B, D), generator=g, device=dev)
    y = torch.randn(B, generator=g, device=dev)
    W = torch.randn((S, D), generator=g, device=dev) / 16
    ctxs = {}
    # (tile rows, waves/SIMD cap, non-temporal loads); append "pk" on the command line to
    # compare the MFMA pass with scalar vs packed backward FMAs instead
    variants = [(8, 0, 1), (16, 0, 1)]
    pk_mode = "pk" in sys.argv
    if pk_mode:
        variants = [(16, 0, 1), (16, 1000, 1)]      # second entry: BSC_BLR_PK=1 (cap field reused as a tag)
    if "wide" in sys.argv:
        # S > 8: sixteen draws per pass (BSC_BLR_WIDE=1, default) against eight per pass
        res = {}
        for S_ in (16, 64):
            Ww = torch.randn((S_, D), generator=g, device=dev) / 16
            Q = torch.zeros(S_, dtype=torch.float64, device=dev)
            G = torch.zeros((S_, D), dtype=torch.float64, device=dev)
            for wide in ("1", "0"):
                os.environ["BSC_BLR_WIDE"] = wide
                c = Context(0)
                c.reserve(32 << 20)
                for alt in (0, 1):
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    times = []
                    for rep in range(rounds + 2):
                        torch.cuda.synchronize()
                        ev0.record(torch.cuda.current_stream())
                        for i in range(per):
                            c.call("bsc_blr_data_pass_sweep", ptr(X), D, ptr(y), B, D, ptr(Ww), S_, ptr(Q), ptr(G),
                                   (1 + (i & 1)) if alt else 0)
                        ev1.record(torch.cuda.current_stream())
                        torch.cuda.synchronize()
                        if rep >= 2:
                            times.append(ev0.elapsed_time(ev1) / per * 1e3)
                    print("S=%d draws per pass=%s sweeps=%s: %.1f us per bsc_blr_data_pass (median of %d bursts of %d)"
                          % (S_, "16" if wide == "1" else "8", "alternate" if alt else "stream",
                             float(np.median(times)), rounds, per))
        return
    sweep_mode = "sweep" in sys.argv
    if sweep_mode:
        # (alternate?, BSC_BLR_KEEP): passes that alternate BSC_SWEEP_FORWARD_KEEP / BACKWARD_KEEP with
        # `keep` trailing windows left in the Infinity Cache (-1 = the library's own choice) against
        # the plain streaming pass
        ctxs = {}
        # alt >= 10: fixed cached zone of `keep` windows + rotated schedule, rotation unit 2^(alt-10) workgroups
        # (alt, keep): alt 0 = streaming pass, 1 = alternating keeping sweeps; + 10 * BSC_BLR_MX
        # (10/11: blr_pass_mx_kernel with the rotated cached zone; 20/21: that kernel, plain sweeps);
        # + 100 * BSC_BLR_ROT
        # BSC_BLR_MX = 4: every position re-reads window 0 (cache hits only: the kernel's compute floor)
        for alt, keep in [(0, 0), (1, -1), (20, 0), (21, -1), (211, 7), (40, 0), (41, 33)]:
            os.environ["BSC_BLR_KEEP"] = str(keep)
            os.environ["BSC_BLR_MX"] = str((alt % 100) // 10)
            os.environ["BSC_PROFILING_BUILDS"] = "1"      # (BSC_BLR_MX=4 is a deletion build: wrong results, timing only)
            os.environ["BSC_BLR_ROT"] = str(alt // 100)
            os.environ["BSC_BLR_TILE_ROWS"] = "16"
            ctxs[(alt, keep)] = Context(0)
            ctxs[(alt, keep)].reserve(16 << 20)
        variants = []
    for rows, wps, nt in variants:
        os.environ["BSC_BLR_PK"] = "1" if (pk_mode and wps == 1000) else "0"
        if pk_mode:
            wps = 0 if wps != 1000 else 1000
        os.environ["BSC_BLR_TILE_ROWS"] = str(rows)
        os.environ["BSC_BLR_WAVES_PER_SIMD"] = str(0 if wps == 1000 else wps)
        os.environ["BSC_BLR_NT"] = str(nt)
        ctxs[(rows, wps, nt)] = Context(0)
        ctxs[(rows, wps, nt)].reserve(16 << 20)
    res = {k: [] for k in ctxs}
    def launch(key, c, i):
        if sweep_mode and key[0] % 10:
            c.call("bsc_blr_data_pass_partial_sweep", ptr(X), D, ptr(y), B, D, ptr(W), S, 1 + (i & 1))
        else:
            c.call("bsc_blr_data_pass_partial", ptr(X), D, ptr(y), B, D, ptr(W), S)

    for rows, c in ctxs.items():     # warm-up
        for i in range(5):
            launch(rows, c, i)
        c.sync()
    for r in range(rounds):
        for rows, c in ctxs.items():
            c.profile(True)
            for i in range(per):
                launch(rows, c, i)
            ms, n = c.profile_read()
            c.profile(False)
            res[rows].append(ms / n * 1e3)
    bytes_ = 4.0 * B * D + 4.0 * B
    print("pure-read probe on this box: %.0f GB/s" % next(iter(ctxs.values())).read_probe(X))
    if sweep_mode:
        for key in ctxs:
            a = np.array(res[key])
            print("alternate=%d keep=%d  per-launch us: median %.1f  min %.1f  max %.1f  -> %.0f GB/s (median)"
                  % (key + (np.median(a), a.min(), a.max(), bytes_ / np.median(a) / 1e3)))
        return
    for key in ctxs:
        a = np.array(res[key])
        print("rows=%d waves/SIMD cap=%d nt=%d  per-launch us: median %.1f  min %.1f  max %.1f  -> %.0f GB/s (median)"
              % (key + (np.median(a), a.min(), a.max(), bytes_ / np.median(a) / 1e3)))


if __name__ == "__main__":
    main()
