"""Interleaved A/B of the config-2 pass kernels in ONE process (round 4): blr_pass_dma_kernel (BSC_BLR_Q=0) against
blr_pass_q_kernel (both contractions on v_mfma_f32_4x4x1) and its deletion builds, at the metric's mini-batch and at
its 2/4/8-GPU shares.  Every launch reads a DIFFERENT resident mini-batch (a 1M x 256 buffer x 3 cut into pieces of the
row count under test), so nothing is served from the Infinity Cache.

    python tools/ab_q.py [rounds] [launches_per_round] [rows,rows,...]  [variants=...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd._ffi import ptr
from bayesic_amd.device import Context

VARIANTS = {
    # name: options of the context (bsc_ctx_set_option)
    "dma": dict(blr_q=0),
    "q": dict(blr_q=1),
    "q-prio1": dict(blr_q=1, blr_q_prio=1),
    "q-prio2": dict(blr_q=1, blr_q_prio=2),
    "q-bias40": dict(blr_q=1, blr_q_bias=40),
    "q-bias100p1": dict(blr_q=1, blr_q_bias=100, blr_q_prio=1),
    "q-bias0": dict(blr_q=1, blr_q_bias=0),
    "q-bias100": dict(blr_q=1, blr_q_bias=100),
    "q-bias130": dict(blr_q=1, blr_q_bias=130),
    "q-steal1": dict(blr_q=1, blr_steal=1),
    "q-steal60": dict(blr_q=1, blr_steal=60),
    "q-steal100": dict(blr_q=1, blr_steal=100),
    "q-steal150": dict(blr_q=1, blr_steal=150),
    "q-steal200": dict(blr_q=1, blr_steal=200),
    "q-steal350": dict(blr_q=1, blr_steal=350),
    "q-steal250": dict(blr_q=1, blr_steal=250),
    "q-nocompute": dict(profiling_builds=1, blr_q=1, blr_q_dbg=1),
    "q-fwdonly": dict(profiling_builds=1, blr_q=1, blr_q_dbg=2),
    "q-bwdonly": dict(profiling_builds=1, blr_q=1, blr_q_dbg=3),
}


def make_ctx(options):
    c = Context(0, options=options)
    c.reserve(32 << 20)
    return c


def stamp_report(c, name):
    st = c.read_stamps()
    if not len(st):
        return
    t0 = st[:, 0].min()
    start = (st[:, 0] - t0).astype(np.float64) / 100.0
    end = (st[:, 1] - t0).astype(np.float64) / 100.0
    xcd = (st[:, 2] & 7).astype(int)
    q = np.percentile(end, [0, 10, 50, 90, 100])
    print("    %-12s workgroup END stamps, us after the first start: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f; "
          "starts span %.1f us" % (name, q[0], q[1], q[2], q[3], q[4], start.max()))
    print("    %-12s per XCD mean end:" % name, " ".join("[%d] %.0f" % (x, end[xcd == x].mean()) for x in range(8) if (xcd == x).any()))
    # the two workgroups that shared a CU (same XCD and HW_ID cu/sh/se bits) -- do they finish together?
    key = xcd * 65536 + ((st[:, 3] >> 8) & 0xff).astype(int)
    per_cu = {}
    for k, e in zip(key, end):
        per_cu.setdefault(int(k), []).append(e)
    cu_end = np.array([max(v) for v in per_cu.values()])
    cu_par = np.array([(k >> 16) & 1 for k in per_cu])
    print("    %-12s per CU (end of its last workgroup): even XCDs mean %.1f max %.1f | odd XCDs mean %.1f max %.1f; all: p10 %.1f p50 %.1f p90 %.1f"
          % (name, cu_end[cu_par == 0].mean(), cu_end[cu_par == 0].max(), cu_end[cu_par == 1].mean(), cu_end[cu_par == 1].max(),
             np.percentile(cu_end, 10), np.percentile(cu_end, 50), np.percentile(cu_end, 90)))
    if VARIANTS[name].get("blr_steal"):
        a = (st[:, 4] - t0).astype(np.float64) / 100.0
        b = (st[:, 5] - t0).astype(np.float64) / 100.0
        print("    %-12s wave 0 of a workgroup: static share read at p10 %.1f p50 %.1f p90 %.1f us; left the queue at p10 %.1f p50 %.1f p90 %.1f max %.1f;"
              " queued tiles taken mean %.2f min %d max %d; end - left p50 %.1f max %.1f"
              % (name, *np.percentile(a, [10, 50, 90]), *np.percentile(b, [10, 50, 90, 100]), st[:, 6].mean(), st[:, 6].min(), st[:, 6].max(),
                 np.median(end - b), (end - b).max()))
        print("    %-12s queued tiles taken by wave 0, mean per XCD:" % name, " ".join("[%d] %.2f" % (x, st[xcd == x, 6].mean()) for x in range(8) if (xcd == x).any()))
    pairs = [v for v in per_cu.values() if len(v) == 2]
    if pairs:
        d = np.array([abs(a - b) for a, b in pairs])
        print("    %-12s %d CUs held two workgroups: |end difference| median %.1f us, max %.1f us; %d distinct (XCD, CU) keys"
              % (name, len(pairs), np.median(d), d.max(), len(per_cu)))


def main():
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    rounds = int(nums[0]) if len(nums) > 0 else 6
    per = int(nums[1]) if len(nums) > 1 else 24
    sizes = [1_000_000, 500_000, 250_000, 125_000]
    names = list(VARIANTS)
    for a in sys.argv[1:]:
        if "," in a and a.replace(",", "").isdigit():
            sizes = [int(v) for v in a.split(",")]
        if a.startswith("variants="):
            names = a[len("variants="):].split(",")
    D, S = 256, 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1234)
    total = 3_000_000
    X = torch.randn((total, D), generator=g, device=dev)
    y = torch.randn(total, generator=g, device=dev)
    W = torch.randn((S, D), generator=g, device=dev) / 16
    ctxs = {n: make_ctx(VARIANTS[n]) for n in names}
    print("pure-read probe on this box: %.0f GB/s" % next(iter(ctxs.values())).read_probe(X[:1_000_000]), flush=True)
    for B in sizes:
        nb = total // B
        res = {n: [] for n in names}
        visit = 0

        def launch(c):
            nonlocal visit
            visit = (visit + 1) % nb
            r0 = visit * B
            c.call("bsc_blr_data_pass_partial", ptr(X[r0:]), D, ptr(y[r0:]), B, D, ptr(W), S)

        for c in ctxs.values():
            for _ in range(8):
                launch(c)
            c.sync()
        for r in range(rounds):
            for n, c in ctxs.items():
                c.profile(True)
                for _ in range(per):
                    launch(c)
                ms, cnt = c.profile_read()
                c.profile(False)
                res[n].append(ms / cnt * 1e3)
        bytes_ = 4.0 * B * D + 4.0 * B
        if "stamps" in sys.argv:
            for n, c in ctxs.items():
                if VARIANTS[n].get("blr_q", 1):
                    c.set_option("blr_stamps", 1)
                    for _ in range(4):
                        launch(c)
                    stamp_report(c, n)
                    c.set_option("blr_stamps", 0)
        for n in names:
            a = np.array(res[n])
            print("rows=%8d  %-12s per-launch us: median %7.2f  min %7.2f  max %7.2f  -> %5.0f GB/s = %.3f of 8 TB/s"
                  % (B, n, np.median(a), a.min(), a.max(), bytes_ / np.median(a) / 1e3, bytes_ / np.median(a) / 1e3 / 8000),
                  flush=True)


if __name__ == "__main__":
    main()
