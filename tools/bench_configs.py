"""Per-config dominant-kernel timings at the BASELINE sizes (one GPU), with the
roofline each is bound by.  Not the driver's bench (that is bench.py, config 2);
this is the evidence table for DESIGN.md section 4.

    python tools/bench_configs.py [--quick]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd import algebra as A
from bayesic_amd.algebra.device_backend import DeviceBackend
from bayesic_amd.device import Context

HBM, F32 = 8.0e12, 157.3e12


def timed(ctx, fn, reps, warm=3, warm_ms=60.0):
    """Mean wall / kernel time of `reps` calls after at least `warm` calls AND `warm_ms` of
    uninterrupted GPU time of the same call: a kernel reaches its steady rate only after
    ~35 ms of continuous running (tools/ramp_probe.py: 228 -> 164 us for the data pass), and
    other kernels or idle gaps in between do not count."""
    e0, e1 = ctx.event(), ctx.event()
    done, elapsed = 0, 0.0
    while done < warm or elapsed < warm_ms:
        e0.record()
        for _ in range(max(warm, 1)):
            fn()
        e1.record()
        elapsed += e0.elapsed_ms(e1)
        done += max(warm, 1)
    ctx.profile(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    wall = e0.elapsed_ms(e1) / reps * 1e-3
    ms, n = ctx.profile_read()
    ctx.profile(0)
    kern = (ms / n * 1e-3) if n else float("nan")
    # per-call distribution (SURVEY 8(d) asks for median and min): one event pair per call, which
    # adds ~4 us of stream time to each -- kept out of the mean above
    evs = [(ctx.event(), ctx.event()) for _ in range(min(reps, 20))]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    per = sorted(a.elapsed_ms(b) * 1e3 for a, b in evs)
    LAST_DISTRIBUTION.update(median_us=per[len(per) // 2], min_us=per[0])
    return wall, kern


LAST_DISTRIBUTION = {}


def main():
    quick = "--quick" in sys.argv
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    want = lambda name: not only or any(o in name for o in only)
    ctx = Context(0)
    dev = ctx.device
    g = torch.Generator(device=dev).manual_seed(0)
    out = []

    def row(name, wall, kern, nbytes, flops, bound):
        t = kern if kern == kern else wall
        rec = dict(config=name, wall_us=wall * 1e6, kernel_us=kern * 1e6,
                   call_median_us=LAST_DISTRIBUTION.get("median_us"), call_min_us=LAST_DISTRIBUTION.get("min_us"),
                   bytes=nbytes, flops=flops,
                   GBps=nbytes / t / 1e9, TFLOPs=flops / t / 1e12, hbm_frac=nbytes / t / HBM,
                   f32_frac=flops / t / F32, bound=bound)
        out.append(rec)
        print(json.dumps(rec), flush=True)

    ctx.reserve(64 << 20)
    if want("cfg2"):
        # ---- config 2: BLR reparam pass ------------------------------------------
        N, D, S = 1_000_000, 256, 8
        X = torch.randn((N, D), generator=g, device=dev)
        y = torch.randn(N, generator=g, device=dev)
        W = torch.randn((S, D), generator=g, device=dev) / 16
        w, k = timed(ctx, lambda: ctx.call("bsc_blr_data_pass_partial", X, D, y, N, D, W, S), 50, warm=20)
        row("cfg2 blr_pass 1Mx256 S=8", w, k, 4.0 * N * D + 4 * N, 4.0 * N * D * S, "hbm")

        # ---- config 2': conjugate statistic X^T X through the executor -----------------
        be = DeviceBackend(ctx)
        Xs = A.var("X", 2)
        gram = A.dot(Xs.T, Xs).compile(be).device_fn
        w, k = timed(ctx, lambda: gram(X=X), 5)
        # the symmetric schedule multiplies the tiles on and above the diagonal only: 3 of the 4 128 x 128
        # tiles here, and that is what the MFMA fraction is taken over (the full 2 N D^2 would read 1.03 of peak)
        tiles = D // 128
        row("cfg2' gram X^T X 256x256x1M (upper-triangle tiles: %d of %d)" % (tiles * (tiles + 1) // 2, tiles * tiles),
            w, k, 4.0 * N * D, 2.0 * N * 128 * 128 * (tiles * (tiles + 1) // 2), "f32-mfma")
        xty = A.dot(Xs.T, A.var("y", 1)).compile(be).device_fn
        w, k = timed(ctx, lambda: xty(X=X, y=y), 5)
        row("cfg2' X^T y", w, k, 4.0 * N * D, 2.0 * N * D, "hbm")
        del X, y

    if want("gemm"):
        # ---- the _tensordot GEMM on a plain square problem (reference point for its tiling) ----
        for n in (4096, 4224, 8192):
            Am = torch.randn((n, n), generator=g, device=dev)
            Bm = torch.randn((n, n), generator=g, device=dev)
            Cm = torch.empty((n, n), device=dev)
            w, k = timed(ctx, lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, n, n, n, Am, 0, n, 1,
                                               Bm, 0, n, 1, Cm, 0, n, 1), 10, warm=3)
            row("gemm %dx%dx%d f32 (A k-contiguous, B n-contiguous)" % (n, n, n), w, k, 12.0 * n * n,
                2.0 * n ** 3, "f32-mfma")
            w, k = timed(ctx, lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, n, n, n, Am, 0, 1, n,
                                               Bm, 0, n, 1, Cm, 0, n, 1), 10, warm=3)
            row("gemm %dx%dx%d f32 (A m-contiguous, B n-contiguous)" % (n, n, n), w, k, 12.0 * n * n,
                2.0 * n ** 3, "f32-mfma")
            del Am, Bm, Cm

    if want("cfg3"):
        # ---- config 3: MoG E-step + statistics ----------------------------------------
        N3, D3, K3 = (2_000_000 if quick else 10_000_000), 16, 64
        X3 = torch.randn((N3, D3), generator=g, device=dev) * 3
        Wm = torch.randn((K3, 2 * D3), generator=g, device=dev) * 0.1
        Wm[:, D3:] = -0.5
        c = torch.zeros(K3, device=dev)
        stats = torch.zeros(K3 * (1 + 2 * D3), dtype=torch.float64, device=dev)
        lse = torch.zeros(1, dtype=torch.float64, device=dev)
        w, k = timed(ctx, lambda: ctx.call("bsc_mog_estep", X3, D3, N3, D3, K3, Wm, c, stats, lse), 20, warm=10)
        row("cfg3 mog_estep %dMx16 K=64" % (N3 // 1_000_000), w, k, 4.0 * N3 * D3,
            8.0 * K3 * D3 * N3, "f32-mfma")
        from bayesic_amd.svi import mog as mog_mod
        from bayesic_amd.svi.mog import MoGNatGradSVI
        eta0 = mog_mod.prior_eta(K3, D3)
        eta_init = mog_mod.init_eta(X3[:2000].cpu().numpy(), K3, D3, seed=2)
        mog = MoGNatGradSVI(X3, K3, eta0, eta_init, n_total=float(N3), ctx=ctx)
        w, _ = timed(ctx, mog.step, 20, warm=10)
        row("cfg3 whole MoG update (expected params + E-step + natural-gradient step)", w,
            float("nan"), 4.0 * N3 * D3, 8.0 * K3 * D3 * N3, "f32-mfma")
        # full-covariance statistic sum_n r_nk x_n x_n^T: one pass instead of the K x D x N product
        Rm = torch.softmax(torch.randn((N3, K3), generator=g, device=dev), dim=1)
        cov = torch.empty((K3, D3, D3), device=dev)
        w, k = timed(ctx, lambda: ctx.call("bsc_weighted_outer", Rm, K3, X3, D3, X3, D3, N3, K3, D3,
                                           D3, 1.0, cov), 10, warm=5)
        pairs = D3 * (D3 + 1) // 2
        row("cfg3' weighted second moment %dMx16 K=64 (bsc_weighted_outer, d<=e pairs)"
            % (N3 // 1_000_000), w, k, 4.0 * N3 * (K3 + D3), 2.0 * N3 * K3 * pairs, "f32-mfma")
        del X3, mog, Rm

    if want("cfg5"):
        # ---- config 5: BBVI log-likelihood pass ------------------------------------------
        N5, D5, G5, S5 = 1_000_000, 256, 1000, 64
        X5 = torch.randn((N5, D5), generator=g, device=dev)
        y5 = (torch.rand(N5, generator=g, device=dev) < 0.4).float()
        g5 = torch.randint(0, G5, (N5,), generator=g, device=dev, dtype=torch.int32)
        Wz = torch.randn((S5, D5), generator=g, device=dev) / 16
        Bz = torch.randn((G5, S5), generator=g, device=dev)
        ell = torch.zeros(S5, dtype=torch.float64, device=dev)
        w, k = timed(ctx, lambda: ctx.call("bsc_logreg_bbvi_loglik", X5, D5, y5, g5, N5, D5, G5, Wz,
                                           Bz, S5, ell), 20, warm=10)
        row("cfg5 logreg_loglik 1Mx256 S=64", w, k, 4.0 * N5 * D5 + 8.0 * N5, 2.0 * N5 * D5 * S5,
            "hbm/f32-mfma")
        from bayesic_amd.svi.bbvi import LogRegBBVI
        bb = LogRegBBVI(X5, y5, g5, G5, n_samples=S5, ctx=ctx)
        w, _ = timed(ctx, bb.step, 20, warm=10)
        row("cfg5 whole BBVI update (sample + pass + control variate + Adam)", w, float("nan"),
            4.0 * N5 * D5 + 8.0 * N5, 2.0 * N5 * D5 * S5, "hbm/f32-mfma")
        del X5, bb

    if want("cfg4"):
        # ---- config 4: LDA local step on one GPU's shard -----------------------------------
        from bayesic_amd.svi.lda import LDAFixedGammaSVI
        docs, V, K4 = (1000 if quick else 6250), 100_000, 128
        C = torch.poisson(torch.full((docs, V), 0.05, device=dev), generator=g)
        gamma = torch.rand((docs, K4), generator=g, device=dev) + 0.5
        lam = torch.rand((K4, V), generator=g, device=dev) + 0.5
        model = LDAFixedGammaSVI(C, gamma, lam, docs_total=50_000, ctx=ctx)
        w, k = timed(ctx, lambda: ctx.call("bsc_lda_sstats", model.C, V, docs, V, K4, model.Th, K4,
                                           model.Bt, V, model.sstats, V), 5, warm=2)
        row("cfg4 lda_sstats %dx100k K=128 (fused kernel)" % docs, w, k, 4.0 * docs * V,
            4.0 * docs * V * K4, "f32-mfma")
        w, _ = timed(ctx, model.step, 3, warm=1)
        row("cfg4 lda_step %dx100k K=128 (whole step, fused kernel)" % docs, w, float("nan"),
            4.0 * docs * V, 4.0 * docs * V * K4, "f32-mfma")
        # the same counts (Poisson 0.05 -> ~4.9 % nonzero) walked as compressed sparse columns
        nz = (C.t() != 0).nonzero()                  # sorted by word, then document
        counts_per_word = torch.bincount(nz[:, 0], minlength=V)
        colptr = torch.zeros(V + 1, dtype=torch.int64, device=dev)
        colptr[1:] = torch.cumsum(counts_per_word, 0)
        rowidx = nz[:, 1].to(torch.int32).contiguous()
        vals = C.t()[nz[:, 0], nz[:, 1]].contiguous()
        nnz = int(vals.numel())
        out_s = torch.empty_like(model.sstats)
        w, k = timed(ctx, lambda: ctx.call("bsc_lda_sstats_csc", colptr, rowidx, vals, docs, V, K4,
                                           model.Th, K4, model.Bt, V, out_s, V), 5, warm=2)
        row("cfg4 lda_sstats_csc %dx100k K=128, %d nonzeros (%.1f%%)" % (docs, nnz, 100.0 * nnz / (docs * V)),
            w, k, 12.0 * nnz + 4.0 * nnz * K4, 4.0 * nnz * K4, "l2/valu")
        del nz, counts_per_word, rowidx, vals
        ex = LDAFixedGammaSVI(C, gamma, lam, docs_total=50_000, ctx=ctx, via="executor")
        w, _ = timed(ctx, ex.step, 3, warm=1)
        row("cfg4 lda_step %dx100k K=128 (whole step, executor: 2 GEMMs + fused elementwise)" % docs,
            w, float("nan"), 4.0 * docs * V, 4.0 * docs * V * K4, "f32-mfma")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
