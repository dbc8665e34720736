"""Launches ONE config's dominant kernel a few times at its BASELINE size, for rocprofv3 --pmc
passes (tools/pmc_collect.sh).  Inputs are random device tensors (counters do not depend on the
values); the call is the same C-ABI entry point bench.py / the drivers use.

    python3 tools/pmc_run.py cfg2|cfg2rot|cfg3|cfg3x|cfg4|cfg4b|cfg4x2|cfg4x3|cfg5|cfg5x2|rowsoftmax|softstats|wouter|gram|skinny|lda1|lda1e|lda2|sq4096nt|sq4096tn [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bayesic_amd.device import Context


def main():
    which = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    options = {}
    if which == "cfg3l2":             # config 3's E-step with X through the default cache policy (option mog_nt = 0)
        options["mog_nt"] = 0
        which = "cfg3"
    ctx = Context(0, options=options)
    dev = ctx.device
    g = torch.Generator(device=dev).manual_seed(0)
    ctx.reserve(64 << 20)
    if which == "cfg2":
        N, D, S = 1_000_000, 256, 8
        X = torch.randn((N, D), generator=g, device=dev)
        y = torch.randn(N, generator=g, device=dev)
        W = torch.randn((S, D), generator=g, device=dev) / 16
        # as the driver issues it: passes over the resident batch alternate their direction
        turn = [0]

        def fn():
            turn[0] += 1
            ctx.call("bsc_blr_data_pass_partial_sweep", X, D, y, N, D, W, S, 1 + (turn[0] & 1))
    elif which == "cfg2stream":
        N, D, S = 1_000_000, 256, 8
        X = torch.randn((N, D), generator=g, device=dev)
        y = torch.randn(N, generator=g, device=dev)
        W = torch.randn((S, D), generator=g, device=dev) / 16
        fn = lambda: ctx.call("bsc_blr_data_pass_partial", X, D, y, N, D, W, S)
    elif which == "cfg2rot":
        # as bench.py's default loop issues it since round 3: a DIFFERENT resident mini-batch every launch, streamed
        N, D, S, nb = 1_000_000, 256, 8, 3
        Xs = [torch.randn((N, D), generator=g, device=dev) for _ in range(nb)]
        y = torch.randn(N, generator=g, device=dev)
        W = torch.randn((S, D), generator=g, device=dev) / 16
        turn = [0]

        def fn():
            turn[0] += 1
            ctx.call("bsc_blr_data_pass_partial_sweep", Xs[turn[0] % nb], D, y, N, D, W, S, 0)
    elif which == "softstats":
        # the derived mixture's WHOLE local step since round 3: softmax of [X | X^2] . coefficients + bias and the
        # statistics R^T [X | X^2 | 1] in one pass, responsibilities not written (10M rows of a 40-column wide operand)
        N, K, C = 10_000_000, 32, 64
        A = torch.randn((N, 40), generator=g, device=dev)
        A[:, 32] = 1.0
        B = torch.randn((40, C), generator=g, device=dev) * 0.3
        stats = torch.empty((C, 40), device=dev)
        lse = torch.empty(1, dtype=torch.float64, device=dev)
        fn = lambda: ctx.call("bsc_gemm_softmax_stats", A, 40, N, K, B, C, 1, C, 1.0, B[32], None, C, stats, 40, lse)
    elif which == "rowsoftmax":
        # the derived mixture's local step: softmax_rows([X | X^2 | 1 | 0..] . coefficients), 10M x 40 -> 64
        N, K, C = 10_000_000, 40, 64
        A = torch.randn((N, K), generator=g, device=dev)
        B = torch.randn((K, C), generator=g, device=dev) * 0.3
        R = torch.empty((N, C), device=dev)
        lse = torch.empty(N, device=dev)
        cross = torch.empty(N, device=dev)
        fn = lambda: ctx.call("bsc_gemm_softmax_rows", A, K, N, K, B, C, 1, C, 1.0, R, C, lse, cross)
    elif which == "cfg3":
        N, D, K = 10_000_000, 16, 64
        X = torch.randn((N, D), generator=g, device=dev) * 3
        Wm = torch.randn((K, 2 * D), generator=g, device=dev) * 0.1
        Wm[:, D:] = -0.5
        c = torch.zeros(K, device=dev)
        stats = torch.zeros(K * (1 + 2 * D), dtype=torch.float64, device=dev)
        lse = torch.zeros(1, dtype=torch.float64, device=dev)
        fn = lambda: ctx.call("bsc_mog_estep", X, D, N, D, K, Wm, c, stats, lse)
    elif which == "cfg3x":
        # config 3's pass on the bf16 MFMA (forward on three terms, backward on two)
        N, D, K = 10_000_000, 16, 64
        X = torch.randn((N, D), generator=g, device=dev) * 3
        Wm = torch.randn((K, 2 * D), generator=g, device=dev) * 0.1
        Wm[:, D:] = -0.5
        c = torch.zeros(K, device=dev)
        stats = torch.zeros(K * (1 + 2 * D), dtype=torch.float64, device=dev)
        lse = torch.zeros(1, dtype=torch.float64, device=dev)
        ctx.call("bsc_ctx_set_mfma_split", 2)
        fn = lambda: ctx.call("bsc_mog_estep", X, D, N, D, K, Wm, c, stats, lse)
    elif which == "wouter":
        N, D, K = 10_000_000, 16, 64
        X = torch.randn((N, D), generator=g, device=dev) * 3
        R = torch.softmax(torch.randn((N, K), generator=g, device=dev), dim=1)
        cov = torch.empty((K, D, D), device=dev)
        fn = lambda: ctx.call("bsc_weighted_outer", R, K, X, D, X, D, N, K, D, D, 1.0, cov)
    elif which == "cfg5":
        N, D, G, S = 1_000_000, 256, 1000, 64
        X = torch.randn((N, D), generator=g, device=dev)
        y = (torch.rand(N, generator=g, device=dev) < 0.4).float()
        gi = torch.randint(0, G, (N,), generator=g, device=dev, dtype=torch.int32)
        Wz = torch.randn((S, D), generator=g, device=dev) / 16
        Bz = torch.randn((G, S), generator=g, device=dev)
        ell = torch.zeros(S, dtype=torch.float64, device=dev)
        fn = lambda: ctx.call("bsc_logreg_bbvi_loglik", X, D, y, gi, N, D, G, Wz, Bz, S, ell)
    elif which == "cfg4":
        docs, V, K = 6250, 100_000, 128
        C = torch.poisson(torch.full((docs, V), 0.05, device=dev), generator=g)
        Th = torch.rand((docs, K), generator=g, device=dev) + 0.5
        Bt = torch.rand((K, V), generator=g, device=dev) + 0.5
        out = torch.empty((K, V), device=dev)
        fn = lambda: ctx.call("bsc_lda_sstats", C, V, docs, V, K, Th, K, Bt, V, out, V)
    elif which == "cfg4b":
        # ... with the words' term of the bound accumulated in the same pass (what the driver issues since round 3)
        docs, V, K = 6250, 100_000, 128
        C = torch.poisson(torch.full((docs, V), 0.05, device=dev), generator=g)
        Th = torch.rand((docs, K), generator=g, device=dev) + 0.5
        Bt = torch.rand((K, V), generator=g, device=dev) + 0.5
        out = torch.empty((K, V), device=dev)
        ll = torch.empty(1, dtype=torch.float64, device=dev)
        fn = lambda: ctx.call("bsc_lda_sstats_bound", C, V, docs, V, K, Th, K, Bt, V, out, V, ll)
    elif which in ("cfg4x2", "cfg4x3"):
        # ... on the operand-split bf16 route (bsc_ctx_set_mfma_split 2 / 3)
        docs, V, K = 6250, 100_000, 128
        C = torch.poisson(torch.full((docs, V), 0.05, device=dev), generator=g)
        Th = torch.rand((docs, K), generator=g, device=dev) + 0.5
        Bt = torch.rand((K, V), generator=g, device=dev) + 0.5
        out = torch.empty((K, V), device=dev)
        ll = torch.empty(1, dtype=torch.float64, device=dev)
        ctx.call("bsc_ctx_set_mfma_split", int(which[-1]))
        fn = lambda: ctx.call("bsc_lda_sstats_bound", C, V, docs, V, K, Th, K, Bt, V, out, V, ll)
    elif which == "cfg5x2":
        # config 5's pass with X and the draws as two bf16 terms
        import math
        N, D, G, S = 1_000_000, 256, 1000, 64
        X = torch.randn((N, D), generator=g, device=dev)
        y = (torch.rand(N, generator=g, device=dev) < 0.4).float()
        ids = torch.randint(G, (N,), generator=g, device=dev, dtype=torch.int32)
        Wz = torch.randn((S, D), generator=g, device=dev) / math.sqrt(D)
        Bz = torch.randn((G, S), generator=g, device=dev)
        ell = torch.empty(S, dtype=torch.float64, device=dev)
        ctx.call("bsc_ctx_set_mfma_split", 2)
        fn = lambda: ctx.call("bsc_logreg_bbvi_loglik", X, D, y, ids, N, D, G, Wz, Bz, S, ell)
    elif which == "gram":
        N, D = 1_000_000, 256
        X = torch.randn((N, D), generator=g, device=dev)
        C = torch.empty((D, D), device=dev)
        fn = lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, D, D, N, X, 0, 1, D, X, 0, D, 1, C, 0, D, 1)
    elif which == "gramx":
        N, D = 1_000_000, 256
        X = torch.randn((N, D), generator=g, device=dev)
        C = torch.empty((D, D), device=dev)
        ctx.call("bsc_ctx_set_mfma_split", 2)
        fn = lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, D, D, N, X, 0, 1, D, X, 0, D, 1, C, 0, D, 1)
    elif which == "skinny":
        N, D, S = 1_000_000, 256, 8
        X = torch.randn((N, D), generator=g, device=dev)
        W = torch.randn((S, D), generator=g, device=dev)
        P = torch.empty((S, N), device=dev)
        fn = lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, S, N, D, W, 0, D, 1, X, 0, 1, D, P, 0, N, 1)
    elif which in ("lda1", "lda1e", "lda2"):
        # config 4's statistic through the executor: Q = C / dot(Th, Bt), S = Bt * dot(Th.T, Q)
        docs, V, K = 6250, 100_000, 128
        Th = torch.rand((docs, K), generator=g, device=dev) + 0.1
        Bt = torch.rand((K, V), generator=g, device=dev) + 0.1
        Cn = torch.rand((docs, V), generator=g, device=dev)
        Q = torch.empty((docs, V), device=dev)
        S = torch.empty((K, V), device=dev)
        if which == "lda1":
            fn = lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, docs, V, K, Th, 0, K, 1, Bt, 0, V, 1, Q, 0, V, 1)
        elif which == "lda1e":
            fn = lambda: ctx.call("bsc_gemm_epilogue", 0, 1, docs, V, K, Th, 0, K, 1, Bt, 0, V, 1, Q, 0, V, 1, -1, 1.0, Cn, 0, V, 1)
        else:
            fn = lambda: ctx.call("bsc_gemm_epilogue", 0, 1, K, V, docs, Th, 0, 1, K, Cn, 0, V, 1, S, 0, V, 1, 1, 1.0, Bt, 0, V, 1)
    elif which in ("sq4096nt", "sq4096tn"):
        n = 4096
        A = torch.randn((n, n), generator=g, device=dev)
        B = torch.randn((n, n), generator=g, device=dev)
        C = torch.empty((n, n), device=dev)
        if which == "sq4096nt":
            fn = lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, n, n, n, A, 0, n, 1, B, 0, 1, n, C, 0, n, 1)
        else:
            fn = lambda: ctx.call("bsc_gemm_strided_batched", 0, 1, n, n, n, A, 0, 1, n, B, 0, n, 1, C, 0, n, 1)
    else:
        raise SystemExit("unknown config %r" % which)
    for _ in range(reps):
        fn()
    ctx.sync()


if __name__ == "__main__":
    main()
