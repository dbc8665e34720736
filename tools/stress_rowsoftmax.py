"""Randomised soak of bsc_gemm_softmax_rows through the C ABI: random row counts (ragged, tiny, one
large), every K = 8 .. 64 and N = 4 .. 64 the entry point takes, leading dimensions above the extents,
both layouts of B, multipliers -- each against float64 numpy, twice (the second call must repeat the
first bit for bit).

    python tools/stress_rowsoftmax.py [cases] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd.device import Context


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = Context(0)
    rs = np.random.RandomState(seed)
    worst = 0.0
    for case in range(cases):
        K = 8 * rs.randint(1, 9)
        N = 4 * rs.randint(1, 17)
        rows = int([rs.randint(1, 40), rs.randint(1, 5000), rs.randint(5000, 400_000)][rs.randint(3)])
        lda = K + 4 * rs.randint(0, 4)
        ldr = N + 4 * rs.randint(0, 3)
        alpha = float([1.0, 0.1, 2.5][rs.randint(3)])
        tb = bool(rs.randint(2))
        A_ = np.full((rows, lda), np.nan, np.float32)           # padding columns must never be read
        A_[:, :K] = rs.standard_normal((rows, K))
        B_ = (rs.standard_normal((K, N)) * rs.uniform(0.2, 3.0)).astype(np.float32)
        Ad = ctx.to_device(A_)
        Bd = ctx.to_device(np.ascontiguousarray(B_.T)) if tb else ctx.to_device(B_)
        ldbk, ldbn = (1, K) if tb else (N, 1)
        outs = []
        for _ in range(2):
            R = torch.full((rows, ldr), 7.0, dtype=torch.float32, device=ctx.device)
            lse = ctx.zeros(rows, torch.float32)
            cross = ctx.zeros(rows, torch.float32)
            ctx.call("bsc_gemm_softmax_rows", Ad, lda, rows, K, Bd, ldbk, ldbn, N, alpha, R, ldr, lse, cross)
            ctx.sync()
            outs.append((R.cpu().numpy(), lse.cpu().numpy(), cross.cpu().numpy()))
        (R1, l1, c1), (R2, l2, c2) = outs
        assert np.array_equal(R1, R2) and np.array_equal(l1, l2) and np.array_equal(c1, c2), "not deterministic"
        assert (R1[:, N:] == 7.0).all(), "wrote beyond N columns of a row"
        logits = alpha * (A_[:, :K].astype(np.float64) @ B_.astype(np.float64))
        m = logits.max(axis=1, keepdims=True)
        e = np.exp(logits - m)
        want = e / e.sum(axis=1, keepdims=True)
        scale = np.abs(logits).max() + 1.0
        err = max(np.abs(R1[:, :N] - want).max() / 2e-5,
                  np.abs(l1 - (m + np.log(e.sum(axis=1, keepdims=True)))[:, 0]).max() / (1e-5 * scale),
                  np.abs(c1 - (want * logits).sum(axis=1)).max() / (2e-5 * scale))
        worst = max(worst, float(err))
        if not err <= 1.0 or not np.isfinite(err):
            print("MISMATCH case %d rows %d K %d N %d lda %d ldr %d alpha %g tb %d: err/tol %g"
                  % (case, rows, K, N, lda, ldr, alpha, tb, err))
            sys.exit(1)
        if case % 20 == 0:
            print("case %4d ok (rows %6d K %2d N %2d lda %2d ldr %2d alpha %.1f tb %d)  worst error / tolerance so far %.3f"
                  % (case, rows, K, N, lda, ldr, alpha, tb, worst), flush=True)
    print("all %d cases passed; worst error / tolerance %.3f" % (cases, worst))


if __name__ == "__main__":
    main()
