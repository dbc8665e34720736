"""Randomised soak of bsc_gemm_strided_batched / bsc_gemm_epilogue through the C ABI: random extents
(ragged, tiny, one large), the four operand layouts, leading dimensions above the extents, batches,
both epilogue powers with matrix / row / column factors -- each compared with a float64 product of
the same operands, and repeated to catch anything order- or timing-dependent.

    python tools/stress_gemm.py [cases] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd.device import Context


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = Context(0)
    dev = ctx.device
    rs = np.random.RandomState(seed)
    g = torch.Generator(device=dev).manual_seed(seed)
    worst = 0.0
    for case in range(cases):
        kind = rs.randint(5)
        if kind == 0:
            M, N, K = [int(4 * rs.randint(1, 200)) for _ in range(3)]
        elif kind == 1:
            M, N, K = int(rs.randint(1, 700)), int(rs.randint(1, 700)), int(rs.randint(1, 3000))
        elif kind == 2:
            M, N, K = int(4 * rs.randint(1, 40)), int(4 * rs.randint(1, 40)), int(4 * rs.randint(2000, 60000))
        elif kind == 3:
            M, N, K = int(4 * rs.randint(500, 6000)), int(4 * rs.randint(1, 40)), int(4 * rs.randint(1, 40))
        else:
            M, N, K = int(128 * rs.randint(1, 24)), int(128 * rs.randint(1, 24)), int(32 * rs.randint(1, 40))
        batch = int(rs.choice([1, 1, 1, 2, 3]))
        a_m, b_n = bool(rs.randint(2)), bool(rs.randint(2))
        pad = lambda n: n + int(4 * rs.randint(0, 3))
        # A stored [b][k][m] (m contiguous) or [b][m][k]; B stored [b][k][n] (n contiguous) or [b][n][k]
        if a_m:
            lda = pad(M); A = torch.randn((batch, K, lda), generator=g, device=dev); sa = (K * lda, 1, lda); A64 = A[:, :, :M].double().transpose(1, 2)
        else:
            lda = pad(K); A = torch.randn((batch, M, lda), generator=g, device=dev); sa = (M * lda, lda, 1); A64 = A[:, :, :K].double()
        if b_n:
            ldb = pad(N); B = torch.randn((batch, K, ldb), generator=g, device=dev); sb = (K * ldb, ldb, 1); B64 = B[:, :, :N].double()
        else:
            ldb = pad(K); B = torch.randn((batch, N, ldb), generator=g, device=dev); sb = (N * ldb, 1, ldb); B64 = B[:, :, :K].double().transpose(1, 2)
        ldc = pad(N)
        want = torch.matmul(A64, B64)
        bound = (A64.pow(2).sum(2).sqrt()[:, :, None] * B64.pow(2).sum(1).sqrt()[:, None, :])
        epi = int(rs.randint(4))          # 0: none, 1: * E, 2: / with E, 3: row-vector factor
        for rep in range(2):
            C = torch.full((batch, M, ldc), float("nan"), device=dev)
            if epi == 0:
                ctx.call("bsc_gemm_strided_batched", 0, batch, M, N, K, A, *sa, B, *sb, C, M * ldc, ldc, 1)
                ref, tol = want, 1e-5 * bound + 1e-12
            else:
                if epi == 3:
                    E = torch.randn((batch, 1, N), generator=g, device=dev) + 2.0; se = (N, 0, 1)
                else:
                    E = torch.randn((batch, M, N), generator=g, device=dev) + 2.0; se = (M * N, N, 1)
                power = -1 if epi == 2 else 1
                ctx.call("bsc_gemm_epilogue", 0, batch, M, N, K, A, *sa, B, *sb, C, M * ldc, ldc, 1, power, 0.5, E, *se)
                if power == 1:
                    ref, tol = 0.5 * want * E.double(), (1e-5 * bound + 1e-12) * E.double().abs()
                else:
                    ok = want.abs() > 0.05 * bound
                    ref = torch.where(ok, 0.5 * E.double() / want, torch.zeros_like(want))
                    got = C[:, :, :N].double()
                    got = torch.where(ok, got, torch.zeros_like(got))
                    err = ((got - ref).abs() / ref.abs().clamp_min(1e-30)).max().item() if ok.any() else 0.0
                    assert err < 5e-4, (case, M, N, K, batch, a_m, b_n, epi, err)
                    continue
            got = C[:, :, :N].double()
            bad = (got - ref).abs() > tol * 2
            assert not bad.any().item(), (case, M, N, K, batch, a_m, b_n, epi, ((got - ref).abs() / tol).max().item())
            assert torch.isnan(C[:, :, N:]).all().item() or ldc == N, "padding columns written"
            worst = max(worst, ((got - ref).abs() / tol).max().item())
        if case % 20 == 0:
            print("case %4d ok   (M %5d N %5d K %6d batch %d layouts %d%d epilogue %d)   worst error / tolerance so far %.3f"
                  % (case, M, N, K, batch, a_m, b_n, epi, worst), flush=True)
    ctx.sync()
    print("all %d cases passed; worst error / tolerance %.3f" % (cases, worst))


if __name__ == "__main__":
    main()
