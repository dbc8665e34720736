"""gram256_bx_kernel: the one-barrier loop of round 3 (option gram_pp = 0) against the ping-pong loop (gram_pp = 1), X^T X at
1M x 256 as two bf16 terms, three X in rotation, interleaved rounds in one process; per CALL (kernel + two reduce
launches) and the kernel alone (the context's profiling slot).

    python tools/ab_gram_pp.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd.device import Context


def main():
    N, D = 1_000_000, 256
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    Xs = [torch.randn((N, D), generator=g, device=dev) for _ in range(3)]
    C = {pp: torch.zeros((D, D), device=dev) for pp in (0, 1)}
    ctxs = {pp: Context(0, options=dict(gram_pp=pp)) for pp in (0, 1)}
    for c in ctxs.values():
        c.call("bsc_ctx_set_mfma_split", 2)
    k = [0]

    def run(pp):
        X = Xs[k[0] % 3]
        k[0] += 1
        ctxs[pp].call("bsc_gemm_strided_batched", 0, 1, D, D, N, X, 0, 1, D, X, 0, D, 1, C[pp], 0, D, 1)

    for pp in (0, 1):
        for _ in range(6):
            run(pp)
        ctxs[pp].sync()
    # same X: same result?
    for pp in (0, 1):
        k[0] = 0
        run(pp)
    torch.cuda.synchronize()
    ref = (Xs[0].double().T @ Xs[0].double())
    for pp in (0, 1):
        err = ((C[pp].double() - ref).abs() / (Xs[0].double().pow(2).sum(0).sqrt()[:, None] * Xs[0].double().pow(2).sum(0).sqrt()[None, :])).max().item()
        print("gram_pp=%d: max |C - float64| / (|x_d| |x_e|) = %.2e; symmetric to the bit: %s" % (pp, err, bool((C[pp] == C[pp].T).all())))
    print("the two loops give the same bits:", bool((C[0] == C[1]).all()))
    per_call = {0: [], 1: []}
    kern = {0: [], 1: []}
    for r in range(6):
        for pp in (0, 1):
            c = ctxs[pp]
            e0, e1 = c.event(), c.event()
            e0.record()
            for _ in range(20):
                run(pp)
            e1.record()
            per_call[pp].append(e0.elapsed_ms(e1) / 20 * 1e3)
            c.profile(True)
            for _ in range(20):
                run(pp)
            ms, cnt = c.profile_read()
            c.profile(False)
            kern[pp].append(ms / cnt * 1e3)
    for pp in (0, 1):
        a, b = np.array(per_call[pp]), np.array(kern[pp])
        print("gram_pp=%d: per call median %.1f us (min %.1f max %.1f); kernel alone median %.1f us (min %.1f max %.1f) = %.2f TB/s of X"
              % (pp, np.median(a), a.min(), a.max(), np.median(b), b.min(), b.max(), 4.0 * N * D / np.median(b) / 1e6))


if __name__ == "__main__":
    main()
