"""Deletion builds of gram256_bx_kernel (X^T X at 1M x 256 as two bf16 terms): where its 257 us go.
BSC_GRAM_DBG bits: 1 no conversion, 2 no MFMAs, 4 no DMAs after the prologue's (results WRONG; BSC_PROFILING_BUILDS=1).
The kernel's own time through the context's profiling slot (the two reduce launches of a call excluded).

    python tools/ab_gram.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BSC_PROFILING_BUILDS"] = "1"

import torch

from bayesic_amd.device import Context


def main():
    N, D = 1_000_000, 256
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    Xs = [torch.randn((N, D), generator=g, device=dev) for _ in range(3)]
    C = torch.zeros((D, D), device=dev)
    names = {0: "as shipped", 1: "no conversion", 2: "no MFMAs", 3: "no conversion, no MFMAs", 4: "no DMAs",
             5: "no DMAs, no conversion", 6: "no DMAs, no MFMAs", 7: "fragment reads and barriers only"}
    for dbg in range(8):
        ctx = Context(0, options=dict(profiling_builds=1, gram_dbg=dbg))
        ctx.call("bsc_ctx_set_mfma_split", 2)
        k = [0]

        def run():
            X = Xs[k[0] % 3]
            k[0] += 1
            ctx.call("bsc_gemm_strided_batched", 0, 1, D, D, N, X, 0, 1, D, X, 0, D, 1, C, 0, D, 1)
        for _ in range(5):
            run()
        ctx.sync()
        best = []
        for _ in range(3):
            e0, e1 = ctx.event(), ctx.event()
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            best.append(e0.elapsed_ms(e1) / 20 * 1e3)
        print("dbg %d  %-36s %7.1f us per call (kernel + two reduce launches)" % (dbg, names[dbg], min(best)), flush=True)


if __name__ == "__main__":
    main()
