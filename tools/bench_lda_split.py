"""Config 4's statistic kernel (one GPU's shard: 6250 x 100 000 counts, K = 128) on the f32 MFMA and on the
operand-split bf16 route (bsc_ctx_set_mfma_split 2 / 3), with the error of each against a float64 reference on a
sample of columns.    python tools/bench_lda_split.py [docs V]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd.device import Context     # noqa: E402


def main():
    docs, V, K = (int(sys.argv[1]), int(sys.argv[2]), 128) if len(sys.argv) > 2 else (6250, 100_000, 128)
    ctx = Context(0)
    g = torch.Generator(device=ctx.device).manual_seed(24)
    C = torch.poisson(torch.full((docs, V), 0.05, device=ctx.device), generator=g)
    Th = torch.rand((docs, K), generator=g, device=ctx.device) + 0.1
    Bt = torch.rand((K, V), generator=g, device=ctx.device) + 0.1
    cols = torch.arange(0, V, max(1, V // 512), device=ctx.device)[:512]
    C64, Th64, Bt64 = C[:, cols].double(), Th.double(), Bt[:, cols].double()
    ref = (Bt64 * (Th64.T @ (C64 / (Th64 @ Bt64)))).cpu().numpy()
    ref_ll = None
    out = ctx.zeros((K, V), torch.float32)
    ll = ctx.zeros(1, torch.float64)
    flops = 4.0 * docs * V * K
    for terms in (0, 2, 3):
        ctx.call("bsc_ctx_set_mfma_split", terms)
        for bound in (False, True):
            def run():
                if bound:
                    ctx.call("bsc_lda_sstats_bound", C, V, docs, V, K, Th, K, Bt, V, out, V, ll)
                else:
                    ctx.call("bsc_lda_sstats", C, V, docs, V, K, Th, K, Bt, V, out, V)
            for _ in range(3):
                run()
            ctx.sync()
            e0, e1 = ctx.event(), ctx.event()
            n = 20
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            ms = e0.elapsed_ms(e1) / n
            got = out[:, cols].cpu().numpy()
            err = np.abs(got - ref) / np.abs(ref)
            if bound and ref_ll is None:
                ref_ll = ll.item()
            print("terms %d%s: %.3f ms per call (split kernels included)  %.1f TF f32-equivalent  %.2f TB/s of C;  "
                  "rel error vs float64: max %.2e, rms %.2e%s"
                  % (terms, " + bound" if bound else "        ", ms, flops / ms * 1e-9, docs * V * 4 / ms * 1e-9,
                     err.max(), np.sqrt((err ** 2).mean()),
                     ";  bound %.10e (rel. to f32 route %.1e)" % (ll.item(), abs(ll.item() - ref_ll) / abs(ref_ll)) if bound else ""))
    ctx.call("bsc_ctx_set_mfma_split", 0)


if __name__ == "__main__":
    main()
