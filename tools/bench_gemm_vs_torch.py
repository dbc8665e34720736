"""bsc_gemm_strided_batched (hand-written f32-MFMA kernels) beside torch.matmul (rocBLAS / hipBLASLt f32) on the same
box and operands -- a yardstick only; the library is not used by the product.   python tools/bench_gemm_vs_torch.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesic_amd.device import Context     # noqa: E402


def main():
    ctx = Context(0)
    dev = ctx.device
    g = torch.Generator(device=dev).manual_seed(0)
    torch.backends.cuda.matmul.allow_tf32 = False
    cases = [("4096^3 (NN)", 4096, 4096, 4096, False, False), ("8192^3 (NN)", 8192, 8192, 8192, False, False),
             ("4096^3 (TN)", 4096, 4096, 4096, True, False), ("X^T X 256 x 1M x 256", 256, 256, 1_000_000, True, False),
             ("1M x 256 . 256 x 256", 1_000_000, 256, 256, False, False),
             ("6250 x 128 . 128 x 100000", 6250, 100_000, 128, False, False),
             ("128 x 6250 . 6250 x 100000 (TN)", 128, 100_000, 6250, True, False)]
    for name, M, N, K, ta, tb in cases:
        A = torch.randn((K, M) if ta else (M, K), generator=g, device=dev)
        B = A if name.startswith("X^T X") else torch.randn((K, N), generator=g, device=dev)
        Av = A.T if ta else A
        C = torch.empty((M, N), device=dev)
        def ours():
            ctx.call("bsc_gemm_strided_batched", 0, 1, M, N, K, Av, 0, Av.stride(0), Av.stride(1), B, 0, B.stride(0), B.stride(1),
                     C, 0, N, 1)
        def lib():
            torch.matmul(Av, B, out=C)
        res = {}
        for label, fn in (("ours", ours), ("torch", lib)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); ctx.sync()
            if label == "torch":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record(); torch.cuda.synchronize()
                res[label] = e0.elapsed_time(e1) / 10
            else:
                a, b = ctx.event(), ctx.event()
                a.record()
                for _ in range(10):
                    fn()
                b.record()
                res[label] = a.elapsed_ms(b) / 10
        fl = 2.0 * M * N * K
        print("%-34s ours %8.3f ms = %6.1f TF   torch.matmul %8.3f ms = %6.1f TF" % (name, res["ours"], fl / res["ours"] * 1e-9, res["torch"], fl / res["torch"] * 1e-9))


if __name__ == "__main__":
    main()
