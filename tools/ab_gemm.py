"""Interleaved in-process A/B of GEMM variants (contexts created under different env knobs),
each burst preceded by its own steady-state warm-up."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bayesic_amd.device import Context

knob = sys.argv[1] if len(sys.argv) > 1 else "BSC_GEMM_PIPE"
ctxs = {}
for v in ("0", "1"):
    os.environ[knob] = v
    ctxs[v] = Context(0)
    ctxs[v].reserve(600 << 20)
dev = ctxs["0"].device
g = torch.Generator(device=dev).manual_seed(1)
cases = []
for n in (4096, 8192):
    A = torch.randn((n, n), generator=g, device=dev); B = torch.randn((n, n), generator=g, device=dev)
    C = torch.empty((n, n), device=dev)
    cases.append(("%d^3 A m-contig" % n, (1, n, n, n, A, 0, 1, n, B, 0, n, 1, C, 0, n, 1), 2.0 * n ** 3, (A, B, C)))
    cases.append(("%d^3 A k-contig" % n, (1, n, n, n, A, 0, n, 1, B, 0, n, 1, C, 0, n, 1), 2.0 * n ** 3, (A, B, C)))
X = torch.randn((1_000_000, 256), generator=g, device=dev); G = torch.empty((256, 256), device=dev)
cases.append(("gram 256x256x1M", (1, 256, 256, 1_000_000, X, 0, 1, 256, X, 0, 256, 1, G, 0, 256, 1), 2.0 * 256 * 256 * 1e6, (X, G)))
for name, args, flops, keep in cases:
    res = {"0": [], "1": []}
    for rnd in range(4):
        for v, c in ctxs.items():
            f = lambda: c.call("bsc_gemm_strided_batched", 0, *args)
            e0, e1 = c.event(), c.event()
            t = 0.0
            while t < 40.0:                      # steady-state warm-up of this very call
                e0.record(); f(); f(); e1.record(); t += e0.elapsed_ms(e1)
            c.profile(True)
            for _ in range(6): f()
            ms, n = c.profile_read(); c.profile(0)
            res[v].append(ms / n * 1e3)
    a, b = np.median(res["0"]), np.median(res["1"])
    print("%-22s %s=0: %8.1f us (%.1f TF)   %s=1: %8.1f us (%.1f TF)   x%.3f" %
          (name, knob, a, flops / a / 1e6, knob, b, flops / b / 1e6, a / b), flush=True)
