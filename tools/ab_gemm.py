"""Interleaved in-process A/B of GEMM variants (contexts created under different env knobs),
each burst preceded by its own steady-state warm-up; whole-call time from hipEvent pairs (a
split-K product counts its reduce), and the two variants' results compared.

    python tools/ab_gemm.py [KNOB [VALUE_A VALUE_B]] [--quick]
"""
import os, sys
os.environ.setdefault("BSC_PROFILING_BUILDS", "1")   # the Context honours BSC_<OPTION> variables only in a process that opts in (device.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bayesic_amd.device import Context

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
knob = pos[0] if pos else "BSC_GEMM_DMA"
quick = "--quick" in sys.argv
va, vb = (pos[1], pos[2]) if len(pos) > 2 else ("0", "1")
ctxs = {}
for v in (va, vb):
    os.environ[knob] = v
    ctxs[v] = Context(0)
    ctxs[v].reserve(2400 << 20)
os.environ.pop(knob)
dev = ctxs[va].device
g = torch.Generator(device=dev).manual_seed(1)
cases = []


def square(n):
    A = torch.randn((n, n), generator=g, device=dev); B = torch.randn((n, n), generator=g, device=dev)
    C = torch.empty((n, n), device=dev)
    # (batch, M, N, K, A, sa_b, sa_m, sa_k, B, sb_b, sb_k, sb_n, C, sc_b, sc_m, sc_n)
    cases.append(("%d^3 TN (A m-, B n-contig)" % n, "gemm", (1, n, n, n, A, 0, 1, n, B, 0, n, 1, C, 0, n, 1), 2.0 * n ** 3, C))
    cases.append(("%d^3 NN (A k-, B n-contig)" % n, "gemm", (1, n, n, n, A, 0, n, 1, B, 0, n, 1, C, 0, n, 1), 2.0 * n ** 3, C))
    cases.append(("%d^3 NT (A k-, B k-contig)" % n, "gemm", (1, n, n, n, A, 0, n, 1, B, 0, 1, n, C, 0, n, 1), 2.0 * n ** 3, C))
    cases.append(("%d^3 TT (A m-, B k-contig)" % n, "gemm", (1, n, n, n, A, 0, 1, n, B, 0, 1, n, C, 0, n, 1), 2.0 * n ** 3, C))


for n in ((4096,) if quick else (4096, 4224, 8192)):
    square(n)
X = torch.randn((1_000_000, 256), generator=g, device=dev); G = torch.empty((256, 256), device=dev)
cases.append(("gram 256x256x1M", "gemm", (1, 256, 256, 1_000_000, X, 0, 1, 256, X, 0, 256, 1, G, 0, 256, 1), 2.0 * 256 * 256 * 1e6, G))
# config 4's statistic through the executor: Q = C / dot(Th, Bt) [6250 x 100k x 128], S = Bt * dot(Th.T, Q)
docs, V, K = 6250, 100_000, 128
Th = torch.rand((docs, K), generator=g, device=dev) + 0.1
Bt = torch.rand((K, V), generator=g, device=dev) + 0.1
Cn = torch.rand((docs, V), generator=g, device=dev)
Q = torch.empty((docs, V), device=dev); S = torch.empty((K, V), device=dev)
cases.append(("lda Q = C / dot(Th, Bt)", "epi", (1, docs, V, K, Th, 0, K, 1, Bt, 0, V, 1, Q, 0, V, 1, -1, 1.0, Cn, 0, V, 1), 2.0 * docs * V * K, Q))
cases.append(("lda dot(Th, Bt) alone", "gemm", (1, docs, V, K, Th, 0, K, 1, Bt, 0, V, 1, Q, 0, V, 1), 2.0 * docs * V * K, Q))
cases.append(("lda S = Bt * dot(Th.T, Q)", "epi", (1, K, V, docs, Th, 0, 1, K, Cn, 0, V, 1, S, 0, V, 1, 1, 1.0, Bt, 0, V, 1), 2.0 * docs * V * K, S))

for name, kind, args, flops, out in cases:
    res = {va: [], vb: []}
    outs = {}
    fn = "bsc_gemm_strided_batched" if kind == "gemm" else "bsc_gemm_epilogue"
    for rnd in range(4):
        for v, c in ctxs.items():
            f = lambda: c.call(fn, 0, *args)
            e0, e1 = c.event(), c.event()
            t = 0.0
            while t < 40.0:                      # steady-state warm-up of this very call
                e0.record(); f(); f(); e1.record(); t += e0.elapsed_ms(e1)
            e0.record()
            for _ in range(6): f()
            e1.record()
            res[v].append(e0.elapsed_ms(e1) / 6 * 1e3)
            if rnd == 0:
                outs[v] = out.double().cpu()
    a, b = np.median(res[va]), np.median(res[vb])
    diff = ((outs[va] - outs[vb]).abs().max() / outs[va].abs().max()).item()
    print("%-30s %s=%s: %8.1f us (%5.1f TF)   =%s: %8.1f us (%5.1f TF)   x%.3f   max|d|/max|.| %.1e" %
          (name, knob, va, a, flops / a / 1e6, vb, b, flops / b / 1e6, a / b, diff), flush=True)
