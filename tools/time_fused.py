"""Ablation timing of blr_fused_update_kernel: slab vs stats input, with/without next draw."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bayesic_amd._ffi import ptr
from bayesic_amd.device import Context

ctx = Context(0)
dev = ctx.device
B, D, S = 1_000_000, 256, 8
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((B, D), generator=g, device=dev); y = torch.randn(B, generator=g, device=dev)
f64 = torch.float64
lam = torch.zeros((2, 2*D+2), dtype=f64, device=dev); lam[0, D:2*D] = -2.3; lam[0, 2*D+1] = -2.3
m1 = torch.zeros(2*D+2, dtype=f64, device=dev); m2 = torch.zeros_like(m1)
eps = torch.zeros((2, S*(D+1)), dtype=f64, device=dev); W = torch.zeros((2, S*D), device=dev); xi = torch.zeros((2, S), dtype=f64, device=dev)
elbo = torch.zeros(1, dtype=f64, device=dev); grad = torch.zeros(2*D+2, dtype=f64, device=dev)
stats = torch.zeros(S*(D+1), dtype=f64, device=dev)
ctx.call("bsc_blr_sample", ptr(lam[0]), D, S, 1, 0, ptr(eps[0]), ptr(W[0]), ptr(xi[0]))
ctx.call("bsc_blr_data_pass", ptr(X), D, ptr(y), B, D, ptr(W[0]), S, ptr(stats[:S]), ptr(stats[S:]))

def fused(use_slab, draw, ready=0):
    ctx.call("bsc_blr_fused_update", None if use_slab else ptr(stats), ptr(lam[0]), ptr(lam[1]), ptr(m1), ptr(m2),
             ptr(eps[0]), ptr(W[0]), ptr(xi[0]), D, S, float(B), 1.0, 1.0, 1.0, 1, 1e-3, 0.9, 0.999, 1e-8, 1, 1,
             ptr(eps[1]) if draw else None, ready, ptr(W[1]) if draw else None, ptr(xi[1]) if draw else None, ptr(elbo), ptr(grad))

def timeit(fn, n=50):
    for _ in range(5): fn()
    ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n): fn()
    e1.record()
    return e0.elapsed_ms(e1) / n * 1e3

def partial():
    ctx.call("bsc_blr_data_pass_partial", ptr(X), D, ptr(y), B, D, ptr(W[0]), S)

partial(); ctx.sync()
print("fused(stats, draw)   %.1f us" % timeit(lambda: fused(False, True)))
print("fused(stats, nodraw) %.1f us" % timeit(lambda: fused(False, False)))
print("fused(slab,  draw)   %.1f us  (slab L2/MALL-warm: re-read back to back)" % timeit(lambda: fused(True, True)))
print("fused(slab,  ready noise) %.1f us" % timeit(lambda: fused(True, True, 1)))
print("fused(slab,  nodraw) %.1f us" % timeit(lambda: fused(True, False)))
tp = timeit(partial, 20)
def both(): partial(); fused(True, True, 1)
tb = timeit(both, 20)
print("pass alone %.1f us; pass+fused %.1f us -> fused adds %.1f us" % (tp, tb, tb - tp))
ctx.call("bsc_blr_sample", ptr(lam[0]), D, S, 1, 0, ptr(eps[0]), ptr(W[0]), ptr(xi[0]))
print("sample kernel %.1f us" % timeit(lambda: ctx.call("bsc_blr_sample", ptr(lam[0]), D, S, 1, 0, ptr(eps[0]), ptr(W[0]), ptr(xi[0]))))
