"""In-process A/B of the executor with and without element-wise fusion
(DeviceBackend(fuse=True/False)) on data-sized expressions.  Prints one JSON line
per case: microseconds per evaluation of the compiled device function, operands
resident in HBM."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bayesic_amd.device import Context
from bayesic_amd.algebra import var, exp, log, dot, sum as asum
from bayesic_amd.algebra.device_backend import DeviceBackend

quick = "--quick" in sys.argv
fused_only = "--fused-only" in sys.argv
no_lda = "--no-lda" in sys.argv
map_only = "--map-only" in sys.argv
ctx = Context(0)
dev = ctx.device
g = torch.Generator(device=dev).manual_seed(3)


def timeit(fn, n, warm_ms=60.0):
    """Mean time of n calls after at least warm_ms of the same call back to back (a kernel
    reaches its steady rate only after ~35 ms of continuous running, tools/ramp_probe.py)."""
    e0, e1 = ctx.event(), ctx.event()
    elapsed = 0.0
    while elapsed < warm_ms:
        e0.record()
        fn()
        fn()
        e1.record()
        elapsed += e0.elapsed_ms(e1)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    return e0.elapsed_ms(e1) / n * 1e3


only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]


def case(name, expr, inputs, algo_bytes, reps):
    if only and not any(o in name for o in only):
        return
    out = {"case": name, "algorithmic_bytes": algo_bytes}
    st0 = torch.cuda.memory_stats()
    for fuse in ((True,) if fused_only else (True, False)):
        f = expr.compile(DeviceBackend(ctx, fuse=fuse)).device_fn
        us = timeit(lambda: f(**inputs), reps)
        out["fused_us" if fuse else "unfused_us"] = us
    if not fused_only:
        out["speedup"] = out["unfused_us"] / out["fused_us"]
    out["fused_GBps_algorithmic"] = algo_bytes / out["fused_us"] / 1e3
    st1 = torch.cuda.memory_stats()
    out["device_mallocs"] = st1.get("num_device_alloc", 0) - st0.get("num_device_alloc", 0)
    out["device_frees"] = st1.get("num_device_free", 0) - st0.get("num_device_free", 0)
    out["reserved_GB"] = st1.get("reserved_bytes.all.current", 0) / 1e9
    print(json.dumps(out), flush=True)


X, Y = var("X", ndim=2), var("Y", ndim=2)
rows = 200_000 if quick else 1_000_000
Xd = torch.randn((rows, 256), generator=g, device=dev)
Yd = torch.randn((rows, 256), generator=g, device=dev)
n = rows * 256
if map_only:
    case("X / Y %dx256" % rows, X / Y, dict(X=Xd, Y=Yd), 12 * n, 20)
    case("exp(X) %dx256" % rows, exp(X), dict(X=Xd), 8 * n, 20)
    sys.exit(0)
case("sum(abs(X)) %dx256" % rows, asum(abs(X)), dict(X=Xd), 4 * n, 20)
case("sum(abs(X), axis=0) %dx256" % rows, asum(abs(X), axis=0), dict(X=Xd), 4 * n, 20)
case("sum(X*Y) %dx256" % rows, asum(X * Y), dict(X=Xd, Y=Yd), 8 * n, 20)
case("sum(X*Y, axis=0) %dx256" % rows, asum(X * Y, axis=0), dict(X=Xd, Y=Yd), 8 * n, 20)
case("sum(X*Y, axis=1) %dx256" % rows, asum(X * Y, axis=1), dict(X=Xd, Y=Yd), 8 * n, 20)
case("sum(exp(X)*Y) %dx256" % rows, asum(exp(X) * Y), dict(X=Xd, Y=Yd), 8 * n, 20)
case("sum(exp(X)*Y, axis=0) %dx256" % rows, asum(exp(X) * Y, axis=0), dict(X=Xd, Y=Yd), 8 * n, 20)
case("X / Y %dx256" % rows, X / Y, dict(X=Xd, Y=Yd), 12 * n, 20)
case("log(exp(X) + exp(Y)) %dx256" % rows, log(exp(X) + exp(Y)), dict(X=Xd, Y=Yd), 12 * n, 20)
vrow = var("v", ndim=1)
ucol = var("u", ndim=1)
vd = torch.randn(256, generator=g, device=dev)
ud = torch.randn(rows, generator=g, device=dev)
from bayesic_amd.algebra import dimshuffle
case("X * v[None,:] %dx256" % rows, X * dimshuffle(vrow, "x", 0), dict(X=Xd, v=vd), 8 * n, 20)
case("X + u[:,None] %dx256" % rows, X + dimshuffle(ucol, 0, "x"), dict(X=Xd, u=ud), 8 * n, 20)
case("sum(X * v[None,:], axis=1) %dx256" % rows, asum(X * dimshuffle(vrow, "x", 0), axis=1),
     dict(X=Xd, v=vd), 4 * n, 20)
case("sum(exp(X) * u[:,None], axis=0) %dx256" % rows, asum(exp(X) * dimshuffle(ucol, 0, "x"), axis=0),
     dict(X=Xd, u=ud), 4 * n, 20)
# an element-wise producer of a product's operand, formed inside the product (bsc_gemm_fused)
W = var("W", ndim=2)
Wd = torch.randn((256, 256), generator=g, device=dev) / 16
Yn = torch.randn((rows, 128), generator=g, device=dev)
Xs = (Xd * 0.25).contiguous()
case("dot(X * X, W) %dx256 . 256x256" % rows, dot(X * X, W), dict(X=Xd, W=Wd), 8 * n, 10)
case("dot(exp(X), W) %dx256 . 256x256" % rows, dot(exp(X), W), dict(X=Xs, W=Wd), 8 * n, 10)
case("dot((X * X).T, Y) 256x%d . %dx128" % (rows, rows), dot((X * X).T, Y), dict(X=Xd, Y=Yn), 4 * n + 4 * rows * 128, 10)
del Xd, Yd, Xs, Yn

if no_lda:
    sys.exit(0)
Th, Bm, C = var("Th", ndim=2), var("Bm", ndim=2), var("C", ndim=2)
docs, V, K = (1000, 100_000, 128) if quick else (6250, 100_000, 128)
Thd = torch.rand((docs, K), generator=g, device=dev) + 0.1
Bmd = torch.rand((K, V), generator=g, device=dev) + 0.1
Cd = torch.poisson(torch.full((docs, V), 0.05, device=dev), generator=g)
case("LDA statistic Bm*dot(Th.T, C/dot(Th,Bm)) %dx%d K=%d" % (docs, V, K),
     Bm * dot(Th.T, C / dot(Th, Bm)), dict(Th=Thd, Bm=Bmd, C=Cd), 4 * docs * V, 5)
