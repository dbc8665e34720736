import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesic_amd.device import Context
from bayesic_amd.svi import mog as mog_mod
ctx = Context(0)
n, D, K = 2_000_000, 16, 128
rs = np.random.RandomState(1)
X = (rs.standard_normal((n, D)) * 3).astype(np.float32)
eta0 = mog_mod.prior_eta(K, D); eta = mog_mod.init_eta(X[:4000], K, D, seed=2)
m = mog_mod.MoGNatGradSVI(ctx.to_device(X), K, eta0, eta, n_total=float(n), ctx=ctx, via="executor")
for _ in range(3): m.step()
ctx.sync(); t0 = time.perf_counter()
for _ in range(10): m.step()
ctx.sync(); print("MoG executor route K=%d D=%d rows=%d: %.3f ms per update" % (K, D, n, (time.perf_counter() - t0) / 10 * 1e3))
