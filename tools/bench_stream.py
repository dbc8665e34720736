"""PCIe-inclusive rate of the config-2 update when every mini-batch is streamed from
host memory (bsc_loader_*): batch t+1 crosses PCIe while update t runs.  Prints one
JSON line.  This is NOT bench.py's `value` (which keeps the batch resident); it is
the number DESIGN.md section 7 quotes beside it."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bayesic_amd.device import Context
from bayesic_amd.svi.blr import BLRReparamSVI
from bayesic_amd.svi.stream import MiniBatchLoader

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D, S, n_host, steps = 256, 8, 4, 24
ctx = Context(0)
rs = np.random.RandomState(0)
host = []
for i in range(n_host):                  # pinned allocations (torch is plumbing here)
    X = torch.empty((rows, D), dtype=torch.float32, pin_memory=True)
    y = torch.empty(rows, dtype=torch.float32, pin_memory=True)
    X.normal_()
    y.normal_()
    host.append((X, y))
X0 = host[0][0].to(ctx.device)
y0 = host[0][1].to(ctx.device)
model = BLRReparamSVI(X0, y0, n_total=float(rows * n_host), n_samples=S, seed=1, lr=1e-3, ctx=ctx)
loader = MiniBatchLoader(ctx, rows, D, n_slots=2)


def run(n):
    loader.submit(*host[0])
    for t in range(n):
        if t + 1 < n:
            loader.submit(*host[(t + 1) % n_host])
        model.set_batch(*loader.acquire())
        model.step()
        loader.release()
    ctx.sync()


run(4)
t0 = time.perf_counter()
run(steps)
dt = time.perf_counter() - t0
model.set_batch(X0, y0)
for _ in range(5):
    model.step()
ctx.sync()
t1 = time.perf_counter()
for _ in range(50):
    model.step()
ctx.sync()
resident = 50 / (time.perf_counter() - t1)
bytes_per = 4.0 * rows * D + 4.0 * rows
print(json.dumps({"workload": "cfg2 update, %dx%d f32 mini-batch streamed from pinned host memory" % (rows, D),
                  "streamed_updates_per_s": steps / dt, "h2d_GBps": steps * bytes_per / dt / 1e9,
                  "resident_updates_per_s": resident, "steps": steps,
                  "note": "two HBM slots, copy stream overlapped with the update stream"}))
loader.close()
