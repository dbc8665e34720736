"""How long after a process starts using the GPU does the data pass reach its steady
rate?  Prints the mean pass-kernel time of successive batches of 25 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesic_amd.device import Context
ctx = Context(0); dev = ctx.device
g = torch.Generator(device=dev).manual_seed(1)
N, D, S = 1_000_000, 256, 8
X = torch.randn((N, D), generator=g, device=dev); y = torch.randn(N, generator=g, device=dev)
W = torch.randn((S, D), generator=g, device=dev) / 16
ctx.reserve(16 << 20)
ctx.sync()
out = []
for b in range(24):
    ctx.profile(True)
    for _ in range(25):
        ctx.call("bsc_blr_data_pass_partial", X, D, y, N, D, W, S)
    ms, n = ctx.profile_read(); ctx.profile(0)
    out.append(ms / n * 1e3)
print("pass us per batch of 25 launches (~4.3 ms each):", " ".join("%.0f" % v for v in out))
