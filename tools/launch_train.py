"""blr_pass_q_kernel launch to launch: a train of 120 launches between TWO events against one event pair PER launch (what
bench.py's roofline burst and tools/ab_q.py use) -- does the per-launch timing itself cost time?

    python tools/launch_train.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd._ffi import ptr
from bayesic_amd.device import Context
D, S = 256, 8
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
total = 3_000_000
X = torch.randn((total, D), generator=g, device=dev)
y = torch.randn(total, generator=g, device=dev)
W = torch.randn((S, D), generator=g, device=dev) / 16
c = Context(0)
c.reserve(32 << 20)
for B in (1_000_000, 500_000, 250_000, 125_000):
    nb = total // B
    k = [0]
    def launch():
        k[0] = (k[0] + 1) % nb
        r0 = k[0] * B
        c.call("bsc_blr_data_pass_partial", ptr(X[r0:]), D, ptr(y[r0:]), B, D, ptr(W), S)
    for _ in range(20): launch()
    c.sync()
    res = []
    for rep in range(5):
        e0, e1 = c.event(), c.event()
        n = 120
        e0.record()
        for _ in range(n): launch()
        e1.record()
        res.append(e0.elapsed_ms(e1) / n * 1e3)
    per = []
    for rep in range(5):
        c.profile(True)
        for _ in range(48): launch()
        ms, cnt = c.profile_read()
        c.profile(False)
        per.append(ms / cnt * 1e3)
    print("rows %8d: train of 120 launches between two events: %.2f us per launch (min %.2f); one event pair per launch: %.2f us (min %.2f)"
          % (B, np.median(res), min(res), np.median(per), min(per)))
