"""How long is the boundary between two launches of blr_pass_q_kernel on one stream?  Two contexts on the SAME stream
(each keeps the s_memrealtime stamps of its own last launch, option blr_stamps) launch alternately; the gap is
(first workgroup start of launch k + 1) - (last workgroup end of launch k) on the device's 100 MHz clock.  Also the
pass -> finish -> pass sequence of an update (the finish kernel between two stamped passes).

    python tools/launch_boundary.py [rows]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bayesic_amd._ffi import ptr
from bayesic_amd.device import Context


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    D, S = 256, 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    total = 3_000_000
    X = torch.randn((total, D), generator=g, device=dev)
    y = torch.randn(total, generator=g, device=dev)
    W = torch.randn((S, D), generator=g, device=dev) / 16
    stream = torch.cuda.current_stream(dev)
    ctxs = [Context(0, stream=stream, options=dict(blr_stamps=1)) for _ in range(2)]
    for c in ctxs:
        c.reserve(32 << 20)
    nb = total // B
    gaps, spans, periods = [], [], []
    prev_end = None
    prev_start = None
    for rep in range(12):
        # a train of launches, alternating contexts; only the LAST launch of each context keeps its stamps
        n = 8 + 2 * rep % 4
        for i in range(n):
            r0 = ((rep * 16 + i) % nb) * B
            ctxs[i % 2].call("bsc_blr_data_pass_partial", ptr(X[r0:]), D, ptr(y[r0:]), B, D, ptr(W), S)
        torch.cuda.synchronize()
        last = ctxs[(n - 1) % 2].read_stamps()
        before = ctxs[(n - 2) % 2].read_stamps()
        if rep >= 2:
            gaps.append((int(last[:, 0].min()) - int(before[:, 1].max())) / 100.0)
            spans.append((int(last[:, 1].max()) - int(last[:, 0].min())) / 100.0)
            periods.append((int(last[:, 0].min()) - int(before[:, 0].min())) / 100.0)
    print("rows %d: in-kernel span (first start -> last end) median %.1f us; boundary (last end of launch k -> first start of "
          "launch k + 1) median %.2f us (min %.2f max %.2f); start-to-start period median %.1f us"
          % (B, np.median(spans), np.median(gaps), min(gaps), max(gaps), np.median(periods)))


if __name__ == "__main__":
    main()
