"""Is the imbalance of blr_pass_q_kernel's static partition SYSTEMATIC (the same workgroups / CUs late in every launch) or
random?  Stamps (option blr_stamps) of several launches over different resident mini-batches: correlation of the
per-workgroup end times between launches, and the mean end per XCD, per shader engine and per CU position.

    python tools/stamps_structure.py [launches] [rows]
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    D, S = 256, 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    total = 3_000_000
    X = torch.randn((total, D), generator=g, device=dev)
    y = torch.randn(total, generator=g, device=dev)
    W = torch.randn((S, D), generator=g, device=dev) / 16
    c = Context(0, options=dict(blr_stamps=1))
    c.reserve(32 << 20)
    nb = total // B
    ends, durs, keys = [], [], None
    for i in range(n + 4):
        r0 = (i % nb) * B
        c.call("bsc_blr_data_pass_partial", ptr(X[r0:]), D, ptr(y[r0:]), B, D, ptr(W), S)
        st = c.read_stamps()
        if i < 4:
            continue
        t0 = st[:, 0].min()
        ends.append((st[:, 1] - t0) / 100.0)
        durs.append((st[:, 1] - st[:, 0]) / 100.0)
        k = np.stack([(st[:, 2] & 7), (st[:, 3] >> 13) & 7, (st[:, 3] >> 12) & 1, (st[:, 3] >> 8) & 15], 1).astype(int)
        if keys is None:
            keys = k
        else:
            print("launch %d: %d of %d workgroups on the same (xcd, se, sh, cu) as in the first launch" % (i, int((k == keys).all(1).sum()), len(k)))
    E = np.array(ends)
    print("rows %d, %d workgroups, %d launches; kernel span (max end) per launch: %s" % (B, E.shape[1], n, np.round(E.max(1), 1)))
    cc = np.corrcoef(E)
    off = cc[~np.eye(n, dtype=bool)]
    print("correlation of per-workgroup end times between launches: mean %.3f min %.3f max %.3f" % (off.mean(), off.min(), off.max()))
    m = E.mean(0)
    print("per-workgroup mean end: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f; residual (launch - mean) std %.2f us"
          % (m.min(), np.percentile(m, 10), np.median(m), np.percentile(m, 90), m.max(), (E - m).std()))
    par = np.arange(E.shape[1]) & 1
    print("mean end by blockIdx parity: even %.1f odd %.1f" % (m[par == 0].mean(), m[par == 1].mean()))
    for name, col in (("xcd", 0), ("se", 1), ("sh", 2), ("cu", 3)):
        vals = sorted(set(keys[:, col]))
        print("mean end by %s: " % name + " ".join("[%d] %.1f" % (v, m[keys[:, col] == v].mean()) for v in vals))
    # per (xcd, se) table
    print("mean end by (xcd, se):")
    for x in range(8):
        print("  xcd %d: " % x + " ".join("%.1f" % m[(keys[:, 0] == x) & (keys[:, 1] == s)].mean() if ((keys[:, 0] == x) & (keys[:, 1] == s)).any() else "  -  " for s in range(8)))
    # first and second workgroup on a CU (blockIdx < 256 and >= 256)
    h = E.shape[1] // 2
    print("mean end of workgroups [0, %d): %.1f; [%d, %d): %.1f" % (h, m[:h].mean(), h, 2 * h, m[h:].mean()))
    np.save(os.path.join(ROOT, "gpurun_out", "stamps_structure.npy"), np.concatenate([E, keys.T.astype(np.float64)]))


if __name__ == "__main__":
    main()
