"""Timeline of the folded finish (option blr_fold) from in-kernel stamps: when the last partial was written, when its
ticket came back, when the role workgroups saw every row, when the launch ended.   python tools/fold_timeline.py [rows]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bayesic_amd.device import Context
from bayesic_amd.svi.blr import BLRReparamSVI
from oracle import svi


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
    ctx = Context(0, options=dict(blr_stamps=1))
    X, y, _ = svi.make_cfg2(B, 256)
    m = BLRReparamSVI(ctx.to_device(X), ctx.to_device(y), n_total=10.0 * B, n_samples=8, seed=1, lr=0.01, ctx=ctx)
    for _ in range(30):
        m.step()
    for rep in range(3):
        m.step()
        st = ctx.read_stamps()
        t0 = st[:, 0].min()
        us = lambda v: (v.astype(np.float64) - float(t0)) / 100.0
        end, wrote, ticketed, seen, ticket = us(st[:, 1]), us(st[:, 4]), us(st[:, 5]), us(st[:, 6]), st[:, 7].astype(int)
        n = len(st)
        roles = ticket >= n - 33
        last = int(np.argmax(ticket))
        print("rows %d, %d workgroups: last partial written %.2f us, its ticket back %.2f; role workgroups saw every row at "
              "%.2f .. %.2f; roles ended %.2f .. %.2f; non-role workgroups ended by %.2f"
              % (B, n, wrote[last], ticketed[last], seen[roles].min(), seen[roles].max(), end[roles].min(), end[roles].max(),
                 end[~roles].max()))
        order = np.argsort(-ticket[roles])        # role = n - 1 - ticket
        dur = (end[roles] - seen[roles])[order]
        print("    role work (us after seeing every row), role 0 .. 32:", " ".join("%.1f" % v for v in dur))


if __name__ == "__main__":
    main()
