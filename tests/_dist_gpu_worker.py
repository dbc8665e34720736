"""Worker for tests/test_distributed_gpu.py: one of two ranks that SHARE GPU 0.

The collective runs over gloo (NCCL/RCCL refuses two ranks on one device), everything
else is the real product path: HIP kernels through the C ABI, the N>1 branch of the
drivers (data pass -> float64 statistics -> all-reduce -> fused finish)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import svi  # noqa: E402  (data generators only)


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesic_amd.device import Context
    from bayesic_amd.svi.blr import BLRReparamSVI
    from bayesic_amd.svi.mog import MoGNatGradSVI
    ctx = Context(0)

    X, y, _ = svi.make_cfg2(20000, 64)
    cuts = [0, 12000, 20000]
    sl = slice(cuts[rank], cuts[rank + 1])
    model = BLRReparamSVI(ctx.to_device(X[sl]), ctx.to_device(y[sl]), n_total=200000, n_samples=8,
                          seed=11, lr=0.02, ctx=ctx)
    assert model.world == 2 and model.batch_rows == 20000.0
    for _ in range(4):
        model.step()
    ctx.sync()

    # reproducible mode: whole virtual shards per rank (8 shards of 2500 rows: 5 + 3)
    rc = [0, 12500, 20000]
    rsl = slice(rc[rank], rc[rank + 1])
    repro = BLRReparamSVI(ctx.to_device(X[rsl]), ctx.to_device(y[rsl]), n_total=200000, n_samples=8,
                          seed=11, lr=0.02, ctx=ctx, reproducible=True)
    assert repro._first_shard == (0, 5)[rank] and repro._n_shards == (5, 3)[rank]
    for _ in range(4):
        repro.step()
    ctx.sync()

    Xm, _, _ = svi.make_cfg3(30000, 8, 5)
    mc = [0, 17000, 30000]
    eta0 = svi.mog_prior_eta(5, 8)
    eta_init = svi.mog_init_eta(Xm[:500], 5, 8, seed=2)
    mog = MoGNatGradSVI(ctx.to_device(Xm[mc[rank]:mc[rank + 1]]), 5, eta0, eta_init, n_total=300000,
                        ctx=ctx)
    for _ in range(3):
        mog.step()
    ctx.sync()
    np.savez(out_path % rank, lam=model.lam.cpu().numpy(), elbo=model.elbo.cpu().numpy(),
             eta=mog.eta.cpu().numpy(), lam_repro=repro.lam.cpu().numpy(), elbo_repro=repro.elbo.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
