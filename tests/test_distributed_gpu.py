"""The N>1 branch of the drivers with REAL kernels: two ranks share GPU 0 and exchange
their float64 statistics over gloo (RCCL needs one device per rank; the 8-GPU run is the
driver's).  Both ranks must end with identical parameters, equal to the single-process
update on the whole mini-batch up to float32 partial-sum rounding: a shard is cut into
workgroup partials differently from the whole batch, and partials are float32 (1e-7
relative per statistic, amplified by a few Adam steps).  With ``reproducible=True`` the two
ranks and the single process agree bit for bit."""
import os
import subprocess
import sys

import numpy as np
import numpy.testing as npt
import pytest

from oracle import svi

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu_equal_single_process(ctx, tmp_path):
    port = 29600 + (os.getpid() % 2000)
    out = str(tmp_path / "rank%d.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_gpu_worker.py"), out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    npt.assert_array_equal(r0["lam"], r1["lam"])
    npt.assert_array_equal(r0["eta"], r1["eta"])

    from bayesic_amd.svi.blr import BLRReparamSVI
    from bayesic_amd.svi.mog import MoGNatGradSVI
    X, y, _ = svi.make_cfg2(20000, 64)
    single = BLRReparamSVI(ctx.to_device(X), ctx.to_device(y), n_total=200000, n_samples=8, seed=11,
                           lr=0.02, ctx=ctx, fused=False)
    for _ in range(4):
        single.step()
    ctx.sync()
    npt.assert_allclose(r0["lam"], single.lam.cpu().numpy(), rtol=5e-6, atol=1e-8)
    npt.assert_allclose(r0["elbo"], single.elbo.cpu().numpy(), rtol=1e-6)
    # reproducible=True: BIT-identical to the one-GPU run of the same mode --
    # a pass per virtual shard, block-sparse all-reduce, fixed-order n-ary add
    npt.assert_array_equal(r0["lam_repro"], r1["lam_repro"])
    one = BLRReparamSVI(ctx.to_device(X), ctx.to_device(y), n_total=200000, n_samples=8, seed=11,
                        lr=0.02, ctx=ctx, reproducible=True)
    for _ in range(4):
        one.step()
    ctx.sync()
    npt.assert_array_equal(r0["lam_repro"], one.lam.cpu().numpy())
    npt.assert_array_equal(r0["elbo_repro"], one.elbo.cpu().numpy())
    npt.assert_allclose(one.lam.cpu().numpy(), single.lam.cpu().numpy(), rtol=5e-6, atol=1e-8)   # and it is the same update
    with pytest.raises(ValueError, match="virtual shards"):
        BLRReparamSVI(ctx.to_device(X[:20001 - 8]), ctx.to_device(y[:20001 - 8]), n_samples=8, ctx=ctx, reproducible=True)
    Xm, _, _ = svi.make_cfg3(30000, 8, 5)
    mog = MoGNatGradSVI(ctx.to_device(Xm), 5, svi.mog_prior_eta(5, 8),
                        svi.mog_init_eta(Xm[:500], 5, 8, seed=2), n_total=300000, ctx=ctx)
    for _ in range(3):
        mog.step()
    ctx.sync()
    npt.assert_allclose(r0["eta"], mog.eta.cpu().numpy(), rtol=2e-5, atol=1e-6)    # float32 partials
