"""The exchange step through the C ABI with RCCL really in the loop ("the
RCCL path exercised at world size 1"), and bench.py started bare for N > 1.

A one-GPU box cannot host two RCCL ranks (one device per rank), so: (1) a one-rank communicator
runs ncclAllReduce on the context stream inside the real update loop and must leave the update
bit-identical to the same loop without a communicator; (2) `python bench.py --gpus 2` with no
launcher and no WORLD_SIZE starts its own ranks (rehearsal: both on GPU 0, gloo exchange)."""
import json
import os
import subprocess
import sys

import numpy as np
import numpy.testing as npt
import pytest
import torch

from oracle import svi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_ctx():
    from bayesic_amd.device import Context
    c = Context(0)
    c.comm_init(c.comm_unique_id(), 0, 1)
    yield c
    c.close()


def test_allreduce_world1_is_identity_and_runs_rccl(rccl_ctx):
    info = rccl_ctx.comm_info()
    assert rccl_ctx.has_comm and info["world"] == 1 and info["rank"] == 0
    assert info["rccl_version"] > 20000, info          # librccl was bound and answered
    for dtype in (torch.float64, torch.float32):
        v = torch.arange(4099, dtype=dtype, device=rccl_ctx.device) * 0.5 - 7
        ref = v.clone()
        rccl_ctx.profile(1)
        rccl_ctx.allreduce_sum(v)
        ms, n = rccl_ctx.profile_read(1)               # the collective was enqueued and timed
        rccl_ctx.profile(0)
        assert n == 1 and ms > 0.0
        npt.assert_array_equal(v.cpu().numpy(), ref.cpu().numpy())
    m = torch.tensor([3.0, -1.0], dtype=torch.float64, device=rccl_ctx.device)
    rccl_ctx.allreduce_max(m)
    npt.assert_array_equal(m.cpu().numpy(), [3.0, -1.0])


def test_second_communicator_on_one_context_is_refused(rccl_ctx):
    from bayesic_amd._ffi import BayesicHipError
    with pytest.raises(BayesicHipError, match="already has a communicator"):
        rccl_ctx.comm_init(rccl_ctx.comm_unique_id(), 0, 1)


def test_update_loop_through_rccl_equals_loop_without(ctx, rccl_ctx):
    """cfg-2 driver: data pass -> float64 statistics -> ncclAllReduce (world 1) -> fused finish,
    against the same N>1 code path with no communicator.  Bit-identical."""
    from bayesic_amd.svi.blr import BLRReparamSVI
    from bayesic_amd.svi.mog import MoGNatGradSVI
    X, y, _ = svi.make_cfg2(30000, 256)
    a = BLRReparamSVI(rccl_ctx.to_device(X), rccl_ctx.to_device(y), n_total=300000, n_samples=8,
                      seed=11, lr=0.02, ctx=rccl_ctx)
    assert a.exchange.rccl and a.world == 1 and a.batch_rows == 30000.0
    b = BLRReparamSVI(ctx.to_device(X), ctx.to_device(y), n_total=300000, n_samples=8, seed=11,
                      lr=0.02, ctx=ctx, fused=False)
    assert not b.exchange.rccl
    rccl_ctx.profile(1)
    for _ in range(4):
        a.step()
        b.step()
    ms, n = rccl_ctx.profile_read(1)
    rccl_ctx.profile(0)
    ctx.sync()
    assert n == 4, "one all-reduce per update"
    npt.assert_array_equal(a.lam.cpu().numpy(), b.lam.cpu().numpy())
    npt.assert_array_equal(a.elbo.cpu().numpy(), b.elbo.cpu().numpy())
    # and the oracle on the same inputs
    lam = svi.blr_init_lam(256)
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    for t in range(1, 5):
        lam, m1, m2, elbo, _ = svi.blr_step(lam, m1, m2, t, X, y, 8, 11, 300000, 0.02)
    npt.assert_allclose(a.lam.cpu().numpy(), lam, rtol=2e-5, atol=2e-6)
    npt.assert_allclose(a.elbo.item(), elbo, rtol=1e-6)

    Xm, _, _ = svi.make_cfg3(30000, 8, 5)
    eta0, eta_init = svi.mog_prior_eta(5, 8), svi.mog_init_eta(Xm[:500], 5, 8, seed=2)
    ma = MoGNatGradSVI(rccl_ctx.to_device(Xm), 5, eta0, eta_init, n_total=300000, ctx=rccl_ctx)
    mb = MoGNatGradSVI(ctx.to_device(Xm), 5, eta0, eta_init, n_total=300000, ctx=ctx)
    for _ in range(3):
        ma.step()
        mb.step()
    ctx.sync()
    npt.assert_array_equal(ma.eta.cpu().numpy(), mb.eta.cpu().numpy())


def test_lda_statistic_in_overlapped_pieces_equals_the_one_collective_update(ctx, rccl_ctx):
    """Config 4: with a communicator the statistic is taken in column ranges -- whole rounds of the
    persistent kernel -- each staged contiguously and all-reduced on the context's second stream while the next range is
    computed (bsc_allreduce_sum_begin / _end); the natural-gradient step of a range waits for its own collective only.
    Bit-identical to the update with ONE collective after the whole statistic, and to the update without a
    communicator; every piece's collective is timed in slot 1."""
    from bayesic_amd.svi.lda import LDAFixedGammaSVI
    docs, V, K = 700, 140_000, 128                   # 1 094 column blocks: two whole rounds of 512 and a tail of 70
    dev = ctx.device
    g = torch.Generator(device=dev).manual_seed(4)
    C = torch.poisson(torch.full((docs, V), 0.05, device=dev), generator=g)
    gamma = torch.rand((docs, K), generator=g, device=dev) + 0.5
    lam0 = torch.rand((K, V), generator=g, device=dev) + 0.5
    models = {
        "pieces": LDAFixedGammaSVI(C, gamma, lam0, docs_total=7000.0, ctx=rccl_ctx),
        "one": LDAFixedGammaSVI(C, gamma, lam0, docs_total=7000.0, ctx=rccl_ctx, overlap=False),
        "alone": LDAFixedGammaSVI(C, gamma, lam0, docs_total=7000.0, ctx=ctx),
    }
    assert models["pieces"].exchange.rccl and not models["alone"].exchange.active
    rccl_ctx.profile(1)
    for _ in range(3):
        models["pieces"].step()
    rccl_ctx.sync()
    ms, n = rccl_ctx.profile_read(1)
    rccl_ctx.profile(0)
    pieces = models["pieces"]._pieces
    assert [c for _, c in pieces] == [65536, 65536, 140_000 - 2 * 65536], pieces
    assert n == 3 * (len(pieces) + 1) and ms > 0.0          # three ranges and the bound's two sums, per update
    for name in ("one", "alone"):
        for _ in range(3):
            models[name].step()
    rccl_ctx.sync()
    ctx.sync()
    want = models["alone"].lam.cpu().numpy()
    npt.assert_array_equal(models["pieces"].lam.cpu().numpy(), want)
    npt.assert_array_equal(models["one"].lam.cpu().numpy(), want)
    # (the words' term is summed per range from float32 per-wave partials: another grouping than in one call)
    npt.assert_allclose(models["pieces"].elbo.item(), models["alone"].elbo.item(), rtol=1e-8)
    npt.assert_allclose(models["one"].elbo.item(), models["alone"].elbo.item(), rtol=1e-12)
    # begin / end bookkeeping: a slot cannot be begun twice, an idle slot ends as a no-op
    from bayesic_amd._ffi import BayesicHipError
    v = torch.ones(1024, device=rccl_ctx.device)
    rccl_ctx.allreduce_sum_begin(v, 3)
    with pytest.raises(BayesicHipError, match="was not ended"):
        rccl_ctx.allreduce_sum_begin(v, 3)
    rccl_ctx.allreduce_sum_end(3)
    rccl_ctx.allreduce_sum_end(3)
    ctx.allreduce_sum_begin(v.to(ctx.device), 0)           # no communicator: both are no-ops
    ctx.allreduce_sum_end(0)
    rccl_ctx.sync()
    npt.assert_array_equal(v.cpu().numpy(), np.ones(1024, np.float32))


def _bench(extra, env_extra=None, timeout=600):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    lines = [l for l in proc.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout.decode()
    return json.loads(lines[0])


def test_bench_bare_two_ranks_rehearsal():
    """`python bench.py --gpus 2` from a bare shell: the parent spawns the ranks itself."""
    out = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--rows", "40000", "--no-cpu-baseline",
                  "--spin-up-ms", "5", "--burst", "8"], {"BSC_BENCH_REHEARSAL": "1"})
    # default --scaling both: value = ONE global mini-batch of the configured size split over the ranks (the metric's
    # reading), value_weak = a full-size mini-batch per rank, both from the one invocation
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "strong"
    assert out["config"]["rows_per_gpu"] == 20000 and out["config"]["global_rows"] == 40000 and out["value"] > 0
    assert out["value_weak"] > 0 and out["config_weak"]["rows_per_gpu"] == 40000
    assert out["timed_blocks"]["n"] >= 1 and out["timed_blocks_weak"]["steps_per_block"] == 6
    assert out["exchange"].startswith("gloo")
    weak = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--rows", "40000", "--scaling", "weak",
                   "--no-cpu-baseline", "--spin-up-ms", "5", "--burst", "8"], {"BSC_BENCH_REHEARSAL": "1"})
    assert weak["scaling"] == "weak" and weak["config"]["rows_per_gpu"] == 40000 and "value_weak" not in weak
    strong = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--rows", "40000", "--scaling",
                     "strong", "--no-cpu-baseline", "--spin-up-ms", "5", "--burst", "8"],
                    {"BSC_BENCH_REHEARSAL": "1"})
    assert strong["scaling"] == "strong" and strong["config"]["rows_per_gpu"] == 20000
    assert strong["config"]["global_rows"] == 40000


def test_bench_world1_rccl_line():
    out = _bench(["--steps", "6", "--warmup", "2", "--rows", "40000", "--no-cpu-baseline", "--rccl-world1",
                  "--spin-up-ms", "5", "--time-every", "1", "--burst", "50"])
    assert out["rccl_ranks"] == 1 and out["allreduce_us"] is not None and out["allreduce_us"] > 0
    assert out["roofline"]["launches"] == 50 and out["roofline"]["bound"] == "hbm"
    assert out["finish_us"] is not None
