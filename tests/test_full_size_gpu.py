"""BASELINE.json's full sizes for configs 3, 4 and 5 on one GPU, checked through properties that
do not need an oracle run of that size (the float64 numpy oracle takes minutes there):
conservation laws, additivity over row shards (the data-parallel contract), closed-form values
at special parameters, run-to-run identity.  Config 2's full-size test lives in
test_blr_gpu.py::test_cfg2_full_size_properties.  Data is generated on the device."""
import math

import numpy as np
import numpy.testing as npt
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cfg3_full_size_mixture_statistics(ctx):
    """10M x 16, K = 64: responsibilities sum to one per row (sum_k R_k = N), first moments are
    conserved (sum_k S1[k, d] = sum_n x_nd), and two row shards add up to the whole."""
    N, D, K = 10_000_000, 16, 64
    g = torch.Generator(device=ctx.device).manual_seed(3)
    centres = torch.randn((K, D), generator=g, device=ctx.device) * 4
    labels = torch.randint(K, (N,), generator=g, device=ctx.device)
    X = centres[labels] + torch.randn((N, D), generator=g, device=ctx.device)
    T = torch.ones((K, D), device=ctx.device)
    Wmat = torch.cat([T * centres, -0.5 * T], dim=1).contiguous()
    c = (-0.5 * (T * centres ** 2).sum(1)).contiguous()

    def estep(rows):
        stats = ctx.zeros((K, 1 + 2 * D), torch.float64)
        lse = ctx.zeros(1, torch.float64)
        ctx.call("bsc_mog_estep", rows, D, rows.shape[0], D, K, Wmat, c, stats, lse)
        ctx.sync()
        return stats.cpu().numpy(), lse.item()

    whole, lse = estep(X)
    npt.assert_allclose(whole[:, 0].sum(), N, rtol=1e-6)
    col = X.double().sum(0).cpu().numpy()
    scale = X.double().abs().sum(0).cpu().numpy()
    assert (np.abs(whole[:, 1:1 + D].sum(0) - col) <= 2e-6 * scale).all()
    sq = (X.double() ** 2).sum(0).cpu().numpy()
    npt.assert_allclose(whole[:, 1 + D:].sum(0), sq, rtol=2e-6)
    a, la = estep(X[:6_000_000])
    b, lb = estep(X[6_000_000:])
    npt.assert_allclose(a + b, whole, rtol=2e-6, atol=1e-3)
    # per-wave float32 accumulation over differently partitioned rows: ~4e-9 expected
    npt.assert_allclose(la + lb, lse, rtol=2e-8)
    # well separated clusters: almost every row belongs to its own centre
    counts = torch.bincount(labels, minlength=K).double().cpu().numpy()
    npt.assert_allclose(whole[:, 0], counts, rtol=2e-3)
    again, _ = estep(X)
    npt.assert_array_equal(again, whole)


def test_cfg4_full_size_counts_are_conserved(ctx):
    """6250 x 100 000 counts, K = 128 (one GPU's shard of config 4): the local step distributes
    every count over the topics, so sum_kv sstats = sum_dv C; the dense and the sparse kernel
    agree; two document shards add up to the whole."""
    docs, V, K = 6250, 100_000, 128
    g = torch.Generator(device=ctx.device).manual_seed(5)
    C = torch.poisson(torch.full((docs, V), 0.05, device=ctx.device), generator=g)
    Th = torch.rand((docs, K), generator=g, device=ctx.device) + 0.1
    Bt = torch.rand((K, V), generator=g, device=ctx.device) + 0.1

    def sstats(Cs, Ths):
        out = torch.empty((K, V), device=ctx.device)
        ctx.call("bsc_lda_sstats", Cs, V, Cs.shape[0], V, K, Ths, K, Bt, V, out, V)
        ctx.sync()
        return out

    whole = sstats(C, Th)
    total = C.double().sum().item()
    npt.assert_allclose(whole.double().sum().item(), total, rtol=1e-5)
    # per word as well: sum_k sstats[k, v] = sum_d C[d, v]
    per_word = C.double().sum(0)
    assert ((whole.double().sum(0) - per_word).abs() <= 2e-5 * per_word + 1e-6).all()
    parts = sstats(C[:4000], Th[:4000]) + sstats(C[4000:], Th[4000:])
    assert ((parts - whole).abs() <= 3e-5 * whole.abs() + 1e-6).all()
    nz = (C.t() != 0).nonzero()
    colptr = torch.zeros(V + 1, dtype=torch.int64, device=ctx.device)
    colptr[1:] = torch.cumsum(torch.bincount(nz[:, 0], minlength=V), 0)
    rowidx = nz[:, 1].to(torch.int32).contiguous()
    vals = C.t()[nz[:, 0], nz[:, 1]].contiguous()
    sparse = torch.empty_like(whole)
    ctx.call("bsc_lda_sstats_csc", colptr, rowidx, vals, docs, V, K, Th, K, Bt, V, sparse, V)
    ctx.sync()
    assert ((sparse - whole).abs() <= 3e-5 * whole.abs() + 1e-6).all()


def test_cfg5_full_size_log_likelihood(ctx):
    """1M x 256, G = 1000, S = 64: with w = 0 and b = 0 every logit is 0 and
    ell_s = -N log 2; with w = 0 and a constant intercept c the value is
    sum_n [y_n c - softplus(c)]; two row shards add up to the whole.  Per-tile float32 sums
    enter float64 accumulators, so even these sums of IDENTICAL terms (systematic rounding, 1.8e-6
    relative with a float32 accumulator) come out to a few 1e-7."""
    N, D, G, S = 1_000_000, 256, 1000, 64
    g = torch.Generator(device=ctx.device).manual_seed(6)
    X = torch.randn((N, D), generator=g, device=ctx.device)
    y = (torch.rand(N, generator=g, device=ctx.device) < 0.3).float()
    grp = torch.randint(G, (N,), generator=g, device=ctx.device).to(torch.int32)

    def loglik(Xs, ys, gs, Wz, Bz):
        ell = ctx.zeros(S, torch.float64)
        ctx.call("bsc_logreg_bbvi_loglik", Xs, D, ys, gs, Xs.shape[0], D, G, Wz, Bz, S, ell)
        ctx.sync()
        return ell.cpu().numpy()

    zeroW = torch.zeros((S, D), device=ctx.device)
    npt.assert_allclose(loglik(X, y, grp, zeroW, torch.zeros((G, S), device=ctx.device)),
                        -N * math.log(2.0), rtol=5e-7)
    cs = torch.linspace(-3.0, 3.0, S, device=ctx.device)
    Bz = cs[None, :].repeat(G, 1).contiguous()
    n1 = y.double().sum().item()
    want = n1 * cs.double().cpu().numpy() - N * np.logaddexp(0.0, cs.double().cpu().numpy())
    bound = N * (np.abs(cs.double().cpu().numpy()) + 1.0)
    assert (np.abs(loglik(X, y, grp, zeroW, Bz) - want) <= 5e-7 * bound).all()
    Wz = torch.randn((S, D), generator=g, device=ctx.device) / 16
    Bz = torch.randn((G, S), generator=g, device=ctx.device)
    whole = loglik(X, y, grp, Wz, Bz)
    parts = loglik(X[:600_000], y[:600_000], grp[:600_000], Wz, Bz) + \
        loglik(X[600_000:], y[600_000:], grp[600_000:], Wz, Bz)
    npt.assert_allclose(parts, whole, rtol=2e-6)
    npt.assert_array_equal(loglik(X, y, grp, Wz, Bz), whole)
    assert (whole < 0).all() and np.isfinite(whole).all()
