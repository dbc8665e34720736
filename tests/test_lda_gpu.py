"""GPU parity for BASELINE config 4 (LDA-style Dirichlet-Multinomial): the
sufficient statistics are evaluated as a bayesic.algebra expression on the device
(two fp32-MFMA GEMMs + element-wise kernels) and compared with the float64 oracle.
Tolerance rtol 2e-5 on contractions (the reference's is 1e-5 at K=5; here the
contracted extent is up to 2000)."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

from oracle import svi

pytestmark = pytest.mark.gpu


def test_dirichlet_expectation_matches_scipy(ctx):
    rs = np.random.RandomState(0)
    for rows, cols in [(1, 1), (7, 33), (128, 5000), (300, 16)]:
        lam = rs.gamma(1.0, 2.0, (rows, cols)).astype(np.float32) + 1e-3
        d = ctx.to_device(lam)
        out = ctx.zeros((rows, cols))
        ctx.call("bsc_dirichlet_expectation", d, rows, cols, cols, out)
        ctx.sync()
        npt.assert_allclose(out.cpu().numpy(), svi.dirichlet_expectation(lam), rtol=2e-6, atol=1e-37)  # float32 subnormals


@pytest.mark.parametrize("docs,V,K", [(60, 500, 16), (257, 1300, 128), (2000, 333, 8)])
def test_lda_steps_match_oracle(ctx, docs, V, K):
    from bayesic_amd.svi.lda import LDAFixedGammaSVI
    rs = np.random.RandomState(docs + V + K)
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    gamma = rs.gamma(100.0, 0.01, (docs, K)).astype(np.float32)
    lam = rs.gamma(100.0, 0.01, (K, V)).astype(np.float32)
    model = LDAFixedGammaSVI(C, gamma, lam, eta=0.01, docs_total=10 * docs, ctx=ctx)
    lam_ref = lam.astype(np.float64)
    for t in range(1, 4):
        model.step()
        # the oracle restarts from the device's float32 lambda so that rounding of the
        # K x V state does not accumulate into the comparison
        lam_ref, ss = svi.lda_svi_step(lam_ref.astype(np.float32), gamma, C, 0.01, 10 * docs,
                                       (t + 1.0) ** -0.7)
        ctx.sync()
        npt.assert_allclose(model.sstats.cpu().numpy(), ss, rtol=5e-5, atol=1e-6)
        npt.assert_allclose(model.lam.cpu().numpy(), lam_ref, rtol=5e-5, atol=1e-6)
        lam_ref = model.lam.cpu().numpy().astype(np.float64)
    # total expected counts are conserved by the local step: sum_kv sstats = sum_dv C
    npt.assert_allclose(model.sstats.double().sum().item(), C.astype(np.float64).sum(), rtol=1e-5)


@pytest.mark.parametrize("docs,V,K", [
    (32, 128, 128),        # one full tile
    (1, 1, 32),            # a single cell
    (45, 130, 64),         # ragged documents and vocabulary
    (700, 1000, 96),       # K = 96 (three topic tiles), V not a multiple of 4
    (5000, 3000, 128),     # several document splits and the partial reduction
    (0, 77, 128),          # no documents: zeros
])
def test_fused_lda_statistics_match_oracle_and_executor(ctx, docs, V, K):
    """bsc_lda_sstats (both contractions + the division in one pass) against the float64
    oracle and against the same expression run through the algebra executor."""
    from bayesic_amd import algebra as A
    from bayesic_amd.algebra.device_backend import DeviceBackend
    rs = np.random.RandomState(docs + V + K)
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    Th = (rs.rand(docs, K) + 0.05).astype(np.float32)
    Bt = (rs.rand(K, V) + 0.05).astype(np.float32)
    dC, dTh, dBt = ctx.to_device(C), ctx.to_device(Th), ctx.to_device(Bt)
    out = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, out, V)
    ctx.sync()
    got = out.cpu().numpy()
    want = svi.lda_sstats(C, Th, Bt) if docs else np.zeros((K, V))
    npt.assert_allclose(got, want, rtol=3e-5, atol=1e-6)
    if docs:
        Thv, Cv, Bm = A.var("Th", 2), A.var("C", 2), A.var("Bm", 2)
        # (fuse=False: the expression as two products and element-wise launches -- with fusion on, the executor
        # recognises it and calls the very kernel under test: test_lda_expression_through_compile_is_one_pass)
        f = (Bm * A.dot(Thv.T, Cv / A.dot(Thv, Bm))).compile(DeviceBackend(ctx, fuse=False)).device_fn
        ex = f(Th=dTh, C=dC, Bm=dBt)
        ctx.sync()
        npt.assert_allclose(got, ex.cpu().numpy(), rtol=3e-5, atol=1e-6)
    # run-to-run identical (fixed reduction order)
    out2 = torch.empty_like(out)
    ctx.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, out2, V)
    ctx.sync()
    npt.assert_array_equal(got, out2.cpu().numpy())


def test_fused_lda_statistics_strided_operands_and_limits(ctx):
    from bayesic_amd._ffi import BayesicHipError
    rs = np.random.RandomState(9)
    docs, V, K = 100, 203, 32
    Cb = rs.poisson(0.3, (docs, V + 5)).astype(np.float32)       # leading dimensions > extents,
    Thb = (rs.rand(docs, K + 3) + 0.05).astype(np.float32)       # not multiples of 4: scalar loads
    Btb = (rs.rand(K, V + 1) + 0.05).astype(np.float32)
    dC, dTh, dBt = ctx.to_device(Cb), ctx.to_device(Thb), ctx.to_device(Btb)
    out = torch.zeros((K, V + 2), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_lda_sstats", dC, V + 5, docs, V, K, dTh, K + 3, dBt, V + 1, out, V + 2)
    ctx.sync()
    want = svi.lda_sstats(Cb[:, :V], Thb[:, :K], Btb[:, :V])
    npt.assert_allclose(out.cpu().numpy()[:, :V], want, rtol=3e-5, atol=1e-6)
    assert (out.cpu().numpy()[:, V:] == 0).all()                 # padding columns untouched
    with pytest.raises(BayesicHipError):                         # unsupported topic count: no fallback
        ctx.call("bsc_lda_sstats", dC, V + 5, docs, V, 48, dTh, K + 3, dBt, V + 1, out, V + 2)


def test_lda_driver_paths_agree(ctx):
    from bayesic_amd.svi.lda import LDAFixedGammaSVI
    rs = np.random.RandomState(1)
    docs, V, K = 300, 900, 64
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    gamma = rs.gamma(100.0, 0.01, (docs, K)).astype(np.float32)
    lam = rs.gamma(100.0, 0.01, (K, V)).astype(np.float32)
    a = LDAFixedGammaSVI(C, gamma, lam, ctx=ctx, via="kernel")
    b = LDAFixedGammaSVI(C, gamma, lam, ctx=ctx, via="executor")
    assert LDAFixedGammaSVI(C, gamma, lam, ctx=ctx).via == "kernel"
    for _ in range(2):
        a.step()
        b.step()
    ctx.sync()
    npt.assert_allclose(a.lam.cpu().numpy(), b.lam.cpu().numpy(), rtol=3e-5)


@pytest.mark.parametrize("docs,V,K,density", [
    (64, 64, 128, 0.3),       # one word tile
    (300, 1000, 64, 0.05),    # config-4-like density, ragged last tile (1000 = 15*64 + 40)
    (1000, 130, 96, 0.01),    # very sparse, K = 96
    (50, 70, 32, 1.0),        # fully dense stored as sparse
    (40, 200, 128, 0.0),      # no nonzeros at all
])
def test_sparse_lda_statistics_match_dense_paths(ctx, docs, V, K, density):
    """bsc_lda_sstats_csc (one pass over the nonzeros, compressed sparse column) against
    the float64 oracle and the dense fused kernel."""
    import scipy.sparse as sparse
    rs = np.random.RandomState(docs + V + K)
    mask = rs.rand(docs, V) < density
    C = (rs.poisson(2.0, (docs, V)) + 1).astype(np.float32) * mask
    Th = (rs.rand(docs, K) + 0.05).astype(np.float32)
    Bt = (rs.rand(K, V) + 0.05).astype(np.float32)
    csc = sparse.csc_matrix(C)
    colptr = torch.as_tensor(csc.indptr.astype(np.int64)).to(ctx.device)
    rowidx = torch.as_tensor(csc.indices.astype(np.int32)).to(ctx.device)
    vals = torch.as_tensor(csc.data.astype(np.float32)).to(ctx.device)
    dTh, dBt = ctx.to_device(Th), ctx.to_device(Bt)
    out = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_lda_sstats_csc", colptr, rowidx, vals, docs, V, K, dTh, K, dBt, V, out, V)
    ctx.sync()
    got = out.cpu().numpy()
    npt.assert_allclose(got, svi.lda_sstats(C, Th, Bt), rtol=3e-5, atol=1e-6)
    dense = torch.empty_like(out)
    ctx.call("bsc_lda_sstats", ctx.to_device(C), V, docs, V, K, dTh, K, dBt, V, dense, V)
    ctx.sync()
    npt.assert_allclose(got, dense.cpu().numpy(), rtol=3e-5, atol=1e-6)
    out2 = torch.empty_like(out)                   # fixed order: run-to-run identical
    ctx.call("bsc_lda_sstats_csc", colptr, rowidx, vals, docs, V, K, dTh, K, dBt, V, out2, V)
    ctx.sync()
    npt.assert_array_equal(got, out2.cpu().numpy())


def test_lda_driver_with_sparse_counts(ctx):
    import scipy.sparse as sparse
    from bayesic_amd.svi.lda import LDAFixedGammaSVI
    rs = np.random.RandomState(2)
    docs, V, K = 200, 700, 64
    C = (rs.poisson(0.05, (docs, V))).astype(np.float32)
    gamma = rs.gamma(100.0, 0.01, (docs, K)).astype(np.float32)
    lam = rs.gamma(100.0, 0.01, (K, V)).astype(np.float32)
    a = LDAFixedGammaSVI(sparse.csr_matrix(C), gamma, lam, ctx=ctx)
    b = LDAFixedGammaSVI(C, gamma, lam, ctx=ctx)
    assert a.via == "csc" and b.via == "kernel"
    for _ in range(2):
        a.step()
        b.step()
    ctx.sync()
    npt.assert_allclose(a.lam.cpu().numpy(), b.lam.cpu().numpy(), rtol=3e-5)


@pytest.mark.parametrize("docs,V,K", [(700, 67840, 128), (33, 66000, 128), (1000, 132, 128), (4100, 1028, 128),
                                      (700, 67840, 64), (4100, 1028, 64), (45, 260, 64),
                                      (700, 67840, 32), (4100, 1028, 32), (33, 132, 32)])
def test_fused_lda_statistics_persistent_kernel_rounds_and_split_blocks(ctx, docs, V, K):
    """K = 128, 64, 32 through lda_sstats_stream_kernel where the 128-column blocks exceed the resident
    workgroups (a whole round, then left-over blocks split along the documents and added by the
    fix-up pass), with leading dimensions larger than the extents, a ragged last block and a short
    last document step -- against the one-block-per-workgroup kernel (option lda_stream = 0) on the same
    operands and against the float64 oracle on a sample of columns."""
    import os
    from bayesic_amd.device import Context
    plain = Context(0, options=dict(lda_stream=0))
    g = torch.Generator(device=ctx.device).manual_seed(docs + V + K)
    ldc, ldth, ldb, ldo = V + 8, K + 4, V + 4, V + 12
    C = torch.poisson(torch.full((docs, ldc), 0.3, device=ctx.device), generator=g)
    Th = torch.rand((docs, ldth), generator=g, device=ctx.device) + 0.05
    Bt = torch.rand((K, ldb), generator=g, device=ctx.device) + 0.05
    outs = []
    for c in (ctx, plain):
        out = torch.full((K, ldo), float("nan"), dtype=torch.float32, device=ctx.device)
        c.call("bsc_lda_sstats", C, ldc, docs, V, K, Th, ldth, Bt, ldb, out, ldo)
        c.sync()
        outs.append(out.cpu().numpy())
    assert np.isnan(outs[0][:, V:]).all()                         # padding columns untouched
    # the same arithmetic in the same order per column, but the blocks are cut differently along the
    # documents: equal to float32 summation error, not bitwise
    npt.assert_allclose(outs[0][:, :V], outs[1][:, :V], rtol=2e-5, atol=1e-6)
    cols = np.unique(np.concatenate([np.arange(0, min(V, 300)), np.arange(max(0, V - 300), V),
                                     np.random.RandomState(1).randint(0, V, 400)]))
    Cn, Thn, Btn = C.cpu().numpy()[:, cols], Th.cpu().numpy()[:, :K], Bt.cpu().numpy()[:, cols]
    npt.assert_allclose(outs[0][:, cols], svi.lda_sstats(Cn, Thn, Btn), rtol=3e-5, atol=1e-6)
    out2 = torch.full((K, ldo), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_lda_sstats", C, ldc, docs, V, K, Th, ldth, Bt, ldb, out2, ldo)
    ctx.sync()
    npt.assert_array_equal(outs[0][:, :V], out2.cpu().numpy()[:, :V])      # run-to-run identical


# ---- the evidence lower bound of config 4 (oracle.svi.lda_elbo) ---------------------------------

def test_dirichlet_bound_matches_oracle(ctx):
    """bsc_dirichlet_expectation_bound: the same expectations plus sum_r -KL(Dir(lam_r) || Dir(prior))."""
    rs = np.random.RandomState(0)
    for rows, cols, prior in [(1, 1, 0.5), (7, 33, 0.01), (128, 5000, 0.01), (300, 16, 1.0 / 16), (2, 3, 1.0)]:
        lam = rs.gamma(1.0, 2.0, (rows, cols)).astype(np.float32) + 1e-3
        d = ctx.to_device(lam)
        out = ctx.zeros((rows, cols))
        bound = ctx.zeros(1, torch.float64)
        ctx.call("bsc_dirichlet_expectation_bound", d, rows, cols, cols, prior, out, bound)
        ctx.sync()
        npt.assert_allclose(out.cpu().numpy(), svi.dirichlet_expectation(lam), rtol=2e-6, atol=1e-37)
        want = float(svi.dirichlet_neg_kl(lam, prior).sum())
        npt.assert_allclose(bound.item(), want, rtol=1e-11, atol=1e-9)
        again = ctx.zeros(1, torch.float64)
        ctx.call("bsc_dirichlet_expectation_bound", d, rows, cols, cols, prior, out, again)
        ctx.sync()
        assert again.item() == bound.item()                                   # fixed order
    # KL(p || p) = 0
    flat = ctx.to_device(np.full((5, 40), 0.3, np.float32))
    ctx.call("bsc_dirichlet_expectation_bound", flat, 5, 40, 40, float(np.float32(0.3)), ctx.zeros((5, 40)), bound)
    ctx.sync()
    assert abs(bound.item()) < 1e-9


@pytest.mark.parametrize("docs,V,K", [(32, 128, 128), (1, 1, 32), (45, 130, 64), (700, 1000, 96), (5000, 3000, 128),
                                      (700, 67840, 128), (4100, 1028, 64), (33, 132, 32), (0, 77, 128)])
def test_words_term_of_the_bound_inside_the_statistic_kernels(ctx, docs, V, K):
    """bsc_lda_sstats_bound / bsc_lda_sstats_csc_bound: sum_dv C log(phinorm) taken where phinorm lives, on every
    kernel variant (persistent LDS-DMA kernel, one block per workgroup, K = 96, sparse), and the statistic
    itself bit-identical to the entry point without the bound."""
    import os
    import scipy.sparse as sparse
    from bayesic_amd.device import Context
    plain = Context(0, options=dict(lda_stream=0))
    rs = np.random.RandomState(docs + V + K)
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    Th = (rs.rand(docs, K) + 0.05).astype(np.float32)
    Bt = (rs.rand(K, V) + 0.05).astype(np.float32)
    dC, dTh, dBt = ctx.to_device(C), ctx.to_device(Th), ctx.to_device(Bt)
    want = svi.lda_local_bound(C, Th, Bt) if docs else 0.0
    scale = float((C.astype(np.float64) * np.abs(np.log(Th.astype(np.float64) @ Bt.astype(np.float64)))).sum()) if docs else 1.0
    for c in (ctx, plain):
        ref = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
        c.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, ref, V)
        out = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
        ll = torch.full((1,), float("nan"), dtype=torch.float64, device=ctx.device)
        c.call("bsc_lda_sstats_bound", dC, V, docs, V, K, dTh, K, dBt, V, out, V, ll)
        c.sync()
        npt.assert_array_equal(out.cpu().numpy(), ref.cpu().numpy())
        # v_log_f32 is good to ~1 ulp of log2, float32 products and per-lane float32 partial sums
        assert abs(ll.item() - want) <= 3e-6 * scale + 1e-9, (ll.item(), want)
    if docs:
        csc = sparse.csc_matrix(C)
        colptr = torch.as_tensor(csc.indptr.astype(np.int64)).to(ctx.device)
        rowidx = torch.as_tensor(csc.indices.astype(np.int32)).to(ctx.device)
        vals = torch.as_tensor(csc.data.astype(np.float32)).to(ctx.device)
        ref = torch.empty((K, V), dtype=torch.float32, device=ctx.device)
        ctx.call("bsc_lda_sstats_csc", colptr, rowidx, vals, docs, V, K, dTh, K, dBt, V, ref, V)
        out = torch.empty_like(ref)
        ll = torch.full((1,), float("nan"), dtype=torch.float64, device=ctx.device)
        ctx.call("bsc_lda_sstats_csc_bound", colptr, rowidx, vals, docs, V, K, dTh, K, dBt, V, out, V, ll)
        ctx.sync()
        npt.assert_array_equal(out.cpu().numpy(), ref.cpu().numpy())
        assert abs(ll.item() - want) <= 3e-6 * scale + 1e-9, (ll.item(), want)


@pytest.mark.parametrize("via", ["kernel", "executor", "csc"])
def test_lda_elbo_tracks_the_oracle_and_rises_under_unit_steps(ctx, via):
    import scipy.sparse as sparse
    from bayesic_amd.svi.lda import LDAFixedGammaSVI
    rs = np.random.RandomState(3)
    docs, V, K = 300, 900, 64
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    gamma = (rs.gamma(2.0, 1.0, (docs, K)) + 0.1).astype(np.float32)
    lam = (rs.gamma(1.0, 1.0, (K, V)) + 0.05).astype(np.float32)
    Carg = sparse.csr_matrix(C) if via == "csc" else C
    model = LDAFixedGammaSVI(Carg, gamma, lam, eta=0.02, docs_total=10 * docs, ctx=ctx, alpha=0.3,
                             via=None if via == "csc" else via)
    assert model.via == via
    for t in range(1, 4):
        before = model.lam.cpu().numpy()
        model.step()
        ctx.sync()
        want = svi.lda_elbo(before, gamma, C, 0.02, 0.3, 10.0 * docs)
        npt.assert_allclose(model.elbo.item(), want, rtol=3e-6)
    full = LDAFixedGammaSVI(Carg, gamma, lam, eta=0.02, docs_total=docs, ctx=ctx, alpha=0.3,
                            via=None if via == "csc" else via)
    bounds = []
    for _ in range(6):
        full.step(rho=1.0)
        ctx.sync()
        bounds.append(full.elbo.item())
    bounds = np.array(bounds)
    assert np.all(np.diff(bounds) >= -2e-6 * np.abs(bounds[:-1])), bounds
    assert bounds[-1] > bounds[0]
    # elbo=False: the same lambda without the bound's work
    quiet = LDAFixedGammaSVI(Carg, gamma, lam, eta=0.02, docs_total=docs, ctx=ctx, alpha=0.3,
                             via=None if via == "csc" else via, elbo=False)
    loud = LDAFixedGammaSVI(Carg, gamma, lam, eta=0.02, docs_total=docs, ctx=ctx, alpha=0.3,
                            via=None if via == "csc" else via)
    quiet.step()
    loud.step()
    ctx.sync()
    npt.assert_allclose(quiet.lam.cpu().numpy(), loud.lam.cpu().numpy(), rtol=1e-5)


def test_lda_expression_through_compile_is_one_pass(ctx):
    """``Bt * dot(Th.T, C / dot(Th, Bt))`` written on the plugin surface and evaluated through
    ``compile()`` reaches bsc_lda_sstats -- ONE pass over the counts, bit-identical to calling the entry point --
    while expressions that only look like it take the general route and still get the right numbers."""
    from bayesic_amd import algebra as A
    from bayesic_amd.algebra.device_backend import DeviceBackend
    rs = np.random.RandomState(11)
    docs, V, K = 700, 1000, 64
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    Th = (rs.rand(docs, K) + 0.05).astype(np.float32)
    Bt = (rs.rand(K, V) + 0.05).astype(np.float32)
    Bt2 = (rs.rand(K, V) + 0.05).astype(np.float32)
    dC, dTh, dBt, dBt2 = (ctx.to_device(v) for v in (C, Th, Bt, Bt2))
    Thv, Cv, Bm, Bo = A.var("Th", 2), A.var("C", 2), A.var("Bm", 2), A.var("Bo", 2)
    be = DeviceBackend(ctx)
    calls = []
    real_call = ctx.call
    ctx.call = lambda name, *a: (calls.append(name), real_call(name, *a))[1]
    try:
        got = (Bm * A.dot(Thv.T, Cv / A.dot(Thv, Bm))).compile(be).device_fn(Th=dTh, C=dC, Bm=dBt)
        ctx.sync()
        assert calls == ["bsc_lda_sstats"], calls
        direct = torch.empty((K, V), dtype=torch.float32, device=ctx.device)
        real_call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, direct, V)
        ctx.sync()
        npt.assert_array_equal(got.cpu().numpy(), direct.cpu().numpy())
        # a scalar factor rides along
        calls.clear()
        scaled = (2.5 * (Bm * A.dot(Thv.T, Cv / A.dot(Thv, Bm)))).compile(be).device_fn(Th=dTh, C=dC, Bm=dBt)
        ctx.sync()
        assert calls.count("bsc_lda_sstats") == 1 and "bsc_gemm_epilogue" not in calls
        npt.assert_allclose(scaled.cpu().numpy(), 2.5 * direct.cpu().numpy(), rtol=1e-6)
        # another matrix outside than inside: not the statistic -- two products, right numbers
        calls.clear()
        other = (Bo * A.dot(Thv.T, Cv / A.dot(Thv, Bm))).compile(be).device_fn(Th=dTh, C=dC, Bm=dBt, Bo=dBt2)
        ctx.sync()
        assert "bsc_lda_sstats" not in calls
        want = Bt2.astype(np.float64) * (Th.astype(np.float64).T @ (C / (Th.astype(np.float64) @ Bt.astype(np.float64))))
        npt.assert_allclose(other.cpu().numpy(), want, rtol=3e-5, atol=1e-6)
        # K outside the kernel's topic counts: general route
        calls.clear()
        K2 = 40
        Th2, Bt3 = ctx.to_device(Th[:, :K2].copy()), ctx.to_device(Bt[:K2].copy())
        gen = (Bm * A.dot(Thv.T, Cv / A.dot(Thv, Bm))).compile(be).device_fn(Th=Th2, C=dC, Bm=Bt3)
        ctx.sync()
        assert "bsc_lda_sstats" not in calls
        npt.assert_allclose(gen.cpu().numpy(), svi.lda_sstats(C, Th[:, :K2], Bt[:K2]), rtol=3e-5, atol=1e-6)
    finally:
        ctx.call = real_call
