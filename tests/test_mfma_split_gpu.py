"""The operand-split bf16 route of the MFMA-bound contractions (bsc_ctx_set_mfma_split; csrc/bsc_bf16split.h):
off by default, and when on it has to pass the SAME comparisons with the float64 oracle, at the SAME tolerances,
as the f32 route (tests/test_lda_gpu.py) -- two terms as well as three."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

from oracle import svi

pytestmark = pytest.mark.gpu


@pytest.fixture
def split_ctx(ctx):
    def set_terms(n):
        ctx.call("bsc_ctx_set_mfma_split", n)
    yield set_terms
    ctx.call("bsc_ctx_set_mfma_split", 0)


def _rel(got, want):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)))


@pytest.mark.parametrize("terms", [2, 3])
@pytest.mark.parametrize("docs,V", [
    (32, 128),             # one full tile
    (1, 4),                # a single document
    (45, 132),             # ragged documents and vocabulary
    (257, 1300),
    (5000, 3000),          # column blocks split along the documents, partial statistics added by the fix-up
    (70, 70000),           # whole rounds of column blocks, then a split tail
])
def test_lda_statistics_and_bound_on_split_operands(ctx, split_ctx, terms, docs, V):
    K = 128
    rs = np.random.RandomState(docs + V + terms)
    C = rs.poisson(0.3, (docs, V)).astype(np.float32)
    Th = (rs.rand(docs, K) + 0.05).astype(np.float32)
    Bt = (rs.rand(K, V) + 0.05).astype(np.float32)
    dC, dTh, dBt = ctx.to_device(C), ctx.to_device(Th), ctx.to_device(Bt)
    want = svi.lda_sstats(C, Th, Bt)
    want_ll = svi.lda_local_bound(C, Th, Bt)
    f32 = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, f32, V)
    split_ctx(terms)
    out = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, out, V)
    out_b = torch.full((K, V), float("nan"), dtype=torch.float32, device=ctx.device)
    ll = ctx.zeros(1, torch.float64)
    ctx.call("bsc_lda_sstats_bound", dC, V, docs, V, K, dTh, K, dBt, V, out_b, V, ll)
    ctx.sync()
    got = out.cpu().numpy()
    npt.assert_allclose(got, want, rtol=3e-5, atol=1e-6)               # tests/test_lda_gpu.py's tolerance
    npt.assert_array_equal(got, out_b.cpu().numpy())
    scale = float((C.astype(np.float64) * np.abs(np.log(Th.astype(np.float64) @ Bt.astype(np.float64)))).sum())
    assert abs(ll.item() - want_ll) <= 3e-6 * scale + 1e-9
    assert not np.array_equal(got, f32.cpu().numpy()) or docs * V < 1000     # it IS another arithmetic
    # three terms: the f32 class -- no further from the float64 oracle than twice the f32 route
    if terms == 3:
        assert _rel(got, want) <= 2.0 * _rel(f32.cpu().numpy(), want) + 1e-7
    # run-to-run identical
    out2 = torch.empty_like(out)
    ctx.call("bsc_lda_sstats", dC, V, docs, V, K, dTh, K, dBt, V, out2, V)
    ctx.sync()
    npt.assert_array_equal(got, out2.cpu().numpy())


def test_split_is_off_by_default_and_rejects_other_term_counts(ctx):
    from bayesic_amd._ffi import BayesicHipError
    with pytest.raises(BayesicHipError):
        ctx.call("bsc_ctx_set_mfma_split", 4)
    with pytest.raises(BayesicHipError):
        ctx.call("bsc_ctx_set_mfma_split", 1)
    ctx.call("bsc_ctx_set_mfma_split", 0)
