"""The uniform-base ("fast") loaders against the general ones they replace for interior tiles.

Kernel selection is a property of the context (bsc_ctx_set_option), so two contexts in one process give
both code paths on the same device buffers.  The GEMM variants do the same arithmetic
in the same order: their results must be bit-identical.  The sparse LDA variants differ in the
division (v_rcp_f32 vs IEEE): 1e-6 relative."""
import os

import numpy as np
import numpy.testing as npt
import pytest

pytestmark = pytest.mark.gpu


def _context_with(monkeypatch, **options):
    from bayesic_amd.device import Context
    return Context(0, options={k: int(v) for k, v in options.items()})


@pytest.mark.parametrize("m,n,k,a_m_contig", [(512, 384, 1024, False), (512, 384, 1024, True),
                                              (300, 260, 777, False), (1024, 256, 4096, True)])
def test_gemm_fast_loader_is_bit_identical(monkeypatch, m, n, k, a_m_contig):
    import torch
    fast = _context_with(monkeypatch, gemm_fast=1)
    slow = _context_with(monkeypatch, gemm_fast=0)
    g = torch.Generator(device=fast.device).manual_seed(m + n + k)
    if a_m_contig:
        A = torch.randn((k, m), generator=g, device=fast.device)       # stored [k][m]
        sa_m, sa_k = 1, m
    else:
        A = torch.randn((m, k), generator=g, device=fast.device)
        sa_m, sa_k = k, 1
    B = torch.randn((k, n), generator=g, device=fast.device)
    outs = []
    for c in (fast, slow):
        C = torch.full((m, n), float("nan"), device=c.device)
        c.call("bsc_gemm_strided_batched", 0, 1, m, n, k, A, 0, sa_m, sa_k, B, 0, n, 1, C, 0, n, 1)
        c.sync()
        outs.append(C.cpu().numpy())
    npt.assert_array_equal(outs[0], outs[1])
    ref = (A.t() if a_m_contig else A).double() @ B.double()
    npt.assert_allclose(outs[0], ref.cpu().numpy(), rtol=0, atol=2e-4 * np.sqrt(k))


def test_sparse_lda_fast_gathers_match_the_general_path(monkeypatch):
    import scipy.sparse as sp
    import torch
    fast = _context_with(monkeypatch, csc_fast=1)
    slow = _context_with(monkeypatch, csc_fast=0)
    rs = np.random.RandomState(12)
    docs, V, K = 700, 1500, 128
    C = rs.poisson(0.08, (docs, V)).astype(np.float32)
    csc = sp.csc_matrix(C)
    dev = fast.device
    colptr = torch.from_numpy(csc.indptr.astype(np.int64)).to(dev)
    rowidx = torch.from_numpy(csc.indices.astype(np.int32)).to(dev)
    vals = torch.from_numpy(csc.data.astype(np.float32)).to(dev)
    Th = torch.from_numpy(rs.uniform(0.1, 1.0, (docs, K)).astype(np.float32)).to(dev)
    Bt = torch.from_numpy(rs.uniform(0.1, 1.0, (K, V)).astype(np.float32)).to(dev)
    outs = []
    for c in (fast, slow):
        out = torch.full((K, V), float("nan"), device=dev)
        c.call("bsc_lda_sstats_csc", colptr, rowidx, vals, docs, V, K, Th, K, Bt, V, out, V)
        c.sync()
        outs.append(out.cpu().numpy())
    npt.assert_allclose(outs[0], outs[1], rtol=2e-6)
    from oracle import svi
    npt.assert_allclose(outs[0], svi.lda_sstats(C, Th.cpu().numpy(), Bt.cpu().numpy()), rtol=3e-5)


def test_pass_packed_backward_gives_the_same_bits(monkeypatch):
    """Option blr_pk (the register-fed MFMA pass, blr_q = blr_dma = 0): v_pk_fma_f32 is two fused multiply-adds --
    the statistics must not change."""
    import torch
    pk = _context_with(monkeypatch, blr_q=0, blr_dma=0, blr_pk=1)
    scalar = _context_with(monkeypatch, blr_q=0, blr_dma=0, blr_pk=0)
    g = torch.Generator(device=pk.device).manual_seed(5)
    B, D, S = 50_000, 256, 8
    X = torch.randn((B, D), generator=g, device=pk.device)
    y = torch.randn(B, generator=g, device=pk.device)
    W = torch.randn((S, D), generator=g, device=pk.device) / 16
    outs = []
    for c in (pk, scalar):
        Q = torch.zeros(S, dtype=torch.float64, device=c.device)
        G = torch.zeros((S, D), dtype=torch.float64, device=c.device)
        c.call("bsc_blr_data_pass", X, D, y, B, D, W, S, Q, G)
        c.sync()
        outs.append((Q.cpu().numpy(), G.cpu().numpy()))
    npt.assert_array_equal(outs[0][0], outs[1][0])
    npt.assert_array_equal(outs[0][1], outs[1][1])
