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
