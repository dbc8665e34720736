"""bsc_softmax_rows: the expectation of a resident Categorical node (responsibilities) and the
bound's log-sum-exp, against float64 numpy."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(1, 1), (7, 3), (1000, 64), (4099, 17), (333, 2), (50, 65), (9, 1000),
                                       (20000, 16), (5, 1024), (0, 8),
                                       # the 16-bytes-per-lane kernel: every lane count, ragged row groups
                                       (100003, 64), (7, 8), (4099, 12), (333, 28), (1001, 100), (50, 256), (13, 132)])
def test_softmax_rows_matches_numpy(ctx, rows, cols):
    rs = np.random.RandomState(rows + cols)
    x = (rs.standard_normal((rows, cols)) * 6.0).astype(np.float32)
    if rows:
        x[0, 0] = 80.0          # a dominant logit and a very negative one
        x[-1, -1] = -90.0
    xd = ctx.to_device(x) if rows else ctx.zeros((1, cols))
    out, lse = ctx.zeros((max(rows, 1), cols)), ctx.zeros(max(rows, 1))
    ctx.call("bsc_softmax_rows", xd, rows, cols, cols, out, cols, lse)
    ctx.sync()
    if rows == 0:
        return
    x64 = x.astype(np.float64)
    m = x64.max(1, keepdims=True)
    w = np.exp(x64 - m)
    # exp2 of a float32 product x * log2(e): relative error ~ |x - max| * 2^-24
    npt.assert_allclose(out.cpu().numpy(), w / w.sum(1, keepdims=True), rtol=3e-5, atol=1e-30)
    npt.assert_allclose(lse.cpu().numpy(), (m + np.log(w.sum(1, keepdims=True)))[:, 0], rtol=1e-6, atol=1e-6)
    npt.assert_allclose(out.cpu().numpy().sum(1), 1.0, rtol=1e-6)


def test_softmax_rows_strided_and_limits(ctx):
    from bayesic_amd._ffi import BayesicHipError
    x = torch.randn((100, 40), device=ctx.device)
    out = torch.zeros((100, 48), device=ctx.device)
    ctx.call("bsc_softmax_rows", x[:, :20], 100, 20, 40, out, 48, 0)        # lse = NULL, padded rows
    ctx.sync()
    npt.assert_allclose(out[:, :20].cpu().numpy(), torch.softmax(x[:, :20].double(), 1).cpu().numpy(), rtol=3e-5)
    assert float(out[:, 20:].abs().max()) == 0.0
    with pytest.raises(BayesicHipError, match="1024"):
        ctx.call("bsc_softmax_rows", x, 1, 2000, 2000, out, 2000, 0)
