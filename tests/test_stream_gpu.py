"""Mini-batch streaming loader (bsc_loader_*): batches pushed from host memory through
two HBM slots must reach the update kernels intact and in order, with the copy of
batch t+1 queued while batch t is being consumed; the update through the loader must
equal the update on a resident copy of the same batch bit for bit."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_batches(n, rows, D, seed=0):
    rs = np.random.RandomState(seed)
    return [(rs.standard_normal((rows, D)).astype(np.float32),
             rs.standard_normal(rows).astype(np.float32)) for _ in range(n)]


def test_batches_arrive_intact_and_in_order(ctx):
    from bayesic_amd.svi.stream import MiniBatchLoader
    D, rows = 64, 3000
    batches = make_batches(5, rows, D)
    batches[3] = (batches[3][0][:1234], batches[3][1][:1234])        # a short batch
    loader = MiniBatchLoader(ctx, rows, D, n_slots=2)
    pinned = []
    for X, y in batches:                                             # page-locked copies (hipHostMalloc)
        Xp, yp = loader.pinned_empty(X.shape), loader.pinned_empty(y.shape)
        Xp[...] = X
        yp[...] = y
        pinned.append((Xp, yp))
    batches = pinned
    W = ctx.to_device(np.ones((1, D), np.float32))
    Q = ctx.zeros(1, torch.float64)
    G = ctx.zeros(D, torch.float64)
    loader.submit(*batches[0])
    for t, (X, y) in enumerate(batches):
        if t + 1 < len(batches):
            loader.submit(*batches[t + 1])
        dX, dy, n = loader.acquire()
        assert n == X.shape[0]
        # one data pass with w = 1: Q = sum (y - x.1)^2, G = sum (y - x.1) x
        ctx.call("bsc_blr_data_pass", dX, D, dy, n, D, W, 1, Q, G)
        loader.release()
        ctx.sync()
        r = y.astype(np.float64) - X.astype(np.float64).sum(1)
        npt.assert_allclose(Q.cpu().numpy()[0], (r ** 2).sum(), rtol=1e-5)
        npt.assert_allclose(G.cpu().numpy(), r @ X.astype(np.float64), rtol=1e-4, atol=1e-2)
    loader.close()


def test_loader_protocol_errors(ctx):
    from bayesic_amd._ffi import BayesicHipError
    from bayesic_amd.svi.stream import MiniBatchLoader
    loader = MiniBatchLoader(ctx, 100, 8, n_slots=2)
    X, y = make_batches(1, 100, 8)[0]
    with pytest.raises(BayesicHipError):
        loader.acquire()                                  # nothing submitted
    loader.submit(X, y)
    loader.submit(X, y)
    with pytest.raises(BayesicHipError):
        loader.submit(X, y)                               # both slots in flight
    loader.acquire()
    with pytest.raises(BayesicHipError):
        loader.acquire()                                  # previous one not released
    loader.release()
    with pytest.raises(BayesicHipError):
        loader.release()                                  # nothing acquired
    with pytest.raises(ValueError):
        loader.submit(X[:, :4], y)                        # wrong width
    with pytest.raises(BayesicHipError):
        loader.submit(np.zeros((200, 8), np.float32), np.zeros(200, np.float32))   # too many rows
    loader.close()
    with pytest.raises(BayesicHipError):
        MiniBatchLoader(ctx, 100, 8, n_slots=1)


def test_streamed_updates_equal_resident_updates(ctx):
    """SVI over a stream of mini-batches: the loader path and a run over resident copies
    of the same batches give identical parameters (same kernels, same inputs)."""
    from bayesic_amd.svi.blr import BLRReparamSVI
    from bayesic_amd.svi.stream import MiniBatchLoader
    D, rows, steps = 32, 4096, 6
    batches = make_batches(steps, rows, D, seed=3)
    resident = [(ctx.to_device(X), ctx.to_device(y)) for X, y in batches]
    a = BLRReparamSVI(resident[0][0], resident[0][1], n_total=steps * rows, n_samples=4, seed=7,
                      lr=1e-2, ctx=ctx)
    b = BLRReparamSVI(resident[0][0], resident[0][1], n_total=steps * rows, n_samples=4, seed=7,
                      lr=1e-2, ctx=ctx)
    loader = MiniBatchLoader(ctx, rows, D, n_slots=2)
    loader.submit(*batches[0])
    for t in range(steps):
        if t + 1 < steps:
            loader.submit(*batches[t + 1])
        a.set_batch(*resident[t])
        a.step()
        b.set_batch(*loader.acquire())
        b.step()
        loader.release()
    ctx.sync()
    npt.assert_array_equal(a.lam.cpu().numpy(), b.lam.cpu().numpy())
    npt.assert_array_equal(a.elbo.cpu().numpy(), b.elbo.cpu().numpy())
    loader.close()
