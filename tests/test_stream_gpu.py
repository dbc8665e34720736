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


# ---- round-2 fault (gpurun_out/t_s_1.log: "Memory access fault by GPU ... on address 0x64d7fdadc000", a heap
# address, raised at the next blocking H2D after the loader tests).  Both ways malloc's pages could be shown to the
# device are gone; one deterministic regression test per mechanism, each run once. ----------

def test_pageable_source_dropped_right_after_submit(ctx):
    """Mechanism 1 -- a source the caller frees while its copy is queued.  A pageable source is copied by the host
    into the slot's page-locked bounce buffer inside submit, so it may be dropped (and its heap pages reused for
    other objects) the moment submit returns; the batches still arrive intact."""
    import gc
    from bayesic_amd.svi.stream import MiniBatchLoader
    D, rows, steps = 16, 700, 6                   # 44 KB + 2.8 KB per batch: brk-heap allocations, not mmap
    loader = MiniBatchLoader(ctx, rows, D, n_slots=2)
    W = ctx.to_device(np.ones((1, D), np.float32))
    Q, G = ctx.zeros(1, torch.float64), ctx.zeros(D, torch.float64)
    want = []

    def submit(t):
        rs = np.random.RandomState(100 + t)
        X, y = rs.standard_normal((rows, D)).astype(np.float32), rs.standard_normal(rows).astype(np.float32)
        want.append(float(((y.astype(np.float64) - X.astype(np.float64).sum(1)) ** 2).sum()))
        loader.submit(X, y)
        del X, y                                                    # dropped while the copy may still be queued
        gc.collect()
        churn = [np.full(3000, float(t), np.float32) for _ in range(64)]     # the freed pages get new tenants
        del churn

    submit(0)
    for t in range(steps):
        if t + 1 < steps:
            submit(t + 1)
        dX, dy, n = loader.acquire()
        ctx.call("bsc_blr_data_pass", dX, D, dy, n, D, W, 1, Q, G)
        loader.release()
        ctx.sync()
        npt.assert_allclose(Q.cpu().numpy()[0], want[t], rtol=1e-5)
    loader.close()
    ctx.to_device(np.ones(1000, np.float32))                        # the next blocking H2D from fresh heap memory
    ctx.sync()


def test_close_while_a_page_locked_copy_is_queued(ctx):
    """Mechanism 2 -- page-locked memory released while the copy engine may still read it.  close() drains the copy
    stream before anything is freed (bsc_loader_destroy), and bsc_host_free drains the device itself; there is no
    register-in-place entry point left whose un-registration could come too early."""
    from bayesic_amd import _ffi
    from bayesic_amd.svi.stream import MiniBatchLoader
    D, rows = 256, 200_000                                          # 205 MB: the copy takes milliseconds
    loader = MiniBatchLoader(ctx, rows, D, n_slots=2)
    X, y = loader.pinned_empty((rows, D)), loader.pinned_empty((rows,))
    X[...] = 1.0
    y[...] = 2.0
    loader.submit(X, y)
    loader.close()                                                  # copy queued a moment ago; nothing acquired
    del X, y
    assert not hasattr(loader, "pin")
    assert "bsc_host_register" not in _ffi.SIGNATURES and "bsc_host_unregister" not in _ffi.SIGNATURES
    again = MiniBatchLoader(ctx, 1000, 8, n_slots=2)                # the device is healthy: stream another batch
    Xs, ys = np.ones((1000, 8), np.float32), np.arange(1000, dtype=np.float32)
    again.submit(Xs, ys)
    dX, dy, n = again.acquire()
    out = torch.empty(1000, dtype=torch.float32, device=ctx.device)
    ctx.call("bsc_memset", out, 0, 4000)
    ctx.sync()
    import ctypes
    host = np.empty(1000, np.float32)
    _ffi.check(ctx.lib.bsc_d2h(ctx.handle, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dy), 4000), "bsc_d2h")
    npt.assert_array_equal(host, ys)
    again.release()
    again.close()
