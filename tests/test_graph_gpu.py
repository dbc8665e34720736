"""hipGraph capture of a launch sequence through the C ABI (bsc_capture_begin / bsc_capture_end /
bsc_graph_launch) and its use by the executor (DeviceBackend.graph_call, ReparamVI(graph=True))."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

from bayesic_amd import _ffi
from bayesic_amd import algebra as A
from bayesic_amd.algebra.device_backend import _i64

pytestmark = pytest.mark.gpu


@pytest.fixture()
def sctx():
    """A context on a stream of its own (the null stream cannot be captured)."""
    from bayesic_amd.device import Context
    prev = torch.cuda.current_stream()
    c = Context(0)
    c.set_stream(torch.cuda.Stream(c.device))
    yield c
    c.sync()
    torch.cuda.set_stream(prev)
    c.close()


def test_capture_replays_a_launch_sequence_on_fresh_inputs(sctx):
    ctx = sctx
    n, d = 5000, 64
    g = torch.Generator(device=ctx.device).manual_seed(1)
    X = torch.randn((n, d), generator=g, device=ctx.device)
    W = torch.randn((d, 32), generator=g, device=ctx.device)
    P = torch.empty((n, 32), device=ctx.device)
    tot = torch.empty(32, device=ctx.device)

    def sequence():
        ctx.call("bsc_gemm_strided_batched", 0, 1, n, 32, d, X, 0, d, 1, W, 0, 32, 1, P, 0, 32, 1)
        ctx.call("bsc_sum", 0, 1, _i64([32]), _i64([1]), 1, _i64([n]), _i64([32]), P, tot)

    sequence()                                   # eager once: workspaces reach their size
    ctx.sync()
    first = tot.cpu().numpy().copy()
    ctx.capture_begin()
    sequence()
    graph = ctx.capture_end(keep=[X, W, P, tot])
    tot.zero_()
    graph.launch()
    ctx.sync()
    npt.assert_array_equal(tot.cpu().numpy(), first)            # the same launches: the same bits
    X.copy_(torch.randn((n, d), generator=g, device=ctx.device))   # new contents, same address
    graph.launch()
    ctx.sync()
    want = (X.double() @ W.double()).sum(0).cpu().numpy()
    npt.assert_allclose(tot.cpu().numpy(), want, rtol=2e-5, atol=2e-3)
    # what cannot be captured fails loudly and leaves the stream usable
    ctx.capture_begin()
    with pytest.raises(_ffi.BayesicHipError):
        ctx.sync()
    ctx.capture_end()
    sequence()
    ctx.sync()


def test_capture_needs_a_stream_of_its_own(ctx):
    if ctx.can_capture:
        pytest.skip("the shared test context is not on the null stream")
    with pytest.raises(_ffi.BayesicHipError):
        ctx.capture_begin()


def test_reparam_engine_with_its_walk_recorded_as_a_graph(sctx):
    """ReparamVI(graph=True): the same draws, the same launches in the same order -> the same
    parameters, bit for bit, as the eager engine over many steps (two eager, one recorded, the rest
    replayed)."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference import ReparamVI
    rs = np.random.RandomState(2)
    N, D, S = 20000, 24, 8
    Xs = rs.standard_normal((N, D)).astype(np.float32)
    w_true = rs.standard_normal(D) / 4
    ys = (Xs @ w_true + 0.5 * rs.standard_normal(N)).astype(np.float32)
    X, y, W = A.var("X", 2), A.var("y", 1), A.var("W", 2)
    r = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
    lj = A.sum(r * r, axis=1) * (-0.5 / 0.25) + A.sum(W * W, axis=1) * (-0.5)
    # (route="general": this test is about the executor's walk recorded as a graph; the default route would
    # recognise the Gaussian-linear data term and run it as one fused pass -- tests/test_plugin_route_gpu.py)
    engines = [ReparamVI(lj, [(W, D)], dict(X=Xs, y=ys), n_samples=S, seed=5, backend=DeviceBackend(sctx),
                         lr=0.02, graph=graph, route="general") for graph in (False, True)]
    for step in range(12):
        a, b = engines[0].step(), engines[1].step()
        assert a == b, (step, a, b)
        npt.assert_array_equal(engines[0].lam, engines[1].lam)
    entry = engines[1].backend._graphs[("reparam", id(engines[1]))]
    assert entry["graph"] is not None and not entry["dead"]         # it really was recorded
    # new data: new buffers, the recorded graph is dropped and a new one made
    Xs2 = Xs[::-1].copy()
    for e in engines:
        e.set_data(X=Xs2)
    for step in range(5):
        a, b = engines[0].step(), engines[1].step()
        assert a == b
    assert np.isfinite(engines[1].lam).all()


def test_compiled_expression_with_its_launches_recorded(sctx):
    """expr.compile(backend, graph=True).device_fn: eager twice, recorded on the third call, replayed
    afterwards on refreshed inputs; a different buffer or a different scalar input is a new recording."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    be = DeviceBackend(sctx)
    X, w, s = A.var("X", 2), A.var("w", 1), A.var("s", 0)
    expr = A.sum(A.exp(A.dot(X, w) * s) * A.dot(X, w)) + A.sum(X * X) * 0.5
    f = expr.compile(be, graph=True)
    plain = expr.compile(DeviceBackend(sctx))
    g = torch.Generator(device=sctx.device).manual_seed(3)
    Xd = torch.randn((30000, 48), generator=g, device=sctx.device) * 0.1
    wd = torch.randn(48, generator=g, device=sctx.device)
    sv = be.from_host(np.float32(0.3), "float32", 0)
    for call in range(6):
        Xd.copy_(torch.randn((30000, 48), generator=g, device=sctx.device) * 0.1)
        got = be.to_host(f.device_fn(X=Xd, w=wd, s=sv))
        want = plain.device_fn(X=Xd, w=wd, s=sv)
        npt.assert_array_equal(got, plain.backend.to_host(want))
    entries = [k for k in be._graphs if k[0] == "compile"]
    assert len(entries) == 1 and be._graphs[entries[0]]["graph"] is not None
    # another scalar: its value is a kernel argument, so a recording of its own
    s2 = be.from_host(np.float32(0.1), "float32", 0)
    for call in range(4):
        got = be.to_host(f.device_fn(X=Xd, w=wd, s=s2))
        npt.assert_array_equal(got, plain.backend.to_host(plain.device_fn(X=Xd, w=wd, s=s2)))
    assert len([k for k in be._graphs if k[0] == "compile"]) == 2
    x64 = Xd.double().cpu().numpy(); w64 = wd.double().cpu().numpy()
    ref = (np.exp(x64 @ w64 * 0.1) * (x64 @ w64)).sum() + 0.5 * (x64 * x64).sum()
    npt.assert_allclose(got, ref, rtol=2e-5)


def test_score_function_engine_with_its_evaluation_recorded(sctx):
    """ScoreFunctionVI(graph=True) against the eager engine: the same parameters, bit for bit."""
    from bayesic_amd.algebra.device_backend import DeviceBackend
    from bayesic_amd.inference import ScoreFunctionVI
    rs = np.random.RandomState(4)
    N, D, S = 5000, 6, 32
    Xs = rs.standard_normal((N, D)).astype(np.float32)
    ys = (Xs @ (rs.standard_normal(D) / 3) + 0.5 * rs.standard_normal(N)).astype(np.float32)
    X, y, W = A.var("X", 2), A.var("y", 1), A.var("W", 2)
    r = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
    lj = A.sum(r * r, axis=1) * (-0.5 / 0.25) + A.sum(W * W, axis=1) * (-0.5)
    engines = [ScoreFunctionVI(lj, [(W, D)], dict(X=Xs, y=ys), n_samples=S, seed=3, backend=DeviceBackend(sctx),
                               lr=0.01, graph=graph) for graph in (False, True)]
    for step in range(8):
        a, b = engines[0].step(), engines[1].step()
        assert a == b
        npt.assert_array_equal(engines[0].lam, engines[1].lam)
    assert any(e["graph"] is not None for e in engines[1].backend._graphs.values())
