"""GPU parity for BASELINE config 5 (BBVI, hierarchical logistic regression) vs
the float64 oracle.

Tolerance: float32 operands, exact-f32 MFMA fma chains (K = 256), fast exp +
log1p in softplus, float32 per-lane partial sums, float64 finish: each ell_s
within 2e-5 * sum_n (|l_ns| + 1); parameter-side results (float64 both sides)
rtol 1e-9."""
import math

import numpy as np
import numpy.testing as npt
import pytest
import torch

from oracle import philox, svi

pytestmark = pytest.mark.gpu


def _loglik(ctx, X, y, g, Wz, Bz):
    D, G, S = X.shape[1], Bz.shape[0], Wz.shape[0]
    Xd = ctx.to_device(X) if X.shape[0] else ctx.zeros((1, D))
    yd = ctx.to_device(y) if X.shape[0] else ctx.zeros(1)
    gd = torch.as_tensor(g, dtype=torch.int32).to(ctx.device) if X.shape[0] else \
        torch.zeros(1, dtype=torch.int32, device=ctx.device)
    Wd, Bd = ctx.to_device(Wz), ctx.to_device(Bz)
    ell = ctx.zeros(S, torch.float64)
    ctx.call("bsc_logreg_bbvi_loglik", Xd, D, yd, gd, X.shape[0], D, G, Wd, Bd, S, ell)
    ctx.sync()
    return ell.cpu().numpy()


@pytest.mark.parametrize("N,D,G", [(32, 256, 7), (1000, 256, 50), (1003, 256, 1000), (5, 256, 3),
                                   (4099, 64, 11), (777, 4, 2), (20000, 252, 100), (0, 256, 4)])
def test_loglik_matches_oracle(ctx, N, D, G):
    rs = np.random.RandomState(N + D + G)
    X = rs.standard_normal((N, D)).astype(np.float32)
    g = rs.randint(G, size=N).astype(np.int32)
    y = (rs.uniform(size=N) < 0.4).astype(np.float32)
    Wz = (rs.standard_normal((64, D)) / math.sqrt(D)).astype(np.float32)
    Bz = rs.standard_normal((G, 64)).astype(np.float32)
    ell = _loglik(ctx, X, y, g, Wz, Bz)
    want = svi.logreg_loglik(X, y, g, Wz, Bz) if N else np.zeros(64)
    L = np.abs(X.astype(np.float64) @ Wz.astype(np.float64).T + Bz.astype(np.float64)[g]) if N else \
        np.zeros((1, 64))
    bound = (L + 1.0).sum(axis=0)
    assert (np.abs(ell - want) <= 2e-5 * bound + 1e-9).all(), np.abs((ell - want) / bound).max()


def test_loglik_operand_layout_with_exact_integers(ctx):
    """Integer X, W with l = 0 for every (row, sample) except through one column per
    sample: any mix-up of k order, sample or row mapping changes which rows count."""
    N, D, G = 96, 256, 4
    X = np.zeros((N, D), np.float32)
    X[np.arange(N), (np.arange(N) * 7) % D] = 1.0            # row n has a single 1 in column c(n)
    Wz = np.zeros((64, D), np.float32)
    for s in range(64):
        Wz[s, (s * 5) % D] = 40.0                            # sample s looks at column 5s
    Bz = np.full((G, 64), -20.0, np.float32)
    g = (np.arange(N) % G).astype(np.int32)
    y = np.ones(N, np.float32)
    ell = _loglik(ctx, X, y, g, Wz, Bz)
    want = svi.logreg_loglik(X, y, g, Wz, Bz)                # rows hit: l=+20 -> ~0 ; else l=-20 -> ~-20
    npt.assert_allclose(ell, want, rtol=1e-6)
    assert len(set(np.round(want, 3))) > 1


def test_sample_and_grad_match_oracle(ctx):
    D, G, S = 24, 9, 64
    P = D + G + 1
    rs = np.random.RandomState(3)
    lam = svi.bbvi_init_lam(P) + 0.05 * rs.standard_normal(2 * P)
    f64 = torch.float64
    lamd = ctx.to_device(lam, f64)
    eps, Wz, Bz, zeta = ctx.zeros(S * P, f64), ctx.zeros(S * D), ctx.zeros(G * S), ctx.zeros(S, f64)
    ctx.call("bsc_bbvi_sample", lamd, D, G, S, 99, 5, eps, Wz, Bz, zeta)
    ctx.sync()
    e_ref, z_ref = svi.bbvi_sample(lam, P, S, 99, step=5)
    npt.assert_allclose(eps.cpu().numpy().reshape(S, P), e_ref, rtol=1e-12, atol=1e-14)
    npt.assert_allclose(Wz.cpu().numpy().reshape(S, D), z_ref[:, :D].astype(np.float32), rtol=1e-6)
    npt.assert_allclose(Bz.cpu().numpy().reshape(G, S), z_ref[:, D:D + G].T.astype(np.float32), rtol=1e-6)
    npt.assert_allclose(zeta.cpu().numpy(), z_ref[:, P - 1], rtol=1e-12)
    ell = rs.uniform(-900, -700, S)
    elld = ctx.to_device(ell, f64)
    elbo, grad, f = ctx.zeros(1, f64), ctx.zeros(2 * P, f64), ctx.zeros(S, f64)
    ctx.call("bsc_bbvi_grad", lamd, eps, elld, D, G, S, 2.5, 1.3, 0.8, elbo, grad, f)
    ctx.sync()
    e_want, g_want, a, f_want = svi.bbvi_elbo_and_grad(lam, e_ref, ell, D, G, 2.5, 1.3, 0.8)
    npt.assert_allclose(f.cpu().numpy(), f_want, rtol=1e-11)
    npt.assert_allclose(elbo.item(), e_want, rtol=1e-11)
    npt.assert_allclose(grad.cpu().numpy(), g_want, rtol=1e-8, atol=1e-9 * np.abs(g_want).max())


def test_bbvi_steps_track_the_oracle(ctx):
    from bayesic_amd.svi.bbvi import LogRegBBVI
    X, y, g, _, _ = svi.make_cfg5(5000, 256, 20)
    D, G, S = 256, 20, 64
    model = LogRegBBVI(X, y, g, G, n_total=50000, n_samples=S, seed=7, lr=0.01, ctx=ctx)
    lam = svi.bbvi_init_lam(D + G + 1)
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    for t in range(1, 5):
        model.step()
        lam, m1, m2, elbo, grad, ell = svi.bbvi_step(lam, m1, m2, t, X, y, g, D, G, S, 7, 50000, 0.01)
        ctx.sync()
        npt.assert_allclose(model.ell.cpu().numpy(), ell, rtol=1e-5)
        npt.assert_allclose(model.elbo.item(), elbo, rtol=1e-5)
        # Adam normalises the step, so parameters agree to ~lr * relative gradient error
        npt.assert_allclose(model.lam.cpu().numpy(), lam, atol=2e-3)


def test_unsupported_shapes_fail_loudly(ctx):
    from bayesic_amd._ffi import BayesicHipError
    X, y = ctx.zeros((8, 256)), ctx.zeros(8)
    g = torch.zeros(8, dtype=torch.int32, device=ctx.device)
    W, Bz, ell = ctx.zeros((129, 256)), ctx.zeros((2, 129)), ctx.zeros(129, torch.float64)
    with pytest.raises(BayesicHipError, match="S <= 128"):
        ctx.call("bsc_logreg_bbvi_loglik", X, 256, y, g, 8, 256, 2, W, Bz, 129, ell)
    with pytest.raises(BayesicHipError, match="D % 4"):
        ctx.call("bsc_logreg_bbvi_loglik", X, 256, y, g, 8, 258, 2, W, Bz, 64, ell)


@pytest.mark.parametrize("S", [1, 8, 16, 17, 32, 48, 64, 100, 128])
@pytest.mark.parametrize("N,D,G", [(1003, 256, 37), (4099, 64, 11), (50, 252, 3)])
def test_loglik_any_sample_count(ctx, S, N, D, G):
    """S Monte-Carlo draws in 16-sample blocks (1, 2, 4 or 8 of them), ragged last block."""
    rs = np.random.RandomState(S * 7 + N)
    X = rs.standard_normal((N, D)).astype(np.float32)
    g = rs.randint(G, size=N).astype(np.int32)
    y = (rs.uniform(size=N) < 0.4).astype(np.float32)
    Wz = (rs.standard_normal((S, D)) / math.sqrt(D)).astype(np.float32)
    Bz = rs.standard_normal((G, S)).astype(np.float32)
    ell = _loglik(ctx, X, y, g, Wz, Bz)
    want = svi.logreg_loglik(X, y, g, Wz, Bz)
    L = np.abs(X.astype(np.float64) @ Wz.astype(np.float64).T + Bz.astype(np.float64)[g])
    bound = (L + 1.0).sum(axis=0)
    assert ell.shape == (S,)
    assert (np.abs(ell - want) <= 2e-5 * bound + 1e-9).all(), np.abs((ell - want) / bound).max()


def test_first_generation_kernel_still_agrees(ctx):
    """option bbvi_kernel = 0 keeps the LDS-staged kernel for in-process A/B runs; same oracle."""
    import os
    from bayesic_amd.device import Context
    old = Context(0, options=dict(bbvi_kernel=0))
    rs = np.random.RandomState(5)
    N, D, G = 3001, 256, 19
    X = rs.standard_normal((N, D)).astype(np.float32)
    g = rs.randint(G, size=N).astype(np.int32)
    y = (rs.uniform(size=N) < 0.4).astype(np.float32)
    Wz = (rs.standard_normal((64, D)) / 16).astype(np.float32)
    Bz = rs.standard_normal((G, 64)).astype(np.float32)
    a, b = _loglik(old, X, y, g, Wz, Bz), _loglik(ctx, X, y, g, Wz, Bz)
    want = svi.logreg_loglik(X, y, g, Wz, Bz)
    npt.assert_allclose(a, want, rtol=2e-6)
    npt.assert_allclose(b, want, rtol=2e-6)
    old.close()


@pytest.mark.parametrize("D,G,S", [(24, 9, 64), (256, 1000, 64), (4, 1, 7), (60, 130, 33)])
def test_fused_update_equals_grad_adam_and_sample(ctx, D, G, S):
    """bsc_bbvi_update (gradient + Adam + the next update's draws in one kernel) against the three
    separate entry points on the same state, and against the oracle's pieces."""
    P = D + G + 1
    rs = np.random.RandomState(D + G + S)
    lam = svi.bbvi_init_lam(P) + 0.05 * rs.standard_normal(2 * P)
    m1, m2 = 0.01 * rs.standard_normal(2 * P), 0.001 * rs.rand(2 * P)
    ell = rs.uniform(-900, -700, S)
    f64 = torch.float64
    t, lr, seed = 7, 0.01, 4242

    def state():
        lamd, m1d, m2d = ctx.to_device(lam, f64), ctx.to_device(m1, f64), ctx.to_device(m2, f64)
        eps, Wz, Bz, zeta = ctx.zeros(S * P, f64), ctx.zeros(S * D), ctx.zeros(G * S), ctx.zeros(S, f64)
        ctx.call("bsc_bbvi_sample", lamd, D, G, S, seed, t - 1, eps, Wz, Bz, zeta)
        return lamd, m1d, m2d, eps, Wz, Bz, zeta, ctx.zeros(1, f64), ctx.zeros(2 * P, f64), ctx.zeros(S, f64)

    elld = ctx.to_device(ell, f64)
    a = state()
    ctx.call("bsc_bbvi_grad", a[0], a[3], elld, D, G, S, 2.5, 1.3, 0.8, a[7], a[8], a[9])
    ctx.call("bsc_adam_ascent", a[0], a[8], a[1], a[2], 2 * P, t, lr, 0.9, 0.999, 1e-8)
    ctx.call("bsc_bbvi_sample", a[0], D, G, S, seed, t, a[3], a[4], a[5], a[6])
    b = state()
    ctx.call("bsc_bbvi_update", b[0], b[3], elld, D, G, S, 2.5, 1.3, 0.8, b[1], b[2], t, lr, 0.9, 0.999, 1e-8,
             seed, t, b[4], b[5], b[6], b[7], b[8], b[9])
    ctx.sync()
    names = ["lam", "m1", "m2", "eps", "Wz", "Bz", "zeta", "elbo", "grad", "f"]
    for name, x, y in zip(names, a, b):
        x, y = x.cpu().numpy(), y.cpu().numpy()
        if name in ("Wz", "Bz"):
            npt.assert_allclose(y, x, rtol=1e-6, atol=1e-7, err_msg=name)
        elif name == "grad":
            npt.assert_allclose(y, x, rtol=1e-9, atol=1e-11 * np.abs(x).max(), err_msg=name)
        else:
            npt.assert_allclose(y, x, rtol=1e-11, atol=1e-13, err_msg=name)
    # the next draws are those of the oracle's sampler at the new parameters
    e_ref, z_ref = svi.bbvi_sample(b[0].cpu().numpy(), P, S, seed, step=t)
    npt.assert_allclose(b[3].cpu().numpy().reshape(S, P), e_ref, rtol=1e-12, atol=1e-14)
    npt.assert_allclose(b[6].cpu().numpy(), z_ref[:, P - 1], rtol=1e-11)
