"""GPU parity: HIP BLR reparam path (through the C ABI) vs the float64 oracle.

Tolerances (stated, float path): the device streams float32 operands, forms
products and per-lane partial sums in float32, and finishes in float64.
 * data pass Q, G: |dev - oracle| <= 2e-5 * scale_of_sum, where scale_of_sum is
   the Cauchy-Schwarz bound of the summed terms (rtol 1e-5 is the reference's
   own tolerance for contractions, bayesic/tests/test_algebra.py:82);
 * sampler eps / xi (float64 both sides): rtol 1e-12; W equal after float32
   rounding up to 1 ulp;
 * ELBO / gradient finish (float64 both sides, same inputs): rtol 1e-10.
"""
import math

import numpy as np
import pytest
import torch

from oracle import philox, svi

pytestmark = pytest.mark.gpu


def _pass(ctx, X, y, W):
    from bayesic_amd._ffi import ptr
    Xd, yd, Wd = ctx.to_device(X), ctx.to_device(y), ctx.to_device(W)
    S, D = W.shape
    Q = ctx.zeros(S, torch.float64)
    G = ctx.zeros((S, D), torch.float64)
    ctx.call("bsc_blr_data_pass", ptr(Xd), Xd.stride(0) if X.shape[0] else D, ptr(yd),
             X.shape[0], D, ptr(Wd), S, ptr(Q), ptr(G))
    ctx.sync()
    return Q.cpu().numpy(), G.cpu().numpy()


def _check_pass(ctx, X, y, W, tol=2e-5):
    Q, G = _pass(ctx, X, y, W)
    Qr, Gr = svi.blr_data_pass_chunked(X, y, W) if X.shape[0] else \
        (np.zeros(W.shape[0]), np.zeros(W.shape))
    X64 = X.astype(np.float64)
    # scale of the summed terms: sum_n r^2 and sqrt(sum r^2 * sum x_d^2)
    np.testing.assert_allclose(Q, Qr, rtol=tol, atol=tol)
    colnorm = np.sqrt((X64 * X64).sum(axis=0))[None, :]
    bound = np.sqrt(Qr)[:, None] * colnorm
    err = np.abs(G - Gr)
    assert (err <= tol * bound + 1e-12).all(), "max err/bound %g" % (err / (bound + 1e-300)).max()
    return Q, G


@pytest.mark.parametrize("B,D,S", [
    (8, 256, 8),          # exactly one tile
    (64, 256, 8),
    (1000, 256, 8),       # ragged: 1000 = 125 tiles, needs multiple waves
    (1003, 256, 8),       # partial last tile
    (7, 256, 8),          # fewer rows than a tile
    (1, 256, 1),
    (4099, 256, 3),       # S < 8 zero-padded
    (513, 64, 8),         # D < 256: masked lanes
    (2050, 4, 2),         # minimum D
    (777, 252, 5),
    (300, 128, 16),       # S > 8: two sample groups
    (2000, 256, 9),       # D = 256, S > 8: sixteen draws per pass (blr_pass_mx_kernel<., 4>), one used in the second slab
    (4099, 256, 16),      # both slabs full
    (1003, 256, 20),      # a 16-draw pass, then a 4-draw pass
    (40_000, 256, 64),    # four 16-draw passes over several windows
    (100, 32, 64),        # S = 64 (config-5 sample count)
])
def test_data_pass_matches_oracle(ctx, B, D, S):
    rs = np.random.RandomState(B * 7 + D + S)
    X = rs.standard_normal((B, D)).astype(np.float32)
    y = rs.standard_normal(B).astype(np.float32)
    W = (rs.standard_normal((S, D)) / math.sqrt(D)).astype(np.float32)
    _check_pass(ctx, X, y, W)


def test_data_pass_empty_batch(ctx):
    Q, G = _pass(ctx, np.zeros((0, 256), np.float32), np.zeros(0, np.float32),
                 np.ones((8, 256), np.float32))
    assert (Q == 0).all() and (G == 0).all()


def test_data_pass_butterfly_maps_every_row_and_sample(ctx):
    """Exact-integer data: any lane/row/sample mix-up in the transposing
    reduction or the readlane broadcast changes an integer result."""
    B, D, S = 64, 256, 8
    n = np.arange(B)[:, None]
    d = np.arange(D)[None, :]
    X = ((n * 3 + d * 5) % 7 - 3).astype(np.float32)
    W = ((np.arange(S)[:, None] * 11 + d * 2) % 5 - 2).astype(np.float32)
    y = ((np.arange(B) * 13) % 17 - 8).astype(np.float32)
    Q, G = _pass(ctx, X, y, W)
    Qr, Gr = svi.blr_data_pass(X, y, W)
    np.testing.assert_array_equal(Q, Qr)      # all sums are exact small integers
    np.testing.assert_array_equal(G, Gr)


def test_data_pass_is_linear_in_y_and_deterministic(ctx):
    rs = np.random.RandomState(5)
    B, D, S = 5000, 256, 8
    X = rs.standard_normal((B, D)).astype(np.float32)
    W = (rs.standard_normal((S, D)) / 16).astype(np.float32)
    y1 = rs.standard_normal(B).astype(np.float32)
    _, G0 = _pass(ctx, X, np.zeros(B, np.float32), W)
    _, G1 = _pass(ctx, X, y1, W)
    _, G1b = _pass(ctx, X, y1, W)
    np.testing.assert_array_equal(G1, G1b)            # bitwise reproducible
    # G(y) - G(0) = X^T y for every sample
    Xty = X.astype(np.float64).T @ y1.astype(np.float64)
    scale = np.sqrt((X.astype(np.float64) ** 2).sum(0) * (y1.astype(np.float64) ** 2).sum())
    assert (np.abs((G1 - G0) - Xty[None, :]) <= 4e-5 * scale[None, :]).all()


def test_data_pass_respects_leading_dimension(ctx):
    from bayesic_amd._ffi import ptr
    rs = np.random.RandomState(9)
    B, D, ld, S = 333, 64, 96, 8
    buf = rs.standard_normal((B, ld)).astype(np.float32)
    y = rs.standard_normal(B).astype(np.float32)
    W = rs.standard_normal((S, D)).astype(np.float32) / 8
    bd, yd, Wd = ctx.to_device(buf), ctx.to_device(y), ctx.to_device(W)
    Q, G = ctx.zeros(S, torch.float64), ctx.zeros((S, D), torch.float64)
    ctx.call("bsc_blr_data_pass", ptr(bd), ld, ptr(yd), B, D, ptr(Wd), S, ptr(Q), ptr(G))
    ctx.sync()
    Qr, Gr = svi.blr_data_pass(buf[:, :D], y, W)
    np.testing.assert_allclose(Q.cpu().numpy(), Qr, rtol=2e-5)
    np.testing.assert_allclose(G.cpu().numpy(), Gr, rtol=1e-4, atol=1e-4 * np.abs(Gr).max())


def test_data_pass_rejects_bad_shapes(ctx):
    from bayesic_amd._ffi import BayesicHipError, ptr
    X = ctx.zeros((8, 260))
    y, W = ctx.zeros(8), ctx.zeros((8, 260))
    Q, G = ctx.zeros(8, torch.float64), ctx.zeros((8, 260), torch.float64)
    with pytest.raises(BayesicHipError, match="multiple of 4"):
        ctx.call("bsc_blr_data_pass", ptr(X), 260, ptr(y), 8, 260, ptr(W), 8, ptr(Q), ptr(G))
    with pytest.raises(BayesicHipError, match="S="):
        ctx.call("bsc_blr_data_pass", ptr(X), 256, ptr(y), 8, 256, ptr(W), 65, ptr(Q), ptr(G))


def test_philox_normal_matches_oracle(ctx):
    from bayesic_amd._ffi import ptr
    for (S, P, stream, step) in [(8, 257, 0, 0), (64, 1001, 1, 3), (1, 1, 0, 7)]:
        eps = ctx.zeros(S * P, torch.float64)
        ctx.call("bsc_philox_normal", 1234, stream, step, S, P, ptr(eps))
        ctx.sync()
        want = philox.normal_draws(1234, S, P, stream=stream, step=step)
        np.testing.assert_allclose(eps.cpu().numpy().reshape(S, P), want, rtol=1e-12, atol=1e-14)


def test_sampler_matches_oracle(ctx):
    from bayesic_amd._ffi import ptr
    D, S = 256, 8
    lam = svi.blr_init_lam(D) + 0.001 * np.arange(2 * D + 2)
    lamd = ctx.to_device(lam, torch.float64)
    eps, W, xi = ctx.zeros(S * (D + 1), torch.float64), ctx.zeros(S * D), ctx.zeros(S, torch.float64)
    ctx.call("bsc_blr_sample", ptr(lamd), D, S, (5 << 32) + 99, 4, ptr(eps), ptr(W), ptr(xi))
    ctx.sync()
    e, w, x = svi.blr_sample(lam, D, S, (5 << 32) + 99, step=4)
    np.testing.assert_allclose(eps.cpu().numpy().reshape(S, D + 1), e, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(xi.cpu().numpy(), x, rtol=1e-13)
    wd = W.cpu().numpy().reshape(S, D)
    assert (np.abs(wd - w) <= np.spacing(np.abs(w))).all()
    assert (wd == w).mean() > 0.999


def test_elbo_grad_and_adam_match_oracle(ctx):
    from bayesic_amd._ffi import ptr
    rs = np.random.RandomState(11)
    D, S, B, scale = 256, 8, 1000.0, 4.0
    lam = svi.blr_init_lam(D) + 0.01 * rs.standard_normal(2 * D + 2)
    eps = rs.standard_normal((S, D + 1))
    W = (lam[:D][None] + np.exp(lam[D:2 * D])[None] * eps[:, :D]).astype(np.float32)
    xi = lam[2 * D] + math.exp(lam[2 * D + 1]) * eps[:, D]
    Q = rs.uniform(200, 400, S)
    G = rs.standard_normal((S, D)) * 30
    f64 = torch.float64
    d = {k: ctx.to_device(v, f64) for k, v in dict(lam=lam, eps=eps, xi=xi, Q=Q, G=G).items()}
    Wd = ctx.to_device(W)
    elbo, grad = ctx.zeros(1, f64), ctx.zeros(2 * D + 2, f64)
    ctx.call("bsc_blr_elbo_grad", ptr(d["lam"]), ptr(d["eps"]), ptr(Wd), ptr(d["xi"]), ptr(d["Q"]),
             ptr(d["G"]), D, S, B, scale, 1.5, 0.7, ptr(elbo), ptr(grad))
    ctx.sync()
    e_ref, g_ref = svi.blr_elbo_and_grad(lam, eps, W, xi, Q, G, B, scale, 1.5, 0.7)
    np.testing.assert_allclose(elbo.cpu().numpy()[0], e_ref, rtol=1e-10)
    np.testing.assert_allclose(grad.cpu().numpy(), g_ref, rtol=1e-10, atol=1e-10 * np.abs(g_ref).max())
    # Adam ascent, three steps
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    m1d, m2d = ctx.zeros(lam.size, f64), ctx.zeros(lam.size, f64)
    lr = 0.01
    lam_ref = lam.copy()
    for t in (1, 2, 3):
        ctx.call("bsc_adam_ascent", ptr(d["lam"]), ptr(grad), ptr(m1d), ptr(m2d), lam.size, t, lr,
                 0.9, 0.999, 1e-8)
        lam_ref, m1, m2 = svi.adam_ascent(lam_ref, g_ref, m1, m2, t, lr)
    ctx.sync()
    np.testing.assert_allclose(d["lam"].cpu().numpy(), lam_ref, rtol=1e-12, atol=1e-14)


def test_full_update_steps_track_the_oracle(ctx):
    """sample -> pass -> gradient -> Adam for several steps, on cfg-2-shaped data
    (small B); compares the variational parameters after each step."""
    from bayesic_amd.svi.blr import BLRReparamSVI
    X, y, _ = svi.make_cfg2(6000, 256)
    model = BLRReparamSVI(X, y, n_total=60000, n_samples=8, seed=1234, lr=0.01, ctx=ctx)
    lam = svi.blr_init_lam(256)
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    for t in range(1, 6):
        model.step()
        lam, m1, m2, elbo, grad = svi.blr_step(lam, m1, m2, t, X, y, 8, 1234, 60000, 0.01)
        ctx.sync()
        g_dev = model.grad.cpu().numpy()
        np.testing.assert_allclose(model.elbo.item(), elbo, rtol=1e-6)
        assert np.abs(g_dev - grad).max() <= 1e-4 * np.abs(grad).max()
        np.testing.assert_allclose(model.lam.cpu().numpy(), lam, atol=2e-4)


def test_suffstats_normal_and_conjugate_update_cfg1(ctx):
    """BASELINE config 1: Gaussian-Gamma, 10k rows, one latent node.  A full-batch
    rho=1 natural-gradient step must land on the exact posterior."""
    from bayesic_amd._ffi import ptr
    x = svi.make_cfg1()
    f64 = torch.float64
    for arr in (x, x[1:], x[:5], x[3:4], x[:0]):
        xd = ctx.to_device(arr) if arr.size else ctx.zeros(1)
        if arr is x[1:]:
            xd = ctx.to_device(x)[1:]            # 4-byte aligned, not 16
        stats = ctx.zeros(3, f64)
        ctx.call("bsc_suffstats_normal", ptr(xd), arr.size, ptr(stats))
        ctx.sync()
        np.testing.assert_allclose(stats.cpu().numpy(), svi.normal_suffstats(arr), rtol=1e-13)
    xd = ctx.to_device(x)
    stats = ctx.zeros(3, f64)
    ctx.call("bsc_suffstats_normal", ptr(xd), x.size, ptr(stats))
    n, sx, sxx = stats.cpu().numpy()
    eta0 = svi.normal_gamma_to_natural(0.0, 1.0, 1.0, 1.0)
    eta = ctx.to_device(eta0, f64).clone()
    msg = ctx.to_device(np.array([sx, n, n, sxx]), f64)
    eta0d = ctx.to_device(eta0, f64)
    ctx.call("bsc_natgrad_update", ptr(eta), ptr(eta0d), ptr(msg), 4, 1.0, 1.0)
    ctx.sync()
    got = svi.normal_gamma_from_natural(eta.cpu().numpy())
    want = svi.normal_gamma_posterior_closed_form(x, 0.0, 1.0, 1.0, 1.0)
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_cfg2_full_size_properties(ctx):
    """BASELINE config 2 size (1M x 256): size-independent properties.
    Q(y=Xw) = 0-ish and G(W=0) = X^T y computed by torch on the device."""
    from bayesic_amd._ffi import ptr
    g = torch.Generator(device=ctx.device).manual_seed(1)
    B, D, S = 1_000_000, 256, 8
    X = torch.randn((B, D), generator=g, device=ctx.device, dtype=torch.float32)
    y = torch.randn(B, generator=g, device=ctx.device, dtype=torch.float32)
    W = torch.zeros((S, D), device=ctx.device)
    Q, G = ctx.zeros(S, torch.float64), ctx.zeros((S, D), torch.float64)
    ctx.call("bsc_blr_data_pass", ptr(X), D, ptr(y), B, D, ptr(W), S, ptr(Q), ptr(G))
    ctx.sync()
    yty = (y.double() ** 2).sum().item()
    np.testing.assert_allclose(Q.cpu().numpy(), yty, rtol=1e-6)
    Xty = (X.double().T @ y.double()).cpu().numpy()
    colnorm = torch.sqrt((X.double() ** 2).sum(0)).cpu().numpy()
    bound = math.sqrt(yty) * colnorm                      # Cauchy-Schwarz scale of the sum
    assert (np.abs(G.cpu().numpy() - Xty[None, :]) <= 2e-5 * bound[None, :]).all()
    # all eight samples saw the same W=0, so the rows of G must be identical
    assert (G.cpu().numpy() == G.cpu().numpy()[0:1]).all()


@pytest.mark.parametrize("D,S", [(256, 8), (100, 16), (8, 1), (36, 64), (252, 3)])
def test_fused_update_from_stats_matches_oracle(ctx, D, S):
    """The multi-GPU finish: all-reduced [Q|G] in, gradient + Adam + next draws out."""
    from bayesic_amd._ffi import ptr
    rs = np.random.RandomState(D + S)
    B, scale, lr, t = 5000.0, 3.0, 0.02, 4
    lam = svi.blr_init_lam(D) + 0.01 * rs.standard_normal(2 * D + 2)
    eps = rs.standard_normal((S, D + 1))
    W = (lam[:D][None] + np.exp(lam[D:2 * D])[None] * eps[:, :D]).astype(np.float32)
    xi = lam[2 * D] + math.exp(lam[2 * D + 1]) * eps[:, D]
    Q = rs.uniform(200, 400, S)
    G = rs.standard_normal((S, D)) * 30
    m1, m2 = rs.standard_normal(2 * D + 2), rs.uniform(0.5, 2.0, 2 * D + 2)
    f64 = torch.float64
    dev = lambda v: ctx.to_device(v, f64)
    lam_in, lam_out = dev(lam), ctx.zeros(2 * D + 2, f64)
    m1d, m2d, epsd, xid = dev(m1), dev(m2), dev(eps), dev(xi)
    stats = dev(np.concatenate([Q, G.ravel()]))
    Wd = ctx.to_device(W)
    eps_n, W_n, xi_n = ctx.zeros(S * (D + 1), f64), ctx.zeros(S * D), ctx.zeros(S, f64)
    elbo, grad = ctx.zeros(1, f64), ctx.zeros(2 * D + 2, f64)
    ctx.call("bsc_blr_fused_update", ptr(stats), ptr(lam_in), ptr(lam_out), ptr(m1d), ptr(m2d),
             ptr(epsd), ptr(Wd), ptr(xid), D, S, B, scale, 1.5, 0.7, t, lr, 0.9, 0.999, 1e-8,
             77, t, ptr(eps_n), 0, ptr(W_n), ptr(xi_n), ptr(elbo), ptr(grad))
    ctx.sync()
    e_ref, g_ref = svi.blr_elbo_and_grad(lam, eps, W, xi, Q, G, B, scale, 1.5, 0.7)
    lam_ref, m1r, m2r = svi.adam_ascent(lam, g_ref, m1, m2, t, lr)
    np.testing.assert_allclose(elbo.item(), e_ref, rtol=1e-10)
    np.testing.assert_allclose(grad.cpu().numpy(), g_ref, rtol=1e-10, atol=1e-10 * np.abs(g_ref).max())
    np.testing.assert_allclose(lam_out.cpu().numpy(), lam_ref, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(m1d.cpu().numpy(), m1r, rtol=1e-12)
    np.testing.assert_allclose(m2d.cpu().numpy(), m2r, rtol=1e-12)
    np.testing.assert_array_equal(lam_in.cpu().numpy(), lam)          # input untouched
    e_n, w_n, x_n = svi.blr_sample(lam_ref, D, S, 77, step=t)
    np.testing.assert_allclose(eps_n.cpu().numpy().reshape(S, D + 1), e_n, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(xi_n.cpu().numpy(), x_n, rtol=1e-12)
    wd = W_n.cpu().numpy().reshape(S, D)
    assert (np.abs(wd - w_n) <= np.spacing(np.abs(w_n))).all()


def test_fused_and_unfused_update_paths_agree(ctx):
    from bayesic_amd.svi.blr import BLRReparamSVI
    X, y, _ = svi.make_cfg2(3001, 256)
    a = BLRReparamSVI(X, y, n_total=30010, n_samples=8, seed=5, lr=0.01, ctx=ctx, fused=True)
    b = BLRReparamSVI(X, y, n_total=30010, n_samples=8, seed=5, lr=0.01, ctx=ctx, fused=False)
    for _ in range(4):
        a.step()
        b.step()
    ctx.sync()
    np.testing.assert_allclose(a.lam.cpu().numpy(), b.lam.cpu().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(a.elbo.item(), b.elbo.item(), rtol=1e-12)


def test_fused_update_requires_pending_partials(ctx):
    from bayesic_amd._ffi import BayesicHipError, ptr
    f64 = torch.float64
    D, S = 8, 2
    z = lambda n: ctx.zeros(n, f64)
    lam_in, lam_out, m1, m2 = z(2 * D + 2), z(2 * D + 2), z(2 * D + 2), z(2 * D + 2)
    eps, W, xi, elbo, grad = z(S * (D + 1)), ctx.zeros(S * D), z(S), z(1), z(2 * D + 2)
    # a full data pass consumes the slab, so nothing is pending afterwards
    X, y = ctx.zeros((16, D)), ctx.zeros(16)
    Q, G = z(S), z(S * D)
    ctx.call("bsc_blr_data_pass", ptr(X), D, ptr(y), 16, D, ptr(W), S, ptr(Q), ptr(G))
    with pytest.raises(BayesicHipError, match="pending"):
        ctx.call("bsc_blr_fused_update", None, ptr(lam_in), ptr(lam_out), ptr(m1), ptr(m2),
                 ptr(eps), ptr(W), ptr(xi), D, S, 16.0, 1.0, 1.0, 1.0, 1, 0.01, 0.9, 0.999, 1e-8,
                 1, 1, None, 0, None, None, ptr(elbo), ptr(grad))
    with pytest.raises(BayesicHipError, match="differ"):
        ctx.call("bsc_blr_fused_update", ptr(Q), ptr(lam_in), ptr(lam_in), ptr(m1), ptr(m2),
                 ptr(eps), ptr(W), ptr(xi), D, S, 16.0, 1.0, 1.0, 1.0, 1, 0.01, 0.9, 0.999, 1e-8,
                 1, 1, None, 0, None, None, ptr(elbo), ptr(grad))


def test_noise_ring_wraps_and_matches_per_step_sampling(ctx):
    """The driver draws noise 32 steps per launch into a 64-row ring; 70 updates cross
    both a block boundary and the wrap-around.  Also checks bsc_blr_noise against the
    oracle's per-step draws and the eps_next_ready path of the fused finish."""
    from bayesic_amd._ffi import ptr
    from bayesic_amd.svi.blr import BLRReparamSVI
    D, S = 16, 4
    noise = ctx.zeros((5, S * (D + 1)), torch.float64)
    ctx.call("bsc_blr_noise", D, S, 321, 7, 5, noise)
    ctx.sync()
    for k in range(5):
        e, _, _ = svi.blr_sample(svi.blr_init_lam(D), D, S, 321, step=7 + k)
        np.testing.assert_allclose(noise[k].cpu().numpy().reshape(S, D + 1), e, rtol=1e-12, atol=1e-14)
    X, y, _ = svi.make_cfg2(800, D)
    model = BLRReparamSVI(X, y, n_total=8000, n_samples=S, seed=321, lr=0.01, ctx=ctx)
    lam = svi.blr_init_lam(D)
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    for t in range(1, 71):
        model.step()
        lam, m1, m2, elbo, _ = svi.blr_step(lam, m1, m2, t, X, y, S, 321, 8000, 0.01)
        if t in (1, 31, 32, 33, 63, 64, 65, 70):
            ctx.sync()
            np.testing.assert_allclose(model.elbo.item(), elbo, rtol=1e-6)
            np.testing.assert_allclose(model.lam.cpu().numpy(), lam, atol=5e-4)


# ---- sweep order (bsc_blr_data_pass_sweep / _partial_sweep; include/bayesic_hip.h) ----------------
def _pass_sweep(ctx, Xd, yd, Wd, sweep, B=None):
    from bayesic_amd._ffi import ptr
    S, D = Wd.shape
    B = Xd.shape[0] if B is None else B
    Q = ctx.zeros(S, torch.float64)
    G = ctx.zeros((S, D), torch.float64)
    ctx.call("bsc_blr_data_pass_sweep", ptr(Xd), Xd.stride(0) if Xd.shape[0] else D, ptr(yd), B, D,
             ptr(Wd), S, ptr(Q), ptr(G), sweep)
    ctx.sync()
    return Q.cpu().numpy(), G.cpu().numpy()


@pytest.mark.parametrize("B,S", [
    (0, 8),               # nothing to sweep
    (5, 8),               # one partial tile: one window
    (16, 3),
    (4099, 8),            # one window, ragged
    (130_003, 8),         # five windows on 256 CUs, the last one ragged
    (70_000, 20),         # S > 8: three sample groups walk forward, back, forward
])
def test_sweep_orders_agree_with_the_oracle_and_are_deterministic(ctx, B, S):
    """Every sweep order computes the same sums (float32 block partials in another order: the
    oracle tolerance of the streaming pass holds for each), and each order repeats bit for bit."""
    rng = np.random.RandomState(B + S)
    D = 256
    X = rng.standard_normal((B, D)).astype(np.float32)
    y = rng.standard_normal(B).astype(np.float32)
    W = (rng.standard_normal((S, D)) / 16).astype(np.float32)
    Xd, yd, Wd = ctx.to_device(X), ctx.to_device(y), ctx.to_device(W)
    Qr, Gr = svi.blr_data_pass_chunked(X, y, W) if B else (np.zeros(S), np.zeros((S, D)))
    colnorm = np.sqrt((X.astype(np.float64) ** 2).sum(axis=0))[None, :]
    bound = np.sqrt(Qr)[:, None] * colnorm
    for sweep in (0, 1, 2):
        Q, G = _pass_sweep(ctx, Xd, yd, Wd, sweep)
        np.testing.assert_allclose(Q, Qr, rtol=2e-5, atol=2e-5)
        assert (np.abs(G - Gr) <= 2e-5 * bound + 1e-12).all(), "sweep %d" % sweep
        Q2, G2 = _pass_sweep(ctx, Xd, yd, Wd, sweep)
        assert np.array_equal(Q, Q2) and np.array_equal(G, G2), "sweep %d is not deterministic" % sweep
    # streaming forward and keeping forward read the rows in the same order: same bits for the
    # first sample group (the second group of a keeping sweep walks back)
    Q0, G0 = _pass_sweep(ctx, Xd, yd, Wd, 0)
    Q1, G1 = _pass_sweep(ctx, Xd, yd, Wd, 1)
    assert np.array_equal(Q0[:8], Q1[:8]) and np.array_equal(G0[:8], G1[:8])


@pytest.mark.parametrize("keep", ["0", "1", "3", "100"])
def test_sweep_keep_window_counts(keep, monkeypatch):
    """Option blr_keep = windows a keeping sweep reads with the allocating policy: 0 (none), fewer
    than the sweep has, and more than it has all give the streaming pass's bits (forward) and the
    oracle's sums (backward)."""
    from bayesic_amd.device import Context
    ctx = Context(0, options=dict(blr_keep=int(keep)))
    rng = np.random.RandomState(int(keep))
    B, D, S = 100_001, 256, 8
    X = rng.standard_normal((B, D)).astype(np.float32)
    y = rng.standard_normal(B).astype(np.float32)
    W = (rng.standard_normal((S, D)) / 16).astype(np.float32)
    Xd, yd, Wd = ctx.to_device(X), ctx.to_device(y), ctx.to_device(W)
    Q0, G0 = _pass_sweep(ctx, Xd, yd, Wd, 0)
    Q1, G1 = _pass_sweep(ctx, Xd, yd, Wd, 1)
    Q2, G2 = _pass_sweep(ctx, Xd, yd, Wd, 2)
    assert np.array_equal(Q0, Q1) and np.array_equal(G0, G1)
    np.testing.assert_allclose(Q2, Q0, rtol=1e-5)
    np.testing.assert_allclose(G2, G0, rtol=1e-4, atol=1e-3 * np.abs(G0).max())


def test_sweep_rejects_unknown_order(ctx):
    from bayesic_amd._ffi import BayesicHipError
    X = ctx.zeros((16, 256), torch.float32)
    y = ctx.zeros(16, torch.float32)
    W = ctx.zeros((8, 256), torch.float32)
    with pytest.raises(BayesicHipError, match="sweep"):
        _pass_sweep(ctx, X, y, W, 3)


def test_alternating_driver_tracks_the_streaming_driver(ctx):
    """BLRReparamSVI(sweep="alternate") -- the default -- and sweep="stream" run the same updates;
    they differ by the float32 summation order of every second pass only."""
    from bayesic_amd.svi.blr import BLRReparamSVI
    X, y, _ = svi.make_cfg2(50_000, 256)
    a = BLRReparamSVI(X, y, n_samples=8, seed=7, lr=0.01, ctx=ctx)
    b = BLRReparamSVI(X, y, n_samples=8, seed=7, lr=0.01, ctx=ctx, sweep="stream")
    assert a.sweep == "alternate"
    codes = []
    for _ in range(6):
        codes.append(a._sweep_next)
        a.step()
        b.step()
    ctx.sync()
    assert codes == [1, 2, 1, 2, 1, 2]
    np.testing.assert_allclose(a.elbo.item(), b.elbo.item(), rtol=1e-6)
    np.testing.assert_allclose(a.lam.cpu().numpy(), b.lam.cpu().numpy(), atol=1e-5)
    # a freshly swapped-in batch is streamed once, then walked back
    a.set_batch(a.X, a.y)
    assert a._take_sweep() == 0 and a._take_sweep() == 2 and a._take_sweep() == 1


@pytest.mark.parametrize("mx", ["1", "2"])
def test_all_mfma_pass_variant_matches_oracle(mx, monkeypatch):
    """Option blr_mx selects blr_pass_mx_kernel (backward rank-1 updates on v_mfma_f32_4x4x1; 1 = with the
    rotated cached-zone schedule for keeping sweeps, 2 = plain sweep orders).  Not the default (DESIGN.md
    section 12: same speed at the read ceiling) but kept selectable, so it is held to the same tolerance."""
    from bayesic_amd.device import Context
    ctx = Context(0, options=dict(blr_mx=int(mx), blr_keep=2))
    for B, S in [(0, 8), (5, 8), (4099, 3), (130_003, 8), (70_000, 20)]:
        rng = np.random.RandomState(B + S)
        D = 256
        X = rng.standard_normal((B, D)).astype(np.float32)
        y = rng.standard_normal(B).astype(np.float32)
        W = (rng.standard_normal((S, D)) / 16).astype(np.float32)
        Xd, yd, Wd = ctx.to_device(X), ctx.to_device(y), ctx.to_device(W)
        Qr, Gr = svi.blr_data_pass_chunked(X, y, W) if B else (np.zeros(S), np.zeros((S, D)))
        colnorm = np.sqrt((X.astype(np.float64) ** 2).sum(axis=0))[None, :]
        bound = np.sqrt(Qr)[:, None] * colnorm
        for sweep in (0, 1, 2):
            Q, G = _pass_sweep(ctx, Xd, yd, Wd, sweep)
            np.testing.assert_allclose(Q, Qr, rtol=2e-5, atol=2e-5)
            assert (np.abs(G - Gr) <= 2e-5 * bound + 1e-12).all(), "B=%d sweep %d" % (B, sweep)
            Q2, G2 = _pass_sweep(ctx, Xd, yd, Wd, sweep)
            assert np.array_equal(Q, Q2) and np.array_equal(G, G2)


def test_sixteen_draws_per_pass_equals_eight_per_pass(monkeypatch):
    """S > 8 at D = 256 runs sixteen draws per pass by default (option blr_wide = 1); eight per pass
    (blr_wide = 0) reads X twice as often and must agree to the float32 summation order."""
    from bayesic_amd.device import Context
    rng = np.random.RandomState(16)
    B, D, S = 90_001, 256, 24
    X = rng.standard_normal((B, D)).astype(np.float32)
    y = rng.standard_normal(B).astype(np.float32)
    W = (rng.standard_normal((S, D)) / 16).astype(np.float32)
    out = {}
    for wide in ("1", "0"):
        ctx = Context(0, options=dict(blr_wide=int(wide)))
        Xd, yd, Wd = ctx.to_device(X), ctx.to_device(y), ctx.to_device(W)
        out[wide] = [_pass_sweep(ctx, Xd, yd, Wd, sweep) for sweep in (0, 1, 2)]
        again = _pass_sweep(ctx, Xd, yd, Wd, 1)
        assert np.array_equal(again[0], out[wide][1][0]) and np.array_equal(again[1], out[wide][1][1])
    Qr, Gr = svi.blr_data_pass_chunked(X, y, W)
    for (Qa, Ga), (Qb, Gb) in zip(out["1"], out["0"]):
        np.testing.assert_allclose(Qa, Qb, rtol=1e-5)
        np.testing.assert_allclose(Qa, Qr, rtol=2e-5)
        np.testing.assert_allclose(Ga, Gb, rtol=1e-4, atol=2e-4 * np.abs(Gr).max())


def test_pass_count_is_the_librarys_answer(ctx, monkeypatch):
    """bsc_blr_pass_count: launches of the pass kernel per update as bsc_blr_data_pass issues them -- eight draws
    per pass; sixteen while more than eight are left at D = 256 (option blr_wide = 0: always eight).  The driver asks
    the library instead of re-deriving the rule from the environment."""
    import ctypes
    from bayesic_amd.device import Context
    from bayesic_amd.svi.blr import BLRReparamSVI
    y = ctx.zeros(64)

    def count(c, D, S):
        n = ctypes.c_int32(-1)
        c.call("bsc_blr_pass_count", y, D, S, ctypes.byref(n))
        return n.value

    for S, D, want in [(1, 256, 1), (8, 256, 1), (9, 256, 1), (16, 256, 1), (17, 256, 2), (24, 256, 2),
                       (25, 256, 2), (33, 256, 3), (64, 256, 4), (20, 128, 3), (64, 64, 8)]:
        assert count(ctx, D, S) == want, (S, D)
    narrow = Context(0, options=dict(blr_wide=0))
    assert count(narrow, 256, 64) == 8
    X = ctx.zeros((64, 256))
    model = BLRReparamSVI(X, y, n_samples=24, ctx=ctx)
    assert model._passes_per_update() == 2
    # a context created with blr_wide = 0 and a driver that runs later agree: the driver asks the library
    model = BLRReparamSVI(X, y, n_samples=64, ctx=narrow)
    assert model._passes_per_update() == 8


def test_profiling_only_builds_need_an_explicit_second_switch(monkeypatch):
    """blr_mx = 4 / gemm_dbg / bbvi_dbg / blr_q_dbg select kernels with parts deleted (wrong results, for timing):
    bsc_ctx_set_option refuses them unless profiling_builds = 1 was set on the same context first.  The LIBRARY reads
    no environment variable; the Python Context honours BSC_<NAME> only in a process that says BSC_PROFILING_BUILDS=1."""
    from bayesic_amd._ffi import BayesicHipError
    from bayesic_amd.device import Context
    for name, value in (("blr_mx", 4), ("gemm_dbg", 1), ("bbvi_dbg", 3), ("blr_q_dbg", 1)):
        with pytest.raises(BayesicHipError, match="WRONG results"):
            Context(0, options={name: value})
        c = Context(0, options={"profiling_builds": 1, name: value})
        assert c.get_option(name) == value
        c.close()
        # the environment alone selects nothing ...
        monkeypatch.setenv("BSC_" + name.upper(), str(value))
        c = Context(0)
        assert c.get_option(name) == 0
        c.close()
        # ... unless the process opts in
        monkeypatch.setenv("BSC_PROFILING_BUILDS", "1")
        c = Context(0)
        assert c.get_option(name) == value
        c.close()
        monkeypatch.delenv("BSC_PROFILING_BUILDS")
        monkeypatch.delenv("BSC_" + name.upper())
    plain = Context(0)
    with pytest.raises(BayesicHipError, match="unknown option"):
        plain.set_option("no_such_option", 1)
    with pytest.raises(BayesicHipError, match="not accepted"):
        plain.set_option("blr_tile_rows", 5)
    assert plain.get_option("blr_q") == 1


@pytest.mark.parametrize("options", [dict(blr_q=0), dict(blr_q=0, blr_dma=0), dict(blr_q=0, blr_dma=0, blr_pk=0)],
                         ids=["dma", "mfma-pk", "mfma"])
def test_earlier_pass_kernels_stay_selectable_and_correct(options):
    """Round 4 made blr_pass_q_kernel (both contractions on v_mfma_f32_4x4x1) the D = 256, S <= 8 pass; the kernels it
    replaced stay behind the options blr_q / blr_dma / blr_pk for A/B runs and are held to the same tolerance."""
    from bayesic_amd.device import Context
    ctx = Context(0, options=options)
    for B, S in [(0, 8), (7, 8), (1003, 8), (4099, 3), (130_003, 8)]:
        rs = np.random.RandomState(B + S)
        X = rs.standard_normal((B, 256)).astype(np.float32)
        y = rs.standard_normal(B).astype(np.float32)
        W = (rs.standard_normal((S, 256)) / 16).astype(np.float32)
        if B:
            _check_pass(ctx, X, y, W)
        else:
            Q, G = _pass(ctx, X, y, W)
            assert (Q == 0).all() and (G == 0).all()


@pytest.mark.parametrize("options", [dict(blr_q_bias=0), dict(blr_q_bias=130), dict(blr_q_bias=400),
                                     dict(blr_steal=0), dict(blr_steal=60), dict(blr_steal=125), dict(blr_steal=500)],
                         ids=["bias0", "bias130", "bias400", "static", "steal60", "steal125", "steal500"])
def test_pass_schedules_cover_every_tile_once(options):
    """blr_pass_q_kernel's schedule gives the workgroups with an even blockIdx further windows (option blr_q_bias, per
    mille; default 70): static and reproducible whatever the value.  Exact-integer data (W = 0, y in {0, 1},
    X in {-1, 0, 1}): a tile read twice or not at all changes an integer; then random data against the oracle at the
    pass's tolerance."""
    from bayesic_amd.device import Context
    ctx = Context(0, options=options)
    for B in (0, 1, 16, 4099, 130_003, 333_333):
        D, S = 256, 8
        n = np.arange(B)[:, None]
        d = np.arange(D)[None, :]
        X = ((n * 3 + d * 5) % 3 - 1).astype(np.float32)
        W = np.zeros((S, D), np.float32)               # residual = y: G = X^T y and Q = y.y in small integers
        y = ((np.arange(B) * 13) % 17 < 9).astype(np.float32)
        Q, G = _pass(ctx, X, y, W)
        if B:
            np.testing.assert_array_equal(Q, np.full(S, float(y.sum())))
            np.testing.assert_array_equal(G, np.tile(X.astype(np.float64).T @ y.astype(np.float64), (S, 1)))
            rs = np.random.RandomState(B)
            X = rs.standard_normal((B, D)).astype(np.float32)
            y = rs.standard_normal(B).astype(np.float32)
            W = (rs.standard_normal((S, D)) / 16).astype(np.float32)
            _check_pass(ctx, X, y, W)
        else:
            assert (Q == 0).all() and (G == 0).all()


def test_stealing_tail_many_launches_of_changing_size():
    """Option blr_steal: the queue heads must be back at zero after every launch (the last workgroup resets them) --
    many launches of changing size on one context, exact-integer data, some with the chip busy on another stream."""
    import torch
    from bayesic_amd.device import Context
    ctx = Context(0, options=dict(blr_steal=125))
    D, S = 256, 8
    W = np.zeros((S, D), np.float32)
    busy = torch.randn((4096, 4096), device="cuda")
    side = torch.cuda.Stream()
    for i, B in enumerate([70_001, 33_000, 262_144, 9_000, 500_017, 33_000, 131_072, 40_000] * 2):
        n = np.arange(B)[:, None]
        d = np.arange(D)[None, :]
        X = ((n * 7 + d * 5 + i) % 3 - 1).astype(np.float32)
        y = ((np.arange(B) * 11 + i) % 17 < 9).astype(np.float32)
        if i % 3 == 0:
            with torch.cuda.stream(side):
                for _ in range(4):
                    busy = torch.tanh(busy @ busy * 1e-3)
        Q, G = _pass(ctx, X, y, W)
        np.testing.assert_array_equal(Q, np.full(S, float(y.sum())))
        np.testing.assert_array_equal(G, np.tile(X.astype(np.float64).T @ y.astype(np.float64), (S, 1)))
    torch.cuda.synchronize()


# ---- round 4: the finish folded into the pass's tail (FoldArgs; bsc_blr_pass_update, bsc_blr_data_pass) -------------
# Built, measured break-even against the two launches (profiles/r04_fold_timeline.txt), hence option blr_fold = 0 by
# default: these tests switch it on.

@pytest.fixture(scope="module")
def fold_ctx():
    from bayesic_amd.device import Context
    c = Context(0, options=dict(blr_fold=1))
    yield c
    c.close()


def _svi_pair(ctx, B, n_total, seed, lr, S=8):
    from bayesic_amd.svi.blr import BLRReparamSVI
    X, y, _ = svi.make_cfg2(B, 256)
    Xd, yd = ctx.to_device(X), ctx.to_device(y)
    one = BLRReparamSVI(Xd, yd, n_total=n_total, n_samples=S, seed=seed, lr=lr, ctx=ctx)
    two = BLRReparamSVI(Xd, yd, n_total=n_total, n_samples=S, seed=seed, lr=lr, ctx=ctx)
    two.one_launch = False
    assert one.one_launch
    return X, y, one, two


@pytest.mark.parametrize("B", [40_000, 130_003, 1_000])
def test_one_launch_update_equals_the_two_launch_update(fold_ctx, B):
    """bsc_blr_pass_update: the last (D + 7) / 8 + 1 workgroups to finish their rows do the finish kernel's work inside
    the pass launch.  Against the two launches it replaces (pass, then bsc_blr_fused_update from the slab): the same
    parameters, ELBO and gradient -- the float64 sum over the block partials runs in another order (4 waves instead
    of 16), nothing else differs -- and against the oracle's update.  B = 1 000 gives a grid below 66 workgroups: the
    entry point then issues the two launches itself."""
    ctx = fold_ctx
    X, y, one, two = _svi_pair(ctx, B, 10.0 * B, seed=21, lr=0.02)
    lam = svi.blr_init_lam(256)
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    calls = []
    real = ctx.call
    ctx.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        for t in range(1, 6):
            one.step()
            two.step()
            lam, m1, m2, elbo, grad = svi.blr_step(lam, m1, m2, t, X, y, 8, 21, 10.0 * B, 0.02)
            np.testing.assert_allclose(one.lam.cpu().numpy(), two.lam.cpu().numpy(), rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(one.elbo.item(), two.elbo.item(), rtol=1e-12)
            np.testing.assert_allclose(one.grad.cpu().numpy(), two.grad.cpu().numpy(), rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(one.elbo.item(), elbo, rtol=1e-6)
            np.testing.assert_allclose(one.lam.cpu().numpy(), lam, atol=1e-4)
    finally:
        ctx.call = real
    assert calls.count("bsc_blr_pass_update") == 5 and calls.count("bsc_blr_fused_update") == 5
    # run to run: bit for bit
    _, _, again, _ = _svi_pair(ctx, B, 10.0 * B, seed=21, lr=0.02)
    for _ in range(5):
        again.step()
    np.testing.assert_array_equal(again.lam.cpu().numpy(), one.lam.cpu().numpy())
    np.testing.assert_array_equal(again.elbo.cpu().numpy(), one.elbo.cpu().numpy())


def test_folded_finish_under_uneven_load_and_many_launches(fold_ctx):
    """The hand-off inside the launch (write-through partials -> arrival counter -> sc1 reads) with the chip busy with
    something else on another stream, for many consecutive launches of changing size (the counters must come back to
    zero every time): every update equal to the two-launch update's."""
    ctx = fold_ctx
    from bayesic_amd.svi.blr import BLRReparamSVI
    dev = ctx.device
    hog_a = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    hog_b = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(dev)
    engines = []
    for B in (40_000, 77_777, 250_000):
        X, y, one, two = _svi_pair(ctx, B, 5.0 * B, seed=B, lr=0.01)
        engines.append((one, two))
    for rep in range(40):
        with torch.cuda.stream(side):
            for _ in range(4):
                hog_b.copy_(hog_a)              # 256 MiB read + write beside the passes: uneven arrival of the workgroups
        for one, two in engines:
            one.step()
        for one, two in engines:
            two.step()
        if rep % 10 == 9:
            for one, two in engines:
                np.testing.assert_allclose(one.lam.cpu().numpy(), two.lam.cpu().numpy(), rtol=1e-12, atol=1e-14)
                np.testing.assert_allclose(one.elbo.item(), two.elbo.item(), rtol=1e-12)
    side.synchronize()
    ctx.sync()


def test_data_pass_folds_its_float64_reduction(fold_ctx):
    """bsc_blr_data_pass (the N > 1 structure's pass): with option blr_fold the float64 statistics [Q | G] come out of
    the pass launch's tail instead of a second launch.  Against a context with blr_fold = 0 and against the oracle."""
    ctx = fold_ctx
    from bayesic_amd.device import Context
    plain = Context(0, options=dict(blr_fold=0))
    assert ctx.get_option("blr_fold") == 1 and Context(0).get_option("blr_fold") == 0
    for B, S in ((130_003, 8), (40_000, 5), (1_000, 8)):
        rs = np.random.RandomState(B + S)
        X = rs.standard_normal((B, 256)).astype(np.float32)
        y = rs.standard_normal(B).astype(np.float32)
        W = (rs.standard_normal((S, 256)) / 16).astype(np.float32)
        Q, G = _check_pass(ctx, X, y, W)
        Q0, G0 = _pass(plain, X, y, W)
        np.testing.assert_allclose(Q, Q0, rtol=1e-13)
        np.testing.assert_allclose(G, G0, rtol=1e-12, atol=1e-12 * np.abs(G0).max())
        Q2, G2 = _pass(ctx, X, y, W)
        assert np.array_equal(Q, Q2) and np.array_equal(G, G2)
    plain.close()
