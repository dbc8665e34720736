"""inference/recognise.py on the CPU: which symbolic log-joints are read as "the data enter only through
c_s * sum_n (y_n - x_n . w_s)^2" (match on the three contractions a Gaussian-linear likelihood expands into),
and which of those have a parameter-sized part of the fused finish kernel's family.  Recognition never
touches data values: the data are given by their shapes."""
import math

import numpy as np
import numpy.testing as npt
import pytest

from bayesic_amd import algebra as A
from bayesic_amd.inference import recognise as R
from bayesic_amd.inference.models import linear_regression_log_joint

N, D, S = 1000, 16, 8
SHAPES = {"X": (N, D), "y": (N,)}


def _want(scale, alpha0, beta0):
    half = 0.5 * (scale * N + D)
    return (-half * math.log(2 * math.pi) + alpha0 * math.log(beta0) - math.lgamma(alpha0), -half - alpha0, scale,
            1.0, beta0)


@pytest.mark.parametrize("scale,alpha0,beta0", [(1.0, 1.0, 1.0), (10.0, 2.5, 0.7), (37.5, 0.3, 4.0)])
def test_config2_written_with_distribution_nodes_is_recognised(scale, alpha0, beta0):
    lj, v = linear_regression_log_joint(scale, alpha0, beta0)
    for latents in ([(v["W"], D), (v["xi"], 1)], [(v["xi"], 1), (v["W"], D)]):
        plan = R.gaussian_linear(lj, latents, SHAPES, S)
        assert plan is not None and (plan.X, plan.y, plan.W) == ("X", "y", "W")
        assert plan.family is not None and plan.family[5] == "xi"
        npt.assert_allclose(plan.family[:5], _want(scale, alpha0, beta0), rtol=1e-9)
    # the surrogate is the oracle's log-joint as a function of (w, xi, Q)
    from oracle import svi
    rng = np.random.RandomState(0)
    w, x, q = rng.standard_normal((3, D)), rng.standard_normal(3), rng.uniform(1, 9, 3)
    got = R._PROBE.evaluate(plan.surrogate, {"W": w, "xi": x[:, None], R.GaussianLinear.Q_NAME: q})
    npt.assert_allclose(got, svi.blr_log_joint(w, x, q, N, scale, alpha0, beta0), rtol=1e-12)


def test_the_same_model_written_by_hand_in_residual_form():
    X, y, W, xi = A.var("X", 2), A.var("y", 1), A.var("W", 2), A.var("xi", 2)
    x = A.sum(xi, axis=1)
    r = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
    e = A.exp(-x)
    lj = (A.sum(r * r, axis=1) * e * (-0.5) - x * (0.5 * N)) * 4.0 \
        + A.sum(W * W, axis=1) * e * (-0.5 * 3.0) - x * (0.5 * D) - x * 2.0 - e * 0.25
    plan = R.gaussian_linear(lj, [(W, D), (xi, 1)], SHAPES, S)
    assert plan is not None and plan.family is not None
    c0, c_xi, s_q, k_w, beta, _ = plan.family
    npt.assert_allclose([c0, c_xi, s_q, k_w, beta], [0.0, -0.5 * (4.0 * N + D) - 2.0, 4.0, 3.0, 0.25], rtol=1e-9,
                        atol=1e-9)


def test_known_noise_variance_has_the_data_term_but_not_the_family():
    X, y, W = A.var("X", 2), A.var("y", 1), A.var("W", 2)
    r = A.dimshuffle(y, "x", 0) - A.dot(W, X.T)
    lj = A.sum(r * r, axis=1) * (-0.5 / 0.25) + A.sum(W * W, axis=1) * (-0.5)
    plan = R.gaussian_linear(lj, [(W, D)], SHAPES, S)
    assert plan is not None and plan.family is None
    npt.assert_allclose(R._PROBE.evaluate(plan.coefficient, {}), -2.0)


def test_what_is_not_a_gaussian_linear_data_term_is_left_alone():
    X, y, W, xi = A.var("X", 2), A.var("y", 1), A.var("W", 2), A.var("xi", 2)
    logits = A.dot(W, X.T)
    yb = A.dimshuffle(y, "x", 0)
    # a Bernoulli-logit likelihood: the data enter through softplus(logits)
    bern = A.sum(yb * logits - A.log(1.0 + A.exp(logits)), axis=1) + A.sum(W * W, axis=1) * (-0.5)
    assert R.gaussian_linear(bern, [(W, D)], SHAPES, S) is None
    # the cross term with the wrong weight: not a multiple of the squared residual
    r2 = A.sum(yb * yb, axis=1) - A.sum(yb * logits, axis=1) * 1.5 + A.sum(logits * logits, axis=1)
    assert R.gaussian_linear(r2 * (-0.5) + A.sum(W * W, axis=1) * (-0.5), [(W, D)], SHAPES, S) is None
    # a noise variance per datum (a data-sized coefficient) is not the pass's statistic either
    V = A.var("V", 2)          # [S, N] latent-by-datum variances
    het = A.sum((yb - logits) * (yb - logits) * A.exp(-V), axis=1) * (-0.5) + A.sum(W * W, axis=1) * (-0.5)
    assert R.gaussian_linear(het, [(W, D), (V, N)], SHAPES, S) is None
    # a positive coefficient would be a log-joint that grows with the residual
    grow = A.sum((yb - logits) * (yb - logits), axis=1) * 0.5
    assert R.gaussian_linear(grow, [(W, D)], SHAPES, S) is None


def test_broadcast_axes_of_opaque_nodes_are_counted_once_per_extent():
    """log(v[:, 'x']) + M[S, N] summed over the rows counts the log N times -- the value semantics of a
    broadcast axis -- also when the broadcast sits inside an element-wise node the einsum form cannot see
    into (conjugacy.expand_terms / _carried_axes)."""
    from bayesic_amd.inference.conjugacy import expand_terms
    v, M = A.var("v", 1), A.var("M", 2)
    expr = A.sum(A.log(A.dimshuffle(v, 0, "x")) + M, axis=1)
    shapes = {"v": (3,), "M": (3, 5)}
    terms = [R._without_shapes(t, shapes) for t in expand_terms(expr)]
    vv, MM = np.array([1.5, 2.0, 0.3]), np.arange(15.0).reshape(3, 5)
    total = sum(np.asarray(R._PROBE.evaluate(t, {"v": vv, "M": MM})) for t in terms)
    npt.assert_allclose(total, 5 * np.log(vv) + MM.sum(axis=1), rtol=1e-14)


def test_parameter_probe_refuses_data_sized_operands():
    from bayesic_amd.inference._param_backend import MAX_ELEMENTS, ParameterBackend, ShapeOnly
    P = ParameterBackend()
    with pytest.raises(ValueError, match="data-sized"):
        P.from_host(np.zeros(MAX_ELEMENTS + 1), "float64", 1)
    with pytest.raises(ValueError, match="data input"):
        P.mul(ShapeOnly((10, 3)), np.ones(3))
    from bayesic_amd.algebra.backend import resolve_backend
    import bayesic_amd.algebra.backend as B
    assert not isinstance(B._default, ParameterBackend)


# ---- recognise.diagonal_mixture: config 3's update rules, read off the DERIVED messages -----------------

def _mixture_vars():
    v = lambda name, nd: A.var(name, nd, "float64")
    return v("X", 2), v("Z", 2), v("pi", 1), (v("TM", 2), v("TM2", 2), v("LT", 2), v("T", 2))


def test_the_symbolic_mixture_is_recognised_with_its_prior():
    from bayesic_amd.inference.mixture import diagonal_mixture_log_joint
    from oracle import svi
    K, D, scale = 5, 3, 10.0
    X, Z, pi, ng = _mixture_vars()
    prior = (1.5, 0.2, 0.05, 2.0, 0.7)                    # alpha0, m0, kappa0, a0, b0
    lj = diagonal_mixture_log_joint(X, Z, pi, *ng, scale, *prior)
    eta0 = R.diagonal_mixture(lj, Z, pi, ng, "X", K, D, scale)
    npt.assert_allclose(eta0, svi.mog_prior_eta(K, D, *prior), rtol=1e-10, atol=1e-12)
    # the data terms written with another scale than the engine is told: not the fused update
    assert R.diagonal_mixture(lj, Z, pi, ng, "X", K, D, 3.0) is None


def test_other_mixtures_are_not_mistaken_for_it():
    from bayesic_amd.inference.mixture import diagonal_mixture_log_joint
    K, D, scale = 4, 3, 2.0
    X, Z, pi, ng = _mixture_vars()
    TM, TM2, LT, T = ng
    row = lambda v: A.dimshuffle(v, "x", 0)
    base = diagonal_mixture_log_joint(X, Z, pi, *ng, scale, 1.0, 0.0, 0.01, 1.0, 1.0)
    # an extra data-dependent term in the assignments' logits (a per-row offset by component: a different model)
    odd = base + A.sum(Z * A.dot(X * X * X, TM.T)) * 0.01
    assert R.diagonal_mixture(odd, Z, pi, ng, "X", K, D, scale) is None
    # the second moments entering with the wrong weight
    lik = A.sum(Z * A.dot(X, TM.T)) + A.sum(Z * A.dot(X * X, T.T)) * (-0.25) \
        + A.sum(Z * row(A.sum(LT, axis=1))) * 0.5 + A.sum(Z * row(A.sum(TM2, axis=1))) * (-0.5)
    wrong = (lik + A.sum(Z * row(A.log(pi)))) * scale + A.sum(A.log(pi)) * 0.5 + A.sum(LT) * 0.5 + A.sum(T) * (-1.0) \
        + A.sum(TM2) * (-0.005)
    assert R.diagonal_mixture(wrong, Z, pi, ng, "X", K, D, scale) is None


# ---- recognise.logistic_hierarchy: config 5 in any parameterisation ---------------------------------------

def test_hierarchical_logistic_regression_is_recognised_with_its_hyperparameters():
    N5, D5, G5, S5 = 3000, 8, 5, 16
    scale, a0, b0 = 10.0, 1.5, 0.7
    Xv, yv, Gm = A.var("X", 2), A.var("y", 1), A.var("Gm", 2)
    W, Bg, Z = A.var("W", 2), A.var("Bg", 2), A.var("Z", 2)
    L = A.dot(Xv, W.T) + A.dot(Gm, Bg.T)
    loglik = A.sum(A.dimshuffle(yv, 0, "x") * L - A.log(1 + A.exp(L)), axis=0)
    zeta = A.sum(Z, axis=1)
    tau = A.exp(zeta)
    shapes = {"X": (N5, D5), "y": (N5,), "Gm": (N5, G5)}
    latents = [(W, D5), (Bg, G5), (Z, 1)]
    prior = A.sum(W * W, axis=1) * (-0.5) + zeta * (0.5 * G5) - 0.5 * (tau * A.sum(Bg * Bg, axis=1)) + a0 * zeta - b0 * tau
    plan = R.logistic_hierarchy(scale * loglik + prior, latents, shapes, S5)            # constants dropped
    assert plan is not None and (plan.X, plan.y, plan.onehot, plan.W, plan.B, plan.zeta) == ("X", "y", "Gm", "W", "Bg", "Z")
    npt.assert_allclose([plan.scale, plan.a0, plan.b0], [scale, a0, b0], rtol=1e-9)
    want_offset = 0.5 * (D5 + G5) * math.log(2 * math.pi) - (a0 * math.log(b0) - math.lgamma(a0))
    npt.assert_allclose(plan.offset, want_offset, rtol=1e-9)
    # a probit-like link, or a group-level variance that does not scale the intercepts: other models
    other = A.sum(A.dimshuffle(yv, 0, "x") * L - A.log(1 + A.exp(L * 2.0)), axis=0) * scale + prior
    assert R.logistic_hierarchy(other, latents, shapes, S5) is None
    flat = scale * loglik + A.sum(W * W, axis=1) * (-0.5) + A.sum(Bg * Bg, axis=1) * (-0.5) + a0 * zeta - b0 * tau
    assert R.logistic_hierarchy(flat, latents, shapes, S5) is None


# ---- recognition is observable: every "no" comes with its reason, and a bug is not a "no" ------------------------------

def test_every_recogniser_says_why_it_declined():
    X, y, W, xi = A.var("X", 2), A.var("y", 1), A.var("W", 2), A.var("xi", 2)
    logits = A.dot(W, X.T)
    yb = A.dimshuffle(y, "x", 0)
    # one term away from config 2: the cross term with the wrong weight
    why = []
    r2 = A.sum(yb * yb, axis=1) - A.sum(yb * logits, axis=1) * 1.5 + A.sum(logits * logits, axis=1)
    assert R.gaussian_linear(r2 * (-0.5) + A.sum(W * W, axis=1) * (-0.5), [(W, D)], SHAPES, S, why=why) is None
    assert "ratio -2 : 1 : 1" in why[-1]
    # a Bernoulli-logit likelihood: a data term that is none of the three contractions
    why = []
    bern = A.sum(yb * logits - A.log(1.0 + A.exp(logits)), axis=1) + A.sum(W * W, axis=1) * (-0.5)
    assert R.gaussian_linear(bern, [(W, D)], SHAPES, S, why=why) is None
    assert "none of sum_nd y X W" in why[-1]
    # no data at all
    why = []
    assert R.gaussian_linear(A.sum(W * W, axis=1) * (-0.5), [(W, D)], SHAPES, S, why=why) is None
    assert "no term of the log-joint mentions a data input" in why[-1]

    from bayesic_amd.inference.mixture import diagonal_mixture_log_joint
    K, Dm, scale = 4, 3, 2.0
    Xm, Z, pi, ng = _mixture_vars()
    lj = diagonal_mixture_log_joint(Xm, Z, pi, *ng, scale, 1.0, 0.0, 0.01, 1.0, 1.0)
    why = []
    assert R.diagonal_mixture(lj, Z, pi, ng, "X", K, Dm, 3.0, why=why) is None        # told another scale
    assert "fixed prior" in why[-1] or "logits" in why[-1]
    why = []
    odd = lj + A.sum(Z * A.dot(Xm * Xm * Xm, ng[0].T)) * 0.01
    assert R.diagonal_mixture(odd, Z, pi, ng, "X", K, Dm, scale, why=why) is None
    assert why and ("logits" in why[-1] or "prior" in why[-1] or "deriving" in why[-1])

    N5, D5, G5, S5 = 300, 8, 5, 4
    Xv, yv, Gm = A.var("X", 2), A.var("y", 1), A.var("Gm", 2)
    Wl, Bg, Zl = A.var("W", 2), A.var("Bg", 2), A.var("Z", 2)
    L = A.dot(Xv, Wl.T) + A.dot(Gm, Bg.T)
    zeta = A.sum(Zl, axis=1)
    tau = A.exp(zeta)
    shapes = {"X": (N5, D5), "y": (N5,), "Gm": (N5, G5)}
    latents = [(Wl, D5), (Bg, G5), (Zl, 1)]
    prior = A.sum(Wl * Wl, axis=1) * (-0.5) + zeta * (0.5 * G5) - 0.5 * (tau * A.sum(Bg * Bg, axis=1)) + 1.5 * zeta - 0.7 * tau
    why = []
    other = A.sum(A.dimshuffle(yv, 0, "x") * L - A.log(1 + A.exp(L * 2.0)), axis=0) * 10.0 + prior      # another link
    assert R.logistic_hierarchy(other, latents, shapes, S5, why=why) is None
    assert "softplus" in why[-1]
    why = []
    assert R.logistic_hierarchy(prior, latents[:2], shapes, S5, why=why) is None
    assert "three latent blocks" in why[-1]


def test_a_bug_inside_a_recogniser_is_not_a_verdict_on_the_model(monkeypatch):
    """recognise.NOT_THIS_MODEL lists what "not this model" may look like as an exception; anything else propagates
    out of the recogniser, and ``guarded_route`` turns it into a RuntimeWarning + reason (route="auto") or lets it
    through (route="fused")."""
    lj, v = linear_regression_log_joint(1.0, 1.0, 1.0)
    latents = [(v["W"], D), (v["xi"], 1)]

    def broken(*a, **k):
        raise AttributeError("'NoneType' object has no attribute 'factors_and_indices'")
    monkeypatch.setattr(R, "_coefficient", broken)
    with pytest.raises(AttributeError):
        R.gaussian_linear(lj, latents, SHAPES, S)

    def try_route():
        R.gaussian_linear(lj, latents, SHAPES, S)
        return None
    with pytest.warns(RuntimeWarning, match="recognition FAILED with AttributeError"):
        reason = R.guarded_route(try_route, strict=False)
    assert "a bug, not a verdict on the model" in reason
    with pytest.raises(AttributeError):
        R.guarded_route(try_route, strict=True)
    # a declining recogniser is no warning
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert R.guarded_route(lambda: "not this model", strict=False) == "not this model"
