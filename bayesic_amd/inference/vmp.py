"""Mean-field variational message passing synthesised from a symbolic log-joint.

For every latent node the coefficients of its sufficient statistics are read off
the log-joint once (``conjugate_coefficients``); an update of q(z) is then

    eta_j(z)  <-  E_{q(others)}[ c_j ]

(README.md:36: "a traditional variational message passing update (equivalently a
unit-step natural gradient update)"), with rho < 1 giving the damped / stochastic
natural-gradient step of README.md:69-79.  The expectation is taken by BINDING:
c_j is multilinear in the other latents' statistics t_k(w) (that is what conjugacy
of the whole blanket means), so evaluating it with every sub-expression t_k(w)
replaced by the number E_q[t_k(w)] is exact.  The data-sized parts of c_j (sums over
observations) run through the executor -- on the MI355X backend as fused map-reduce
launches over the resident data.

Model-writing rule (the usual VMP one): a latent must enter the log-joint only
through the statistic expressions its node declares, e.g. ``mu ** 2`` and not
``mu * mu`` for a Normal node -- the product form is bilinear in mu and binding
E[mu] twice would give E[mu]^2.
"""
import math

import numpy as np

from .. import algebra as A
from ..distribution.special import digamma  # noqa: F401  (documented dependency of GammaNode)
from .conjugacy import conjugate_coefficients


class LatentNode(object):
    """A latent variable with an exponential-family variational distribution.

    ``resident`` nodes keep their natural parameters and expectations as BACKEND values (device
    tensors on the MI355X backend): what a data-sized local latent -- the N x K assignments of a
    mixture -- needs, so that no update moves it across PCIe.  Parameter-sized global nodes stay on
    the host in float64 (their transfers are a few KB per message)."""
    resident = False

    def __init__(self, variable):
        self.var = variable
        self.eta = None

    @property
    def statistics(self):
        raise NotImplementedError

    def expectations(self):
        """E_q[t_j(z)] for the current natural parameters, as numpy arrays."""
        raise NotImplementedError

    def entropy(self):
        """H[q] (a float), for the evidence lower bound."""
        raise NotImplementedError


class NormalNode(LatentNode):
    """q(z) = N(mean, variance), element-wise over z's shape: t = (z, z^2),
    eta = (mean / variance, -1 / (2 variance))."""

    def __init__(self, variable, mean=0.0, variance=1.0):
        LatentNode.__init__(self, variable)
        self.set(mean, variance)

    @property
    def statistics(self):
        return (self.var, self.var ** 2)

    def set(self, mean, variance):
        mean, variance = np.asarray(mean, np.float64), np.asarray(variance, np.float64)
        self.eta = [mean / variance, -0.5 / variance]

    @property
    def variance(self):
        return -0.5 / self.eta[1]

    @property
    def mean(self):
        return self.eta[0] * self.variance

    def expectations(self):
        return [self.mean, self.mean ** 2 + self.variance]

    def entropy(self):
        return float(np.sum(0.5 * np.log(2.0 * math.pi * math.e * self.variance)))


class GammaNode(LatentNode):
    """q(z) = Gamma(shape a, rate b): t = (log z, z), eta = (a - 1, -b)."""

    def __init__(self, variable, shape=1.0, rate=1.0):
        LatentNode.__init__(self, variable)
        self.eta = [np.asarray(shape, np.float64) - 1.0, -np.asarray(rate, np.float64)]

    @property
    def statistics(self):
        return (A.log(self.var), self.var)

    @property
    def shape(self):
        return self.eta[0] + 1.0

    @property
    def rate(self):
        return -self.eta[1]

    def expectations(self):
        from scipy.special import digamma as psi       # parameter-sized, host side
        return [psi(self.shape) - np.log(self.rate), self.shape / self.rate]

    def entropy(self):
        from scipy.special import digamma as psi, gammaln
        a, b = self.shape, self.rate
        return float(np.sum(a - np.log(b) + gammaln(a) + (1.0 - a) * psi(a)))


class InverseGammaNode(LatentNode):
    """q(z) = InverseGamma(shape a, scale b): t = (log z, 1 / z), eta = (-(a + 1), -b) -- a
    variance (config 2's sigma^2 ~ InvGamma(1, 1)).  Write the reciprocal as ``z ** -1``."""

    def __init__(self, variable, shape=1.0, scale=1.0):
        LatentNode.__init__(self, variable)
        self.eta = [-(np.asarray(shape, np.float64) + 1.0), -np.asarray(scale, np.float64)]

    @property
    def statistics(self):
        return (A.log(self.var), self.var ** -1)

    @property
    def shape(self):
        return -self.eta[0] - 1.0

    @property
    def scale(self):
        return -self.eta[1]

    def expectations(self):
        from scipy.special import digamma as psi       # parameter-sized, host side
        return [np.log(self.scale) - psi(self.shape), self.shape / self.scale]

    def entropy(self):
        from scipy.special import digamma as psi, gammaln
        a, b = self.shape, self.scale
        return float(np.sum(a + np.log(b) + gammaln(a) - (1.0 + a) * psi(a)))


class MVNormalNode(LatentNode):
    """q(w) = N(m, Sigma) over the LAST axis of w (leading axes index independent vectors, e.g.
    mixture components): t = (w, w w^T), eta = (Lambda m, -Lambda / 2).

    The second moment enters the log-joint through a variable of its own (``second_moment``, of
    shape w.shape + (D,)): the front end flattens ``outer(w, w)`` into the surrounding einsum
    (bayesic/algebra.py:314-508), which would leave no sub-expression to bind E[w w^T] to and
    would bind E[w] twice instead.  Write ``sum(dot(X.T, X) * W2)`` for w^T X^T X w."""

    def __init__(self, variable, second_moment, mean, covariance):
        LatentNode.__init__(self, variable)
        self.second = second_moment
        mean = np.asarray(mean, np.float64)
        precision = np.linalg.inv(np.asarray(covariance, np.float64))
        self.eta = [np.einsum("...ij,...j->...i", precision, mean), -0.5 * precision]

    @property
    def statistics(self):
        return (self.var, self.second)

    @property
    def precision(self):
        lam = -2.0 * self.eta[1]
        return 0.5 * (lam + np.swapaxes(lam, -1, -2))     # a message is symmetric up to rounding

    @property
    def covariance(self):
        return np.linalg.inv(self.precision)

    @property
    def mean(self):
        return np.linalg.solve(self.precision, self.eta[0][..., None])[..., 0]

    def expectations(self):
        m = self.mean
        return [m, self.covariance + m[..., :, None] * m[..., None, :]]

    def entropy(self):
        d = self.eta[0].shape[-1]
        return float(np.sum(0.5 * (d * math.log(2.0 * math.pi * math.e)
                                   - np.linalg.slogdet(self.precision)[1])))


class WishartNode(LatentNode):
    """q(Lambda) = Wishart(nu, V) over the last two axes of a [..., D, D] precision (leading axes:
    independent matrices): t = (log det Lambda, Lambda), eta = ((nu - D - 1) / 2, -V^-1 / 2).
    E[Lambda] = nu V,  E[log det Lambda] = sum_i psi((nu + 1 - i) / 2) + D log 2 + log det V."""

    def __init__(self, variable, dof, scale):
        LatentNode.__init__(self, variable)
        scale = np.asarray(scale, np.float64)
        self.dim = scale.shape[-1]
        dof = np.broadcast_to(np.asarray(dof, np.float64), scale.shape[:-2])
        self.eta = [0.5 * (dof - self.dim - 1.0), -0.5 * np.linalg.inv(scale)]

    @property
    def statistics(self):
        from ..distribution.core import logdet
        return (logdet(self.var), self.var)

    @property
    def dof(self):
        return 2.0 * self.eta[0] + self.dim + 1.0

    @property
    def scale(self):
        inv = -2.0 * self.eta[1]
        return np.linalg.inv(0.5 * (inv + np.swapaxes(inv, -1, -2)))

    def expectations(self):
        from scipy.special import digamma as psi       # parameter-sized, host side
        nu, V = np.asarray(self.dof), self.scale
        elogdet = psi(0.5 * (nu[..., None] - np.arange(self.dim))).sum(-1) \
            + self.dim * math.log(2.0) + np.linalg.slogdet(V)[1]
        return [np.asarray(elogdet), nu[..., None, None] * V]

    def entropy(self):
        from scipy.special import multigammaln
        d, nu, V = self.dim, np.asarray(self.dof), self.scale
        elogdet = self.expectations()[0]
        log_b = -0.5 * nu * np.linalg.slogdet(V)[1] - 0.5 * nu * d * math.log(2.0) \
            - np.vectorize(lambda t: multigammaln(0.5 * t, d))(nu)
        return float(np.sum(-log_b - 0.5 * (nu - d - 1.0) * elogdet + 0.5 * nu * d))


class DirichletNode(LatentNode):
    """q(theta) = Dirichlet(alpha) over the LAST axis of theta: t = (log theta,),
    eta = (alpha - 1,).  E[log theta_k] = psi(alpha_k) - psi(sum_k alpha_k)."""

    def __init__(self, variable, alpha=1.0):
        LatentNode.__init__(self, variable)
        self.eta = [np.asarray(alpha, np.float64) - 1.0]

    @property
    def statistics(self):
        return (A.log(self.var),)

    @property
    def alpha(self):
        return self.eta[0] + 1.0

    def expectations(self):
        from scipy.special import digamma as psi       # parameter-sized, host side
        a = self.alpha
        return [psi(a) - psi(a.sum(axis=-1, keepdims=True))]

    def entropy(self):
        from scipy.special import digamma as psi, gammaln
        a = self.alpha
        a0 = a.sum(axis=-1)
        log_b = gammaln(a).sum(axis=-1) - gammaln(a0)
        return float(np.sum(log_b + (a0 - a.shape[-1]) * psi(a0) - ((a - 1.0) * psi(a)).sum(axis=-1)))


class CategoricalNode(LatentNode):
    """q(z) = product over leading axes of Categorical over the LAST axis, z one-hot
    (the discrete latent of a mixture, README.md:43): t = (z,), eta = (unnormalised log
    probabilities,).  E[z] = softmax(eta) -- responsibilities.

    ``resident=True``: eta and E[z] live on the backend (``Backend.softmax_rows``: one pass over the
    N x K logits, bsc_softmax_rows on the device); ``log_prob`` may then be None (uniform start,
    never materialised on the host) with ``shape`` = (N, K)."""

    def __init__(self, variable, log_prob=None, resident=False, shape=None):
        LatentNode.__init__(self, variable)
        self.resident = bool(resident)
        self._backend = None
        self._cache = None          # (E[z], log-sum-exp) of the current eta, on the backend
        self._cross = None          # sum_c E[z] * eta per row when eta itself was not stored (set_softmax)
        if log_prob is None:
            if not self.resident or shape is None:
                raise ValueError("log_prob=None needs resident=True and the shape (N, K)")
            self._shape = tuple(int(d) for d in shape)
            self.eta = [None]       # uniform until the first update: E[z] = 1 / K
        else:
            self.eta = [np.asarray(log_prob, np.float64)]
            self._shape = self.eta[0].shape

    @property
    def statistics(self):
        return (self.var,)

    # -- resident protocol (MeanFieldVMP) --------------------------------------------------------
    def bind(self, backend):
        self._backend = backend
        if self.eta[0] is not None and isinstance(self.eta[0], np.ndarray):
            self.eta = [backend.from_host(self.eta[0].astype(np.float32), "float32", len(self._shape))]
        self._drop_cache() if self._cache is not None else None
        self._cache = None

    FUSED = "logits not materialised"     # eta[0] of a node whose softmax was taken inside the product

    def set_softmax(self, r, lse, cross):
        """The node's update when the backend took the softmax inside the logits' product
        (DeviceBackend.evaluate_softmax_rows): responsibilities, log-sum-exp and sum_c r * logits per
        row -- everything expectations and entropy need; the logits themselves were never stored."""
        self._drop_cache()
        self.eta[0] = self.FUSED
        self._cache = (r, lse)
        self._cross = cross
        if hasattr(self._backend, "mark_constant_tensor"):
            self._backend.mark_constant_tensor(r)

    def _drop_cache(self):
        # the responsibilities are marked constant while they stand (the executor then computes a
        # reduction that several messages share -- their column sums -- once); un-mark before the
        # buffer can be reused
        if self._cache is not None and hasattr(self._backend, "unmark_constant"):
            self._backend.unmark_constant(self._cache[0])
        self._cache = None
        self._cross = None

    def set_eta(self, j, value):
        self.eta[j] = value
        self._drop_cache()

    def expectations_backend(self):
        b = self._backend
        if self._cache is None:
            if self.eta[0] is None:
                k = self._shape[-1]
                r = b.broadcast_to(b.constant(1.0 / k), self._shape)
                lse = b.broadcast_to(b.constant(math.log(k)), self._shape[:-1])
                self._cache = (b.materialize(r), lse)
            else:
                self._cache = b.softmax_rows(self.eta[0])
            if hasattr(b, "mark_constant_tensor"):
                b.mark_constant_tensor(self._cache[0])
        return [self._cache[0]]

    def expectations(self):
        if self.resident and self._backend is not None:
            return [np.asarray(self._backend.to_host(self.expectations_backend()[0]), np.float64)]
        e = self.eta[0] - self.eta[0].max(axis=-1, keepdims=True)
        w = np.exp(e)
        return [w / w.sum(axis=-1, keepdims=True)]

    def entropy(self):
        if self.resident and self._backend is not None:
            # H = sum_n (lse_n - sum_k r_nk eta_nk): no log of a responsibility that underflowed
            b = self._backend
            r = self.expectations_backend()[0]
            if hasattr(r, "entropy_terms"):        # responsibilities that were never written: from their statistics
                lse_total, cross_total = r.entropy_terms()
                return lse_total - cross_total
            if self.eta[0] is None:
                return float(np.prod(self._shape[:-1]) * math.log(self._shape[-1]))
            lse = self._cache[1]
            total = b.to_host(b.sum(lse, list(range(len(self._shape) - 1))))
            if self._cross is not None:
                cross = b.to_host(b.sum(self._cross, list(range(len(self._shape) - 1))))
            else:
                cross = b.to_host(b.sum(b.mul(r, self.eta[0]), list(range(len(self._shape)))))
            return float(np.asarray(total, np.float64) - np.asarray(cross, np.float64))
        r = self.expectations()[0]
        return float(-np.sum(np.where(r > 0.0, r * np.log(np.where(r > 0.0, r, 1.0)), 0.0)))


class NormalGammaNode(LatentNode):
    """q(mu, tau) = N(mu | m, 1 / (kappa tau)) Gamma(tau | a, rate b), element-wise over the node's
    shape -- the conjugate pair of a Gaussian with unknown mean AND precision (the per-component,
    per-column factor of BASELINE config 3's mixture).  Sufficient statistics
        t = (tau mu, tau mu^2, log tau, tau),   eta = (kappa m, -kappa / 2, a - 1/2, -b - kappa m^2 / 2)
    are carried by FOUR variables of the log-joint (as MVNormalNode carries w w^T): the front end
    would flatten tau * mu into the surrounding einsum and leave nothing to bind E[tau mu] to."""

    def __init__(self, tau_mu, tau_mu2, log_tau, tau, m=0.0, kappa=1.0, a=1.0, b=1.0):
        LatentNode.__init__(self, tau_mu)
        self._vars = (tau_mu, tau_mu2, log_tau, tau)
        m, kappa, a, b = (np.asarray(v, np.float64) for v in np.broadcast_arrays(m, kappa, a, b))
        self.eta = [kappa * m, -0.5 * kappa, a - 0.5, -b - 0.5 * kappa * m * m]

    @property
    def statistics(self):
        return self._vars

    @property
    def kappa(self):
        return -2.0 * self.eta[1]

    @property
    def m(self):
        return self.eta[0] / self.kappa

    @property
    def a(self):
        return self.eta[2] + 0.5

    @property
    def b(self):
        return -self.eta[3] - 0.5 * self.kappa * self.m ** 2

    def expectations(self):
        from scipy.special import digamma as psi       # parameter-sized, host side
        m, kappa, a, b = self.m, self.kappa, self.a, self.b
        e_tau = a / b
        return [m * e_tau, 1.0 / kappa + m * m * e_tau, psi(a) - np.log(b), e_tau]

    def entropy(self):
        from scipy.special import digamma as psi, gammaln
        kappa, a, b = self.kappa, self.a, self.b
        # H[N(mu | m, 1/(kappa tau))] averaged over tau, plus H[Gamma(a, b)]
        h_mu = 0.5 * math.log(2.0 * math.pi * math.e) - 0.5 * (np.log(kappa) + psi(a) - np.log(b))
        h_tau = a - np.log(b) + gammaln(a) + (1.0 - a) * psi(a)
        return float(np.sum(h_mu + h_tau))


class NotConjugateMessage(ValueError):
    pass


class _ResidentGlobal(object):
    """Mixin for a parameter-sized factor whose natural parameters and expectations live on the
    backend (``resident=True`` in DiagonalMixtureVMP): no message is read back and no expectation
    uploaded per update, so an update issues its launches without a single host synchronisation and
    the Python walk of the next message runs while the device works on the previous one.  float32 on
    the device backend (the host-side nodes keep float64)."""
    resident = True

    def _eta_shapes(self):
        """Shape of each natural parameter (element-wise families: one common shape)."""
        common = tuple(np.broadcast_shapes(*[np.shape(e) for e in self.eta]))
        return [common] * len(self.eta)

    def bind(self, backend):
        self._backend = backend
        self._shapes = [tuple(sh) for sh in self._eta_shapes()]
        self._shape = self._shapes[0]
        # (float64 handed over: a float64 backend keeps it, the device backend stores float32)
        # (np.ascontiguousarray would turn a scalar factor's 0-d parameter into shape (1,))
        self.eta = [backend.from_host(np.array(np.broadcast_to(np.asarray(e, np.float64), sh), order="C"),
                                      "float32", len(sh))
                    for e, sh in zip(self.eta, self._shapes)]
        self._exp = None

    def set_eta(self, j, value):
        b = self._backend
        value = b.materialize(value)
        if tuple(np.shape(value)) != self._shapes[j]:
            # a coefficient that does not depend on one of the statistic's axes comes back with that
            # axis broadcast (extent 1): sum_d LT_kd inside a term gives c_kd = c_k
            value = b.materialize(b.elemwise("add", b.broadcast_to(value, self._shapes[j]), b.constant(0.0)))
        self.eta[j] = value
        self._exp = None

    def host_eta(self):
        return [np.asarray(self._backend.to_host(e), np.float64).reshape(sh) for e, sh in zip(self.eta, self._shapes)]

    def expectations(self):
        return [np.asarray(self._backend.to_host(e), np.float64) for e in self.expectations_backend()]

    def entropy(self):
        return self.host_copy().entropy()


class ResidentDirichletNode(_ResidentGlobal, DirichletNode):
    def expectations_backend(self):
        if self._exp is None:
            b = self._backend
            alpha = b.elemwise("add", self.eta[0], b.constant(1.0))
            last = len(self._shape) - 1
            total = b.sum(alpha, [last])
            axes = list(range(last)) + ["x"]
            psi_total = b.dimshuffle(b.elemwise("digamma", total), axes)
            self._exp = [b.materialize(b.elemwise("add", b.elemwise("digamma", alpha),
                                                  b.mul(b.constant(-1.0), psi_total)))]
        return self._exp

    def host_copy(self):
        return DirichletNode(self.var, alpha=self.host_eta()[0] + 1.0)


class ResidentNormalGammaNode(_ResidentGlobal, NormalGammaNode):
    def expectations_backend(self):
        """(E[tau mu], E[tau mu^2], E[log tau], E[tau]) from eta = (kappa m, -kappa / 2, a - 1/2,
        -b - kappa m^2 / 2), element-wise on the backend."""
        if self._exp is None:
            b = self._backend
            e1, e2, e3, e4 = self.eta
            c = b.constant
            inv = lambda x: b.elemwise("pow", x, c(-1.0))
            kappa = b.materialize(b.mul(c(-2.0), e2))
            m = b.materialize(b.mul(e1, inv(kappa)))
            a = b.materialize(b.elemwise("add", e3, c(0.5)))
            rate = b.materialize(b.elemwise("add", b.mul(c(-1.0), e4), b.mul(c(-0.5), kappa, m, m)))
            a_over_b = b.materialize(b.mul(a, inv(rate)))
            self._exp = [b.materialize(b.mul(m, a_over_b)),
                         b.materialize(b.elemwise("add", inv(kappa), b.mul(m, m, a_over_b))),
                         b.materialize(b.elemwise("add", b.elemwise("digamma", a),
                                                  b.mul(c(-1.0), b.elemwise("log", rate)))),
                         a_over_b]
        return self._exp

    def host_copy(self):
        e1, e2, e3, e4 = self.host_eta()
        kappa = -2.0 * e2
        m = e1 / kappa
        return NormalGammaNode(*self._vars, m=m, kappa=kappa, a=e3 + 0.5, b=-e4 - 0.5 * kappa * m * m)


class ResidentNormalNode(_ResidentGlobal, NormalNode):
    def expectations_backend(self):
        if self._exp is None:
            b = self._backend
            c = b.constant
            var = b.materialize(b.mul(c(-0.5), b.elemwise("pow", self.eta[1], c(-1.0))))
            mean = b.materialize(b.mul(self.eta[0], var))
            self._exp = [mean, b.materialize(b.elemwise("add", b.mul(mean, mean), var))]
        return self._exp

    def host_copy(self):
        e1, e2 = self.host_eta()
        return NormalNode(self.var, mean=e1 * (-0.5 / e2), variance=-0.5 / e2)


class ResidentGammaNode(_ResidentGlobal, GammaNode):
    def expectations_backend(self):
        if self._exp is None:
            b = self._backend
            c = b.constant
            a = b.materialize(b.elemwise("add", self.eta[0], c(1.0)))
            rate = b.materialize(b.mul(c(-1.0), self.eta[1]))
            self._exp = [b.materialize(b.elemwise("add", b.elemwise("digamma", a),
                                                  b.mul(c(-1.0), b.elemwise("log", rate)))),
                         b.materialize(b.mul(a, b.elemwise("pow", rate, c(-1.0))))]
        return self._exp

    def host_copy(self):
        e1, e2 = self.host_eta()
        return GammaNode(self.var, shape=e1 + 1.0, rate=-e2)


class ResidentInverseGammaNode(_ResidentGlobal, InverseGammaNode):
    def expectations_backend(self):
        if self._exp is None:
            b = self._backend
            c = b.constant
            a = b.materialize(b.elemwise("add", b.mul(c(-1.0), self.eta[0]), c(-1.0)))
            scale = b.materialize(b.mul(c(-1.0), self.eta[1]))
            self._exp = [b.materialize(b.elemwise("add", b.elemwise("log", scale),
                                                  b.mul(c(-1.0), b.elemwise("digamma", a)))),
                         b.materialize(b.mul(a, b.elemwise("pow", scale, c(-1.0))))]
        return self._exp

    def host_copy(self):
        e1, e2 = self.host_eta()
        return InverseGammaNode(self.var, shape=-e1 - 1.0, scale=-e2)


class ResidentMVNormalNode(_ResidentGlobal, MVNormalNode):
    """(E[w], E[w w^T]) = (Sigma eta_1, Sigma + m m^T), Sigma = (-2 eta_2)^-1 by the backend's SPD inverse
    (``bsc_inverse_spd``): the D x D precision message of a regression never leaves the device."""

    def _eta_shapes(self):
        return [np.shape(self.eta[0]), np.shape(self.eta[1])]

    def expectations_backend(self):
        if self._exp is None:
            b = self._backend
            c = b.constant
            r = len(self._shapes[0])                  # axes of w: lead..., D
            lead = list(range(r - 1))
            swapped = b.dimshuffle(self.eta[1], lead + [r, r - 1])
            cov = b.materialize(b.inverse_spd(b.materialize(b.mul(c(-1.0), b.elemwise("add", self.eta[1], swapped)))))
            eta1_row = b.dimshuffle(self.eta[0], lead + ["x", r - 1])
            mean = b.materialize(b.sum(b.mul(cov, eta1_row), [r]))
            outer = b.mul(b.dimshuffle(mean, lead + [r - 1, "x"]), b.dimshuffle(mean, lead + ["x", r - 1]))
            self._exp = [mean, b.materialize(b.elemwise("add", cov, outer))]
        return self._exp

    def host_copy(self):
        e1, e2 = self.host_eta()
        lam = -(e2 + np.swapaxes(e2, -1, -2))
        cov = np.linalg.inv(lam)
        return MVNormalNode(self.var, self.second, mean=np.einsum("...ij,...j->...i", cov, e1), covariance=cov)


class ResidentWishartNode(_ResidentGlobal, WishartNode):
    """(E[log det Lambda], E[Lambda]) = (sum_i psi((nu - i) / 2) + D log 2 - log det(-2 eta_2), nu (-2 eta_2)^-1)."""

    def _eta_shapes(self):
        return [np.shape(self.eta[0]), np.shape(self.eta[1])]

    def expectations_backend(self):
        if self._exp is None:
            b = self._backend
            c = b.constant
            d = self.dim
            r = len(self._shapes[0])                  # leading axes
            lead = list(range(r))
            # (a message is symmetric only in what it says about the symmetric Lambda: sum(Lam * outer(sx, mu)) carries
            # sx mu^T, not its symmetric part)
            swapped = b.dimshuffle(self.eta[1], lead + [r + 1, r])
            vinv = b.materialize(b.mul(c(-1.0), b.elemwise("add", self.eta[1], swapped)))
            V = b.materialize(b.inverse_spd(vinv))
            nu = b.materialize(b.elemwise("add", b.mul(c(2.0), self.eta[0]), c(d + 1.0)))
            steps = b.from_host(np.arange(d, dtype=np.float64).reshape([1] * r + [d]) * -0.5, "float32", r + 1)
            half = b.elemwise("add", b.mul(c(0.5), b.dimshuffle(nu, lead + ["x"])), steps)
            psi_sum = b.sum(b.elemwise("digamma", half), [r])
            # (from_host, not constant: a float64 backend then keeps all of D log 2)
            elogdet = b.materialize(b.elemwise("add", psi_sum, b.from_host(np.asarray(d * math.log(2.0)), "float32", 0),
                                               b.mul(c(-1.0), b.logdet(vinv))))
            self._exp = [elogdet, b.materialize(b.mul(b.dimshuffle(nu, lead + ["x", "x"]), V))]
        return self._exp

    def host_copy(self):
        e1, e2 = self.host_eta()
        inv = -(e2 + np.swapaxes(e2, -1, -2))
        return WishartNode(self.var, dof=2.0 * e1 + self.dim + 1.0, scale=np.linalg.inv(inv))


class MeanFieldVMP(object):
    """Coordinate-ascent mean field on a conjugate-exponential log-joint.

    log_joint : scalar expression (or list of summands) over the data and latent vars
    nodes     : LatentNode objects, updated in this order by ``sweep``
    data      : {input name: array}; uploaded to the backend once
    Raises NotConjugate at construction when a node is not conjugate in its blanket.
    """

    def __init__(self, log_joint, nodes, data, backend=None):
        from ..algebra.backend import resolve_backend
        self.backend = resolve_backend(backend)
        self.fuse_softmax = True    # (False: a resident Categorical node's logits are always materialised)
        # the responsibilities of a resident Categorical node are not written when only their statistics are wanted
        self.defer_responsibilities = True
        self._log_joint = list(log_joint) if isinstance(log_joint, (list, tuple)) else [log_joint]
        self._elbo_fns = None
        self.nodes = list(nodes)
        self._by_name = {n.var.name: n for n in self.nodes}
        self._messages = {}
        for node in self.nodes:
            coefficients, _ = conjugate_coefficients(log_joint, node.var, node.statistics)
            others = [m for m in self.nodes if m is not node]
            compiled = []
            for c in coefficients:
                if c is None:
                    compiled.append(None)
                    continue
                bindings = {}
                for m in others:
                    for k, t in enumerate(m.statistics):
                        if self._carrier(t) is not None:
                            continue                 # a statistic that IS a variable feeds it directly
                        bindings[t] = "_E_%s_%d" % (m.var.name, k)
                compiled.append((c, self.backend.compile(c, bindings), bindings))
            self._messages[node.var.name] = compiled
        types = {}
        for compiled in self._messages.values():
            for entry in compiled:
                if entry is not None:
                    types.update(entry[0].input_types)
        # inputs that occur only in latent-free terms of the log-joint carry no message but are
        # part of the bound: elbo() needs them too
        for piece in self._log_joint:
            for name, t in A.wrap_if_literal(piece).input_types.items():
                types.setdefault(name, t)
        self._types = types
        for node in self.nodes:
            if node.resident:
                node.bind(self.backend)
        self._data = {name: self.backend.from_host(value, *types[name])
                      for name, value in data.items() if name in types}
        self._marked = []
        if hasattr(self.backend, "mark_constant"):
            # the data never changes between updates: element-wise values of data alone (x^2 in every
            # message of a Gaussian model) are computed once by the device executor, not per message.
            # The marks are THIS model's: close() (or the model's collection) takes them back, with every
            # value the executor cached from them, so a backend shared by successive models neither keeps
            # their data alive nor serves one model's cached values to the next.
            self._marked = list(self._data.values())
            self.backend.mark_constant(*self._marked)
        carried = {self._carrier(t) for n in self.nodes for t in n.statistics} - {None}
        missing = [n for n in types
                   if n not in self._data and n not in carried and n not in self._by_name]
        if missing:
            raise TypeError("log-joint inputs neither given as data nor declared latent: %s"
                            % ", ".join(sorted(missing)))

    @staticmethod
    def _carrier(statistic):
        """Name of the input variable a statistic is carried by (the latent itself for the
        identity statistic, a second-moment variable, ...), or None for a derived expression."""
        return statistic.name if isinstance(statistic, A.var) else None

    def _expectation_inputs(self, exclude):
        values = {}
        for m in self.nodes:
            if m is exclude:
                continue
            if m.resident:          # already backend values: nothing crosses the host
                for k, (t, e) in enumerate(zip(m.statistics, m.expectations_backend())):
                    values[self._carrier(t) or "_E_%s_%d" % (m.var.name, k)] = e
                continue
            # the expectations of a host-side node change only when its natural parameters are
            # replaced (update() assigns new arrays): their device copies are kept until then, instead
            # of being computed and uploaded again for every message that reads them
            cache = self.__dict__.setdefault("_expectation_cache", {})
            held = cache.get(m.var.name)
            if held is None or len(held[0]) != len(m.eta) or any(a is not b for a, b in zip(held[0], m.eta)):
                uploaded = {}
                for k, (t, e) in enumerate(zip(m.statistics, m.expectations())):
                    name = self._carrier(t) or "_E_%s_%d" % (m.var.name, k)
                    uploaded[name] = self.backend.from_host(np.asarray(e, np.float64), "float32", t.ndim)
                held = cache[m.var.name] = (list(m.eta), uploaded)
            values.update(held[1])
        return values

    def message(self, name):
        """E_q(others)[c_j] for node `name`: the natural parameters VMP assigns to it."""
        node = self._by_name[name]
        inputs = dict(self._data)
        inputs.update(self._expectation_inputs(node))
        out = []
        for entry in self._messages[name]:
            if entry is None:
                out.append(None)
                continue
            c, f, _ = entry
            needed = {k: v for k, v in inputs.items()}
            value = f.device_fn(**needed)
            if node.resident:       # a data-sized message stays where it was computed
                value = self.backend.materialize(value)
            out.append(value)
        if not node.resident:
            # read back only after EVERY statistic's launches are queued: a read-back waits for the
            # device, and the walk of the next expression would otherwise run beside an idle GPU
            out = [None if v is None else np.asarray(self.backend.to_host(v), np.float64) for v in out]
        return out

    def set_data(self, **arrays):
        """Replace data inputs (a new mini-batch): with the data terms of the log-joint written
        times N / B, ``update(name, rho_t)`` is then the stochastic natural-gradient step of
        README.md:69-79 (SVI) for that node -- global nodes only; a local latent such as the
        assignments of a mixture is simply re-created per mini-batch."""
        for name, value in arrays.items():
            if name not in self._types:
                raise TypeError("%s is not an input of the log-joint" % name)
            self._data[name] = self.backend.from_host(value, *self._types[name])
        if hasattr(self.backend, "mark_constant"):
            # only this model's marks: other models on the same backend keep theirs
            self.backend.unmark_constant(*self._marked)
            self._marked = list(self._data.values())
            self.backend.mark_constant(*self._marked)
        if self._elbo_fns is not None:
            self._elbo_data = dict(self._data)

    def close(self):
        """Give the data's constant marks (and the values the executor cached from them) back."""
        marked, self._marked = getattr(self, "_marked", []), []
        if not hasattr(self.backend, "unmark_constant"):
            return
        if marked:
            self.backend.unmark_constant(*marked)
        for node in getattr(self, "nodes", []):        # a resident node's responsibilities are marked while they stand
            cache = getattr(node, "_cache", None)
            if node.resident and cache is not None and hasattr(cache[0], "data_ptr"):
                self.backend.unmark_constant(cache[0])

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: the backend may already be gone
            pass

    def elbo(self):
        """E_q[log p(data, latents)] + sum of the factors' entropies, up to whatever constants
        the log-joint was written without.  The expectation is taken the way the messages are:
        the log-joint is multilinear in the nodes' statistics, so it is evaluated with every
        statistic bound to its expectation.  Coordinate ascent can only raise it -- the check
        the tests apply to every update rule."""
        if self._elbo_fns is None:
            bindings = {}
            for m in self.nodes:
                for k, t in enumerate(m.statistics):
                    if self._carrier(t) is None:
                        bindings[t] = "_E_%s_%d" % (m.var.name, k)
            self._elbo_fns = [self.backend.compile(A.wrap_if_literal(piece), bindings)
                              for piece in self._log_joint]
            self._elbo_data = {n: v for n, v in self._data.items()}
        inputs = dict(self._elbo_data)
        inputs.update(self._expectation_inputs(None))
        total = 0.0
        for f in self._elbo_fns:
            total += float(np.asarray(self.backend.to_host(f.device_fn(**inputs)), np.float64))
        return total + sum(n.entropy() for n in self.nodes)

    def update(self, name, rho=1.0, message_scale=1.0):
        """eta <- (1 - rho) eta + rho * message_scale * message; rho = 1 is the VMP update.
        ``message_scale``: with the data terms of the log-joint written times N / B (mini-batch
        SVI), a LOCAL latent -- one factor per datum, like a mixture's assignments -- is updated
        with 1 / (N / B): its own terms are not replicated."""
        node = self._by_name[name]
        if node.resident and rho == 1.0 and isinstance(node, CategoricalNode) and \
                self.fuse_softmax and hasattr(self.backend, "evaluate_softmax_rows") and \
                len(self._messages[name]) == 1 and self._messages[name][0] is not None and \
                len(node._shape) == 2:
            # a resident Categorical node replaced outright: its logits are only wanted through their
            # softmax, which the backend may take inside the product that forms them
            c, _, bound = self._messages[name][0]
            inputs = dict(self._data)
            inputs.update(self._expectation_inputs(node))
            # (defer: when every neighbour asks only for statistics of the responsibilities against the features
            # the logits were formed from, the device takes them in the softmax's own pass and the [rows, K]
            # responsibilities are never written -- device_backend.DeferredSoftmax)
            r, lse, cross, logits = self.backend.evaluate_softmax_rows(c, inputs, bound, scale=message_scale,
                                                                       **({"defer": True} if self.defer_responsibilities else {}))
            if logits is None:
                node.set_softmax(r, lse, cross)
            else:
                node.set_eta(0, logits)
                node._cache = (r, lse)
                if hasattr(self.backend, "mark_constant_tensor"):
                    self.backend.mark_constant_tensor(r)
            return node
        message = self.message(name)
        if node.resident:
            b = self.backend
            for j, m in enumerate(message):
                if m is None:
                    raise NotConjugateMessage("resident node %s: statistic %d receives no message" % (name, j))
                if message_scale != 1.0:
                    m = b.materialize(b.mul(b.constant(float(message_scale)), m))
                if rho == 1.0 or node.eta[j] is None:
                    node.set_eta(j, m)
                else:
                    node.set_eta(j, b.materialize(b.elemwise(
                        "add", b.mul(b.constant(1.0 - rho), node.eta[j]), b.mul(b.constant(rho), m))))
            return node
        for j, m in enumerate(message):
            # a statistic no term of the log-joint touches receives the message 0: under damping
            # its natural parameter decays like the others instead of keeping its initial value
            if m is None:
                m = 0.0
            elif m.size == np.size(node.eta[j]):
                m = m.reshape(np.shape(node.eta[j]))
            else:
                # a coefficient that does not depend on one of the statistic's axes comes back
                # with that axis broadcast (extent 1): sum_d LT_kd inside a term gives c_kd = c_k
                m = np.broadcast_to(m, np.shape(node.eta[j]))
            node.eta[j] = (1.0 - rho) * node.eta[j] + rho * message_scale * m
        return node

    def sweep(self, rho=1.0):
        for node in self.nodes:
            self.update(node.var.name, rho)
