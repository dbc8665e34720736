"""BASELINE config 3's model written as a symbolic log-joint, its updates DERIVED by the mean-field
engine (inference/vmp.py) instead of hand-fused (svi/mog.py, csrc/bsc_mog.hip):

    z_n ~ Cat(pi),  x_nd | z_n = k ~ N(mu_kd, 1 / tau_kd),
    pi ~ Dirichlet(alpha0),  (mu_kd, tau_kd) ~ NormalGamma(m0, kappa0, a0, b0)

README.md:43,72 (finite discrete latents marginalised by summation, also under mini-batching) and
README.md:36,75-77 (VMP = unit-step natural gradient; SVI).  The N x K assignments are a RESIDENT
node: their logits, responsibilities and every message computed from them stay on the backend.
So are, by default, the parameter-sized factors (``resident_globals``): an update then reads nothing
back and uploads nothing -- no host synchronisation, the Python walk of one message runs while the
device works on the previous one (10M x 16, K = 64: 3.05 -> 2.45 ms per update).
One ``step(rho)`` = local update of q(z) (rho = 1), then the damped natural-gradient step on the
global factors -- the same update as ``oracle.svi.mog_svi_step`` / ``MoGNatGradSVI.step``.

``route``: the symbolic log-joint is the plugin surface, not the execution plan.  ``"derived"`` walks the
derived messages through the executor (since round 3 the responsibilities are not written: their
statistics are taken in the softmax's own pass, bsc_gemm_softmax_stats -- 1.5 ms per update at 10M x 16,
K = 64).  ``"auto"`` (default) first asks ``recognise.diagonal_mixture`` whether the update rules that
``match`` derived from the log-joint ARE the ones csrc/bsc_mog.hip computes -- checked by running them on a
six-row instance in host float64 at random parameters -- and if so hands the update to those three kernels
(svi/mog.py: expected parameters, E-step + statistics, natural-gradient step; 0.85 ms), the prior's
natural parameters read off the derived messages.  The node objects then lag behind the device state until
``sync_nodes()``.
"""
import numpy as np

from .. import algebra as A
from .vmp import (CategoricalNode, DirichletNode, MeanFieldVMP, NormalGammaNode, ResidentDirichletNode,
                  ResidentNormalGammaNode)


def diagonal_mixture_log_joint(X, Z, pi, TM, TM2, LT, T, scale, alpha0, m0, kappa0, a0, b0):
    """Constants dropped; data terms times ``scale`` = N / B (README.md:69-79).  TM, TM2, LT, T
    carry tau mu, tau mu^2, log tau, tau of the NormalGamma factors ([K, D] each)."""
    row = lambda v: A.dimshuffle(v, "x", 0)
    lik = A.sum(Z * A.dot(X, TM.T)) + A.sum(Z * A.dot(X * X, T.T)) * (-0.5) \
        + A.sum(Z * row(A.sum(LT, axis=1))) * 0.5 + A.sum(Z * row(A.sum(TM2, axis=1))) * (-0.5)
    prior_z = A.sum(Z * row(A.log(pi)))
    prior_pi = A.sum(A.log(pi)) * (alpha0 - 1.0)
    prior_ng = A.sum(LT) * (a0 - 0.5) + A.sum(T) * (-b0 - 0.5 * kappa0 * m0 * m0) \
        + A.sum(TM2) * (-0.5 * kappa0) + A.sum(TM) * (kappa0 * m0)
    return (lik + prior_z) * scale + prior_pi + prior_ng


class DiagonalMixtureVMP(object):
    def __init__(self, X, K, n_total=None, alpha0=1.0, m0=0.0, kappa0=0.01, a0=1.0, b0=1.0, init=None,
                 backend=None, dtype="float32", resident=True, resident_globals=None, route="auto"):
        """X: [N, D] host array (uploaded once).  ``init`` = (alpha, m, kappa, a, b) of the starting
        factors ([K] and [K, D] arrays)."""
        N, D = X.shape
        self.N, self.D, self.K = int(N), int(D), int(K)
        scale = float(n_total) / N if n_total is not None else 1.0
        self.scale = scale
        v = lambda name, nd: A.var(name, nd, dtype)
        Xv, Z, pi = v("X", 2), v("Z", 2), v("pi", 1)
        TM, TM2, LT, T = v("TM", 2), v("TM2", 2), v("LT", 2), v("T", 2)
        lj = diagonal_mixture_log_joint(Xv, Z, pi, TM, TM2, LT, T, scale, alpha0, m0, kappa0, a0, b0)
        alpha, m, kappa, a, b = init
        self.z = CategoricalNode(Z, log_prob=None if resident else np.zeros((N, K)), resident=resident,
                                 shape=(N, K))
        # resident_globals: the parameter-sized factors live on the backend too (no read-back, no
        # upload, no host synchronisation inside an update)
        if resident_globals is None:
            resident_globals = resident
        self.resident_globals = bool(resident_globals)
        Dir, NG = (ResidentDirichletNode, ResidentNormalGammaNode) if resident_globals else \
            (DirichletNode, NormalGammaNode)
        self.pi = Dir(pi, alpha=np.asarray(alpha, np.float64))
        self.ng = NG(TM, TM2, LT, T, m=m, kappa=kappa, a=a, b=b)
        self.vmp = MeanFieldVMP(lj, [self.z, self.pi, self.ng], {"X": X}, backend=backend)
        self.t = 0
        if route not in ("auto", "derived", "fused"):
            raise ValueError("route must be 'auto', 'derived' or 'fused'")
        self.route, self._fused = "derived", None
        self.route_reason = None        # why route="auto" stayed on the derived route (None: it did not, or was not asked)
        if route != "derived":
            from .recognise import guarded_route
            why = guarded_route(lambda: self._try_fused_route(lj, Z, pi, (TM, TM2, LT, T), init), strict=route == "fused")
            self.route_reason = why
            if why is not None and route == "fused":
                raise ValueError("route='fused': %s" % why)

    def _try_fused_route(self, lj, Z, pi, ng_vars, init):
        from . import recognise
        backend = self.vmp.backend
        if not hasattr(backend, "ctx"):
            return "the fused kernels run on the MI355X backend"
        import torch
        X = self.vmp._data["X"]
        if not (isinstance(X, torch.Tensor) and X.dtype == torch.float32 and X.stride(1) == 1):
            return "the fused E-step streams row-major float32 data"
        said = []
        eta0 = recognise.diagonal_mixture(lj, Z, pi, ng_vars, "X", self.K, self.D, self.scale, why=said)
        if eta0 is None:
            return ("the derived update rules are not those of a diagonal Gaussian mixture with Dirichlet / Normal-Gamma "
                    "factors: %s" % (said[-1] if said else "no reason recorded"))
        from ..svi.mog import MoGNatGradSVI
        alpha, m, kappa, a, b = (np.asarray(v, np.float64) for v in init)
        shape = (self.K, self.D)
        m, kappa, a, b = (np.broadcast_to(v, shape) for v in (m, kappa, a, b))
        eta = np.concatenate([alpha - 1.0, (kappa * m).ravel(), kappa.ravel(), (2.0 * a - 1.0).ravel(),
                              (2.0 * b + kappa * m * m).ravel()])
        self._fused = MoGNatGradSVI(X, self.K, eta0, eta, n_total=self.scale * self.N, ctx=backend.ctx)
        self.route = "fused: bsc_mog_expected_params + bsc_mog_estep + bsc_mog_natgrad (via=%s)" % self._fused.via
        return None

    def sync_nodes(self):
        """Bring the node objects (``pi``, ``ng``, ``z`` and with them ``vmp.elbo()``) up to the device state of the
        fused route: the global factors' natural parameters are copied in, the local factor is updated from them."""
        if self._fused is None:
            return
        eta = self._fused.eta.cpu().numpy()
        K, D, KD = self.K, self.D, self.K * self.D
        e = [eta[K + j * KD:K + (j + 1) * KD].reshape(K, D) for j in range(4)]
        ng_eta = [e[0], -0.5 * e[1], 0.5 * e[2], -0.5 * e[3]]          # the nodes' (kappa m, -kappa/2, a - 1/2, -b - kappa m^2/2)
        b = self.vmp.backend
        if self.resident_globals:
            self.pi.set_eta(0, b.from_host(eta[:K], "float32", 1))
            for j in range(4):
                self.ng.set_eta(j, b.from_host(np.ascontiguousarray(ng_eta[j]), "float32", 2))
        else:
            self.pi.eta[0] = eta[:K].copy()
            self.ng.eta = [np.ascontiguousarray(v) for v in ng_eta]
        self.vmp.__dict__.pop("_expectation_cache", None)
        self.vmp.update("Z", 1.0, message_scale=1.0 / self.scale)

    def elbo(self):
        """The mini-batch estimate of the bound (oracle.svi.mog_elbo): on the fused route the device scalar of the
        last step (the bound at the parameters that step started from); on the derived route ``vmp.elbo()`` -- the
        same bound up to the constants the symbolic log-joint was written without."""
        if self._fused is not None:
            return float(self._fused.elbo.item())
        return self.vmp.elbo()

    def step(self, rho=None):
        self.t += 1
        if rho is None:
            rho = (self.t + 1.0) ** -0.6
        if self._fused is not None:
            self._fused.step(rho)
            return rho
        self.vmp.update("Z", 1.0, message_scale=1.0 / self.scale)     # a local latent: its terms are not replicated
        self.vmp.update("pi", rho)
        self.vmp.update("TM", rho)
        return rho

    def close(self):
        """Release the data's constant marks on the backend (and what the executor cached from them)."""
        self.vmp.close()

    def eta_fused_layout(self):
        """Natural parameters in the layout of svi/mog.py and oracle.svi:
        [alpha - 1 | kappa m | kappa | 2a - 1 | 2b + kappa m^2]."""
        if self._fused is not None:
            return self._fused.eta.cpu().numpy()
        ng, pi = self.ng, self.pi
        if self.resident_globals:
            ng, pi = ng.host_copy(), pi.host_copy()
        return np.concatenate([pi.alpha - 1.0, (ng.kappa * ng.m).ravel(), ng.kappa.ravel(),
                               (2.0 * ng.a - 1.0).ravel(), (2.0 * ng.b + ng.kappa * ng.m ** 2).ravel()])
