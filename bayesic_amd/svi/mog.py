"""Natural-gradient SVI for a mixture of Gaussians with the discrete latent
marginalised by summation (BASELINE config 3).

README.md:43,72 (marginalise finite discrete latents by summation, also under
mini-batching) + README.md:36,75-77 (VMP == unit-step natural gradient; SVI, ref
[4]).  One update = expected-parameter kernel -> fused E-step/statistics pass
(fp32 MFMA) -> float64 reduction -> (all-reduce of the K(1+2D) statistics + the
bound term when data-parallel) -> natural-gradient step.  All on the device.
"""
import numpy as np
import torch

from ..device import default_context
from .exchange import Exchange


# -- natural-parameter layout (host side, parameter-sized) ------------------------------------
# eta = [alpha-1 (K) | kappa*m (K*D) | kappa (K*D) | 2a-1 (K*D) | 2b + kappa*m^2 (K*D)]:
# Dirichlet(alpha) over the weights, Normal-Gamma(m, kappa, a, b) per component and dimension
# (the layout csrc/bsc_mog.hip reads).

def prior_eta(K, D, alpha0=1.0, m0=0.0, kappa0=0.01, a0=1.0, b0=1.0):
    """Natural parameters of the prior."""
    return np.concatenate([np.full(K, alpha0 - 1.0), np.full(K * D, kappa0 * m0),
                           np.full(K * D, kappa0), np.full(K * D, 2.0 * a0 - 1.0),
                           np.full(K * D, 2.0 * b0 + kappa0 * m0 * m0)])


def message(stats, K, D):
    """Summed statistics [K, 1+2D] = (sum r, sum r x, sum r x^2) -> natural-parameter increment."""
    stats = np.asarray(stats, np.float64).reshape(K, 1 + 2 * D)
    Rk, Sx, Sxx = stats[:, 0], stats[:, 1:1 + D], stats[:, 1 + D:]
    Rkd = np.repeat(Rk[:, None], D, axis=1)
    return np.concatenate([Rk, Sx.ravel(), Rkd.ravel(), Rkd.ravel(), Sxx.ravel()])


def init_eta(X_sample, K, D, seed=0, **prior):
    """Prior plus one pseudo-observation per component at K randomly chosen rows of
    X_sample (breaks the symmetry between components)."""
    X_sample = np.asarray(X_sample, np.float64)
    picks = X_sample[np.random.RandomState(seed).choice(len(X_sample), K, replace=False)]
    stats = np.zeros((K, 1 + 2 * D))
    stats[:, 0] = 1.0
    stats[:, 1:1 + D] = picks
    stats[:, 1 + D:] = picks * picks + 1.0
    return prior_eta(K, D, **prior) + message(stats, K, D)


def unpack(eta, K, D):
    """(alpha [K], m, kappa, a, b [K, D]) from natural parameters."""
    eta = np.asarray(eta, np.float64)
    alpha = eta[:K] + 1.0
    e1, e2, e3, e4 = (eta[K + i * K * D: K + (i + 1) * K * D].reshape(K, D) for i in range(4))
    m = e1 / e2
    return alpha, m, e2, 0.5 * (e3 + 1.0), 0.5 * (e4 - e2 * m * m)


class MoGNatGradSVI:
    """``via="kernel"`` (default when K <= 64 and D <= 16): the fused E-step of csrc/bsc_mog.hip.
    ``via="executor"`` (any K up to 1024, any D): the same local step written with the algebra
    front end and run by the MI355X executor -- logits = dot(X, B.T) + dot(X*X, A.T) + c through
    the GEMM and one fused add, responsibilities by bsc_softmax_rows, the statistics
    (sum r, r^T X, r^T X^2) as three more contractions.  Slower (the N x K responsibilities are
    materialised) but not limited to one MFMA tile; it is also the cross-check of the kernel."""

    def __init__(self, X, K, eta0, eta_init, n_total=None, ctx=None, group=None, via=None):
        self.ctx = ctx or default_context()
        dev = self.ctx.device
        self.X = X if isinstance(X, torch.Tensor) else self.ctx.to_device(X, torch.float32)
        if self.X.dtype != torch.float32 or self.X.dim() != 2 or self.X.stride(1) != 1:
            raise ValueError("X must be a row-major float32 [B, D] tensor")
        self.B, self.D = self.X.shape
        self.K = int(K)
        self.group = group
        self.exchange = Exchange(self.ctx, group)   # RCCL behind the C ABI when ctx has a communicator
        self.world = self.exchange.world
        self.batch_rows = self.exchange.global_count(self.B, dev)
        self.n_total = float(n_total) if n_total is not None else self.batch_rows
        f64 = torch.float64
        n_eta = self.K + 4 * self.K * self.D
        self.eta0 = torch.as_tensor(eta0, dtype=f64).to(dev).contiguous()
        self.eta = torch.as_tensor(eta_init, dtype=f64).to(dev).clone().contiguous()
        if self.eta0.numel() != n_eta or self.eta.numel() != n_eta:
            raise ValueError("eta must have K + 4*K*D = %d entries" % n_eta)
        self.Wmat = torch.zeros((self.K, 2 * self.D), dtype=torch.float32, device=dev)
        self.c = torch.zeros(self.K, dtype=torch.float32, device=dev)
        # [stats (K*(1+2D)) | lse] contiguous so that one all-reduce covers both
        self.buf = torch.zeros(self.K * (1 + 2 * self.D) + 1, dtype=f64, device=dev)
        self.stats = self.buf[:-1]
        self.lse = self.buf[-1:]
        # the evidence lower bound at the parameters each step STARTS from (README.md:30-37, 69-79):
        # scale * sum_n logsumexp_k + E_q[log p(theta)] - E_q[log q(theta)], formed on the device by the
        # two parameter kernels that run anyway (oracle.svi.mog_elbo)
        self._bound = torch.zeros(1, dtype=f64, device=dev)
        self.elbo = torch.zeros(1, dtype=f64, device=dev)
        self._prior_A = torch.zeros(1, dtype=f64, device=dev)      # A(eta0): a constant of the model, taken once
        self.ctx.call("bsc_mog_log_normalizer", self.eta0, self.K, self.D, self._prior_A)
        self.t = 0
        if via is None:
            via = "kernel" if (self.K <= 64 and self.D <= 16) else "executor"
        if via not in ("kernel", "executor"):
            raise ValueError("via must be 'kernel' or 'executor'")
        self.via = via
        if via == "kernel":
            self.ctx.reserve((2 * self.ctx.info()["cu_count"] + 8) * (64 * 33 + 1) * 4)
        else:
            from .. import algebra as A
            from ..algebra.device_backend import DeviceBackend
            self.backend = be = DeviceBackend(self.ctx)
            Xv, Bv, Av, cv, Rv = A.var("X", 2), A.var("B", 2), A.var("A", 2), A.var("c", 1), A.var("R", 2)
            self._logits_expr = A.dot(Xv, Bv.T) + A.dot(Xv * Xv, Av.T) + A.dimshuffle(cv, "x", 0)
            self._s0 = A.sum(Rv, axis=0).compile(be).device_fn
            self._s1 = A.dot(Rv.T, Xv).compile(be).device_fn
            self._s2 = A.dot(Rv.T, Xv * Xv).compile(be).device_fn
            self._lse_sum = A.sum(A.var("l", 1)).compile(be).device_fn
            # the mini-batch does not change between updates: the executor then keeps X * X, runs the
            # logits as ONE product [X | X^2 | 1] . [B | A | c]^T and -- with the responsibilities marked
            # while they stand -- the three statistics as one product against the same wide operand
            import os
            self._constants = os.environ.get("BSC_MOG_EXECUTOR_CONSTANTS", "1") != "0"   # (0: for A/B runs)
            if self._constants:
                be.mark_constant(self.X)

    def expected_params(self):
        self.ctx.call("bsc_mog_expected_params_bound", self.eta, self.eta0, self._prior_A, self.K, self.D,
                      self.Wmat, self.c, self._bound)

    def local_step(self):
        if self.via == "kernel":
            self.ctx.call("bsc_mog_estep", self.X, self.X.stride(0), self.B, self.D, self.K,
                          self.Wmat, self.c, self.stats, self.lse)
            return
        be, D = self.backend, self.D
        # (K <= 64 and 2 D + 1 <= 64 with a constant mini-batch: the softmax is taken inside the one
        # product that forms the logits -- bsc_gemm_softmax_rows; otherwise product, then bsc_softmax_rows)
        R, lse, _, _ = be.evaluate_softmax_rows(
            self._logits_expr, dict(X=self.X, B=self.Wmat[:, :D], A=self.Wmat[:, D:], c=self.c))
        if self._constants:
            be.mark_constant_tensor(R)
        stats = self.stats.view(self.K, 1 + 2 * D)
        f64 = torch.float64
        # (dtype conversion through bsc_convert; the slice copies are device-to-device plumbing)
        stats[:, 0].copy_(be._convert(be.materialize(self._s0(R=R)), f64))
        stats[:, 1:1 + D].copy_(be._convert(be.materialize(self._s1(R=R, X=self.X)), f64))
        stats[:, 1 + D:].copy_(be._convert(be.materialize(self._s2(R=R, X=self.X)), f64))
        self.lse.copy_(be._convert(be.materialize(self._lse_sum(l=lse)).reshape(1), f64))
        if self._constants:
            be.unmark_constant(R)       # before the buffer can be reused

    def step(self, rho=None):
        """One SVI update; rho defaults to the Robbins-Monro schedule (t+1)^-0.6.  ``self.elbo`` (device,
        float64) then holds the mini-batch estimate of the bound at the parameters the step started from."""
        self.t += 1
        if rho is None:
            rho = (self.t + 1.0) ** -0.6
        self.expected_params()
        self.local_step()
        self.exchange.all_reduce(self.buf)
        self.ctx.call("bsc_mog_natgrad_elbo", self.eta, self.eta0, self.stats, self.K,
                      self.D, self.n_total / self.batch_rows, float(rho), self.lse, self._bound, self.elbo)
