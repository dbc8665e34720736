"""Black-box VI (score-function gradient with control variate) for hierarchical
logistic regression (BASELINE config 5).

README.md:52 -> ref [3] (Ranganath et al.), mini-batched per README.md:69-79.
One update = Philox draw of S=64 parameter samples -> ONE pass over the
mini-batch giving the 64 log-likelihoods (fp32 MFMA) -> (all-reduce of those 64
float64 values when data-parallel) -> f_s, control variate, gradient -> Adam.
"""
import math

import torch

from ..device import default_context
from .exchange import Exchange


class LogRegBBVI:
    def __init__(self, X, y, g, n_groups, n_total=None, n_samples=64, seed=1234, lr=1e-2,
                 a0=1.0, b0=1.0, ctx=None, group=None, lam0=None):
        self.ctx = ctx or default_context()
        dev = self.ctx.device
        as_t = lambda v, dt: v if isinstance(v, torch.Tensor) else torch.as_tensor(v).to(dt).to(dev)
        self.X = as_t(X, torch.float32)
        self.y = as_t(y, torch.float32)
        self.g = as_t(g, torch.int32)
        if self.X.dim() != 2 or self.X.stride(1) != 1 or self.X.dtype != torch.float32:
            raise ValueError("X must be a row-major float32 [B, D] tensor")
        if self.g.dtype != torch.int32 or self.y.dtype != torch.float32:
            raise TypeError("y must be float32 (0/1) and g int32")
        self.B, self.D = self.X.shape
        self.G, self.S = int(n_groups), int(n_samples)
        self.P = self.D + self.G + 1
        self.seed, self.lr, self.a0, self.b0 = int(seed), float(lr), float(a0), float(b0)
        self.group = group
        self.exchange = Exchange(self.ctx, group)   # RCCL behind the C ABI when ctx has a communicator
        self.world = self.exchange.world
        self.batch_rows = self.exchange.global_count(self.B, dev)
        self.n_total = float(n_total) if n_total is not None else self.batch_rows
        f64 = torch.float64
        P, S = self.P, self.S
        self.lam = torch.zeros(2 * P, dtype=f64, device=dev)
        if lam0 is None:
            self.lam[P:] = math.log(0.05)
        else:
            self.lam.copy_(torch.as_tensor(lam0, dtype=f64))
        self.m1, self.m2 = torch.zeros_like(self.lam), torch.zeros_like(self.lam)
        self.grad = torch.zeros_like(self.lam)
        self.eps = torch.zeros(S * P, dtype=f64, device=dev)
        self.Wz = torch.zeros(S * self.D, dtype=torch.float32, device=dev)
        self.Bz = torch.zeros(self.G * S, dtype=torch.float32, device=dev)
        self.zeta = torch.zeros(S, dtype=f64, device=dev)
        self.ell = torch.zeros(S, dtype=f64, device=dev)
        self.f = torch.zeros(S, dtype=f64, device=dev)
        self.elbo = torch.zeros(1, dtype=f64, device=dev)
        self.t = 0
        self._drawn = False
        self.ctx.reserve((2 * self.ctx.info()["cu_count"] + 8) * 64 * 4)

    def step(self):
        """One update.  The draws of update t are made at the end of update t - 1 (inside
        ``bsc_bbvi_update``, from the parameters that update has just written), so between two
        updates ``self.eps`` / ``Wz`` / ``Bz`` hold the NEXT update's noise; assigning to
        ``self.lam`` from outside must be followed by ``invalidate_draws()``."""
        self.t += 1
        c = self.ctx
        if not self._drawn:
            c.call("bsc_bbvi_sample", self.lam, self.D, self.G, self.S, self.seed, self.t - 1,
                   self.eps, self.Wz, self.Bz, self.zeta)
        c.call("bsc_logreg_bbvi_loglik", self.X, self.X.stride(0), self.y, self.g, self.B, self.D,
               self.G, self.Wz, self.Bz, self.S, self.ell)
        self.exchange.all_reduce(self.ell)
        # f, control variate, gradient, Adam and the next update's draws: three launches
        c.call("bsc_bbvi_update", self.lam, self.eps, self.ell, self.D, self.G, self.S,
               self.n_total / self.batch_rows, self.a0, self.b0, self.m1, self.m2, self.t, self.lr,
               0.9, 0.999, 1e-8, self.seed, self.t, self.Wz, self.Bz, self.zeta, self.elbo, self.grad,
               self.f)
        self._drawn = True

    def invalidate_draws(self):
        """Call after changing ``lam`` (or ``seed`` / ``t``) by hand: the next update redraws."""
        self._drawn = False
