"""Reparameterised-ELBO SVI for Bayesian linear regression (BASELINE config 2).

Host-side driver of the device path; every numeric step is a C-ABI call into
libbayesic_hip.so.  The reference has no inference code -- this implements
README.md:51 (reparameterisation-trick gradient, refs [10][11][12]) with
mini-batch scaling per README.md:69-79, on the likelihood decomposition of
bayesic/distribution/base.py:47-69.

Model:  y_n ~ N(x_n.w, s2),  w | s2 ~ N(0, s2 I),  s2 ~ InvGamma(alpha0, beta0)
q:      w ~ N(m, diag e^{2 rho}),  log s2 ~ N(a, e^{2b});  lam = [m, rho, a, b].

Data parallelism: each rank holds a contiguous block of mini-batch rows.  The
only exchange per update is ONE all-reduce(sum) of the float64 vector
[Q (S), G (S*D)] (16 KB at S=8, D=256); noise is keyed by (seed, step, sample,
parameter) and never by rank, so all ranks apply the identical update.
"""
import math

import torch

from ..device import default_context
from .._ffi import ptr


class BLRReparamSVI:
    def __init__(self, X, y, n_total=None, n_samples=8, seed=1234, lr=1e-2, alpha0=1.0,
                 beta0=1.0, ctx=None, group=None, lam0=None):
        self.ctx = ctx or default_context()
        dev = self.ctx.device
        self.X = X if isinstance(X, torch.Tensor) else self.ctx.to_device(X, torch.float32)
        self.y = y if isinstance(y, torch.Tensor) else self.ctx.to_device(y, torch.float32)
        if self.X.dtype != torch.float32 or self.y.dtype != torch.float32:
            raise TypeError("X and y must be float32")
        if self.X.dim() != 2 or self.y.dim() != 1 or self.X.shape[0] != self.y.shape[0]:
            raise ValueError("X must be [B, D] and y [B]")
        if self.X.stride(1) != 1:
            raise ValueError("X must be row-major (unit stride along columns)")
        self.B, self.D = self.X.shape
        self.S = int(n_samples)
        self.seed = int(seed)
        self.lr = float(lr)
        self.alpha0, self.beta0 = float(alpha0), float(beta0)
        self.group = group
        self.world = 1
        if group is not None or (torch.distributed.is_available()
                                 and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
        # global mini-batch rows (all ranks); ranks may hold unequal blocks
        rows = torch.tensor([float(self.B)], dtype=torch.float64, device=dev)
        if self.world > 1:
            torch.distributed.all_reduce(rows, group=self.group)
        self.batch_rows = float(rows.item())
        self.n_total = float(n_total) if n_total is not None else self.batch_rows
        D, S = self.D, self.S
        f64 = torch.float64
        self.lam = torch.zeros(2 * D + 2, dtype=f64, device=dev)
        if lam0 is None:
            self.lam[D:2 * D] = math.log(0.1)
            self.lam[2 * D + 1] = math.log(0.1)
        else:
            self.lam.copy_(torch.as_tensor(lam0, dtype=f64))
        self.m1 = torch.zeros_like(self.lam)
        self.m2 = torch.zeros_like(self.lam)
        self.grad = torch.zeros_like(self.lam)
        self.elbo = torch.zeros(1, dtype=f64, device=dev)
        self.eps = torch.zeros(S * (D + 1), dtype=f64, device=dev)
        self.W = torch.zeros(S * D, dtype=torch.float32, device=dev)
        self.xi = torch.zeros(S, dtype=f64, device=dev)
        self.stats = torch.zeros(S * (D + 1), dtype=f64, device=dev)  # [Q | G]
        self.Q = self.stats[:S]
        self.G = self.stats[S:]
        self.t = 0
        # size the slab once so step() never allocates
        self.ctx.reserve(2 * self.ctx.info()["cu_count"] * (8 * 256 + 8) * 4)

    # -- the four phases of one update --------------------------------------
    def sample(self, step):
        self.ctx.call("bsc_blr_sample", ptr(self.lam), self.D, self.S, self.seed, step,
                      ptr(self.eps), ptr(self.W), ptr(self.xi))

    def data_pass(self):
        self.ctx.call("bsc_blr_data_pass", ptr(self.X), self.X.stride(0), ptr(self.y), self.B,
                      self.D, ptr(self.W), self.S, ptr(self.Q), ptr(self.G))

    def all_reduce(self):
        if self.world > 1:
            torch.distributed.all_reduce(self.stats, group=self.group)

    def elbo_grad(self):
        self.ctx.call("bsc_blr_elbo_grad", ptr(self.lam), ptr(self.eps), ptr(self.W),
                      ptr(self.xi), ptr(self.Q), ptr(self.G), self.D, self.S, self.batch_rows,
                      self.n_total / self.batch_rows, self.alpha0, self.beta0, ptr(self.elbo),
                      ptr(self.grad))

    def adam(self):
        self.ctx.call("bsc_adam_ascent", ptr(self.lam), ptr(self.grad), ptr(self.m1),
                      ptr(self.m2), self.lam.numel(), self.t, self.lr, 0.9, 0.999, 1e-8)

    def step(self):
        """One ELBO-gradient update; asynchronous on the context stream."""
        self.t += 1
        self.sample(self.t - 1)
        self.data_pass()
        self.all_reduce()
        self.elbo_grad()
        self.adam()

    # -- host views -----------------------------------------------------------
    def params(self):
        lam = self.lam.cpu().numpy()
        D = self.D
        return dict(m=lam[:D], rho=lam[D:2 * D], a=lam[2 * D], b=lam[2 * D + 1])
