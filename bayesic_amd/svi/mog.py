"""Natural-gradient SVI for a mixture of Gaussians with the discrete latent
marginalised by summation (BASELINE config 3).

README.md:43,72 (marginalise finite discrete latents by summation, also under
mini-batching) + README.md:36,75-77 (VMP == unit-step natural gradient; SVI, ref
[4]).  One update = expected-parameter kernel -> fused E-step/statistics pass
(fp32 MFMA) -> float64 reduction -> (all-reduce of the K(1+2D) statistics + the
bound term when data-parallel) -> natural-gradient step.  All on the device.
"""
import torch

from ..device import default_context


class MoGNatGradSVI:
    def __init__(self, X, K, eta0, eta_init, n_total=None, ctx=None, group=None):
        self.ctx = ctx or default_context()
        dev = self.ctx.device
        self.X = X if isinstance(X, torch.Tensor) else self.ctx.to_device(X, torch.float32)
        if self.X.dtype != torch.float32 or self.X.dim() != 2 or self.X.stride(1) != 1:
            raise ValueError("X must be a row-major float32 [B, D] tensor")
        self.B, self.D = self.X.shape
        self.K = int(K)
        self.group = group
        self.world = 1
        if group is not None or (torch.distributed.is_available()
                                 and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
        rows = torch.tensor([float(self.B)], dtype=torch.float64, device=dev)
        if self.world > 1:
            torch.distributed.all_reduce(rows, group=self.group)
        self.batch_rows = float(rows.item())
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
        self.t = 0
        self.ctx.reserve((2 * self.ctx.info()["cu_count"] + 8) * (64 * 33 + 1) * 4)

    def expected_params(self):
        self.ctx.call("bsc_mog_expected_params", self.eta, self.K, self.D, self.Wmat,
                      self.c)

    def local_step(self):
        self.ctx.call("bsc_mog_estep", self.X, self.X.stride(0), self.B, self.D, self.K,
                      self.Wmat, self.c, self.stats, self.lse)

    def step(self, rho=None):
        """One SVI update; rho defaults to the Robbins-Monro schedule (t+1)^-0.6."""
        self.t += 1
        if rho is None:
            rho = (self.t + 1.0) ** -0.6
        self.expected_params()
        self.local_step()
        if self.world > 1:
            torch.distributed.all_reduce(self.buf, group=self.group)
        self.ctx.call("bsc_mog_natgrad", self.eta, self.eta0, self.stats, self.K,
                      self.D, self.n_total / self.batch_rows, float(rho))
