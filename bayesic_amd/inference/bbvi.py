"""Black-box variational inference for a model written as a bayesic.algebra expression.

README.md:52 plans, for "non-conjugate/non-exp-family continuous nodes", "the gradient
estimator from Black box variational inference [3] -- either the Rao-Blackwellized one or
the control variate one".  The reference never wrote it; bayesic_amd has a hand-fused
kernel path for BASELINE config 5 (svi/bbvi.py).  This module is the GENERAL form of the
same estimator: the model is any scalar log-joint expression whose latent variables carry
a leading Monte-Carlo sample axis, evaluated by the executor (on the MI355X backend the
data-sized work is fused map-reduce launches and MFMA GEMMs over the resident data);
everything parameter-sized (S x P numbers) is host float64.

    q(z) = N(mu, diag e^{2 rho}),  lam = [mu (P) | rho (P)],  z = concatenation of the latents
    f_s  = log p(data, z_s) - log q(z_s)
    h_s  = grad_lam log q(z_s) = [eps_s / sigma | eps_s^2 - 1]
    a    = sum_i Cov(f h_i, h_i) / sum_i Var(h_i)          (scalar control variate)
    grad = mean_s (f_s - a) h_s,   ELBO estimate = mean_s f_s,   Adam ascent on lam

Noise is Philox4x32-10 keyed (parameter block, sample, stream 2, step) -- the keying of
bsc_bbvi_sample -- so with the same seed this engine and the fused config-5 kernel see the
same draws (tests/test_inference.py cross-checks them).
"""
import math

import numpy as np

_LOG_2PI = math.log(2.0 * math.pi)


class ScoreFunctionVI(object):
    """log_joint : expression of ndim 1 (one value per Monte-Carlo sample, mini-batch scaling
                   included) over data inputs and the latent vars
    latents     : list of (var, size); each var has ndim 2 = [S, size]; their concatenation,
                  in this order, is z
    data        : {input name: array}, uploaded once
    """

    def __init__(self, log_joint, latents, data, n_samples, seed=0, lr=1e-2, backend=None,
                 lam0=None, graph=False):
        from ..algebra.backend import resolve_backend
        from ..algebra.device_backend import DeviceBackend
        self.backend = resolve_backend(backend)
        if not isinstance(self.backend, DeviceBackend):
            raise TypeError("ScoreFunctionVI runs on the MI355X backend (its noise is drawn on the device)")
        if log_joint.ndim != 1:
            raise ValueError("log_joint must have one value per sample (ndim 1), got ndim %d"
                             % log_joint.ndim)
        self.latents = [(v, int(n)) for v, n in latents]
        for v, _ in self.latents:
            if v.ndim != 2:
                raise ValueError("latent %s must be [samples, size] (ndim 2)" % v.name)
        self.P = sum(n for _, n in self.latents)
        self.S, self.seed, self.lr = int(n_samples), int(seed), float(lr)
        self.lam = np.zeros(2 * self.P)
        if lam0 is None:
            self.lam[self.P:] = math.log(0.05)
        else:
            self.lam[:] = np.asarray(lam0, np.float64)
        self.m1, self.m2 = np.zeros_like(self.lam), np.zeros_like(self.lam)
        self.t = 0
        # graph=True: the evaluation's launches are recorded once as a hipGraph and replayed
        # (DeviceBackend.compile(graph=True)); the latent draws then live in fixed device buffers
        self._graph = bool(graph)
        self._f = log_joint.compile(self.backend, graph=True) if self._graph else log_joint.compile(self.backend)
        self._z_dev = None
        types = log_joint.input_types
        latent_names = {v.name for v, _ in self.latents}
        missing = [n for n in types if n not in data and n not in latent_names]
        if missing:
            raise TypeError("log-joint inputs neither data nor latent: %s" % ", ".join(sorted(missing)))
        self._data = {n: self.backend.from_host(data[n], *types[n]) for n in data if n in types}
        self._types = types
        import torch
        self._eps_dev = torch.zeros((self.S, self.P), dtype=torch.float64, device=self.backend.ctx.device)
        self.elbo, self.grad, self.f = None, None, None

    def set_data(self, **arrays):
        """Replace data inputs (the next mini-batch; write the data term times N / B)."""
        for name, value in arrays.items():
            if name not in self._types or name in {v.name for v, _ in self.latents}:
                raise TypeError("%s is not a data input of the log-joint" % name)
            self._data[name] = self.backend.from_host(value, *self._types[name])

    def draw(self, step):
        """eps [S, P] for Philox step `step` (device draw, downloaded: parameter-sized)."""
        ctx = self.backend.ctx
        ctx.call("bsc_philox_normal", self.seed, 2, int(step), self.S, self.P, self._eps_dev)
        ctx.sync()
        return self._eps_dev.cpu().numpy()

    def log_joint_values(self, z):
        """log p(data, z_s) for the rows of z [S, P]; the latents go to the device as float32."""
        inputs = dict(self._data)
        offset = 0
        if self._graph and self._z_dev is None:
            self._z_dev = {v.name: self.backend.from_host(np.zeros((self.S, n), np.float32), *self._types[v.name])
                           for v, n in self.latents}
        for v, n in self.latents:
            block = np.ascontiguousarray(z[:, offset:offset + n], dtype=np.float32)
            if self._graph:
                import torch
                self._z_dev[v.name].copy_(torch.from_numpy(block))
                inputs[v.name] = self._z_dev[v.name]
            else:
                inputs[v.name] = self.backend.from_host(block, *self._types[v.name])
            offset += n
        out = self.backend.to_host(self._f.device_fn(**inputs))
        return np.asarray(out, np.float64).reshape(self.S)

    def estimate(self, step):
        """(ELBO estimate, gradient, f) at the current lam with the noise of Philox step `step`."""
        P, S = self.P, self.S
        mu, rho = self.lam[:P], self.lam[P:]
        eps = self.draw(step)
        sigma = np.exp(rho)
        z = mu[None, :] + sigma[None, :] * eps
        log_q = (-0.5 * _LOG_2PI - rho[None, :] - 0.5 * eps * eps).sum(axis=1)
        f = self.log_joint_values(z) - log_q
        h = np.concatenate([eps / sigma[None, :], eps * eps - 1.0], axis=1)
        fh = f[:, None] * h
        cov = ((fh - fh.mean(0)) * (h - h.mean(0))).sum(0) / (S - 1)
        var = ((h - h.mean(0)) ** 2).sum(0) / (S - 1)
        a = cov.sum() / var.sum()
        grad = ((f - a)[:, None] * h).mean(axis=0)
        return f.mean(), grad, f

    def step(self):
        self.t += 1
        self.elbo, self.grad, self.f = self.estimate(self.t - 1)
        b1, b2, eps = 0.9, 0.999, 1e-8
        self.m1 = b1 * self.m1 + (1 - b1) * self.grad
        self.m2 = b2 * self.m2 + (1 - b2) * self.grad ** 2
        mhat = self.m1 / (1 - b1 ** self.t)
        vhat = self.m2 / (1 - b2 ** self.t)
        self.lam = self.lam + self.lr * mhat / (np.sqrt(vhat) + eps)
        return self.elbo
