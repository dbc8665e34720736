"""Reparameterisation-trick variational inference for a model written as an algebra expression.

README.md:51 plans "the reparameterization trick [10][11][12]" for continuous non-conjugate
nodes; the reference never wrote it and would have leaned on ``theano.grad`` for the pathwise
derivative.  bayesic_amd has a hand-fused kernel path for BASELINE config 2 (svi/blr.py); this is
the GENERAL form: the model is any log-joint expression whose latent variables carry a leading
Monte-Carlo sample axis, and the derivative of log p with respect to the draws comes from
``bayesic_amd.algebra.autodiff`` -- the executor's own fused map-reduce launches and MFMA GEMMs
run in reverse over the resident data.  Everything parameter-sized (S x P numbers) is host float64.

    q(z) = N(mu, diag e^{2 rho}),  lam = [mu (P) | rho (P)],  z_s = mu + e^{rho} eps_s
    ELBO estimate = mean_s log p(data, z_s) + sum(rho) + P/2 (1 + log 2 pi)
    d/d mu        = mean_s  g_s,                g_s = d log p / d z at z_s
    d/d rho       = mean_s  g_s * eps_s * e^{rho} + 1

Noise is Philox4x32-10 keyed as in ``ScoreFunctionVI`` (and bsc_blr_noise's stream layout is
NOT assumed: this engine shares draws with ScoreFunctionVI, so the two estimators can be compared
on identical noise).
"""
import math

import numpy as np

from ..algebra.autodiff import value_and_grad

_LOG_2PI = math.log(2.0 * math.pi)


class ReparamVI(object):
    """log_joint : expression of ndim 1 (one value per Monte-Carlo sample, mini-batch scaling
                   included) over data inputs and the latent vars
    latents     : list of (var, size); each var has ndim 2 = [S, size]; their concatenation,
                  in this order, is z
    data        : {input name: array}, uploaded once
    noise       : optional callable step -> eps [S, P] (float64); default: Philox draws on the
                  MI355X backend's device
    graph       : device backend only.  The evaluation of log p and its gradient is a few dozen short
                  launches behind a Python walk of the expression (forward, then the tape backwards);
                  with ``graph=True`` that walk is recorded once into a hipGraph
                  (``DeviceBackend.graph_call``) and replayed on every later step: the latent draws
                  are written into fixed device buffers, one graph launch, two results read back.
                  Needs a context on a stream of its own (``Context.set_stream``); falls back to the
                  eager walk otherwise.
    """

    def __init__(self, log_joint, latents, data, n_samples, seed=0, lr=1e-2, backend=None,
                 lam0=None, noise=None, graph=False):
        from ..algebra.backend import resolve_backend
        self.backend = resolve_backend(backend)
        if log_joint.ndim != 1:
            raise ValueError("log_joint must have one value per sample (ndim 1), got ndim %d"
                             % log_joint.ndim)
        self.log_joint = log_joint
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
        types = log_joint.input_types
        names = {v.name for v, _ in self.latents}
        missing = [n for n in types if n not in data and n not in names]
        if missing:
            raise TypeError("log-joint inputs neither data nor latent: %s" % ", ".join(sorted(missing)))
        self._types = types
        self._data = {n: self.backend.from_host(data[n], *types[n]) for n in data if n in types}
        self._noise = noise
        self._eps_dev = None
        self._graph = bool(graph) and hasattr(self.backend, "graph_call")
        self._z_dev = None
        self.elbo, self.grad = None, None

    def set_data(self, **arrays):
        """Replace data inputs (the next mini-batch; write the data term times N / B)."""
        for name, value in arrays.items():
            if name not in self._types or name in {v.name for v, _ in self.latents}:
                raise TypeError("%s is not a data input of the log-joint" % name)
            self._data[name] = self.backend.from_host(value, *self._types[name])     # (a new buffer: a recorded graph is dropped)

    def draw(self, step):
        if self._noise is not None:
            return np.asarray(self._noise(step), np.float64).reshape(self.S, self.P)
        import torch
        ctx = self.backend.ctx
        if self._eps_dev is None:
            self._eps_dev = torch.zeros((self.S, self.P), dtype=torch.float64, device=ctx.device)
        ctx.call("bsc_philox_normal", self.seed, 2, int(step), self.S, self.P, self._eps_dev)
        ctx.sync()
        return self._eps_dev.cpu().numpy()

    def _log_joint_and_gradient_graph(self, z):
        """The same through one recorded hipGraph: fixed input buffers, refreshed in place."""
        import torch
        names = [v.name for v, _ in self.latents]
        if self._z_dev is None:
            self._z_dev = {v.name: self.backend.from_host(np.zeros((self.S, n), self._types[v.name][0]),
                                                          *self._types[v.name]) for v, n in self.latents}
        offset = 0
        for v, n in self.latents:
            block = np.ascontiguousarray(z[:, offset:offset + n], dtype=self._types[v.name][0])
            self._z_dev[v.name].copy_(torch.from_numpy(block))
            offset += n
        inputs = dict(self._data)
        inputs.update(self._z_dev)

        def walk():
            out, grads = value_and_grad(self.backend, self.log_joint, inputs, names)
            return [out] + [grads[name] for name in names]

        res = self.backend.graph_call(("reparam", id(self)), walk,
                                      [self._z_dev[name] for name in names] + list(self._data.values()))
        f = np.asarray(self.backend.to_host(res[0]), np.float64).reshape(self.S)
        g = np.concatenate([np.asarray(self.backend.to_host(r), np.float64).reshape(self.S, n)
                            for r, (_, n) in zip(res[1:], self.latents)], axis=1)
        return f, g

    def log_joint_and_gradient(self, z):
        """(log p(data, z_s) [S], d log p / d z [S, P]) through the executor."""
        if self._graph:
            return self._log_joint_and_gradient_graph(z)
        inputs = dict(self._data)
        offset = 0
        for v, n in self.latents:
            dtype = self._types[v.name][0]
            block = np.ascontiguousarray(z[:, offset:offset + n], dtype=dtype)
            inputs[v.name] = self.backend.from_host(block, *self._types[v.name])
            offset += n
        out, grads = value_and_grad(self.backend, self.log_joint, inputs, [v.name for v, _ in self.latents])
        f = np.asarray(self.backend.to_host(out), np.float64).reshape(self.S)
        g = np.concatenate([np.asarray(self.backend.to_host(grads[v.name]), np.float64).reshape(self.S, n)
                            for v, n in self.latents], axis=1)
        return f, g

    def estimate(self, step):
        """(ELBO estimate, pathwise gradient) at the current lam with the noise of step `step`."""
        P = self.P
        mu, rho = self.lam[:P], self.lam[P:]
        eps = self.draw(step)
        sigma = np.exp(rho)
        f, g = self.log_joint_and_gradient(mu[None, :] + sigma[None, :] * eps)
        elbo = f.mean() + rho.sum() + 0.5 * P * (1.0 + _LOG_2PI)
        grad = np.concatenate([g.mean(axis=0), (g * eps).mean(axis=0) * sigma + 1.0])
        return elbo, grad

    def step(self):
        self.t += 1
        self.elbo, self.grad = self.estimate(self.t - 1)
        b1, b2, eps = 0.9, 0.999, 1e-8
        self.m1 = b1 * self.m1 + (1 - b1) * self.grad
        self.m2 = b2 * self.m2 + (1 - b2) * self.grad ** 2
        mhat = self.m1 / (1 - b1 ** self.t)
        vhat = self.m2 / (1 - b2 ** self.t)
        self.lam = self.lam + self.lr * mhat / (np.sqrt(vhat) + eps)
        return self.elbo
