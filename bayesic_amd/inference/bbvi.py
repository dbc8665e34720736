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

**The fused route** (``route="auto"``, the default).  ``recognise.logistic_hierarchy`` evaluates the
log-joint on a seven-row instance in host float64 and asks whether it IS
scale * sum_n [y_n l_ns - softplus(l_ns)] + the hierarchical Gaussian / Gamma prior of config 5 for some
(scale, a0, b0), the group of a row given by a one-hot matrix ``Gm`` (``dot(Gm, B.T)`` is how the plugin
surface writes a gathered intercept).  If so -- and ``Gm`` really is one-hot, which is checked on the
device -- the update is svi/bbvi.py's: bsc_logreg_bbvi_loglik (ONE pass over X on fp32 MFMA) and
bsc_bbvi_update (f, control variate, gradient, Adam, next draws), state on the device, ``step()``
asynchronous, ``elbo`` / ``grad`` / ``f`` / ``lam`` read back on access; same draws, so the same update as
the general route up to float32 evaluation order.  ``route="general"`` never takes it, ``"fused"`` insists.
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
                 lam0=None, graph=False, route="auto"):
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
        self._fused = None
        self._lam = np.zeros(2 * self.P)
        if lam0 is None:
            self._lam[self.P:] = math.log(0.05)
        else:
            self._lam[:] = np.asarray(lam0, np.float64)
        self.m1, self.m2 = np.zeros_like(self._lam), np.zeros_like(self._lam)
        self._t = 0
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
        self._elbo, self._grad, self._fv = None, None, None
        if route not in ("auto", "general", "fused"):
            raise ValueError("route must be 'auto', 'general' or 'fused'")
        self.route, self.plan = "general", None
        self.route_reason = None        # why route="auto" did not take the fused route (None: it did, or was not asked)
        if route != "general":
            from .recognise import guarded_route
            why = guarded_route(lambda: self._try_fused_route(log_joint), strict=route == "fused")
            self.route_reason = why
            if why is not None and route == "fused":
                raise ValueError("route='fused': %s" % why)

    # -- the fused route (module docstring) -------------------------------------------------------
    def _try_fused_route(self, log_joint):
        import torch
        from . import recognise
        shapes = {n: tuple(int(k) for k in v.shape) for n, v in self._data.items()}
        said = []
        plan = recognise.logistic_hierarchy(log_joint, self.latents, shapes, self.S, why=said)
        if plan is None:
            return ("the log-joint is not config 5's hierarchical logistic regression in any parameterisation: %s"
                    % (said[-1] if said else "no reason recorded"))
        self.plan = plan
        X, y, Gm = self._data[plan.X], self._data[plan.y], self._data[plan.onehot]
        if not all(isinstance(t, torch.Tensor) and t.dtype == torch.float32 for t in (X, y, Gm)):
            return "the fused pass streams float32 data"
        N, D = X.shape
        G = Gm.shape[1]
        if D > 256 or D % 4 or self.S > 128 or X.stride(1) != 1:
            return "outside the fused pass's envelope (D <= 256 and a multiple of 4, S <= 128, row-major X)"
        # the group matrix must be one-hot: every entry 0 or 1 (sum of squares = sum) and one per row (sum = N,
        # every row sum 1) -- reductions through the executor; the index vector is the matrix times 0 .. G - 1
        b = self.backend
        from .. import algebra as A
        Gv = A.var("Gm", 2)
        ramp = b.from_host(np.arange(G, dtype=np.float32), "float32", 1)
        # (no negative entry: sum |x| = sum x; rows sum to one: sum_n r_n = sum_n r_n^2 = N; then sum x^2 = N iff one-hot)
        checks = [A.sum(Gv), A.sum(Gv * Gv), A.sum(A.sum(Gv, axis=1) * A.sum(Gv, axis=1)), A.sum(A.abs_(Gv))]
        total, squares, row_squares, absolute = (float(np.asarray(b.to_host(e.compile(b).device_fn(Gm=Gm))))
                                                 for e in checks)
        if not (total == float(N) and squares == float(N) and row_squares == float(N) and absolute == float(N)):
            return "the group matrix %s is not one-hot" % plan.onehot
        g = b.materialize(A.dot(Gv, A.var("ramp", 1)).compile(b).device_fn(Gm=Gm, ramp=ramp))
        g = g.round().to(torch.int32)                     # (dtype conversion: plumbing)
        order = [v.name for v, _ in self.latents]
        if order != [plan.W, plan.B, plan.zeta]:
            return "latents must be listed as (weights, group intercepts, log precision) for the fused layout"
        from ..svi.bbvi import LogRegBBVI
        self._fused = LogRegBBVI(X, y, g, G, n_total=plan.scale * N, n_samples=self.S, seed=self.seed, lr=self.lr,
                                 a0=plan.a0, b0=plan.b0, ctx=b.ctx, lam0=self._lam)
        self.route = "fused: bsc_logreg_bbvi_loglik + bsc_bbvi_update"
        return None

    @property
    def lam(self):
        return self._fused.lam.cpu().numpy() if self._fused is not None else self._lam

    @lam.setter
    def lam(self, value):
        if self._fused is not None:
            raise AttributeError("on the fused route the variational parameters live on the device; build the "
                                 "engine with lam0=")
        self._lam = value

    @property
    def t(self):
        return self._fused.t if self._fused is not None else self._t

    @t.setter
    def t(self, value):
        self._t = value

    @property
    def elbo(self):
        if self._fused is not None:
            return float(self._fused.elbo.item()) + self.plan.offset if self._fused.t else None
        return self._elbo

    @elbo.setter
    def elbo(self, value):
        self._elbo = value

    @property
    def grad(self):
        if self._fused is not None:
            return self._fused.grad.cpu().numpy() if self._fused.t else None
        return self._grad

    @grad.setter
    def grad(self, value):
        self._grad = value

    @property
    def f(self):
        if self._fused is not None:
            return self._fused.f.cpu().numpy() + self.plan.offset if self._fused.t else None
        return self._fv

    @f.setter
    def f(self, value):
        self._fv = value

    def set_data(self, **arrays):
        """Replace data inputs (the next mini-batch; write the data term times N / B)."""
        if self._fused is not None:         # (before anything is touched: the engine stays as it was)
            raise NotImplementedError("set_data on the fused route: build a new engine for another mini-batch "
                                      "(the one-hot group matrix is converted to an index vector at construction)")
        for name in arrays:
            if name not in self._types or name in {v.name for v, _ in self.latents}:
                raise TypeError("%s is not a data input of the log-joint" % name)
        for name, value in arrays.items():
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
        if self._fused is not None:
            self._fused.step()          # asynchronous: one pass over X + the fused update, on the context's stream
            return None
        self.t += 1
        self.elbo, self.grad, self.f = self.estimate(self.t - 1)
        b1, b2, eps = 0.9, 0.999, 1e-8
        self.m1 = b1 * self.m1 + (1 - b1) * self.grad
        self.m2 = b2 * self.m2 + (1 - b2) * self.grad ** 2
        mhat = self.m1 / (1 - b1 ** self.t)
        vhat = self.m2 / (1 - b2 ** self.t)
        self.lam = self.lam + self.lr * mhat / (np.sqrt(vhat) + eps)
        return self.elbo
