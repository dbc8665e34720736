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

**The fused route.**  The expression is the plugin surface (bayesic/algebra.py:42-58; Distribution nodes,
bayesic/distribution/base.py:9-172), not the execution plan.  Before anything runs, ``recognise.gaussian_linear``
reads the data-sized structure off the log-joint with ``match`` (bayesic/algebra.py:1037-1063): when the data
enter only through ``c_s * sum_n (y_n - x_n . w_s)^2`` -- any model with a Gaussian likelihood whose mean is
``dot(W, X.T)`` -- and the parameter-sized remainder is of the family ``bsc_blr_fused_update_general`` computes
(log noise variance as the second latent, a zero-mean Gaussian prior on the weights scaled by it, a
log-variance prior linear in xi and e^{-xi}), the whole update is the two launches of ``svi/blr.py``: ONE
pass over X (csrc/bsc_blr.hip) and the fused finish (ELBO, pathwise gradient, Adam, next draw).  Config 2
written with ``Normal`` / ``InverseGamma`` nodes is such a model: 0.17 ms per update instead of 0.85.
``route="auto"`` (default) takes the fused route when the model qualifies, ``"general"`` never does,
``"fused"`` insists (ValueError otherwise).

**The pass route.**  When the data term is recognised but the parameter-sized remainder is NOT of that family
(a known noise variance, another prior, more latents), the step still needs the data only through
Q_s = sum_n (y_n - x_n . w_s)^2 and G_s = sum_n (y_n - x_n . w_s) x_n: ONE ``bsc_blr_data_pass_sweep`` gives
both, and the remainder -- the recogniser's *surrogate* ``rest + coefficient * Q`` with Q an input -- is
parameter-sized work for the executor and autodiff (d surrogate / d Q is the coefficient, so
d log p / d w = d surrogate / d w - 2 (d surrogate / d Q) G).  Two skinny products over X and a dozen [S, N]
element-wise launches become one pass; the estimator and its noise are the general route's.  On the fused route the state (lam, Adam moments) lives on the
device, ``step()`` is asynchronous and returns None, ``elbo`` / ``grad`` / ``lam`` read back on access, and
the draws are ``bsc_blr_noise``'s (Philox streams 0 and 1, as ``oracle.svi.blr_sample``) -- with the
same seed the update equals ``oracle.svi.blr_step``.
"""
import math

import numpy as np

from ..algebra.autodiff import value_and_grad

_LOG_2PI = math.log(2.0 * math.pi)


def _outside_pass_envelope(X, y, D, S):
    """None, or why bsc_blr_data_pass would refuse these operands -- the checks of check_pass_args
    (csrc/bsc_blr.hip) restated, so that route="auto" falls back to the general route instead of raising at
    the first step."""
    if D > 256 or S > 64:
        return "outside the fused pass's envelope (D <= 256, S <= 64)"
    if D < 4 or D % 4:
        return "outside the fused pass's envelope (D = %d is not a multiple of 4)" % D
    if X.dim() != 2 or X.stride(1) != 1 or (X.shape[0] > 1 and (X.stride(0) < D or X.stride(0) % 4)):
        return "outside the fused pass's envelope (row-major X with a leading dimension >= D that is a multiple of 4)"
    if X.shape[0] > 1 and X.stride(0) >= 1 << 26:
        return "outside the fused pass's envelope (leading dimension of X below 2^26)"
    if X.data_ptr() % 16:
        return "outside the fused pass's envelope (X 16-byte aligned)"
    if y.dim() != 1 or (y.shape[0] > 1 and y.stride(0) != 1):
        return "outside the fused pass's envelope (contiguous y)"
    return None


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
    resident    : device backend only, general and pass routes (None, the default: ON for the pass route -- one term
                  away from config 2 a model then steps in 0.30 ms instead of 0.49 -- and off for the general one).  The state (lam, Adam moments), the draws, z = mu + e^rho eps,
                  the ELBO estimate, the pathwise gradient and the Adam step all stay on the device -- the
                  parameter-sized arithmetic as compiled algebra expressions in float64, the update by
                  ``bsc_adam_ascent`` -- so a step contains NO host synchronisation: ``step()`` returns None and
                  the walk of the next step overlaps the device's work on this one; ``elbo`` / ``grad`` / ``lam``
                  read back on access (as on the fused route).  Without it every step waits for the device twice
                  (the draws come back, the gradient comes back) and the host does the update in numpy.
    replay      : resident engines only (default on).  From its third step on the engine re-issues the RECORDED list of
                  C-ABI calls of a step -- draws to gradient, every launch whose arguments do not change -- instead of
                  walking the expression again (``DeviceBackend.replay_call``): the walk costs ~28 us of Python per
                  launch, the list ~2.  Unlike ``graph`` it needs no stream of the context's own; ``graph=True`` wins
                  when both are set.  ``set_data`` with new buffers records again.
    """

    def __init__(self, log_joint, latents, data, n_samples, seed=0, lr=1e-2, backend=None,
                 lam0=None, noise=None, graph=False, route="auto", resident=None, replay=True):
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
        self._fused = None
        self._lam = np.zeros(2 * self.P)
        if lam0 is None:
            self._lam[self.P:] = math.log(0.05)
        else:
            self._lam[:] = np.asarray(lam0, np.float64)
        self.m1, self.m2 = np.zeros_like(self._lam), np.zeros_like(self._lam)
        self._t = 0
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
        self._replay = bool(replay) and hasattr(self.backend, "replay_call")     # (used by the resident general route)
        self._z_dev = None
        self._elbo, self._grad = None, None
        self._pass_plan = None
        if route not in ("auto", "general", "fused"):
            raise ValueError("route must be 'auto', 'general' or 'fused'")
        self.route = "general"
        self.route_reason = None        # why route="auto" did not take a fused route (None: it did, or was not asked)
        self.plan = None
        if route != "general":
            from .recognise import guarded_route
            why = guarded_route(self._try_fused_route, strict=route == "fused")
            self.route_reason = why
            if why is not None and route == "fused":
                raise ValueError("route='fused': %s" % why)
        self._resident = None
        if resident is None:        # the pass route keeps its state on the device wherever it can
            resident = self._pass_plan is not None and self._noise is None and hasattr(self.backend, "ctx")
        if resident and self._fused is None:
            if self._noise is not None or not hasattr(self.backend, "ctx"):
                raise ValueError("resident=True needs the MI355X backend and its own device-side noise")
            self._init_resident()

    # -- the general route with its state on the device (class docstring, `resident`) -------------
    def _init_resident(self):
        import torch
        from .. import algebra as A
        b, S, P = self.backend, self.S, self.P
        dev = b.ctx.device
        f64 = dict(dtype=torch.float64, device=dev)
        st = {"lam": torch.from_numpy(np.asarray(self._lam, np.float64)).to(dev),
              "m1": torch.zeros(2 * P, **f64), "m2": torch.zeros(2 * P, **f64),
              "eps": torch.zeros((S, P), **f64), "elbo": None, "gmu": None, "grho": None}
        if len(self.latents) > 1:
            st["g"] = torch.zeros((S, P), dtype=torch.float32, device=dev)
        EPS, MU, RHO, G, F = A.var("eps", 2), A.var("mu", 1), A.var("rho", 1), A.var("g", 2), A.var("f", 1)
        sigma = A.dimshuffle(A.exp(RHO), "x", 0)
        st["z_fn"] = (A.dimshuffle(MU, "x", 0) + sigma * EPS).compile(b).device_fn
        st["gmu_fn"] = (A.sum(G, axis=0) * (1.0 / S)).compile(b).device_fn
        st["grho_fn"] = (A.sum(G * EPS, axis=0) * A.exp(RHO) * (1.0 / S) + 1.0).compile(b).device_fn
        st["elbo_fn"] = (A.sum(F) * (1.0 / S) + A.sum(RHO) + 0.5 * P * (1.0 + _LOG_2PI)).compile(b).device_fn
        self._resident = st
        if self._pass_plan is not None:
            # d Q_s / d w_s = -2 G_s joins the executor's gradient of the parameter-sized surrogate (all float64, [S, D])
            GW, C, GS = A.var("gw", 2), A.var("c", 1), A.var("G", 2)
            st["gw_fn"] = (GW + A.dimshuffle(C, 0, "x") * GS * (-2.0)).compile(b).device_fn
            self.route += ", state resident on the device"
        else:
            self.route = "general, state resident on the device"

    def _step_resident(self):
        import torch
        from ..algebra.device_backend import _DT, _i64
        b, st, S, P = self.backend, self._resident, self.S, self.P
        ctx = b.ctx
        self._t += 1
        ctx.call("bsc_philox_normal", self.seed, 2, int(self._t - 1), S, P, st["eps"])      # (its step counter changes)
        mu, rho = st["lam"][:P], st["lam"][P:]
        names = [v.name for v, _ in self.latents]
        fixed = self._graph or self._replay
        if fixed and self._z_dev is None:
            # fixed buffers for the draws: the walk (forward, then the tape backwards) is recorded once and replayed
            self._z_dev = {v.name: b.from_host(np.zeros((S, n), self._types[v.name][0]), *self._types[v.name])
                           for v, n in self.latents}

        def draws_to_gradient():
            """lam, eps -> z -> log p and its gradient -> ELBO estimate and pathwise gradient: every launch of a step
            whose arguments do not change from step to step."""
            z = b.materialize(st["z_fn"](eps=st["eps"], mu=mu, rho=rho))                  # float64 [S, P]
            inputs, offset = dict(self._data), 0
            for v, n in self.latents:
                if fixed:
                    zb = self._z_dev[v.name]
                    ctx.call("bsc_convert", _DT[z.dtype], _DT[zb.dtype], 2, _i64(zb.shape), z[:, offset:offset + n],
                             _i64((z.stride(0), 1)), zb, _i64(zb.stride()))
                    inputs[v.name] = zb
                else:
                    want = torch.float64 if str(np.dtype(self._types[v.name][0])) == "float64" else torch.float32
                    inputs[v.name] = b._convert(z[:, offset:offset + n], want)
                offset += n
            if self._pass_plan is not None:
                # the data term by ONE fused pass over X, y; the executor differentiates the parameter-sized surrogate
                plan = self._pass_plan
                X, y = self._data[plan.X], self._data[plan.y]
                ctx.call("bsc_blr_data_pass_sweep", X, X.stride(0), y, X.shape[0], X.shape[1], inputs[plan.W], S,
                         self._pass_Q, self._pass_G, 0)
                small = {name: inputs[name] for name in names}
                small[plan.Q_NAME] = b._convert(self._pass_Q, torch.float32)
                out, grads = value_and_grad(b, plan.surrogate, small, names + [plan.Q_NAME])
                f, gs = b.materialize(out), []
                for v, n in self.latents:
                    gv = grads.get(v.name)      # (None: a latent the parameter-sized part does not mention -- no prior on it)
                    gv = torch.zeros((S, n), dtype=torch.float64, device=X.device) if gv is None else \
                        b._convert(b.materialize(gv), torch.float64)
                    if v.name == plan.W:
                        c = b._convert(b.materialize(grads[plan.Q_NAME]), torch.float64)
                        gv = b.materialize(st["gw_fn"](gw=gv, c=c, G=self._pass_G))
                    gs.append(gv)
            elif self._graph:
                def walk():
                    out, grads = value_and_grad(b, self.log_joint, inputs, names)
                    return [out] + [grads[name] for name in names]
                res = b.graph_call(("reparam", id(self)), walk,
                                   [self._z_dev[name] for name in names] + list(self._data.values()))
                f, gs = res[0], list(res[1:])
            else:
                out, grads = value_and_grad(b, self.log_joint, inputs, names)
                f, gs = b.materialize(out), [b.materialize(grads[name]) for name in names]
            if len(self.latents) == 1:
                g = gs[0]
            else:
                g, offset = st["g"], 0
                for (v, n), gv in zip(self.latents, gs):            # (a strided copy through the C ABI: recordable)
                    ctx.call("bsc_convert", _DT[gv.dtype], _DT[g.dtype], 2, _i64(gv.shape), gv, _i64(gv.stride()),
                             g[:, offset:offset + n], _i64((g.stride(0), 1)))
                    offset += n
            gmu = b.materialize(st["gmu_fn"](g=g))
            if gmu.dtype != torch.float64:
                gmu = b._convert(gmu, torch.float64)
            grho = b.materialize(st["grho_fn"](g=g, eps=st["eps"], rho=rho))
            if grho.dtype != torch.float64:
                grho = b._convert(grho, torch.float64)
            return [b.materialize(st["elbo_fn"](f=f, rho=rho)), gmu, grho]

        if self._replay and not self._graph:
            # the third step records this region's C-ABI calls, later steps re-issue the list (DeviceBackend.replay_call):
            # ~2 us of host time per launch instead of the ~28 us of walking the expression again
            elbo, gmu, grho = b.replay_call(("reparam-step", id(self)), draws_to_gradient,
                                            [st["eps"], st["lam"]] + [self._z_dev[name] for name in names]
                                            + list(self._data.values()))
        else:
            elbo, gmu, grho = draws_to_gradient()
        st["elbo"], st["gmu"], st["grho"] = elbo, gmu, grho
        for lo, grad in ((0, gmu), (P, grho)):          # (element-wise: two calls on the halves are the one step)
            ctx.call("bsc_adam_ascent", st["lam"][lo:lo + P], grad, st["m1"][lo:lo + P], st["m2"][lo:lo + P], P,
                     int(self._t), self.lr, 0.9, 0.999, 1e-8)
        return None

    # -- the fused route (module docstring) ------------------------------------------------------
    def _try_fused_route(self):
        """Build the svi/blr.py driver behind this engine when the model qualifies; returns None, or the
        reason it does not."""
        from . import recognise
        if self._noise is not None:
            return "a caller-supplied noise source cannot drive the device-side sampler"
        if not hasattr(self.backend, "ctx"):
            return "the fused kernels run on the MI355X backend"
        shapes = {n: tuple(int(k) for k in v.shape) for n, v in self._data.items()}
        said = []
        plan = recognise.gaussian_linear(self.log_joint, self.latents, shapes, self.S, why=said)
        if plan is None:
            return ("the data do not enter the log-joint as coefficient_s * sum_n (y_n - x_n . w_s)^2: %s"
                    % (said[-1] if said else "no reason recorded"))
        self.plan = plan
        import torch
        X, y = self._data[plan.X], self._data[plan.y]
        D = int(X.shape[1])
        if not (isinstance(X, torch.Tensor) and X.dtype == torch.float32 and y.dtype == torch.float32):
            return "the fused pass streams float32 data"
        why = _outside_pass_envelope(X, y, D, self.S)
        if why is not None:
            return why
        self._planned_shape = (int(X.shape[0]), D)
        if plan.family is None:
            # the pass route: data term by the fused pass, the rest by the executor (module docstring)
            self._pass_plan = plan
            self._pass_Q = torch.zeros(self.S, dtype=torch.float64, device=X.device)
            self._pass_G = torch.zeros((self.S, D), dtype=torch.float64, device=X.device)
            self.backend.ctx.reserve((4 * self.backend.ctx.info()["cu_count"] + 8) * (8 * 256 + 8) * 4 * 2)
            self.route = "pass: bsc_blr_data_pass_sweep + executor on the parameter-sized surrogate"
            return ("Gaussian-linear data term recognised (one pass over X per step), but the parameter-sized part "
                    "is not of the family c0 + c_xi xi + e^{-xi} (-s_q Q / 2 - k_w |w|^2 / 2 - beta) with one "
                    "scalar latent xi")
        from ..svi.blr import BLRReparamSVI
        c0, c_xi, s_q, k_w, beta, xi_name = plan.family
        self._order = [v.name for v, _ in self.latents]          # [W, xi] or [xi, W]
        self._w_first = self._order[0] == plan.W
        self._fused = BLRReparamSVI(X, y, n_samples=self.S, seed=self.seed, lr=self.lr, ctx=self.backend.ctx,
                                    lam0=self._to_blr_layout(self._lam), family=(c0, c_xi, s_q, k_w, beta))
        self._fused_D = D
        self.route = "fused: bsc_blr_data_pass + bsc_blr_fused_update_general"
        return None

    def _to_blr_layout(self, lam):
        """[mu (P) | rho (P)] in the order of ``latents`` -> svi/blr.py's [m (D) | rho (D) | a | b]."""
        P, D = self.P, self.P - 1
        mu, rho = np.asarray(lam[:P], np.float64), np.asarray(lam[P:], np.float64)
        w = slice(0, D) if self._w_first else slice(1, P)
        x = D if self._w_first else 0
        return np.concatenate([mu[w], rho[w], [mu[x]], [rho[x]]])

    def _from_blr_layout(self, v):
        P, D = self.P, self.P - 1
        v = np.asarray(v, np.float64)
        m, rho, a, b = v[:D], v[D:2 * D], v[2 * D], v[2 * D + 1]
        if self._w_first:
            return np.concatenate([m, [a], rho, [b]])
        return np.concatenate([[a], m, [b], rho])

    @property
    def lam(self):
        if self._fused is not None:
            return self._from_blr_layout(self._fused.lam.cpu().numpy())
        if getattr(self, "_resident", None) is not None:
            return self._resident["lam"].cpu().numpy()
        return self._lam

    @lam.setter
    def lam(self, value):
        if self._fused is not None:
            raise AttributeError("on the fused route the variational parameters live on the device; build the "
                                 "engine with lam0=")
        if getattr(self, "_resident", None) is not None:
            import torch
            self._resident["lam"].copy_(torch.from_numpy(np.asarray(value, np.float64)))
            return
        self._lam = value

    @property
    def t(self):
        return self._fused.t if self._fused is not None else self._t

    @t.setter
    def t(self, value):
        self._t = value

    @property
    def elbo(self):
        """Monte-Carlo ELBO estimate of the last step (fused route: reads the device scalar, synchronises)."""
        if self._fused is not None:
            return float(self._fused.elbo.item()) if self._fused.t else None
        if getattr(self, "_resident", None) is not None:
            e = self._resident["elbo"]
            return None if e is None else float(e.reshape(-1)[0].item())
        return self._elbo

    @elbo.setter
    def elbo(self, value):
        self._elbo = value

    @property
    def grad(self):
        if self._fused is not None:
            return self._from_blr_layout(self._fused.grad.cpu().numpy()) if self._fused.t else None
        if getattr(self, "_resident", None) is not None:
            st = self._resident
            return None if st["gmu"] is None else np.concatenate([st["gmu"].cpu().numpy(), st["grho"].cpu().numpy()])
        return self._grad

    @grad.setter
    def grad(self, value):
        self._grad = value

    def set_data(self, **arrays):
        """Replace data inputs (the next mini-batch; write the data term times N / B)."""
        for name in arrays:
            if name not in self._types or name in {v.name for v, _ in self.latents}:
                raise TypeError("%s is not a data input of the log-joint" % name)
        previous = {name: self._data[name] for name in arrays if name in self._data}
        for name, value in arrays.items():
            self._data[name] = self.backend.from_host(value, *self._types[name])     # (a new buffer: a recorded graph is dropped)
        if self._fused is not None or self._pass_plan is not None:
            # fused route AND pass route: every shape(X, 0) of the log-joint was resolved to a constant when the model
            # was recognised (recognise.normalise), so another row count would keep the old N in the normaliser terms
            X, y = self._data[self.plan.X], self._data[self.plan.y]
            if tuple(X.shape) != self._planned_shape or tuple(y.shape) != (self._planned_shape[0],):
                for name, value in previous.items():
                    self._data[name] = value
                raise ValueError("the %s route was planned for mini-batches of %d x %d (the mini-batch extent is part of "
                                 "the recognised coefficients); build a new engine for another batch size"
                                 % (("fused" if self._fused is not None else "pass",) + self._planned_shape))
            why = _outside_pass_envelope(X, y, self._planned_shape[1], self.S)
            if why is not None:
                for name, value in previous.items():
                    self._data[name] = value
                raise ValueError("set_data: " + why)
            if self._fused is not None:
                self._fused.set_batch(X, y)

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

    def _log_joint_and_gradient_pass(self, z):
        """The same with the data term through ONE fused pass (module docstring, "the pass route")."""
        import torch
        plan, b = self._pass_plan, self.backend
        inputs, offset, W_dev = {}, 0, None
        for v, n in self.latents:
            block = np.ascontiguousarray(z[:, offset:offset + n], dtype=np.float32)
            inputs[v.name] = b.from_host(block, "float32", 2)
            if v.name == plan.W:
                W_dev = inputs[v.name]
            offset += n
        X, y = self._data[plan.X], self._data[plan.y]
        b.ctx.call("bsc_blr_data_pass_sweep", X, X.stride(0), y, X.shape[0], X.shape[1], W_dev, self.S,
                   self._pass_Q, self._pass_G, 0)
        inputs[plan.Q_NAME] = b._convert(self._pass_Q, torch.float32)
        names = [v.name for v, _ in self.latents]
        out, grads = value_and_grad(b, plan.surrogate, inputs, names + [plan.Q_NAME])
        f = np.asarray(b.to_host(out), np.float64).reshape(self.S)
        coefficient = np.asarray(b.to_host(grads[plan.Q_NAME]), np.float64).reshape(self.S)
        G = self._pass_G.cpu().numpy()
        blocks = []
        for v, n in self.latents:
            gv = grads.get(v.name)           # (a latent the parameter-sized part does not mention: no prior on it)
            gv = np.zeros((self.S, n)) if gv is None else np.asarray(b.to_host(gv), np.float64).reshape(self.S, n)
            if v.name == plan.W:
                gv = gv - 2.0 * coefficient[:, None] * G          # d Q_s / d w_s = -2 G_s
            blocks.append(gv)
        return f, np.concatenate(blocks, axis=1)

    def log_joint_and_gradient(self, z):
        """(log p(data, z_s) [S], d log p / d z [S, P]) through the executor."""
        if self._pass_plan is not None:
            return self._log_joint_and_gradient_pass(z)
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
        if self._fused is not None:
            self._fused.step()          # asynchronous: pass + fused finish on the context's stream
            return None
        if self._resident is not None:
            return self._step_resident()
        self.t += 1
        self.elbo, self.grad = self.estimate(self.t - 1)
        b1, b2, eps = 0.9, 0.999, 1e-8
        self.m1 = b1 * self.m1 + (1 - b1) * self.grad
        self.m2 = b2 * self.m2 + (1 - b2) * self.grad ** 2
        mhat = self.m1 / (1 - b1 ** self.t)
        vhat = self.m2 / (1 - b2 ** self.t)
        self.lam = self.lam + self.lr * mhat / (np.sqrt(vhat) + eps)
        return self.elbo
