"""Reparameterised-ELBO SVI for Bayesian linear regression (BASELINE config 2).

Host-side driver of the device path; every numeric step is a C-ABI call into
libbayesic_hip.so.  The reference has no inference code -- this implements
README.md:51 (reparameterisation-trick gradient, refs [10][11][12]) with
mini-batch scaling per README.md:69-79, on the likelihood decomposition of
bayesic/distribution/base.py:47-69.

Model:  y_n ~ N(x_n.w, s2),  w | s2 ~ N(0, s2 I),  s2 ~ InvGamma(alpha0, beta0)
q:      w ~ N(m, diag e^{2 rho}),  log s2 ~ N(a, e^{2b});  lam = [m, rho, a, b].

``sweep="alternate"`` (default): the passes over a resident mini-batch alternate their direction
so that each starts in the rows the previous one left in the Infinity Cache (158 instead of 164 us
at 1M x 256; a shard below 256 MiB is read from the cache entirely).  ``sweep="stream"`` always walks
forward with non-temporal loads.

One update is two launches on one GPU -- the streaming data pass and a fused
finish (float64 reduction of the pass partials, ELBO + pathwise gradient, Adam
step, next step's Philox draws) -- with lam and the draws double-buffered.

``reproducible=True``: results are bit-identical on 1, 2, 4 and 8 GPUs.
The global mini-batch is cut into V = 8 virtual shards of equal size; a rank runs one data pass
per shard it holds (the pass's workgroup partition then depends on the shard alone, not on p),
the ranks all-reduce a [V, S(D+1)] float64 buffer in which every row is non-zero on exactly one
rank (adding zeros is exact, so the collective's order does not matter), and every rank adds
the V rows in shard order with one n-ary bsc_elemwise launch.  Off by default: on one GPU it is
eight 125k-row passes instead of one 1M-row pass.

Data parallelism: each rank holds a contiguous block of mini-batch rows.  The
only exchange per update is ONE all-reduce(sum) of the float64 vector
[Q (S), G (S*D)] (16 KB at S=8, D=256) between the pass and the finish; noise is
keyed by (seed, step, sample, parameter) and never by rank, so all ranks apply
the identical update.
"""
import math

import torch

from ..device import default_context
from .exchange import Exchange

# include/bayesic_hip.h: BSC_SWEEP_*
SWEEP_STREAM, SWEEP_FORWARD_KEEP, SWEEP_BACKWARD_KEEP = 0, 1, 2


class BLRReparamSVI:
    NOISE_BLOCK = 32

    VIRTUAL_SHARDS = 8

    def __init__(self, X, y, n_total=None, n_samples=8, seed=1234, lr=1e-2, alpha0=1.0,
                 beta0=1.0, ctx=None, group=None, lam0=None, fused=True, reproducible=False,
                 sweep="alternate", family=None):
        """``family`` = (c0, c_xi, s_q, k_w, beta): the log-joint per draw as a member of
        f(w, xi; Q) = c0 + c_xi xi + e^{-xi} (-s_q Q / 2 - k_w |w|^2 / 2 - beta) instead of the
        Normal-InverseGamma model above (bsc_blr_fused_update_general) -- what ``inference.ReparamVI``
        passes when it has recognised such a model in a symbolic log-joint."""
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
        self._Xarg, self._yarg, self._ldx = self.X, self.y, self.X.stride(0)
        self.S = int(n_samples)
        self.seed = int(seed)
        self.lr = float(lr)
        self.alpha0, self.beta0 = float(alpha0), float(beta0)
        self.family = None if family is None else tuple(float(v) for v in family)
        if self.family is not None and len(self.family) != 5:
            raise ValueError("family must be (c0, c_xi, s_q, k_w, beta)")
        self.group = group
        self.exchange = Exchange(self.ctx, group)   # RCCL behind the C ABI when ctx has a communicator
        self.world = self.exchange.world
        # global mini-batch rows (all ranks); ranks may hold unequal blocks
        self.batch_rows = self.exchange.global_count(self.B, dev)
        self.n_total = float(n_total) if n_total is not None else self.batch_rows
        self.fused = bool(fused)
        # one launch per update (bsc_blr_pass_update: the finish in the pass's tail) where the library offers the entry
        # point; a test double without it, or one_launch = False, takes the two launches
        self.one_launch = hasattr(getattr(self.ctx, "lib", None), "bsc_blr_pass_update")
        self.reproducible = bool(reproducible)
        if sweep not in ("alternate", "stream"):
            raise ValueError("sweep must be 'alternate' or 'stream'")
        # "alternate": successive passes over the SAME resident mini-batch walk it forward, then
        # backward, ..., each leaving the rows it read last in the 256 MiB Infinity Cache for the
        # pass that starts there (include/bayesic_hip.h, bsc_blr_data_pass_sweep).  A batch that
        # has just been swapped in (set_batch) is streamed on its first pass.
        self.sweep = sweep
        self._sweep_next = SWEEP_FORWARD_KEEP
        self._fresh_batch = False
        if self.reproducible:
            V = self.VIRTUAL_SHARDS
            total = int(round(self.batch_rows))
            if total % V != 0:
                raise ValueError("reproducible=True needs the global mini-batch (%d rows) to be a multiple "
                                 "of %d virtual shards" % (total, V))
            self._shard_rows = total // V
            first = self.exchange.row_offset(self.B, dev)
            if first % self._shard_rows != 0 or self.B % self._shard_rows != 0:
                raise ValueError("reproducible=True needs every rank to hold whole virtual shards of %d rows "
                                 "(this rank: rows %d..%d)" % (self._shard_rows, first, first + self.B))
            self._first_shard = first // self._shard_rows
            self._n_shards = self.B // self._shard_rows
        D, S = self.D, self.S
        f64 = torch.float64
        # double-buffered state: index t & 1 is current at the start of step t+1
        self._lam = torch.zeros((2, 2 * D + 2), dtype=f64, device=dev)
        if lam0 is None:
            self._lam[0, D:2 * D] = math.log(0.1)
            self._lam[0, 2 * D + 1] = math.log(0.1)
        else:
            self._lam[0].copy_(torch.as_tensor(lam0, dtype=f64))
        # noise ring: NOISE_BLOCK steps are drawn per launch, two blocks resident
        self._ring = 2 * self.NOISE_BLOCK
        self._eps = torch.zeros((self._ring, S * (D + 1)), dtype=f64, device=dev)
        self._noise_upto = 0   # noise of Philox steps [0, _noise_upto) has been requested
        self._W = torch.zeros((2, S * D), dtype=torch.float32, device=dev)
        self._xi = torch.zeros((2, S), dtype=f64, device=dev)
        self.m1 = torch.zeros(2 * D + 2, dtype=f64, device=dev)
        self.m2 = torch.zeros(2 * D + 2, dtype=f64, device=dev)
        self.grad = torch.zeros(2 * D + 2, dtype=f64, device=dev)
        self.elbo = torch.zeros(1, dtype=f64, device=dev)
        self.stats = torch.zeros(S * (D + 1), dtype=f64, device=dev)  # [Q | G]
        if self.reproducible:
            self._vstats = torch.zeros((self.VIRTUAL_SHARDS, S * (D + 1)), dtype=f64, device=dev)
        self.Q = self.stats[:S]
        self.G = self.stats[S:]
        self.t = 0
        self._drawn = False
        # size the slab once so step() never allocates
        self.ctx.reserve((4 * self.ctx.info()["cu_count"] + 8) * (8 * 256 + 8) * 4)

    def set_batch(self, X, y, rows=None, ldx=None):
        """Point the next update at another device-resident mini-batch of the same width:
        torch tensors, or raw device pointers with `rows` (and `ldx`, default D) -- what
        MiniBatchLoader.acquire() returns.  The mini-batch scaling n_total / batch_rows
        keeps the batch size the model was built with."""
        if isinstance(X, torch.Tensor):
            if X.dtype != torch.float32 or y.dtype != torch.float32 or X.dim() != 2 or \
                    X.shape[1] != self.D or X.stride(1) != 1 or y.shape[0] != X.shape[0]:
                raise ValueError("batch must be float32 X [rows, %d] row-major and y [rows]" % self.D)
            self.X, self.y = X, y
            self._Xarg, self._yarg, self._ldx, self.B = X, y, X.stride(0), X.shape[0]
            self._fresh_batch = True
        else:
            if rows is None:
                raise ValueError("raw device pointers need `rows`")
            self.X = self.y = None
            self._Xarg, self._yarg = int(X), int(y)
            self._ldx, self.B = int(ldx if ldx is not None else self.D), int(rows)
            self._fresh_batch = True

    def _take_sweep(self):
        """Sweep order of the pass about to be issued (and book-keeping for the one after)."""
        if self.sweep != "alternate" or self.reproducible:
            return SWEEP_STREAM
        if self._fresh_batch:
            # nothing of this batch is cached and it may never be read again: stream it; if it IS
            # read again, that pass walks back from the end
            self._fresh_batch = False
            self._sweep_next = SWEEP_BACKWARD_KEEP
            return SWEEP_STREAM
        code = self._sweep_next
        if self._passes_per_update() & 1:   # an odd number of passes ends at the other side
            self._sweep_next = 3 - code
        return code

    def _passes_per_update(self):
        """Launches of the pass kernel per update, as bsc_blr_data_pass issues them (asked of the library
        once per batch pointer: eight draws per pass, or sixteen while more than eight are left at D = 256)."""
        if self.S <= 8:
            return 1
        key = self._yarg if isinstance(self._yarg, int) else self._yarg.data_ptr()
        if getattr(self, "_pass_count", (None, 0))[0] != key:
            if hasattr(self.ctx, "lib"):
                import ctypes
                n = ctypes.c_int32(0)
                self.ctx.call("bsc_blr_pass_count", self._yarg, self.D, self.S, ctypes.byref(n))
                self._pass_count = (key, int(n.value))
            else:                                   # a test double without the library: eight per pass
                self._pass_count = (key, (self.S + 7) // 8)
        return self._pass_count[1]

    # -- current views ---------------------------------------------------------
    @property
    def cur(self):
        return self.t & 1

    @property
    def lam(self):
        return self._lam[self.cur]

    @property
    def W(self):
        return self._W[self.cur]

    @property
    def eps(self):
        return self._eps[self.t % self._ring]

    def _ensure_noise(self, step):
        """Noise of Philox step `step` is in ring row step % ring (drawn a block ahead)."""
        nb = self.NOISE_BLOCK
        while self._noise_upto <= step:
            start = self._noise_upto
            r0 = start % self._ring
            self.ctx.call("bsc_blr_noise", self.D, self.S, self.seed, start, nb,
                          self._eps[r0:r0 + nb])
            self._noise_upto = start + nb

    @property
    def xi(self):
        return self._xi[self.cur]

    # -- unfused phases (also the multi-sample-group path) ---------------------
    def sample(self, step):
        c = self.cur
        self._ensure_noise(step)        # (bsc_blr_sample rewrites this row with the same values)
        self.ctx.call("bsc_blr_sample", self._lam[c], self.D, self.S, self.seed, step,
                      self._eps[step % self._ring], self._W[c], self._xi[c])
        self._drawn = True

    def data_pass(self):
        if self.reproducible:
            return self._data_pass_by_shard()
        self.ctx.call("bsc_blr_data_pass_sweep", self._Xarg, self._ldx, self._yarg, self.B,
                      self.D, self.W, self.S, self.Q, self.G, self._take_sweep())

    def _data_pass_by_shard(self):
        """One pass per virtual shard this rank holds, each into its own row of the [V, n] buffer
        (the other rows stay zero until the all-reduce)."""
        if self.X is None:
            raise ValueError("reproducible=True needs tensor batches (set_batch with raw pointers is not sharded)")
        S, rows = self.S, self._shard_rows
        self.ctx.call("bsc_memset", self._vstats, 0, self._vstats.numel() * 8)
        for k in range(self._n_shards):
            r0 = k * rows
            out = self._vstats[self._first_shard + k]
            self.ctx.call("bsc_blr_data_pass", self.X[r0:r0 + rows], self._ldx, self.y[r0:r0 + rows], rows,
                          self.D, self.W, S, out[:S], out[S:])

    def all_reduce(self):
        if not self.reproducible:
            self.exchange.all_reduce(self.stats)
            return
        import ctypes
        self.exchange.all_reduce(self._vstats)          # every row is non-zero on one rank only: exact
        V, n = self._vstats.shape
        i64 = lambda v: (ctypes.c_int64 * len(v))(*v)
        ptrs = (ctypes.c_void_p * V)(*[self._vstats[v].data_ptr() for v in range(V)])
        # stats = ((V0 + V1) + V2) + ... in shard order: one n-ary add, the same on every rank count
        self.ctx.call("bsc_elemwise", 0, 1, 1, i64([n]), self.stats, i64([1]), V, ptrs, i64([1] * V))

    def _finish(self, stats):
        """Fused gradient + Adam + next draw; flips the double buffer."""
        c, n = self.cur, 1 - self.cur
        t = self.t + 1                  # Adam step count; Philox step of the NEXT draw
        self._ensure_noise(t)
        if self.family is None:
            self.ctx.call("bsc_blr_fused_update", stats,
                          self._lam[c], self._lam[n], self.m1, self.m2,
                          self._eps[self.t % self._ring], self._W[c], self._xi[c], self.D, self.S,
                          self.batch_rows, self.n_total / self.batch_rows, self.alpha0, self.beta0,
                          t, self.lr, 0.9, 0.999, 1e-8, self.seed, t,
                          self._eps[t % self._ring], 1, self._W[n], self._xi[n], self.elbo,
                          self.grad)
        else:
            c0, c_xi, s_q, k_w, beta = self.family
            self.ctx.call("bsc_blr_fused_update_general", stats,
                          self._lam[c], self._lam[n], self.m1, self.m2,
                          self._eps[self.t % self._ring], self._W[c], self._xi[c], self.D, self.S,
                          c0, c_xi, s_q, k_w, beta,
                          t, self.lr, 0.9, 0.999, 1e-8, self.seed, t,
                          self._eps[t % self._ring], 1, self._W[n], self._xi[n], self.elbo,
                          self.grad)
        self.t = t

    def _pass_update(self):
        """bsc_blr_pass_update[_general]: data pass + gradient + Adam + next draw as ONE launch; flips the double buffer."""
        c, n = self.cur, 1 - self.cur
        t = self.t + 1                  # Adam step count; Philox step of the NEXT draw
        self._ensure_noise(t)
        head = (self._Xarg, self._ldx, self._yarg, self.B, self.D, self._take_sweep(),
                self._lam[c], self._lam[n], self.m1, self.m2, self._eps[self.t % self._ring], self._W[c], self._xi[c],
                self.S)
        tail = (t, self.lr, 0.9, 0.999, 1e-8, self.seed, t, self._eps[t % self._ring], 1, self._W[n], self._xi[n],
                self.elbo, self.grad)
        if self.family is None:
            self.ctx.call("bsc_blr_pass_update", *head, self.batch_rows, self.n_total / self.batch_rows, self.alpha0,
                          self.beta0, *tail)
        else:
            self.ctx.call("bsc_blr_pass_update_general", *head, *self.family, *tail)
        self.t = t

    def step(self):
        """One ELBO-gradient update; asynchronous on the context stream."""
        if not self._drawn:
            self.sample(self.t)  # Philox step index == number of completed updates
        if self.fused and self.world == 1 and not self.exchange.rccl and self.S <= 8 and not self.reproducible:
            if self.one_launch:
                self._pass_update()         # the pass with the finish in its tail: one launch (falls back inside the library)
            else:
                self.ctx.call("bsc_blr_data_pass_partial_sweep", self._Xarg, self._ldx,
                              self._yarg, self.B, self.D, self.W, self.S, self._take_sweep())
                self._finish(None)
        else:
            self.data_pass()
            self.all_reduce()
            self._finish(self.stats)

    # -- host views -----------------------------------------------------------
    def params(self):
        lam = self.lam.cpu().numpy()
        D = self.D
        return dict(m=lam[:D], rho=lam[D:2 * D], a=lam[2 * D], b=lam[2 * D + 1])
