"""SVI for an LDA-style Dirichlet-Multinomial model with a fixed-gamma local
step (BASELINE config 4).

The data-sized work is the pair of contractions that bayesic.algebra lowers to
GEMMs (cfg 4):

    sstats = Bt * dot(Th.T, C / dot(Th, Bt))

Two device paths compute it.  ``via="kernel"`` (default when K is 32, 64, 96 or
128) is bsc_lda_sstats: both contractions and the division in one pass over C,
no docs x V intermediate.  ``via="executor"`` builds the expression ONCE with
the algebra front end and evaluates it on the MI355X backend (two fp32 MFMA
GEMMs + fused element-wise launches) -- any K, and the cross-check of the fused
kernel.  ``via="csc"`` (default when C is a scipy.sparse matrix) keeps the counts
in compressed-sparse-column form and walks only the nonzeros (bsc_lda_sstats_csc).
The Dirichlet expectation and the natural-gradient step on lambda [K, V] are their
own kernels.  Data-parallel over documents: one all-reduce of sstats (K*V float32,
51.2 MB at K=128, V=100k) per update -- taken and exchanged in column ranges (whole rounds of the
persistent kernel, ``_step_overlapped``): the collective of a finished range runs on a second stream
beside the kernel of the next, only the last range's is exposed (``overlap=False``: one collective
after the whole statistic, bit-identical results).

``self.elbo`` (device, float64) holds, after every ``step()``, the mini-batch estimate of the evidence
lower bound at the lambda the step started from (README.md:30-37, 69-79; oracle.svi.lda_elbo):

    (docs_total / docs) * [sum_dv C log(phinorm) - sum_d KL(Dir(gamma_d) || Dir(alpha))]
        - sum_k KL(Dir(lambda_k) || Dir(eta))

The words' term is accumulated INSIDE the statistic kernel, where phinorm lives in registers
(bsc_lda_sstats_bound); the topics' term comes out of the Dirichlet-expectation kernel of lambda
(bsc_dirichlet_expectation_bound); the documents' term is constant (gamma is fixed) and taken once.
Data-parallel: one more all-reduce of ONE float64 per update.  ``elbo=False`` skips all of it.
"""
import torch

from .. import algebra as A
from ..algebra.device_backend import DeviceBackend
from ..device import default_context
from .exchange import Exchange


class LDAFixedGammaSVI:
    def __init__(self, C, gamma, lam0, eta=0.01, docs_total=None, ctx=None, group=None, via=None,
                 alpha=None, elbo=True, overlap=True):
        self.ctx = ctx or default_context()
        dev = self.ctx.device
        f32 = torch.float32
        self._csc = None
        try:
            import scipy.sparse as sparse
            is_sparse = sparse.issparse(C)
        except ImportError:
            is_sparse = False
        if is_sparse:
            if via not in (None, "csc"):
                raise ValueError("sparse counts need via='csc'")
            via = "csc"
            csc = C.tocsc()
            csc.sum_duplicates()
            self._csc = (torch.as_tensor(csc.indptr.astype("int64")).to(dev),
                         torch.as_tensor(csc.indices.astype("int32")).to(dev),
                         torch.as_tensor(csc.data.astype("float32")).to(dev))
            self.C = None
            shape_C = csc.shape
        else:
            self.C = C if isinstance(C, torch.Tensor) else torch.as_tensor(C, dtype=f32).to(dev)
            shape_C = tuple(self.C.shape)
        gamma = gamma if isinstance(gamma, torch.Tensor) else torch.as_tensor(gamma, dtype=f32).to(dev)
        self.lam = (lam0 if isinstance(lam0, torch.Tensor)
                    else torch.as_tensor(lam0, dtype=f32).to(dev)).clone().contiguous()
        if (self.C is not None and self.C.dtype != f32) or gamma.dtype != f32 or self.lam.dtype != f32:
            raise TypeError("C, gamma and lambda must be float32")
        self.docs, self.V = shape_C
        self.K = self.lam.shape[0]
        if gamma.shape != (self.docs, self.K) or self.lam.shape != (self.K, self.V):
            raise ValueError("shapes: C [docs, V], gamma [docs, K], lambda [K, V]")
        self.eta = float(eta)
        self.group = group
        self.exchange = Exchange(self.ctx, group)   # RCCL behind the C ABI when ctx has a communicator
        self.world = self.exchange.world
        self.batch_docs = self.exchange.global_count(self.docs, dev)
        self.docs_total = float(docs_total) if docs_total is not None else self.batch_docs
        self.Th = torch.empty((self.docs, self.K), dtype=f32, device=dev)
        self.Bt = torch.empty((self.K, self.V), dtype=f32, device=dev)
        self.alpha = float(alpha) if alpha is not None else 1.0 / self.K      # the documents' Dirichlet prior
        self.with_elbo = bool(elbo)
        self.overlap = bool(overlap)    # data-parallel, via="kernel": the statistic's all-reduce in pieces, overlapped
        self._pieces = None
        f64 = torch.float64
        # [words' term | documents' term]: the two per-rank sums one all-reduce carries
        self._local = torch.zeros(2, dtype=f64, device=dev)
        self._ll, self._theta_bound = self._local[:1], self._local[1:]
        self._beta_bound = torch.zeros(1, dtype=f64, device=dev)
        self.elbo = torch.zeros(1, dtype=f64, device=dev)
        if self.with_elbo:      # gamma is fixed: Th and -sum_d KL(Dir(gamma_d) || Dir(alpha)) are computed once
            self.ctx.call("bsc_dirichlet_expectation_bound", gamma.contiguous(), self.docs, self.K, self.K,
                          self.alpha, self.Th, self._theta_bound)
            self._theta_local = self._theta_bound.clone()
        else:
            self.ctx.call("bsc_dirichlet_expectation", gamma.contiguous(), self.docs, self.K, self.K,
                          self.Th)
        if via is None:
            via = "kernel" if self.K in (32, 64, 96, 128) else "executor"
        if via not in ("kernel", "executor", "csc"):
            raise ValueError("via must be 'kernel', 'executor' or 'csc'")
        if via == "csc" and self._csc is None:
            raise ValueError("via='csc' needs a scipy.sparse count matrix")
        self.via = via
        if via == "executor":
            Th, Cv, Bm = A.var("Th", 2), A.var("C", 2), A.var("Bm", 2)
            self.expr = Bm * A.dot(Th.T, Cv / A.dot(Th, Bm))
            self.backend = DeviceBackend(self.ctx)
            self._sstats_fn = self.expr.compile(self.backend).device_fn
            # the words' term as an expression of its own (a third product over C: this route is the
            # general one, not the fast one); counts of zero contribute nothing whatever phinorm is
            self._ll_fn = A.sum(Cv * A.log(A.dot(Th, Bm))).compile(self.backend).device_fn
        if self.C is not None and self.C.stride(1) != 1:
            self.C = self.C.contiguous()
        self.sstats = torch.zeros((self.K, self.V), dtype=f32, device=dev)
        self.t = 0

    def local_step(self):
        """sstats = Bt * dot(Th.T, C / dot(Th, Bt)) for the current lambda (and, with ``elbo``, the words'
        and the topics' terms of the bound at it)."""
        if not self.with_elbo:
            self.ctx.call("bsc_dirichlet_expectation", self.lam, self.K, self.V, self.V, self.Bt)
            if self.via == "csc":
                colptr, rowidx, vals = self._csc
                self.ctx.call("bsc_lda_sstats_csc", colptr, rowidx, vals, self.docs, self.V, self.K,
                              self.Th, self.K, self.Bt, self.V, self.sstats, self.V)
            elif self.via == "kernel":
                self.ctx.call("bsc_lda_sstats", self.C, self.C.stride(0), self.docs, self.V, self.K,
                              self.Th, self.K, self.Bt, self.V, self.sstats, self.V)
            else:
                self.sstats = self._sstats_fn(Th=self.Th, C=self.C, Bm=self.Bt)
            return self.sstats
        self.ctx.call("bsc_dirichlet_expectation_bound", self.lam, self.K, self.V, self.V, self.eta, self.Bt,
                      self._beta_bound)
        if self.via == "csc":
            colptr, rowidx, vals = self._csc
            self.ctx.call("bsc_lda_sstats_csc_bound", colptr, rowidx, vals, self.docs, self.V, self.K,
                          self.Th, self.K, self.Bt, self.V, self.sstats, self.V, self._ll)
        elif self.via == "kernel":
            self.ctx.call("bsc_lda_sstats_bound", self.C, self.C.stride(0), self.docs, self.V, self.K,
                          self.Th, self.K, self.Bt, self.V, self.sstats, self.V, self._ll)
        else:
            self.sstats = self._sstats_fn(Th=self.Th, C=self.C, Bm=self.Bt)
            ll = self.backend.materialize(self._ll_fn(Th=self.Th, C=self.C, Bm=self.Bt))
            self._ll.copy_(self.backend._convert(ll.reshape(1), torch.float64))
        return self.sstats

    # -- the update with its collective overlapped (data-parallel, via="kernel") -------------------------------------
    def _plan_pieces(self):
        """Column ranges [(c0, cols)] the statistic is taken and all-reduced in: whole rounds of the persistent
        kernel (bsc_lda_sstats_round_columns -- a range that starts at a multiple of a round gives every column block
        the schedule it has in the one-call statistic: bit-identical), the rest as the last piece."""
        import ctypes
        cols = ctypes.c_int64()
        self.ctx.call("bsc_lda_sstats_round_columns", self.K, ctypes.byref(cols))
        step = int(cols.value)
        pieces = [(c0, min(step, self.V - c0)) for c0 in range(0, self.V, step)]
        if len(pieces) > 15:            # (bsc_allreduce_sum_begin has 16 slots; the last is the bound's)
            pieces = pieces[:14] + [(pieces[14][0], self.V - pieces[14][0])]
        return pieces

    def _step_overlapped(self, rho):
        """step() for more than one rank: piece i of the statistic goes to a contiguous staging block [K, cols_i]
        and its all-reduce starts (second stream) while the kernel of piece i + 1 runs; the natural-gradient step of a
        piece waits for that piece's collective only.  Exposed: the LAST piece's collective (17.7 of 51.2 MB at
        V = 100 000 on 256 CUs) instead of all of it."""
        f32, f64 = torch.float32, torch.float64
        if getattr(self, "_pieces", None) is None:
            self._pieces = self._plan_pieces()
            self._stage = torch.empty(self.K * self.V, dtype=f32, device=self.lam.device)
            # [words' term per piece ... | documents' term]: one small all-reduce carries them all
            self._local_pieces = torch.zeros(len(self._pieces) + 1, dtype=f64, device=self.lam.device)
        n = len(self._pieces)
        ctx, K, V = self.ctx, self.K, self.V
        if self.with_elbo:
            ctx.call("bsc_dirichlet_expectation_bound", self.lam, K, V, V, self.eta, self.Bt, self._beta_bound)
        else:
            ctx.call("bsc_dirichlet_expectation", self.lam, K, V, V, self.Bt)
        blocks, off = [], 0
        for i, (c0, cols) in enumerate(self._pieces):
            block = self._stage[off:off + K * cols]
            off += K * cols
            blocks.append(block)
            if self.with_elbo:
                ctx.call("bsc_lda_sstats_bound", self.C[:, c0:], self.C.stride(0), self.docs, cols, K, self.Th, K,
                         self.Bt[:, c0:], V, block, cols, self._local_pieces[i:i + 1])
            else:
                ctx.call("bsc_lda_sstats", self.C[:, c0:], self.C.stride(0), self.docs, cols, K, self.Th, K,
                         self.Bt[:, c0:], V, block, cols)
            self.exchange.begin(block, i)
        if self.with_elbo:
            self._local_pieces[n:].copy_(self._theta_local)         # (the all-reduce sums in place)
            self.exchange.begin(self._local_pieces, 15)
        scale = self.docs_total / self.batch_docs
        for i, (c0, cols) in enumerate(self._pieces):
            self.exchange.end(i)
            last = i == n - 1 and self.with_elbo
            if last:
                self.exchange.end(15)
            ctx.call("bsc_natgrad_update_f32_2d", self.lam[:, c0:], V, self.eta, blocks[i], cols, K, cols, scale,
                     float(rho), self._local_pieces if last else None, n if last else 0,
                     self._local_pieces[n:] if last else None, self._beta_bound if last else None,
                     self.elbo if last else None)

    def step(self, rho=None):
        self.t += 1
        if rho is None:
            rho = (self.t + 1.0) ** -0.7
        if self.overlap and self.exchange.active and self.via == "kernel":
            return self._step_overlapped(rho)
        self.local_step()
        self.exchange.all_reduce(self.sstats)
        if not self.with_elbo:
            self.ctx.call("bsc_natgrad_update_f32", self.lam, self.eta, self.sstats, self.lam.numel(),
                          self.docs_total / self.batch_docs, float(rho))
            return
        if self.world > 1 or self.exchange.rccl:
            self._theta_bound.copy_(self._theta_local)      # (the all-reduce below sums in place)
            self.exchange.all_reduce(self._local)
        self.ctx.call("bsc_natgrad_update_f32_elbo", self.lam, self.eta, self.sstats, self.lam.numel(),
                      self.docs_total / self.batch_docs, float(rho), self._ll, self._theta_bound,
                      self._beta_bound, self.elbo)
