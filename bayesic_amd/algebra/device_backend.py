"""MI355X backend of the algebra front end: every numeric hook is a C-ABI call
into libbayesic_hip.so (include/bayesic_hip.h).

Values are strided device tensors.  torch supplies the storage and the *views*
(``_dimshuffle`` and ``_diagonal`` are zero-copy stride manipulations, exactly
what they are in the reference: ``X.dimshuffle`` bayesic/algebra.py:1319,
``T.Diagonal`` :1407); no torch arithmetic is used -- sums, products, element-wise
functions and contractions are the HIP kernels in csrc/bsc_tensor.hip and
csrc/bsc_gemm.hip.  0-d integer bookkeeping (``X.shape[i]``, ``X.size``, the ``n``
of ``eye(n)``, Python literal scalars) stays on the host as ``HostScalar`` until
it meets a device tensor.

Fusion: unary element-wise nodes and n-ary add / mul are
not launched when they are met.  They become a ``Lazy`` value -- an operand list
with a unary op per operand, a combine op, a scale/shift and a unary post op --
that is folded into whichever node consumes it: another add / mul of the same
kind splices its operands, ``_dimshuffle`` re-views every operand, ``_sum`` and a
full contraction (``_tensordot`` with no free axes) become ONE ``bsc_map_reduce``
launch that reads every operand once and writes only the result.  Anything else
(GEMM operands, ``_diagonal``, ``logdet``, the function result) forces the value
with one fused element-wise launch.  So ``sum(exp(X) * Y)`` or
``C * dot(Th, B) ** -1`` never write an intermediate the size of the data.

Memory plan: the intermediates of a compiled expression are allocated on its
first evaluation and reused by every later one (same shapes), so the hot path
makes no allocator calls except for the result, which always gets fresh storage
because the caller owns it.  (Measured need: with one allocator request per
intermediate, the LDA statistic at 6250 x 100k made ~25 device mallocs per
evaluation through block splitting, 6 ms -> 60+ ms.)

dtype rules (the reference's came from Theano and are pinned by no test):
float32 unless any operand is float64; Python scalars take the dtype of the
tensors they meet; integer arrays are converted to float64 on upload.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _ffi
from .._ffi import BayesicHipError
from ..device import Context, default_context
from .backend import Backend

_OPS = {"add": 0, "mul": 1, "log": 2, "exp": 3, "pow": 4, "abs_": 5, "copy": 6, "gammaln": 7,
        "digamma": 8, "scale": 9}
_DT = {torch.float32: 0, torch.float64: 1}
_MAX_RANK = 6


class HostScalar(object):
    """0-d value kept on the host (shape arithmetic, literals)."""

    def __init__(self, value):
        self.value = value

    @property
    def ndim(self):
        return 0


class Lazy(object):
    """post(scale * COMBINE_i pre_i(t_i) + shift), not yet computed.

    ``terms`` is a list of (tensor, pre_op, pre_arg); every tensor has rank
    ``len(shape)`` and extent 1 where it broadcasts."""

    __slots__ = ("combine", "terms", "scale", "shift", "post", "shape", "dtype")

    def __init__(self, combine, terms, shape, dtype, scale=1.0, shift=0.0, post=None):
        self.combine, self.terms, self.shape, self.dtype = combine, terms, tuple(shape), dtype
        self.scale, self.shift, self.post = scale, shift, post

    @property
    def ndim(self):
        return len(self.shape)

    def dim(self):
        return len(self.shape)

    @property
    def plain(self):
        """No scale, shift or post op: the operand list can be spliced into a
        consumer that combines the same way."""
        return self.post is None and self.scale == 1.0 and self.shift == 0.0


class LazyGemm(object):
    """A matrix-matrix _tensordot that has not been launched yet, so that its consumer can be
    folded into the store (bsc_gemm_epilogue): ``scale * dot(x, y) ** power * E``.  The lowering
    emits ``_mul(B, _tensordot(...))`` and ``_mul(C, pow(_tensordot(...), -1))`` for
    ``B * dot(X, Y)`` and ``C / dot(X, Y)`` (bayesic/algebra.py:741-765, 1419-1432); Theano's graph
    optimiser would have fused them, here the executor does."""

    __slots__ = ("gemm", "shape", "dtype", "power", "scale", "E", "pre")

    def __init__(self, gemm, shape, dtype, power=1, scale=1.0, E=None, pre=None):
        self.gemm, self.shape, self.dtype = gemm, tuple(shape), dtype
        self.power, self.scale, self.E = power, scale, E
        # pre = (code_x, code_y): element-wise producers of the operands folded into the product's fragment reads
        # (bsc_gemm_fused; 1 square, 2 exp, 3 abs) -- `gemm` then holds the producers' SOURCE tensors
        self.pre = pre

    @property
    def ndim(self):
        return len(self.shape)

    def dim(self):
        return len(self.shape)


class LazyLda(LazyGemm):
    """``dot(Th.T, C / dot(Th, Bt))`` not launched yet: the inner product with its division folded in
    (``inner``, a LazyGemm with power -1 and E = C) under an outer product with the SAME Th.  If the consumer is
    ``Bt * (.)`` with the same Bt -- the fixed-gamma statistic of an LDA-style Dirichlet-Multinomial model,
    ``Bt * dot(Th.T, C / dot(Th, Bt))`` (cfg 4) -- the whole expression is ONE pass over the
    count matrix (bsc_lda_sstats, csrc/bsc_lda.hip: neither docs x V intermediate exists); anything else forces
    the two products as before."""

    __slots__ = ("inner", "outer_x", "outer_axes", "C", "Th", "Bt")

    def __init__(self, inner, outer_x, outer_axes, C, Th, Bt, shape):
        LazyGemm.__init__(self, None, shape, torch.float32)
        self.inner, self.outer_x, self.outer_axes = inner, outer_x, outer_axes
        self.C, self.Th, self.Bt = C, Th, Bt


def _i64(values):
    values = list(values)
    return (ctypes.c_int64 * max(len(values), 1))(*values)


class DeferredSoftmax(object):
    """R = softmax_rows(alpha * X_wide . Y) that has NOT been written: the responsibilities of a resident
    Categorical node right after its update (``evaluate_softmax_rows(..., defer=True)``).  What the node's
    neighbours ask of them is, for an exponential-family mixture, always a contraction over the rows with
    constituents of the SAME wide operand the logits came from -- ``dot(R.T, X)``, ``dot(R.T, X * X)``,
    ``sum(R, 0)`` (bayesic/distribution/base.py:329-332: statistics of iid draws add up) -- i.e. column blocks
    of R^T . X_wide, and that the device forms in the pass that forms the softmax (bsc_gemm_softmax_stats,
    csrc/bsc_rowsoftmax.hip): the [rows, N] responsibilities never reach memory.  Anything else asked of the
    object (an element-wise use, a product with other data) writes them after all -- ``realise()``, the
    launch ``evaluate_softmax_rows`` would have made -- and carries on with the tensor."""

    def __init__(self, backend, xcat, wkey, ycat, alpha, transposed=False, state=None, factor=1.0):
        self.backend, self.xcat, self.wkey, self.ycat, self.alpha = backend, xcat, wkey, ycat, float(alpha)
        self.transposed = transposed
        self.factor = float(factor)          # a scalar the responsibilities have been multiplied by (N / B ...)
        self._state = state if state is not None else {}      # shared between a view and its transpose
        m, n = xcat.shape[0], ycat.shape[1]
        self.shape = (n, m) if transposed else (m, n)
        self.dtype = torch.float32

    def dim(self):
        return 2

    @property
    def ndim(self):
        return 2

    @property
    def T(self):
        return DeferredSoftmax(self.backend, self.xcat, self.wkey, self.ycat, self.alpha, not self.transposed,
                               self._state, self.factor)

    def scaled(self, c):
        return DeferredSoftmax(self.backend, self.xcat, self.wkey, self.ycat, self.alpha, self.transposed,
                               self._state, self.factor * float(c))

    def statistics(self):
        """(R^T . X_wide [N, kp] float32, sum_rows lse float64 [1]) -- one pass, computed once."""
        if "stats" not in self._state:
            ctx = self.backend.ctx
            m, kp = self.xcat.shape
            n = self.ycat.shape[1]
            stats = ctx.empty((n, kp), torch.float32)
            lse = ctx.empty((1,), torch.float64)
            # a ones column that closes the wide operand (everything after it padding) goes out of the product:
            # it becomes the kernel's bias row, and its statistic -- the column sums -- lands in its own column
            ones = self.backend._wide.get(("ones", m))
            k_feat, bias = kp, None
            if ones is not None and ones[0] == self.wkey and ones[1] % 8 == 0 and 8 <= ones[1] <= 56 \
                    and kp - ones[1] <= 8:
                k_feat = ones[1]
                bias = self.ycat[k_feat]
                ctx.call("bsc_memset", stats, 0, stats.numel() * 4)        # (the padding columns are never written)
            ctx.call("bsc_gemm_softmax_stats", _ffi.ptr(self.xcat), self.xcat.stride(0), m, k_feat,
                     _ffi.ptr(self.ycat), self.ycat.stride(0), self.ycat.stride(1), n, self.alpha,
                     None if bias is None else _ffi.ptr(bias), None, n, _ffi.ptr(stats), kp, _ffi.ptr(lse))
            self._state["stats"] = (stats, lse)
        return self._state["stats"]

    def realise(self):
        """(R [rows, N], lse [rows], cross [rows]) written after all (bsc_gemm_softmax_rows)."""
        if "rows" not in self._state:
            ctx = self.backend.ctx
            m, kp = self.xcat.shape
            n = self.ycat.shape[1]
            R, lse, cross = ctx.empty((m, n), torch.float32), ctx.empty((m,), torch.float32), \
                ctx.empty((m,), torch.float32)
            ctx.call("bsc_gemm_softmax_rows", _ffi.ptr(self.xcat), self.xcat.stride(0), m, kp, _ffi.ptr(self.ycat),
                     self.ycat.stride(0), self.ycat.stride(1), n, self.alpha, _ffi.ptr(R), n, _ffi.ptr(lse),
                     _ffi.ptr(cross))
            self._state["rows"] = (R, lse, cross)
        return self._state["rows"]

    def tensor(self):
        R = self.realise()[0]
        R = R.t() if self.transposed else R
        return R if self.factor == 1.0 else self.backend._force(self.backend._combine("mul", [HostScalar(self.factor), R]))

    def block(self, part):
        """Column block of R^T . X_wide for the constituent ``part`` of the wide operand, or None."""
        entry = self.backend._wide.get(part)
        if entry is None or entry[0] != self.wkey or self.backend._const_cache.get(self.wkey) is not self.xcat:
            return None
        width = 1 if part[0] == "ones" else part[1]
        block = self.statistics()[0][:, entry[1]:entry[1] + width]
        if self.factor == 1.0:
            return block
        return self.backend._force(self.backend._combine("mul", [HostScalar(self.factor), block]))   # (parameter-sized)

    def entropy_terms(self):
        """(sum_rows lse, sum_rows sum_c r * logit) as host floats: the factor's entropy is their difference.
        The second is <coefficients, statistics> -- logit = alpha * X_wide . Y is linear in the features."""
        stats, lse = self.statistics()
        self.backend.ctx.sync()
        cross = self.alpha * float((stats.double().cpu().numpy() * self.ycat.double().cpu().numpy().T).sum())
        return float(lse.item()), cross


class DeviceBackend(Backend):
    name = "mi355x-hip"

    _MAX_PLANS = 64

    def __init__(self, ctx=None, fuse=True):
        self._plans = {}          # id(expr) -> [expr, [flat buffers in allocation order]]
        self._plan = None         # buffers of the evaluation in progress
        self._cursor = 0
        """fuse=False launches every element-wise node on its own (the unfused
        baseline of tools/bench_fusion.py); results are identical up to rounding."""
        self.fuse = bool(fuse)
        self._one = None          # a resident float32 1.0 (broadcast_to of a host scalar)
        self._keep = None         # buffers made inside an open graph capture
        # storages the caller promised not to change (mark_constant): storage address -> the tensor.  The
        # strong reference is what makes an ADDRESS a valid identity: while a tensor is marked its block
        # cannot go back to the allocator, so no other tensor can turn up at the address the cached values
        # are keyed by (a freed model's data once handed its address -- and its cached X * X -- to the next
        # model's upload).  Unmarking evicts every value computed from the tensor before releasing it.
        self._const = {}
        self._const_ptrs = {}     # single tensors under the same promise (mark_constant_tensor): address -> tensor
        self._wide = {}           # (ptr, columns, row stride) of a constituent -> (key of its wide operand, column offset)
        self._const_cache = {}    # element-wise values of constants only: computed once, LRU by bytes
        self._const_gen = 0       # bumped whenever a constant is withdrawn: recorded call lists (replay_call) are then stale
        self._replays = {}        # key -> recorded C-ABI call list (replay_call)
        self._const_bytes = 0
        self._graphs = {}         # graph_call: key -> recorded hipGraph
        self.ctx = ctx if ctx is not None else default_context()
        if not isinstance(self.ctx, Context):
            raise TypeError("ctx must be a bayesic_amd.device.Context")

    # -- a host-side walk recorded once, replayed as a hipGraph ------------------------------
    def _kept(self, t):
        """Every device buffer made while a capture is open is held by the graph (which stores its
        address, not a reference)."""
        if self._keep is not None:
            self._keep.append(t)
        return t

    _MAX_GRAPHS = 32

    def graph_call(self, key, fn, inputs):
        """``fn()`` issues launches that read the tensors in ``inputs`` (a list; the caller refreshes
        them in place) and returns a list of device tensors.  The first two calls under a ``key``
        run eagerly (workspaces and allocator pools reach their final size); the third is recorded
        into a hipGraph (bsc_capture_begin / _end) and replayed from then on, without the host-side
        walk of ``fn``, as long as the inputs keep their addresses.  The returned tensors are the
        graph's own output buffers: consume them before the next call.  Falls back to eager for good
        when the context is on the null stream or something inside ``fn`` cannot be captured (the call
        in which the capture failed is re-run eagerly: a capture records, it does not execute).

        What a recording holds on to: its inputs, its outputs and EVERY device buffer the walk touched while
        it was recorded -- the capture runs with the per-expression memory plans switched off, so all
        intermediates are fresh allocations owned by the graph (a plan's buffers may be evicted, resized or
        rewritten by an eager evaluation at any later time).  The context's workspace is checked by
        bsc_graph_launch itself; a graph it refuses as stale is recorded again."""
        entry = self._graphs.get(key)
        ptrs = tuple(t.data_ptr() for t in inputs)
        if entry is not None and entry["graph"] is not None:
            if entry["ptrs"] == ptrs:
                try:
                    entry["graph"].launch()
                    return entry["outs"]
                except BayesicHipError as e:
                    if "stale graph" not in str(e):
                        raise
                    entry["graph"].destroy()
                    entry.update(graph=None, outs=None, calls=2)      # the next call records again
                    return [self._force(o) for o in fn()]
            entry["graph"].destroy()
            self._graphs.pop(key, None)
            entry = None                                  # other buffers: start over
        if entry is None:
            if len(self._graphs) >= self._MAX_GRAPHS:            # (a key that keeps changing -- a scalar input that
                old = next(iter(self._graphs))                    # varies per call -- must not pile up recordings)
                dropped = self._graphs.pop(old)
                if dropped["graph"] is not None:
                    dropped["graph"].destroy()
            entry = self._graphs[key] = {"calls": 0, "graph": None, "ptrs": ptrs, "outs": None, "dead": False}
        entry["calls"] += 1
        if entry["dead"] or entry["calls"] < 3 or not self.ctx.can_capture or entry["ptrs"] != ptrs:
            entry["ptrs"] = ptrs
            return [self._force(o) for o in fn()]
        self._keep = []                   # (evaluate() then allocates every intermediate afresh: _plan_for)
        self.ctx.capture_begin()
        try:
            outs = [self._force(o) for o in fn()]
        except Exception:
            self._keep = None
            try:
                self.ctx.capture_end()
            except Exception:
                pass
            entry["dead"] = True
            # nothing ran (a capture records): give the caller this call's result by the eager walk; only
            # if that fails too is the error the caller's
            return [self._force(o) for o in fn()]
        keep, self._keep = self._keep, None
        try:
            graph = self.ctx.capture_end(keep + list(outs) + list(inputs))
        except Exception:
            entry["dead"] = True                          # not capturable: eager from now on
            return [self._force(o) for o in fn()]
        entry["graph"], entry["outs"] = graph, outs
        graph.launch()                                    # a capture records, it does not run
        return outs

    _CONST_CACHE_BYTES = 8 << 30

    def replay_call(self, key, fn, inputs):
        """``graph_call`` without a graph: the third call under a ``key`` runs ``fn()`` once more with every C-ABI call
        RECORDED (``Context.record_begin``: bound function + raw arguments) and every intermediate a fresh allocation
        owned by the recording; later calls re-issue that list (``Context.replay``) while the inputs keep their
        addresses, shapes and strides.  What it removes is the host-side walk of the expression -- ~28 us of Python per
        launch, the whole gap between the general engines and their kernels on short launches -- and what it needs is
        less than a capture does: no stream of the context's own, nothing that cannot be captured to avoid.  What it does
        NOT do is what a graph does on the device (one submission, ~1.2 us between kernels): each launch is still an
        eager one.  Contract as for ``graph_call``: ``fn`` reads ``inputs`` (refreshed in place by the caller), contains
        no host read-back, and returns device tensors that belong to the recording -- consume them before the next call."""
        entry = self._replays.get(key)
        sig = tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype) for t in inputs)
        if entry is not None and entry["calls"] is not None:
            if entry["sig"] == sig and entry["gen"] == self._const_gen:
                self.ctx.replay(entry["calls"])
                return entry["outs"]
            self._replays.pop(key, None)                  # other buffers, or a constant was withdrawn: start over
            entry = None
        if entry is None:
            if len(self._replays) >= self._MAX_GRAPHS:
                self._replays.pop(next(iter(self._replays)))
            entry = self._replays[key] = {"n": 0, "calls": None, "outs": None, "keep": None, "sig": sig,
                                          "gen": self._const_gen}
        entry["n"] += 1
        if entry["n"] < 3 or entry["sig"] != sig or self._keep is not None or self.ctx._record is not None:
            entry["sig"] = sig
            return [self._force(o) for o in fn()]
        self._keep = []                   # (evaluate() then allocates every intermediate afresh: _plan_for)
        self.ctx.record_begin()
        try:
            outs = [self._force(o) for o in fn()]
        finally:
            calls = self.ctx.record_end()
            keep, self._keep = self._keep, None
        entry.update(calls=calls, outs=outs, keep=keep + list(outs) + list(inputs), gen=self._const_gen)
        return outs

    def mark_constant(self, *tensors):
        """A promise that these device tensors (a model's data) are not written while they stay marked:
        an element-wise value of constants only -- ``X * X`` in every message of a Gaussian model --
        is then computed once and kept (up to _CONST_CACHE_BYTES, least recently used first out)
        instead of being recomputed by every evaluation that contains it."""
        for t in tensors:
            if isinstance(t, torch.Tensor):
                self._const[t.untyped_storage().data_ptr()] = t

    def mark_constant_tensor(self, *tensors):
        """The same promise for tensors that may share their storage with others (an intermediate
        the executor allocated from its arena, such as a mixture's responsibilities): only a tensor
        starting at exactly this address counts, not its storage's other tenants."""
        for t in tensors:
            if isinstance(t, torch.Tensor):
                self._const_ptrs[t.data_ptr()] = t

    def _is_const(self, t):
        return t.untyped_storage().data_ptr() in self._const or t.data_ptr() in self._const_ptrs

    def forget_constants(self):
        self._const_gen += 1
        self._const.clear()
        self._const_ptrs.clear()
        self._const_cache.clear()
        self._wide.clear()
        self._const_bytes = 0

    def unmark_constant(self, *tensors):
        """The caller is about to change or release these tensors: cached values computed from them
        are dropped and they are no longer constants (call BEFORE the storage can be reused)."""
        self._const_gen += 1
        for t in tensors:
            if not isinstance(t, torch.Tensor):
                continue
            ptr = t.data_ptr()
            for key in [k for k in self._const_cache if self._key_mentions(k, ptr)]:
                self._evict(key)        # (while the tensor is still held: nothing can have taken its address)
            if ptr in self._const_ptrs:
                self._const_ptrs.pop(ptr, None)
            else:
                self._const.pop(t.untyped_storage().data_ptr(), None)

    def _evict(self, key):
        """Drop one cached value and everything computed FROM it (a wide operand built from a cached
        element-wise value, products with that wide operand): once its buffer is released another
        tensor may get the address the dependants are keyed by."""
        old = self._const_cache.pop(key, None)
        if old is None:
            return
        self._const_gen += 1
        self._const_bytes -= old.numel() * old.element_size()
        self._const.pop(old.untyped_storage().data_ptr(), None)
        ptr = old.data_ptr()
        for part in [p for p, (wkey, _) in self._wide.items() if wkey == key or p[0] == ptr]:
            self._wide.pop(part, None)
        for dep in [k for k in self._const_cache
                    if self._key_mentions(k, ptr) or (k[0] == "kprod" and k[5] == ptr)]:
            self._evict(dep)

    @staticmethod
    def _key_mentions(key, ptr):
        if key and key[0] in ("sum", "kprod"):
            return key[1] == ptr
        return any(term[0] == ptr for term in key[1])

    def compile(self, expr, bindings=None, graph=False):
        """As ``Backend.compile``.  ``graph=True``: ``f.device_fn(**device_inputs)`` records its launches
        once (third call, see ``graph_call``) and replays them while it is called with the same device
        buffers -- where the reference compiles a Theano function (bayesic/algebra.py:50-58) this
        records a hipGraph.  The tensor it returns is then the graph's own output buffer, overwritten
        by the next call.  (``f(**arrays)`` uploads fresh buffers on every call and stays eager.)"""
        f = Backend.compile(self, expr, bindings)
        if not graph:
            return f
        eager = f.device_fn
        backend = self

        def device_fn(**device_inputs):
            names = sorted(device_inputs)
            tensors = [device_inputs[n] for n in names if isinstance(device_inputs[n], torch.Tensor)]
            scalars = tuple((n, device_inputs[n].value) for n in names if isinstance(device_inputs[n], HostScalar))
            return backend.graph_call(("compile", id(expr), scalars), lambda: [eager(**device_inputs)], tensors)[0]

        device_fn.expr = expr            # (keeps id(expr) unique for the life of the function)
        f.device_fn = device_fn
        return f

    # -- host <-> device ---------------------------------------------------------
    def from_host(self, array, dtype, ndim):
        if isinstance(array, torch.Tensor) and array.is_cuda:
            # already resident (a mini-batch the caller keeps in HBM): taken as it is, no copy
            if array.dim() != ndim:
                raise ValueError("input has ndim %d, expected %d" % (array.dim(), ndim))
            want = {"float32": torch.float32, "float64": torch.float64}.get(str(np.dtype(dtype)))
            if want is None or array.dtype != want:
                raise TypeError("a device tensor passed as input must already have dtype %s (got %s)"
                                % (dtype, array.dtype))
            return array
        a = np.asarray(array)
        if a.ndim != ndim:
            raise ValueError("input has ndim %d, expected %d" % (a.ndim, ndim))
        if ndim == 0 and np.dtype(dtype).kind == "f":
            # scalar parameters travel as kernel arguments (double), not as device tensors
            return HostScalar(float(a))
        if np.dtype(dtype).kind in "iub":
            if ndim == 0:
                return HostScalar(int(a))
            a = a.astype(np.float64)
        else:
            a = a.astype(dtype, copy=False)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float32)
        # (np.ascontiguousarray would turn a 0-d array into shape (1,))
        a = np.array(a, order="C", copy=True)
        return self._kept(torch.from_numpy(a).to(self.ctx.device))

    def to_host(self, value):
        value = self._force(value)
        if isinstance(value, HostScalar):
            return np.asarray(value.value)
        self.ctx.sync()
        return value.cpu().numpy()

    def constant(self, value):
        a = np.asarray(value)
        if a.ndim == 0:
            return HostScalar(a.item())
        return self.from_host(a, "float64" if a.dtype == np.float64 else
                              ("float32" if a.dtype.kind == "f" else str(a.dtype)), a.ndim)

    def shape(self, x, axis):
        return HostScalar(int(x.shape[axis]))

    def _host_int(self, v):
        if isinstance(v, HostScalar):
            return int(v.value)
        v = self._force(v)
        self.ctx.sync()
        return int(v.item())

    def eye(self, n):
        n = self._host_int(n)
        out = self._empty((n, n), torch.float32)
        self.ctx.call("bsc_eye", 0, _ffi.ptr(out), n)
        return out

    # -- helpers ----------------------------------------------------------------------
    def _empty(self, shape, dtype):
        """Uninitialised device tensor; inside evaluate() the k-th request reuses the
        k-th buffer of the previous evaluation of the same expression."""
        shape = tuple(int(n) for n in shape)
        if self._plan is None:
            return self._kept(self.ctx.empty(shape, dtype))       # (checks torch's current stream is the context's)
        numel = math.prod(shape)
        k = self._cursor
        self._cursor += 1
        if k < len(self._plan):
            buf = self._plan[k]
            if buf is not None and buf.dtype == dtype and buf.numel() == numel:
                return buf.view(shape)
        buf = self._kept(self.ctx.empty((numel,), dtype))
        if k < len(self._plan):
            self._plan[k] = buf
        else:
            self._plan.append(buf)
        return buf.view(shape)

    def _upload_scalar(self, s, dtype, ndim):
        t = self._kept(torch.tensor(float(s.value), dtype=dtype, device=self.ctx.device))
        return t.reshape((1,) * ndim)

    def _convert(self, t, dtype):
        out = self._empty(t.shape, dtype)
        self.ctx.call("bsc_convert", _DT[t.dtype], _DT[dtype], t.dim(), _i64(t.shape),
                      _ffi.ptr(t), _i64(t.stride()), _ffi.ptr(out), _i64(out.stride()))
        return out

    def _contiguous(self, t):
        return t if t.is_contiguous() else self._convert(t, t.dtype)

    def _common(self, args):
        """Device tensors of one dtype and one rank from a mix of tensors / host scalars."""
        tensors = [a for a in args if not isinstance(a, HostScalar)]
        dtype = torch.float64 if any(t.dtype == torch.float64 for t in tensors) else torch.float32
        ndim = max(t.dim() for t in tensors)
        out = []
        for a in args:
            if isinstance(a, HostScalar):
                a = self._upload_scalar(a, dtype, ndim)
            elif a.dtype != dtype:
                a = self._convert(a, dtype)
            out.append(a)
        return out, dtype, ndim

    @staticmethod
    def _host_elemwise(op_name, values):
        if op_name == "add":
            return sum(values)
        if op_name == "mul":
            return math.prod(values)
        with np.errstate(all="ignore"):
            if op_name == "log":
                return float(np.log(values[0]))
            if op_name == "exp":
                return float(np.exp(values[0]))
            if op_name == "pow":
                return float(np.power(float(values[0]), values[1])) \
                    if not all(isinstance(v, int) and v >= 0 for v in values) else values[0] ** values[1]
            if op_name == "abs_":
                return abs(values[0])
            if op_name == "gammaln":
                return math.lgamma(values[0])
            if op_name == "digamma":
                from scipy.special import digamma      # host scalars only (shape arithmetic)
                return float(digamma(values[0]))
        raise ValueError("unknown elementwise op %r" % op_name)

    def _elemwise(self, op_name, args):
        """Immediate element-wise launch (binary pow with a tensor exponent)."""
        args = [self._force(a) for a in args]
        if all(isinstance(a, HostScalar) for a in args):
            return HostScalar(self._host_elemwise(op_name, [a.value for a in args]))
        args, dtype, ndim = self._common(args)
        if ndim > _MAX_RANK:
            raise _ffi.BayesicHipError("rank %d exceeds the kernels' limit %d" % (ndim, _MAX_RANK))
        shape = self._broadcast_shape([a.shape for a in args], ndim)
        out = self._empty(shape, dtype)
        strides = []
        for a in args:
            strides += [0 if (a.shape[ax] == 1 and shape[ax] != 1) else a.stride(ax)
                        for ax in range(ndim)]
        ptrs = (ctypes.c_void_p * len(args))(*[a.data_ptr() for a in args])
        self.ctx.call("bsc_elemwise", _OPS[op_name], _DT[dtype], ndim, _i64(shape), _ffi.ptr(out),
                      _i64(out.stride()), len(args), ptrs, _i64(strides))
        return out

    @staticmethod
    def _broadcast_shape(shapes, ndim):
        shape = []
        for axis in range(ndim):
            extents = {s[axis] for s in shapes}
            big = extents - {1}
            if len(big) > 1:
                raise ValueError("shapes do not broadcast on axis %d: %s" % (axis, sorted(extents)))
            shape.append(big.pop() if big else 1)
        return shape

    # -- deferred element-wise values -------------------------------------------------
    def _launch(self, lazy, red_axes=()):
        """One bsc_map_reduce launch: the value of `lazy`, summed over red_axes."""
        rank = lazy.ndim
        if rank > _MAX_RANK:
            raise _ffi.BayesicHipError("rank %d exceeds the kernels' limit %d" % (rank, _MAX_RANK))
        red = sorted(a % rank for a in red_axes) if rank else []
        keep = [a for a in range(rank) if a not in red]
        shape = lazy.shape
        terms = lazy.terms
        out = None
        ckey = None
        if (self._const or self._const_ptrs) and not red and self._keep is None and \
                all(self._is_const(t) for t, _, _ in terms):
            ckey = (lazy.combine, tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride()), op, float(arg))
                                        for t, op, arg in terms),
                    float(lazy.scale), float(lazy.shift), lazy.post, tuple(shape), lazy.dtype)
            hit = self._const_cache.pop(ckey, None)
            if hit is not None:
                self._const_cache[ckey] = hit            # most recently used last
                return hit
        if ckey is not None:
            # kept across evaluations: not a buffer of the per-expression memory plan
            nbytes = math.prod(shape) * (8 if lazy.dtype == torch.float64 else 4)
            while self._const_cache and self._const_bytes + nbytes > self._CONST_CACHE_BYTES:
                self._evict(next(iter(self._const_cache)))
            if nbytes <= self._CONST_CACHE_BYTES:
                out = self.ctx.empty([shape[a] for a in keep], lazy.dtype)
            else:
                ckey = None
        if out is None and not red and rank > 1:
            # Pure map: lay the result out the way its biggest operand lies in memory (a
            # transposed view stays a transposed view) so that reads and writes both stream;
            # a row-major result of a column-major operand is an uncoalesced transpose.
            ref = max((t for t, _, _ in terms), key=lambda t: t.numel())
            order = sorted(range(rank), key=lambda ax: (-abs(ref.stride(ax)) if ref.shape[ax] != 1 else 0,
                                                        ax))
            order = [ax for ax in order if ref.shape[ax] != 1] + [ax for ax in range(rank)
                                                                   if ref.shape[ax] == 1]
            if order != list(range(rank)) and ref.numel() == math.prod(shape):
                base = self._empty([shape[ax] for ax in order], lazy.dtype)
                inverse = [order.index(ax) for ax in range(rank)]
                out = base.permute(inverse)
        if out is None:
            out = self._empty([shape[a] for a in keep], lazy.dtype)

        def strides(t, axes):
            return [0 if (t.shape[ax] == 1 and shape[ax] != 1) else t.stride(ax) for ax in axes]

        keep_strides, red_strides = [], []
        for t, _, _ in terms:
            keep_strides += strides(t, keep)
            red_strides += strides(t, red)
        n = len(terms)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _, _ in terms])
        pre_ops = (ctypes.c_int32 * n)(*[_OPS[op or "copy"] for _, op, _ in terms])
        pre_args = (ctypes.c_double * n)(*[float(arg) for _, _, arg in terms])
        post_op, post_arg = lazy.post if lazy.post is not None else ("copy", 0.0)
        self.ctx.call("bsc_map_reduce", _DT[lazy.dtype], _OPS[lazy.combine], len(keep),
                      _i64(shape[a] for a in keep), len(red), _i64(shape[a] for a in red), n, ptrs,
                      _i64(keep_strides), _i64(red_strides), pre_ops, pre_args, float(lazy.scale),
                      float(lazy.shift), _OPS[post_op], float(post_arg), _ffi.ptr(out),
                      _i64(out.stride()))
        if ckey is not None:
            self._const_cache[ckey] = out
            self._const_bytes += out.numel() * out.element_size()
            self._const[out.untyped_storage().data_ptr()] = out      # a value of constants is a constant
        return out

    def _force(self, v):
        if isinstance(v, LazyGemm):
            return self._launch_gemm(v)
        if isinstance(v, DeferredSoftmax):
            return v.tensor()
        return self._launch(v) if isinstance(v, Lazy) else v

    def _launch_gemm(self, g):
        if isinstance(g, LazyLda):          # not consumed by Bt * (.): the two products after all
            q = self._launch_gemm(g.inner)
            x_dot, y_dot = g.outer_axes
            return self._force(self.tensordot(g.outer_x, q, x_dot, y_dot, [], []))
        xb, m, n, k, x, sxb, sxm, sxk, y, syb, syk, syn = g.gemm
        out = self._empty(g.shape, g.dtype)
        if g.pre is not None:
            E = g.E
            se_m = se_n = 0
            if E is not None:
                se_m = 0 if E.shape[0] == 1 else E.stride(0)
                se_n = 0 if E.shape[1] == 1 else E.stride(1)
            plain = g.power == 1 and E is None and g.scale == 1.0
            handled = ctypes.c_int32(0)
            self.ctx.call("bsc_gemm_fused", _DT[g.dtype], xb, m, n, k, _ffi.ptr(x), sxb, sxm, sxk, g.pre[0],
                          _ffi.ptr(y), syb, syk, syn, g.pre[1], _ffi.ptr(out), m * n, n, 1,
                          0 if plain else int(g.power), float(g.scale), _ffi.ptr(E) if E is not None else None,
                          0, se_m, se_n, ctypes.byref(handled))
            if handled.value:
                return out
            # the shape takes a kernel without prologues: the producers are launched after all (same strides:
            # an element-wise value has its source's layout)
            x = self._apply_pre(x, g.pre[0])
            y = self._apply_pre(y, g.pre[1])
            (sxb, sxm, sxk), (syb, syk, syn) = self._restride(g.gemm[4], x, sxb, sxm, sxk), self._restride(g.gemm[8], y, syb, syk, syn)
        if g.power == 1 and g.E is None and g.scale == 1.0:
            self.ctx.call("bsc_gemm_strided_batched", _DT[g.dtype], xb, m, n, k,
                          _ffi.ptr(x), sxb, sxm, sxk, _ffi.ptr(y), syb, syk, syn,
                          _ffi.ptr(out), m * n, n, 1)
            return out
        E = g.E
        se_m = se_n = 0
        if E is not None:
            se_m = 0 if E.shape[0] == 1 else E.stride(0)
            se_n = 0 if E.shape[1] == 1 else E.stride(1)
        self.ctx.call("bsc_gemm_epilogue", _DT[g.dtype], xb, m, n, k,
                      _ffi.ptr(x), sxb, sxm, sxk, _ffi.ptr(y), syb, syk, syn,
                      _ffi.ptr(out), m * n, n, 1, int(g.power), float(g.scale),
                      _ffi.ptr(E) if E is not None else None, 0, se_m, se_n)
        return out

    _PRE_CODES = {1: ("pow", 2.0), 2: ("exp", 0.0), 3: ("abs_", 0.0)}

    def _apply_pre(self, t, code):
        if not code:
            return t
        op, arg = self._PRE_CODES[code]
        return self._launch(Lazy("mul", [(t, op, arg)], t.shape, t.dtype))

    @staticmethod
    def _restride(src, new, s_b, s_m, s_k):
        """Strides of the merged (batch, free, contracted) axis groups of `src`, for `new` = an element-wise
        value of it: the same when the layouts agree (they do for a dense source), else not representable."""
        if new is src or tuple(new.stride()) == tuple(src.stride()):
            return s_b, s_m, s_k
        raise ValueError("prologue fallback: the materialised operand does not have its source's layout")

    def _prologue_of(self, v):
        """(source tensor, code, scale) when `v` is an element-wise producer the GEMM can apply to its fragments:
        exp(T), abs(T), T ** 2, T * T (times a host scalar), T a dense float32 tensor of v's shape that is not a
        constant of the model (a constant's X * X is computed once and cached instead: _const_cache)."""
        if not (self.fuse and isinstance(v, Lazy)) or v.post is not None or v.shift != 0.0 or v.combine != "mul" \
                or v.dtype != torch.float32:
            return None
        terms = v.terms
        code = None
        if len(terms) == 1:
            t, op, arg = terms[0]
            code = {"exp": 2, "abs_": 3}.get(op) if op in ("exp", "abs_") else (1 if op == "pow" and float(arg) == 2.0 else None)
        elif len(terms) == 2 and terms[0][1] is None and terms[1][1] is None:
            t, u = terms[0][0], terms[1][0]
            if t.data_ptr() == u.data_ptr() and tuple(t.shape) == tuple(u.shape) and tuple(t.stride()) == tuple(u.stride()):
                code = 1
        if code is None:
            return None
        if tuple(t.shape) != tuple(v.shape) or t.dtype != torch.float32 or not t.is_contiguous() or self._is_const(t):
            return None
        return t, code, float(v.scale)

    def _unary(self, op_name, x, arg=0.0):
        if isinstance(x, LazyLda):
            x = self._force(x)
        if isinstance(x, LazyGemm):
            if self.fuse and op_name == "pow" and float(arg) in (1.0, -1.0) and x.E is None and \
                    x.power == 1 and x.scale == 1.0:
                return LazyGemm(x.gemm, x.shape, x.dtype, power=int(arg), pre=x.pre)
            x = self._force(x)
        if isinstance(x, Lazy):
            if x.post is None and self.fuse:
                return Lazy(x.combine, x.terms, x.shape, x.dtype, x.scale, x.shift, (op_name, arg))
            x = self._force(x)
        out = Lazy("mul", [(x, op_name, arg)], x.shape, x.dtype)
        return out if self.fuse else self._launch(out)

    def _combine(self, op_name, args):
        """n-ary add / mul as a deferred value; host scalars fold into scale / shift."""
        mul = op_name == "mul"
        host = [a.value for a in args if isinstance(a, HostScalar)]
        rest = [a for a in args if not isinstance(a, HostScalar)]
        if not rest:
            return HostScalar(self._host_elemwise(op_name, host))
        ldas = [a for a in rest if isinstance(a, LazyLda)]
        if ldas:
            out = self._lda_statistic(rest, host) if (mul and len(ldas) == 1) else None
            if out is not None:
                return out
            rest = [self._force(a) if isinstance(a, LazyLda) else a for a in rest]
        gemms = [a for a in rest if isinstance(a, LazyGemm)]
        if gemms:
            fused = self._fold_into_gemm(rest, host) if (mul and self.fuse and len(gemms) == 1) else None
            if fused is None and not mul and self.fuse and len(gemms) >= 2:
                fused = self._concat_products(rest, host)
            if fused is not None:
                return fused
            rest = [self._force(a) if isinstance(a, LazyGemm) else a for a in rest]
            args = rest + [HostScalar(h) for h in host]
        coef = (math.prod(host) if mul else sum(host)) if host else (1.0 if mul else 0.0)
        ndim = max(a.dim() for a in rest)
        dtype = torch.float64 if any(a.dtype == torch.float64 for a in rest) else torch.float32
        terms = []
        for a in rest:
            if a.dim() != ndim:                       # numpy-style left padding
                a = self._force(a)
                a = a.reshape((1,) * (ndim - a.dim()) + tuple(a.shape))
            if isinstance(a, Lazy):
                single = len(a.terms) == 1
                if a.post is None and (a.combine == op_name or single) and \
                        (a.plain or (mul and a.shift == 0.0) or (not mul and a.scale == 1.0)):
                    # splice: (s * prod) * rest == s * (prod * rest);  (sum + c) + rest likewise
                    if mul:
                        coef = coef * a.scale
                    else:
                        coef = coef + a.shift
                    terms += a.terms
                    continue
                if not mul and single and a.post is None and a.shift == 0.0 and a.terms[0][1] is None \
                        and a.dtype == torch.float32:
                    # c * t as an addend: the coefficient rides on the operand (BSC_OP_SCALE), so that
                    # (1 - rho) eta + rho m -- every damped update -- is one launch instead of three
                    terms.append((a.terms[0][0], "scale", float(a.scale)))
                    continue
                a = self._force(a)
            terms.append((a, None, 0.0))
        if mul and len(terms) >= 2:
            # x * x: ONE operand with the pre-op pow 2 -- the same tensor as two operands is two load instructions per
            # value (sum(R * R, axis=0) at 10M x 64: 820 us against 475 for sum(R, axis=0))
            merged, seen = [], {}
            for t, op, arg in terms:
                key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype) if op is None else None
                at = seen.get(key) if key is not None else None
                if at is not None and merged[at][1] is None:
                    merged[at] = (t, "pow", 2.0)
                    continue
                if key is not None and at is None:
                    seen[key] = len(merged)
                merged.append((t, op, arg))
            terms = merged
        terms = [(t if t.dtype == dtype else self._convert(t, dtype), op, arg)
                 for t, op, arg in terms]
        while len(terms) > 8:                         # the kernel takes up to 8 operands
            head = Lazy(op_name, terms[:8], self._broadcast_shape([t.shape for t, _, _ in terms[:8]],
                                                                  ndim), dtype)
            terms = [(self._launch(head), None, 0.0)] + terms[8:]
        shape = self._broadcast_shape([t.shape for t, _, _ in terms], ndim)
        out = Lazy(op_name, terms, shape, dtype, scale=coef if mul else 1.0,
                   shift=0.0 if mul else coef)
        return out if self.fuse else self._launch(out)

    def _map_into(self, out_view, shape, operands, scale=1.0, shift=0.0):
        """out_view[...] = scale * sum(operands) + shift for float32 operands given as (tensor, strides)
        over `shape`: one bsc_map_reduce launch writing through the view's strides."""
        n = len(operands)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in operands])
        strides = []
        for _, st in operands:
            strides += list(st)
        pre_ops = (ctypes.c_int32 * n)(*([_OPS["copy"]] * n))
        pre_args = (ctypes.c_double * n)(*([0.0] * n))
        self.ctx.call("bsc_map_reduce", _DT[torch.float32], _OPS["add"], len(shape), _i64(shape), 0,
                      _i64([]), n, ptrs, _i64(strides), _i64([]), pre_ops, pre_args, float(scale),
                      float(shift), _OPS["copy"], 0.0, _ffi.ptr(out_view), _i64(out_view.stride()))

    def _concat_products(self, rest, host):
        """sum_i s_i dot(X_i, Y_i) + row vectors + scalars  ->  ONE product dot([X_1 | X_2 | .. | 1],
        [s_1 Y_1 ; s_2 Y_2 ; .. ; bias]) when every X_i is a CONSTANT with contiguous rows (a model's
        data and cached element-wise values of it: the logits of an exponential-family mixture are
        sum_j T_j(x) . eta_j + c).  The wide left operand is built once and kept with the constants;
        the stacked right operand is a few small launches per evaluation.  Saves one [M, N] result per
        product and the n-ary add over them.  None when the addends do not have that form."""
        gemms = [a for a in rest if isinstance(a, LazyGemm)]
        others = [a for a in rest if not isinstance(a, LazyGemm)]
        g0 = gemms[0]
        if len(g0.shape) != 2 or g0.dtype != torch.float32:
            return None
        m, n = g0.shape
        for g in gemms:
            xb, gm, gn, k, x, sxb, sxm, sxk, y, syb, syk, syn = g.gemm
            if g.power != 1 or g.E is not None or g.pre is not None or xb != 1 or (gm, gn) != (m, n) or g.dtype != torch.float32 \
                    or sxk != 1 or sxm < k or not self._is_const(x) or self._keep is not None:
                return None
        rows = []
        for a in others:                    # only bias-like addends: [1, n] (or [n]) float32
            if isinstance(a, Lazy):
                if tuple(a.shape) not in ((1, n), (n,)):
                    return None
                a = self._force(a)
            if not isinstance(a, torch.Tensor) or a.dtype != torch.float32 or tuple(a.shape) not in ((1, n), (n,)):
                return None
            rows.append((a, [0, a.stride(-1)]))
        bias = bool(rows) or any(h != 0.0 for h in host)
        ks = [g.gemm[3] for g in gemms]
        ktot = sum(ks) + (1 if bias else 0)
        kp = (ktot + 7) // 8 * 8          # (a multiple of 8: bsc_gemm_softmax_rows pairs column s with K/2 + s)
        if kp > 1024 or m * kp * 4 > self._CONST_CACHE_BYTES // 2:
            return None
        ckey = ("kcat", tuple((g.gemm[4].data_ptr(), g.gemm[3], g.gemm[6]) for g in gemms), bias, m, kp)
        xcat = self._const_cache.pop(ckey, None)
        if xcat is None:
            if self._one is None:
                self._one = self.from_host(np.ones(1, np.float32), "float32", 1)
            xcat = self.ctx.empty((m, kp), torch.float32)
            for stale in [key for key in self._const_cache if key[0] == "kprod" and key[5] == xcat.data_ptr()]:
                self._evict(stale)                                                # products with a former tenant
            self.ctx.call("bsc_memset", xcat, 0, xcat.numel() * 4)
            off = 0
            for g in gemms:
                k, x, sxm = g.gemm[3], g.gemm[4], g.gemm[6]
                self._map_into(xcat[:, off:off + k], [m, k], [(x, [sxm, 1])])
                off += k
            if bias:
                self._map_into(xcat[:, off:off + 1], [m, 1], [(self._one, [0, 0])])
            self._const_bytes += xcat.numel() * 4
            self._const[xcat.untyped_storage().data_ptr()] = xcat
        self._const_cache[ckey] = xcat
        off = 0
        for g in gemms:
            self._wide[(g.gemm[4].data_ptr(), g.gemm[3], g.gemm[6])] = (ckey, off)
            off += g.gemm[3]
        if bias:
            self._wide[("ones", m)] = (ckey, off)
        ycat = self._empty((kp, n), torch.float32)
        self.ctx.call("bsc_memset", ycat, 0, ycat.numel() * 4)
        off = 0
        for g in gemms:
            k, y, syk, syn = g.gemm[3], g.gemm[8], g.gemm[10], g.gemm[11]
            self._map_into(ycat[off:off + k, :], [k, n], [(y, [syk, syn])], scale=g.scale)
            off += k
        if bias:
            if rows:
                self._map_into(ycat[off:off + 1, :], [1, n], rows, shift=float(sum(host)))
            else:
                self._map_into(ycat[off:off + 1, :], [1, n], [(self._one, [0, 0])], scale=float(sum(host)))
        return LazyGemm((1, m, n, kp, xcat, 0, kp, 1, ycat, 0, n, 1), (m, n), torch.float32)

    def _product_with_wide(self, x, m, k, sxm, sxk, part):
        """dot(x^T-like [m, k], Y) where Y [k, n] is a constituent of a cached wide operand
        [Y_1 | Y_2 | .. | 1] (see _concat_products) and x is a constant: the product with the WHOLE wide
        operand is computed once, kept while x stays marked, and every constituent's product -- and,
        through the ones column, the sums of x over the contracted axis -- is a column block of it.
        The responsibility-weighted statistics of a mixture (R^T X, R^T X^2, sum_n R) are then one pass
        over R instead of three.  `part` = (ptr, columns, row stride) of Y, or ("ones", k)."""
        if part is None or not self._is_const(x):
            return None
        entry = self._wide.get(part)
        if entry is None:
            return None
        wkey, off = entry
        xcat = self._const_cache.get(wkey)
        if xcat is None or xcat.shape[0] != k:
            self._wide.pop(part, None)
            return None
        kp = xcat.shape[1]
        pkey = ("kprod", x.data_ptr(), m, sxm, sxk, xcat.data_ptr())
        prod = self._const_cache.pop(pkey, None)
        if prod is None:
            prod = self.ctx.empty((m, kp), torch.float32)
            self.ctx.call("bsc_gemm_strided_batched", _DT[torch.float32], 1, m, kp, k,
                          _ffi.ptr(x), 0, sxm, sxk, _ffi.ptr(xcat), 0, kp, 1, _ffi.ptr(prod), m * kp, kp, 1)
            self._const_bytes += prod.numel() * 4
        self._const_cache[pkey] = prod
        width = 1 if part[0] == "ones" else part[1]
        return prod[:, off:off + width]

    def _deferred_statistics(self, x, y, x_dot, y_dot, x_batch, y_batch):
        """dot(R^T-like, Y) with R deferred (never written) and Y a constituent of the wide operand R's logits
        were formed from: a column block of the statistics the fused pass leaves -- or None."""
        if x_batch or y_batch or len(x_dot) != 1 or len(y_dot) != 1:
            return None
        swap = not isinstance(x, DeferredSoftmax)
        r, other, r_dot, o_dot = (y, x, y_dot[0], x_dot[0]) if swap else (x, y, x_dot[0], y_dot[0])
        if isinstance(other, (DeferredSoftmax, Lazy, LazyGemm, HostScalar)) or other.dim() != 2:
            return None
        rows_axis = 1 if r.transposed else 0
        if r_dot != rows_axis or other.dtype != torch.float32:
            return None
        free = 1 - o_dot
        if other.stride(free) != 1 or not self._is_const(other):
            return None
        block = r.block((other.data_ptr(), other.shape[free], other.stride(o_dot)))
        if block is None:
            return None
        return block.t() if swap else block          # [N, width], or [width, N] when R is the right operand

    def _lda_pattern(self, x, y, x_dot, y_dot):
        """x = Th^T (a view), y = C / dot(Th, Bt) still deferred: a LazyLda, or None."""
        if not (isinstance(x, torch.Tensor) and x.dim() == 2 and x.dtype == torch.float32 and y.scale == 1.0 and y.pre is None
                and y.dtype == torch.float32 and len(y.shape) == 2 and list(y_dot) == [0] and len(x_dot) == 1):
            return None
        xb, docs, V, K, Th, sxb, ldth, sxk, Bt, syb, ldb, syn = y.gemm
        C = y.E
        kx = 1 - x_dot[0]                           # x's free axis: the topics
        if xb != 1 or sxk != 1 or syn != 1 or K not in (32, 64, 96, 128) or x.shape[kx] != K \
                or x.shape[x_dot[0]] != docs or x.data_ptr() != Th.data_ptr() or x.stride(kx) != 1 \
                or x.stride(x_dot[0]) != ldth or not isinstance(C, torch.Tensor) or tuple(C.shape) != (docs, V) \
                or C.stride(1) != 1 or ldth < K or ldb < V or C.stride(0) < V:
            return None
        out = LazyLda(y, x, (list(x_dot), list(y_dot)), C, (Th, ldth), (Bt, ldb), (K, V) if kx == 0 else (V, K))
        return out if kx == 0 else None

    def _lda_statistic(self, rest, host):
        """Bt * LazyLda with the LazyLda's own Bt: bsc_lda_sstats (one pass over C), else None."""
        g = next(a for a in rest if isinstance(a, LazyLda))
        others = [a for a in rest if a is not g]
        (Bt, ldb), (Th, ldth) = g.Bt, g.Th
        K, V = g.shape
        if len(others) != 1 or not isinstance(others[0], torch.Tensor):
            return None
        E = others[0]
        if E.dtype != torch.float32 or tuple(E.shape) != (K, V) or E.data_ptr() != Bt.data_ptr() or \
                E.stride(1) != 1 or E.stride(0) != ldb:
            return None
        docs = g.C.shape[0]
        out = self._empty((K, V), torch.float32)
        self.ctx.call("bsc_lda_sstats", _ffi.ptr(g.C), g.C.stride(0), docs, V, K, _ffi.ptr(Th), ldth, _ffi.ptr(Bt),
                      ldb, _ffi.ptr(out), V)
        scale = math.prod(host) if host else 1.0
        return out if scale == 1.0 else self._combine("mul", [HostScalar(scale), out])

    def _fold_into_gemm(self, rest, host):
        """scale * dot ** power * E as ONE launch when the product has exactly one other operand,
        a float32 matrix of the result's shape (or broadcast along one of its axes); else None."""
        g = next(a for a in rest if isinstance(a, LazyGemm))
        others = [a for a in rest if a is not g]
        if g.E is not None or len(g.shape) != 2 or g.dtype != torch.float32 or len(others) > 1:
            return None
        E = None
        if others:
            E = others[0]
            if isinstance(E, Lazy):
                if E.dim() != 2 or any(E.shape[a] not in (1, g.shape[a]) for a in range(2)):
                    return None
                E = self._force(E)
            if not isinstance(E, torch.Tensor) or E.dim() != 2 or E.dtype != torch.float32 or \
                    any(E.shape[a] not in (1, g.shape[a]) for a in range(2)):
                return None
        return LazyGemm(g.gemm, g.shape, g.dtype, power=g.power, scale=g.scale * math.prod(host), E=E, pre=g.pre)

    def _plan_for(self, expr):
        """The buffers the previous evaluation of ``expr`` left behind -- or, while a graph is being
        recorded, an empty throw-away plan: the recorded launches keep the ADDRESSES of their
        intermediates, so those must belong to the graph (``_kept``), not to a plan that a later eager
        evaluation rewrites, resizes or evicts."""
        if self._keep is not None:
            return []
        entry = self._plans.get(id(expr))
        if entry is None or entry[0] is not expr:
            if len(self._plans) >= self._MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            entry = self._plans[id(expr)] = [expr, []]
        return entry[1]

    def evaluate(self, expr, inputs, bindings=None):
        self._plan, self._cursor = self._plan_for(expr), 0
        try:
            out = self._force(Backend.evaluate(self, expr, inputs, bindings))
        finally:
            plan, self._plan = self._plan, None
        if isinstance(out, torch.Tensor):
            # the caller owns the result: its storage leaves the plan
            where = out.untyped_storage().data_ptr()
            for k, buf in enumerate(plan):
                if buf is not None and buf.untyped_storage().data_ptr() == where:
                    plan[k] = None
        return out

    # -- hooks --------------------------------------------------------------------------
    def elemwise(self, op_name, *args):
        args = [a.tensor() if isinstance(a, DeferredSoftmax) else a for a in args]   # (an element-wise use writes them)
        if all(isinstance(a, HostScalar) for a in args):
            return HostScalar(self._host_elemwise(op_name, [a.value for a in args]))
        if op_name in ("add", "mul"):
            return self._combine(op_name, args)
        if op_name == "pow":
            if isinstance(args[1], HostScalar):
                return self._unary("pow", args[0], float(args[1].value))
            return self._elemwise("pow", args)
        return self._unary(op_name, args[0])

    def mul(self, *factors):
        deferred = [f for f in factors if isinstance(f, DeferredSoftmax)]
        if len(deferred) == 1 and all(isinstance(f, HostScalar) for f in factors if f is not deferred[0]):
            # a scalar multiple of responsibilities that were not written is still not written
            return deferred[0].scaled(math.prod(float(f.value) for f in factors if f is not deferred[0]))
        return self._combine("mul", [f.tensor() if isinstance(f, DeferredSoftmax) else f for f in factors])

    def sum(self, x, axes):
        if isinstance(x, DeferredSoftmax):
            # column sums of responsibilities that were never written: the ones column of R^T . X_wide
            rows_axis = 1 if x.transposed else 0
            col = x.block(("ones", x.xcat.shape[0])) if [a % 2 for a in axes] == [rows_axis] else None
            if col is not None:
                return col.reshape(x.ycat.shape[1])
            x = x.tensor()
        if isinstance(x, LazyGemm):
            x = self._force(x)
        if isinstance(x, HostScalar):
            return x
        if isinstance(x, Lazy):
            return self._launch(x, axes)
        axes = [a % x.dim() for a in axes]
        keep = [a for a in range(x.dim()) if a not in axes]
        ckey = None
        if self._wide and self._keep is None and x.dim() == 2 and axes == [0] and x.dtype == torch.float32 \
                and ("ones", x.shape[0]) in self._wide and self._is_const(x):
            col = self._product_with_wide(x, x.shape[1], x.shape[0], x.stride(1), x.stride(0),
                                          ("ones", x.shape[0]))
            if col is not None:
                return col.reshape(x.shape[1])
        if (self._const or self._const_ptrs) and self._keep is None and self._is_const(x):
            # a sum of a constant (the column sums of a mixture's responsibilities occur in the
            # messages of three factors): computed once while its operand stays marked
            ckey = ("sum", x.data_ptr(), tuple(x.shape), tuple(x.stride()), tuple(axes), x.dtype)
            hit = self._const_cache.pop(ckey, None)
            if hit is not None:
                self._const_cache[ckey] = hit
                return hit
        out_shape = [x.shape[a] for a in keep]
        small = math.prod(out_shape) * x.element_size() <= (1 << 20)
        out = self.ctx.empty(out_shape, x.dtype) if (ckey is not None and small) else self._empty(out_shape, x.dtype)
        self.ctx.call("bsc_sum", _DT[x.dtype], len(keep), _i64(x.shape[a] for a in keep),
                      _i64(x.stride(a) for a in keep), len(axes), _i64(x.shape[a] for a in axes),
                      _i64(x.stride(a) for a in axes), _ffi.ptr(x), _ffi.ptr(out))
        if ckey is not None and small:
            self._const_cache[ckey] = out
            self._const_bytes += out.numel() * out.element_size()
        return out

    @staticmethod
    def _view_dimshuffle(x, axes):
        y = x.permute([a for a in axes if a != "x"])
        for position, a in enumerate(axes):
            if a == "x":
                y = y.unsqueeze(position)
        return y

    def dimshuffle(self, x, axes):
        if isinstance(x, DeferredSoftmax):
            if tuple(axes) == (1, 0):
                return x.T
            if tuple(axes) == (0, 1):
                return x
            x = x.tensor()
        if isinstance(x, LazyGemm):
            x = self._force(x)
        if isinstance(x, HostScalar):
            return x          # broadcast axes of a scalar are re-created where it is used
        if isinstance(x, Lazy):     # element-wise values commute with views: re-view every operand
            terms = [(self._view_dimshuffle(t, axes), op, arg) for t, op, arg in x.terms]
            shape = [1 if a == "x" else x.shape[a] for a in axes]
            return Lazy(x.combine, terms, shape, x.dtype, x.scale, x.shift, x.post)
        return self._view_dimshuffle(x, axes)

    def diagonal(self, x, axis1, axis2):
        x = self._force(x)
        return torch.diagonal(x, 0, axis1, axis2)     # view; the diagonal axis goes last

    def evaluate_softmax_rows(self, expr, inputs, bindings=None, scale=1.0, defer=False):
        """softmax over the last axis of ``scale`` times the value of ``expr`` (a resident Categorical node's update).
        Returns (R, lse, cross, logits): when the value is a tall-skinny product nothing else reads --
        the logits of a mixture, after _concat_products one product [X | X^2 | 1] . coefficients --
        ONE launch (bsc_gemm_softmax_rows) produces R, lse and cross = sum_c R * logits, and
        ``logits`` is None: they never reach memory.  Otherwise the logits are evaluated as usual and
        returned with their softmax (cross is then None).

        ``defer=True``: in that one-product case with the left operand a cached wide operand, NOTHING is launched:
        R comes back as a ``DeferredSoftmax`` (lse and cross None) whose statistics against constituents of the
        wide operand are formed, when first asked for, in the pass that takes the softmax -- the responsibilities
        are not written at all unless something else wants them."""
        self._plan, self._cursor = self._plan_for(expr), 0
        logits = None
        try:
            root = Backend.evaluate(self, expr, inputs, bindings)
            g = root if (isinstance(root, LazyGemm) and not isinstance(root, LazyLda)) else None
            if g is not None and self._keep is None:
                xb, m, n, k, x, sxb, sxm, sxk, y, syb, syk, syn = g.gemm
                if g.power == 1 and g.E is None and g.pre is None and xb == 1 and sxk == 1 and k % 8 == 0 \
                        and k <= 64 and n <= 64 and n % 4 == 0 and sxm % 4 == 0 and x.data_ptr() % 16 == 0 \
                        and g.dtype == torch.float32 and len(g.shape) == 2:
                    wkey = next((key for key, t in self._const_cache.items() if t is x and key[0] == "kcat"), None) \
                        if defer else None
                    if wkey is not None and syn == 1 and syk == n:
                        # (the stacked coefficients live in a plan buffer the next evaluation rewrites: kept apart)
                        ycat = self.ctx.empty((k, n), torch.float32)
                        self._map_into(ycat, [k, n], [(y, [syk, syn])])
                        return DeferredSoftmax(self, x, wkey, ycat, float(g.scale) * float(scale)), None, None, None
                    R = self.ctx.empty((m, n), torch.float32)
                    lse = self.ctx.empty((m,), torch.float32)
                    cross = self.ctx.empty((m,), torch.float32)
                    self.ctx.call("bsc_gemm_softmax_rows", _ffi.ptr(x), sxm, m, k, _ffi.ptr(y), syk, syn, n,
                                  float(g.scale) * float(scale), _ffi.ptr(R), n, _ffi.ptr(lse), _ffi.ptr(cross))
                    return R, lse, cross, None
            if float(scale) != 1.0:
                root = self._combine("mul", [root, HostScalar(float(scale))])
            logits = self._force(root)
        finally:
            plan, self._plan = self._plan, None
        if isinstance(logits, torch.Tensor):
            where = logits.untyped_storage().data_ptr()
            for k_, buf in enumerate(plan):
                if buf is not None and buf.untyped_storage().data_ptr() == where:
                    plan[k_] = None
        R, lse = self.softmax_rows(logits)
        return R, lse, None, logits

    def softmax_rows(self, x):
        x = self._force(x)
        if x.dtype != torch.float32:
            raise TypeError("softmax_rows: float32 only")
        cols = x.shape[-1]
        x2 = x.reshape(-1, cols) if x.dim() != 2 else x
        if x2.stride(1) != 1:
            x2 = self._contiguous(x2)
        rows = x2.shape[0]
        out = self._empty((rows, cols), torch.float32)
        lse = self._empty((rows,), torch.float32)
        self.ctx.call("bsc_softmax_rows", _ffi.ptr(x2), rows, cols, x2.stride(0), _ffi.ptr(out), cols, _ffi.ptr(lse))
        return out.reshape(x.shape), lse.reshape(x.shape[:-1])

    def materialize(self, value):
        return self._force(value)

    def broadcast_to(self, g, shape):
        """A stride-0 view of g over `shape` (a stride of 0 broadcasts in every C-ABI entry point);
        a host scalar becomes scale * (one resident 1.0 viewed over `shape`), still deferred."""
        shape = [int(d) for d in shape]
        if isinstance(g, HostScalar):
            if self._one is None:
                self._one = self.from_host(np.ones(1, np.float32), "float32", 1)
            base = self._one.expand(shape) if shape else self._one.reshape(())
            return base if float(g.value) == 1.0 else self.mul(g, base)
        g = self._force(g)
        if g.dim() == 0:
            g = g.reshape([1] * len(shape))
        return g.expand(shape)

    def logdet(self, x):
        x = self._force(x)
        lead = list(x.shape[:-2])
        n = x.shape[-1]
        if x.shape[-2] != n:
            raise ValueError("logdet needs square matrices")
        xb = x.reshape([-1, n, n]) if x.dim() != 3 else x     # view when possible
        if not (x.dim() == 3 or xb.data_ptr() == x.data_ptr()):
            xb = self._contiguous(x).reshape([-1, n, n])
        out = self._empty(lead, x.dtype)
        self.ctx.call("bsc_logdet_spd", _DT[x.dtype], xb.shape[0], n, _ffi.ptr(xb), xb.stride(0),
                      xb.stride(1), xb.stride(2), _ffi.ptr(out))
        return out

    def inverse_spd(self, x):
        x = self._force(x)
        n = x.shape[-1]
        if x.dim() < 2 or x.shape[-2] != n:
            raise ValueError("inverse_spd needs square matrices")
        xb = x.reshape([-1, n, n]) if x.dim() != 3 else x     # view when possible
        if not (x.dim() == 3 or xb.data_ptr() == x.data_ptr()):
            xb = self._contiguous(x).reshape([-1, n, n])
        out = self._empty(list(x.shape), x.dtype)
        self.ctx.call("bsc_inverse_spd", _DT[x.dtype], xb.shape[0], n, _ffi.ptr(xb), xb.stride(0), xb.stride(1),
                      xb.stride(2), _ffi.ptr(out), None)
        return out

    @staticmethod
    def _merge(t, axes):
        """(extent, stride) of the axes `axes` of t collapsed into one, or None when
        their strides do not allow it."""
        extent, stride = 1, 0
        live = [(t.shape[a], t.stride(a)) for a in axes if t.shape[a] != 1]
        if not live:
            return (math.prod(t.shape[a] for a in axes) if axes else 1), 0
        for (n_outer, s_outer), (n_inner, s_inner) in zip(live, live[1:]):
            if s_outer != s_inner * n_inner:
                return None
        extent = math.prod(n for n, _ in live)
        stride = live[-1][1]
        return extent, stride

    def _dot_products(self, x, y, x_dot, y_dot, x_batch, y_batch):
        """_tensordot in which y has no free axes: out[batch, x free] = sum_dot x * y in one
        fused pass -- y is broadcast over x's free axes (operands may be deferred
        element-wise values; nothing is materialised)."""
        # line y's axes up with x's, then it is sum(x * y) over x's dot axes
        x, y = (self._force(v) if isinstance(v, LazyGemm) else v for v in (x, y))
        order = ["x"] * x.dim()
        for ax, ay in zip(list(x_batch) + list(x_dot), list(y_batch) + list(y_dot)):
            order[ax] = ay
        y = self.dimshuffle(y, order)
        for ax in list(x_batch) + list(x_dot):
            if x.shape[ax] != y.shape[ax]:
                raise ValueError("tensordot: contracted / batch extents differ (%d vs %d)"
                                 % (x.shape[ax], y.shape[ax]))
        prod = self._combine("mul", [x, y])
        out = self._launch(prod, x_dot)               # kept axes in x's axis order
        kept = [a for a in range(x.dim()) if a not in x_dot]
        x_other = [a for a in kept if a not in x_batch]
        perm = [kept.index(a) for a in list(x_batch) + x_other]
        return out.permute(perm) if perm != sorted(perm) else out

    def _weighted_outer(self, x, y, x_dot, y_dot):
        """sum_n A[k,n] B[d,n] Y[n,e] -- what the front end makes of
        einsum(out_kde = sum_n R_nk X_nd Y_ne): _tensordot(_mul(_dimshuffle(R,1,'x',0),
        _dimshuffle(X,'x',1,0)), Y, [2], [0]) (bayesic/algebra.py:632-636).  Run as one pass of
        bsc_weighted_outer instead of materialising the K x D x N product; None when the
        shapes are not that pattern or outside the kernel's limits."""
        if not (x.combine == "mul" and x.post is None and x.shift == 0.0 and len(x.terms) == 2
                and x.dim() == 3 and len(x_dot) == 1 and len(y_dot) == 1
                and isinstance(y, torch.Tensor) and y.dim() == 2
                and x.dtype == torch.float32 and y.dtype == torch.float32
                and all(op is None and isinstance(t, torch.Tensor) for t, op, _ in x.terms)):
            return None
        n_ax, yd = x_dot[0], y_dot[0]
        ye = 1 - yd
        f0, f1 = [a for a in range(3) if a != n_ax]
        n = x.shape[n_ax]
        if y.shape[yd] != n:
            raise ValueError("tensordot: contracted / batch extents differ (%d vs %d)"
                             % (n, y.shape[yd]))

        def role(t):    # the free axis this factor varies along (it must not vary along the other)
            if t.shape[n_ax] != n:
                return None
            if t.shape[f1] == 1 and t.shape[f0] == x.shape[f0]:
                return f0
            if t.shape[f0] == 1 and t.shape[f1] == x.shape[f1]:
                return f1
            return None

        (ta, _, _), (tb, _, _) = x.terms
        ra, rb = role(ta), role(tb)
        if ra is None or rb is None or ra == rb:
            return None
        if ra != f0:
            ta, tb = tb, ta                         # ta varies along f0, tb along f1
        # the kernel takes K <= 64 on its first factor and D <= 32 on its second
        if x.shape[f0] <= 64 and x.shape[f1] <= 32:
            r, ax_r, xx, ax_x, swap = ta, f0, tb, f1, False
        elif x.shape[f1] <= 64 and x.shape[f0] <= 32:
            r, ax_r, xx, ax_x, swap = tb, f1, ta, f0, True
        else:
            return None
        K, D, E = r.shape[ax_r], xx.shape[ax_x], y.shape[ye]
        ldr, ldx, ldy = r.stride(n_ax), xx.stride(n_ax), y.stride(yd)
        if K % 4 or D % 4 or E % 4 or E > 32 or n == 0 or \
                r.stride(ax_r) != 1 or xx.stride(ax_x) != 1 or y.stride(ye) != 1 or \
                ldr % 4 or ldx % 4 or ldy % 4 or ldr < K or ldx < D or ldy < E or \
                max(ldr, ldx, ldy) >= 1 << 22 or \
                (r.data_ptr() | xx.data_ptr() | y.data_ptr()) % 16:
            return None
        out = self._empty((K, D, E), torch.float32)
        self.ctx.call("bsc_weighted_outer", _ffi.ptr(r), ldr, _ffi.ptr(xx), ldx, _ffi.ptr(y), ldy,
                      n, K, D, E, float(x.scale), _ffi.ptr(out))
        return out.permute(1, 0, 2) if swap else out

    def tensordot(self, x, y, x_dot, y_dot, x_batch, y_batch):
        if self.fuse and isinstance(y, LazyGemm) and not isinstance(y, LazyLda) and y.power == -1 and y.E is not None \
                and not x_batch and not y_batch and self._plan is not None:
            lda = self._lda_pattern(x, y, x_dot, y_dot)
            if lda is not None:
                return lda
        if isinstance(x, DeferredSoftmax) or isinstance(y, DeferredSoftmax):
            # (the partner may be a deferred element-wise value of constants -- X * X --, whose forced value is
            # the cached constituent of the wide operand)
            x = x if isinstance(x, DeferredSoftmax) else self._force(x)
            y = y if isinstance(y, DeferredSoftmax) else self._force(y)
            out = self._deferred_statistics(x, y, x_dot, y_dot, x_batch, y_batch)
            if out is not None:
                return out
            x, y = self._force(x), self._force(y)
        free_x = x.dim() - len(x_dot) - len(x_batch)
        free_y = y.dim() - len(y_dot) - len(y_batch)
        if self.fuse:
            if free_x == 0 and free_y == 0:                # (batched) dot products
                return self._dot_products(x, y, x_dot, y_dot, x_batch, y_batch)
            # matrix-vector shapes with a deferred element-wise operand: one fused pass instead
            # of materialising it for the GEMV (sum_n exp(X)_nd u_n and the like)
            if free_y == 0 and isinstance(x, Lazy):
                return self._dot_products(x, y, x_dot, y_dot, x_batch, y_batch)
            if free_x == 0 and isinstance(y, Lazy):
                return self._dot_products(y, x, y_dot, x_dot, y_batch, x_batch)
            if isinstance(x, Lazy) and not x_batch and not y_batch:
                out = self._weighted_outer(x, y, x_dot, y_dot)
                if out is not None:
                    return out
        pre = None
        if self.fuse and self._plan is not None and not x_batch and not y_batch and free_x >= 1 and free_y >= 1:
            # an element-wise producer of a matrix-matrix product's operand goes into the product (bsc_gemm_fused)
            px, py = self._prologue_of(x), self._prologue_of(y)
            if (px or py) and not (px and py and px[1] == 2 and py[1] == 2) and \
                    (px or not isinstance(x, (Lazy, LazyGemm, DeferredSoftmax))) and \
                    (py or not isinstance(y, (Lazy, LazyGemm, DeferredSoftmax))):
                pre = (px[1] if px else 0, py[1] if py else 0, (px[2] if px else 1.0) * (py[2] if py else 1.0))
                x = px[0] if px else x
                y = py[0] if py else y
        x, y = self._force(x), self._force(y)
        (x, y), dtype, _ = self._common([x, y]) if x.dtype != y.dtype else ((x, y), x.dtype, 0)
        x_other = [a for a in range(x.dim()) if a not in x_dot and a not in x_batch]
        y_other = [a for a in range(y.dim()) if a not in y_dot and a not in y_batch]
        out_shape = [x.shape[a] for a in x_batch] + [x.shape[a] for a in x_other] + \
                    [y.shape[a] for a in y_other]

        def groups(t, batch, free, dot):
            merged = [self._merge(t, g) for g in (batch, free, dot)]
            if any(m is None for m in merged):
                t = self._convert(t.permute(list(batch) + list(free) + list(dot)), t.dtype)
                nb, nf = len(batch), len(free)
                merged = [self._merge(t, range(0, nb)), self._merge(t, range(nb, nb + nf)),
                          self._merge(t, range(nb + nf, t.dim()))]
            return t, merged

        x, ((xb, sxb), (m, sxm), (k, sxk)) = groups(x, x_batch, x_other, x_dot)
        y, ((yb, syb), (n, syn), (k2, syk)) = groups(y, y_batch, y_other, y_dot)
        if k != k2 or xb != yb:
            raise ValueError("tensordot: contracted / batch extents differ (%d vs %d, %d vs %d)"
                             % (k, k2, xb, yb))
        if self.fuse and dtype == torch.float32 and len(out_shape) == 2 and xb == 1 and self._wide and \
                self._keep is None:
            # (y must still be the constant the wide operand was built from: a cached value that has
            # been evicted may have handed its address to something else)
            part = self._product_with_wide(x, m, k, sxm, sxk,
                                           (y.data_ptr(), n, syk) if (syn == 1 and self._is_const(y)) else None)
            if part is not None:
                return part
        if self.fuse and dtype == torch.float32 and len(out_shape) == 2 and xb == 1 and m > 1 and n > 1 \
                and k > 0 and self._plan is not None:
            # deferred: a _mul / pow(., -1) consumer folds into the store (inside evaluate() only --
            # a value handed to the caller is always a tensor)
            return LazyGemm((xb, m, n, k, x, sxb, sxm, sxk, y, syb, syk, syn), out_shape, dtype,
                            scale=pre[2] if pre else 1.0, pre=pre[:2] if pre else None)
        if pre is not None:      # (not a deferred product after all: the producers are launched)
            x2, y2 = self._apply_pre(x, pre[0]), self._apply_pre(y, pre[1])
            (sxb, sxm, sxk), (syb, syk, syn) = self._restride(x, x2, sxb, sxm, sxk), self._restride(y, y2, syb, syk, syn)
            x, y = x2, y2
            if pre[2] != 1.0:
                x = self._force(self._combine("mul", [HostScalar(pre[2]), x]))
        out = self._empty(out_shape, dtype)
        self.ctx.call("bsc_gemm_strided_batched", _DT[dtype], xb, m, n, k,
                      _ffi.ptr(x), sxb, sxm, sxk, _ffi.ptr(y), syb, syk, syn,
                      _ffi.ptr(out), m * n, n, 1)
        return out
