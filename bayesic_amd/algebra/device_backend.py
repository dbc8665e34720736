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

dtype rules (the reference's came from Theano and are pinned by no test):
float32 unless any operand is float64; Python scalars take the dtype of the
tensors they meet; integer arrays are converted to float64 on upload.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _ffi
from ..device import Context, default_context
from .backend import Backend

_OPS = {"add": 0, "mul": 1, "log": 2, "exp": 3, "pow": 4, "abs_": 5, "copy": 6}
_DT = {torch.float32: 0, torch.float64: 1}
_MAX_RANK = 6


class HostScalar(object):
    """0-d value kept on the host (shape arithmetic, literals)."""

    def __init__(self, value):
        self.value = value

    @property
    def ndim(self):
        return 0


def _i64(values):
    values = list(values)
    return (ctypes.c_int64 * max(len(values), 1))(*values)


class DeviceBackend(Backend):
    name = "mi355x-hip"

    def __init__(self, ctx=None):
        self.ctx = ctx if ctx is not None else default_context()
        if not isinstance(self.ctx, Context):
            raise TypeError("ctx must be a bayesic_amd.device.Context")

    # -- host <-> device ---------------------------------------------------------
    def from_host(self, array, dtype, ndim):
        a = np.asarray(array)
        if a.ndim != ndim:
            raise ValueError("input has ndim %d, expected %d" % (a.ndim, ndim))
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
        return torch.from_numpy(a).to(self.ctx.device)

    def to_host(self, value):
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
        self.ctx.sync()
        return int(v.item())

    def eye(self, n):
        n = self._host_int(n)
        out = torch.empty((n, n), dtype=torch.float32, device=self.ctx.device)
        self.ctx.call("bsc_eye", 0, _ffi.ptr(out), n)
        return out

    # -- helpers ----------------------------------------------------------------------
    def _upload_scalar(self, s, dtype, ndim):
        t = torch.tensor(float(s.value), dtype=dtype, device=self.ctx.device)
        return t.reshape((1,) * ndim)

    def _convert(self, t, dtype):
        out = torch.empty(t.shape, dtype=dtype, device=self.ctx.device)
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
        raise ValueError("unknown elementwise op %r" % op_name)

    def _elemwise(self, op_name, args):
        if all(isinstance(a, HostScalar) for a in args):
            return HostScalar(self._host_elemwise(op_name, [a.value for a in args]))
        if op_name in ("add", "mul") and len(args) > 8:      # kernel takes up to 8 inputs
            head = self._elemwise(op_name, args[:8])
            return self._elemwise(op_name, [head] + list(args[8:]))
        args, dtype, ndim = self._common(args)
        if ndim > _MAX_RANK:
            raise _ffi.BayesicHipError("rank %d exceeds the kernels' limit %d" % (ndim, _MAX_RANK))
        shape = []
        for axis in range(ndim):
            extents = {a.shape[axis] for a in args}
            big = extents - {1}
            if len(big) > 1:
                raise ValueError("shapes do not broadcast on axis %d: %s" % (axis, sorted(extents)))
            shape.append(big.pop() if big else 1)
        out = torch.empty(shape, dtype=dtype, device=self.ctx.device)
        strides = []
        for a in args:
            strides += [0 if (a.shape[ax] == 1 and shape[ax] != 1) else a.stride(ax)
                        for ax in range(ndim)]
        ptrs = (ctypes.c_void_p * len(args))(*[a.data_ptr() for a in args])
        self.ctx.call("bsc_elemwise", _OPS[op_name], _DT[dtype], ndim, _i64(shape), _ffi.ptr(out),
                      _i64(out.stride()), len(args), ptrs, _i64(strides))
        return out

    # -- hooks --------------------------------------------------------------------------
    def elemwise(self, op_name, *args):
        return self._elemwise(op_name, list(args))

    def mul(self, *factors):
        return self._elemwise("mul", list(factors))

    def sum(self, x, axes):
        if isinstance(x, HostScalar):
            return x
        axes = [a % x.dim() for a in axes]
        keep = [a for a in range(x.dim()) if a not in axes]
        out = torch.empty([x.shape[a] for a in keep], dtype=x.dtype, device=self.ctx.device)
        self.ctx.call("bsc_sum", _DT[x.dtype], len(keep), _i64(x.shape[a] for a in keep),
                      _i64(x.stride(a) for a in keep), len(axes), _i64(x.shape[a] for a in axes),
                      _i64(x.stride(a) for a in axes), _ffi.ptr(x), _ffi.ptr(out))
        return out

    def dimshuffle(self, x, axes):
        if isinstance(x, HostScalar):
            return x          # broadcast axes of a scalar are re-created where it is used
        y = x.permute([a for a in axes if a != "x"])
        for position, a in enumerate(axes):
            if a == "x":
                y = y.unsqueeze(position)
        return y

    def diagonal(self, x, axis1, axis2):
        return torch.diagonal(x, 0, axis1, axis2)     # view; the diagonal axis goes last

    def logdet(self, x):
        lead = list(x.shape[:-2])
        n = x.shape[-1]
        if x.shape[-2] != n:
            raise ValueError("logdet needs square matrices")
        xb = x.reshape([-1, n, n]) if x.dim() != 3 else x     # view when possible
        if not (x.dim() == 3 or xb.data_ptr() == x.data_ptr()):
            xb = self._contiguous(x).reshape([-1, n, n])
        out = torch.empty(lead, dtype=x.dtype, device=self.ctx.device)
        self.ctx.call("bsc_logdet_spd", _DT[x.dtype], xb.shape[0], n, _ffi.ptr(xb), xb.stride(0),
                      xb.stride(1), xb.stride(2), _ffi.ptr(out))
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

    def tensordot(self, x, y, x_dot, y_dot, x_batch, y_batch):
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
        out = torch.empty(out_shape, dtype=dtype, device=self.ctx.device)
        self.ctx.call("bsc_gemm_strided_batched", _DT[dtype], xb, m, n, k,
                      _ffi.ptr(x), sxb, sxm, sxk, _ffi.ptr(y), syb, syk, syn,
                      _ffi.ptr(out), m * n, n, 1)
        return out
