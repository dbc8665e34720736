"""Host float64 evaluation of PARAMETER-SIZED expressions -- model recognition only.

``inference/recognise.py`` decides at construction time whether a symbolic log-joint belongs to a
family one of the fused kernels computes (csrc/bsc_blr.hip, bsc_mog.hip, bsc_bbvi.hip).  The
data-sized structure is read off with ``match`` (bayesic/algebra.py:1037-1063); what is left are
scalar coefficients and functions of a handful of latent numbers, which are identified by
evaluating them at a few probe points.  That evaluation happens here, in numpy float64, exactly as
``ReparamVI`` / ``ScoreFunctionVI`` already keep their S x P parameters and Adam state on the host.

This is NOT an execution backend: ``resolve_backend`` never returns it, it refuses any operand
above ``MAX_ELEMENTS`` (data belongs on the device and never comes near it -- data inputs are
represented by ``ShapeOnly`` stand-ins that carry an extent and nothing else), and no update,
statistic or bound is ever computed with it.
"""
import numpy as np

from ..algebra.backend import Backend

MAX_ELEMENTS = 1 << 16


class ShapeOnly(object):
    """A data input as far as recognition may see it: its extents."""

    def __init__(self, shape):
        self.shape = tuple(int(n) for n in shape)
        self.ndim = len(self.shape)


class ParameterBackend(Backend):
    name = "parameter-probe"

    def _small(self, a):
        if isinstance(a, ShapeOnly):
            raise ValueError("a data input reached arithmetic in the parameter probe: the term is not "
                             "parameter-sized")
        a = np.asarray(a, np.float64)
        if a.size > MAX_ELEMENTS:
            raise ValueError("operand of %d elements in the parameter probe (limit %d): data-sized work "
                             "belongs on the device" % (a.size, MAX_ELEMENTS))
        return a

    def from_host(self, array, dtype, ndim):
        if isinstance(array, ShapeOnly):
            return array
        a = self._small(array)
        if a.ndim != ndim:
            raise ValueError("expected ndim %d, got %d" % (ndim, a.ndim))
        return a

    def to_host(self, value):
        return np.asarray(value, np.float64)

    def constant(self, value):
        return self._small(value)

    def shape(self, x, axis):
        return np.asarray(float(x.shape[axis]))

    def eye(self, n):
        return np.eye(int(n))

    def elemwise(self, op_name, *args):
        args = [self._small(a) for a in args]
        if op_name == "add":
            out = args[0]
            for a in args[1:]:
                out = out + a
            return out
        with np.errstate(all="ignore"):
            if op_name == "log":
                return np.log(args[0])
            if op_name == "exp":
                return np.exp(args[0])
            if op_name == "pow":
                return np.power(args[0], args[1])
            if op_name == "abs_":
                return np.abs(args[0])
            if op_name == "gammaln":
                import math
                return np.vectorize(math.lgamma, otypes=[np.float64])(args[0])
            if op_name == "digamma":
                from scipy.special import digamma
                return digamma(args[0])
        raise ValueError("element-wise op %r is not known to the parameter probe" % op_name)

    def sum(self, x, axes):
        return self._small(x).sum(axis=tuple(axes))

    def mul(self, *factors):
        out = self._small(factors[0])
        for f in factors[1:]:
            out = out * self._small(f)
        return out

    def dimshuffle(self, x, axes):
        x = self._small(x)
        y = np.transpose(x, [a for a in axes if a != "x"])
        for position, a in enumerate(axes):
            if a == "x":
                y = np.expand_dims(y, position)
        return y

    def tensordot(self, x, y, x_dot, y_dot, x_batch, y_batch):
        x, y = self._small(x), self._small(y)
        if not x_batch:
            return np.tensordot(x, y, (list(x_dot), list(y_dot)))
        letters = iter("abcdefghijklmnopqrstuvwxyz")
        xs, ys = [None] * x.ndim, [None] * y.ndim
        batch = []
        for xa, ya in zip(x_batch, y_batch):
            xs[xa] = ys[ya] = c = next(letters)
            batch.append(c)
        for xa, ya in zip(x_dot, y_dot):
            xs[xa] = ys[ya] = next(letters)
        x_free, y_free = [], []
        for idx, free in ((xs, x_free), (ys, y_free)):
            for i, c in enumerate(idx):
                if c is None:
                    idx[i] = next(letters)
                    free.append(idx[i])
        return np.einsum("%s,%s->%s" % ("".join(xs), "".join(ys), "".join(batch + x_free + y_free)), x, y)

    def diagonal(self, x, axis1, axis2):
        return np.diagonal(self._small(x), 0, axis1, axis2)

    def broadcast_to(self, g, shape):
        return np.broadcast_to(np.asarray(g, np.float64), tuple(shape))

    def inverse_spd(self, x):
        return np.linalg.inv(self._small(x))
