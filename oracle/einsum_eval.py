"""float64/float32 numpy executor of the algebra front end (oracle; TEST
INFRASTRUCTURE -- never imported by bayesic_amd).

Two independent evaluators:

* ``NumpyBackend`` implements the backend hook set (bayesic_amd/algebra/backend.py)
  with the numpy calls that correspond to the Theano calls of the reference
  (bayesic/algebra.py:132-134,155,217,258,1291,1306,1319,1351,1407,1436-1448).
  numpy is the reference's own numeric oracle: every numeric test there compares
  against a numpy expression (bayesic/tests/test_algebra.py:44-191).  Batched
  ``_tensordot`` -- broken in the reference (:1358-1383) -- is defined by the
  einsum semantics (:334-338).
* ``einsum_semantics`` evaluates an Einsum straight from its definition
  T_out = sum_{sum indices} prod factors (:334-338) with ``np.einsum``, without
  going through the lowering at all -- the check on the planner.
"""
import numpy as np

from bayesic_amd.algebra.backend import Backend


class NumpyBackend(Backend):
    name = "numpy-oracle"

    def __init__(self, dtype=None):
        self.force_dtype = dtype   # e.g. np.float64 to evaluate everything in double

    def from_host(self, array, dtype, ndim):
        a = np.asarray(array, dtype=self.force_dtype or dtype)
        if a.ndim != ndim:
            raise ValueError("expected ndim %d, got %d" % (ndim, a.ndim))
        return a

    def to_host(self, value):
        return np.asarray(value)

    def constant(self, value):
        a = np.asarray(value)
        if self.force_dtype is not None and a.dtype.kind == "f":
            a = a.astype(self.force_dtype)
        return a

    def shape(self, x, axis):
        return np.asarray(x.shape[axis])

    def eye(self, n):
        return np.eye(int(n), dtype=self.force_dtype or np.float32)

    def elemwise(self, op_name, *args):
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
                from scipy.special import gammaln
                return gammaln(args[0])
            if op_name == "digamma":
                from scipy.special import digamma
                return digamma(args[0])
        raise ValueError("unknown elementwise op %r" % op_name)

    def sum(self, x, axes):
        return x.sum(axis=tuple(axes))

    def mul(self, *factors):
        out = factors[0]
        for f in factors[1:]:
            out = out * f
        return out

    def dimshuffle(self, x, axes):
        perm = [a for a in axes if a != "x"]
        y = np.transpose(x, perm)
        for position, a in enumerate(axes):
            if a == "x":
                y = np.expand_dims(y, position)
        return y

    def tensordot(self, x, y, x_dot, y_dot, x_batch, y_batch):
        if not x_batch:
            return np.tensordot(x, y, (list(x_dot), list(y_dot)))
        # einsum semantics: result axes = batch, x others, y others
        letters = iter("abcdefghijklmnopqrstuvwxyz")
        xs, ys = [None] * x.ndim, [None] * y.ndim
        batch_letters = []
        for xa, ya in zip(x_batch, y_batch):
            c = next(letters)
            xs[xa] = ys[ya] = c
            batch_letters.append(c)
        for xa, ya in zip(x_dot, y_dot):
            c = next(letters)
            xs[xa] = ys[ya] = c
        x_other, y_other = [], []
        for i in range(x.ndim):
            if xs[i] is None:
                xs[i] = next(letters)
                x_other.append(xs[i])
        for i in range(y.ndim):
            if ys[i] is None:
                ys[i] = next(letters)
                y_other.append(ys[i])
        spec = "%s,%s->%s" % ("".join(xs), "".join(ys),
                              "".join(batch_letters + x_other + y_other))
        return np.einsum(spec, x, y)

    def diagonal(self, x, axis1, axis2):
        return np.diagonal(x, 0, axis1, axis2)

    def broadcast_to(self, g, shape):
        return np.broadcast_to(np.asarray(g), tuple(shape))

    def logdet(self, x):
        sign, value = np.linalg.slogdet(np.asarray(x, dtype=np.float64))
        return value.astype(x.dtype)

    def inverse_spd(self, x):
        x = np.asarray(x)
        return np.linalg.inv(0.5 * (x + np.swapaxes(x, -1, -2)).astype(np.float64)).astype(x.dtype)

    def softmax_rows(self, x):
        x = np.asarray(x)
        m = x.max(axis=-1, keepdims=True)
        w = np.exp(x - m)
        z = w.sum(axis=-1, keepdims=True)
        return w / z, (m + np.log(z))[..., 0]


def einsum_semantics(e, inputs, dtype=np.float64):
    """Evaluate an (un-lowered) Einsum from its definition; factors that are not
    einsums themselves are evaluated with NumpyBackend."""
    from bayesic_amd.algebra import Einsum
    backend = NumpyBackend(dtype)
    if not isinstance(e, Einsum):
        return backend.evaluate(e, inputs)
    letters = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
    names = {}

    def name(index):
        if index not in names:
            names[index] = letters[len(names)]
        return names[index]

    operands, specs = [], []
    for factor, indices in e.factors_and_indices:
        operands.append(np.asarray(backend.evaluate(factor, inputs), dtype=dtype))
        specs.append("".join(name(i) for i in indices))
    present = [("out", n) for n in range(e.ndim) if ("out", n) in names]
    out_spec = "".join(name(i) for i in present)
    if not operands:
        result = np.asarray(1.0, dtype=dtype)
    else:
        result = np.einsum(",".join(specs) + "->" + out_spec, *operands)
    # an out index that never occurs is a broadcastable (size-1) axis
    for n in range(e.ndim):
        if ("out", n) not in names:
            result = np.expand_dims(result, n)
    return result
