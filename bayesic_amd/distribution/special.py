"""Special functions as opaque element-wise nodes of the algebra front end
(``elemwise`` accepts any op object with ``.scalar_op.name``,
bayesic/algebra.py:195-209).  They are needed by the log-normalisers and
expectations of the Gamma / Dirichlet / Wishart families and are deliberately
not part of ``bayesic_amd.algebra``'s public names, which mirror the reference's.
On the device they are unary ops of bsc_map_reduce (BSC_OP_LGAMMA, BSC_OP_DIGAMMA)
and fuse with their neighbours like log and exp do."""
from ..algebra.expr import ElementwiseOp, elemwise

_GAMMALN = ElementwiseOp("gammaln")
_DIGAMMA = ElementwiseOp("digamma")


def gammaln(X):
    """log Gamma(X), element-wise."""
    return elemwise(_GAMMALN, X)


def digamma(X):
    """d/dx log Gamma(X), element-wise."""
    return elemwise(_DIGAMMA, X)
