"""Core exponential-family nodes: Normal(mean, variance) and
MultivariateNormal(mean, precision).

Sufficient statistics and natural parameters follow the reference
(bayesic/distribution/core.py:16-20, 41-47).  Its log-normalisers are wrong as
written (:22-25 has the sign of the log 2 pi term flipped and (mean/variance)**2
where mean**2/variance is meant; :49-52 likewise) and its MultivariateNormal
refers to undefined names (:44,46-52); the forms used here are the correct ones
and are checked against scipy.stats in tests/test_distribution.py:

    Normal:  A = 1/2 log 2pi + 1/2 log var + mean^2 / (2 var)
    MVN:     A = D/2 log 2pi - 1/2 log det Lambda + 1/2 mean^T Lambda mean

All methods accept extra leading (observation) dimensions.
"""
import math

from .. import algebra as A
from ..algebra.expr import Expression
from .base import ExponentialFamily

floatX = "float32"   # the reference imports this from a module that does not exist (core.py:3)
_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


class logdet(Expression):
    """log det of the trailing two axes of a symmetric positive-definite matrix
    (the reference calls Theano's T.logdet, core.py:50)."""

    def __init__(self, X):
        X = A.wrap_if_literal(X)
        if X.ndim < 2:
            raise ValueError("logdet needs at least a matrix")
        self.ndim = X.ndim - 2
        super(logdet, self).__init__([X])

    def _emit(self, backend, X):
        return backend.logdet(X)


class Normal(ExponentialFamily):
    """Scalar Gaussian given by mean and variance."""

    parameter_types = dict(mean=(floatX, 0), variance=(floatX, 0))
    data_type = (floatX, 0)

    def statistic_ndims(self):
        return [0, 0]

    def sufficient_statistics(self, data):
        data = A.wrap_if_literal(data)
        return data, data ** 2

    def natural_parameters(self, mean, variance):
        mean, variance = A.wrap_if_literal(mean), A.wrap_if_literal(variance)
        return mean / variance, -0.5 / variance

    def log_normalizer(self, mean, variance, data_shape=None):
        mean, variance = A.wrap_if_literal(mean), A.wrap_if_literal(variance)
        return _HALF_LOG_2PI + 0.5 * A.log(variance) + 0.5 * (mean ** 2 / variance)

    def log_likelihood_data_term(self, data):
        return 0

    def expected_sufficient_statistics(self, mean, variance):
        """E[(x, x^2)] under N(mean, variance) -- what a mean-field message carries."""
        mean, variance = A.wrap_if_literal(mean), A.wrap_if_literal(variance)
        return mean, mean ** 2 + variance


def _lead(n, *trailing):
    """Index list: n shared leading out axes followed by the given trailing indices."""
    return [("out", i) for i in range(n)] + list(trailing)


class MultivariateNormal(ExponentialFamily):
    """Multivariate Gaussian given by mean [D] and precision matrix [D, D]."""

    parameter_types = dict(mean=(floatX, 1), precision=(floatX, 2))
    data_type = (floatX, 1)

    def statistic_ndims(self):
        return [1, 2]

    def sufficient_statistics(self, data):
        data = A.wrap_if_literal(data)
        n = data.ndim - 1
        second = A.einsum([(data, _lead(n, ("out", n))), (data, _lead(n, ("out", n + 1)))], n + 2)
        return data, second

    def natural_parameters(self, mean, precision):
        mean, precision = A.wrap_if_literal(mean), A.wrap_if_literal(precision)
        n = mean.ndim - 1
        first = A.einsum([(precision, _lead(n, ("out", n), ("sum", 0))),
                          (mean, _lead(n, ("sum", 0)))], n + 1)
        return first, -0.5 * precision

    def log_normalizer(self, mean, precision, data_shape=None):
        mean, precision = A.wrap_if_literal(mean), A.wrap_if_literal(precision)
        n = mean.ndim - 1
        D = data_shape[0] if data_shape is not None else A.shape(mean, n)
        quad = A.einsum([(mean, _lead(n, ("sum", 0))),
                         (precision, _lead(n, ("sum", 0), ("sum", 1))),
                         (mean, _lead(n, ("sum", 1)))], n)
        return D * _HALF_LOG_2PI - 0.5 * logdet(precision) + 0.5 * quad

    def log_likelihood_data_term(self, data):
        return 0
