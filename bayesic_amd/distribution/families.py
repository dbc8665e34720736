"""Further exponential-family nodes behind the contract of
bayesic/distribution/base.py:271-314 (sufficient_statistics / natural_parameters
of matching shapes, log_normalizer, data term): Gamma, InverseGamma, Bernoulli,
Categorical, Multinomial, Dirichlet, Wishart -- the priors and likelihoods the
BASELINE configurations use.  The reference defines only
Normal and MultivariateNormal (bayesic/distribution/core.py); these follow the
same pattern and are pinned by scipy.stats in tests/test_distribution.py.

Each family also gives ``expected_sufficient_statistics(**params)``: E[t(x)]
under the family itself, which is what a mean-field / VMP message carries
(README.md:30-37).  All methods accept extra leading (observation) dimensions.
"""
import math

import numpy as np

from .. import algebra as A
from .base import ExponentialFamily, _sum_trailing
from .core import floatX, logdet
from .special import digamma, gammaln


def _w(*xs):
    out = tuple(A.wrap_if_literal(x) for x in xs)
    return out if len(out) > 1 else out[0]


def _keep_last(x):
    """[..., ] -> [..., 1-broadcast]: re-insert a summed trailing axis."""
    return A.dimshuffle(x, *(list(range(x.ndim)) + ["x"]))


class Gamma(ExponentialFamily):
    """Gamma(shape a, rate b):  log p = (a-1) log x - b x - [lgamma(a) - a log b]."""

    parameter_types = dict(shape=(floatX, 0), rate=(floatX, 0))
    data_type = (floatX, 0)

    def statistic_ndims(self):
        return [0, 0]

    def sufficient_statistics(self, data):
        data = _w(data)
        return A.log(data), data

    def natural_parameters(self, shape, rate):
        shape, rate = _w(shape, rate)
        return shape - 1, -rate

    def log_normalizer(self, shape, rate, data_shape=None):
        shape, rate = _w(shape, rate)
        return gammaln(shape) - shape * A.log(rate)

    def log_likelihood_data_term(self, data):
        return 0

    def expected_sufficient_statistics(self, shape, rate):
        shape, rate = _w(shape, rate)
        return digamma(shape) - A.log(rate), shape / rate


class InverseGamma(ExponentialFamily):
    """InverseGamma(shape a, scale b):  log p = -(a+1) log x - b / x - [lgamma(a) - a log b]."""

    parameter_types = dict(shape=(floatX, 0), scale=(floatX, 0))
    data_type = (floatX, 0)

    def statistic_ndims(self):
        return [0, 0]

    def sufficient_statistics(self, data):
        data = _w(data)
        return A.log(data), data ** -1

    def natural_parameters(self, shape, scale):
        shape, scale = _w(shape, scale)
        return -shape - 1, -scale

    def log_normalizer(self, shape, scale, data_shape=None):
        shape, scale = _w(shape, scale)
        return gammaln(shape) - shape * A.log(scale)

    def log_likelihood_data_term(self, data):
        return 0

    def expected_sufficient_statistics(self, shape, scale):
        shape, scale = _w(shape, scale)
        return A.log(scale) - digamma(shape), shape / scale


class Bernoulli(ExponentialFamily):
    """Bernoulli(probability p) on x in {0, 1}:  log p = x logit(p) + log(1 - p)."""

    parameter_types = dict(probability=(floatX, 0))
    data_type = ("int8", 0)

    def statistic_ndims(self):
        return [0]

    def sufficient_statistics(self, data):
        return (_w(data),)

    def natural_parameters(self, probability):
        p = _w(probability)
        return (A.log(p) - A.log(1 - p),)

    def log_normalizer(self, probability, data_shape=None):
        return -A.log(1 - _w(probability))

    def log_likelihood_data_term(self, data):
        return 0

    def expected_sufficient_statistics(self, probability):
        return (_w(probability),)


class Categorical(ExponentialFamily):
    """Categorical(probabilities p[K]) on one-hot x[K]:  log p = <x, log p> - log sum_k p_k
    (the normaliser is zero for normalised p and makes unnormalised weights legal)."""

    parameter_types = dict(probabilities=(floatX, 1))
    data_type = ("int8", 1)

    def statistic_ndims(self):
        return [1]

    def sufficient_statistics(self, data):
        return (_w(data),)

    def natural_parameters(self, probabilities):
        return (A.log(_w(probabilities)),)

    def log_normalizer(self, probabilities, data_shape=None):
        return A.log(_sum_trailing(_w(probabilities), 1))

    def log_likelihood_data_term(self, data):
        return 0

    def expected_sufficient_statistics(self, probabilities):
        p = _w(probabilities)
        return (p / _keep_last(_sum_trailing(p, 1)),)


class Multinomial(Categorical):
    """Multinomial(total_count n fixed, probabilities p[K]) on count vectors x[K]:
    log p = <x, log p> - n log sum_k p_k + lgamma(n+1) - sum_k lgamma(x_k + 1)."""

    def __init__(self, total_count):
        self.total_count = total_count

    data_type = ("int32", 1)

    def log_normalizer(self, probabilities, data_shape=None):
        return self.total_count * A.log(_sum_trailing(_w(probabilities), 1))

    def log_likelihood_data_term(self, data):
        data = _w(data)
        return math.lgamma(self.total_count + 1.0) - _sum_trailing(gammaln(data + 1), 1)

    def expected_sufficient_statistics(self, probabilities):
        (p,) = Categorical.expected_sufficient_statistics(self, probabilities)
        return (self.total_count * p,)


class Dirichlet(ExponentialFamily):
    """Dirichlet(concentration alpha[K]) on the simplex:
    log p = <log x, alpha - 1> - [sum_k lgamma(alpha_k) - lgamma(sum_k alpha_k)]."""

    parameter_types = dict(concentration=(floatX, 1))
    data_type = (floatX, 1)

    def statistic_ndims(self):
        return [1]

    def sufficient_statistics(self, data):
        return (A.log(_w(data)),)

    def natural_parameters(self, concentration):
        return (_w(concentration) - 1,)

    def log_normalizer(self, concentration, data_shape=None):
        alpha = _w(concentration)
        return _sum_trailing(gammaln(alpha), 1) - gammaln(_sum_trailing(alpha, 1))

    def log_likelihood_data_term(self, data):
        return 0

    def expected_sufficient_statistics(self, concentration):
        alpha = _w(concentration)
        return (digamma(alpha) - _keep_last(digamma(_sum_trailing(alpha, 1))),)


class Wishart(ExponentialFamily):
    """Wishart over D x D SPD matrices with degrees of freedom nu and RATE matrix
    W = scale^{-1} (so no matrix inverse is needed):
    log p = (nu-D-1)/2 log det X - 1/2 tr(W X)
            - [nu D / 2 log 2 - nu/2 log det W + log Gamma_D(nu / 2)]."""

    def __init__(self, dim):
        self.dim = int(dim)

    parameter_types = dict(dof=(floatX, 0), rate=(floatX, 2))
    data_type = (floatX, 2)

    def statistic_ndims(self):
        return [0, 2]

    def sufficient_statistics(self, data):
        data = _w(data)
        return logdet(data), data

    def natural_parameters(self, dof, rate):
        dof, rate = _w(dof, rate)
        return 0.5 * (dof - (self.dim + 1)), -0.5 * rate

    def log_normalizer(self, dof, rate, data_shape=None):
        dof, rate = _w(dof, rate)
        D = self.dim
        n = dof.ndim
        offsets = A.constant(-0.5 * np.arange(D, dtype=np.float64))
        offsets = A.dimshuffle(offsets, *(["x"] * n + [0])) if n else offsets
        half = 0.5 * dof
        log_gamma_d = 0.25 * D * (D - 1) * math.log(math.pi) + \
            _sum_trailing(gammaln(_keep_last(half) + offsets), 1)
        return (0.5 * D * math.log(2.0)) * dof - half * logdet(rate) + log_gamma_d

    def log_likelihood_data_term(self, data):
        return 0
