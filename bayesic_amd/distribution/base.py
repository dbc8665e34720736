"""Distribution / exponential-family node interface.

The reference file (bayesic/distribution/base.py) is an interface SKETCH that
does not parse (SyntaxError at line 205; undefined names at 212, 231-233, 290,
335).  This module implements the contract its docstrings state, on top of
bayesic_amd.algebra expressions instead of raw Theano calls:

  * log_likelihood = data term + interaction term - log normaliser (:47-69,98-100);
  * with extra LEADING dimensions on data and parameters the result has one value
    per observation, shape == those leading dimensions (:72-95);
  * ExponentialFamily: interaction term = sum_j <s_j(x), eta_j(theta)> (:271-291);
  * IndependentObservations: parameter copies and iid draws as leading dimensions,
    terms summed; the normaliser is computed once per parameter copy and multiplied
    by the number of iid draws (:226-244);
  * ExpFamIndependentObservations: sufficient statistics summed over the iid-draw
    dimensions (:328-332) -- because everything is an einsum, sum_n x_n x_n^T
    lowers to ONE tensordot(X^T, X), the "optimised summed statistics" the
    reference's MultivariateNormal leaves as a TODO (core.py:42-43).

Every method takes and returns algebra expressions; `expr.compile()` runs them on
the MI355X backend.
"""
from .. import algebra as A

_DISCRETE_DTYPES = ("int8", "int16", "int32", "int64")


def _sum_trailing(expr, n_trailing):
    """Sum over the last n_trailing axes (the per-datum axes of a statistic)."""
    if n_trailing == 0:
        return expr
    return A.sum(expr, axis=list(range(expr.ndim - n_trailing, expr.ndim)))


class ConditionalDistribution(object):
    """A conditional probability distribution = a parameterised family."""

    @property
    def parameter_types(self):
        """{parameter name: (dtype, ndim)}"""
        raise NotImplementedError

    @property
    def data_type(self):
        """(dtype, ndim) of one datum"""
        raise NotImplementedError

    def is_discrete(self):
        # the reference compares the whole (dtype, ndim) tuple with dtype strings
        # (base.py:23), which is always False; the dtype is what is meant
        return self.data_type[0] in _DISCRETE_DTYPES

    def log_likelihood(self, data, **params):
        """Normalised log-likelihood expression; one value per observation when
        data/params carry extra leading dimensions."""
        return self.log_likelihood_data_term(data) \
            + self.log_likelihood_interaction_term(data, **params) \
            - self.log_normalizer(data_shape=data.shape, **params)

    def log_normalizer(self, data_shape, **params):
        """Terms depending on the parameters (and the data's SHAPE) only."""
        raise NotImplementedError

    def log_likelihood_interaction_term(self, data, **params):
        """Terms depending on both parameters and data."""
        raise NotImplementedError

    def log_likelihood_data_term(self, data):
        """Terms depending on the data only."""
        raise NotImplementedError

    def independent_observations(self, param_copy_ndim=1, iid_draw_ndim=0):
        return IndependentObservations(self, param_copy_ndim, iid_draw_ndim)

    def iid(self, extra_ndim=1):
        return self.independent_observations(param_copy_ndim=0, iid_draw_ndim=extra_ndim)


class IndependentObservations(ConditionalDistribution):
    """Tensor of independent observations: `param_copy_ndim` leading dimensions
    index copies of the parameters, the next `iid_draw_ndim` index iid draws from
    each copy (same number of draws per copy)."""

    def __init__(self, distribution, param_copy_ndim=1, iid_draw_ndim=0):
        self.param_copy_ndim = param_copy_ndim
        self.iid_draw_ndim = iid_draw_ndim
        self.underlying = distribution

    @property
    def parameter_types(self):
        return {name: (dtype, self.param_copy_ndim + ndim)
                for name, (dtype, ndim) in self.underlying.parameter_types.items()}

    @property
    def data_type(self):
        dtype, ndim = self.underlying.data_type
        return dtype, self.param_copy_ndim + self.iid_draw_ndim + ndim

    def _broadcast_params_over_iid_draws(self, params):
        """[copies..., param...] -> [copies..., 1 (x iid_draw_ndim), param...]"""
        out = {}
        for name, (dtype, ndim) in self.underlying.parameter_types.items():
            p = A.wrap_if_literal(params[name])
            axes = list(range(self.param_copy_ndim)) + ["x"] * self.iid_draw_ndim + \
                [self.param_copy_ndim + d for d in range(ndim)]
            out[name] = A.dimshuffle(p, *axes) if self.iid_draw_ndim else p
        return out

    def log_normalizer(self, data_shape, **params):
        lead = self.param_copy_ndim + self.iid_draw_ndim
        per_copy = self.underlying.log_normalizer(data_shape=tuple(data_shape[lead:]), **params)
        per_copy = A.wrap_if_literal(per_copy)
        total = A.sum(per_copy) if self.param_copy_ndim > 0 else per_copy
        if self.iid_draw_ndim > 0:
            draws = A.mul(*data_shape[self.param_copy_ndim:lead])
            return total * draws
        return total

    def log_likelihood_interaction_term(self, data, **params):
        broadcast = self._broadcast_params_over_iid_draws(params)
        return A.sum(self.underlying.log_likelihood_interaction_term(data, **broadcast))

    def log_likelihood_data_term(self, data):
        term = A.wrap_if_literal(self.underlying.log_likelihood_data_term(data))
        return A.sum(term) if term.ndim > 0 else term


class ExponentialFamily(ConditionalDistribution):
    """log p(x | theta) = data_term(x) + sum_j <s_j(x), eta_j(theta)> - A(theta)."""

    def statistic_ndims(self):
        """Per-datum ndim of every sufficient statistic (same as its natural parameter)."""
        raise NotImplementedError

    def log_likelihood_interaction_term(self, data, **params):
        stats = self.sufficient_statistics(data)
        nats = self.natural_parameters(**params)
        terms = [_sum_trailing(A.mul(s, eta), nd)
                 for s, eta, nd in zip(stats, nats, self.statistic_ndims())]
        return terms[0] if len(terms) == 1 else A.add(*terms)

    def sufficient_statistics(self, data):
        raise NotImplementedError

    def natural_parameters(self, **params):
        raise NotImplementedError

    def independent_observations(self, param_copy_ndim=1, iid_draw_ndim=0):
        return ExpFamIndependentObservations(self, param_copy_ndim, iid_draw_ndim)


class ExpFamIndependentObservations(IndependentObservations):
    def statistic_ndims(self):
        return [self.param_copy_ndim + nd for nd in self.underlying.statistic_ndims()]

    def sufficient_statistics(self, data):
        """Statistics of iid draws from the same parameters add up."""
        draws = list(range(self.param_copy_ndim, self.param_copy_ndim + self.iid_draw_ndim))
        stats = self.underlying.sufficient_statistics(data)
        return tuple(A.sum(s, axis=draws) if draws else s for s in stats)

    def natural_parameters(self, **params):
        return self.underlying.natural_parameters(**params)

    def log_likelihood_interaction_term(self, data, **params):
        # <sum_n s(x_n), eta> : the statistics are reduced first (one streaming pass
        # over the data), then dotted with the natural parameters
        stats = self.sufficient_statistics(data)
        nats = self.natural_parameters(**params)
        terms = [A.sum(A.mul(s, eta)) for s, eta in zip(stats, nats)]
        return terms[0] if len(terms) == 1 else A.add(*terms)
