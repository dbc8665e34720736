"""Distribution / exponential-family node interface (bayesic/distribution/)."""
from .base import (ConditionalDistribution, ExpFamIndependentObservations, ExponentialFamily,
                   IndependentObservations)
from .core import MultivariateNormal, Normal, logdet
from .families import (Bernoulli, Categorical, Dirichlet, Gamma, InverseGamma, Multinomial,
                       Wishart)
from .special import digamma, gammaln

__all__ = ["ConditionalDistribution", "IndependentObservations", "ExponentialFamily",
           "ExpFamIndependentObservations", "Normal", "MultivariateNormal", "logdet",
           "Gamma", "InverseGamma", "Bernoulli", "Categorical", "Multinomial", "Dirichlet",
           "Wishart", "gammaln", "digamma"]
