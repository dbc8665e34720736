"""Distribution / exponential-family node interface (bayesic/distribution/)."""
from .base import (ConditionalDistribution, ExpFamIndependentObservations, ExponentialFamily,
                   IndependentObservations)
from .core import MultivariateNormal, Normal, logdet

__all__ = ["ConditionalDistribution", "IndependentObservations", "ExponentialFamily",
           "ExpFamIndependentObservations", "Normal", "MultivariateNormal", "logdet"]
