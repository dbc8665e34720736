"""Conjugacy detection and mean-field / VMP update synthesis from a symbolic
log-joint (the purpose bayesic/algebra.py:1-6 and
README.md:30-37 state for the algebra front end)."""
from .conjugacy import (NotConjugate, conjugate_coefficients, depends_on, expand_terms)
from .bbvi import ScoreFunctionVI
from .reparam import ReparamVI
from .vmp import (CategoricalNode, DirichletNode, GammaNode, InverseGammaNode, MeanFieldVMP,
                  MVNormalNode, NormalGammaNode, NormalNode, WishartNode, ResidentDirichletNode,
                  ResidentGammaNode, ResidentInverseGammaNode, ResidentMVNormalNode, ResidentNormalGammaNode,
                  ResidentNormalNode, ResidentWishartNode)

__all__ = ["NotConjugate", "conjugate_coefficients", "depends_on", "expand_terms",
           "MeanFieldVMP", "NormalNode", "GammaNode", "DirichletNode", "CategoricalNode", "MVNormalNode", "InverseGammaNode", "WishartNode", "NormalGammaNode",
           "ResidentDirichletNode", "ResidentGammaNode", "ResidentInverseGammaNode", "ResidentMVNormalNode",
           "ResidentNormalGammaNode", "ResidentNormalNode", "ResidentWishartNode",
           "ScoreFunctionVI", "ReparamVI"]
