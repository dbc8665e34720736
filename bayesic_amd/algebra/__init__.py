"""bayesic_amd.algebra -- symbolic tensor algebra with einsum rewriting, the
plugin surface of the MI355X backend.

Same public names and behaviour as the reference module ``bayesic.algebra``
(``from bayesic.algebra import *``, bayesic/tests/test_algebra.py:1-4), written
from scratch and with the numeric side behind an explicit backend interface
(bayesic_amd/algebra/backend.py) instead of inline Theano calls.  Like the
reference module it has no ``__all__``-style filtering of helper imports: ``np``,
``it``, ``Counter`` and ``defaultdict`` are re-exported because the reference's
own tests rely on the star import providing ``Counter``.
"""
import itertools as it
from collections import Counter, defaultdict

import numpy as np

from .einsum_form import Einsum, einsum
from .expr import (Expression, add, autobroadcast_or_match, constant, elemwise, eye, shape, var,
                   with_wrapped_literals, wrap_if_literal)
from .matching import match
from .multiset import (equivalence_classes, find_bijection, find_bijections, find_duplicate,
                       find_injection, find_injections, submultisets_of_size)
from .ops import (_diagonal, _dimshuffle, _mul, _sum, _tensordot, abs_, diagonal, dimshuffle, div,
                  dot, exp, log, mul, neg, outer, pow, sub, sum, tensordot, trace, transpose)

__all__ = [
    "np", "it", "Counter", "defaultdict",
    "Expression", "var", "constant", "shape", "wrap_if_literal", "with_wrapped_literals",
    "autobroadcast_or_match", "elemwise", "add", "eye", "find_duplicate", "equivalence_classes",
    "einsum", "Einsum", "match", "submultisets_of_size", "find_bijection", "find_bijections",
    "find_injection", "find_injections", "dot", "tensordot", "mul", "outer", "sum", "trace",
    "diagonal", "transpose", "dimshuffle", "div", "neg", "sub", "log", "exp", "pow", "abs_",
]
