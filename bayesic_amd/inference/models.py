"""BASELINE configurations written on the plugin surface -- Distribution nodes
(bayesic/distribution/base.py:9-172) and ``bayesic.algebra`` expressions -- for the general engines
(``ReparamVI``, ``ScoreFunctionVI``, ``MeanFieldVMP``).  Nothing here names a kernel: which launches an
update turns into is the engines' business (``inference/recognise.py``).
"""
from .. import algebra as A
from ..distribution import InverseGamma, Normal


def linear_regression_log_joint(n_total_over_batch=1.0, alpha0=1.0, beta0=1.0, dtype="float32"):
    """Config 2: y_n ~ N(x_n . w, s2), w | s2 ~ N(0, s2 I), s2 ~ InvGamma(alpha0, beta0),
    with a leading Monte-Carlo sample axis on the latents ``W`` [S, D] and ``xi`` = log s2 [S, 1]:

        log p(y, w, xi) = scale * sum_n log N(y_n | x_n . w, e^xi) + sum_d log N(w_d | 0, e^xi)
                          + log InvGamma(e^xi | alpha0, beta0) + xi          (the Jacobian of s2 = e^xi)

    each density through the node's three-term decomposition (data term + interaction term - log-normaliser,
    bayesic/distribution/base.py:47-69).  Returns (log-joint [S], [(W, D) ...] is the caller's: the latent
    vars ``W`` and ``xi`` and the data vars ``X`` [N, D], ``y`` [N])."""
    X, y = A.var("X", 2, dtype), A.var("y", 1, dtype)
    W, xi = A.var("W", 2, dtype), A.var("xi", 2, dtype)
    x = A.sum(xi, axis=1)                                   # [S]
    s2 = A.exp(x)
    s2_rows = A.dimshuffle(s2, 0, "x")                      # [S, 1]: one variance per draw, broadcast along the data
    likelihood = A.sum(Normal().log_likelihood(A.dimshuffle(y, "x", 0), mean=A.dot(W, X.T), variance=s2_rows),
                       axis=1) * float(n_total_over_batch)
    prior_w = A.sum(Normal().log_likelihood(W, mean=0.0, variance=s2_rows), axis=1)
    prior_xi = InverseGamma().log_likelihood(s2, shape=float(alpha0), scale=float(beta0)) + x
    return likelihood + prior_w + prior_xi, dict(X=X, y=y, W=W, xi=xi)
