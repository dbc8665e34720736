"""CPU oracle for the Bayesic SVI hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker -- never as the thing measured or
shipped.  ``bayesic_amd`` must never import from here.

Pinning status (see DESIGN.md "Oracle"):

* einsum semantics / lowered five-op tree (``oracle.einsum_eval``): PINNED by
  the reference's own numeric tests, which use numpy expressions as their
  oracle (``bayesic/tests/test_algebra.py:44-191``), and by the symbolic golden
  fixtures in ``tests/golden/`` generated from the reference's pure-Python
  front end.
* exponential-family nodes: the node classes (``bayesic_amd.distribution``) build
  algebra expressions, which the tests evaluate in float64 with
  ``oracle.einsum_eval.NumpyBackend`` -- the reference's formulas
  (``bayesic/distribution/core.py:16-20,41-47``) with corrected log-normalisers,
  pinned by ``scipy.stats`` known answers (the reference files do not import).
* ELBO / reparameterisation sampler / natural-gradient update / BBVI
  (``oracle.svi``, ``oracle.philox``): **PARITY UNPINNED** -- the reference has
  no implementation of this path (``README.md:24-80`` is prose only).  These are
  float64 restatements of the published algorithms, validated against exact
  conjugate posteriors, finite differences and the Random123 Philox4x32-10
  known-answer vectors.
* ``oracle/c/oracle_kernels.c`` (loaded by ``oracle.cbuild``): the data-sized sums
  of configs 2, 3, 4 and 5 once more in plain C + OpenMP, written independently of
  ``oracle.svi`` and checked against it; **PARITY UNPINNED** for the same reason.
  It exists so the HIP kernels can be compared on identical data at full size.
"""
