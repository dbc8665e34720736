"""N>1 data-parallel host logic on CPU: two gloo ranks on row shards must
reproduce the single-process update (config 2, 3, 4 and 5 drivers).  See
tests/_dist_worker.py for what is real and what is a test double."""
import os
import subprocess
import sys

import numpy as np

from oracle import svi

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_gloo_ranks_equal_single_process(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    out = str(tmp_path / "rank%d.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    # every rank applied the identical update
    np.testing.assert_array_equal(r0["lam"], r1["lam"])
    np.testing.assert_array_equal(r0["eta"], r1["eta"])
    # ... and it is the single-process update on the whole mini-batch
    X, y, _ = svi.make_cfg2(900, 16)
    lam = svi.blr_init_lam(16)
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    for t in range(1, 4):
        lam, m1, m2, elbo, _ = svi.blr_step(lam, m1, m2, t, X, y, 4, 11, 9000, 0.02)
    np.testing.assert_allclose(r0["lam"], lam, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r0["elbo"][0], elbo, rtol=1e-10)
    # reproducible mode (whole virtual shards per rank, block-sparse all-reduce, ordered add): the
    # host logic; both ranks identical, and equal to the oracle on the first 800 rows
    np.testing.assert_array_equal(r0["lam_rep"], r1["lam_rep"])
    lam = svi.blr_init_lam(16)
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    for t in range(1, 4):
        lam, m1, m2, _, _ = svi.blr_step(lam, m1, m2, t, X[:800], y[:800], 4, 11, 9000, 0.02)
    np.testing.assert_allclose(r0["lam_rep"], lam, rtol=1e-10, atol=1e-12)
    Xm, _, _ = svi.make_cfg3(1200, 4, 3)
    eta0 = svi.mog_prior_eta(3, 4)
    eta = svi.mog_init_eta(Xm[:300], 3, 4, seed=2)
    for t in range(1, 4):
        before = eta
        eta, _, lse = svi.mog_svi_step(eta, eta0, Xm, 12000, (t + 1.0) ** -0.6, 3, 4)
    np.testing.assert_allclose(r0["eta"], eta, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r0["lse"][0], lse, rtol=1e-10)
    # the bound at the parameters the last step started from: local part all-reduced, global part replicated
    np.testing.assert_allclose(r0["mog_elbo"][0], svi.mog_elbo(before, eta0, lse, 10.0, 3, 4), rtol=1e-10)
    np.testing.assert_array_equal(r0["mog_elbo"], r1["mog_elbo"])
    # config 5: the S log-likelihoods are the only exchanged object
    np.testing.assert_array_equal(r0["bbvi_lam"], r1["bbvi_lam"])
    X5, y5, g5, _, _ = svi.make_cfg5(600, 8, 5)
    lam5 = svi.bbvi_init_lam(8 + 5 + 1)
    m1, m2 = np.zeros_like(lam5), np.zeros_like(lam5)
    for t in range(1, 4):
        lam5, m1, m2, elbo5, _, _ = svi.bbvi_step(lam5, m1, m2, t, X5, y5, g5, 8, 5, 16, 5, 6000, 0.05)
    np.testing.assert_allclose(r0["bbvi_lam"], lam5, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(r0["bbvi_elbo"][0], elbo5, rtol=1e-9)
    # config 4: the K x V statistics are all-reduced, every rank holds the same lambda
    np.testing.assert_array_equal(r0["lda_lam"], r1["lda_lam"])
    rs = np.random.RandomState(9)
    C4 = rs.poisson(0.3, (90, 40)).astype(np.float32)
    gamma4 = rs.gamma(100.0, 0.01, (90, 32)).astype(np.float32)
    lam4 = rs.gamma(100.0, 0.01, (32, 40)).astype(np.float32).astype(np.float64)
    for t in range(1, 3):
        before4 = lam4.astype(np.float32)
        lam4, _ = svi.lda_svi_step(lam4.astype(np.float32), gamma4, C4, 0.01, 900, (t + 1.0) ** -0.7)
    np.testing.assert_allclose(r0["lda_lam"], lam4, rtol=2e-6)
    # the words' and the documents' terms are per-rank sums (one all-reduce of two float64), the topics' term is replicated
    np.testing.assert_allclose(r0["lda_elbo"][0], svi.lda_elbo(before4, gamma4, C4, 0.01, 1.0 / 32, 900.0), rtol=1e-6)
    np.testing.assert_array_equal(r0["lda_elbo"], r1["lda_elbo"])
