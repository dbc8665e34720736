"""Worker for tests/test_distributed_cpu.py: one rank of a gloo job (CPU).

The product path needs a GPU, so here the *context* is a test double whose entry
points are computed by the oracle on the rank's CPU tensors.  What is under test
is the HOST logic of the data-parallel drivers: world-size detection, the global
row count and mini-batch scale, the single all-reduce between pass and finish,
rank-independent noise, the double-buffer flip -- i.e. that N ranks on row shards
reproduce the single-process update.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import svi  # noqa: E402


class OracleContext:
    """Duck-types bayesic_amd.device.Context on the CPU (tests only)."""
    device = torch.device("cpu")

    def reserve(self, nbytes):
        pass

    def info(self):
        return {"cu_count": 256}

    def sync(self):
        pass

    def call(self, name, *a):
        getattr(self, name)(*a)

    @staticmethod
    def _np(t):
        return t.numpy()

    def bsc_blr_noise(self, D, S, seed, step0, n_steps, eps):
        for k in range(n_steps):
            e, _, _ = svi.blr_sample(np.zeros(2 * D + 2), D, S, seed, step=step0 + k)
            eps[k].copy_(torch.from_numpy(e.ravel()))

    def bsc_blr_sample(self, lam, D, S, seed, step, eps, W, xi):
        e, w, x = svi.blr_sample(lam.numpy(), D, S, seed, step=step)
        eps.copy_(torch.from_numpy(e.ravel()))
        W.copy_(torch.from_numpy(w.ravel()))
        xi.copy_(torch.from_numpy(x))

    def bsc_blr_data_pass(self, X, ldx, y, B, D, W, S, Q, G):
        q, g = svi.blr_data_pass(X.numpy(), y.numpy(), W.numpy().reshape(S, D))
        Q.copy_(torch.from_numpy(q))
        G.copy_(torch.from_numpy(g.ravel()))

    def bsc_blr_data_pass_partial(self, X, ldx, y, B, D, W, S):
        self._pending = svi.blr_data_pass(X.numpy(), y.numpy(), W.numpy().reshape(S, D))

    def bsc_blr_data_pass_sweep(self, X, ldx, y, B, D, W, S, Q, G, sweep):
        assert sweep in (0, 1, 2)       # the order of the rows does not change the sums
        self.bsc_blr_data_pass(X, ldx, y, B, D, W, S, Q, G)

    def bsc_blr_data_pass_partial_sweep(self, X, ldx, y, B, D, W, S, sweep):
        assert sweep in (0, 1, 2)
        self.bsc_blr_data_pass_partial(X, ldx, y, B, D, W, S)

    def bsc_blr_fused_update(self, stats, lam_in, lam_out, m1, m2, eps, W, xi, D, S, batch_rows,
                             scale, alpha0, beta0, t, lr, b1, b2, adam_eps, seed, next_step,
                             eps_n, eps_ready, W_n, xi_n, elbo, grad):
        if stats is None:
            Q, G = self._pending
        else:
            s = stats.numpy()
            Q, G = s[:S], s[S:].reshape(S, D)
        e, g = svi.blr_elbo_and_grad(lam_in.numpy(), eps.numpy().reshape(S, D + 1),
                                     W.numpy().reshape(S, D), xi.numpy(), Q, G, batch_rows, scale,
                                     alpha0, beta0)
        new, a, b = svi.adam_ascent(lam_in.numpy(), g, m1.numpy(), m2.numpy(), t, lr, b1, b2, adam_eps)
        lam_out.copy_(torch.from_numpy(new))
        m1.copy_(torch.from_numpy(a))
        m2.copy_(torch.from_numpy(b))
        elbo[0] = e
        grad.copy_(torch.from_numpy(g))
        self.bsc_blr_sample(lam_out, D, S, seed, next_step, eps_n, W_n, xi_n)

    def bsc_memset(self, t, value, nbytes):
        t.view(-1).view(torch.uint8)[:nbytes].fill_(value)

    def bsc_elemwise(self, op, dtype, rank, shape, out, out_strides, n_in, ptrs, strides):
        """The n-ary float64 add of the reproducible mode: out = ((in0 + in1) + in2) + ..."""
        import ctypes
        assert op == 0 and dtype == 1 and rank == 1
        n = int(shape[0])
        acc = None
        for k in range(n_in):
            a = np.ctypeslib.as_array((ctypes.c_double * n).from_address(int(ptrs[k])))
            acc = a.copy() if acc is None else acc + a
        out.copy_(torch.from_numpy(acc))

    def bsc_mog_expected_params(self, eta, K, D, Wmat, c):
        w, cc = svi.mog_expected_params(eta.numpy(), K, D)
        Wmat.copy_(torch.from_numpy(w))
        c.copy_(torch.from_numpy(cc))

    def bsc_mog_log_normalizer(self, eta, K, D, A_out):
        alpha, m, kappa, a, b = svi.mog_unpack(eta.numpy(), K, D)
        A_out[0] = float(svi.dirichlet_log_normalizer(alpha) + svi.normal_gamma_log_normalizer(kappa, a, b).sum())

    def bsc_mog_expected_params_bound(self, eta, eta0, prior_A, K, D, Wmat, c, bound):
        self.bsc_mog_expected_params(eta, K, D, Wmat, c)
        bound[0] = svi.mog_global_bound(eta.numpy(), eta0.numpy(), K, D)

    def bsc_mog_natgrad_elbo(self, eta, eta0, stats, K, D, scale, rho, lse, bound, elbo):
        elbo[0] = scale * float(lse[0]) + float(bound[0])
        self.bsc_mog_natgrad(eta, eta0, stats, K, D, scale, rho)

    def bsc_mog_estep(self, X, ldx, N, D, K, Wmat, c, stats, lse):
        s, l = svi.mog_local_step(X.numpy(), Wmat.numpy(), c.numpy())
        stats.copy_(torch.from_numpy(s.ravel()))
        lse[0] = l

    def bsc_mog_natgrad(self, eta, eta0, stats, K, D, scale, rho):
        new = svi.natgrad_update(eta.numpy(), eta0.numpy(),
                                 svi.mog_message(stats.numpy().reshape(K, 1 + 2 * D), K, D), scale, rho)
        eta.copy_(torch.from_numpy(new))


    # -- config 5 (BBVI) ---------------------------------------------------------
    def bsc_bbvi_sample(self, lam, D, G, S, seed, step, eps, Wz, Bz, zeta):
        P = D + G + 1
        e, z = svi.bbvi_sample(lam.numpy(), P, S, seed, step=step)
        eps.copy_(torch.from_numpy(e.ravel()))
        Wz.copy_(torch.from_numpy(z[:, :D].astype(np.float32).ravel()))
        Bz.copy_(torch.from_numpy(np.ascontiguousarray(z[:, D:D + G].T).astype(np.float32).ravel()))
        zeta.copy_(torch.from_numpy(z[:, D + G].copy()))

    def bsc_logreg_bbvi_loglik(self, X, ldx, y, g, N, D, G, Wz, Bz, S, ell):
        out = svi.logreg_loglik(X.numpy(), y.numpy(), g.numpy(), Wz.numpy().reshape(S, D),
                                Bz.numpy().reshape(G, S))
        ell.copy_(torch.from_numpy(out))

    def bsc_bbvi_grad(self, lam, eps, ell, D, G, S, scale, a0, b0, elbo, grad, f_out):
        P = D + G + 1
        e, g, _, f = svi.bbvi_elbo_and_grad(lam.numpy(), eps.numpy().reshape(S, P), ell.numpy(),
                                            D, G, scale, a0, b0)
        elbo[0] = float(e)
        grad.copy_(torch.from_numpy(g))
        f_out.copy_(torch.from_numpy(f))

    def bsc_bbvi_update(self, lam, eps, ell, D, G, S, scale, a0, b0, m1, m2, t, lr, b1, b2, eps_adam, seed,
                        next_step, Wz, Bz, zeta, elbo, grad, f_out):
        self.bsc_bbvi_grad(lam, eps, ell, D, G, S, scale, a0, b0, elbo, grad, f_out)
        self.bsc_adam_ascent(lam, grad, m1, m2, lam.numel(), t, lr, b1, b2, eps_adam)
        self.bsc_bbvi_sample(lam, D, G, S, seed, next_step, eps, Wz, Bz, zeta)

    def bsc_adam_ascent(self, lam, grad, m1, m2, n, t, lr, b1, b2, eps):
        new, a, b = svi.adam_ascent(lam.numpy(), grad.numpy(), m1.numpy(), m2.numpy(), t, lr, b1, b2, eps)
        lam.copy_(torch.from_numpy(new))
        m1.copy_(torch.from_numpy(a))
        m2.copy_(torch.from_numpy(b))

    # -- config 4 (LDA) ------------------------------------------------------------
    def bsc_dirichlet_expectation(self, lam, rows, cols, ld, out):
        out.copy_(torch.from_numpy(svi.dirichlet_expectation(lam.numpy()).astype(np.float32)))

    def bsc_lda_sstats(self, C, ldc, docs, V, K, Th, ldth, Bt, ldb, out, ldo):
        # (C, Bt: views of V columns; out: [K, V] with leading dimension ldo -- a block of the staging buffer when the
        # driver takes the statistic in column ranges)
        res = svi.lda_sstats(C.numpy()[:, :V], Th.numpy(), Bt.numpy()[:, :V]).astype(np.float32)
        out.view(-1)[:K * ldo].view(K, ldo)[:, :V].copy_(torch.from_numpy(res))

    def bsc_lda_sstats_round_columns(self, K, out):
        out._obj.value = 16            # (a "round" of 16 columns: 40 columns are taken in three ranges)

    def bsc_natgrad_update_f32_2d(self, eta, ld_eta, eta0, message, ld_msg, rows, cols, scale, rho, ll, n_ll,
                                  local_bound, global_bound, elbo):
        if elbo is not None:
            words = 0.0
            for i in range(n_ll):
                words += float(ll[i])
            elbo[0] = scale * (words + float(local_bound[0])) + float(global_bound[0])
        e = eta[:, :cols]
        m = message.view(-1)[:rows * ld_msg].view(rows, ld_msg)[:, :cols]
        new = (1.0 - rho) * e.numpy().astype(np.float64) + rho * (eta0 + scale * m.numpy().astype(np.float64))
        e.copy_(torch.from_numpy(new.astype(np.float32)))

    def bsc_dirichlet_expectation_bound(self, lam, rows, cols, ld, prior, out, bound):
        self.bsc_dirichlet_expectation(lam, rows, cols, ld, out)
        bound[0] = float(svi.dirichlet_neg_kl(lam.numpy(), prior).sum())

    def bsc_lda_sstats_bound(self, C, ldc, docs, V, K, Th, ldth, Bt, ldb, out, ldo, ll):
        self.bsc_lda_sstats(C, ldc, docs, V, K, Th, ldth, Bt, ldb, out, ldo)
        ll[0] = svi.lda_local_bound(C.numpy()[:, :V], Th.numpy(), Bt.numpy()[:, :V])

    def bsc_natgrad_update_f32_elbo(self, eta, eta0, message, n, scale, rho, ll, local_bound, global_bound, elbo):
        elbo[0] = scale * (float(ll[0]) + float(local_bound[0])) + float(global_bound[0])
        self.bsc_natgrad_update_f32(eta, eta0, message, n, scale, rho)

    def bsc_natgrad_update_f32(self, eta, eta0, message, n, scale, rho):
        new = (1.0 - rho) * eta.numpy().astype(np.float64) + \
            rho * (eta0 + scale * message.numpy().astype(np.float64))
        eta.copy_(torch.from_numpy(new.astype(np.float32)))


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesic_amd.svi.blr import BLRReparamSVI
    from bayesic_amd.svi.mog import MoGNatGradSVI

    # --- config 2 shape (small): unequal shards on purpose ---------------------
    X, y, _ = svi.make_cfg2(900, 16)
    cuts = [0, 500, 900] if world == 2 else np.linspace(0, 900, world + 1).astype(int)
    Xs = torch.from_numpy(X[cuts[rank]:cuts[rank + 1]].copy())
    ys = torch.from_numpy(y[cuts[rank]:cuts[rank + 1]].copy())
    model = BLRReparamSVI(Xs, ys, n_total=9000, n_samples=4, seed=11, lr=0.02, ctx=OracleContext())
    assert model.world == world and model.batch_rows == 900.0
    for _ in range(3):
        model.step()
    lam = model.lam.numpy().copy()

    # --- reproducible mode: whole virtual shards per rank (8 shards of 100 rows: 5 + 3) --------
    rcut = [0, 500, 800] if world == 2 else np.linspace(0, 800, world + 1).astype(int)
    rep = BLRReparamSVI(torch.from_numpy(X[rcut[rank]:rcut[rank + 1]].copy()),
                        torch.from_numpy(y[rcut[rank]:rcut[rank + 1]].copy()), n_total=9000, n_samples=4,
                        seed=11, lr=0.02, ctx=OracleContext(), reproducible=True)
    assert rep.batch_rows == 800.0 and rep._shard_rows == 100
    assert rep._first_shard == rcut[rank] // 100 and rep._n_shards == (rcut[rank + 1] - rcut[rank]) // 100
    for _ in range(3):
        rep.step()
    lam_rep = rep.lam.numpy().copy()

    # --- config 3 shape (small) ---------------------------------------------------
    Xm, _, _ = svi.make_cfg3(1200, 4, 3)
    mc = [0, 700, 1200] if world == 2 else np.linspace(0, 1200, world + 1).astype(int)
    eta0 = svi.mog_prior_eta(3, 4)
    eta_init = svi.mog_init_eta(Xm[:300], 3, 4, seed=2)
    mog = MoGNatGradSVI(torch.from_numpy(Xm[mc[rank]:mc[rank + 1]].copy()), 3, eta0, eta_init,
                        n_total=12000, ctx=OracleContext())
    for _ in range(3):
        mog.step()

    # --- config 5 shape (small): one all-reduce of the S log-likelihoods ------------
    from bayesic_amd.svi.bbvi import LogRegBBVI
    from bayesic_amd.svi.lda import LDAFixedGammaSVI
    X5, y5, g5, _, _ = svi.make_cfg5(600, 8, 5)
    c5 = [0, 350, 600] if world == 2 else np.linspace(0, 600, world + 1).astype(int)
    sl = slice(c5[rank], c5[rank + 1])
    bb = LogRegBBVI(torch.from_numpy(X5[sl].copy()), torch.from_numpy(y5[sl].copy()),
                    torch.from_numpy(g5[sl].copy()), 5, n_total=6000, n_samples=16, seed=5, lr=0.05,
                    ctx=OracleContext())
    assert bb.world == world and bb.batch_rows == 600.0
    for _ in range(3):
        bb.step()

    # --- config 4 shape (small): one all-reduce of the K x V statistics --------------
    rs = np.random.RandomState(9)
    C4 = rs.poisson(0.3, (90, 40)).astype(np.float32)
    gamma4 = rs.gamma(100.0, 0.01, (90, 32)).astype(np.float32)
    lam4 = rs.gamma(100.0, 0.01, (32, 40)).astype(np.float32)
    c4 = [0, 50, 90] if world == 2 else np.linspace(0, 90, world + 1).astype(int)
    sl = slice(c4[rank], c4[rank + 1])
    lda = LDAFixedGammaSVI(torch.from_numpy(C4[sl].copy()), torch.from_numpy(gamma4[sl].copy()),
                           torch.from_numpy(lam4.copy()), eta=0.01, docs_total=900, ctx=OracleContext(),
                           via="kernel")
    assert lda.world == world and lda.batch_docs == 90.0
    # the same model with ONE collective after the whole statistic (overlap=False): the column ranges of the default
    # route (three of them here, each all-reduced on its own while the next is computed) must give the same bits
    mono = LDAFixedGammaSVI(torch.from_numpy(C4[sl].copy()), torch.from_numpy(gamma4[sl].copy()),
                            torch.from_numpy(lam4.copy()), eta=0.01, docs_total=900, ctx=OracleContext(),
                            via="kernel", overlap=False)
    for _ in range(2):
        lda.step()
        mono.step()
    assert lda._pieces == [(0, 16), (16, 16), (32, 8)] and mono._pieces is None
    assert torch.equal(lda.lam, mono.lam), "overlapped pieces differ from the one-collective update"
    np.testing.assert_allclose(lda.elbo.numpy(), mono.elbo.numpy(), rtol=1e-13)
    np.savez(out_path % rank, lam=lam, lam_rep=lam_rep, elbo=model.elbo.numpy(), eta=mog.eta.numpy(),
             lse=mog.lse.numpy(), bbvi_lam=bb.lam.numpy(), bbvi_elbo=bb.elbo.numpy(),
             lda_lam=lda.lam.numpy(), mog_elbo=mog.elbo.numpy(), lda_elbo=lda.elbo.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
