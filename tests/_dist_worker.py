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

    def bsc_mog_expected_params(self, eta, K, D, Wmat, c):
        w, cc = svi.mog_expected_params(eta.numpy(), K, D)
        Wmat.copy_(torch.from_numpy(w))
        c.copy_(torch.from_numpy(cc))

    def bsc_mog_estep(self, X, ldx, N, D, K, Wmat, c, stats, lse):
        s, l = svi.mog_local_step(X.numpy(), Wmat.numpy(), c.numpy())
        stats.copy_(torch.from_numpy(s.ravel()))
        lse[0] = l

    def bsc_mog_natgrad(self, eta, eta0, stats, K, D, scale, rho):
        new = svi.natgrad_update(eta.numpy(), eta0.numpy(),
                                 svi.mog_message(stats.numpy().reshape(K, 1 + 2 * D), K, D), scale, rho)
        eta.copy_(torch.from_numpy(new))


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

    # --- config 3 shape (small) ---------------------------------------------------
    Xm, _, _ = svi.make_cfg3(1200, 4, 3)
    mc = [0, 700, 1200] if world == 2 else np.linspace(0, 1200, world + 1).astype(int)
    eta0 = svi.mog_prior_eta(3, 4)
    eta_init = svi.mog_init_eta(Xm[:300], 3, 4, seed=2)
    mog = MoGNatGradSVI(torch.from_numpy(Xm[mc[rank]:mc[rank + 1]].copy()), 3, eta0, eta_init,
                        n_total=12000, ctx=OracleContext())
    for _ in range(3):
        mog.step()
    np.savez(out_path % rank, lam=lam, elbo=model.elbo.numpy(), eta=mog.eta.numpy(),
             lse=mog.lse.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
