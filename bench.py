"""ELBO-gradient updates/sec on a 1M-row mini-batch (BASELINE.json's metric), MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling both|weak|strong] [--config cfg2]

Default workload = BASELINE config 2 (the configuration the metric is quoted on): Bayesian
linear regression, 1M x 256 float32 mini-batch, reparameterisation-trick ELBO, S = 8, Adam.
One "step" = one full update on the HBM-resident mini-batch: Philox draw -> fused data pass
over X, y -> float64 reduction -> (all-reduce over RCCL when N > 1, through the C ABI on the
context's stream) -> ELBO + pathwise gradient -> Adam step.  --config cfg3 | cfg4 | cfg5 run
the other BASELINE configurations' updates through the same harness and schema.

N > 1: one process per GPU.  Started by the driver's torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) -- or bare, `python bench.py --gpus N`: then THIS
process never touches the GPU, starts N rank processes itself, relays rank 0's JSON line and
exits non-zero if any rank failed.  torch.distributed (gloo) is only the host channel (unique
id, barrier, max over ranks); the data path's one collective is bsc_allreduce_sum (RCCL/xGMI).

--scaling strong: ONE global mini-batch of the configured size (the metric's "1M-row mini-batch at
1/2/4/8 GPUs") is split into N row blocks.  --scaling weak: every rank holds its own full-size
mini-batch, `value` counts 1M-row mini-batch equivalents per second over all ranks.  Default
(`both`): at N > 1 the line carries BOTH -- `value` / `scaling` = the metric's reading (strong) and
`value_weak` beside it; at N = 1 the two coincide.

What `value` is on config 2 (the metric's configuration): the update loop ROTATES over --batches
(default 3) distinct mini-batches resident in HBM, a different one every step, as an SVI loop over
fresh subsamples does (README.md:69-79) -- nothing of a step's 1.03 GB is left in the 256 MiB
Infinity Cache for the next.  The same loop over ONE resident mini-batch (alternating sweeps, each
pass starting in the rows the previous one left on-die) is timed beside it: `value_same_batch`,
`roofline.same_batch`.

The timed region is EXACTLY K steps between barrier + synchronize; it is repeated (same K) until
--min-timed-s of it has been measured, `ms_per_step` / `value` are the MEDIAN block's and
`timed_blocks` gives min / median / max.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36
F32_MFMA_PEAK_TF = 157.3  # fp32-input MFMA = fp32 vector peak, same guide :41-42
BF16_MFMA_PEAK_TF = 2500.0   # dense bf16 MFMA, same guide :43 (the operand-split route, --mfma-split)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"])
    ap.add_argument("--via", default="driver", choices=["driver", "plugin"],
                    help="cfg2: driver = the hand-written host driver (svi/blr.py); plugin = the SAME model written "
                         "with Normal / InverseGamma nodes and bayesic.algebra expressions (inference/models.py) "
                         "and stepped by the general engine inference.ReparamVI, which has to recognise the data "
                         "term itself (inference/recognise.py) to reach the fused kernels")
    ap.add_argument("--batches", type=int, default=3,
                    help="cfg2: distinct resident mini-batches the update loop rotates over (1 = the same "
                         "mini-batch every step)")
    ap.add_argument("--min-timed-s", type=float, default=0.25,
                    help="repeat the timed block of K steps until this much of it has been measured")
    ap.add_argument("--max-blocks", type=int, default=100)
    ap.add_argument("--rows", type=int, default=None,
                    help="mini-batch rows (cfg4: documents): per GPU when weak, global when strong; "
                         "default = the BASELINE size of the config")
    ap.add_argument("--dim", type=int, default=256, help="cfg2 / cfg5 columns")
    ap.add_argument("--samples", type=int, default=None, help="Monte-Carlo draws (cfg2: 8, cfg5: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="cfg4 with a communicator: ONE all-reduce after the whole statistic instead of one per finished "
                         "column range on a second stream (svi/lda.py overlap=False)")
    ap.add_argument("--no-via-plugin", action="store_true",
                    help="cfg2: skip the second timed loop through the plugin surface (value_via_plugin)")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0,
                    help="CPU seconds per leg of the cpu_baseline (there are up to three legs)")
    ap.add_argument("--mfma-split", type=int, default=0, choices=[0, 2, 3],
                    help="MFMA-bound contractions on the bf16 MFMA with every f32 operand a sum of 2 / 3 bf16 terms "
                         "(3 / 6 products; bsc_ctx_set_mfma_split).  0 = f32 MFMA, the default and the dtype of the "
                         "reported figures; honoured by cfg3 (forward always on three terms), cfg4 (2, 3) and cfg5 (2)")
    ap.add_argument("--unfused", action="store_true",
                    help="cfg2, N=1 through the multi-GPU code path (float64 statistics between the "
                         "pass and the finish, no collective): what the N>1 step costs besides RCCL")
    ap.add_argument("--reproducible", action="store_true",
                    help="cfg2: the bit-reproducible mode (one pass per virtual shard, block-sparse "
                         "all-reduce, ordered add): identical results on 1, 2, 4 and 8 GPUs")
    ap.add_argument("--sweep", default="alternate", choices=["alternate", "stream"],
                    help="cfg2: alternate = successive passes over the resident mini-batch walk it "
                         "forward / backward so each starts in the rows the Infinity Cache still holds "
                         "(the driver's default); stream = always forward, non-temporal")
    ap.add_argument("--rccl-world1", action="store_true",
                    help="N=1 with a one-rank RCCL communicator: the collective is really enqueued")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="no hipEvent pairs inside the timed region and no timed burst after it")
    ap.add_argument("--spin-up-ms", type=float, default=60.0,
                    help="untimed launches of the dominant kernel before the W warm-up steps, until "
                         "about this much GPU work has been queued: a process that has just started "
                         "using the GPU runs its first ~35 ms of kernels 10-40 %% slow "
                         "(tools/ramp_probe.py); a training job lives in the steady state")
    ap.add_argument("--time-every", type=int, default=8,
                    help="inside the timed region, time every n-th launch per slot (an event pair "
                         "costs ~4 us of stream time per use)")
    ap.add_argument("--burst", type=int, default=64,
                    help="dominant-kernel launches timed one by one AFTER the timed region; "
                         "roofline.avg_launch_us comes from these")
    ap.add_argument("--synthetic", default="survey", choices=["survey", "device"],
                    help="survey: the SURVEY 8(d) numpy RandomState inputs (host-generated, seconds); "
                         "device: torch.randn on the GPU")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------
# bare `--gpus N`: start the ranks.  Nothing in this function may touch the GPU or import torch.
# ------------------------------------------------------------------------------------------
def spawn_ranks(args):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", BSC_BENCH_SPAWNED="1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr,
                                      stderr=sys.stderr))
    deadline = time.time() + float(os.environ.get("BSC_BENCH_TIMEOUT_S", "1500"))
    failed = None
    pending = set(range(args.gpus))
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0:
                    failed = (r, rc)
        if time.time() > deadline:
            failed = (-1, 124)
        if pending and failed is None:
            time.sleep(0.2)
    if failed is not None:
        for r in pending:                 # exact PIDs of our own children, nothing else
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    out = procs[0].stdout.read().decode() if procs[0].stdout else ""
    sys.stdout.write(out)
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with status %d\n" % failed)
        return failed[1] if failed[1] > 0 else 1
    return 0


# ------------------------------------------------------------------------------------------
# synthetic inputs; rank r of a weak-scaling run draws from seed + r
# ------------------------------------------------------------------------------------------
def _normal_f32(np, seed, shape, chunk_rows=125_000):
    """RandomState(seed).standard_normal(shape) as float32, drawn in row chunks (the stream of
    one RandomState is the same whether it is drawn at once or in pieces)."""
    rs = np.random.RandomState(seed)
    if len(shape) == 1:
        return rs.standard_normal(shape[0]).astype(np.float32)
    out = np.empty(shape, np.float32)
    for r0 in range(0, shape[0], chunk_rows):
        r1 = min(shape[0], r0 + chunk_rows)
        out[r0:r1] = rs.standard_normal((r1 - r0,) + tuple(shape[1:]))
    return out


def rank_rows(total, rank, world):
    """Contiguous row block of `rank`: [rank*total/world, (rank+1)*total/world)."""
    return (total * rank) // world, (total * (rank + 1)) // world


class Workload:
    """One BASELINE configuration behind the harness."""
    name = ""
    dtype = "f32"

    def spin(self):            # one untimed launch of the dominant kernel (state untouched)
        raise NotImplementedError

    def step(self):
        raise NotImplementedError


class Cfg2(Workload):
    """Bayesian linear regression, reparameterisation-trick ELBO (the metric's config)."""
    spin_is_one_kernel = True       # spin() = bsc_blr_data_pass_partial_sweep = ONE launch of the pass kernel
    name = "cfg2"
    default_rows, default_samples = 1_000_000, 8
    kernel_ms = 0.17

    def __init__(self, args, ctx, torch, rank, world):
        import numpy as np
        from bayesic_amd.svi.blr import BLRReparamSVI
        self.ctx, self.args = ctx, args
        D, S = args.dim, args.samples or self.default_samples
        total = args.rows or self.default_rows
        if args.scaling == "strong":
            r0, r1 = rank_rows(total, rank, world)
            global_rows, seed_off = total, 0
        else:
            r0, r1, global_rows, seed_off = 0, total, total * world, rank
        rows = r1 - r0
        dev = ctx.device
        if args.synthetic == "survey":
            # X = RandomState(1234).standard_normal((N, D)), w* = RandomState(1)/16,
            # y = X w* + 0.5 RandomState(2) -- the rank keeps its row block
            Xh = _normal_f32(np, 1234 + seed_off, (r1, D))[r0:]
            w_true = np.random.RandomState(1).standard_normal(D) / 16.0
            noise = np.random.RandomState(2 + 7 * seed_off).standard_normal(r1)[r0:]
            yh = (Xh.astype(np.float64) @ w_true + 0.5 * noise).astype(np.float32)
            X, y = torch.from_numpy(Xh).to(dev), torch.from_numpy(yh).to(dev)
            self.host = (Xh, yh)
        else:
            g = torch.Generator(device=dev).manual_seed(1234 + seed_off)
            X = torch.randn((rows, D), generator=g, device=dev, dtype=torch.float32)
            w_true = torch.randn(D, generator=torch.Generator(device=dev).manual_seed(1), device=dev) / 16
            y = X @ w_true + 0.5 * torch.randn(rows, generator=g, device=dev)
            self.host = None
        self.X, self.y, self.rows, self.D, self.S = X, y, rows, D, S
        # further resident mini-batches of the same shape (drawn on the device: their values are
        # not compared with anything); the loop visits batch t % n at step t
        # (one allocation, so that the read-ceiling probe can stream through all of them: nb x 1.03 GB
        # against a 256 MiB Infinity Cache -- a ceiling for FRESH data, like the rotation itself)
        nb = max(1, args.batches)
        self.X_all = torch.empty((nb * rows, D), dtype=torch.float32, device=dev)
        self.X_all[:rows].copy_(X)
        X = self.X = self.X_all[:rows]
        self.batches = [(X, y)]
        for b in range(1, nb):
            gb = torch.Generator(device=dev).manual_seed(99_991 * b + 1234 + seed_off)
            Xb = self.X_all[b * rows:(b + 1) * rows]
            Xb.normal_(generator=gb)
            wb = torch.from_numpy((np.random.RandomState(1).standard_normal(D) / 16.0).astype(np.float32)).to(dev)
            yb = Xb @ wb + 0.5 * torch.randn(rows, generator=gb, device=dev)
            self.batches.append((Xb, yb))
        self.rotate = len(self.batches) > 1 and not args.reproducible
        self._visit = 0
        self._spin_visit = 0
        self.n_total = float(global_rows)       # the resident global batch is the data set
        self.engine = None
        if getattr(args, "via", "driver") == "plugin":
            if world > 1 or args.unfused or args.reproducible:
                raise SystemExit("bench.py: --via plugin is the single-GPU fused route")
            self.attach_plugin_engine()
        else:
            self.model = BLRReparamSVI(X, y, n_total=self.n_total, n_samples=S, seed=1234, lr=1e-3, ctx=ctx,
                                       group=args.exchange_group, fused=not args.unfused,
                                       reproducible=args.reproducible, sweep=args.sweep)
        self.units_per_step = global_rows / 1e6 if args.scaling == "strong" else world * rows / 1e6
        self.describe = ("cfg2: Bayesian linear regression (Normal-InverseGamma), %dx%d f32 mini-batch %s, "
                         "reparam-trick ELBO, S=%d, Adam"
                         % (rows, D, "per GPU" if args.scaling == "weak" else
                            "block of a global %d-row batch" % global_rows, S))
        self.config = {"rows_per_gpu": rows, "global_rows": global_rows, "dim": D, "mc_samples": S,
                       "reproducible": bool(args.reproducible), "sweep": args.sweep,
                       "via": ("plugin surface: inference.ReparamVI over Normal / InverseGamma nodes + bayesic.algebra "
                               "(route = %s)" % self.engine.route) if self.engine is not None else
                              "hand-written host driver svi/blr.py",
                       "minibatches_resident": len(self.batches),
                       "value_is": ("update loop rotating over %d distinct HBM-resident mini-batches, a different "
                                    "one every step (no cross-update Infinity-Cache reuse)" % len(self.batches))
                       if self.rotate else "update loop over ONE resident mini-batch (sweep = %s)" % args.sweep}

    def attach_plugin_engine(self):
        """The SAME model on the plugin surface (Normal / InverseGamma nodes + bayesic.algebra expressions, stepped by
        inference.ReparamVI) over the mini-batches already resident: from here on step() goes through the engine."""
        import numpy as np
        from bayesic_amd.algebra.device_backend import DeviceBackend
        from bayesic_amd.inference import ReparamVI
        from bayesic_amd.inference.models import linear_regression_log_joint
        D, S = self.D, self.S
        lj, v = linear_regression_log_joint(self.n_total / self.rows, 1.0, 1.0)
        lam0 = np.zeros(2 * (D + 1))
        lam0[D + 1:] = np.log(0.1)              # the driver's starting point (oracle.svi.blr_init_lam)
        X, y = self.batches[0]
        self.engine = ReparamVI(lj, [(v["W"], D), (v["xi"], 1)], dict(X=X, y=y), n_samples=S, seed=1234,
                                lr=1e-3, backend=DeviceBackend(self.ctx), lam0=lam0, route="fused")
        self.model = self.engine._fused         # (the harness reads W / sweep bookkeeping off the driver behind it)
        self.model.sweep = self.args.sweep
        return self.engine.route

    def set_mode(self, mode):
        """'rotate' (a different resident mini-batch every step) or 'same' (batch 0 every step)."""
        self.rotate = mode == "rotate" and len(self.batches) > 1 and not self.args.reproducible
        if not self.rotate:
            self._set_batch(*self.batches[0])

    def _set_batch(self, X, y):
        if self.engine is not None:
            self.engine.set_data(X=X, y=y)
        else:
            self.model.set_batch(X, y)

    def spin(self):
        # the same launch the update loop makes: streaming over the next batch of the rotation, or --
        # one resident batch -- in the loop's sweep order (alternating directions unless --sweep stream)
        code = 0
        X, y = self.batches[0]
        if self.rotate:
            self._spin_visit = (self._spin_visit + 1) % len(self.batches)
            X, y = self.batches[self._spin_visit]
        elif self.model.sweep == "alternate" and not self.model.reproducible:
            code = self._spin_sweep = 3 - getattr(self, "_spin_sweep", 2)
        self.ctx.call("bsc_blr_data_pass_partial_sweep", X, X.stride(0), y, self.rows,
                      self.D, self.model.W, min(self.S, 8), code)

    def step(self):
        if self.rotate:
            self._visit = (self._visit + 1) % len(self.batches)
            self._set_batch(*self.batches[self._visit])
        (self.engine or self.model).step()

    def result(self):
        return {"final_elbo": float(self.model.elbo.item())}

    def roofline(self, avg_s):
        algo = 4.0 * self.rows * self.D + 4.0 * self.rows      # X and y read once per launch
        # which pass kernel the context's options select (csrc/bsc_blr.hip launch_pass), asked of the library
        opt = self.ctx.get_option
        if self.D == 256 and opt("blr_tile_rows") == 16 and min(self.S, 8) == self.S:
            kernel = ("blr_pass_mx_kernel" if opt("blr_mx") else "blr_pass_q_kernel" if opt("blr_q") else
                      "blr_pass_dma_kernel" if opt("blr_dma") and opt("blr_pk") else "blr_pass_mfma_kernel")
        elif self.D == 256 and opt("blr_tile_rows") == 16:
            kernel = "blr_pass_mx_kernel"
        else:
            kernel = "blr_pass_kernel"
        achieved = algo / avg_s / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(kernel, self.rows == 1_000_000 and self.D == 256),
                "algorithmic_bytes_per_launch": algo,
                "launches_per_step": self.model._passes_per_update()}

    def cpu_baseline(self, budget_s):
        """Three legs on this host, each bounded to ~budget_s: (a) numpy float32/BLAS executing the
        lowered op tree at all cores [the reported baseline, SURVEY 8(d)], (b) the same at 1 core,
        (c) the float64 C + OpenMP restatement of the pass (oracle/c)."""
        import numpy as np
        from oracle import lowered_baseline as lb
        from oracle import svi
        Xh, yh = self.host if self.host is not None else (self.X.cpu().numpy(), self.y.cpu().numpy())
        S, D = self.S, self.D

        def run(data_pass, budget, max_updates=100):
            lam = svi.blr_init_lam(D)
            m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
            lam, m1, m2, _, _ = svi.blr_step(lam, m1, m2, 1, Xh, yh, S, 1234, self.n_total, 0.01,
                                             chunked=True, data_pass=data_pass)    # warm-up
            done, t0 = 0, time.perf_counter()
            while done < max_updates and (time.perf_counter() - t0 < budget or done == 0):
                lam, m1, m2, _, _ = svi.blr_step(lam, m1, m2, done + 2, Xh, yh, S, 1234, self.n_total,
                                                 0.01, chunked=True, data_pass=data_pass)
                done += 1
            dt = time.perf_counter() - t0
            return done, dt

        unit = "updates/s (%d-row mini-batch)" % Xh.shape[0]
        facts = lb.host_facts()
        pools = [p.get("num_threads") for p in facts["blas"]] if isinstance(facts.get("blas"), list) else []
        blas_threads = max([t for t in pools if t] or [1])
        n, dt = run(lb.blr_data_pass_lowered, budget_s)
        # cores = the threads this leg actually ran on: the BLAS pool (the GEMMs of the lowered tree); its
        # element-wise nodes are single-threaded numpy.  The host's logical CPU count is in host.cpu_count.
        out = {"value": n / dt, "unit": unit, "cores": blas_threads, "threads": blas_threads, "kind": "port",
               "sample": "%d full updates on the same %dx%d mini-batch in %.1f s: numpy float32 / BLAS "
                         "executing the lowered op tree of the pass (%s), float64 numpy finish; "
                         "numpy restatement, not Theano"
                         % (n, Xh.shape[0], D, dt, lb.blr_pass_functions()["lowered"]),
               "host": facts}
        with lb.threads(1):
            n, dt = run(lb.blr_data_pass_lowered, budget_s)
        out["one_core"] = {"value": n / dt, "unit": unit, "cores": 1, "threads": 1,
                           "sample": "%d updates in %.1f s, BLAS threads = 1" % (n, dt)}
        try:
            from oracle import cbuild
            threads = cbuild.use_cpu_share()      # (the container's CPU share, not the host's CPU count)
            n, dt = run(cbuild.blr_data_pass, budget_s)
            out["c_port"] = {"value": n / dt, "unit": unit, "cores": threads, "threads": threads,
                             "sample": "%d updates in %.1f s: float64 C + OpenMP restatement of the pass "
                                       "(oracle/c), scalar inner loops" % (n, dt)}
        except Exception as e:   # no gcc and no prebuilt library on this host
            out["c_port"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        # (d) what a host implementation that wanted to be fast would do: ONE fused float32 pass over X, every
        # thread streaming its own row block from memory it touched first (the honest gap)
        try:
            from oracle import cbuild
            threads = cbuild.use_cpu_share()      # (the container's CPU share, not the host's CPU count)
            if S <= 16 and D <= 1024:
                keep = (Xh, yh)
                Xh, yh = cbuild.first_touch_copy(Xh), cbuild.first_touch_copy(yh)
                n, dt = run(cbuild.blr_data_pass_f32, min(budget_s, 6.0), max_updates=400)
                Xh, yh = keep
                out["fused_f32"] = {"value": n / dt, "unit": unit, "cores": threads, "threads": threads, "kind": "port",
                                    "gbytes_per_s": n / dt * (4.0 * Xh.shape[0] * D + 4.0 * Xh.shape[0]) / 1e9,
                                    "sample": "%d updates in %.1f s: one fused float32 OpenMP pass over X per update "
                                              "(oracle/c oracle_blr_data_pass_f32: AVX2 + FMA vectors, eight draws per "
                                              "row in registers, per-thread accumulators in L1, float64 across threads), "
                                              "row blocks first-touched by the threads that stream them, float64 numpy "
                                              "finish" % (n, dt)}
        except Exception as e:
            out["fused_f32"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        return out


def _mfma_roofline(kernel, flops, algo_bytes, avg_s, traffic, split=0):
    achieved = flops / avg_s / 1e12
    out = {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": F32_MFMA_PEAK_TF,
           "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TF, "traffic": traffic,
           "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": algo_bytes,
           "hbm_frac": algo_bytes / avg_s / 1e9 / HBM_PEAK_GBS, "launches_per_step": 1}
    if split:
        # every f32 product is 3 (two terms) or 6 (three terms) bf16 products: the flops the matrix pipe executes
        # are priced against the dense bf16 peak; `f32_equivalent_tflops` is the algorithmic rate
        products = 3 if split == 2 else 6
        out.update({"achieved": products * achieved, "peak": BF16_MFMA_PEAK_TF,
                    "frac": products * achieved / BF16_MFMA_PEAK_TF, "f32_equivalent_tflops": achieved,
                    "bf16_products_per_f32_product": products})
    return out


class Cfg3(Workload):
    """Mixture of Gaussians K=64, discrete latent marginalised by summation, natural-gradient SVI."""
    name = "cfg3"
    default_rows = 10_000_000
    kernel_ms = 0.72

    def __init__(self, args, ctx, torch, rank, world):
        import numpy as np
        from bayesic_amd.svi import mog as mog_mod
        self.ctx = ctx
        D, K = 16, 64
        total = args.rows or self.default_rows
        if args.scaling == "strong":
            r0, r1 = rank_rows(total, rank, world)
            global_rows, seed_off = total, 0
        else:
            r0, r1, global_rows, seed_off = 0, total, total * world, rank
        rows = r1 - r0
        # centres RandomState(3)*4, labels RandomState(4), unit noise RandomState(5)
        centres = np.random.RandomState(3).standard_normal((K, D)) * 4.0
        labels = np.random.RandomState(4 + 11 * seed_off).randint(K, size=r1)[r0:]
        Xh = (centres[labels] + _normal_f32(np, 5 + 11 * seed_off, (r1, D))[r0:]).astype(np.float32)
        self.host = Xh
        self.X = torch.from_numpy(Xh).to(ctx.device)
        self.rows, self.D, self.K = rows, D, K
        self.mfma_split = args.mfma_split
        eta0 = mog_mod.prior_eta(K, D)
        eta_init = mog_mod.init_eta(_normal_f32(np, 5, (2000, D)) +
                                    centres[np.random.RandomState(4).randint(K, size=2000)], K, D, seed=2)
        self.n_total = float(global_rows)
        self.model = mog_mod.MoGNatGradSVI(self.X, K, eta0, eta_init, n_total=self.n_total, ctx=ctx,
                                           group=args.exchange_group)
        self.model.expected_params()
        self.units_per_step = 1.0 if args.scaling == "strong" else float(world)
        self.describe = ("cfg3: mixture of Gaussians K=%d, %dx%d f32 mini-batch %s, discrete latent "
                         "marginalised by summation, natural-gradient SVI"
                         % (K, rows, D, "per GPU" if args.scaling == "weak" else "block of a global "
                            "%d-row batch" % global_rows))
        self.config = {"rows_per_gpu": rows, "global_rows": global_rows, "dim": D, "components": K,
                       "inputs": "SURVEY 8(d): centres RandomState(3) * 4, labels RandomState(4), unit noise RandomState(5), "
                                 "drawn on the host",
                       "cpu_baseline_is": "timed on the first min(rows, 1 000 000) rows and scaled to the batch"}

    def spin(self):
        self.model.local_step()

    def step(self):
        self.model.step()

    def result(self):
        return {"final_elbo": float(self.model.elbo.item()), "final_bound_term": float(self.model.lse.item())}

    def roofline(self, avg_s):
        if self.mfma_split:
            # forward on three terms (6 products), backward on two (3): 4.5 bf16 products per f32 product on average.
            # (The split pass is bound by its vector instructions -- 490 a tile against 36 MFMAs -- not by either roof.)
            out = _mfma_roofline("mog_estep_bx_kernel", 8.0 * self.K * self.D * self.rows, 4.0 * self.rows * self.D,
                                 avg_s, pmc_traffic("mog_estep_bx_kernel", self.rows == 10_000_000), split=2)
            f32eq = out["f32_equivalent_tflops"]
            out.update({"achieved": 4.5 * f32eq, "frac": 4.5 * f32eq / BF16_MFMA_PEAK_TF, "bf16_products_per_f32_product": 4.5})
            return out
        return _mfma_roofline("mog_estep_kernel", 8.0 * self.K * self.D * self.rows,
                              4.0 * self.rows * self.D, avg_s, pmc_traffic("mog_estep_kernel",
                                                                           self.rows == 10_000_000))

    def cpu_baseline(self, budget_s):
        from oracle import cbuild
        import numpy as np
        n = min(self.rows, 1_000_000)
        W, c = self.model.Wmat.cpu().numpy(), self.model.c.cpu().numpy()
        threads = cbuild.use_cpu_share()      # (the container's CPU share, not the host's CPU count)
        cbuild.mog_estep(self.host[:50_000], W, c)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s or done == 0:
            cbuild.mog_estep(self.host[:n], W, c)
            done += 1
        dt = time.perf_counter() - t0
        return {"value": done * (n / self.rows) / dt, "unit": "updates/s (%d-row mini-batch)" % self.rows,
                "cores": threads, "kind": "port",
                "sample": "%d E-step passes over the first %d rows (scaled to the %d-row batch) in %.1f s: "
                          "float64 C + OpenMP restatement (oracle/c); the parameter-sized steps are not "
                          "timed" % (done, n, self.rows, dt)}


class Cfg5(Workload):
    """Hierarchical logistic regression, BBVI score-function gradient + control variate, S=64."""
    name = "cfg5"
    default_rows, default_samples = 1_000_000, 64
    kernel_ms = 0.31

    def __init__(self, args, ctx, torch, rank, world):
        import numpy as np
        from bayesic_amd.svi.bbvi import LogRegBBVI
        self.ctx = ctx
        D, G, S = args.dim, 1000, args.samples or self.default_samples
        total = args.rows or self.default_rows
        if args.scaling == "strong":
            r0, r1 = rank_rows(total, rank, world)
            global_rows, seed_off = total, 0
        else:
            r0, r1, global_rows, seed_off = 0, total, total * world, rank
        rows = r1 - r0
        Xh = _normal_f32(np, 1234 + seed_off, (r1, D))[r0:]
        w_true = np.random.RandomState(1).standard_normal(D) / 16.0
        b_true = np.random.RandomState(7).standard_normal(G) * 0.5
        gh = np.random.RandomState(6 + 13 * seed_off).randint(G, size=r1).astype(np.int32)[r0:]
        logits = Xh.astype(np.float64) @ w_true + b_true[gh]
        yh = (np.random.RandomState(8 + 13 * seed_off).uniform(size=r1)[r0:] <
              1.0 / (1.0 + np.exp(-logits))).astype(np.float32)
        self.host = (Xh, yh, gh)
        dev = ctx.device
        self.X, self.y, self.g = (torch.from_numpy(a).to(dev) for a in (Xh, yh, gh))
        self.rows, self.D, self.G, self.S = rows, D, G, S
        self.mfma_split = args.mfma_split
        self.n_total = float(global_rows)
        self.model = LogRegBBVI(self.X, self.y, self.g, G, n_total=self.n_total, n_samples=S, seed=1234,
                                lr=1e-3, ctx=ctx, group=args.exchange_group)
        self.units_per_step = global_rows / 1e6 if args.scaling == "strong" else world * rows / 1e6
        self.describe = ("cfg5: hierarchical logistic regression, %dx%d f32 mini-batch %s, G=%d groups, "
                         "BBVI score-function gradient with control variate, S=%d, Adam"
                         % (rows, D, "per GPU" if args.scaling == "weak" else "block of a global %d-row "
                            "batch" % global_rows, G, S))
        self.config = {"rows_per_gpu": rows, "global_rows": global_rows, "dim": D, "groups": G,
                       "mc_samples": S,
                       "inputs": "SURVEY 8(d): X RandomState(1234), groups RandomState(6), y ~ Bernoulli(sigmoid(X w* + b*_g)) "
                                 "with w* RandomState(1) / 16, b* RandomState(7) / 2, uniforms RandomState(8); drawn on the host",
                       "cpu_baseline_is": "timed on the first min(rows, 200 000) rows and scaled to the batch"}

    def spin(self):
        m = self.model
        self.ctx.call("bsc_logreg_bbvi_loglik", m.X, m.X.stride(0), m.y, m.g, m.B, m.D, m.G, m.Wz, m.Bz,
                      m.S, m.ell)

    def step(self):
        self.model.step()

    def result(self):
        return {"final_elbo": float(self.model.elbo.item())}

    def roofline(self, avg_s):
        # 32 flop/B: the fp32-MFMA time (209 us) exceeds the HBM time (129 us) -> MFMA binds
        dma = self.S % 4 == 0 and 32 < self.S <= 64
        kernel = "logreg_loglik_dma_kernel" if dma else "logreg_loglik_xreg_kernel"
        flops, algo = 2.0 * self.rows * self.D * self.S, 4.0 * self.rows * self.D + 8.0 * self.rows
        if dma and self.mfma_split == 2:
            # X and the draws as two bf16 terms: 3 x 32.8 GF on the bf16 MFMA is 40 us of matrix-pipe time against
            # 129 us of HBM time at the 8 TB/s peak -- the pass is priced against HBM
            kernel = "logreg_loglik_dma_bx_kernel"
            achieved = algo / avg_s / 1e9
            return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(kernel, self.rows == 1_000_000 and self.D == 256),
                    "algorithmic_bytes_per_launch": algo, "algorithmic_flops_per_launch": flops,
                    "f32_equivalent_tflops": flops / avg_s / 1e12, "bf16_products_per_f32_product": 3,
                    "launches_per_step": 1}
        return _mfma_roofline(kernel, flops, algo, avg_s, pmc_traffic(kernel, self.rows == 1_000_000 and self.D == 256))

    def cpu_baseline(self, budget_s):
        from oracle import cbuild
        Xh, yh, gh = self.host
        n = min(self.rows, 200_000)
        m = self.model
        Wz = m.Wz.cpu().numpy().reshape(self.S, self.D)
        Bz = m.Bz.cpu().numpy().reshape(self.G, self.S)
        threads = cbuild.use_cpu_share()      # (the container's CPU share, not the host's CPU count)
        cbuild.logreg_loglik(Xh[:20_000], yh[:20_000], gh[:20_000], Wz, Bz)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s or done == 0:
            cbuild.logreg_loglik(Xh[:n], yh[:n], gh[:n], Wz, Bz)
            done += 1
        dt = time.perf_counter() - t0
        return {"value": done * (n / self.rows) / dt * (self.rows / 1e6),
                "unit": "updates/s (1M-row mini-batch)", "cores": threads, "kind": "port",
                "sample": "%d log-likelihood passes over the first %d rows (scaled to 1M rows) in %.1f s: "
                          "float64 C + OpenMP restatement (oracle/c)" % (done, n, dt)}


class Cfg4(Workload):
    """LDA-style Dirichlet-Multinomial, fixed-gamma local step, K=128 topics, V=100k words."""
    name = "cfg4"
    default_rows = 6_250          # documents per GPU (50 000 over 8 GPUs)
    kernel_ms = 2.7

    def __init__(self, args, ctx, torch, rank, world):
        from bayesic_amd.svi.lda import LDAFixedGammaSVI
        self.ctx = ctx
        V, K = 100_000, 128
        if args.scaling == "strong":
            total = args.rows or 50_000
            r0, r1 = rank_rows(total, rank, world)
            docs, global_docs = r1 - r0, total
        else:
            docs = args.rows or self.default_rows
            global_docs = docs * world
        dev = ctx.device
        # C ~ Poisson(0.05) (RandomState(5).poisson) drawn on the device: 2.5 GB per
        # 6250-document shard is minutes of numpy; gamma, lambda as tools/bench_configs.py
        g = torch.Generator(device=dev).manual_seed(5 + rank)
        C = torch.empty((docs, V), device=dev)
        for d0 in range(0, docs, 1024):
            d1 = min(docs, d0 + 1024)
            C[d0:d1] = torch.poisson(torch.full((d1 - d0, V), 0.05, device=dev), generator=g)
        gp = torch.Generator(device=dev).manual_seed(17)
        gamma = torch.rand((docs, K), generator=g, device=dev) + 0.5
        lam = torch.rand((K, V), generator=gp, device=dev) + 0.5      # replicated: same on every rank
        self.docs, self.V, self.K = docs, V, K
        self.mfma_split = args.mfma_split
        self.model = LDAFixedGammaSVI(C, gamma, lam, docs_total=float(global_docs), ctx=ctx,
                                      group=args.exchange_group, overlap=not args.no_overlap)
        self.units_per_step = 1.0 if args.scaling == "strong" else float(world)
        self.describe = ("cfg4: LDA-style Dirichlet-Multinomial, %d docs x %d vocab f32 dense counts %s, "
                         "K=%d, fixed-gamma local step + natural-gradient step (all-reduce of %d x %d f32)"
                         % (docs, V, "per GPU" if args.scaling == "weak" else "block of %d docs" % global_docs,
                            K, K, V))
        self.config = {"docs_per_gpu": docs, "global_docs": global_docs, "vocab": V, "topics": K,
                       "exchange_pieces": ("the statistic's all-reduce in column ranges of whole kernel rounds, each on a "
                                           "second stream beside the next range's kernel (allreduce_us = mean per piece)")
                       if (self.model.overlap and self.model.exchange.active and self.model.via == "kernel") else
                       "one all-reduce after the whole statistic",
                       "inputs": "counts: torch.poisson(0.05) on the device, generator seed 5 + rank -- NOT the "
                                 "RandomState(5).poisson(0.05) stream SURVEY 8(d) names (2.5 GB a shard is minutes of "
                                 "numpy); same law, other draws.  gamma, lambda: torch.rand + 0.5 on the device",
                       "cpu_baseline_is": "timed on the first 64 documents and scaled to the shard"}

    def spin(self):
        self.model.local_step()

    def step(self):
        self.model.step()

    def result(self):
        return {"final_elbo": float(self.model.elbo.item()), "lambda_sum": float(self.model.lam.sum().item())}

    def roofline(self, avg_s):
        split = self.mfma_split
        # (the driver takes the words' term of the bound in the same pass: the ..._bound_kernel instantiations)
        kernel = "lda_sstats_bx%d_bound_kernel" % split if split else "lda_sstats_stream_bound_kernel"
        return _mfma_roofline(kernel, 4.0 * self.docs * self.V * self.K, 4.0 * self.docs * self.V, avg_s,
                              pmc_traffic(kernel, self.docs == 6250), split=split)

    def cpu_baseline(self, budget_s):
        from oracle import cbuild
        m = self.model
        n = min(self.docs, 64)
        C, Th, Bt = m.C[:n].cpu().numpy(), m.Th[:n].cpu().numpy(), m.Bt.cpu().numpy()
        threads = cbuild.use_cpu_share()      # (the container's CPU share, not the host's CPU count)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s or done == 0:
            cbuild.lda_sstats(C, Th, Bt)
            done += 1
        dt = time.perf_counter() - t0
        return {"value": done * (n / self.docs) / dt, "unit": "updates/s (%d-document shard)" % self.docs,
                "cores": threads, "kind": "port",
                "sample": "%d statistic passes over the first %d documents (scaled to %d) in %.1f s: "
                          "float64 C + OpenMP restatement (oracle/c)" % (done, n, self.docs, dt)}


WORKLOADS = {"cfg2": Cfg2, "cfg3": Cfg3, "cfg4": Cfg4, "cfg5": Cfg5}


def pmc_traffic(kernel, at_counted_size):
    """HBM bytes per launch from the PMC pass recorded in profiles/pmc_traffic.json -- only when
    the counters were taken at this problem size AND from the kernel source as it is now (the file
    stores the sha1 of the .hip file the counted build came from)."""
    if not at_counted_size:
        return None
    try:
        import hashlib
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(kernel)
        if not rec:
            return None
        src = os.path.join(ROOT, "bayesic_amd", "csrc", rec.get("source", ""))
        if rec.get("source_sha1") and os.path.exists(src):
            if hashlib.sha1(open(src, "rb").read()).hexdigest() != rec["source_sha1"]:
                return None      # counters predate the current kernel: not this kernel's traffic
        return rec.get("hbm_bytes_per_launch")
    except Exception:
        return None


def agree_on_exchange(dist, torch, rank, init, drop, fallback):
    """Every rank tries ``init()`` (bsc_comm_init_rank through svi.exchange.init_comm); the ranks then agree over
    the HOST channel whether ALL of them succeeded.  If any failed, every rank drops what it built (``drop()``) and
    all of them build the fallback group together (``fallback()``: a torch.distributed RCCL process group -- same
    library, same xGMI links, two stream hops more).  A communicator that came up on some ranks only would hang the
    first collective; a rank that raised alone would leave the others waiting in it.  Returns (group or None, note
    or None).  Collective: all ranks must call it.  (tests/test_bench_fallback_cpu.py drives it under gloo.)"""
    err = None
    try:
        init()
    except Exception as e:      # noqa: BLE001 -- reported, not swallowed
        err = "%s: %s" % (type(e).__name__, e)
        sys.stderr.write("bench.py rank %d: bsc_comm_init_rank failed (%s)\n" % (rank, err))
    flag = torch.tensor([0 if err is None else 1], dtype=torch.int32)
    dist.all_reduce(flag)                                   # host tensor over the gloo channel
    failed = int(flag.item())
    if failed == 0:
        return None, None
    drop()
    group = fallback()
    note = ("torch.distributed RCCL process group (fallback: bsc_comm_init_rank failed on %d rank(s)%s)"
            % (failed, "; this rank: " + err if err else ""))
    return group, note


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # rehearsal on a one-GPU box: BSC_BENCH_REHEARSAL=1 puts every rank on GPU 0 and exchanges
    # over gloo (RCCL needs one device per rank); numbers from it mean nothing
    rehearsal = os.environ.get("BSC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the HSA runtime starts
    # the contract is ONE line on stdout: keep the real stdout for it and point fd 1 at stderr for
    # everything else (gloo announces "[Gloo] Rank 0 is connected ..." on fd 1, RCCL can be chatty too)
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    if world > 1:
        # host channel only: unique id, barrier, max over ranks
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from bayesic_amd.device import Context
    from bayesic_amd.svi.exchange import init_comm

    ctx = Context(local_rank)
    if args.mfma_split:
        ctx.call("bsc_ctx_set_mfma_split", args.mfma_split)
    args.exchange_group = None
    exchange_note = None
    if world > 1 and not rehearsal:
        # the product route: an RCCL communicator held by the context (bsc_comm_init_rank).  Should
        # that fail on ANY rank (it has only ever run at world size 1: gpurun boxes have one GPU), all
        # ranks agree over the host channel to exchange through torch.distributed's own RCCL process
        # group instead -- same library, same xGMI links, two stream hops more -- and the line says so.
        args.exchange_group, exchange_note = agree_on_exchange(
            dist, torch, rank, init=lambda: init_comm(ctx),
            drop=lambda: ctx.comm_destroy() if ctx.has_comm else None,
            fallback=lambda: dist.new_group(backend="nccl"))
    elif world == 1 and args.rccl_world1:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_max(values):
        t = torch.tensor(list(values), dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def measure(wl, spin_up=True, burst=True):
        """Spin-up, W warm-up steps, then timed blocks of EXACTLY K steps each (barrier + synchronize on both
        sides, max over ranks), repeated until --min-timed-s has been measured; then the dominant kernel's
        burst.  Returns a dict."""
        spin_launches = int(args.spin_up_ms / wl.kernel_ms) if (spin_up and args.spin_up_ms > 0) else 0
        for _ in range(spin_launches):
            wl.spin()
        for _ in range(args.warmup):
            wl.step()
        torch.cuda.synchronize()
        timing = not args.no_kernel_timing
        ctx.profile(max(1, args.time_every) if timing else 0)
        blocks, n_blocks = [], 1
        while len(blocks) < n_blocks:
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                wl.step()
            torch.cuda.synchronize()
            barrier()
            blocks.append(time.perf_counter() - t0)
            if len(blocks) == 1:       # every rank derives the same count from the same (max-reduced) time
                first = reduce_max(blocks)[0]
                n_blocks = max(1, min(args.max_blocks, int(-(-args.min_timed_s // max(first, 1e-9)))))
        in_region = {s: ctx.profile_read(s) for s in (0, 1, 2)} if timing else {}
        ctx.profile(0)
        # dedicated burst AFTER the timed region: every launch of the dominant kernel timed on the
        # launch stream, so roofline.avg_launch_us rests on >= 50 samples whatever --steps was
        burst_ms, burst_n, pairs = 0.0, 0, (0.0, 0)
        if timing and burst and args.burst > 0:
            # (a) the launches back to back between TWO events: launch-to-launch period, nothing else in the stream --
            # the figure rocprofv3's per-dispatch durations add up to.  (b) one event pair per launch, as rounds 1-3
            # reported it: the records themselves cost 1.6-2.2 us per launch (tools/launch_train.py), kept for comparison.
            ctx.profile(1)
            for _ in range(args.burst):
                wl.spin()
            pairs = ctx.profile_read(0)
            ctx.profile(0)
            burst_ms, burst_n = pairs
            if getattr(wl, "spin_is_one_kernel", False):        # (a spin() that also launches a reduce is timed per kernel)
                e0, e1 = ctx.event(), ctx.event()
                e0.record()
                for _ in range(args.burst):
                    wl.spin()
                e1.record()
                burst_ms, burst_n = e0.elapsed_ms(e1), args.burst
        blocks = sorted(reduce_max(blocks))
        median = blocks[len(blocks) // 2] if len(blocks) % 2 else 0.5 * (blocks[len(blocks) // 2 - 1] + blocks[len(blocks) // 2])
        return {"blocks": blocks, "median_s": median, "in_region": in_region, "burst": (burst_ms, burst_n),
                "burst_pairs": pairs, "spin_launches": spin_launches}

    def block_stats(m):
        k = args.steps
        return {"n": len(m["blocks"]), "steps_per_block": k, "ms_per_step_min": m["blocks"][0] / k * 1e3,
                "ms_per_step_median": m["median_s"] / k * 1e3, "ms_per_step_max": m["blocks"][-1] / k * 1e3,
                "timed_s": sum(m["blocks"])}

    import copy
    if args.scaling == "both":
        modes = ["strong", "weak"] if world > 1 else ["weak"]
    else:
        modes = [args.scaling]
    runs = {}
    wl = None
    for mode in modes:
        del wl
        margs = copy.copy(args)
        margs.scaling = mode
        wl = WORKLOADS[args.config](margs, ctx, torch, rank, world)
        m = measure(wl)
        m["units_per_step"], m["describe"], m["config"] = wl.units_per_step, wl.describe, wl.config
        if mode == modes[0]:
            m["roofline_wl"] = wl
            # config 2: the same loop over ONE resident mini-batch (alternating sweeps) beside the rotation
            if args.config == "cfg2" and getattr(wl, "rotate", False):
                wl.set_mode("same")
                m["same"] = measure(wl, spin_up=False)
                wl.set_mode("rotate")
            m["read_ceiling"] = ctx.read_probe(getattr(wl, "X_all", wl.X)) if (rank == 0 and args.config == "cfg2") else None
            m["extra"] = wl.result()
            # config 2 by default: the same loop once more with the model written on the reference's plugin surface
            # (row P of the coverage table: recognition must reach the same kernels at the same speed)
            if (args.config == "cfg2" and world == 1 and args.via == "driver" and not args.unfused
                    and not args.reproducible and not args.no_via_plugin):
                try:
                    route = wl.attach_plugin_engine()
                    wl.set_mode("rotate")
                    m["plugin"] = measure(wl, spin_up=False, burst=False)
                    m["plugin"]["route"] = route
                except Exception as e:      # noqa: BLE001 -- reported in the line, the headline stands
                    m["plugin"] = {"error": "%s: %s" % (type(e).__name__, e)}
        runs[mode] = m
    head = runs[modes[0]]

    if rank == 0:
        k = args.steps
        ms_per_step = head["median_s"] / k * 1e3
        value = head["units_per_step"] * k / head["median_s"]
        roofline = None
        burst_ms, burst_n = head["burst"]
        rwl = head["roofline_wl"]
        if burst_n:
            avg_s = burst_ms / burst_n * 1e-3
            roofline = rwl.roofline(avg_s)
            train = getattr(rwl, "spin_is_one_kernel", False)
            roofline.update({"avg_launch_us": avg_s * 1e6, "launches": burst_n,
                             "timed": ("burst of %d consecutive launches after the timed region between two hipEvents on "
                                       "the launch stream (launch-to-launch period)" if train else
                                       "burst of %d consecutive launches after the timed region, one hipEvent pair "
                                       "each on the launch stream") % burst_n})
            pm, pn = head.get("burst_pairs", (0.0, 0))
            if pn and train:
                roofline["avg_launch_us_event_pair_each"] = pm / pn * 1e3     # (rounds 1-3 reported this one)
            ms0, n0 = head["in_region"].get(0, (0.0, 0))
            if n0:
                roofline["avg_launch_us_in_timed_region"] = ms0 / n0 * 1e3     # (one event pair per sampled launch)
                roofline["launches_in_timed_region"] = n0
            if head.get("read_ceiling"):
                roofline["read_ceiling_this_box"] = head["read_ceiling"]
                roofline["frac_of_read_ceiling"] = roofline["achieved"] / head["read_ceiling"]
            if "same" in head and head["same"]["burst"][1]:
                sb_ms, sb_n = head["same"]["burst"]
                sb = rwl.roofline(sb_ms / sb_n * 1e-3)
                roofline["passes"] = "each launch streams a mini-batch the previous launch did not touch"
                roofline["same_batch"] = {"achieved": sb["achieved"], "frac": sb["frac"],
                                          "avg_launch_us": sb_ms / sb_n * 1e3, "launches": sb_n,
                                          "passes": "alternating sweeps over ONE resident mini-batch: each launch "
                                                    "starts in the ~230 MB the previous one left in the Infinity Cache"}
        ms1, n1 = head["in_region"].get(1, (0.0, 0))
        ms2, n2 = head["in_region"].get(2, (0.0, 0))
        info = ctx.comm_info()
        out = {
            "metric": "ELBO-grad updates/sec (1M-row mini-batch)",
            "value": value,
            "unit": "updates/s (1M-row mini-batch equivalents, all GPUs)" if args.config in ("cfg2", "cfg5")
                    else "updates/s (whole-job mini-batch updates, all GPUs)",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": modes[0],
            "vs_baseline": None,
            "dtype": "f32" if not args.mfma_split else
                     "f32 operands as sums of %d bf16 terms, %d bf16 products per product, f32 accumulation "
                     "(--mfma-split; NOT the headline dtype)" % (args.mfma_split, 3 if args.mfma_split == 2 else 6),
            "data": "synthetic",
            "config": dict({"workload": head["describe"], "parallelism": "dp%d" % world}, **head["config"]),
            "timed_blocks": block_stats(head),
            "roofline": roofline,
            "allreduce_us": (ms1 / n1 * 1e3) if n1 else None,
            "finish_us": (ms2 / n2 * 1e3) if n2 else None,
            "rccl_ranks": info["world"] if ctx.has_comm else 0,
            "rccl_version": info["rccl_version"] if ctx.has_comm else None,
            "exchange": ("rccl via bsc_allreduce_sum on the ctx stream" if ctx.has_comm else
                         exchange_note if exchange_note else
                         ("gloo (rehearsal: all ranks on one GPU)" if world > 1 else "none (one rank)")),
            "spin_up_launches": head["spin_launches"],
        }
        if "plugin" in head:
            pl = head["plugin"]
            if "error" in pl:
                out["value_via_plugin"] = None
                out["via_plugin"] = pl
            else:
                out["value_via_plugin"] = head["units_per_step"] * k / pl["median_s"]
                out["via_plugin"] = {"ms_per_step": pl["median_s"] / k * 1e3, "route": pl["route"],
                                     "timed_blocks": block_stats(pl),
                                     "what": "the same update loop with the model written as Normal / InverseGamma nodes and "
                                             "bayesic.algebra expressions (inference/models.py), stepped by inference.ReparamVI: "
                                             "the engine has to recognise the data term itself (inference/recognise.py)"}
        if "same" in head:
            same = head["same"]
            out["value_same_batch"] = head["units_per_step"] * k / same["median_s"]
            out["ms_per_step_same_batch"] = same["median_s"] / k * 1e3
            out["timed_blocks_same_batch"] = block_stats(same)
        if len(modes) > 1:
            weak = runs["weak"]
            out["value_weak"] = weak["units_per_step"] * k / weak["median_s"]
            out["ms_per_step_weak"] = weak["median_s"] / k * 1e3
            out["timed_blocks_weak"] = block_stats(weak)
            out["config_weak"] = dict({"workload": weak["describe"]}, **weak["config"])
            out["scaling_note"] = ("value = ONE global mini-batch of the configured size split over the ranks "
                                   "(the metric's reading); value_weak = one full-size mini-batch PER rank")
        out.update(head["extra"])
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = rwl.cpu_baseline(args.cpu_budget_s)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        else:
            out["cpu_baseline"] = None
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    barrier()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
