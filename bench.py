"""ELBO-gradient updates/sec on BASELINE config 2 (Bayesian linear regression,
1M x 256 float32 mini-batch per GPU, reparameterisation-trick ELBO, S=8).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full update on the resident mini-batch: Philox sample ->
fused data pass over X,y -> float64 slab reduce -> (all-reduce of 16 KB over
RCCL when N>1) -> ELBO + pathwise gradient -> Adam step.  Inputs are resident in
HBM before the timed region.  Before the W warm-up steps the device is spun up with untimed
data-pass launches (--spin-up-ms, default 60): a process that has just started using the GPU
runs its first ~35 ms of kernels 10-40 % slower (tools/ramp_probe.py), and the metric is the
steady rate of a long-running job.  Weak scaling: every rank holds its own 1M rows, so
`value` counts 1M-row mini-batch equivalents per second over all ranks.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=1_000_000, help="mini-batch rows per GPU")
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true",
                    help="N=1 through the multi-GPU code path (float64 statistics between the pass "
                         "and the finish, no collective): what the N>1 step costs besides RCCL")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the hipEvent pair around the pass kernel")
    ap.add_argument("--spin-up-ms", type=float, default=60.0,
                    help="untimed data-pass launches before the W warm-up steps, until about this "
                         "much GPU work has been queued: a process that has just started using the "
                         "GPU runs the pass at 230 -> 165 us over its first ~35 ms "
                         "(tools/ramp_probe.py); a training job lives in the steady state")
    ap.add_argument("--time-every", type=int, default=8,
                    help="time every n-th pass-kernel launch inside the timed region (an event "
                         "pair costs ~4 us of stream time per use)")
    return ap.parse_args()


def make_data(torch, device, rank, rows, dim):
    """Synthetic cfg-2-shaped shard, generated on the device (seed depends on the
    rank so shards differ): X ~ N(0,1), y = X w* + 0.5 noise."""
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    X = torch.randn((rows, dim), generator=g, device=device, dtype=torch.float32)
    gw = torch.Generator(device=device).manual_seed(1)
    w_true = torch.randn(dim, generator=gw, device=device, dtype=torch.float32) / 16.0
    noise = torch.randn(rows, generator=g, device=device, dtype=torch.float32)
    y = X @ w_true + 0.5 * noise
    return X, y


def cpu_baseline(X_host, y_host, samples, n_total, budget_s=12.0, max_updates=100):
    """The oracle timed on this host: the data pass in plain C + OpenMP (oracle/c, float64
    accumulate) when gcc built it, else the numpy restatement.  Bounded sample: full 1M-row
    updates until ~budget_s of CPU work."""
    import numpy as np
    from oracle import svi
    data_pass, how = None, "numpy float64 restatement (oracle.svi.blr_step)"
    try:
        from oracle import cbuild
        threads = cbuild.load().oracle_threads()
        data_pass, how = cbuild.blr_data_pass, "C + OpenMP data pass (oracle/c), numpy finish"
    except Exception:
        try:
            from threadpoolctl import threadpool_info
            threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        except Exception:
            threads = os.cpu_count() or 1
    D = X_host.shape[1]
    lam = svi.blr_init_lam(D)
    m1, m2 = np.zeros_like(lam), np.zeros_like(lam)
    done, t0 = 0, time.perf_counter()
    while done < max_updates and (time.perf_counter() - t0 < budget_s or done == 0):
        lam, m1, m2, _, _ = svi.blr_step(lam, m1, m2, done + 1, X_host, y_host, samples, 1234,
                                         n_total, 0.01, chunked=True, data_pass=data_pass)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "updates/s (1M-row mini-batch)", "cores": int(threads),
            "kind": "port",
            "sample": "%d full updates on the same %dx%d mini-batch, %s, %.1f s"
                      % (done, X_host.shape[0], D, how, dt)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # rehearsal on a one-GPU box: BSC_BENCH_REHEARSAL=1 puts every rank on GPU 0 and uses gloo
    # for the collective (RCCL needs one device per rank); numbers from it mean nothing
    rehearsal = os.environ.get("BSC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    from bayesic_amd.device import Context
    from bayesic_amd.svi.blr import BLRReparamSVI

    ctx = Context(local_rank)
    X, y = make_data(torch, device, rank, args.rows, args.dim)
    n_total = float(args.rows * world)  # the resident global batch is the data set
    model = BLRReparamSVI(X, y, n_total=n_total, n_samples=args.samples, seed=1234, lr=1e-3,
                          ctx=ctx, fused=not args.unfused)

    def barrier():
        if world > 1:
            dist.barrier()

    # device spin-up (not steps: the model state is untouched, only the partial slab is written)
    spin_launches = int(args.spin_up_ms / 0.17) if args.spin_up_ms > 0 else 0
    for _ in range(spin_launches):
        ctx.call("bsc_blr_data_pass_partial", X, X.stride(0), y, args.rows, args.dim, model.W,
                 args.samples if args.samples <= 8 else 8)
    for _ in range(args.warmup):
        model.step()
    torch.cuda.synchronize()
    ctx.profile(0 if args.no_kernel_timing else max(1, args.time_every))
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    pass_ms, launches = ctx.profile_read() if not args.no_kernel_timing else (0.0, 0)
    ctx.profile(0)
    # what a kernel that only reads X achieves on THIS box (boxes differ by up to 20 %)
    read_ceiling = ctx.read_probe(X) if rank == 0 else None

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    elbo = float(model.elbo.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * (args.rows / 1e6) * args.steps / elapsed
        algo_bytes = 4.0 * args.rows * args.dim + 4.0 * args.rows  # X and y read once
        launches_per_step = (args.samples + 7) // 8
        roofline = None
        if launches:
            avg_s = pass_ms / launches * 1e-3
            achieved = algo_bytes / avg_s / 1e9
            traffic = None
            # D == 256: the variant with the forward pass on the MFMA pipe (csrc/bsc_blr.hip)
            kernel_name = "blr_pass_mfma_kernel" if args.dim == 256 and \
                os.environ.get("BSC_BLR_TILE_ROWS", "16") == "16" else "blr_pass_kernel"
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    if args.rows == 1_000_000 and args.dim == 256:   # the size the counters were taken at
                        traffic = json.load(open(pmc)).get(kernel_name, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes,
                        "avg_launch_us": avg_s * 1e6, "launches": launches,
                        "timed_every": args.time_every,
                        "launches_per_step": launches_per_step,
                        "read_ceiling_this_box": read_ceiling,
                        "frac_of_read_ceiling": achieved / read_ceiling if read_ceiling else None}
        out = {
            "metric": "ELBO-grad updates/sec (1M-row mini-batch)",
            "value": value,
            "unit": "updates/s (1M-row mini-batch equivalents, all GPUs)",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg2: Bayesian linear regression (Normal-InverseGamma), "
                            "%dx%d f32 mini-batch per GPU, reparam-trick ELBO, S=%d, Adam"
                            % (args.rows, args.dim, args.samples),
                "rows_per_gpu": args.rows, "dim": args.dim, "mc_samples": args.samples,
                "parallelism": "dp%d" % world,
            },
            "roofline": roofline,
            "spin_up_launches": spin_launches,
            "final_elbo": elbo,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X.cpu().numpy(), y.cpu().numpy(), args.samples,
                                               n_total)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
