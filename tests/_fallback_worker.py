"""One rank of tests/test_bench_fallback_cpu.py: bench.agree_on_exchange under gloo (CPU)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    out_path, failing = sys.argv[1], {int(r) for r in sys.argv[2].split(",") if r}
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []

    def init():
        log.append("init")
        if rank in failing:
            raise RuntimeError("ncclCommInitRank: unhandled system error (simulated on rank %d)" % rank)

    def drop():
        log.append("drop")

    def fallback():
        log.append("fallback")
        return dist.new_group(backend="gloo")           # (the product builds an RCCL group here: same collective call)

    group, note = bench.agree_on_exchange(dist, torch, rank, init, drop, fallback)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, group=group)                      # the exchange the drivers would make over what was agreed
    json.dump({"log": log, "group": group is not None, "note": note, "sum": float(t.item())}, open(out_path % rank, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
