"""The one exchange step of a data-parallel update (README.md:69-79).

Rows are iid (bayesic/distribution/base.py:146-172) and every per-batch quantity is a sum
over rows, so rank r holds a row block and the ranks exchange ONE vector per update: the
all-reduce(sum) of the concatenated statistic / gradient vector.

The product route is RCCL behind the C ABI: ``bsc_allreduce_sum`` on the context's own
stream (pass -> all-reduce -> finish is one in-order queue).  ``init_comm`` builds that
communicator from a torch.distributed job -- torch is only the host channel that carries
the 128-byte unique id from rank 0; any other channel does.

A context WITHOUT a communicator inside a torch.distributed job (the gloo CPU tests with
the oracle test double, or several ranks rehearsing on one GPU, where RCCL cannot run)
exchanges through ``torch.distributed.all_reduce`` instead.
"""
import torch


def _dist_ready():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def init_comm(ctx, group=None):
    """Give ``ctx`` an RCCL communicator spanning the ranks of the torch.distributed job
    (or of ``group``).  Collective.  Returns the world size."""
    if not _dist_ready():
        raise RuntimeError("init_comm needs an initialised torch.distributed job to carry the "
                           "unique id (any backend); for a world of one call "
                           "ctx.comm_init(ctx.comm_unique_id(), 0, 1)")
    rank = torch.distributed.get_rank(group)
    world = torch.distributed.get_world_size(group)
    box = [ctx.comm_unique_id() if rank == 0 else None]
    src = torch.distributed.get_global_rank(group, 0) if group is not None else 0
    torch.distributed.broadcast_object_list(box, src=src, group=group)
    ctx.comm_init(box[0], rank, world)
    return world


class Exchange:
    """all-reduce(sum) over the ranks that share one mini-batch."""

    def __init__(self, ctx, group=None):
        self.ctx = ctx
        self.group = group
        self.rccl = getattr(ctx, "comm_world", 1) > 1 or bool(getattr(ctx, "has_comm", False))
        if self.rccl:
            self.world = ctx.comm_world
        elif group is not None or _dist_ready():
            self.world = torch.distributed.get_world_size(group)
        else:
            self.world = 1

    def all_reduce(self, tensor):
        """In place; asynchronous on the context stream on the RCCL route."""
        if self.rccl:
            self.ctx.allreduce_sum(tensor)
        elif self.world > 1:
            torch.distributed.all_reduce(tensor, group=self.group)
        return tensor

    def begin(self, tensor, slot):
        """Start the all-reduce of `tensor` (contiguous) so that it OVERLAPS what the caller enqueues next;
        ``end(slot)`` before anything reads the sum.  RCCL route: a second stream of the context
        (bsc_allreduce_sum_begin); torch.distributed route: an asynchronous collective."""
        if self.rccl:
            self.ctx.allreduce_sum_begin(tensor, slot)
        elif self.world > 1:
            self._pending = getattr(self, "_pending", {})
            self._pending[slot] = torch.distributed.all_reduce(tensor, group=self.group, async_op=True)

    def end(self, slot):
        if self.rccl:
            self.ctx.allreduce_sum_end(slot)
        elif self.world > 1:
            work = getattr(self, "_pending", {}).pop(slot, None)
            if work is not None:
                work.wait()

    @property
    def active(self):
        """Whether a collective is actually issued (more than one rank, or a one-rank RCCL communicator)."""
        return self.rccl or self.world > 1

    @property
    def rank(self):
        if self.rccl:
            return self.ctx.comm_info()["rank"]
        if self.world > 1:
            return torch.distributed.get_rank(self.group)
        return 0

    def row_offset(self, local, device):
        """Global index of this rank's first row when rank r holds the r-th contiguous block:
        the exclusive prefix sum of the ranks' row counts (one all-reduce of a [world] vector in
        which every rank fills its own slot -- exact whatever the reduction order)."""
        if self.world == 1:
            return 0
        t = torch.zeros(self.world, dtype=torch.float64, device=device)
        t[self.rank] = float(local)
        self.all_reduce(t)
        return int(round(float(t[:self.rank].sum().item())))

    def global_count(self, local, device):
        """Sum of a per-rank scalar (the rank's rows) over all ranks, as a float."""
        t = torch.tensor([float(local)], dtype=torch.float64, device=device)
        if self.rccl or self.world > 1:
            self.all_reduce(t)
        return float(t.item())
