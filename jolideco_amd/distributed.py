"""Multi-GPU plumbing for the joint fit: one process per GPU, `torch.distributed` with the
"nccl" backend (= RCCL over xGMI on ROCm).  The reference has no distributed code at all; this is
new.

Partitioning: the datasets go to the ranks by longest-processing-time-first on an estimated cost per dataset
(`lpt_assignment`; equal costs give the round-robin d mod R), the GMM prior is split by contiguous patch rows in
shares that top every rank up to the same estimated load (`balanced_shares`, `split_range`: a rank that owns a
dataset with a wide PSF evaluates fewer patch rows); theta, optimizer state and GMM constants are replicated.  Every
rank computes the same partition from the same inputs.  Per optimizer step: ONE sum all-reduce
of a flat buffer holding every component's likelihood gradient followed by the epoch's loss scalars
(16.8 MB per component at 2048^2), started asynchronously, and -- while it is in flight -- the rank's
band of the prior gradient, exchanged with ONE all-gather of the compact bands (`FitSession`,
JOLIDECO_DIST_OVERLAP=0: the prior accumulated into the flat buffer before one blocking all-reduce).
Every rank then applies the identical update, so no broadcast is needed.
"""
import os

import torch
import torch.distributed as dist

__all__ = ["DistContext", "init_from_env", "lpt_assignment", "balanced_shares", "split_range"]


def lpt_assignment(costs, world_size):
    """Owner rank of every item by longest-processing-time-first: items in order of falling cost (ties: by index), each to
    the least loaded rank (ties: the lowest rank).  Deterministic -- every rank computes the same table; equal costs give
    the round-robin ``i mod world_size``.  Returns (owners, loads per rank)."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads, owners = [0.0] * world_size, [0] * len(costs)
    for i in order:
        rank = min(range(world_size), key=lambda r: (loads[r], r))
        owners[i] = rank
        loads[rank] += float(costs[i])
    return owners, loads


def balanced_shares(loads, divisible):
    """Shares (sum 1) of a divisible piece of work of total cost ``divisible`` that top the ranks' fixed ``loads`` up to a
    common level where that is possible (water filling: a rank already above the level gets nothing); None when every
    rank gets the same share anyway."""
    n = len(loads)
    if n < 2 or divisible <= 0 or max(loads) - min(loads) <= 1e-12 * max(max(loads), 1.0):
        return None
    level_order = sorted(loads)
    level = None
    for k in range(n, 0, -1):  # the k least loaded ranks share the work
        candidate = (sum(level_order[:k]) + divisible) / k
        if candidate >= level_order[k - 1]:
            level = candidate
            break
    shares = [max(level - load, 0.0) / divisible for load in loads]
    total = sum(shares)
    return [s / total for s in shares]


def split_range(n, world_size, shares=None):
    """Contiguous [begin, end) slices of range(n), one per rank.  ``shares`` None: balanced, the first n mod R ranks get one
    extra element; else rank r gets about shares[r] * n elements (boundaries at the rounded cumulative shares)."""
    if shares is None:
        base, extra = divmod(n, world_size)
        out, begin = [], 0
        for r in range(world_size):
            end = begin + base + (1 if r < extra else 0)
            out.append((begin, end))
            begin = end
        return out
    total, acc, bounds = float(sum(shares)), 0.0, [0]
    for r in range(world_size):
        acc += float(shares[r])
        bounds.append(n if r == world_size - 1 else max(bounds[-1], min(n, int(round(n * acc / total)))))
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


class DistContext:
    """Rank / world size and the two sharding rules + the gradient all-reduce."""

    def __init__(self, rank=0, world_size=1, group=None, dry_run=False, force_collectives=False):
        self.rank = rank
        self.world_size = world_size
        self.group = group
        # dry_run: shard like rank `rank` of `world_size` but skip the collectives (single-process
        # timing of one rank's share of the step, bench.py --shard-of)
        self.dry_run = dry_run
        # force_collectives: take the sharded code path -- async all-reduce of the flat buffer, all-gather of the prior
        # bands, band sum + optimizer step -- even with ONE rank, through an initialised process group: how RCCL is
        # exercised on a single-GPU box (tests/test_gpu_distributed.py; the result is the un-sharded step's bit for bit)
        self.force_collectives = force_collectives

    @property
    def sharded(self):
        """The joint step goes through the collectives (more than one rank, or forced)."""
        return self.world_size > 1 or self.force_collectives

    @classmethod
    def current(cls):
        if dist.is_available() and dist.is_initialized():
            return cls(rank=dist.get_rank(), world_size=dist.get_world_size(), force_collectives=force_collectives_requested())
        return cls()

    def shard_items(self, items, costs=None):
        """This rank's items, in their original order.  Without ``costs`` round-robin (item i belongs to rank
        i mod world_size); with an estimated cost per item the longest-processing-time-first table of `lpt_assignment`."""
        if costs is None:
            return [item for i, item in enumerate(items) if i % self.world_size == self.rank]
        owners, _ = lpt_assignment(costs, self.world_size)
        return [item for item, owner in zip(items, owners) if owner == self.rank]

    def shard_range(self, n, shares=None):
        """Contiguous [begin, end) slice of range(n) for this rank: balanced (the first n mod R ranks get one extra
        element), or by ``shares`` per rank (`split_range`)."""
        return split_range(n, self.world_size, shares)[self.rank]

    def all_reduce_sum(self, buffer):
        """In-place sum all-reduce of one flat tensor (a no-op for a single process)."""
        if self.sharded and not self.dry_run:
            dist.all_reduce(buffer, op=dist.ReduceOp.SUM, group=self.group)
        return buffer

    def all_reduce_sum_async(self, buffer):
        """Start the in-place sum all-reduce of one flat tensor and return a handle whose ``wait()`` orders the
        current stream behind it (None for a single process).  RCCL runs the collective on its own stream, so kernels
        enqueued on the current stream after this call overlap with it."""
        if self.sharded and not self.dry_run:
            return dist.all_reduce(buffer, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def all_gather_flat(self, out, piece):
        """out[r * n : (r + 1) * n] <- rank r's ``piece`` (n = piece.numel()) on every rank.  With gloo (tests: several
        ranks on one GPU) device tensors are staged through the host; a dry run only places this rank's piece."""
        n = piece.numel()
        if out.numel() != n * self.world_size:
            raise ValueError("all_gather_flat: out must hold world_size pieces")
        if not self.sharded or self.dry_run:
            out[self.rank * n : (self.rank + 1) * n].copy_(piece)
            return out
        if dist.get_backend(self.group) == "gloo" and piece.is_cuda:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, piece.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, piece, group=self.group)
        return out

    def assert_same_on_all_ranks(self, values, what):
        """Raise on every rank if the int64 vector ``values`` (host) differs between ranks (one synchronising
        exchange at set-up time, e.g. the state of the cycle-spin generators that must draw identical shifts)."""
        if not self.sharded or self.dry_run:
            return
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else "cpu"
        mine = torch.as_tensor(values, dtype=torch.int64).to(device)
        both = torch.stack([mine, -mine])
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)  # max(v) and max(-v) = -min(v)
        if not torch.equal(both[0], -both[1]):
            raise RuntimeError(f"{what} differs between the ranks of the sharded fit")

    def barrier(self):
        if self.sharded and not self.dry_run:
            dist.barrier(group=self.group)


def force_collectives_requested():
    """JOLIDECO_FORCE_COLLECTIVES=1: a ONE-rank process group still runs the sharded joint step with its collectives
    (`DistContext.force_collectives`) -- RCCL on a single-GPU box."""
    return os.environ.get("JOLIDECO_FORCE_COLLECTIVES", "0") not in ("", "0")


def init_from_env(backend=None):
    """Initialise the default process group from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT) and bind this process to its GPU.  Returns a DistContext;
    a single-process run (no WORLD_SIZE or WORLD_SIZE=1) initialises nothing -- unless
    JOLIDECO_FORCE_COLLECTIVES=1 asks for a one-rank group whose collectives really run."""
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size <= 1 and not force_collectives_requested():
        return DistContext()
    os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1"), os.environ.setdefault("MASTER_PORT", "29533")
    # the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer registration fails with
    # "hipIpcGetMemHandle: invalid argument" (read by the HSA runtime when the first GPU call initialises it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        # JOLIDECO_DIST_BACKEND=gloo lets several ranks share ONE GPU (tests on a single-GPU box; gloo
        # stages device tensors through the host); the production backend is RCCL ("nccl")
        backend = os.environ.get("JOLIDECO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, **kwargs)
    return DistContext.current()
