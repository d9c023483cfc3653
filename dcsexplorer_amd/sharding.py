"""Multi-GPU partitioning of the decode path: plain range partition over streams, one process per GPU,
no data-path collective (streams are independent units).  torch.distributed is used
only for the rendezvous, the barrier around the timed region and the max-over-ranks of the time."""
import inspect
import os

from . import workloads


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def rank_streams(workload, rank):
    """the streams rank `rank` decodes: stream k of the rank is stream rank*n + k of the seeded corpus
    (same shape on every rank = weak scaling, disjoint content)"""
    fn = workloads.WORKLOADS[workload]
    if rank == 0:
        return fn()
    n = inspect.signature(fn).parameters["n_streams"].default
    return workloads.shifted(fn, rank * n)


def partition_by_frames(frame_counts, world):
    """cut points (world + 1) of contiguous stream ranges balanced by total FRAME count (the C ABI's
    dcs_partition_streams): rank r owns streams [cut[r], cut[r+1])"""
    from . import api
    return [int(c) for c in api.partition_streams(frame_counts, world)]


def rank_corpus(manifest, world, rank):
    """(lo, hi): the range of a corpus manifest rank `rank` decodes -- the same corpus cut `world` ways
    (strong scaling), ragged streams balanced by frames, no data-path collective"""
    from . import workloads
    cut = partition_by_frames(workloads.corpus_frames(manifest), world)
    return cut[rank], cut[rank + 1]


def partition_range(n_items, world, rank):
    """contiguous range [lo, hi) of n_items owned by `rank` (sizes differ by at most one)"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device=None):
    """max of a python float over all ranks (identity when not distributed)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
