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


# --------------------------------------------------------------------------------------------- what the ranks exchange
class CommError(RuntimeError):
    pass


class Comm:
    """The little the N ranks of one node exchange: a barrier, the max of a float, small rows of floats (never frame data).
    backend: "single" (one rank, nothing to do), "nccl" (RCCL; the rows live on the rank's GPU), "gloo", or "store" -- the
    rendezvous' own TCP key-value store, the last resort when no process group could be formed on every rank."""

    def __init__(self, backend, rank, world, device=None, store=None, timeout_s=120.0, attempts=None):
        self.backend, self.rank, self.world = backend, rank, world
        self.device = device if backend == "nccl" else None          # where the small all-reduces live
        self._store, self._seq, self._timeout_s = store, 0, timeout_s
        self.attempts = attempts or []                               # [(backend, [vote of rank 0, ...]), ...] for the bench line

    def gather_rows(self, vec):
        """-> world rows, row r = rank r's `vec` (floats); every rank gets all rows"""
        import datetime
        import json
        vec = [float(x) for x in vec]
        if self.backend == "single":
            return [vec]
        if self.backend in ("nccl", "gloo"):
            import torch
            import torch.distributed as dist
            t = torch.zeros(self.world, max(1, len(vec)), dtype=torch.float64, device=self.device if self.device is not None else "cpu")
            if vec:
                t[self.rank] = torch.tensor(vec, dtype=torch.float64)
            dist.all_reduce(t)
            return [[float(x) for x in row[:len(vec)]] for row in t.cpu()]
        self._seq += 1
        self._store.set("op%d_r%d" % (self._seq, self.rank), json.dumps(vec))
        keys = ["op%d_r%d" % (self._seq, r) for r in range(self.world)]
        try:
            self._store.wait(keys, datetime.timedelta(seconds=self._timeout_s))
        except Exception as e:
            raise CommError("rank %d: the other ranks did not reach exchange %d within %.0f s (%s)" % (self.rank, self._seq, self._timeout_s, e))
        return [json.loads(self._store.get(k)) for k in keys]

    def post(self, key, text):
        """leave a note for the other ranks in the rendezvous store (how a failing rank says why)"""
        if self._store is not None:
            try:
                self._store.set("note_" + key, text[:2000])
            except Exception:
                pass

    def read(self, key):
        if self._store is None:
            return None
        try:
            if self._store.check(["note_" + key]):
                return self._store.get("note_" + key).decode(errors="replace")
        except Exception:
            pass
        return None

    def barrier(self):
        if self.backend in ("nccl", "gloo"):
            import torch.distributed as dist
            dist.barrier()
        elif self.backend == "store":
            self.gather_rows([])

    def max(self, value):
        if self.backend in ("nccl", "gloo"):
            import torch
            import torch.distributed as dist
            t = torch.tensor([value], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return max(row[0] for row in self.gather_rows([value]))

    def close(self):
        if self.backend in ("nccl", "gloo"):
            import torch.distributed as dist
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass


def _sabotage(backend, rank):
    """test hook: DCS_COMM_SABOTAGE="nccl,gloo:hang@1" makes the attempt on that backend fail (or never return), on every rank
    or on rank @r only -- how tests reach open_comm's fallbacks without a broken machine"""
    import time
    for item in filter(None, os.environ.get("DCS_COMM_SABOTAGE", "").split(",")):
        what, _, only = item.partition("@")
        name, _, mode = what.partition(":")
        if name == backend and (not only or int(only) == rank):
            if mode == "hang":
                while True:
                    time.sleep(1.0)
            raise RuntimeError("sabotaged %s (DCS_COMM_SABOTAGE)" % backend)


def open_comm(rank, world, want, device=None, init_timeout_s=120.0, log=None):
    """Form the ranks' process group without ever hanging for good and without two ranks disagreeing about what was formed.
    The env:// rendezvous gives every rank the launcher's TCP store; each attempt (`want`, then gloo) runs
    init_process_group(timeout=...) plus one probe all-reduce in a thread the caller waits for with a bound, every rank posts
    how it went in the store, and a backend is used only if EVERY rank reports it working.  If none is, the store itself
    carries the barrier and the rows (backend "store").  `device` (a torch.device) is the rank's GPU for nccl.
    Raises CommError when not even the store can be reached."""
    import datetime
    import threading
    import torch
    import torch.distributed as dist
    log = log or (lambda s: None)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    try:
        store, _, _ = next(iter(dist.rendezvous("env://", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=init_timeout_s))))
        store.set_timeout(datetime.timedelta(seconds=init_timeout_s + 60))
    except Exception as e:
        raise CommError("rank %d: no rendezvous store at %s:%s (%s)" % (rank, os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"], e))
    ctl = dist.PrefixStore("dcsbench", store)
    attempts = []
    order = [want] + (["gloo"] if want != "gloo" else [])
    for i, backend in enumerate(order):
        result = {}

        def work(backend=backend, i=i):
            try:
                _sabotage(backend, rank)
                kw = dict(backend=backend, store=dist.PrefixStore("pg%d" % i, store), rank=rank, world_size=world,
                          timeout=datetime.timedelta(seconds=init_timeout_s))
                if backend == "nccl":
                    kw["device_id"] = device
                dist.init_process_group(**kw)
                # (one collective now, so that a broken set-up shows here and not inside the timed region)
                probe = torch.ones(1, device=device if backend == "nccl" else "cpu")
                dist.all_reduce(probe)
                if backend == "nccl":
                    torch.cuda.synchronize(device)
                result["status"] = "ok" if int(probe.item()) == world else "error: probe all-reduce gave %r" % probe.item()
            except Exception as e:
                result["status"] = "error: %s" % (str(e).splitlines() or [type(e).__name__])[0][:200]

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(init_timeout_s + 15)
        mine = result.get("status", "hung")
        ctl.set("vote%d_r%d" % (i, rank), mine)
        try:
            votes = [ctl.get("vote%d_r%d" % (i, r)).decode() for r in range(world)]
        except Exception as e:
            raise CommError("rank %d: the other ranks did not report on backend %s (%s)" % (rank, backend, e))
        attempts.append((backend, votes))
        if all(v == "ok" for v in votes):
            return Comm(backend, rank, world, device=device, store=ctl, timeout_s=init_timeout_s, attempts=attempts)
        log("process group over %s not formed on every rank: %s" % (backend, votes))
        # (whenever a group came to exist on this rank -- also when its probe all-reduce then failed: a group left behind would
        # make the next backend's init_process_group fail here and keep the other ranks waiting for this one's vote)
        try:
            if mine == "ok" or dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        if any(v == "hung" for v in votes):
            break                       # a thread is stuck inside init_process_group somewhere: no second attempt
    log("barrier and max over the rendezvous store")
    return Comm("store", rank, world, store=ctl, timeout_s=init_timeout_s, attempts=attempts)
