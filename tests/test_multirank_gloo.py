"""The N>1 path on CPU: two processes over gloo.  Each rank takes its range of the seeded stream corpus
(no data-path collective exists on this path); the ranks agree on a max-over-ranks time and their ranges
are disjoint and together equal the corpus."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from dcsexplorer_amd import sharding
dist.init_process_group(backend="gloo")
rank, local_rank, world = sharding.rank_info()
streams = sharding.rank_streams("dcs93_4096", rank)
mine = [hashlib.sha1(s[1]).hexdigest() for s in streams]
frames = sum((s[1][0] << 8) | s[1][1] for s in streams)
dist.barrier()
t = sharding.max_over_ranks(1.0 + rank)            # rank 1 is "slower"
gathered = [None] * world
dist.all_gather_object(gathered, (rank, mine, frames))
lo, hi = sharding.partition_range(10, world, rank)
if rank == 0:
    print(json.dumps(dict(world=world, tmax=t, ranks=gathered, part=[lo, hi])))
dist.destroy_process_group()
''' % ROOT


def test_two_ranks_partition_the_corpus(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world"] == 2 and d["tmax"] == 2.0
    (r0, s0, f0), (r1, s1, f1) = sorted(d["ranks"])
    assert f0 == f1 == 4096                            # weak scaling: same shape per rank
    assert len(set(s0)) == 64 and len(set(s1)) == 64 and not (set(s0) & set(s1))
    # together they are the first 128 streams of the corpus
    sys.path.insert(0, ROOT)
    from dcsexplorer_amd import workloads
    corpus = [hashlib.sha1(s[1]).hexdigest() for s in workloads.streams_dcs93_4096(n_streams=128)]
    assert s0 + s1 == corpus


def test_partition_range_covers_everything():
    sys.path.insert(0, ROOT)
    from dcsexplorer_amd.sharding import partition_range
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [partition_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
