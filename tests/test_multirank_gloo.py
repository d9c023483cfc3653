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


CORPUS_WORKER = r'''
import hashlib, json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from dcsexplorer_amd import sharding, workloads
dist.init_process_group(backend="gloo")
rank, local_rank, world = sharding.rank_info()
manifest = workloads.corpus_manifest(titles=8, streams_per_title=6, max_frames=300)
lo, hi = sharding.rank_corpus(manifest, world, rank)
streams = workloads.corpus_streams(manifest, lo, hi)
mine = [hashlib.sha1(s[1]).hexdigest() for s in streams]
frames = sum((s[1][0] << 8) | s[1][1] for s in streams)
gathered = [None] * world
dist.all_gather_object(gathered, (rank, lo, hi, mine, frames))
if rank == 0:
    print(json.dumps(dict(world=world, ranks=gathered)))
dist.destroy_process_group()
''' % ROOT


def test_two_ranks_cut_the_ragged_corpus_by_frames(tmp_path):
    """BASELINE configs[4] shape: ONE ragged corpus, contiguous stream ranges balanced by total frame count
    (SURVEY 8e); the ranges are disjoint, their union is the corpus, no rank is more than one stream off"""
    script = tmp_path / "corpus_worker.py"
    script.write_text(CORPUS_WORKER)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    (r0, lo0, hi0, s0, f0), (r1, lo1, hi1, s1, f1) = sorted(d["ranks"])
    sys.path.insert(0, ROOT)
    from dcsexplorer_amd import workloads
    manifest = workloads.corpus_manifest(titles=8, streams_per_title=6, max_frames=300)
    whole = workloads.corpus_streams(manifest)
    assert lo0 == 0 and hi0 == lo1 and hi1 == len(manifest)           # disjoint, contiguous, complete
    assert s0 + s1 == [hashlib.sha1(s[1]).hexdigest() for s in whole]
    counts = workloads.corpus_frames(manifest)
    assert f0 == int(counts[lo0:hi0].sum()) and f1 == int(counts[lo1:hi1].sum())
    assert f0 + f1 == int(counts.sum())
    assert abs(f0 - f1) <= int(counts.max())                         # at most one stream's length apart
    assert len(set(m["os"] for m in manifest)) == 4                   # titles of all four OS generations


def test_frame_balanced_partition_properties():
    """dcs_partition_streams: contiguous, complete, every range within one (longest) stream of the ideal share"""
    sys.path.insert(0, ROOT)
    import numpy as np
    import dcsexplorer_amd as D
    rng = np.random.default_rng(5)
    for n in (0, 1, 3, 17, 580, 5000):
        counts = rng.integers(20, 2001, size=n).astype(np.uint32)
        for world in (1, 2, 3, 4, 8):
            cut = D.partition_streams(counts, world)
            assert cut[0] == 0 and cut[-1] == n and all(cut[i] <= cut[i + 1] for i in range(world))
            if n == 0:
                continue
            loads = [int(counts[cut[r]:cut[r + 1]].sum()) for r in range(world)]
            assert sum(loads) == int(counts.sum())
            ideal = counts.sum() / world
            assert max(abs(l - ideal) for l in loads) <= counts.max()
    # skewed: one huge stream among small ones still gives contiguous, complete ranges
    counts = np.array([10] * 50 + [60000] + [10] * 50, dtype=np.uint32)
    cut = D.partition_streams(counts, 4)
    assert cut[0] == 0 and cut[-1] == 101 and sum(int(counts[cut[r]:cut[r + 1]].sum()) for r in range(4)) == counts.sum()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without RANK in the environment starts 2 ranks itself; on a box without GPUs the
    --rehearse path (gloo) runs launcher, partition, host planning, barrier and max-over-ranks and reports n_gpus 2"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for wl, scaling in (("corpus", "strong"), ("dcs93_4096", "weak")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1",
                              "--workload", wl, "--corpus-titles", "8", "--corpus-streams", "4"],
                             capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] is None and "rehearsal" in d
        assert len(d["config"]["frames_per_rank"]) == 2 and all(f > 0 for f in d["config"]["frames_per_rank"])


def _bench(extra_args, extra_env=None, timeout=900):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None), len(lines)


def test_bench_rehearses_eight_ranks():
    """the driver's --gpus 8 shape on CPU: launcher, eight ranks, rendezvous store, the vote on the process group, every rank's
    share, barrier, max over ranks, ONE line from rank 0"""
    out, d, n = _bench(["--gpus", "8", "--rehearse", "--steps", "1", "--workload", "dcs93_4096"])
    assert out.returncode == 0 and n == 1, out.stderr[-2000:]
    assert d["n_gpus"] == 8 and d["config"]["frames_per_rank"] == [4096] * 8
    assert d["dist"]["backend"] == "gloo" and d["dist"]["attempts"][0]["votes"] == ["ok"] * 8


def test_bench_ranks_fall_back_to_the_rendezvous_store():
    """no process group on every rank (here: gloo sabotaged on rank 1 only, so rank 0's attempt runs into its timeout) -> all
    ranks agree on the store as carrier of barrier and max, the line is printed, the attempts are in it"""
    out, d, n = _bench(["--gpus", "2", "--rehearse", "--steps", "1", "--workload", "dcs93_4096", "--dist-timeout", "10"],
                       {"DCS_COMM_SABOTAGE": "gloo@1"})
    assert out.returncode == 0 and n == 1, out.stderr[-2000:]
    assert d["dist"]["backend"] == "store" and d["n_gpus"] == 2 and d["config"]["frames_per_rank"] == [4096, 4096]
    votes = d["dist"]["attempts"][0]["votes"]
    assert votes[1].startswith("error: sabotaged") and votes[0].startswith("error")


def test_bench_a_rank_stuck_in_init_does_not_hang_the_job():
    """init_process_group never returns on rank 1: both ranks see "hung" in the vote, make no second attempt and carry on over
    the store (no silent hang)"""
    out, d, n = _bench(["--gpus", "2", "--rehearse", "--steps", "1", "--workload", "dcs93_4096", "--dist-timeout", "5"],
                       {"DCS_COMM_SABOTAGE": "gloo:hang@1"}, timeout=300)
    assert out.returncode == 0 and n == 1, out.stderr[-2000:]
    assert d["dist"]["backend"] == "store" and d["dist"]["attempts"][0]["votes"][1] == "hung" and len(d["dist"]["attempts"]) == 1


def test_bench_a_rank_that_fails_in_setup_yields_an_error_line_not_a_hang():
    out, d, n = _bench(["--gpus", "2", "--rehearse", "--steps", "1", "--workload", "dcs93_4096", "--dist-timeout", "20"],
                       {"DCS_BENCH_INJECT_FAIL": "setup_rank1"}, timeout=300)
    assert out.returncode != 0 and n == 1
    assert d["value"] is None and d["n_gpus"] == 2 and "rank(s) [1]" in d["error"]
    assert "injected failure in setup_rank1" in d["ranks_failed"]["rank1"]


def test_sections_guard_exceptions_and_timeouts(monkeypatch):
    """bench.Sections: an exception or a section that never returns leaves {"error": ...}; what follows an abandoned section is
    skipped (it may still hold the GPU); the total budget is kept"""
    sys.path.insert(0, ROOT)
    import time
    import bench
    sec = bench.Sections(30.0)
    assert sec.run("a", lambda: {"x": 1}, 5) == {"x": 1}
    r = sec.run("b", lambda: 1 / 0, 5)
    assert r["error"].startswith("ZeroDivisionError")
    monkeypatch.setenv("DCS_BENCH_INJECT_FAIL", "c")
    assert "injected failure in c" in sec.run("c", lambda: 1, 5)["error"]
    t0 = time.perf_counter()
    r = sec.run("d", lambda: time.sleep(60), 1.5)
    assert r["error"].startswith("timeout") and time.perf_counter() - t0 < 5 and sec.abandoned == "d"
    assert "skipped" in sec.run("e", lambda: 1, 5)["error"]
    sec2 = bench.Sections(0.5)
    assert "budget is spent" in sec2.run("f", lambda: 1, 5)["error"]
