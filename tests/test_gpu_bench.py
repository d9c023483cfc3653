"""bench.py on the GPU box: the N-rank code with the one GPU there is (RCCL process group of one), and a line that survives
its optional sections.  The loop being sharded is /root/reference/DCSExplorer/DCSExplorer.cpp:1628-1907."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, env=None, timeout=900):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None), len(lines)


def test_rank_path_over_rccl_with_one_rank():
    """RANK=0 WORLD_SIZE=1 --force-dist: init_process_group(backend="nccl", device_id=...), the probe all-reduce, the NUMA bind,
    barrier / max / rows as device-side all-reduces around the timed region, and end_to_end over the ranks -- everything of
    `--gpus N` that one GPU can execute, on RCCL"""
    r, d, n = _bench(["--force-dist", "--steps", "5", "--warmup", "2", "--e2e-device-depth", "8", "--e2e-lists", "16"],
                     {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29611"})
    assert r.returncode == 0 and n == 1, r.stderr[-3000:]
    assert d["dist"]["backend"] == "nccl" and d["dist"]["attempts"] == [{"backend": "nccl", "votes": ["ok"]}], d["dist"]
    assert d["n_gpus"] == 1 and d["bit_exact"] is True and d["bit_exact_ranks"] == [True]
    assert d["value"] > 1e10 and d["roofline"]["frac"] > 0.05
    e = d["end_to_end"]
    assert e["ranks_failed"] == [] and e["sustained"]["ranks"] == 1 and e["sustained"]["value"] > 1e9
    assert e["per_rank"][0]["cpu_ms_per_list"] > 0


def test_line_survives_a_failing_and_a_hanging_section():
    """an exception inside device_full_path and a second_workload that never returns: `value`, `roofline` and `bit_exact` are in
    the one line all the same, the sections carry {"error": ...}, what follows the abandoned section is skipped"""
    r, d, n = _bench(["--steps", "5", "--warmup", "2", "--no-end-to-end", "--no-cpu-baseline"], {"DCS_BENCH_INJECT_FAIL": "device_full_path"})
    assert r.returncode == 0 and n == 1, r.stderr[-3000:]
    assert d["value"] > 1e10 and d["roofline"]["frac"] > 0.05 and d["bit_exact"] is True
    assert "injected failure in device_full_path" in d["device_full_path"]["error"]
    assert d["second_workload"]["bit_exact"] is True and d["roofline_cold"]["frac"] > 0.05      # the sections around it ran

    r, d, n = _bench(["--steps", "5", "--warmup", "2", "--no-cpu-baseline"], {"DCS_BENCH_INJECT_HANG": "third_workload", "DCS_BENCH_SECTION_BUDGET_S": "20"})
    assert r.returncode == 0 and n == 1, r.stderr[-3000:]
    assert d["value"] > 1e10 and d["roofline"]["frac"] > 0.05 and d["bit_exact"] is True
    assert d["second_workload"]["bit_exact"] is True
    assert d["third_workload"]["error"].startswith("timeout")
    assert all("skipped" in d[k]["error"] for k in ("roofline_cold", "device_full_path", "end_to_end"))


def test_a_rank_whose_pipeline_fails_still_reaches_the_exchanges():
    """two ranks on the one GPU (gloo): rank 1's end_to_end fails, rank 0's figures are reported, the failed rank is named, the
    headline is untouched"""
    r, d, n = _bench(["--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1", "--e2e-device-depth", "8", "--e2e-lists", "16"],
                     {"DCS_BENCH_INJECT_FAIL": "end_to_end_ranks@1"})
    assert r.returncode == 0 and n == 1, r.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["bit_exact_ranks"] == [True, True] and d["value"] > 1e9
    e = d["end_to_end"]
    assert e["ranks_failed"] == [1] and e["sustained"]["ranks"] == 1 and [p["rank"] for p in e["per_rank"]] == [0]
