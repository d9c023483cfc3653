"""the C-ABI library loads and exports every symbol include/dcs_hip.h declares (no GPU needed)"""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "dcs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dcs_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(dcs):
    L = dcs.load_library()
    names = declared_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(L, n), "libdcs_hip.so does not export %s" % n
    from dcsexplorer_amd.api import EXPORTS
    assert set(EXPORTS) == set(names)


def test_abi_version_and_struct_sizes(dcs):
    hdr = open(os.path.join(ROOT, "include", "dcs_hip.h")).read()
    declared = int(re.search(r"#define\s+DCS_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert dcs.load_library().dcs_abi_version() == declared == dcs.api.ABI_VERSION == 9
    assert dcs.SRC_DTYPE.itemsize == 160
    assert dcs.JOB_DTYPE.itemsize == 16
    assert dcs.INDEX_DTYPE.itemsize == 148


def test_loading_the_library_leaves_the_environment_untouched():
    """no load-time constructor edits the host's environment.  A fresh process loads libdcs_hip.so with
    GPU_MAX_HW_QUEUES unset and compares os.environ / the C environ before and after; the explicit dcs_runtime_defaults() then
    sets the variable, never overwrites one, and DCS_NO_RUNTIME_DEFAULTS keeps the library's first HIP call from making it"""
    import subprocess
    import sys
    code = r'''
import ctypes, os, sys
libc = ctypes.CDLL(None)
libc.getenv.restype = ctypes.c_char_p
def c_environ():
    env = ctypes.POINTER(ctypes.c_char_p).in_dll(libc, "environ")
    out, i = [], 0
    while env[i]:
        out.append(env[i]); i += 1
    # (GLOG_*: exported by librocprofiler-register when ANY library that carries a HIP code object registers it with the runtime
    #  at load -- the HIP runtime's doing for every HIP program, not this library's)
    return sorted(e for e in out if not e.startswith(b"GLOG_"))
before = c_environ()
L = ctypes.CDLL(sys.argv[1])
assert c_environ() == before, "loading the library changed environ"
assert libc.getenv(b"GPU_MAX_HW_QUEUES") is None
if os.environ.get("DCS_NO_RUNTIME_DEFAULTS"):
    L.dcs_device_count()                                   # the library's first call into HIP
    assert libc.getenv(b"GPU_MAX_HW_QUEUES") is None, "opt-out ignored"
    print("optout-ok")
else:
    L.dcs_runtime_defaults.restype = ctypes.c_int
    assert L.dcs_runtime_defaults() == 1 and libc.getenv(b"GPU_MAX_HW_QUEUES") == b"8"
    libc.setenv(b"GPU_MAX_HW_QUEUES", b"4", 1)
    assert L.dcs_runtime_defaults() == 0 and libc.getenv(b"GPU_MAX_HW_QUEUES") == b"4"    # never overwritten
    print("explicit-ok")
'''
    lib = os.path.join(ROOT, "dcsexplorer_amd", "libdcs_hip.so")
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "DCS_NO_RUNTIME_DEFAULTS")}
    r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "explicit-ok" in r.stdout, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True, env=dict(env, DCS_NO_RUNTIME_DEFAULTS="1"), timeout=300)
    assert r.returncode == 0 and "optout-ok" in r.stdout, r.stderr[-2000:]


def test_no_gpu_means_loud_failure_not_fallback(dcs):
    """without a gfx950 device the context constructor must raise; nothing decodes on the CPU"""
    if dcs.device_count() > 0:
        return
    try:
        dcs.Context(0)
    except dcs.DcsError as e:
        assert e.status == -2
    else:
        raise AssertionError("Context() succeeded without a GPU")


def test_product_never_references_oracle():
    """the shipped package must not import, include or link anything under oracle/"""
    pkg = os.path.join(ROOT, "dcsexplorer_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in text and "dcs_oracle" not in text and "libdcsref" not in text, \
                    "%s references the oracle" % os.path.join(dirpath, f)
    for hdr in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        assert "dcs_oracle" not in text


def test_product_reads_no_test_data():
    """the package opens nothing under tests/ (the encoder-made recordings of one workload are handed to it:
    workloads.register_recordings)"""
    pkg = os.path.join(ROOT, "dcsexplorer_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            text = open(os.path.join(pkg, f)).read()
            assert '"tests"' not in text and "tests/golden/" not in text.replace("tests/golden/encoder_golden.npz (made", "").replace(
                "tests/golden/make_encoder_golden.py", "").replace("with tests/golden/encoder_golden.npz", ""), f


def test_counter_profiles_are_tied_to_the_library_build(dcs, tmp_path):
    """bench.py reports committed PMC counters only for the library build they were taken with (dcs_build_id): counters
    of another build -- the kernel edited, tools/prof.sh not run again -- are withheld with a note, never reported stale"""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    bid = dcs.build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", bid)
    json.dump({"traffic_bytes_fetch_x2": 1.0, "lib_build_id": bid, "lib_sha256": "0" * 64}, open(tmp_path / "traffic_same.json", "w"))
    json.dump({"traffic_bytes_fetch_x2": 1.0, "lib_build_id": "f" * 16, "lib_sha256": "0" * 64}, open(tmp_path / "traffic_other.json", "w"))
    json.dump({"traffic_bytes_fetch_x2": 1.0}, open(tmp_path / "traffic_untagged.json", "w"))
    t, note = bench.load_counters("same", str(tmp_path))
    assert t is not None and note is None
    for stale in ("other", "untagged"):
        t, note = bench.load_counters(stale, str(tmp_path))
        assert t is None and "withheld" in note
    assert bench.load_counters("absent", str(tmp_path)) == (None, None)
