"""The reference's extraction and listing loops through DCSDecoderHIP, behind the reference's REAL base class (VERDICT r2
item 1).  tests/cpp/dcs_extract_driver.cpp is the shape of the caller (`DCSExplorer --extract-streams[=raw]`,
`--extract-tracks`, `--streams`; DCSExplorer.cpp:1628-1939, :696-770) as a template over the decoder class; oracle/Makefile
(target extract) builds it over the unmodified DCSDecoderNative (dcs_extract_native: the expected files) and over
DCSDecoderHIP with -DDCSHIP_USE_REFERENCE_BASE (dcs_extract_hip: the files under test).  Expected results are committed as
tests/golden/extract_golden.json (make_extract_golden.py: log text + SHA-256 of every file)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import dcsexplorer_amd as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_extract_golden as X                 # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "extract_golden.json")))
PAIRS = [(name, mode) for name, _, modes in X.CASES for mode in modes]
BUILD = {name: build for name, build, _ in X.CASES}


def need(exe):
    if not os.path.exists(exe):
        pytest.skip("%s not built (needs /root/reference; `make -C oracle extract`)" % os.path.relpath(exe, ROOT))


def test_goldens_cover_every_os_version_and_mode():
    """four OS versions x (list, raw, wav, tracks) on the ROM-ingestion sets, five sequencer sets x tracks"""
    assert len(PAIRS) == 4 * 4 + 5 and set(GOLD) == {"%s/%s" % p for p in PAIRS}
    for key, g in GOLD.items():
        if not key.endswith("/list"):
            assert g["files"], key
        assert "decoder " in g["log"]


@pytest.mark.parametrize("name,mode", PAIRS[::4], ids=["%s-%s" % p for p in PAIRS[::4]])
def test_native_driver_reproduces_the_goldens(tmp_path, name, mode):
    """(build container) the driver over the unmodified DCSDecoderNative still writes what the goldens record"""
    need(X.NATIVE)
    log, files = X.run_driver(X.NATIVE, mode, BUILD[name](), str(tmp_path))
    g = GOLD["%s/%s" % (name, mode)]
    assert log == g["log"]
    assert {k: hashlib.sha256(v).hexdigest() for k, v in files.items()} == g["files"]


@pytest.mark.parametrize("name", [n for n, _, m in X.CASES if "tracks" in m])
def test_tracks_plan_names_the_references_files(name):
    """dcs_romset_extract_tracks_plan (host only): the tracks the reference's loop wrote a WAV for, and frame counts whose WAV
    sizes add up to the bytes the reference wrote"""
    rs = D.RomSet(images=BUILD[name]().images)
    rs.check()
    plan = rs.extract_tracks_plan()
    g = GOLD["%s/tracks" % name]
    assert ["x_%04x.wav" % t for t, _ in plan] == sorted(g["files"])
    assert sum(44 + 480 * n for _, n in plan) == g["bytes"]
    for t, n in plan:
        ti = rs.track_info(t)
        assert ti.type == 1 and n == ((ti.time & 0xFFFF) + 2) & 0xFFFF


def test_hip_driver_fails_loudly_without_a_gpu(tmp_path):
    """no GPU: the class behind the real base ends SoftBoot in an error state and every sample is silence -- never the
    reference's PCM from some other path"""
    if D.device_count() > 0:
        pytest.skip("a GPU is present")
    need(X.HIP)
    log, files = X.run_driver(X.HIP, "wav", BUILD["dcs95-os95"](), str(tmp_path))
    assert "decoder in error state" in log
    g = GOLD["dcs95-os95/wav"]
    assert sorted(files) == sorted(g["files"])
    assert all(not np.frombuffer(v[44:], dtype=np.int16).any() for v in files.values())


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", PAIRS, ids=["%s-%s" % p for p in PAIRS])
def test_hip_decoder_writes_what_the_reference_writes(tmp_path, name, mode):
    """every file of --extract-streams (wav and raw) and --extract-tracks, and the --streams table, byte for byte: against
    the committed hashes, and -- where the native driver is there too -- against its files directly"""
    need(X.HIP)
    rs = BUILD[name]()
    log, files = X.run_driver(X.HIP, mode, rs, str(tmp_path / "hip"))
    g = GOLD["%s/%s" % (name, mode)]
    assert log == g["log"]
    assert {k: hashlib.sha256(v).hexdigest() for k, v in files.items()} == g["files"]
    if os.path.exists(X.NATIVE):
        wlog, want = X.run_driver(X.NATIVE, mode, rs, str(tmp_path / "native"))
        assert wlog == log and sorted(want) == sorted(files)
        for k in want:
            assert want[k] == files[k], "%s differs" % k


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n, _, m in X.CASES if "tracks" in m])
def test_batch_entry_decodes_all_tracks_in_one_launch(gpu_ctx, tmp_path, name):
    """dcs_romset_extract_tracks_plan + dcs_extract_tracks: the whole --extract-tracks loop planned ahead on the host
    sequencer and decoded in ONE launch gives the PCM of the reference's track WAV files (their hashes are committed)"""
    rs_py = BUILD[name]()
    rs = D.RomSet()
    for chip, image in sorted(rs_py.images.items()):
        rs.add_rom(chip, image)
    rs.check()
    plan = rs.extract_tracks_plan()
    g = GOLD["%s/tracks" % name]
    assert ["x_%04x.wav" % t for t, _ in plan] == sorted(g["files"])
    pcm, first = gpu_ctx.extract_tracks(rs, plan)
    assert int(first[-1]) == sum(n for _, n in plan) == pcm.shape[0]
    for k, (t, n) in enumerate(plan):
        hdr = np.zeros(44, dtype=np.uint8)
        D.load_library().dcs_wav_header(n, hdr.ctypes.data)
        wav = hdr.tobytes() + pcm[int(first[k]):int(first[k + 1])].tobytes()
        assert hashlib.sha256(wav).hexdigest() == g["files"]["x_%04x.wav" % t], "track %04x" % t
