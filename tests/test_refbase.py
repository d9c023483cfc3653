"""DCSDecoderHIP behind the reference's REAL base class.  oracle/_ref/dcs_refbase_test is the
class compiled with -DDCSHIP_USE_REFERENCE_BASE against /root/reference/DCSDecoder/DCSDecoder.h and linked with the
reference's own DCSDecoder.cpp (oracle/Makefile, target refbase; build container only, the binary travels).  Its driver
takes the decoder out of the real DCSDecoder::GetRegistrationMap() and touches it through a DCSDecoder* only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import dcsexplorer_amd as D
import romkit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_seq_golden as G                     # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "dcs_refbase_test")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "seq_golden.json")))


def run_driver(tmp_path, case, script):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/dcs_refbase_test not built (needs /root/reference; `make -C oracle refbase`)")
    n, ev = romkit.SCRIPTS[script]
    rs = G.build(case)
    args = []
    for chip, image in sorted(rs.images.items()):
        path = tmp_path / ("u%d.rom" % chip)
        path.write_bytes(image)
        args.append("%d=%s" % (chip, path))
    evf = tmp_path / "events.txt"
    evf.write_text("".join("%d %d %d\n" % e for e in sorted(ev, key=lambda x: x[0])))
    prefix = str(tmp_path / "out")
    r = subprocess.run([EXE, str(G.VOLUME), str(n), str(evf), prefix] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    info = dict(l.split("=", 1) for l in open(prefix + ".info").read().splitlines() if "=" in l)
    pcm = np.fromfile(prefix + ".pcm", dtype=np.int16).reshape(n, 240)
    host = [[int(x) for x in l.split()] for l in open(prefix + ".host").read().splitlines()]
    return info, pcm, host


def test_real_base_build_fails_loudly_without_a_gpu(tmp_path):
    """on a box without a GPU the object still comes out of the real registration map and takes the ROMs through the
    base class, and SoftBoot ends in InitializationError with the cause named: no silent CPU path"""
    if D.device_count() > 0:
        pytest.skip("a GPU is present: the loud-failure path is the CPU container's")
    info, pcm, host = run_driver(tmp_path, G.CASES[3], "port-only")
    assert info["name"] == "MI355X HIP batch decoder"
    assert info["post"] == "1"                                  # the base's own CheckROMs accepted the images
    assert info["ok"] == "0" and "HIP decoder unavailable" in info["error"]
    assert not pcm.any() and host == []


@pytest.mark.gpu
@pytest.mark.parametrize("case", G.CASES, ids=[c[0] for c in G.CASES])
def test_through_a_plain_base_pointer_equals_reference(tmp_path, case):
    """ROMs in through DCSDecoder::AddROM, commands in through DCSDecoder::WriteDataPort, samples out through
    DCSDecoder::GetNextSample: PCM hash, host bytes (with their ticks) and final state equal what the unmodified
    DCSDecoderNative did on the same ROMs and events (tests/golden/seq_golden.json, made by ref_seq_run)"""
    from oracle.dcs_oracle import Oracle, Reference, reference_available
    info, pcm, host = run_driver(tmp_path, case, "port-only")
    gold = GOLD["%s/port-only" % case[0]]
    assert info["post"] == "1" and info["running"] == "1"
    assert (info["ok"] == "0") == gold["fatal"]
    assert host == gold["host_bytes"]
    assert "%016x" % Oracle().fnv1a64(pcm) == gold["pcm_fnv1a64"]
    if reference_available():
        n, ev = romkit.SCRIPTS["port-only"]
        want, hb, _ = G.ref_run(Reference(), G.build(case), G.VOLUME, n, ev)
        bad = np.nonzero((pcm != want).any(axis=1))[0]
        assert bad.size == 0, "first differing ticks: %s" % bad[:8]


ENC = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_golden.json")))["cases"]


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead", [0, 1, 64])
@pytest.mark.parametrize("case", ENC, ids=[c["name"] for c in ENC])
def test_rom_less_recipe_behind_the_real_base(tmp_path, case, lookahead):
    """InitStandalone / SetDefaultVolume / SoftBoot / LoadAudioStream(0, ROMPointer(0, p), level) / GetNextSample, as
    DCSEncoder.cpp:522-571 and EncoderTester.cpp:85-137 drive the native decoder, on the 24 streams the reference's own
    encoder made (six layouts, all four OS versions; volume / level 255 / 0x64, 220 / 0x7F, 0x67 / 0x64, 240 / 0x50): the
    PCM the unmodified DCSDecoderNative produced for the same calls (tests/golden/encoder_golden.json), whether the class
    decodes tick by tick or 64 ticks per launch"""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/dcs_refbase_test not built (needs /root/reference; `make -C oracle refbase`)")
    from oracle.dcs_oracle import Oracle
    z = np.load(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz"))
    sf = tmp_path / "s.bin"
    sf.write_bytes(z[case["name"] + "/stream"].tobytes())
    out = tmp_path / "o.pcm"
    n = case["frames_out"]
    r = subprocess.run([EXE, "standalone", str(case["os"]), str(case["volume"]), str(case["levels"][0]), str(lookahead), str(n), str(sf), str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    pcm = np.fromfile(out, dtype=np.int16).reshape(n, 240)
    assert "%016x" % Oracle().fnv1a64(pcm) == case["pcm_fnv1a64"]
    if case["name"] + "/pcm" in z:
        assert np.array_equal(pcm, z[case["name"] + "/pcm"])


FUZZ = os.path.join(ROOT, "oracle", "_ref", "dcs_class_fuzz")


@pytest.mark.gpu
@pytest.mark.parametrize("os_", [D.OS93A, D.OS93B, D.OS94, D.OS95], ids=["os93a", "os93b", "os94", "os95"])
def test_a_random_caller_in_lock_step_with_the_reference(tmp_path, os_):
    """tests/cpp/dcs_class_fuzz.cpp: DCSDecoderHIP behind the reference's real base class and the reference's unmodified DCSDecoderNative
    in one process, the same seeded random calls on both -- LoadAudioStream on any channel at any level, ClearTracks, SetMasterVolume,
    IsStreamPlaying, pulls of 0 to 2 500 frames in between -- every sample and every answer compared.  Streams of 3 to 1 800 frames
    (the long ones are walked on a second thread while their first frames go out), two of them damaged; the caller never mentions
    look-ahead, and the same seeds once more tick by tick."""
    if not os.path.exists(FUZZ):
        pytest.skip("oracle/_ref/dcs_class_fuzz not built (needs /root/reference; `make -C oracle fuzz`)")
    from util import make_stream, corrupt, splitmix
    fmts = [f for f in range(6) if os_ in (D.format_os(f), D.format_os(f, prefer_95=True), D.format_os(f, prefer_93a=True))]
    g = splitmix(0xF022 + os_)
    paths, clean = [], []
    for k, n in enumerate([3, 17, 64, 65, 200, 385, 700, 1800, 40, 500]):
        data = make_stream(fmts[next(g) % len(fmts)], n, seed=0xF0220 + 16 * os_ + k, profile=(0, 1, 2, 5)[next(g) % 4])
        path = tmp_path / ("c%d.bin" % k)
        path.write_bytes(data)
        clean.append(str(path))
        if k >= 8:
            data = corrupt(data, seed=70 + k)                   # (frame errors: the channel stops on the next tick)
        path = tmp_path / ("f%d.bin" % k)
        path.write_bytes(data)
        paths.append(str(path))
    damaged_runs = 0
    for seed, n_ops, lookahead in ((1, 300, -1), (2, 300, -1), (3, 120, 1), (4, 300, 0), (5, 200, 37)):
        args = [str(os_), str(1000 * os_ + seed), str(n_ops), str(lookahead)]
        # the reference has undefined behaviour on some damaged streams (unchecked table indices: it may crash): a seed whose calls
        # kill the REFERENCE ALONE runs on the undamaged streams instead
        alone = subprocess.run([FUZZ] + args + paths, capture_output=True, text=True, timeout=900, env=dict(os.environ, DCS_FUZZ_REF_ONLY="1"))
        use = paths if alone.returncode == 0 else clean
        damaged_runs += use is paths
        r = subprocess.run([FUZZ] + args + use, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), "seed %d look-ahead %d: %s %s" % (seed, lookahead, r.stdout[-500:], r.stderr[-500:])
