"""The reference's own test -- `DCSExplorer --vol=220 --autoplay --silent --terse --validate=<log>` (DCSDecoder/Tests/test-all.bat:59,
DCSExplorer.cpp:1029-1566) -- with DCSDecoderHIP as the decoder under test.  tests/cpp/dcs_validate_driver.cpp is the shape of that
caller: two decoders booted alike, autoplay sends every type-1 track's two bytes to both, 240 samples from each per frame compared
sample by sample, their bytes to the host compared per frame, differing frames logged in the reference's layout, the report at the
end.  The reference validates against DCSDecoderEmulated, which needs the ROM's ADSP code; here the reference decoder is the
reference's unmodified DCSDecoderNative (oracle/Makefile, target validate).  Expected reports (both decoders native):
tests/golden/validate_golden.json."""
import json
import os
import sys

import numpy as np
import pytest

import dcsexplorer_amd as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_validate_golden as V                # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "validate_golden.json")))
NAMES = [n for n, _ in V.CASES]
BUILD = dict(V.CASES)


def need():
    if not os.path.exists(V.EXE):
        pytest.skip("oracle/_ref/dcs_validate_hip not built (needs /root/reference; `make -C oracle validate`)")


def same_but_for_the_name(text):
    return text.replace("MI355X HIP batch decoder", "Universal native decoder")


def test_goldens_say_what_a_faultless_run_prints():
    assert set(GOLD) == set(NAMES) and len(NAMES) == 9
    for name, g in GOLD.items():
        assert "Result:            Validation Succeeded" in g["report"] and "tracks tested" in g["report"]
        assert "--- Frame" not in g["log"]


def test_native_against_native_reproduces_the_goldens(tmp_path):
    need()
    for name in (NAMES[0], NAMES[6]):
        rep, log = V.run(BUILD[name](), name, str(tmp_path / name), ["--native"])
        assert rep == GOLD[name]["report"] and log == GOLD[name]["log"]


def test_a_differing_frame_is_logged_in_the_references_layout(tmp_path):
    """one sample changed in frame 300 of the decoder under test: the run fails, and the block the log holds for that frame is
    the block dcs_frame_diff (the C ABI's writer of it) formats from the same samples"""
    need()
    name = "seq-os95"
    rep, log = V.run(BUILD[name](), name, str(tmp_path), ["--native", "--flip", "300"])
    assert "Result:            Validation Failed" in rep
    assert "Total number of non-matching PCM samples: 1" in rep and "Number of frames containing differences:  1" in rep
    start = log.index("--- Frame 300 - 1 sample differences ---")
    block = log[start:log.index("\n\n", start) + 2]
    rows = [l for l in block.splitlines()[1:] if l.strip()]
    mine = np.array([int(x) for l in rows for x in l.split("|")[0].split()], dtype=np.int16)
    theirs = np.array([int(x) for l in rows for x in l.split("|")[1].split()], dtype=np.int16)
    n, text = D.frame_diff(300, mine, theirs)
    assert n == 1 and text == block
    assert log[:start].rstrip().splitlines()[-1].startswith("Recent commands: Frame ")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_decoder_validates_against_the_reference_decoder(tmp_path, name):
    """HardBoot / StartSelfTests (the bong included) / SetDefaultVolume, then autoplay over every type-1 track: no PCM sample and
    no data-port byte differs from the unmodified DCSDecoderNative in lock step, the run takes the same tracks and frames and ends
    the same way (some sets end in DecoderFatalError) as when both decoders are the reference's"""
    need()
    rep, log = V.run(BUILD[name](), name, str(tmp_path))
    assert "Decoder tested:    MI355X HIP batch decoder" in rep
    assert "Result:            Validation Succeeded" in rep, log[-3000:]
    assert same_but_for_the_name(rep) == GOLD[name]["report"]
    assert same_but_for_the_name(log) == GOLD[name]["log"]
