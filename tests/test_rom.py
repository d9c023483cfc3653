"""ROM ingestion (csrc/dcs_rom.cpp): catalog, version detection, ROM addressing, track programs,
stream lists and the zip loader, on the synthetic ROM sets of romkit.py.  The expected answers are the
UNMODIFIED reference's (tests/golden/rom_golden.json, made by tests/golden/make_rom_golden.py; compared live
as well where oracle/_ref is present).  The zip member-recognition heuristics have no compiled twin here (the
reference's zip loader needs <Windows.h>): they are tested against the rules stated in
DCSDecoderZipLoader.cpp:106-203 only -- parity unpinned for that one function."""
import json
import os
import sys

import numpy as np
import pytest

import dcsexplorer_amd as D
import romkit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_rom_golden as G                     # noqa: E402  (the case list and builders; no reference needed to import)

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "rom_golden.json")))
IDS = [c[0] for c in G.CASES]


@pytest.mark.parametrize("case", G.CASES, ids=IDS)
def test_rom_set_dump_equals_reference_golden(case):
    gold = GOLD[case[0]]
    rs = D.RomSet(images=G.build(case).images)
    assert rs.dump(*gold["force"]) == gold["dump"]


@pytest.mark.parametrize("case", G.CASES, ids=IDS)
def test_rom_set_dump_equals_reference_live(case, reference):
    rs_img = G.build(case)
    force = GOLD[case[0]]["force"]
    assert D.RomSet(images=rs_img.images).dump(*force) == G.ref_dump(reference, rs_img, *force)


def test_detection_results_are_the_expected_ones():
    for case in G.CASES:
        name, hw, os_, cat, seed, code, damage = case
        c = D.RomSet(images=G.build(case).images).check()
        assert c.catalogOffset == cat and c.hw == hw
        assert c.status == (3 if damage == -1 else 1)               # U3 fails its checksum in that case
        if code:
            assert c.os == os_
            assert c.nominalVersion == (0x0104 if hw == romkit.HW95 else 0)
        assert c.signature.decode().startswith("Synthetic Pinball")


def test_pointers_stay_inside_their_images():
    rs_img = G.build(G.CASES[3])
    rs = D.RomSet(images=rs_img.images)
    rs.check()
    for addr, data in rs_img.streams.items():
        assert rs.stream_bytes(addr)[:len(data)] == data
    # an unpopulated chip reads as 8 KB of 0xFF, offsets wrap inside the image
    chip, _, avail = rs.pointer((7 << 21) | 0x12345)
    assert chip == 9 and avail == 0x2000 - (0x12345 & 0x1FFF)
    assert rs.stream_bytes((7 << 21) | 5)[:4] == b"\xff\xff\xff\xff"
    assert rs.track_info(10 ** 6) is None and len(rs.decompile(3)) == 0


@pytest.mark.parametrize("compress", [True, False])
def test_zip_loader_recognises_the_chips(tmp_path, compress):
    rs_img = G.build(G.CASES[2])
    want = D.RomSet(images=rs_img.images).dump()
    junk = {"readme2.txt": b"not a rom, although the name has a 2",
            "notes_3.txt": b"S4 wrong digit in the signature 01/01/94\0",
            "dir/": b""}
    z = rs_img.zip_bytes(extra=junk, compress=compress)
    assert D.RomSet(zip_bytes=z).dump() == want
    path = tmp_path / "synth.zip"
    path.write_bytes(z)
    assert D.RomSet(zip_path=path).dump() == want
    # U2 named explicitly when the heuristics cannot find it (no '2' in its name)
    z2 = rs_img.zip_bytes(names={2: "boot.rom", 3: "s3.rom", 4: "s4.rom"})
    with pytest.raises(D.DcsError):
        D.RomSet(zip_bytes=z2)
    assert D.RomSet(zip_bytes=z2, explicit_u2="BOOT.ROM").dump() == want
    with pytest.raises(D.DcsError):
        D.RomSet(zip_bytes=z[:-30])                                 # truncated archive


def test_zip_loader_cactus_canyon_u7_quirk():
    rs_img = G.build(G.CASES[3])
    u7 = bytearray(b"\xFF" * 0x80000)
    sig = b"S6 Cactus Canyon mislabelled 09/08/98\0"
    u7[:len(sig)] = sig
    rs_img.images = dict(rs_img.images); rs_img.images[7] = bytes(u7)
    names = {2: "cc_u2.rom", 3: "cc_s3.rom", 4: "cc_s4.rom", 7: "cc_s7.rom"}
    size_of_u7 = lambda rs: (rs.check(), rs.pointer(5 << 21)[2])[1]
    assert size_of_u7(D.RomSet(zip_bytes=rs_img.zip_bytes(names=names), zip_name="cc_13.zip")) == 0x80000
    assert size_of_u7(D.RomSet(zip_bytes=rs_img.zip_bytes(names=names), zip_name="afm_113.zip")) == 0x2000


@pytest.mark.gpu
@pytest.mark.parametrize("case", G.CASES, ids=IDS)
def test_extract_streams_from_rom_set(gpu_ctx, oracle, case):
    """ROM images -> plan -> one launch: the PCM of the reference's --extract-streams loop (hash committed
    with the goldens) and of the oracle, stream by stream"""
    from oracle.dcs_oracle import fnv1a64
    gold = GOLD[case[0]]
    rs_img = G.build(case)
    rs = D.RomSet(images=rs_img.images)
    rs.check()
    if gold["force"][0] >= 0:
        rs.set_version(*gold["force"])
    items, pcm, first = gpu_ctx.extract_streams(rs, volume=255)
    streams = [rs_img.streams[int(a)] + bytes(64) for a in items["address"]]
    want = oracle.decode_sequence(case[2], 255, streams, [int(l) for l in items["level"]], 2)
    assert np.array_equal(pcm, want)
    assert "%016x" % fnv1a64(pcm.tobytes()) == gold["extract_pcm_fnv1a64"]
    assert first[-1] == pcm.shape[0]
