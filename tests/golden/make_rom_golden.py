#!/usr/bin/env python3
"""Generates tests/golden/rom_golden.json from the UNMODIFIED reference (oracle/_ref/libdcsref.so): for each
synthetic ROM set of tests/romkit.py, the text dump of everything DCSDecoder derives from the images
(oracle/ref_driver.cpp:ref_rom_dump), and the reference's PCM hash of the --extract-streams loop over the
set's streams.  Build container only; the output is committed.  A fixture is data: the ROM-set recipe
(constructor arguments of romkit.RomSet) + the reference's answers."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dcsexplorer_amd as D                                  # noqa: E402
from oracle.dcs_oracle import Reference, fnv1a64             # noqa: E402
import romkit                                                # noqa: E402

# (name, hw, os, catalog offset, seed, plant the version-detection code?, corrupt seed or None)
CASES = [
    ("dcs93-os93a", romkit.HW93, D.OS93A, 0x3000, 11, True, None),
    ("dcs93-os93b", romkit.HW93, D.OS93B, 0x3000, 12, True, None),
    ("dcs93-os94", romkit.HW93, D.OS94, 0x4000, 13, True, None),
    ("dcs95-os95", romkit.HW95, D.OS95, 0x6000, 14, True, None),
    ("dcs95-os95-nocode", romkit.HW95, D.OS95, 0x6000, 15, False, None),
    ("dcs93-os93a-forced", romkit.HW93, D.OS93A, 0x4000, 16, False, None),
    ("dcs93-os94-damaged-programs", romkit.HW93, D.OS94, 0x4000, 17, True, 99),
    ("dcs95-os95-bad-checksum", romkit.HW95, D.OS95, 0x6000, 18, True, -1),
]


def build(case):
    name, hw, os_, cat, seed, code, damage = case
    rs = romkit.RomSet(hw, os_, cat, seed, version_code=code)
    if damage is not None:
        rs = romkit.damage(rs, damage)
    return rs


def ref_dump(ref, rs, force_hw, force_os):
    roms = (ctypes.c_char_p * 8)(*[rs.images.get(c) for c in range(2, 10)])
    sizes = (ctypes.c_size_t * 8)(*[len(rs.images.get(c, b"")) for c in range(2, 10)])
    buf = ctypes.create_string_buffer(1 << 22)
    f = ref.lib.ref_rom_dump
    f.restype = ctypes.c_size_t
    n = f(roms, sizes, force_hw, force_os, buf, len(buf))
    assert n < len(buf)
    return buf.value.decode()


def main():
    ref = Reference()
    out = {}
    for case in CASES:
        name, hw, os_, cat, seed, code, damage = case
        rs = build(case)
        force = (hw, os_) if not code else (-1, -1)
        dump = ref_dump(ref, rs, *force)
        # the reference's own extraction loop over the planned streams
        plan = [l.split() for l in dump.splitlines() if l.startswith("extract ")]
        addrs = [int(p[3].split("=")[1], 16) for p in plan]
        levels = [int(p[4].split("=")[1]) for p in plan]
        streams = [rs.streams[a] + bytes(64) for a in addrs if a in rs.streams]
        h = None
        if len(streams) == len(addrs) and streams:
            pcm = ref.decode_sequence(os_, 255, streams, levels, 2)
            h = "%016x" % fnv1a64(pcm.tobytes())
        out[name] = dict(hw=hw, os=os_, catalog=cat, seed=seed, version_code=code, damage=damage,
                         force=list(force), dump=dump, extract_pcm_fnv1a64=h)
        print(name, len(dump.splitlines()), "lines,", len(plan), "streams, pcm hash", h)
    with open(os.path.join(ROOT, "tests", "golden", "rom_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
