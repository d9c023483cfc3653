#!/usr/bin/env python3
"""Generates tests/golden/seq_golden.json from the UNMODIFIED reference (oracle/_ref/libdcsref.so): for the
sequencer ROM sets of tests/romkit.py and each event script, what the real decoder does when driven tick by
tick -- the bytes it sends to the host (with their tick), whether it ends in DecoderFatalError, and the
FNV-1a-64 of its PCM.  Build container only; the output is committed."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dcsexplorer_amd as D                                  # noqa: E402
from oracle.dcs_oracle import Reference, fnv1a64             # noqa: E402
import romkit                                                # noqa: E402

# (name, hw, os, catalog offset, seed, nominal version planted in the DCS-95 code)
CASES = [
    ("seq-os93a", romkit.HW93, D.OS93A, 0x3000, 31, 0),
    ("seq-os93b", romkit.HW93, D.OS93B, 0x3000, 32, 0),
    ("seq-os94", romkit.HW93, D.OS94, 0x4000, 33, 0),
    ("seq-os95", romkit.HW95, D.OS95, 0x6000, 34, 0x0104),
    ("seq-os95-v105", romkit.HW95, D.OS95, 0x6000, 35, 0x0105),      # the channel-5 maximum-level override
]
VOLUME = 0xE0


def build(case):
    name, hw, os_, cat, seed, nominal = case
    return romkit.SeqRomSet(hw, os_, cat, seed, nominal=nominal or 0x0104)


def ref_run(ref, rs, volume, n_ticks, events):
    roms = (ctypes.c_char_p * 8)(*[rs.images.get(c) for c in range(2, 10)])
    sizes = (ctypes.c_size_t * 8)(*[len(rs.images.get(c, b"")) for c in range(2, 10)])
    ev = np.array(sorted(events, key=lambda x: x[0]), dtype=np.uint32).reshape(-1)
    pcm = np.zeros((n_ticks, 240), dtype=np.int16)
    hb = np.zeros((8192, 2), dtype=np.uint32)
    nh, fatal = ctypes.c_int(), ctypes.c_int()
    ref.lib.ref_seq_run(roms, sizes, -1, -1, -1, volume, ev.ctypes.data_as(ctypes.c_void_p), len(events), n_ticks,
                        pcm.ctypes.data_as(ctypes.c_void_p), hb.ctypes.data_as(ctypes.c_void_p), 8192,
                        ctypes.byref(nh), ctypes.byref(fatal))
    assert nh.value <= 8192
    return pcm, [[int(t), int(b)] for t, b in hb[:nh.value]], bool(fatal.value)


def main():
    ref = Reference()
    out = {}
    for case in CASES:
        rs = build(case)
        for sname, (n, ev) in romkit.SCRIPTS.items():
            pcm, hb, fatal = ref_run(ref, rs, VOLUME, n, ev)
            out["%s/%s" % (case[0], sname)] = dict(host_bytes=hb, fatal=fatal, pcm_fnv1a64="%016x" % fnv1a64(pcm.tobytes()),
                                                  nonzero_frames=int((pcm != 0).any(axis=1).sum()))
            print(case[0], sname, "host bytes", len(hb), "fatal", fatal, "nonzero frames", out["%s/%s" % (case[0], sname)]["nonzero_frames"])
    with open(os.path.join(ROOT, "tests", "golden", "seq_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
