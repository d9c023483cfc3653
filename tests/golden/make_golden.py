#!/usr/bin/env python3
"""Generates tests/golden/dcs_golden.npz + dcs_golden_hashes.json from the UNMODIFIED reference decoder
compiled into oracle/_ref/libdcsref.so (oracle/Makefile, target `ref`).  Runs only in the build
container (needs /root/reference for that build); the outputs are committed and travel to the GPU box.

A fixture is data: the input stream bytes + the reference's int16 PCM for it.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dcsexplorer_amd as D                                  # noqa: E402
from dcsexplorer_amd import workloads                        # noqa: E402
from oracle.dcs_oracle import Reference, fnv1a64             # noqa: E402
from util import ALL_FORMATS, FORMAT_NAMES, make_stream, os_for   # noqa: E402

# Known-answer streams of SURVEY.md Appendix D (hand-assembled, 2 frames + 4 pad bytes each) with the
# sample values listed there: frame 0 [0..7], [16..19], [236..239].
KATS = [
    ("KAT-94-T0", 2, 255, 0x7F, "0002202060ffffffffffffffffffffffffff504a82541214283c5064788df7ede3d9cfc5bbb0060c12181e242a31c2c547ca4ccf51febd7c3af9b87735c0c1824303c485460000000000",
     [-316, -1844, -4240, -7404, -9857, -10835, -10346, -8691], [-27736, -29307, -28701, -25670], [-32768, -32768, 7261, -32768]),
    ("KAT-94-T1s0", 2, 255, 0x64, "0002a02060ffffffffffffffffffffffffff504a82541214283c5064788df7ede3d9cfc5bbb0060c12181e242a31c2c547ca4ccf51febd7c3af9b87735c0c1824303c485460000000000",
     [-264, -1519, -3309, -5253, -6961, -8070, -8372, -7896], [-7632, -6965, -5472, -3165], [-32768, -32768, -32768, -32768]),
    ("KAT-94-T1s3", 3, 220, 0x64, "0002a0a060ffffffffffffffffffffffffff504a82541214283c5064788df7ede3d9cfc5bbb0060c12181e242a31c2c547ca4ccf51febd7c3af9b87735c0c1824303c485460000000000",
     [-136, -781, -1702, -2702, -3579, -4152, -4305, -4066], [-3926, -3581, -2813, -1627], [-28879, -28575, -27628, -26046]),
    ("KAT-93b-T0", 1, 255, 0x7F, "00022020ffffffffffffffffffffffffffffdc1c3854708ca8c4e0fd1935516d89a5c0ff9f1e9e1d9d1c9c1b9b1a9a199918981b84078b0e9215991ca023a72aae31b5389ff3e3d3c3b3a39383736353433323130000000000",
     [-90, 11, 1693, 5948, -9906, -11640, -17524, 17399], [-9234, -8710, 27783, -5371], [-14334, 9784, -32093, -2286]),
    ("KAT-93a-T0", 0, 255, 0x7F, "00022020ffffffffffffffffffffffffffffdc1c3854708ca8c4e0fd1935516d89a5c0ff9f1e9e1d9d1c9c1b9b1a9a199918981b84078b0e9215991ca023a72aae31b5389ff3e3d3c3b3a39383736353433323130000000000",
     [-90, 11, 1693, 5948, -9906, -11640, -17524, 17399], [-9234, -8710, 27783, -5371], [-14334, 9784, -32093, -2286]),
    ("KAT-93b-T1", 1, 255, 0x64, "0002a020ffffffffffffffffffffffffffff4cbc2081840a18388122858c1a3878997fdf5dfa71ddaf465c57ee59ad4e823f7f000417efe00082fdfc00105040407f040407f040407f040407f000000000",
     [84, 1118, -4192, -4320, -2753, -247, 653, -3608], [9693, 6570, 10387, 20343], [27592, -19487, 3031, 27993]),
    ("KAT-93a-T1", 0, 255, 0x7F, "0002838ab678a32d9d2000000000",
     [1, 8, 19, 36, 56, 80, 107, 133], [202, 201, 201, 200], [18, 21, 25, 30]),
]


def main():
    ref = Reference()
    arrays = {}
    meta = []

    for name, os_, vol, lvl, hexs, head, mid, tail in KATS:
        s = bytes.fromhex(hexs)
        pcm = ref.decode(os_, vol, [s], [lvl], 4)
        assert list(pcm[0, :8]) == head and list(pcm[0, 16:20]) == mid and list(pcm[0, 236:240]) == tail, name
        arrays[name + "/stream"] = np.frombuffer(s, dtype=np.uint8)
        arrays[name + "/pcm"] = pcm
        meta.append(dict(name=name, os=os_, volume=vol, levels=[lvl], streams=1, frames_out=4,
                         survey_head=head, survey_mid=mid, survey_tail=tail))

    # one synthetic stream per unpack layout and profile, full PCM incl. two taper frames
    settings = [(255, 0x64), (220, 0x7F), (0x67, 0x64), (255, 0x20)]
    for fmt in ALL_FORMATS:
        for profile in range(4):
            nf = 24
            s = make_stream(fmt, nf, seed=0x601D0000 + fmt * 16 + profile, profile=profile,
                            stride_from=16 if profile != 1 else 9)
            os_ = os_for(fmt, profile)
            vol, lvl = settings[profile]
            pcm = ref.decode(os_, vol, [s], [lvl], nf + 2)
            name = "SYN-%s-p%d" % (FORMAT_NAMES[fmt], profile)
            arrays[name + "/stream"] = np.frombuffer(s, dtype=np.uint8)
            arrays[name + "/pcm"] = pcm
            meta.append(dict(name=name, os=os_, volume=vol, levels=[lvl], streams=1, frames_out=nf + 2))

    # BASELINE.json config 1 as SURVEY.md 8(d) words it: ONE OS93a Type-0 stream, 64 frames, seed 0x93010001, volume 255, level 0x64,
    # all of its PCM from the compiled reference (the "bit-exact vs DCSDecoderEmu" half of that config needs ROM images: SURVEY 8c)
    s = make_stream(D.FMT_93_T0, 64, seed=0x93010001, profile=6, nbands=12)
    pcm = ref.decode(D.OS93A, 255, [s], [0x64], 64 + 2)
    arrays["CONFIG-1/stream"] = np.frombuffer(s, dtype=np.uint8)
    arrays["CONFIG-1/pcm"] = pcm
    meta.append(dict(name="CONFIG-1", os=D.OS93A, volume=255, levels=[0x64], streams=1, frames_out=66))

    # multi-channel mixes (frequency-domain mixing of several streams before one transform)
    for fam, fmts, nch in ((2, [3, 4, 5, 5], 4), (1, [0, 1, 0], 3), (0, [0, 2], 2), (3, [5] * 8, 8)):
        streams = [make_stream(fmts[c], 12 + 4 * c, seed=0x3C0000 + fam * 64 + c, profile=c % 3) for c in range(nch)]
        levels = [0x6A - 4 * c for c in range(nch)]
        nout = 12 + 4 * nch + 2
        pcm = ref.decode(fam, 240, streams, levels, nout)
        name = "MIX-os%d-%dch" % (fam, nch)
        for c, s in enumerate(streams):
            arrays["%s/stream%d" % (name, c)] = np.frombuffer(s, dtype=np.uint8)
        arrays[name + "/pcm"] = pcm
        meta.append(dict(name=name, os=fam, volume=240, levels=levels, streams=nch, frames_out=nout))

    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dcs_golden.npz"), **arrays)

    # hashes of the full-size seeded workloads (each stream decoded alone by the reference)
    hashes = {}
    for wl in ("dcs93_4096", "dcs94_65536", "mixed_16384", "survey3_65536"):
        streams = workloads.WORKLOADS[wl]()
        h = 0xcbf29ce484222325
        per_stream = []
        for os_, s, vol, lvl in streams:
            nf = (s[0] << 8) | s[1]
            pcm = ref.decode(os_, vol, [s], [lvl], nf)
            per_stream.append(fnv1a64(pcm.tobytes()))
        # checksum of checksums, in stream order
        blob = np.array(per_stream, dtype=np.uint64).tobytes()
        hashes[wl] = dict(streams=len(streams), frames=int(sum((s[0] << 8) | s[1] for _, s, _, _ in streams)),
                          fnv1a64_of_stream_hashes="%016x" % fnv1a64(blob),
                          first_stream_hash="%016x" % per_stream[0], last_stream_hash="%016x" % per_stream[-1],
                          stream_hashes=["%016x" % x for x in per_stream])
    with open(os.path.join(ROOT, "tests", "golden", "dcs_golden_hashes.json"), "w") as f:
        json.dump(dict(cases=meta, workloads=hashes), f, indent=1)
    print("wrote %d arrays, %d cases, %d workload hashes" % (len(arrays), len(meta), len(hashes)))


if __name__ == "__main__":
    main()
