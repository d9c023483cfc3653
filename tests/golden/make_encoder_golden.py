#!/usr/bin/env python3
"""Generates tests/golden/encoder_golden.npz + encoder_golden.json: streams made by the reference's OWN encoder
(DCSEncoder::OpenStream / WriteStream / CloseStream, DCSEncoder.cpp:520-579 is its round trip) from a deterministic
signal, in every layout it can write (1994+ Type 0, Type 1 sub-type 0 and 3; OS93b Type 0 and 1; OS93a Type 0 -- it
cannot write OS93a Type 1, DCSEncoder.cpp:2485-2527), and the PCM the UNMODIFIED reference decoder
(oracle/_ref/libdcsref.so) produces for them.  Real audio visits band-type and zero-band statistics the seeded
stream writer does not necessarily visit (missing #5).

Build container only.  The encoder is compiled from where it lies under /root/reference with g++, a force-included
header for five MSVC CRT names (encoder/enc_shim.h) and a pass-through for the four libsamplerate calls
(encoder/enc_resample_stub.c: the signal is generated at 31 250 Hz, ratio 1; the vendored libsamplerate lacks
high_qual_coeffs.h).  The encoder is only a SOURCE of stream bytes: a fixture is those bytes plus the reference
decoder's PCM.  Outputs are data and travel to the GPU box."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
HERE = os.path.join(ROOT, "tests", "golden", "encoder")
EXE = os.path.join(ROOT, "oracle", "_ref", "dcs_ref_encoder")

from oracle.dcs_oracle import Oracle, Reference              # noqa: E402

# (name, encoder format version, type, sub-type, OS version the decoder is told)
LAYOUTS = [("94-T0", "9400", 0, 0, 2), ("94-T1s0", "9400", 1, 0, 3), ("94-T1s3", "9400", 1, 3, 3),
           ("93b-T0", "9302", 0, 0, 1), ("93b-T1", "9302", 1, 0, 1), ("93a-T0", "9301", 0, 0, 0)]
VARIANTS = 4
FRAMES = 256


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        stub = os.path.join(tmp, "stub.o")
        subprocess.check_call(["gcc", "-O2", "-w", "-I%s/libsamplerate/src" % REF, "-c", os.path.join(HERE, "enc_resample_stub.c"), "-o", stub])
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-include", os.path.join(HERE, "enc_shim.h"),
                               "-I%s/DCSEncoder" % REF, "-I%s/libsamplerate/src" % REF, "-o", EXE,
                               os.path.join(HERE, "enc_driver.cpp"), "%s/DCSEncoder/DCSEncoder.cpp" % REF,
                               "%s/DCSDecoder/DCSDecoder.cpp" % REF, "%s/DCSDecoder/DCSDecoderNative.cpp" % REF, stub, "-lpthread"])


def workload_streams(arrays):
    """realistic_65536: 256 streams x 256 frames out of the 24 encoder-made streams, each replica at its own volume and
    mixing level (dcsexplorer_amd/workloads.py builds the same list from the committed arrays)"""
    names = ["ENC-%s-v%d" % (l[0], v) for v in range(VARIANTS) for l in LAYOUTS]
    os_of = {l[0]: l[4] for l in LAYOUTS}
    out = []
    for k in range(256):
        name = names[k % len(names)]
        lay = name[4:name.rindex("-v")]
        os_ = os_of[lay]
        if lay.startswith("94") and (k & 1):
            os_ = 2 if os_ == 3 else 3              # OS94 and OS95 share the codec
        out.append((os_, arrays[name + "/stream"].tobytes(), 200 + (k % 56), 0x60 + (k % 16)))
    return out


def main():
    build()
    ref, orc = Reference(), Oracle()
    arrays, meta = {}, []
    with tempfile.TemporaryDirectory() as tmp:
        for v in range(VARIANTS):
            for name, fv, typ, sub, os_ in LAYOUTS:
                path = os.path.join(tmp, "s.bin")
                subprocess.check_call([EXE, fv, str(typ), str(sub), str(FRAMES), path, str(v)], stderr=subprocess.DEVNULL)
                s = open(path, "rb").read()
                assert ((s[0] << 8) | s[1]) == FRAMES
                key = "ENC-%s-v%d" % (name, v)
                arrays[key + "/stream"] = np.frombuffer(s, dtype=np.uint8)
                vol, lvl = [(255, 0x64), (220, 0x7F), (0x67, 0x64), (240, 0x50)][v]
                pcm = ref.decode(os_, vol, [s], [lvl], FRAMES + 2)
                info = ref.stream_info(os_, s)
                assert info["formatType"] == typ and info["nFrames"] == FRAMES
                m = dict(name=key, os=os_, volume=vol, levels=[lvl], streams=1, frames_out=FRAMES + 2, bytes=len(s),
                         bytes_per_frame=round(len(s) / FRAMES, 1), pcm_fnv1a64="%016x" % orc.fnv1a64(pcm))
                if v == 0:
                    arrays[key + "/pcm"] = pcm          # full PCM for one recording, hashes for the others
                meta.append(m)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz"), **arrays)
    streams = workload_streams(arrays)
    hashes = ["%016x" % orc.fnv1a64(ref.decode(o, vol, [s], [lvl], FRAMES)) for o, s, vol, lvl in streams]
    out = dict(cases=meta, workloads=dict(realistic_65536=dict(
        streams=len(streams), frames=len(streams) * FRAMES, bytes_per_frame=round(sum(len(s[1]) for s in streams) / (len(streams) * FRAMES), 1),
        fnv1a64_of_stream_hashes="%016x" % orc.fnv1a64(np.array([int(h, 16) for h in hashes], dtype=np.uint64)), stream_hashes=hashes)))
    with open(os.path.join(ROOT, "tests", "golden", "encoder_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    for m in meta:
        print(m["name"], m["bytes"], "bytes,", m["bytes_per_frame"], "B/frame")
    print("realistic_65536:", out["workloads"]["realistic_65536"]["bytes_per_frame"], "B/frame")


if __name__ == "__main__":
    main()
