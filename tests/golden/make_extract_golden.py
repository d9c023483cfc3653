#!/usr/bin/env python3
"""Generates tests/golden/extract_golden.json: what the reference's extraction and listing loops write
(`DCSExplorer --extract-streams[=raw]`, `--extract-tracks`, `--streams`; DCSExplorer.cpp:1628-1939, :696-770) on the synthetic ROM
sets of tests/romkit.py, as produced by oracle/_ref/dcs_extract_native -- tests/cpp/dcs_extract_driver.cpp, the shape of that
caller, over the UNMODIFIED DCSDecoderNative.  Per case and mode: the text the loop prints and the SHA-256 of every file it
writes.  Build container only (`make -C oracle extract` first); the output is committed."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_rom_golden as R                                  # noqa: E402
import make_seq_golden as S                                  # noqa: E402

NATIVE = os.path.join(ROOT, "oracle", "_ref", "dcs_extract_native")
HIP = os.path.join(ROOT, "oracle", "_ref", "dcs_extract_hip")

# (case name, builder, modes): the ROM-ingestion sets that carry version-detection code (sane streams: every mode), and
# the sequencer sets (programs that use every opcode, loops, deferred tracks) for the track loop -- one of their tracks
# plays a "stream" that is no stream at all, which the reference's stream modes do not survive on every OS version
CASES = [(c[0], (lambda c=c: R.build(c)), ("list", "raw", "wav", "tracks")) for c in R.CASES[:4]] + \
        [(c[0], (lambda c=c: S.build(c)), ("tracks",)) for c in S.CASES]


def run_driver(exe, mode, rs, workdir, timeout=1200):
    """-> (log text, {file name: bytes}) of one run of a driver over the ROM set"""
    os.makedirs(workdir, exist_ok=True)
    args = []
    for chip, image in sorted(rs.images.items()):
        path = os.path.join(workdir, "u%d.rom" % chip)
        with open(path, "wb") as f:
            f.write(image)
        args.append("%d=%s" % (chip, path))
    out = os.path.join(workdir, mode)
    os.makedirs(out, exist_ok=True)
    r = subprocess.run([exe, mode, os.path.join(out, "x")] + args, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("%s %s: exit %d: %s" % (os.path.basename(exe), mode, r.returncode, r.stderr[-1000:]))
    files = {}
    for name in sorted(os.listdir(out)):
        with open(os.path.join(out, name), "rb") as f:
            files[name] = f.read()
    log = files.pop("x.log").decode()
    return log, files


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, build, modes in CASES:
            rs = build()
            for mode in modes:
                log, files = run_driver(NATIVE, mode, rs, os.path.join(tmp, name))
                out["%s/%s" % (name, mode)] = dict(log=log, files={k: hashlib.sha256(v).hexdigest() for k, v in files.items()},
                                                  bytes=sum(len(v) for v in files.values()))
                print(name, mode, len(files), "files,", sum(len(v) for v in files.values()), "bytes")
    with open(os.path.join(ROOT, "tests", "golden", "extract_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
