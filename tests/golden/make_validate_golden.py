#!/usr/bin/env python3
"""Generates tests/golden/validate_golden.json: the report of the reference's `--autoplay --validate` loop (DCSExplorer.cpp:1029-1566;
the shape of that caller is tests/cpp/dcs_validate_driver.cpp) when BOTH decoders are the reference's unmodified DCSDecoderNative,
on the synthetic ROM sets of tests/romkit.py at the reference's test volume (--vol=220, Tests/test-all.bat:59): which tracks
autoplay starts, how many frames the run takes, how it ends.  A run with DCSDecoderHIP under test must print the same report but for
the decoder's name.  Build container only (`make -C oracle validate` first); the output is committed."""
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

EXE = os.path.join(ROOT, "oracle", "_ref", "dcs_validate_hip")
VOLUME = 220
CASES = [(c[0], (lambda c=c: R.build(c))) for c in R.CASES[:4]] + [(c[0], (lambda c=c: S.build(c))) for c in S.CASES]


def run(rs, name, workdir, options=(), timeout=1200):
    """-> (report text, log text) of one validation run"""
    os.makedirs(workdir, exist_ok=True)
    args = []
    for chip, image in sorted(rs.images.items()):
        path = os.path.join(workdir, "u%d.rom" % chip)
        with open(path, "wb") as f:
            f.write(image)
        args.append("%d=%s" % (chip, path))
    log, rep = os.path.join(workdir, "log.txt"), os.path.join(workdir, "report.txt")
    r = subprocess.run([EXE, str(VOLUME), log, rep] + list(options) + [name] + args, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("dcs_validate_hip %s: exit %d: %s" % (name, r.returncode, r.stderr[-1000:]))
    return open(rep).read(), open(log).read()


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, build in CASES:
            rep, log = run(build(), name, os.path.join(tmp, name), ["--native"])
            assert "Validation Succeeded" in rep
            out[name] = dict(report=rep, log=log)
            print(name, rep.strip().splitlines()[-1])
    with open(os.path.join(ROOT, "tests", "golden", "validate_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
