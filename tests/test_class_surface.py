"""DCSDecoderHIP (include/DCSDecoderHIP.h): the reference's decoder class surface on top of the C ABI,
driven by a C++ client written like the reference's own callers (tests/cpp/dcs_class_test.cpp)."""
import os
import subprocess

import numpy as np
import pytest

import dcsexplorer_amd as D
from util import make_stream, os_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "dcsexplorer_amd", "dcs_class_test")


def run(args, tmp_path):
    out = str(tmp_path / "out.pcm")
    full = [EXE] + [a if a != "OUT" else out for a in args]
    p = subprocess.run(full, capture_output=True, text=True, timeout=300)
    return p, out


def write_stream(tmp_path, name, data):
    path = str(tmp_path / name)
    open(path, "wb").write(data)
    return path


def test_class_fails_loudly_without_gpu(tmp_path):
    """no device: SoftBoot ends in InitializationError, IsOK() is false, nothing is decoded on the CPU"""
    assert os.path.exists(EXE), "build with make -C dcsexplorer_amd/csrc"
    if D.device_count() > 0:
        pytest.skip("a GPU is present")
    s = write_stream(tmp_path, "s.bin", make_stream(D.FMT_94_T1_S3, 4, seed=1))
    p, _ = run(["live", "3", "255", "1", "OUT", "4", "100", s], tmp_path)
    assert p.returncode == 4 and "HIP decoder unavailable" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead", [0, 1, 7])
def test_live_playback_matches_oracle(tmp_path, oracle, lookahead):
    """look-ahead 0 = the decoder's own (what a caller gets who never calls SetLookahead), 1 = tick by tick, 7 = fixed"""
    for fmt in (D.FMT_94_T1_S3, D.FMT_93_T0, D.FMT_93A_T1):
        data = make_stream(fmt, 23, seed=21000 + fmt)
        os_ = os_for(fmt)
        s = write_stream(tmp_path, "s%d.bin" % fmt, data)
        p, out = run(["live", str(os_), "230", str(lookahead), "OUT", "26", str(0x66), s], tmp_path)
        assert p.returncode == 0, p.stderr
        got = np.fromfile(out, dtype=np.int16).reshape(26, 240)
        want = oracle.decode(os_, 230, [data], [0x66], 26)
        assert np.array_equal(got, want), "fmt %d" % fmt
        # IsStreamPlaying after every frame is the state behind THAT frame, however far the decoder has run ahead: the
        # stream's last frame ends it (DCSDecoderNative.cpp:1377, :1575-1589)
        assert open(out + ".playing").read().split() == ["1"] * 22 + ["0"] * 4


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [D.FMT_94_T1_S3, D.FMT_93B_T1])
def test_long_stream_at_the_default_lookahead(tmp_path, oracle, fmt):
    """a caller that pulls samples in a bare loop and never heard of SetLookahead (DCSEncoder.cpp:565-567): the decoder runs
    64, 512, 4 096 ticks ahead on its own and stops two ticks behind the stream's end; 700 frames cross every snapshot and
    refill boundary"""
    data = make_stream(fmt, 700, seed=21500 + fmt, profile=5)
    os_ = os_for(fmt)
    s = write_stream(tmp_path, "long%d.bin" % fmt, data)
    p, out = run(["live", str(os_), "255", "0", "OUT", "705", str(0x64), s], tmp_path)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, dtype=np.int16).reshape(705, 240)
    want = oracle.decode(os_, 255, [data], [0x64], 705)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "first differing frames: %s" % bad[:8]
    assert open(out + ".playing").read().split() == ["1"] * 699 + ["0"] * 6


@pytest.mark.gpu
def test_live_multichannel_matches_oracle(tmp_path, oracle):
    datas = [make_stream(f, 9 + 6 * i, seed=22000 + i, profile=i % 3) for i, f in
             enumerate((D.FMT_94_T0, D.FMT_94_T1_S0, D.FMT_94_T1_S3, D.FMT_94_T1_S3))]
    levels = [0x68, 0x60, 0x64, 0x5C]
    want = oracle.decode(2, 255, datas, levels, 40)
    for lookahead in (5, 0, 1):
        args = ["live", "2", "255", str(lookahead), "OUT", "40"]
        for i, d in enumerate(datas):
            args += [str(levels[i]), write_stream(tmp_path, "m%d.bin" % i, d)]
        p, out = run(args, tmp_path)
        assert p.returncode == 0, p.stderr
        got = np.fromfile(out, dtype=np.int16).reshape(40, 240)
        assert np.array_equal(got, want), "look-ahead %d" % lookahead
        # channel c plays 9 + 6 c frames
        rows = open(out + ".playing").read().split()
        assert rows == ["".join("1" if f + 1 < 9 + 6 * c else "0" for c in range(4)) for f in range(40)]


@pytest.mark.gpu
def test_batch_submit_matches_oracle(tmp_path, oracle):
    datas = [make_stream(D.FMT_93B_T1, 12 + 3 * i, seed=23000 + i) for i in range(5)]
    args = ["batch", "1", "240", "2", "OUT", str(0x64)] + [write_stream(tmp_path, "b%d.bin" % i, d) for i, d in enumerate(datas)]
    p, out = run(args, tmp_path)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, dtype=np.int16).reshape(-1, 240)
    want = np.concatenate([oracle.decode(1, 240, [d], [0x64], ((d[0] << 8) | d[1]) + 2) for d in datas])
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead", [0, 1, 16])
def test_extract_streams_loop_matches_oracle(tmp_path, oracle, lookahead):
    """the --extract-streams loop on ONE DCSDecoderHIP object: decoder state carries over between streams
    (incl. a stream stopped by an error, whose mixer level is reset), WAV files as ExtractToWAV writes them"""
    from util import corrupt
    datas, levels = [], [0x64, 0x7F, 0x30, 0x64, 0x55]
    for i in range(5):
        d = make_stream(D.FMT_94_T1_S3 if i % 2 else D.FMT_94_T0, 6 + 4 * i, seed=24000 + i, profile=i % 4)
        if i == 2:
            d = corrupt(d, seed=5)
        datas.append(d)
    args = ["extract", "3", "255", str(lookahead), "OUT", "0"]
    for i, d in enumerate(datas):
        args += [str(levels[i]), write_stream(tmp_path, "e%d.bin" % i, d)]
    p, out = run(args, tmp_path)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, dtype=np.int16).reshape(-1, 240)
    want = oracle.decode_sequence(3, 255, datas, levels, 2)
    assert np.array_equal(got, want)
    pos = 0
    for k, d in enumerate(datas):
        n = ((d[0] << 8) | d[1]) + 2
        raw = open("%s.%d.wav" % (out, k), "rb").read()
        assert raw[:44] == D.wav_header(n) and raw[44:] == want[pos:pos + n].tobytes()
        pos += n


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead", [0, 1, 9, 100])
@pytest.mark.parametrize("script", ["main", "fatal-opcode", "volume-and-clear"])
def test_rom_mode_script_through_the_class(tmp_path, lookahead, script):
    """DCSDecoderHIP driven like DCSExplorer drives a decoder: AddROM, CheckROMs, SoftBoot, WriteDataPort /
    AddTrackCommand while pulling samples.  PCM, the bytes sent to the host (with their tick) and the fatal
    state equal the reference decoder's, whatever the look-ahead."""
    import json, sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_seq_golden as G
    import romkit
    from oracle.dcs_oracle import fnv1a64
    case = G.CASES[3]                                   # DCS-95 board, OS95
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "seq_golden.json")))["%s/%s" % (case[0], script)]
    rs = G.build(case)
    n, ev = romkit.SCRIPTS[script]
    sf = tmp_path / "script.txt"
    sf.write_text("".join("%d %d %d\n" % e for e in sorted(ev, key=lambda x: x[0])))
    roms = [write_stream(tmp_path, "u%d.rom" % c, rs.images[c]) for c in (2, 3, 4)]
    p, out = run(["script", "3", str(G.VOLUME), str(lookahead), "OUT", str(n), str(sf)] + roms, tmp_path)
    assert p.returncode == 0, p.stderr
    assert "CheckROMs 1, version 0104" in p.stderr
    pcm = np.fromfile(out, dtype=np.int16)
    lines = open(out + ".host").read().split("\n")
    host = [[int(x) for x in l.split()] for l in lines if l and not l.startswith("fatal")]
    assert host == gold["host_bytes"]
    assert ("fatal 1" in lines) == gold["fatal"]
    assert "%016x" % fnv1a64(pcm.tobytes()) == gold["pcm_fnv1a64"]
