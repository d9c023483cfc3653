"""Output formats of the reference's extraction / validation modes: the WAV header of
ExtractToWAV (DCSExplorer.cpp:1686-1699), the "DCSa" raw container (:1831-1866, DCSExplorer/README.md:274-289,
reader DCSEncoder.cpp:358-400) and the --validate frame log (:1424-1447).  Byte layouts are checked
against the field tables of those places; the WAV is also read back with the standard library."""
import ctypes
import io
import struct
import wave

import numpy as np
import pytest

import dcsexplorer_amd as D
from util import ALL_FORMATS, make_stream, os_for


@pytest.mark.parametrize("nframes", [0, 1, 66, 65535 + 2])
def test_wav_header_fields(nframes):
    h = D.wav_header(nframes)
    data = nframes * 240 * 2
    assert len(h) == 44
    assert h[0:4] == b"RIFF" and h[8:16] == b"WAVEfmt " and h[36:40] == b"data"
    riff, = struct.unpack_from("<I", h, 4)
    fmtlen, kind, chans, rate, bps, align, bits = struct.unpack_from("<IHHIIHH", h, 16)
    datalen, = struct.unpack_from("<I", h, 40)
    assert (riff, fmtlen, kind, chans, rate, bps, align, bits, datalen) == \
           (data + 44 - 8, 16, 1, 1, 31250, 62500, 2, 16, data)


def test_wav_file_reads_back(tmp_path):
    pcm = (np.arange(5 * 240, dtype=np.int32) * 37 - 20000).astype(np.int16).reshape(5, 240)
    path = tmp_path / "x.wav"
    L = D.load_library()
    assert L.dcs_write_wav(str(path).encode(), pcm.ctypes.data_as(ctypes.c_void_p), 5) == 0
    raw = path.read_bytes()
    assert raw[:44] == D.wav_header(5) and len(raw) == 44 + 5 * 480
    with wave.open(io.BytesIO(raw)) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 31250, 1200)
        assert np.array_equal(np.frombuffer(w.readframes(1200), dtype="<i2"), pcm.reshape(-1))


@pytest.mark.parametrize("os_,ver", [(D.OS93A, 0x9301), (D.OS93B, 0x9302), (D.OS94, 0x9400), (D.OS95, 0x9400)])
def test_dcsa_container_round_trip(os_, ver, tmp_path):
    fmt = [f for f in ALL_FORMATS if os_for(f) == os_ or os_for(f, 1) == os_][0]
    s = make_stream(fmt, 9, seed=31 + os_)
    _, info = D.index_stream(os_, s)
    h = D.dcsa_header(os_, info.nBytes)
    assert len(h) == 36 and h[0:4] == b"DCSa"
    assert struct.unpack_from(">HHH", h, 4) == (ver, 1, 0x7A12)
    assert h[10:32] == bytes(22)
    assert struct.unpack_from(">I", h, 32) == (info.nBytes,)
    path = tmp_path / "s.dcs"
    buf = np.frombuffer(s + bytes(8), dtype=np.uint8)
    assert D.load_library().dcs_write_dcsa(str(path).encode(), os_, buf.ctypes.data_as(ctypes.c_void_p), info.nBytes) == 0
    raw = path.read_bytes()
    assert raw[:36] == h and len(raw) == 36 + info.nBytes
    got_os, got = D.dcsa_parse(raw)
    assert got_os == (D.OS94 if os_ in (D.OS94, D.OS95) else os_)
    assert got == (s + bytes(8))[:info.nBytes]
    # the extracted bytes decode to the same stream: the index pass sees the same frames
    idx_a, _ = D.index_stream(os_, s)
    idx_b, _ = D.index_stream(os_, got + bytes(8))
    assert idx_a.tobytes() == idx_b.tobytes()


def test_dcsa_reader_rejects_what_the_reference_rejects():
    good = D.dcsa_header(D.OS94, 4) + b"\0\1\2\3"
    D.dcsa_parse(good)
    for pos, val in ((0, 0x58), (4, 0x95), (7, 2), (8, 0x7B)):
        bad = bytearray(good); bad[pos] = val
        with pytest.raises(D.DcsError):
            D.dcsa_parse(bytes(bad))
    with pytest.raises(D.DcsError):
        D.dcsa_parse(good[:-1])             # data section shorter than its size field
    with pytest.raises(D.DcsError):
        D.dcsa_parse(good[:20])


def test_validate_frame_log_block():
    a = np.arange(240, dtype=np.int16) - 120
    b = a.copy()
    assert D.frame_diff(7, a, b) == (0, "")
    b[3] = 999; b[239] = -32768
    n, text = D.frame_diff(123456789012, a, b)
    lines = text.split("\n")
    assert n == 2 and lines[0] == "--- Frame 123456789012 - 2 sample differences ---"
    assert len(lines) == 1 + 15 + 2 and lines[-1] == "" and lines[-2] == ""
    first = lines[1]
    left, right = first.split(" | ")
    assert left == " ".join("%6d" % v for v in a[:16]) and right == " ".join("%6d" % v for v in b[:16])
    assert lines[15].endswith("%6d" % -32768)
