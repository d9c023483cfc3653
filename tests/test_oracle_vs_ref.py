"""Pins oracle/dcs_oracle.c against the UNMODIFIED reference compiled into oracle/_ref/libdcsref.so.
Runs only where that build exists (this container; it also travels to the GPU box with gpurun)."""
import numpy as np
import pytest

from util import ALL_FORMATS, FORMAT_NAMES, make_stream, os_for, splitmix


@pytest.mark.parametrize("fmt", ALL_FORMATS, ids=[FORMAT_NAMES[f] for f in ALL_FORMATS])
@pytest.mark.parametrize("profile", [0, 1, 2, 3, 4, 5])
def test_whole_decoder_pcm_and_probes(oracle, reference, fmt, profile):
    for k in range(3):
        stream = make_stream(fmt, 48, seed=7000 + 100 * fmt + 10 * profile + k, profile=profile,
                             stride_from=16 if k == 0 else 6 + k)
        os_ = os_for(fmt, k)
        vol, lvl = [(255, 0x64), (220, 0x7F), (0x67, 0x40)][k]
        a, pa = oracle.decode(os_, vol, [stream], [lvl], 51, probes=True)
        b, pb = reference.decode(os_, vol, [stream], [lvl], 51, probes=True)
        assert np.array_equal(a, b)
        for f in range(51):
            assert bytes(pa[f]) == bytes(pb[f]), "probe %d differs" % f
        assert oracle.stream_info(os_, stream) == reference.stream_info(os_, stream)


@pytest.mark.parametrize("fmt", ALL_FORMATS, ids=[FORMAT_NAMES[f] for f in ALL_FORMATS])
def test_decompress_frame_level(oracle, reference, fmt):
    stream = make_stream(fmt, 32, seed=8100 + fmt, profile=3)
    os_ = os_for(fmt)
    for mix in (0x7FFF, 0xFF00, 0x0001, 0x1234):
        a = oracle.decompress(os_, stream, mix, 32)
        b = reference.decompress(os_, stream, mix, 32)
        assert np.array_equal(a[0], b[0])           # frequency-domain words
        assert np.array_equal(a[1], b[1])           # bit offsets
        assert np.array_equal(a[2], b[2])           # band types after each frame
        assert np.array_equal(a[3] & 1, b[3])       # stop flags


@pytest.mark.parametrize("os_", [0, 1, 2, 3])
def test_transform_on_random_spectra(oracle, reference, os_):
    g = splitmix(900 + os_)
    rng = np.random.default_rng(next(g) & 0xFFFFFFFF)
    for trial in range(60):
        kind = trial % 4
        if kind == 0:
            fb = rng.integers(0, 65536, 512)                        # full range: saturation / wrap everywhere
        elif kind == 1:
            fb = rng.integers(-600, 600, 512) & 0xFFFF              # moderate
        elif kind == 2:
            fb = rng.choice([0, 0x8000, 0x7FFF, 1, 0xFFFF, 0x4000, 0xC000], 512)   # rounding edge cases
        else:
            fb = np.zeros(512, dtype=np.int64); fb[rng.integers(0, 256, 6)] = rng.integers(0, 65536, 6)
        fb = fb.astype(np.uint16)
        fb[256:] = 0
        ovl = rng.integers(0, 65536, 16).astype(np.uint16)
        vs = int(rng.integers(0, 9))
        a = oracle.transform(os_, fb, vs, ovl)
        b = reference.transform(os_, fb, vs, ovl)
        assert np.array_equal(a[0], b[0]), "pcm differs (trial %d)" % trial
        assert np.array_equal(a[1], b[1]), "overlap tail differs"


def test_volume_and_mixing_parameters(oracle, reference):
    for v in range(256):
        assert oracle.volume_multiplier(v) == reference.volume_multiplier(v)
    for os_ in range(4):
        for ls in list(range(-8300, 8300, 61)) + [-8191, 8191, 0, 63, 64, -64]:
            for cv in (0, 1, 0x40, 0xFF):
                assert oracle.mixing_multiplier(os_, ls, cv) == reference.mixing_multiplier(os_, ls, cv)


def test_multichannel_mixing(oracle, reference):
    """up to 8 streams of different lengths mixed in the frequency domain before one transform"""
    for fam, fmts in ((2, [3, 4, 5]), (1, [0, 1]), (0, [0, 2])):
        for nch in (2, 3, 8):
            streams = [make_stream(fmts[c % len(fmts)], 10 + 5 * c, seed=9300 + 17 * c + fam, profile=c % 3)
                       for c in range(nch)]
            levels = [0x64 - 3 * c for c in range(nch)]
            a = oracle.decode(fam, 230, streams, levels, 60)
            b = reference.decode(fam, 230, streams, levels, 60)
            assert np.array_equal(a, b)


def test_looping_is_not_used_but_short_streams_work(oracle, reference):
    for fmt in ALL_FORMATS:
        s = make_stream(fmt, 1, seed=77 + fmt)
        assert np.array_equal(oracle.decode(os_for(fmt), 255, [s], [0x64], 4),
                              reference.decode(os_for(fmt), 255, [s], [0x64], 4))


def _sequence_case(os_, seed, with_error=False):
    """streams of the layouts one OS version plays, ragged lengths, different levels"""
    from util import corrupt
    fmts = [f for f in ALL_FORMATS if os_for(f) == os_ or os_for(f, 1) == os_]
    streams, levels = [], []
    for i in range(6):
        fmt = fmts[i % len(fmts)]
        s = make_stream(fmt, 3 + 5 * i, seed=seed + i, profile=i % 6)
        if with_error and i in (1, 4):
            s = corrupt(s, seed=seed + i)
        streams.append(s)
        levels.append([0x64, 0x7F, 0x20, 0x64, 0x50, 0x7F][i])
    return streams, levels


@pytest.mark.parametrize("os_", [0, 1, 2, 3])
@pytest.mark.parametrize("with_error", [False, True])
def test_extract_streams_sequence_on_one_decoder(oracle, reference, os_, with_error):
    """the --extract-streams loop: decoder state carries from stream to stream"""
    streams, levels = _sequence_case(os_, 5100 + 10 * os_, with_error)
    for vol, extra in ((255, 2), (200, 3)):
        a = oracle.decode_sequence(os_, vol, streams, levels, extra)
        b = reference.decode_sequence(os_, vol, streams, levels, extra)
        assert np.array_equal(a, b)
    # the carry-over is observable: frame 0 of the second stream differs from a fresh decoder's
    a = oracle.decode_sequence(os_, 255, streams[:2], levels[:2], 2)
    n0 = ((streams[0][0] << 8) | streams[0][1]) + 2
    fresh = oracle.decode(os_, 255, [streams[1]], [levels[1]], 1)
    assert not np.array_equal(a[n0], fresh[0]) or not a[n0].any()
