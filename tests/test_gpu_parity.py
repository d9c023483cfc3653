"""HIP path vs the oracle, through the C ABI, on a real MI355X.  Bit-exact or fail."""
import json
import os

import numpy as np
import pytest

import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
from oracle.dcs_oracle import fnv1a64
from util import ALL_FORMATS, FORMAT_NAMES, make_stream, os_for, corrupt

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_streams(oracle, streams, extra=0):
    out = []
    for os_, s, vol, lvl in streams:
        nf = (s[0] << 8) | s[1]
        out.append(oracle.decode(os_, vol, [s], [lvl], nf + extra))
    return np.concatenate(out)


def assert_same(got, want, what=""):
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        f, i = bad[0]
        raise AssertionError("%s: %d samples differ in %d frames; first at frame %d sample %d: got %d want %d"
                             % (what, len(bad), len(set(bad[:, 0])), f, i, got[f, i], want[f, i]))


def test_native_library_is_what_runs(gpu_ctx):
    assert os.path.exists(D.lib_path())
    assert D.device_count() >= 1


@pytest.mark.parametrize("fmt", ALL_FORMATS, ids=[FORMAT_NAMES[f] for f in ALL_FORMATS])
@pytest.mark.parametrize("profile", [0, 1, 2, 3, 4, 5])
def test_single_stream_every_layout(gpu_ctx, oracle, fmt, profile):
    for k in range(2):
        s = make_stream(fmt, 70, seed=11000 + fmt * 32 + profile * 4 + k, profile=profile,
                        stride_from=16 if k == 0 else 7)
        streams = [(os_for(fmt, k), s, [255, 220, 0x67, 255, 240, 200][profile], [0x64, 0x7F, 0x64, 0x20, 0x50, 0x70][profile])]
        pcm, err, _ = gpu_ctx.decode_streams(streams, extra_frames=2)
        assert_same(pcm, oracle_streams(oracle, streams, extra=2), FORMAT_NAMES[fmt])
        assert not err.any()


def test_band_15_shared_by_two_lanes(gpu_ctx, oracle):
    """sixteen lanes per frame (4 frames per wavefront): band 15 of a 1994+ frame, twice as long as any other, is unpacked
    by two lanes, the second starting where the index pass saw the first code boundary past the middle.  Streams with many
    two-zeros codes, strided and not, so that codes run across the middle in both kinds; against the oracle, and the same
    PCM from the other two kernel variants."""
    streams, straddles = [], [0, 0]
    for fmt in (D.FMT_94_T0, D.FMT_94_T1_S0, D.FMT_94_T1_S3):
        for k in range(6):
            s = make_stream(fmt, 90, seed=9900 + fmt * 16 + k, profile=2 if k % 2 else 0, stride_from=16 if k < 3 else 12)
            idx, info = D.index_stream(os_for(fmt, k), s)
            straddles[1 if info.header[15] & 0x40 else 0] += int(((idx["split"][:, 14]["prvDelta"] >> 9) & 1).sum())
            streams.append((os_for(fmt, k), s, 255 - 7 * k, 0x64))
    assert straddles[0] > 10 and straddles[1] > 10
    want = oracle_streams(oracle, streams, extra=1)
    try:
        for fpw in (4, 8, 16):
            gpu_ctx.set_frames_per_wave(fpw)
            pcm, err, _ = gpu_ctx.decode_streams(streams, extra_frames=1)
            assert_same(pcm, want, "fpw=%d" % fpw)
            assert not err.any()
    finally:
        gpu_ctx.set_frames_per_wave(0)


def test_seeded_fuzz_lists_for_a_minute():
    """tools/fuzz_parity.py: random lists (every layout, 1..18 bands, any stride start, damaged and truncated streams) through
    the three kernel variants against the oracle, the device index walk against the host walk, every eighth list through
    the pipeline's three modes, every fourth seed a multi-channel mix; a minute of it here (some 3 000 lists), tens of
    minutes of it per round (profiles/)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "60", "424242"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "all bit-exact" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("handoff", [True, False], ids=["handoff", "halo"])
@pytest.mark.parametrize("fpw", [4, 8, 16])
def test_frames_per_wave_variants_and_chunk_boundaries(gpu_ctx, oracle, fpw, handoff):
    """chunk boundaries fall inside streams: the overlap tail must cross them, through the hand-off buffer
    (published by the wavefront that decodes the predecessor) or by the halo re-decode"""
    streams = [(os_for(f, f), make_stream(f, 45 + 13 * f, seed=12000 + f, profile=f % 3), 240, 0x62 + f)
               for f in ALL_FORMATS]
    gpu_ctx.set_frames_per_wave(fpw)
    gpu_ctx.set_tail_handoff(handoff)
    try:
        pcm, err, _ = gpu_ctx.decode_streams(streams, extra_frames=1)
    finally:
        gpu_ctx.set_frames_per_wave(0)
        gpu_ctx.set_tail_handoff(True)
    assert_same(pcm, oracle_streams(oracle, streams, extra=1), "fpw=%d" % fpw)
    assert not err.any()


def test_handoff_survives_many_launches_of_one_batch(gpu_ctx, oracle):
    """the hand-off words carry the launch's epoch: a resident batch run again and again never sees a stale tail,
    and a batch built on recycled device buffers (another batch's old words) does not either"""
    for rep in range(2):
        streams = [(os_for(f), make_stream(f, 90, seed=12100 + 7 * rep + f, profile=(f + rep) % 4), 255, 0x64)
                   for f in (D.FMT_93_T0, D.FMT_94_T1_S3, D.FMT_93B_T1)]
        b = D.build_stream_batch(streams)
        gpu_ctx.set_frames_per_wave(4)
        try:
            bt = gpu_ctx.batch(b["blob"], b["srcs"], b["jobs"])
        finally:
            gpu_ctx.set_frames_per_wave(0)
        want = oracle_streams(oracle, streams)
        for k in range(40):
            bt.run()
            if k in (0, 1, 39):
                pcm, err = bt.download()[:2]
                assert_same(pcm, want, "launch %d" % k)
                assert not err.any()
        bt.close()


def test_golden_vectors(gpu_ctx):
    """committed reference PCM (tests/golden/make_golden.py), single-stream cases"""
    meta = json.load(open(os.path.join(GOLD, "dcs_golden_hashes.json")))
    arrays = np.load(os.path.join(GOLD, "dcs_golden.npz"))
    n = 0
    for case in meta["cases"]:
        if case["streams"] != 1:
            continue
        s = arrays[case["name"] + "/stream"].tobytes()
        nf = (s[0] << 8) | s[1]
        pcm, err, _ = gpu_ctx.decode_streams([(case["os"], s, case["volume"], case["levels"][0])],
                                             extra_frames=case["frames_out"] - nf)
        assert_same(pcm, arrays[case["name"] + "/pcm"], case["name"])
        n += 1
    assert n >= 32 and any(c["name"] == "CONFIG-1" for c in meta["cases"])      # (BASELINE configs[0] as SURVEY 8(d) words it is among them)


def test_streams_made_by_the_reference_encoder(gpu_ctx, oracle):
    """real-audio band statistics: the 24 recordings of tests/golden/encoder_golden.npz (made by the reference's own
    encoder) against the unmodified reference decoder's PCM, and the realistic_65536 workload built from them against
    its committed per-stream hashes"""
    meta = json.load(open(os.path.join(GOLD, "encoder_golden.json")))
    arrays = np.load(os.path.join(GOLD, "encoder_golden.npz"))
    for c in meta["cases"]:
        s = arrays[c["name"] + "/stream"].tobytes()
        nf = (s[0] << 8) | s[1]
        pcm, err, _ = gpu_ctx.decode_streams([(c["os"], s, c["volume"], c["levels"][0])], extra_frames=c["frames_out"] - nf)
        assert not err.any()
        assert "%016x" % oracle.fnv1a64(pcm) == c["pcm_fnv1a64"], c["name"]
        if c["name"] + "/pcm" in arrays:
            assert_same(pcm, arrays[c["name"] + "/pcm"], c["name"])
    streams = workloads.WORKLOADS["realistic_65536"]()
    pcm, err, first = gpu_ctx.decode_streams(streams)
    assert not err.any()
    got = ["%016x" % oracle.fnv1a64(pcm[first[k]:first[k + 1]]) for k in range(len(streams))]
    assert got == meta["workloads"]["realistic_65536"]["stream_hashes"]


def test_interleaved_mixed_format_batch(gpu_ctx, oracle):
    """configs[3] shape at reduced size: neighbouring frames alternate among the six layouts"""
    b = workloads.build("mixed_16384", n_streams=30, n_frames=50)
    pcm, err = gpu_ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
    want = oracle_streams(oracle, b["streams"])[b["perm"]]
    assert_same(pcm, want, "interleaved")
    assert not err.any()


def test_corrupted_streams_match_oracle_error_semantics(gpu_ctx, oracle):
    """bit-flipped payloads: zeroed bands, stop at the failing frame, silence afterwards"""
    streams = []
    for fmt in ALL_FORMATS:
        for k in range(12):
            s = corrupt(make_stream(fmt, 20, seed=13000 + fmt * 16 + k, profile=k % 4), seed=100 + k, nflips=3)
            streams.append((os_for(fmt), s + bytes(1024), 255, 0x64))
    pcm, err, first = gpu_ctx.decode_streams(streams, extra_frames=2)
    want = oracle_streams(oracle, streams, extra=2)
    assert_same(pcm, want, "corrupted")
    assert err.any()


def test_external_tails_and_tails_out(gpu_ctx, oracle):
    """streaming use: decode a stream in two calls, carrying the 16-sample overlap tail by hand"""
    for fmt in (D.FMT_93_T0, D.FMT_94_T1_S3):
        s = make_stream(fmt, 40, seed=14000 + fmt)
        os_ = os_for(fmt)
        b = D.build_stream_batch([(os_, s, 255, 0x64)])
        want = oracle_streams(oracle, [(os_, s, 255, 0x64)])
        jobs_a = b["jobs"][:17].copy()
        pcm_a, _, tails_a = gpu_ctx.decode_batch(b["blob"], b["srcs"], jobs_a, want_tails=True)
        jobs_b = b["jobs"][17:].copy()
        jobs_b["prev"] = np.arange(jobs_b.size, dtype=np.int64) - 1
        jobs_b["prev"][0] = D.PREV_EXT | 0
        pcm_b, _ = gpu_ctx.decode_batch(b["blob"], b["srcs"], jobs_b, tails_in=tails_a[16:17])
        assert_same(np.concatenate([pcm_a, pcm_b]), want, "two-call streaming")


def test_resident_batch_keeps_chain_end_tails_unless_asked_for_all(gpu_ctx):
    """dcs_ctx_set_batch_tails: by default a resident batch stores the tail of the LAST frame of every chain (what carries a
    stream into its next batch), the other rows read as zero; with all_frames every row is what dcs_decode_batch returns.
    Three streams of unlike lengths and layouts, so that chain ends fall inside chunks."""
    streams = [(os_for(f), make_stream(f, n, seed=15000 + f), 255, 0x64) for f, n in ((D.FMT_94_T1_S3, 21), (D.FMT_93_T0, 9), (D.FMT_94_T0, 14))]
    b = D.build_stream_batch(streams)
    _, _, want = gpu_ctx.decode_batch(b["blob"], b["srcs"], b["jobs"], want_tails=True)
    ends = np.asarray(b["first_job"][1:], dtype=np.int64) - 1
    assert want[ends].any()
    for fpw in (4, 8, 16):
        gpu_ctx.set_frames_per_wave(fpw)
        try:
            bt = gpu_ctx.batch(b["blob"], b["srcs"], b["jobs"])
            bt.run(); bt.run()
            _, _, got = bt.download(want_tails=True)
            bt.close()
            assert np.array_equal(got[ends], want[ends]), fpw
            rest = np.ones(got.shape[0], bool); rest[ends] = False
            assert not got[rest].any(), fpw
            gpu_ctx.set_batch_tails(True)
            bt = gpu_ctx.batch(b["blob"], b["srcs"], b["jobs"])
            bt.run()
            _, _, got = bt.download(want_tails=True)
            bt.close()
            assert np.array_equal(got, want), fpw
        finally:
            gpu_ctx.set_batch_tails(False)
            gpu_ctx.set_frames_per_wave(0)


def test_multichannel_mix(gpu_ctx, oracle):
    """several sources mixed into one output frame (MainLoop's per-channel DecodeStream loop, :272-273)"""
    meta = json.load(open(os.path.join(GOLD, "dcs_golden_hashes.json")))
    arrays = np.load(os.path.join(GOLD, "dcs_golden.npz"))
    from mixer_ref import build_mix_batch
    for case in meta["cases"]:
        if case["streams"] == 1:
            continue
        streams = [arrays["%s/stream%d" % (case["name"], c)].tobytes() for c in range(case["streams"])]
        b = build_mix_batch(case["os"], case["volume"], streams, case["levels"], case["frames_out"])
        pcm, err = gpu_ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
        assert_same(pcm, arrays[case["name"] + "/pcm"], case["name"])


@pytest.mark.parametrize("wl", ["dcs93_4096", "dcs94_65536", "mixed_16384", "survey3_65536"])
def test_full_size_workloads_hash_and_sampled_oracle(gpu_ctx, oracle, wl):
    """BASELINE.json sizes: per-stream FNV-1a of the HIP PCM == the reference's committed hashes, plus
    a direct oracle comparison on a sample of streams"""
    meta = json.load(open(os.path.join(GOLD, "dcs_golden_hashes.json")))["workloads"][wl]
    b = workloads.build(wl)
    pcm, err = gpu_ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
    assert not err.any()
    if "perm" in b:
        inv = np.empty_like(b["perm"]); inv[b["perm"]] = np.arange(b["perm"].size)
        pcm = pcm[inv]
    first = b["first_job"]
    got = ["%016x" % fnv1a64(pcm[first[k]:first[k + 1]].tobytes()) for k in range(len(first) - 1)]
    assert got == meta["stream_hashes"]
    for k in range(0, len(b["streams"]), max(1, len(b["streams"]) // 16)):
        assert_same(pcm[first[k]:first[k + 1]], oracle_streams(oracle, [b["streams"][k]]), "%s stream %d" % (wl, k))


@pytest.mark.parametrize("wl", ["dcs94_65536", "dcs93_4096"])
def test_full_size_every_plan_gives_the_same_pcm(gpu_ctx, wl):
    """size-independent property at BASELINE sizes: the PCM does not depend on how the batch is cut into chunks
    (4 / 8 / 16 frames per wavefront: one to many rounds of workgroups, every seam of the XCD mapping) nor on how the
    overlap tails cross the cuts (hand-off buffer / halo re-decode); the reference hashes pin one of the plans"""
    b = workloads.build(wl)
    meta = json.load(open(os.path.join(GOLD, "dcs_golden_hashes.json")))["workloads"][wl]
    first = b["first_job"]
    want = None
    try:
        for fpw, handoff in ((16, True), (8, True), (4, True), (8, False), (16, False)):
            gpu_ctx.set_frames_per_wave(fpw)
            gpu_ctx.set_tail_handoff(handoff)
            pcm, err = gpu_ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
            assert not err.any()
            if want is None:
                want = pcm
                got = ["%016x" % fnv1a64(pcm[first[k]:first[k + 1]].tobytes()) for k in range(len(first) - 1)]
                assert got == meta["stream_hashes"]
            else:
                assert np.array_equal(pcm, want), "fpw=%d handoff=%s" % (fpw, handoff)
    finally:
        gpu_ctx.set_frames_per_wave(0)
        gpu_ctx.set_tail_handoff(True)


def test_pinned_download_view_and_buffer_reuse(gpu_ctx):
    """dcs_batch_download_view hands out the result in pinned memory; buffers of closed batches are reused"""
    b = workloads.build("dcs93_4096", n_streams=6, n_frames=20)
    want = None
    for _ in range(3):                                  # the 2nd and 3rd batch run on recycled buffers
        bt = gpu_ctx.batch(b["blob"], b["srcs"], b["jobs"])
        bt.run(); bt.sync()
        pcm, err = bt.download()
        pv, ev = bt.download_view()
        assert np.array_equal(pv, pcm) and np.array_equal(ev, err)
        want = pcm if want is None else want
        assert np.array_equal(pcm, want)
        bt.close()


@pytest.mark.gpu
def test_batch_is_idempotent_and_resident(gpu_ctx):
    """a resident batch run twice gives identical PCM; timing entry returns a positive duration"""
    b = workloads.build("dcs93_4096", n_streams=16, n_frames=32)
    bt = gpu_ctx.batch(b["blob"], b["srcs"], b["jobs"])
    bt.run(); p1, _ = bt.download()
    ms = bt.time(3)
    p2, _ = bt.download()
    assert np.array_equal(p1, p2) and ms > 0
    assert bt.algorithmic_bytes > 480 * 512
    bt.close()


@pytest.mark.gpu
def test_gpu_index_pass_equals_host_index_pass(gpu_ctx, oracle):
    """dcs_index_streams_gpu (one wavefront per stream) must return the host walker's records bit for bit --
    valid streams of every layout, corrupted streams, ragged lengths -- and decode to the oracle's PCM."""
    streams = []
    for i, fmt in enumerate(ALL_FORMATS * 4):
        s = make_stream(fmt, 3 + 11 * i, seed=4000 + i, profile=i % 6)
        if i % 6 == 5:
            s = corrupt(s, seed=i)
        streams.append((os_for(fmt, i), s, 0xE0, 0x64))
    got = gpu_ctx.index_streams_gpu(streams)
    for (os_, data, _, _), (idx, info) in zip(streams, got):
        want_idx, want_info = D.index_stream(os_, data)
        assert idx.tobytes() == want_idx.tobytes()
        assert bytes(info) == bytes(want_info)
    b = D.build_stream_batch(streams, extra_frames=2, indexer=gpu_ctx.index_streams_gpu)
    pcm, err = gpu_ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
    assert_same(pcm, oracle_streams(oracle, streams, extra=2), "gpu-indexed batch")


@pytest.mark.gpu
def test_gpu_index_pass_many_streams_fills_waves(gpu_ctx):
    """more streams than one-lane-per-wave placement covers: the lanes-per-wave > 1 path"""
    streams = []
    for i in range(5000):
        fmt = ALL_FORMATS[i % len(ALL_FORMATS)]
        streams.append((os_for(fmt, i), make_stream(fmt, 1 + i % 3, seed=70000 + i, profile=i % 4), 0xFF, 0x64))
    got = gpu_ctx.index_streams_gpu(streams)
    want = D.api.index_streams(streams)
    for (gi, ginfo), (wi, winfo) in zip(got, want):
        assert gi.tobytes() == wi.tobytes()
        assert bytes(ginfo) == bytes(winfo)


@pytest.mark.gpu
def test_gpu_index_pass_rejects_bad_shapes(gpu_ctx):
    s = make_stream(ALL_FORMATS[0], 4, seed=1)
    blob, locs = D.api.pack_streams([(os_for(ALL_FORMATS[0]), s, 0xFF, 0x64)])
    b = np.frombuffer(blob, dtype=np.uint8)
    out = np.zeros(4, dtype=D.api.INDEX_DTYPE)
    infos = np.zeros(1, dtype=D.api.INFO_DTYPE)
    L = gpu_ctx.L
    bad = locs.copy(); bad["len"] = b.size + 100
    assert L.dcs_index_streams_gpu(gpu_ctx.h, b.ctypes.data, b.size, bad.ctypes.data, 1, out.ctypes.data, 4, infos.ctypes.data) == D.api.ERR_INVALID_ARG
    assert L.dcs_index_streams_gpu(gpu_ctx.h, b.ctypes.data, b.size, locs.ctypes.data, 1, out.ctypes.data, 3, infos.ctypes.data) == D.api.ERR_CAPACITY


@pytest.mark.gpu
@pytest.mark.parametrize("os_", [0, 1, 2, 3])
def test_stream_sequence_on_one_decoder(gpu_ctx, oracle, os_):
    """dcs_decode_stream_sequence = the --extract-streams loop in one launch: frame 0 of each stream is mixed
    with what the previous stream's level left behind (a stream cut short by an error leaves level 0)"""
    from test_oracle_vs_ref import _sequence_case
    for with_error in (False, True):
        streams, levels = _sequence_case(os_, 6100 + 10 * os_, with_error)
        for vol, extra in ((255, 2), (190, 4)):
            pcm, err, first = gpu_ctx.decode_stream_sequence(os_, vol, streams, levels, extra)
            assert_same(pcm, oracle.decode_sequence(os_, vol, streams, levels, extra), "sequence os %d" % os_)
            assert first[-1] == pcm.shape[0]
    L = gpu_ctx.L
    assert L.dcs_decode_stream_sequence(gpu_ctx.h, None, 0, 2, None, 0, None, None) == D.api.ERR_INVALID_ARG


@pytest.mark.gpu
def test_caller_made_index_records_cannot_break_the_launch(gpu_ctx):
    """the batch ABI takes index records from the caller: inconsistent ones are refused before anything is
    launched, and consistent ones that do not belong to the bytes (records of another stream) decode to
    garbage but stay inside the kernel's LDS (bounded readers, bounded accumulator indices)"""
    a = make_stream(D.FMT_94_T1_S3, 40, seed=91001, profile=0)
    b_ = make_stream(D.FMT_94_T1_S3, 40, seed=91002, profile=3)
    os_ = os_for(D.FMT_94_T1_S3)
    batch = D.build_stream_batch([(os_, a, 255, 0x64), (os_, b_, 255, 0x64)])
    srcs = batch["srcs"].copy()
    bad = srcs.copy()
    bad["idx"]["split"][3]["bitDelta"][5] = 60000                # beyond the frame
    with pytest.raises(D.DcsError):
        gpu_ctx.decode_batch(batch["blob"], bad, batch["jobs"])
    bad = srcs.copy()
    bad["idx"]["split"][3]["state"][2] = 0x1FF                  # output index outside the row
    with pytest.raises(D.DcsError):
        gpu_ctx.decode_batch(batch["blob"], bad, batch["jobs"])
    bad = srcs.copy()
    bad["idx"]["split"][7]["prv"][14] = 50000                    # 1994+: the middle of band 15 beyond the frame
    with pytest.raises(D.DcsError):
        gpu_ctx.decode_batch(batch["blob"], bad, batch["jobs"])
    bad = srcs.copy()
    bad["idx"]["split"][7]["prvDelta"][14] = 0x1F0               # ... its output index outside the row
    with pytest.raises(D.DcsError):
        gpu_ctx.decode_batch(batch["blob"], bad, batch["jobs"])
    # swap the records of the two streams (stream offsets stay): self-consistent, but not these bytes' records
    n = 40
    swapped = srcs.copy()
    swapped["idx"][:n], swapped["idx"][n:] = srcs["idx"][n:].copy(), srcs["idx"][:n].copy()
    for fpw in (4, 8, 16):
        gpu_ctx.set_frames_per_wave(fpw)
        pcm, err = gpu_ctx.decode_batch(batch["blob"], swapped, batch["jobs"])
        assert pcm.shape == (80, 240)
    gpu_ctx.set_frames_per_wave(0)
    good, _ = gpu_ctx.decode_batch(batch["blob"], srcs, batch["jobs"])
    assert good.any()


def _slice_batch(b, lo, hi, first_prev):
    """frames [lo, hi) of a one-stream batch as a batch of their own: the first takes its tail from outside"""
    jobs = b["jobs"][lo:hi].copy()
    s0 = int(jobs["firstSrc"][jobs["nSrc"] > 0].min()) if (jobs["nSrc"] > 0).any() else 0
    s1 = int((jobs["firstSrc"] + jobs["nSrc"])[jobs["nSrc"] > 0].max()) if (jobs["nSrc"] > 0).any() else 0
    jobs["firstSrc"] = np.where(jobs["nSrc"] > 0, jobs["firstSrc"] - s0, 0)
    jobs["prev"] = np.arange(hi - lo, dtype=np.int64) - 1
    jobs["prev"][0] = first_prev
    return b["srcs"][s0:s1], jobs


@pytest.mark.parametrize("zc", ["default", "always-copy", "never-copy"])
def test_live_decoder_call_after_call(dcs, oracle, zc, monkeypatch):
    """dcs_decode_batch_live, the path DCSDecoderHIP's pump takes: ONE context, call after call of 1, 2, 5, 64, 65 and 163 frames
    of one stream per layout, every call's first frame taking the tail the call before left (DCS_PREV_EXT) -- nothing allocated,
    cleared or created per call, hand-off words told apart by a launch counter that only grows.  With the packages read and the
    PCM written over the link (small calls by default), with copies both ways, and with neither."""
    if zc == "always-copy":
        monkeypatch.setenv("DCS_LIVE_ZC_UP_KB", "0"); monkeypatch.setenv("DCS_LIVE_ZC_DOWN_FRAMES", "0")
    elif zc == "never-copy":
        monkeypatch.setenv("DCS_LIVE_ZC_UP_KB", "1000000"); monkeypatch.setenv("DCS_LIVE_ZC_DOWN_FRAMES", "1000000")
    ctx = dcs.Context(0)
    try:
        for fmt in ALL_FORMATS:
            s = make_stream(fmt, 300, seed=31000 + fmt, profile=fmt % 4)
            streams = [(os_for(fmt), s, 255, 0x64)]
            b = D.build_stream_batch(streams, extra_frames=2)
            want = oracle_streams(oracle, streams, extra=2)
            got, tail, lo = [], None, 0
            for n in (1, 2, 5, 64, 65, 163, 2):
                srcs, jobs = _slice_batch(b, lo, lo + n, D.PREV_NONE if tail is None else (D.PREV_EXT | 0))
                pcm, err, tails = ctx.decode_batch_live(b["blob"], srcs, jobs, tails_in=tail)
                assert not err.any()
                got.append(pcm)
                tail = tails[-1:].copy()
                lo += n
            assert_same(np.concatenate(got), want, "%s %s" % (FORMAT_NAMES[fmt], zc))
    finally:
        ctx.close()


def test_live_decoder_keeps_named_blobs_resident(dcs, oracle):
    """multi-channel frames read their second and later sources from the blob on the device: under a name (blobId) the blob is
    uploaded once and only what is appended later goes up; another name, or none, replaces it; the PCM is the same"""
    meta = json.load(open(os.path.join(GOLD, "dcs_golden_hashes.json")))
    arrays = np.load(os.path.join(GOLD, "dcs_golden.npz"))
    from mixer_ref import build_mix_batch
    cases = [c for c in meta["cases"] if c["streams"] > 1]
    ctx = dcs.Context(0)
    try:
        for rep, case in enumerate(cases + cases[:1]):
            streams = [arrays["%s/stream%d" % (case["name"], c)].tobytes() for c in range(case["streams"])]
            b = build_mix_batch(case["os"], case["volume"], streams, case["levels"], case["frames_out"])
            want = arrays[case["name"] + "/pcm"]
            n = len(b["jobs"])
            half = n // 2
            # first half under the blob's name, second half with the blob grown (appended bytes only go up), tail carried
            srcs, jobs = _slice_batch(b, 0, half, D.PREV_NONE)
            pcm_a, err, tails = ctx.decode_batch_live(b["blob"], srcs, jobs, blob_id=100 + rep)
            grown = b["blob"] + bytes(range(256)) * 3
            srcs, jobs = _slice_batch(b, half, n, D.PREV_EXT | 0)
            pcm_b, err, _ = ctx.decode_batch_live(grown, srcs, jobs, tails_in=tails[-1:], blob_id=100 + rep)
            assert_same(np.concatenate([pcm_a, pcm_b]), want, case["name"] + " named")
            # the same without a name, and through the one-shot call
            pcm, err, _ = ctx.decode_batch_live(b["blob"], b["srcs"], b["jobs"])
            assert_same(pcm, want, case["name"] + " unnamed")
            pcm, err = ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
            assert_same(pcm, want, case["name"] + " one-shot")
    finally:
        ctx.close()
