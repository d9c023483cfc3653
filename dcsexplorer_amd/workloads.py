"""Seeded synthetic workloads: the BASELINE.json configurations restated on the library's own
stream writer (the reference ships no audio).  Used by bench.py and by the parity tests; everything is
derived from integer seeds so the GPU box regenerates byte-identical inputs.

  dcs93_4096     configs[1]: 4096 DCS-93 frames = 64 streams x 64 frames, OS93 Type 0, to the letter of SURVEY.md section
                 8(d) Config 2 (12 populated bands, scale codes 0x20..0x34, the stated band-type code mix: synth profile 6)
  dcs94_65536    configs[2]: 65536 1994+ ("DCS-95 format") frames = 256 streams x 256 frames,
                 80 % Type 1 sub-type 3, 10 % Type 1 sub-type 0, 10 % Type 0
  survey3_65536  configs[2] to the letter of SURVEY.md section 8(d), Config 3: the same 256 x 256 frames and layout mix with
                 12 populated bands, band-type deltas 0 / +-1 / +-2 / other at 70 / 20 / 8 / 2 %, a quarter of the
                 Huffman-coded values zero, 120 bytes a frame (synth profile 5); dcs94_65536 has 16 bands at 96
  mixed_16384    configs[3]: 128 streams x 128 frames over all six unpack layouts, frames interleaved
                 so that neighbouring frames of the batch alternate formats
  realistic_65536  (not a BASELINE config) 256 x 256 frames of streams made by the reference's own encoder
  corpus         configs[4] stand-in: `titles` synthetic titles of `streams_per_title` streams with
                 U[20, max_frames] frames each (no ROM corpus exists in the reference tree)
"""
import numpy as np

from . import api as D


def _splitmix(seed):
    x = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def streams_dcs93_4096(n_streams=64, n_frames=64, first=0):
    out = []
    for k in range(first, first + n_streams):
        # SURVEY 8(d) Config 2: bands 0-11 populated, 12-15 empty; one stream in ten carries the stride bit on bands >= 6
        # (a strided Type-0 band spans 32 slots, so the writer keeps as many of the twelve as fit the frame buffer: ten)
        strided = (k % 10) == 9
        s = D.synth_stream(D.FMT_93_T0, n_frames, seed=0x93020002 + k, nbands=12,
                           stride_from=6 if strided else 16, profile=6)
        out.append((D.OS93A if (k & 1) else D.OS93B, s, 255, 0x64))
    return out


def streams_dcs94_65536(n_streams=256, n_frames=256, first=0):
    out = []
    for k in range(first, first + n_streams):
        m = k % 10
        fmt = D.FMT_94_T0 if m == 0 else D.FMT_94_T1_S0 if m == 1 else D.FMT_94_T1_S3
        s = D.synth_stream(fmt, n_frames, seed=0x94000003 + k, nbands=16, stride_from=16 if (k % 7) else 12,
                           profile=0)
        out.append((D.OS95 if (k & 1) else D.OS94, s, 220, 0x64))
    return out


def streams_survey3_65536(n_streams=256, n_frames=256, first=0):
    out = []
    for k in range(first, first + n_streams):
        m = k % 10
        fmt = D.FMT_94_T0 if m == 0 else D.FMT_94_T1_S0 if m == 1 else D.FMT_94_T1_S3
        s = D.synth_stream(fmt, n_frames, seed=0x94000003 + k, nbands=12, stride_from=16, profile=5)
        out.append((D.OS95 if (k & 1) else D.OS94, s, 220, 0x64))
    return out


def streams_mixed_16384(n_streams=128, n_frames=128, first=0):
    out = []
    for k in range(first, first + n_streams):
        fmt = k % 6
        os_ = D.format_os(fmt, prefer_95=bool(k & 8), prefer_93a=bool(k & 8))
        s = D.synth_stream(fmt, n_frames, seed=0x00040004 + k, nbands=18 if fmt == D.FMT_93A_T1 else 16,
                           stride_from=16, profile=k % 3)
        out.append((os_, s, 200 + (k % 56), 0x60 + (k % 16)))
    return out


def corpus_manifest(titles=29, streams_per_title=600, max_frames=2000, seed=0x0005):
    """BASELINE config 5 stand-in: `titles` synthetic titles x `streams_per_title` streams of
    U[20, max_frames] frames, the era of a title fixing its OS version and the mix of unpack layouts.  Returns the
    list of stream recipes (os, format, frames, level, synth seed, nbands) WITHOUT writing any stream: the frame
    counts are all a partition needs, so every rank derives the same manifest and writes only its own range."""
    out = []
    for t in range(titles):
        g = _splitmix(seed * 1000003 + t)
        era = t % 4                      # 0: OS93a, 1: OS93b, 2: OS94, 3: OS95
        for k in range(streams_per_title):
            r = next(g)
            nf = 20 + r % (max_frames - 19)
            if era == 0:
                fmt = D.FMT_93A_T1 if (r >> 20) % 8 == 0 else D.FMT_93_T0
            elif era == 1:
                fmt = D.FMT_93B_T1 if (r >> 20) % 2 else D.FMT_93_T0
            else:
                m = (r >> 20) % 10
                fmt = D.FMT_94_T0 if m == 0 else D.FMT_94_T1_S0 if m == 1 else D.FMT_94_T1_S3
            out.append(dict(os=era, format=fmt, frames=int(nf), level=0x60 + (r >> 40) % 16,
                            seed=(seed << 32) + t * 100000 + k, nbands=18 if fmt == D.FMT_93A_T1 else 16,
                            title=t, stream=k))
    return out


def corpus_frames(manifest):
    return np.array([m["frames"] for m in manifest], dtype=np.uint32)


def corpus_streams(manifest, lo=0, hi=None, threads=0):
    """write streams [lo, hi) of a manifest: -> [(os, bytes, volume, level)].  The writer is the library's C
    function (ctypes releases the GIL), so a thread pool scales it over the host cores."""
    part = manifest[lo:hi]
    def one(m):
        return (m["os"], D.synth_stream(m["format"], m["frames"], seed=m["seed"], nbands=m["nbands"], stride_from=16,
                                        profile=0), 255, m["level"])
    threads = threads or min(32, D.host_threads())
    if threads <= 1 or len(part) < 4:
        return [one(m) for m in part]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(one, part, chunksize=max(1, len(part) // (threads * 8))))


def streams_corpus(titles=29, streams_per_title=600, max_frames=2000, seed=0x0005):
    return corpus_streams(corpus_manifest(titles, streams_per_title, max_frames, seed))


RECORDINGS = None


def register_recordings(recordings):
    """hand the package the encoder-made recordings the `realistic_65536` workload is built from: a mapping with
    "ENC-<layout>-v<k>/stream" -> bytes-like for the six layouts x four variants.  The package itself reads no data
    files: bench.py and the tests load tests/golden/encoder_golden.npz (made by tests/golden/make_encoder_golden.py) and
    register it."""
    global RECORDINGS
    RECORDINGS = recordings


def streams_realistic_65536(n_streams=256, n_frames=256, first=0):
    """256 streams x 256 frames made by the REFERENCE'S OWN ENCODER from a deterministic signal (24 recordings: six
    layouts x four pitch / noise variants; see register_recordings), each replica at its own volume and mixing level.
    Real-audio band statistics instead of the seeded writer's; there is no encoder for OS93a Type 1."""
    if RECORDINGS is None:
        raise RuntimeError("realistic_65536 needs the encoder-made recordings: call workloads.register_recordings() first "
                           "(bench.py and tests/conftest.py do, with tests/golden/encoder_golden.npz)")
    ENCODER_GOLDEN = RECORDINGS
    layouts = [("94-T0", D.OS94), ("94-T1s0", D.OS95), ("94-T1s3", D.OS95), ("93b-T0", D.OS93B), ("93b-T1", D.OS93B), ("93a-T0", D.OS93A)]
    names = [("ENC-%s-v%d" % (l, v), l, o) for v in range(4) for l, o in layouts]
    assert n_frames == 256
    out = []
    for k in range(first, first + n_streams):
        name, lay, os_ = names[k % len(names)]
        if lay.startswith("94") and (k & 1):
            os_ = D.OS94 if os_ == D.OS95 else D.OS95       # OS94 and OS95 share the codec
        out.append((os_, bytes(np.asarray(ENCODER_GOLDEN[name + "/stream"]).tobytes()), 200 + (k % 56), 0x60 + (k % 16)))
    return out


def interleave(batch):
    """re-order the jobs of a build_stream_batch() result round-robin over the streams, remapping the
    overlap links; returns (new batch dict, perm) with new_jobs[i] = old_jobs[perm[i]]"""
    first = batch["first_job"]
    n_streams = len(first) - 1
    lens = np.diff(first)
    order = []
    for f in range(int(lens.max())):
        for s in range(n_streams):
            if f < lens[s]:
                order.append(first[s] + f)
    perm = np.array(order, dtype=np.int64)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    jobs = batch["jobs"][perm].copy()
    prev = jobs["prev"].astype(np.int64)
    link = (prev != D.PREV_NONE) & ((prev & D.PREV_EXT) == 0)
    prev[link] = inv[prev[link]]
    jobs["prev"] = prev.astype(np.uint32)
    nb = dict(batch)
    nb["jobs"] = jobs
    return nb, perm


def shifted(fn, stream_offset):
    """the same workload shape over a different range of the (unbounded) seeded corpus: stream k of the
    result is stream k + stream_offset of the corpus.  Used to give every rank its own range."""
    return fn(first=stream_offset)


def streams_one_layout(fmt, n_streams=512, n_frames=128, first=0):
    """diagnostic (tools/per_format.py, tools/stamps.py layout_<k>): one unpack layout on its own, 65 536 frames"""
    out = []
    for k in range(first, first + n_streams):
        s = D.synth_stream(fmt, n_frames, seed=0x5150000 + fmt * 4096 + k, nbands=18 if fmt == D.FMT_93A_T1 else 16,
                           stride_from=16, profile=k % 3)
        out.append((D.format_os(fmt, prefer_95=bool(k & 1), prefer_93a=bool(k & 1)), s, 230, 0x64))
    return out


WORKLOADS = {
    "layout_0": lambda **kw: streams_one_layout(0, **kw), "layout_1": lambda **kw: streams_one_layout(1, **kw),
    "layout_2": lambda **kw: streams_one_layout(2, **kw), "layout_3": lambda **kw: streams_one_layout(3, **kw),
    "layout_4": lambda **kw: streams_one_layout(4, **kw), "layout_5": lambda **kw: streams_one_layout(5, **kw),
    "dcs93_4096": streams_dcs93_4096,
    "dcs94_65536": streams_dcs94_65536,
    "survey3_65536": streams_survey3_65536,
    "mixed_16384": streams_mixed_16384,
    "realistic_65536": streams_realistic_65536,
}


def build(name, **kw):
    """returns dict(blob, srcs, jobs, first_job, streams)"""
    streams = WORKLOADS[name](**kw)
    b = D.build_stream_batch(streams)
    b["streams"] = streams
    if name == "mixed_16384":
        nb, perm = interleave(b)
        nb["streams"] = streams
        nb["perm"] = perm
        return nb
    return b
