"""host side of the library (no GPU): index pass, mixing parameters, chunk planner, stream writer"""
import numpy as np
import pytest

import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
from util import ALL_FORMATS, FORMAT_NAMES, make_stream, os_for, corrupt


@pytest.mark.parametrize("fmt", ALL_FORMATS, ids=[FORMAT_NAMES[f] for f in ALL_FORMATS])
def test_index_pass_matches_oracle_probes(oracle, fmt):
    for k in range(4):
        s = make_stream(fmt, 60, seed=4200 + fmt * 8 + k, profile=k, stride_from=16 if k < 2 else 8)
        os_ = os_for(fmt, k)
        idx, info = D.index_stream(os_, s)
        _, probes = oracle.decode(os_, 255, [s], [0x64], 60, probes=True)
        assert info.nFrames == 60 and info.nValidFrames == 60 and info.format == fmt
        assert info.hdrLen == (1 if fmt == D.FMT_93A_T1 else 16)
        _, probes2 = oracle.decode(os_, 255, [s + bytes(8)], [0x64], 61, probes=True)
        for f in range(60):
            assert idx[f]["bitOff"] == probes[f].bitOff
            if fmt >= D.FMT_94_T0:
                # 1994+: the record holds the codes AFTER the frame's header deltas = what frame f+1 carries in
                if f < 59:
                    assert list(idx[f]["bandType"]) == list(probes[f + 1].bandType)
            elif fmt == D.FMT_93B_T1:
                assert list(idx[f]["bandType"]) == list(probes[f].bandType)
            assert 1 <= idx[f]["nBands"] <= (18 if fmt == D.FMT_93A_T1 else 16)
            for q in range(min(int(idx[f]["nBands"]), 16) - 1):
                sp = idx[f]["split"][q]
                assert sp["bitDelta"] <= idx[f]["nBits"]
        assert np.all(np.diff(idx["bitOff"].astype(np.int64)) == idx["nBits"][:-1])
        assert int((idx["flags"] >> 4).max()) == 0
        assert np.all((idx["flags"] & D.IDX_SERIAL) == 0)         # only frames with errors are serial
        oi = oracle.stream_info(os_, s)
        assert (info.nBytes, info.formatType, info.formatSubType, bytes(info.header)) == \
               (oi["nBytes"], oi["formatType"], oi["formatSubType"], oi["header"])


def test_index_pass_records_the_middle_of_band_15():
    """1994+: band 15 has 32 samples, twice any other band; the index pass notes where its second half starts (the first
    code boundary with half of the samples done) in the two fields of that band's split record the 1993 formats use, so
    that two lanes can share the band.  Checked here: position and output index are consistent with the band's own
    record, a two-zeros code across the middle shows as one sample more, an empty band has no middle."""
    seen = {"plain": 0, "straddle": 0, "strided_plain": 0, "strided_straddle": 0, "empty": 0}
    for fmt in (D.FMT_94_T0, D.FMT_94_T1_S0, D.FMT_94_T1_S3):
        for k in range(6):
            s = make_stream(fmt, 80, seed=9900 + fmt * 16 + k, profile=2 if k % 2 else 0, stride_from=16 if k < 3 else 12)
            idx, info = D.index_stream(os_for(fmt, k), s)
            strided = (info.header[15] & 0x40) != 0
            for r in idx:
                sp = r["split"][14]
                start, mid, st = int(sp["state"]) & 0x1FF, int(sp["prvDelta"]) & 0x1FF, int(sp["prvDelta"]) >> 9
                if r["bandType"][15] == 0:
                    assert int(sp["prv"]) == 0 and int(sp["prvDelta"]) == 0
                    seen["empty"] += 1
                    continue
                assert int(sp["bitDelta"]) < int(sp["prv"]) < int(r["nBits"])
                assert st in (0, 1) and mid == start + 16 + (2 if strided else 1) * st
                seen[("strided_" if strided else "") + ("straddle" if st else "plain")] += 1
    assert all(v > 0 for v in seen.values()), seen


def test_fast_walk_equals_literal_walk():
    """dcs_index_stream (64-bit window, byte pointer computed, several 1994+ codes per look) against the walk with the
    literal restatement of the reference's reader, one code per look: same records, same StreamInfo incl. nBytes, on
    every layout and profile, on corrupted streams and on streams cut short in the middle of a frame"""
    import ctypes
    n = 0
    for fmt in ALL_FORMATS:
        for k in range(12):
            s = make_stream(fmt, 40 + k, seed=8800 + fmt * 32 + k, profile=k % 6, stride_from=16 if k % 3 else 8)
            variants = [s, corrupt(s, 200 + k, nflips=4), s[:len(s) * 2 // 3], s + bytes(7)]
            for v in variants:
                os_ = os_for(fmt, k)
                try:
                    a, ia = D.index_stream(os_, v)
                except D.DcsError as e:
                    with pytest.raises(D.DcsError):
                        D.index_stream(os_, v, literal=True)
                    continue
                b, ib = D.index_stream(os_, v, literal=True)
                assert a.tobytes() == b.tobytes(), (fmt, k)
                assert bytes(ctypes.string_at(ctypes.byref(ia), ctypes.sizeof(ia))) == bytes(ctypes.string_at(ctypes.byref(ib), ctypes.sizeof(ib)))
                n += 1
    assert n > 250


def test_index_pass_error_semantics_match_oracle(oracle):
    """corrupted payloads: the stream ends at the first frame that raises STOP/FATAL, like the
    reference's channel.stop sweep (DCSDecoderNative.cpp:95-116)"""
    hit = 0
    for fmt in ALL_FORMATS:
        for k in range(40):
            s = corrupt(make_stream(fmt, 30, seed=5100 + fmt * 64 + k, profile=k % 6), seed=k, nflips=4)
            os_ = os_for(fmt)
            idx, info = D.index_stream(os_, s)
            _, _, _, stops = oracle.decompress(os_, s, 0x7FFF, 30)
            first_bad = next((f for f in range(30) if stops[f]), None)
            if first_bad is None:
                assert info.nValidFrames == 30
            else:
                hit += 1
                assert info.nValidFrames == first_bad + 1
                assert (idx[first_bad]["flags"] >> 4) == stops[first_bad]
                assert idx[first_bad]["flags"] & D.IDX_SERIAL
    assert hit > 5      # the corruption does trigger the error paths


def test_mixing_parameters_match_oracle(oracle):
    for v in range(256):
        assert D.volume_multiplier(v) == oracle.volume_multiplier(v)
    for os_ in range(4):
        for ls in list(range(-8300, 8300, 97)) + [8191, -8191, 0]:
            for cv in (0, 0x7F, 0xFF):
                assert D.mixing_multiplier(os_, ls, cv) == oracle.mixing_multiplier(os_, ls, cv)
        for vol, lvl in [(255, 0x7F), (255, 0x64), (220, 0x64), (0x67, 0x64), (0, 0x64), (1, 0), (255, 0xFF), (37, -5)]:
            mm, vs = D.stream_params(os_, vol, lvl, 6)
            omm, ovs = oracle.frame_params(os_, vol, lvl, 6)
            assert np.array_equal(mm, omm) and np.array_equal(vs, ovs)
    # values measured on the reference by the survey
    mm, vs = D.stream_params(D.OS94, 255, 0x7F, 3)
    assert (mm[0], vs[0], mm[1], vs[1]) == (0x7FFD, 0, 0xFEFC, 0)
    mm, vs = D.stream_params(D.OS94, 220, 0x64, 3)
    assert (mm[1], vs[1]) == (0x7E26, 1)
    mm, vs = D.stream_params(D.OS94, 0x67, 0x64, 3)
    assert (mm[1], vs[1]) == (0x6D56, 4)


def test_stream_writer_is_deterministic_and_seed_sensitive():
    a = D.synth_stream(D.FMT_94_T1_S3, 50, seed=1)
    assert a == D.synth_stream(D.FMT_94_T1_S3, 50, seed=1)
    assert a != D.synth_stream(D.FMT_94_T1_S3, 50, seed=2)
    assert (a[0] << 8 | a[1]) == 50 and (a[2] & 0x80) and ((a[3] | a[4]) & 0x80)
    b = D.synth_stream(D.FMT_93A_T1, 9, seed=5, nbands=18)
    assert (b[2] & 0x80) and (b[2] & 0x1F) == 18


@pytest.mark.parametrize("handoff", [True, False])
@pytest.mark.parametrize("fpw", [4, 8, 16])
def test_chunk_plan_properties(fpw, handoff):
    def check(plan, jobs):
        flat = plan.reshape(-1)
        real = flat[(flat["flags"] & 0x81) == 0]
        # every job exactly once as a real slot
        assert np.array_equal(np.sort(real["job"]), np.arange(jobs.size))
        home = {}               # job -> (chunk, position) of its real slot
        last_live = {}
        for c, chunk in enumerate(plan):
            for pos, sl in enumerate(chunk):
                if not (sl["flags"] & 0x80):
                    last_live[c] = pos
                    if not (sl["flags"] & 1):
                        home[int(sl["job"])] = (c, pos)
        n_import = 0
        for c, chunk in enumerate(plan):
            pad = False
            for pos, sl in enumerate(chunk):
                if sl["flags"] & 0x80:
                    pad = True
                    continue
                assert not pad, "padding only at the end of a chunk"
                j = int(sl["job"])
                if not (sl["flags"] & 1):
                    prev = int(jobs[j]["prev"])
                    if prev == D.PREV_NONE:
                        assert sl["prevSlot"] == 0xFF and not (sl["flags"] & 8)
                    elif sl["flags"] & 8:
                        # tail through the hand-off buffer: published by the LAST frame of an EARLIER chunk
                        n_import += 1
                        pc, ppos = home[prev]
                        assert handoff and sl["prevSlot"] == 0xFF
                        assert pc < c and last_live[pc] == ppos and (plan[pc][ppos]["flags"] & 4)
                    else:
                        assert sl["prevSlot"] < pos and int(chunk[sl["prevSlot"]]["job"]) == prev
        return n_import

    b = workloads.build("mixed_16384", n_streams=24, n_frames=37)
    jobs = b["jobs"]
    plan = D.plan_chunks(jobs, fpw, b["srcs"], handoff=handoff)
    n_import = check(plan, jobs)
    halos_interleaved = int(((plan["flags"] & 1) != 0).sum())
    b2 = workloads.build("dcs93_4096", n_streams=5, n_frames=100)
    plan2 = D.plan_chunks(b2["jobs"], fpw, b2["srcs"], handoff=handoff)
    n_import2 = check(plan2, b2["jobs"])
    halos = int(((plan2["flags"] & 1) != 0).sum())
    if handoff:
        # along a chain the predecessor of a chunk's first frame is the last frame of the chunk before: no frame is
        # decoded twice, and the chunks are full
        assert halos == 0 and halos_interleaved == 0 and n_import > 0
        assert plan2.shape[0] == sum((100 + fpw - 1) // fpw for _ in range(5)) or plan2.shape[0] == (500 + fpw - 1) // fpw
        # the chunks stay in chain order (round 6: no consumer waits for its tail any more, so nothing is reordered): the chunks that
        # import nothing are the streams' first ones (one per stream, fewer where a chunk holds the end of one stream and the
        # start of the next), and a chunk that imports does so from the chunk right before it
        assert n_import2 > 0
        firsts = [c for c, chunk in enumerate(plan2) if not any((sl["flags"] & 0x88) == 8 for sl in chunk)]
        assert 1 <= len(firsts) <= 5 and firsts[0] == 0
        for c, chunk in enumerate(plan2):
            for sl in chunk:
                if (sl["flags"] & 0x88) == 8:
                    assert c not in firsts
    else:
        # stream-contiguous order needs one halo per chunk at most, and so does a job list that interleaves the
        # streams frame by frame: the planner follows the chains
        assert n_import == 0
        assert 0 < halos <= plan2.shape[0]
        assert 0 < halos_interleaved <= plan.shape[0]


@pytest.mark.parametrize("workload", ["mixed_16384", "dcs94_65536"], ids=["mixed-layouts", "all-1994+"])
@pytest.mark.parametrize("fpw", [4, 8, 16])
def test_chunk_packages_hold_exactly_what_round_0_needs(fpw, workload):
    """the host packer (what a batch uploads): per slot the slot word, descriptor head, pool offset and aligned header; per-lane
    split records (4 bytes each when every source is a 1994+ frame, else 8) and a bit-pool image from which every frame's bits
    can be read back at the position its slot names"""
    b = workloads.build(workload, n_streams=18, n_frames=29)
    blob, srcs, jobs = bytes(b["blob"]), b["srcs"], b["jobs"]
    pk = D.pack_chunks(blob, srcs, jobs, fpw)
    plan = D.plan_chunks(jobs, fpw, srcs)
    assert pk.shape[0] == plan.shape[0]
    sub = 64 // fpw
    split4 = bool((srcs["format"] >= D.FMT_94_T0).all())
    assert split4 == (workload == "dcs94_65536")
    sb = 4 if split4 else 8
    off_split = fpw * 80
    off_pool = (fpw * 80 + 64 * sb + 127) // 128 * 128
    # the image of the bit pool is as long as the plan's fullest chunk needs (a multiple of 128 bytes, at most the pool)
    pool_cap = max(fpw * 56, 320)
    img_dw = (pk.shape[1] - off_pool) // 4
    assert pk.shape[1] == off_pool + img_dw * 4 and img_dw % 32 == 0 and 32 <= img_dw <= pool_cap
    fullest = 0
    src_bytes = srcs.view(np.uint8).reshape(srcs.size, -1)
    seen = 0
    n_shared, n_93a, n_export = [0], [0], [0]
    flat_plan = {}
    for c in range(pk.shape[0]):
        pool = pk[c, off_pool:].view("<u4")
        for s in range(fpw):
            blk = pk[c, 80 * s: 80 * s + 80]
            slot = blk[:16].view("<u4")
            flags = (int(slot[1]) >> 8) & 0xFF
            assert int(slot[0]) == int(plan[c, s]["job"]) and flags == int(plan[c, s]["flags"])
            if flags & 0x80:
                assert not blk[16:56].any() and not blk[64:].any()
                continue
            job = jobs[int(slot[0])]
            assert int(slot[2]) == int(job["firstSrc"]) and ((int(slot[1]) >> 16) & 0xFF) == int(job["nSrc"])
            if job["nSrc"] == 0:
                continue
            sd = srcs[int(job["firstSrc"])]
            # descriptor head: the first 40 bytes of the DcsSrcDesc
            assert np.array_equal(blk[16:56], src_bytes[int(job["firstSrc"]), :40])
            pool_off = int(blk[56:58].view("<u2")[0])
            bpl = int(blk[58])
            assert blk[59] == 0
            # bytes 60..63: the job this frame's tail is due at, where the frame hands its tail to another chunk (DCS_SLOT_EXPORT)
            next_job = int(blk[60:64].view("<u4")[0])
            if flags & 0x04:
                assert int(jobs[next_job]["prev"]) == int(slot[0]), "the exporter's nextJob names the frame that follows it"
                n_export[0] += 1
            else:
                assert next_job == 0
            # header: the bytes behind the U16 frame count
            so, hl = int(sd["streamOff"]), int(sd["hdrLen"])
            want_hdr = np.zeros(16, np.uint8); want_hdr[:hl] = np.frombuffer(blob[so + 2: so + 2 + hl], np.uint8)
            assert np.array_equal(blk[64:80], want_hdr)
            # per-lane records: lane q's first band (bits 12..15 of the state word; bit 15 of bitDelta = no bands) with
            # the split record of that band's start.  The lanes cover the bands contiguously, in order: bpl bands each for
            # the 1993 layouts; in a 1994+ frame (bands of 7, 8, 13 x 16 and 32 samples) bands 0 and 1 count as one and
            # band 15 as two, so every lane after the first starts one band later; with one band per lane the last lane
            # takes the second half of band 15.
            nb16 = min(int(sd["idx"]["nBands"]), 16)
            is94 = int(sd["format"]) >= D.FMT_94_T0
            is93a = int(sd["format"]) == D.FMT_93A_T1
            nb_end = min(int(sd["idx"]["nBands"]), 18) if is93a else nb16
            mid = sd["idx"]["split"][14]
            shared15 = is94 and bpl == 1 and nb16 == 16 and int(mid["prv"]) != 0
            bases = [0]
            for q in range(1, sub):
                lane = s + q * fpw
                raw = pk[c, off_split + sb * lane: off_split + sb * lane + sb].copy()
                if split4:
                    # bitDelta | state << 16: the 8-byte record without its prv / prvDelta words
                    rec = np.zeros(8, np.uint8); rec[0:2] = raw[0:2]; rec[6:8] = raw[2:4]
                else:
                    rec = raw
                if bpl == 0:
                    assert not rec.any()
                    continue
                r16 = rec.view("<u2")
                if q == sub - 1 and shared15:
                    # the second half of band 15: where the index pass saw the first code with half of the samples done
                    assert list(r16) == [int(mid["prv"]), 0, 0, (int(mid["prvDelta"]) & 0x3FF) | 0x800 | (15 << 12)]
                    assert 0 < int(mid["prv"]) < int(sd["idx"]["nBits"]) and int(mid["prv"]) > int(mid["bitDelta"])
                    bases.append(15)
                    continue
                if r16[0] & 0x8000:
                    assert int(r16[0]) == 0x8000 and not rec[2:].any()
                    bases.append(nb_end)
                    continue
                base = int(r16[3]) >> 12
                if is93a and (int(r16[3]) & 0x200):
                    # OS93a Type 1, a lane that starts at band 16 or 17: the record from the frame record's bandType bytes
                    base += 16
                    r16[3] &= 0x0DFF
                    assert np.array_equal(rec[:6], np.asarray(sd["idx"]["bandType"][(base - 16) * 8:(base - 16) * 8 + 6], np.uint8))
                    assert int(r16[3]) == (int(sd["idx"]["bandType"][(base - 16) * 8 + 6]) | int(sd["idx"]["bandType"][(base - 16) * 8 + 7]) << 8) & 0x0DFF
                else:
                    sp = sd["idx"]["split"][base - 1]
                    want16 = [int(sp["bitDelta"]), 0 if split4 else int(sp["prv"]), 0 if split4 else int(sp["prvDelta"]),
                              (int(sp["state"]) & 0x0FFF) | (base << 12)]
                    assert list(r16) == want16
                bases.append(base)
            if bpl:
                if is93a:
                    def start(q):
                        return (2 * q if q < 2 else q + 2) if bpl == 1 else (3 * q if q < 2 else 2 * q + 2) if bpl == 2 else (0, 7, 11, 14)[q]
                    want = [min(start(q), nb_end) for q in range(sub)]
                    n_93a[0] += 1
                else:
                    want = [min(q * bpl + (1 if is94 and q else 0), nb16) for q in range(sub)]
                if shared15:
                    want[sub - 1] = 15
                    n_shared[0] += 1
                assert bases[:sub] == want
            # the frame's bits, read MSB-first from the pool image at the slot's position, are the stream's
            bit0 = (so + 2 + hl) * 8 + int(sd["idx"]["bitOff"])
            nbits = int(sd["idx"]["nBits"])
            fullest = max(fullest, pool_off + ((bit0 & 31) + nbits + 31) // 32 + 3)
            for k in (0, nbits // 2, max(nbits - 1, 0)):
                pos = (bit0 & 31) + k
                got = (int(pool[pool_off + (pos >> 5)]) >> (31 - (pos & 31))) & 1
                byte = blob[(bit0 + k) >> 3] if ((bit0 + k) >> 3) < len(blob) else 0
                assert got == (byte >> (7 - ((bit0 + k) & 7))) & 1
            seen += 1
    assert seen >= jobs.size
    assert img_dw == min(pool_cap, max(32, (fullest + 31) // 32 * 32))      # exactly the fullest chunk's runs, rounded up
    assert (n_shared[0] > 0) == (fpw == 4)          # one band per lane: band 15 of the 1994+ frames goes to two lanes
    assert (n_93a[0] > 0) == (workload == "mixed_16384")


def test_resident_plan_shortens_packages_but_opens_no_further_generation():
    """dcsPlanChunksCapped (what dcs_batch_create plans with): the default workload's 65 536 frames stay 8 192 chunks -- exactly two
    generations of an MI355X's 4 096 wavefront places; a plan capped at what 97 % of the chunks need would make 8 216 of them and a
    third generation (measured: 36.2 us against 33.4) -- with the image sized by the fullest chunk and 4-byte split records; a batch
    that is well inside one generation takes the capped plan: more chunks than frames / 8, shorter packages, fewer bytes in all"""
    b = workloads.build("survey3_65536")
    pk = D.pack_chunks(bytes(b["blob"]), b["srcs"], b["jobs"], 8)
    assert pk.shape[0] == 8192
    img_dw = (pk.shape[1] - 896) // 4
    assert pk.shape[1] == 896 + img_dw * 4 and img_dw % 32 == 0 and 256 <= img_dw < 448
    small = workloads.build("survey3_65536", n_streams=40)                 # 10 240 frames: 1 280 chunks, a third of a generation
    pk2 = D.pack_chunks(bytes(small["blob"]), small["srcs"], small["jobs"], 8)
    n_min = small["jobs"].size // 8
    assert n_min <= pk2.shape[0] <= n_min + n_min // 16
    if pk2.shape[0] > n_min:                                               # the cap was taken: it must have paid
        assert pk2.shape[1] < pk.shape[1] and pk2.size < n_min * pk.shape[1]
    jobs_seen = np.sort(np.concatenate([pk2[c, 80 * s: 80 * s + 4].view("<u4") for c in range(pk2.shape[0]) for s in range(8)]))
    jobs_seen = jobs_seen[jobs_seen != 0xFFFFFFFF]
    assert np.array_equal(jobs_seen, np.arange(small["jobs"].size, dtype=np.uint32))      # every frame once, halo-free


def test_workload_builders_shape():
    b = workloads.build("dcs93_4096")
    assert b["jobs"].size == 4096 and b["srcs"].size == 4096
    assert set(np.unique(b["jobs"]["xform"])) == {D.XFORM_93}
    payload = int(b["srcs"]["idx"]["nBits"].astype(np.int64).sum()) // 8
    assert 90 * 4096 < payload < 200 * 4096            # ~125 compressed bytes per frame


def _mixed_stream_set():
    streams = []
    for i, fmt in enumerate(ALL_FORMATS * 3):
        s = make_stream(fmt, 5 + 7 * i, seed=900 + i, profile=i % 4)
        if i % 5 == 4:
            s = corrupt(s, seed=i)
        streams.append((os_for(fmt, i), s, 0xFF, 0x64))
    return streams


@pytest.mark.parametrize("threads", [1, 3, 0])
def test_threaded_indexer_equals_single_stream_indexer(threads):
    streams = _mixed_stream_set()
    many = D.api.index_streams(streams, threads=threads)
    assert len(many) == len(streams)
    for (os_, data, _, _), (idx, info) in zip(streams, many):
        want_idx, want_info = D.index_stream(os_, data)
        assert idx.tobytes() == want_idx.tobytes()
        assert bytes(info) == bytes(want_info)


def test_batch_built_from_threaded_indexer_is_identical():
    streams = _mixed_stream_set()
    a = D.build_stream_batch(streams, extra_frames=2)
    b = D.build_stream_batch(streams, extra_frames=2, indexer=D.api.index_streams)
    assert a["blob"] == b["blob"]
    assert a["srcs"].tobytes() == b["srcs"].tobytes()
    assert a["jobs"].tobytes() == b["jobs"].tobytes()


def test_dcs93_4096_follows_survey_config_2_to_the_letter():
    """SURVEY 8(d) Config 2: 64 streams x 64 frames of OS93 Type 0; header bands 0-11 populated with scale codes uniform in 0x20..0x34,
    bands 12-15 empty (0xFF / 0x7F); one stream in ten carries the 0x40 stride bit on bands >= 6 (a strided Type-0 band spans 32 slots,
    so such a stream keeps the ten bands that fit the 255-slot frame); band-type codes 0 / 1-3 / 4-6 / 7-9 at 15 / 35 / 40 / 10 %"""
    from dcsexplorer_amd import workloads
    import dcsexplorer_amd as D
    streams = workloads.streams_dcs93_4096()
    assert len(streams) == 64
    codes = []
    for k, (os_, s, vol, lvl) in enumerate(streams):
        assert ((s[0] << 8) | s[1]) == 64 and not (s[2] & 0x80)                # 64 frames, Type 0
        hdr = s[2:18]
        strided = (k % 10) == 9
        populated = [b for b in range(16) if (hdr[b] & 0x7F) != 0x7F]
        assert populated == list(range(10 if strided else 12))
        for b in populated:
            assert 0x20 <= (hdr[b] & 0x3F) <= 0x34
            assert bool(hdr[b] & 0x40) == (strided and b >= 6)
        recs, info = D.index_stream(os_, s)
        assert info.nValidFrames == 64
        codes.append(len(s) / 64.0)
    assert 80 < sum(codes) / len(codes) < 140                                   # bytes per frame


def test_several_gpu_entries_fail_loudly_without_a_gpu(dcs):
    """dcs_node_create / dcs_decode_streams_sharded / dcs_device_numa_node on a box without a GPU: an error, no crash, no fallback"""
    if dcs.device_count() > 0:
        return
    assert dcs.device_numa_node(0) == -1 and dcs.bind_process_to_device_numa(0) is None
    try:
        dcs.Node([0, 0], depth=2)
    except dcs.DcsError as e:
        assert e.status == -2
    else:
        raise AssertionError("Node() succeeded without a GPU")
    s = dcs.synth_stream(dcs.FMT_94_T1_S3, 8, seed=1)
    try:
        dcs.decode_streams_sharded([0], [(dcs.OS95, s, 255, 0x64)])
    except dcs.DcsError as e:
        assert e.status == -2
    else:
        raise AssertionError("dcs_decode_streams_sharded succeeded without a GPU")
