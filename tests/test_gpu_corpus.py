"""BASELINE configs[4] on one MI355X: the reduced ragged corpus against the reference's committed per-stream hashes,
the several-GPU entry (two contexts on the one device of the test box), batches in flight, and lost tails."""
import json
import os
import sys

import numpy as np
import pytest

import dcsexplorer_amd as D
from dcsexplorer_amd import workloads
from util import ALL_FORMATS, make_stream, os_for

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def corpus():
    g = json.load(open(os.path.join(GOLD, "corpus_golden.json")))
    manifest = workloads.corpus_manifest(**g["corpus"])
    return g, manifest, workloads.corpus_streams(manifest)


def stream_hashes(oracle, pcm, first):
    return ["%016x" % oracle.fnv1a64(pcm[first[k]:first[k + 1]]) for k in range(len(first) - 1)]


def test_corpus_on_one_gpu_matches_reference_hashes(gpu_ctx, oracle, corpus):
    """29 titles x 20 streams, U[20, 2000] frames, all six layouts, one launch: every stream's PCM hash equals the
    hash of the unmodified reference's PCM (tests/golden/make_corpus_golden.py)"""
    g, manifest, streams = corpus
    assert g["streams"] == len(streams) == 580 and g["formats"] == [0, 1, 2, 3, 4, 5]
    b = D.build_stream_batch(streams, indexer=D.index_streams)
    assert int(b["jobs"].size) == g["frames"]
    batch = gpu_ctx.batch(b["blob"], b["srcs"], b["jobs"])
    batch.run()
    pcm, err = batch.download_view()
    assert not err.any()
    got = stream_hashes(oracle, pcm, b["first_job"])
    bad = [k for k in range(len(got)) if got[k] != g["stream_hashes"][k]]
    assert not bad, "streams %s differ from the reference" % bad[:8]
    assert "%016x" % oracle.fnv1a64(np.array([int(h, 16) for h in got], dtype=np.uint64)) == g["fnv1a64_of_stream_hashes"]
    batch.close()


def test_corpus_cut_for_several_ranks_decodes_to_the_same_pcm(gpu_ctx, oracle, corpus):
    """what rank r of N decodes (its frame-balanced range) is exactly its part of the one-GPU result"""
    g, manifest, streams = corpus
    counts = workloads.corpus_frames(manifest)
    for world in (2, 8):
        cut = D.partition_streams(counts, world)
        for r in (0, world - 1):
            part = streams[cut[r]:cut[r + 1]]
            pcm, err, first = gpu_ctx.decode_streams(part)
            assert not err.any()
            assert stream_hashes(oracle, pcm, first) == g["stream_hashes"][cut[r]:cut[r + 1]]


def test_sharded_entry_two_contexts_one_device(oracle, corpus):
    """dcs_decode_streams_sharded with the device list [0, 0]: two host threads, two contexts, the range partition and
    the disjoint output ranges of the several-GPU path, on the one GPU a test box has"""
    g, manifest, streams = corpus
    sub = streams[:120]
    pcm, err, first, cut = D.decode_streams_sharded([0, 0], sub, extra_frames=1)
    assert list(cut) == list(D.partition_streams([((s[1][0] << 8) | s[1][1]) + 1 for s in sub], 2))
    assert 0 < cut[1] < len(sub) and not err.any()
    ctx = D.Context(0)
    want, _, wfirst = ctx.decode_streams(sub, extra_frames=1)
    ctx.close()
    assert np.array_equal(first, wfirst) and np.array_equal(pcm, want)
    # a frame of every stream without its taper frame hashes like the reference's
    k = 5
    nf = (sub[k][1][0] << 8) | sub[k][1][1]
    assert "%016x" % oracle.fnv1a64(pcm[first[k]:first[k] + nf]) == g["stream_hashes"][k]


def test_sharded_entry_survives_a_concurrent_cache_release(corpus):
    """the node a sharded call runs on is held while the call uses it (it used to be a raw pointer looked up under a
    mutex that was dropped before the call): one thread decodes the same list over [0, 0] and [0, 0, 0] again and again while another
    releases the node cache all the time -- every call returns the same PCM and nothing crashes"""
    import threading
    g, manifest, streams = corpus
    sub = streams[40:72]
    want, werr, wfirst, _ = D.decode_streams_sharded([0, 0], sub, extra_frames=1)
    stop, bad = threading.Event(), []

    def releaser():
        while not stop.is_set():
            D.node_cache_release()

    th = threading.Thread(target=releaser)
    th.start()
    try:
        for k in range(12):
            devs = [0, 0] if k % 2 == 0 else [0, 0, 0]
            pcm, err, first, cut = D.decode_streams_sharded(devs, sub, extra_frames=1)
            if not (np.array_equal(pcm, want) and np.array_equal(err, werr) and np.array_equal(first, wfirst)):
                bad.append(k)
    finally:
        stop.set()
        th.join()
        D.node_cache_release()
    assert not bad


@pytest.mark.parametrize("fpw", [4, 8, 16])
def test_device_packer_lays_out_the_same_packages_as_the_host_packer(gpu_ctx, fpw):
    """the packer on the device (DCS_PIPE_PACK_ON_DEVICE) against the host packer: byte for byte the same packages,
    on a batch of all six layouts, damaged streams included"""
    from util import corrupt
    streams = []
    for f in ALL_FORMATS:
        for k in range(3):
            s = make_stream(f, 37 + 5 * k, seed=61000 + 8 * f + k, profile=k)
            if k == 2:
                s = corrupt(s, 9 + f, nflips=3) + bytes(512)
            streams.append((os_for(f, k), s, 240, 0x60 + k))
    b = D.build_stream_batch(streams, extra_frames=2)
    host = D.pack_chunks(b["blob"], b["srcs"], b["jobs"], fpw)
    dev = gpu_ctx.pack_chunks_device(b["blob"], b["srcs"], b["jobs"], fpw)
    assert host.shape == dev.shape
    bad = np.argwhere(host != dev)
    assert bad.size == 0, "first difference: chunk %d byte %d" % (bad[0][0], bad[0][1])


@pytest.mark.parametrize("on_device", [0, 1, 2, 3], ids=["host-index", "device-index", "device-index-and-pack", "device-index-plan-and-pack"])
def test_pipeline_returns_lists_in_order(gpu_ctx, corpus, on_device):
    """dcs_pipeline: several lists in flight come back in submission order with the PCM of dcs_decode_streams, whether
    the index pass runs on the host pool or on the device"""
    g, manifest, streams = corpus
    lists = [streams[0:40], streams[40:45], streams[45:140], streams[140:141], streams[141:200]]
    want = [gpu_ctx.decode_streams(l, extra_frames=2) for l in lists]
    pipe = gpu_ctx.pipeline(3, index_on_device=on_device >= 1, pack_on_device=on_device >= 2, plan_on_device=on_device == 3)
    got = []
    for k, l in enumerate(lists):
        pipe.submit(l, extra_frames=2)              # (blocks while 3 lists are in flight)
        if k >= 2:
            pcm, err, first, _, _ = pipe.collect()
            got.append((pcm.copy(), err.copy(), first.copy()))
    while len(got) < len(lists):
        pcm, err, first, _, _ = pipe.collect()
        got.append((pcm.copy(), err.copy(), first.copy()))
    pipe.close()
    for (p, e, f), (wp, we, wf) in zip(got, want):
        assert np.array_equal(f, wf) and np.array_equal(e, we) and np.array_equal(p, wp)


@pytest.mark.parametrize("fpw", [4, 8, 16])
def test_tails_meet_whatever_the_chunk_order(gpu_ctx, oracle, fpw):
    """Tails between chunks are a rendezvous (round 6): producer and consumer each exchange one word per tail sample on the
    producing chunk's row, whoever arrives second finishes the consumer's first samples; nobody waits, so no order of the chunks
    can matter.  Test hook: the planner's chunks in a seeded random order (consumers dispatched long before their producers and
    the other way round), no XCD ranges: resident batches relaunched, the one-shot call and the live decoder, all bit-exact,
    no frame flagged."""
    streams = [(os_for(f, f), make_stream(f, 140 + 37 * f, seed=41000 + f, profile=f % 3), 240, 0x64) for f in ALL_FORMATS]
    want = np.concatenate([oracle.decode(os_, vol, [s], [lvl], (s[0] << 8) | s[1]) for os_, s, vol, lvl in streams])
    b = D.build_stream_batch(streams)
    plan = D.plan_chunks(b["jobs"], fpw, b["srcs"])
    assert any((s["flags"] & 0x08) and not (s["flags"] & 0x80) for s in plan.reshape(-1)), "no frame takes its tail from another chunk"
    gpu_ctx.set_frames_per_wave(fpw)
    try:
        for seed in (0, 1, 7, 12345):
            gpu_ctx.set_test_hooks(chunk_order_seed=seed, no_xcd_ranges=True)
            batch = gpu_ctx.batch(b["blob"], b["srcs"], b["jobs"])
            for k in range(6):
                batch.run()
                if k in (0, 5):
                    pcm, err = batch.download()[:2]
                    assert not err.any(), "seed %d launch %d" % (seed, k)
                    bad = np.nonzero((pcm != want).any(axis=1))[0]
                    assert bad.size == 0, "seed %d launch %d: frames %s" % (seed, k, bad[:8])
            batch.close()
            pcm2, err2 = gpu_ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
            assert not err2.any() and np.array_equal(pcm2, want), "one-shot, seed %d" % seed
    finally:
        gpu_ctx.set_test_hooks(0, False)
        gpu_ctx.set_frames_per_wave(0)


def test_batch_outlives_caller_stream_ordering(gpu_ctx, oracle):
    """a run on a caller's own non-blocking stream, then download without any synchronisation by the caller: the batch
    waits for its own last launch"""
    import torch
    streams = [(D.OS95, make_stream(D.FMT_94_T1_S3, 400, seed=777, profile=0), 255, 0x64)]
    want = oracle.decode(D.OS95, 255, [streams[0][1]], [0x64], 400)
    b = D.build_stream_batch(streams)
    st = torch.cuda.Stream()
    for _ in range(5):
        batch = gpu_ctx.batch(b["blob"], b["srcs"], b["jobs"])
        batch.run_many(3, st.cuda_stream)
        pcm, err = batch.download()
        assert np.array_equal(pcm, want) and not err.any()
        batch.close()                               # destroy right away: buffers are recycled only once the launch is done


@pytest.mark.parametrize("pack_on_device", [0, 1, 2], ids=["host-pack", "device-pack", "device-plan-and-pack"])
def test_pipeline_device_index_with_damaged_and_rom_sized_buffers(gpu_ctx, oracle, pack_on_device):
    """device index pass: a stream handed over with a buffer far longer than the stream (the rest of a ROM image) is cut
    to what it can use; a truncated stream (runs past its buffer, the missing bytes read as zero) sends its list down
    the host path; corrupted streams keep the reference's error semantics.  PCM equals dcs_decode_streams'."""
    from util import corrupt
    base = [(os_for(f, 1), make_stream(f, 60 + 5 * f, seed=52000 + f, profile=f % 3), 230, 0x66) for f in ALL_FORMATS]
    rom_sized = [(o, s + bytes(200000), v, l) for o, s, v, l in base[:3]] + base[3:]
    truncated = base[:5] + [(base[5][0], base[5][1][:len(base[5][1]) // 2], base[5][2], base[5][3])]
    damaged = [(o, corrupt(s, 77 + k, nflips=4), v, l) for k, (o, s, v, l) in enumerate(base)]
    pipe = gpu_ctx.pipeline(3, index_on_device=True, pack_on_device=pack_on_device >= 1, plan_on_device=pack_on_device == 2)
    for extra in (2, 0):                # (without taper frames a stream's last frame and the next one's first share chunks)
        for lst in (rom_sized, truncated, damaged):
            want = gpu_ctx.decode_streams(lst, extra_frames=extra)
            pipe.submit(lst, extra_frames=extra)
            pcm, err, first, _, _ = pipe.collect()
            assert np.array_equal(first, want[2]) and np.array_equal(err, want[1]) and np.array_equal(pcm, want[0])
    pipe.close()


@pytest.mark.parametrize("pack_on_device", [0, 1, 2], ids=["host-pack", "device-pack", "device-plan-and-pack"])
def test_pipeline_device_index_stream_that_announces_far_more_than_it_holds(gpu_ctx, oracle, pack_on_device):
    """the LAST stream of a small list claims 65 535 frames, and its header makes the zero bits behind its few bytes parse
    as valid frames (1993 Type 0: "no sub-type change, code 0" is five zero bits per band), so the device walk goes on for
    650 KB past the stream: it must read those bytes as zero WITHOUT touching memory behind the stream (the pipeline's
    launch has no allocation bound: stream locations are absolute addresses).  Same PCM as the synchronous call."""
    good = [(os_for(f, 1), make_stream(f, 20 + f, seed=88000 + f), 255, 0x64) for f in ALL_FORMATS[:3]]
    runaway = bytes([0xFF, 0xFF]) + bytes([0x28] * 12 + [0x7F] * 4) + bytes(6)
    lst = good + [(D.OS93B, runaway, 255, 0x64)]
    want = gpu_ctx.decode_streams(lst, extra_frames=2)
    assert want[0].shape[0] == sum(20 + f + 2 for f in range(3)) + 65535 + 2
    pipe = gpu_ctx.pipeline(2, index_on_device=True, pack_on_device=pack_on_device >= 1, plan_on_device=pack_on_device == 2)
    for _ in range(2):
        pipe.submit(lst, extra_frames=2)
    for _ in range(2):
        pcm, err, first, _, _ = pipe.collect()
        assert np.array_equal(first, want[2]) and np.array_equal(err, want[1]) and np.array_equal(pcm, want[0])
    pipe.close()
    # ... and through the one-shot device index entry: the records equal the host walk's
    recs_h, info_h = D.index_stream(D.OS93B, runaway)
    (recs_d, info_d), = gpu_ctx.index_streams_gpu([(D.OS93B, runaway)])
    assert info_d.nValidFrames == info_h.nValidFrames == 65535 and info_d.nBytes == info_h.nBytes
    assert recs_d.tobytes() == recs_h.tobytes()


def test_saturated_frames_fill_the_bit_pool_and_the_device_planner_plans_again_or_hands_the_list_back(gpu_ctx, oracle):
    """synth profile 4: every band at its widest code in every frame, ~500 bytes a frame where the workloads have ~100.  The
    chunks' compressed bytes then no longer fit the kernel's bit pool: the host planner closes such chunks early, the
    device planner (arithmetic chunks) flags the list and the pipeline decodes it through the host planner -- same PCM as
    the oracle's either way, with 4, 8 and 16 frames per wavefront; DcsPipelineResult.path tells which stages ran where."""
    lst = [(os_for(f, f & 1), make_stream(f, 48 + f, seed=66000 + f, profile=4), 255, 0x64) for f in ALL_FORMATS]
    assert max(len(s) / (48.0 + f) for f, (_, s, _, _) in zip(ALL_FORMATS, lst)) > 480
    easy = [(os_for(f, f & 1), make_stream(f, 48 + f, seed=66100 + f), 255, 0x64) for f in ALL_FORMATS]
    want = np.concatenate([oracle.decode(o, v, [s], [l], ((s[0] << 8) | s[1]) + 2) for o, s, v, l in lst])
    try:
        for fpw in (4, 8, 16):
            gpu_ctx.set_frames_per_wave(fpw)
            pcm, err, _ = gpu_ctx.decode_streams(lst, extra_frames=2)
            assert np.array_equal(pcm, want) and not err.any(), "fpw %d" % fpw
            for mode in range(4):
                pipe = gpu_ctx.pipeline(2, index_on_device=mode >= 1, pack_on_device=mode >= 2, plan_on_device=mode == 3)
                pipe.submit(lst, extra_frames=2)
                pipe.submit(easy, extra_frames=2)
                pcm, err, _, _, _ = pipe.collect()
                assert np.array_equal(pcm, want) and not err.any(), "fpw %d mode %d" % (fpw, mode)
                hard_path = pipe.last_path
                pipe.collect()
                easy_path = pipe.last_path
                pipe.close()
                assert easy_path == (0, 1, 3, 7)[mode]
                if mode == 3:
                    # planned again on the device with three quarters, then half of the slots per chunk; where even that overflows
                    # the pool (8 and 16 frames per wavefront: 448 bytes per frame) the list is handed back: host index pass, planner, packer
                    assert hard_path == (7 if fpw == 4 else 0)
    finally:
        gpu_ctx.set_frames_per_wave(0)


def test_large_frames_are_planned_again_on_the_device_with_fewer_frames_per_chunk(gpu_ctx, oracle):
    """frames of 260 to 330 bytes (ten of sixteen bands at their widest codes): eight of them overflow a chunk's bit pool (224 bytes
    per slot), so the arithmetic plan of the device planner flags the list -- and the pipeline plans it AGAIN on the device with six,
    then four frames per chunk instead of handing it to the host (round 4: one such stream among 600 used to cost a 600 000-frame list
    the device path).  DcsPipelineResult.path says 7; the PCM is the oracle's."""
    lst = [(os_for(f, f & 1), make_stream(f, 60 + f, seed=67000 + f, profile=4, nbands=10), 255, 0x64) for f in (0, 1, 3)]
    lst += [(os_for(f, 1), make_stream(f, 40, seed=67100 + f), 255, 0x64) for f in ALL_FORMATS]
    assert 280 < max(len(s) / (((s[0] << 8) | s[1]) + 0.0) for _, s, _, _ in lst) < 340
    want = np.concatenate([oracle.decode(o, v, [s], [l], ((s[0] << 8) | s[1]) + 2) for o, s, v, l in lst])
    try:
        for fpw in (8, 16):
            gpu_ctx.set_frames_per_wave(fpw)
            pipe = gpu_ctx.pipeline(2, index_on_device=True, pack_on_device=True, plan_on_device=True)
            pipe.submit(lst, extra_frames=2)
            pcm, err, _, _, _ = pipe.collect()
            path = pipe.last_path
            pipe.close()
            assert np.array_equal(pcm, want) and not err.any(), "fpw %d" % fpw
            assert path == 7, "fpw %d: path %d" % (fpw, path)
    finally:
        gpu_ctx.set_frames_per_wave(0)


@pytest.mark.parametrize("mode", [0, 1, 2, 3], ids=["host-index", "device-index", "device-index-and-pack", "device-index-plan-and-pack"])
def test_pipeline_reports_a_bad_list_and_carries_on(gpu_ctx, oracle, mode):
    """a list with an unusable stream (zero frames) comes back with an error status in its turn; the lists around it are
    decoded as if nothing had happened; destroying a pipeline with lists still in flight finishes them first"""
    good = [(os_for(f, 0), make_stream(f, 30 + f, seed=71000 + f), 255, 0x64) for f in ALL_FORMATS]
    bad = good[:2] + [(D.OS94, bytes([0, 0]) + bytes(40), 255, 0x64)] + good[2:]
    want = gpu_ctx.decode_streams(good)
    pipe = gpu_ctx.pipeline(4, index_on_device=mode >= 1, pack_on_device=mode >= 2, plan_on_device=mode == 3)
    pipe.submit(good)
    pipe.submit(bad)
    pipe.submit(good)
    pcm, err, first, _, _ = pipe.collect()
    assert np.array_equal(pcm, want[0]) and np.array_equal(first, want[2])
    with pytest.raises(D.DcsError) as e:
        pipe.collect()
    assert e.value.status == D.api.ERR_BAD_STREAM
    pcm, err, first, _, _ = pipe.collect()
    assert np.array_equal(pcm, want[0])
    pipe.submit(good)                       # still in flight when the pipeline is closed
    pipe.close()


def test_sharded_entry_reports_errors():
    """bad arguments and an unusable device come back as status codes, not crashes"""
    good = [(D.OS95, make_stream(D.FMT_94_T1_S3, 20, seed=5), 255, 0x64)]
    with pytest.raises(D.DcsError):
        D.decode_streams_sharded([0, 99], good * 4)         # device 99 does not exist
    with pytest.raises(D.DcsError):
        D.decode_streams_sharded([0], [(D.OS94, bytes([0, 0, 0, 0]), 255, 0x64)])    # zero frames
    pcm, err, first, cut = D.decode_streams_sharded([0, 0, 0, 0], good)              # more devices than streams
    assert pcm.shape[0] == 20 and list(cut)[0] == 0 and list(cut)[-1] == 1


@pytest.mark.parametrize("mode", [0, 1, 2, 3], ids=["host-index", "device-index", "device-index-and-pack", "device-index-plan-and-pack"])
def test_pipeline_soak(gpu_ctx, mode):
    """a few hundred lists through 12 slots in flight: every list's PCM is checked (tools/pipe_soak.py runs the same for
    thousands of lists and watches the memory)"""
    import zlib
    base = workloads.streams_mixed_16384(n_streams=24, n_frames=40)
    variants = [base[i:] + base[:i] for i in (0, 5, 11)]
    want = [zlib.crc32(gpu_ctx.decode_streams(v)[0].tobytes()) for v in variants]
    refs = [D.make_refs(v) for v in variants]
    pipe = gpu_ctx.pipeline(12, index_on_device=mode >= 1, pack_on_device=mode >= 2, plan_on_device=mode == 3)
    n, done, bad = 240, 0, 0
    for k in range(n):
        pipe.submit_refs(refs[k % 3][0], len(variants[k % 3]))
        if k >= 11:
            pcm, err, _, _, _ = pipe.collect()
            bad += zlib.crc32(pcm.tobytes()) != want[done % 3] or bool(err.any()); done += 1
    while done < n:
        pcm, err, _, _, _ = pipe.collect()
        bad += zlib.crc32(pcm.tobytes()) != want[done % 3] or bool(err.any()); done += 1
    pipe.close()
    assert bad == 0


@pytest.mark.parametrize("on_device", [2, 1, 0], ids=["walk-shared-by-host-and-device", "parts-on-device", "parts-behind-host-index"])
def test_one_call_on_a_large_list_goes_through_the_pipeline_in_parts(gpu_ctx, oracle, corpus, on_device):
    """dcs_decode_streams cuts a large list into eight parts that go through the context's own pipeline -- index walk,
    planner and packer on the device with the host pool walking the first parts next to the index kernel (the default),
    everything on the device, or behind the host pool's index pass (dcs_ctx_set_large_list_path 2, 1, 0);
    same PCM, error words and frame offsets as the one-batch path, including taper frames, a damaged stream in the middle
    and a truncated one (whose part the device stages hand back to the host's)"""
    from util import corrupt
    g, manifest, streams = corpus
    streams = list(streams[:200])
    k = 77
    streams[k] = (streams[k][0], corrupt(streams[k][1], 3, nflips=5) + bytes(2048), streams[k][2], streams[k][3])
    k = 150
    streams[k] = (streams[k][0], streams[k][1][:len(streams[k][1]) // 2], streams[k][2], streams[k][3])
    gpu_ctx.set_large_list_path(on_device)
    try:
        # (the second call finds the pipeline made; with the walk shared, the number of parts the host takes moves from call to
        # call -- every split must give the same PCM)
        for _ in range(2 if on_device != 2 else 6):
            pcm, err, first = gpu_ctx.decode_streams(streams, extra_frames=2)    # > 32 768 frames: in parts
            if on_device == 2:
                if _ == 0:
                    first_pcm, first_err = pcm.copy(), err.copy()
                else:
                    assert np.array_equal(pcm, first_pcm) and np.array_equal(err, first_err)
    finally:
        gpu_ctx.set_large_list_path(2)
    b = D.build_stream_batch(streams, extra_frames=2, indexer=D.index_streams)
    want, werr = gpu_ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])           # one batch, one launch
    assert np.array_equal(first, b["first_job"])
    assert np.array_equal(err, werr) and np.array_equal(pcm, want)
    j = 5
    nf = int(first[j + 1] - first[j]) - 2
    assert "%016x" % oracle.fnv1a64(pcm[first[j]:first[j] + nf]) == g["stream_hashes"][j]


def test_bench_two_ranks_on_one_gpu_runs_launcher_ranks_and_kernel(tmp_path):
    """bench.py --gpus 2 --share-gpu: the launcher starts its two ranks (gloo carries barrier and max; both decode on GPU 0),
    every rank times its K launches, and every rank runs a pipeline of its own for the N-rank end_to_end figure: the line
    says n_gpus 2, carries per-rank host figures, and EVERY rank's PCM equals the reference's hashes of its range"""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1",
                        "--e2e-device-depth", "8", "--e2e-lists", "16"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["bit_exact"] is True and "share_gpu" in out
    assert out["bit_exact_ranks"] == [True, True]           # every rank held ITS streams' PCM against the reference's hashes
    assert out["value"] > 1e9 and out["config"]["frames_all_ranks_per_step"] == 2 * out["config"]["frames_rank0_per_step"]
    e = out["end_to_end"]
    assert e["sustained"]["ranks"] == 2 and len(e["per_rank"]) == 2 and e["usable_cpus"] >= 1
    assert all(p["worker_host_ms"] > 0 and p["cpu_ms_per_list"] > 0 for p in e["per_rank"])
    assert e["sustained"]["value"] > 1e9


def test_full_size_corpus_through_the_pipeline_matches_reference_hashes(gpu_ctx, oracle):
    """BASELINE configs[4] at SURVEY 8(d) Config 5's size: 29 titles x 600 streams of U[20, 2000] frames = 17 667 184 frames,
    all six layouts.  One title per list through dcs_pipeline (index walk, planner and packer on the device, two lists in
    flight); EVERY stream's PCM hash equals the hash of the unmodified reference's PCM (tests/golden/corpus_golden_full.json,
    made by tests/golden/make_corpus_golden.py --full).  The reference loop is DCSExplorer.cpp:1628-1907."""
    from concurrent.futures import ThreadPoolExecutor
    g = json.load(open(os.path.join(GOLD, "corpus_golden_full.json")))
    spec = g["corpus"]
    assert (spec["titles"], spec["streams_per_title"]) == (29, 600) and g["streams"] == 17400 and g["frames"] == 17667184
    manifest = workloads.corpus_manifest(**spec)
    per = spec["streams_per_title"]
    pipe = gpu_ctx.pipeline(2, index_on_device=True, pack_on_device=True, plan_on_device=True)
    pool = ThreadPoolExecutor(max_workers=min(16, D.host_threads()))
    frames, bad, on_device = 0, [], 0

    def check(t):
        nonlocal frames, on_device
        pcm, err, first, _, _ = pipe.collect()
        on_device += int(pipe.last_path == 7)
        assert not err.any(), "title %d: error flags" % t
        got = list(pool.map(lambda k: "%016x" % oracle.fnv1a64(pcm[first[k]:first[k + 1]]), range(per)))
        want = g["stream_hashes"][t * per:(t + 1) * per]
        bad.extend(t * per + k for k in range(per) if got[k] != want[k])
        frames += int(first[-1])

    for t in range(spec["titles"]):
        pipe.submit(workloads.corpus_streams(manifest, t * per, (t + 1) * per))     # (blocks while two lists are in flight)
        if t >= 1:
            check(t - 1)
    check(spec["titles"] - 1)
    pipe.close()
    pool.shutdown()
    assert not bad, "%d streams differ from the reference, first %s" % (len(bad), bad[:8])
    assert frames == g["frames"]
    assert on_device >= spec["titles"] - 2, "only %d of %d lists took the device path" % (on_device, spec["titles"])


@pytest.mark.parametrize("extra", [0, 2])
def test_device_path_resident_streams_to_resident_pcm(gpu_ctx, oracle, corpus, extra):
    """dcs_device_path: the streams uploaded once, then index walk, planner, packer and decode kernels back to back with
    nothing crossing PCIe -- the PCM of dcs_decode_streams (and therefore the reference's), on the first pass and after
    timed passes; per-kernel times come back positive and add up to about a pass"""
    g, manifest, streams = corpus
    part = streams[100:180]                     # ragged, several layouts
    want_pcm, want_err, want_first = gpu_ctx.decode_streams(part, extra_frames=extra)
    path = gpu_ctx.device_path(part, extra_frames=extra)
    pcm, err, first = path.download()
    assert np.array_equal(first, want_first) and np.array_equal(err, want_err) and np.array_equal(pcm, want_pcm)
    t = path.run(6)
    assert t["planFlags"] == 0 and t["nStreams"] == len(part) and t["nFrames"] == want_pcm.shape[0]
    assert t["indexMs"] > 0 and t["planMs"] > 0 and t["packMs"] > 0 and t["decodeMs"] > 0
    assert 0.5 * t["passMs"] < t["indexMs"] + t["planMs"] + t["packMs"] + t["decodeMs"] < 2.0 * t["passMs"]
    pcm, err, first = path.download()
    assert np.array_equal(pcm, want_pcm) and not err.any()
    if extra == 0:
        assert stream_hashes(oracle, pcm, first) == g["stream_hashes"][100:180]
    path.close()


def _rss_kb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS:"):
            return int(line.split()[1])
    return 0


def test_node_two_contexts_on_one_device_200_lists_in_submission_order_memory_flat(gpu_ctx):
    """dcs_node with the device list [0, 0]: two persistent contexts with a pipeline each, lists dealt by frames in flight;
    200 lists of unlike sizes come back in SUBMISSION order with the single-context PCM, both contexts take their share, and
    the process's resident memory does not grow over the second hundred"""
    import zlib
    base = workloads.streams_mixed_16384(n_streams=30, n_frames=48)
    variants = [base[:30], base[3:14], base[10:30] + base[:6], base[5:6]]
    want = [zlib.crc32(gpu_ctx.decode_streams(v, extra_frames=1)[0].tobytes()) for v in variants]
    refs = [D.make_refs(v) for v in variants]
    node = D.Node([0, 0], depth=6)
    assert node.n_devices == 2 and node.device_info(0)[0] == 0 and node.device_info(1)[0] == 0
    n, inflight, done, bad, by_dev = 200, 10, 0, 0, [0, 0]
    rss_mid = None
    for k in range(n):
        v = (k * 7 + k // 5) % 4
        node.submit_refs(refs[v][0], len(variants[v]), extra_frames=1)
        if k >= inflight - 1:
            pcm, err, first, _, _, dev = node.collect()
            w = (done * 7 + done // 5) % 4
            bad += zlib.crc32(pcm.tobytes()) != want[w] or bool(err.any()) or len(first) != len(variants[w]) + 1
            by_dev[dev] += 1; done += 1
            if done == n // 2:
                rss_mid = _rss_kb()
    while done < n:
        pcm, err, first, _, _, dev = node.collect()
        w = (done * 7 + done // 5) % 4
        bad += zlib.crc32(pcm.tobytes()) != want[w] or bool(err.any())
        by_dev[dev] += 1; done += 1
    rss_end = _rss_kb()
    assert bad == 0
    assert min(by_dev) >= n // 5, "lists per context: %s" % by_dev
    assert [node.device_info(i)[2] for i in range(2)] == by_dev
    assert rss_end - rss_mid < 64 * 1024, "resident memory grew by %d KB over the second hundred lists" % (rss_end - rss_mid)
    node.close()


def test_node_submit_and_collect_on_two_threads(gpu_ctx):
    """dcs_node_submit on one thread, dcs_node_collect on another, from the very first list on (a device's pipeline is created
    by its first submit): a list counts as submitted once submit has returned, every list comes back in submission order"""
    import threading, zlib
    base = workloads.streams_mixed_16384(n_streams=24, n_frames=40)
    variants = [base[:24], base[2:9], base[8:24] + base[:3]]
    want = [zlib.crc32(gpu_ctx.decode_streams(v)[0].tobytes()) for v in variants]
    refs = [D.make_refs(v) for v in variants]
    node = D.Node([0, 0], depth=4)
    n, submitted, got, errors = 60, threading.Semaphore(0), [], []

    def collector():
        try:
            for _ in range(n):
                submitted.acquire()
                pcm, err, first, _, _, dev = node.collect()
                got.append((zlib.crc32(pcm.tobytes()), bool(err.any()), len(first) - 1))
        except Exception as e:          # (reported by the main thread)
            errors.append(e)

    t = threading.Thread(target=collector)
    t.start()
    for k in range(n):
        v = (k * 5 + k // 7) % 3
        node.submit_refs(refs[v][0], len(variants[v]))
        submitted.release()
    t.join(120)
    assert not t.is_alive() and not errors, errors
    assert got == [(want[(k * 5 + k // 7) % 3], False, len(variants[(k * 5 + k // 7) % 3])) for k in range(n)]
    node.close()


def test_sharded_entry_keeps_its_contexts_between_calls(oracle, corpus):
    """dcs_decode_streams_sharded runs on the persistent contexts of a cached node: the second call with the same device
    list creates nothing (markedly faster than the first, which creates two contexts), same PCM both times"""
    import time
    g, manifest, streams = corpus
    sub = streams[200:260]
    D.node_cache_release()
    t0 = time.perf_counter(); a = D.decode_streams_sharded([0, 0], sub); t1 = time.perf_counter()
    b = D.decode_streams_sharded([0, 0], sub); t2 = time.perf_counter()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and not a[1].any()
    assert stream_hashes(oracle, b[0], b[2]) == g["stream_hashes"][200:260]
    assert (t2 - t1) < (t1 - t0), "second call %.1f ms, first %.1f ms" % ((t2 - t1) * 1e3, (t1 - t0) * 1e3)
    D.node_cache_release()


def test_buffer_cache_is_bounded_and_can_be_trimmed(dcs, corpus):
    """the context's buffer cache: limits derived from the card and the host, never exceeded, changeable; a trim
    gives everything back; decoding goes on afterwards with the same PCM"""
    g, manifest, streams = corpus
    ctx = dcs.Context(0)
    dev, pin, dev_lim, pin_lim = ctx.cache_bytes()
    assert dev == 0 and pin == 0 and 0 < dev_lim <= 32 << 30 and 0 < pin_lim <= 8 << 30
    want = ctx.decode_streams(streams[:30])
    pipe = ctx.pipeline(4, index_on_device=True, pack_on_device=True, plan_on_device=True)
    for _ in range(4):
        pipe.submit(streams[:30])
    for _ in range(4):
        pcm, err, first, _, _ = pipe.collect()
        assert np.array_equal(pcm, want[0])
    pipe.close()
    dev, pin, _, _ = ctx.cache_bytes()
    assert dev > 0 and pin > 0 and dev <= dev_lim and pin <= pin_lim
    released = ctx.trim_cache()
    assert released == (dev, pin) and ctx.cache_bytes()[:2] == (0, 0)
    ctx.set_cache_limits(1 << 20, 1 << 20)          # next to nothing may be kept: every buffer goes back to the runtime at once
    got = ctx.decode_streams(streams[:30])
    assert np.array_equal(got[0], want[0])
    dev, pin, dev_lim, pin_lim = ctx.cache_bytes()
    assert (dev_lim, pin_lim) == (1 << 20, 1 << 20) and dev <= 1 << 20 and pin <= 1 << 20
    ctx.close()


@pytest.mark.parametrize("fpw", [4, 8, 16])
def test_xcd_range_launches_give_the_same_pcm(dcs, corpus, fpw):
    """dcs_ctx_set_concurrent_batches: chunks in chain order, logical workgroups mapped to the XCDs in ranges -- the PCM of the
    default plan (depth order, index order), for every kernel variant, on ragged streams of all six layouts"""
    g, manifest, streams = corpus
    part = streams[300:380]
    ctx = dcs.Context(0)
    ctx.set_frames_per_wave(fpw)
    want = ctx.decode_streams(part, extra_frames=1)
    ctx.set_concurrent_batches(True)
    got = ctx.decode_streams(part, extra_frames=1)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    ctx.close()


@pytest.mark.parametrize("ranges", [True, False], ids=["xcd-ranges", "plain-order-shuffled"])
def test_two_contexts_launching_side_by_side_lose_no_tail(dcs, oracle, ranges):
    """two contexts on one GPU, each relaunching a resident 65 536-frame batch from a thread of its own (decode kernels of different
    launches side by side, as several ranks or pipelines on one card make them): no frame is flagged and both PCMs equal the
    reference's hashes -- launched in XCD ranges (what rounds 4 and 5 needed for this) and, since nobody waits for a tail any more,
    just as well without them and with the chunks in a random order"""
    import threading
    gold = json.load(open(os.path.join(GOLD, "dcs_golden_hashes.json")))["workloads"]["survey3_65536"]["stream_hashes"]
    streams = workloads.streams_survey3_65536()
    b = D.build_stream_batch(streams, indexer=D.index_streams)
    ctxs = [dcs.Context(0), dcs.Context(0)]
    batches = []
    for k, c in enumerate(ctxs):
        c.set_concurrent_batches(ranges)
        if not ranges:
            c.set_test_hooks(chunk_order_seed=99 + k, no_xcd_ranges=True)
        batches.append(c.batch(b["blob"], b["srcs"], b["jobs"]))
    def drive(bt):
        for _ in range(10):
            bt.run_many(100)
            bt.sync()
    th = [threading.Thread(target=drive, args=(bt,)) for bt in batches]
    for t in th: t.start()
    for t in th: t.join()
    for bt in batches:
        pcm, err = bt.download()
        assert not (err & D.FRAME_TAIL_LOST).any() and not err.any()
        assert stream_hashes(oracle, pcm, b["first_job"]) == gold
        bt.close()
    for c in ctxs:
        c.close()
