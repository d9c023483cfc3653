"""The track-program sequencer (csrc/dcs_sequencer.cpp) against the UNMODIFIED reference decoder
driven tick by tick with ROMs (oracle/ref_driver.cpp:ref_seq_run): same bytes to the host at the same ticks,
same fatal-error outcome, and -- on the GPU -- the same PCM for whole event scripts decoded in ONE launch.
Expected values are committed (tests/golden/seq_golden.json) and also compared live where oracle/_ref exists."""
import json
import os
import sys

import numpy as np
import pytest

import dcsexplorer_amd as D
import romkit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_seq_golden as G                     # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "seq_golden.json")))
PAIRS = [(c, s) for c in G.CASES for s in romkit.SCRIPTS]
IDS = ["%s/%s" % (c[0], s) for c, s in PAIRS]


def make_sequencer(case):
    rs = D.RomSet(images=G.build(case).images)
    c = rs.check()
    assert c.status == 1 and c.os == case[2] and c.nominalVersion == case[5]
    return rs, D.Sequencer(rs, G.VOLUME)


@pytest.mark.parametrize("case,script", PAIRS, ids=IDS)
def test_host_bytes_and_fatal_state_equal_reference_golden(case, script):
    n, ev = romkit.SCRIPTS[script]
    rs, seq = make_sequencer(case)
    seq.run_script(n, ev)
    gold = GOLD["%s/%s" % (case[0], script)]
    assert seq.pending_ticks == n
    assert [list(x) for x in seq.host_bytes()] == gold["host_bytes"]
    assert seq.fatal == gold["fatal"]


@pytest.mark.parametrize("case,script", PAIRS, ids=IDS)
def test_host_bytes_equal_reference_live(case, script, reference):
    n, ev = romkit.SCRIPTS[script]
    _, hb, fatal = G.ref_run(reference, G.build(case), G.VOLUME, n, ev)
    rs, seq = make_sequencer(case)
    seq.run_script(n, ev)
    assert [list(x) for x in seq.host_bytes()] == hb and seq.fatal == fatal


def test_sequencer_needs_versions_and_u2():
    rs = D.RomSet(images=G.build(G.CASES[2]).images)
    with pytest.raises(D.DcsError):
        D.Sequencer(rs)                         # versions not detected / set yet
    rs.check()
    D.Sequencer(rs)


@pytest.mark.gpu
@pytest.mark.parametrize("case,script", PAIRS, ids=IDS)
def test_script_pcm_equals_reference(gpu_ctx, case, script):
    """a whole event script -- up to 8 channels, fades, loops, deferred tracks -- planned on the host and decoded in
    one launch: PCM hash == the reference's, and PCM == the live reference where its build travelled along"""
    from oracle.dcs_oracle import Reference, reference_available, fnv1a64
    n, ev = romkit.SCRIPTS[script]
    rs, seq = make_sequencer(case)
    # decode in two launches to exercise the tail carry between plans
    half = n // 2
    seq.run_script(half, [e for e in ev if e[0] < half])
    pcm_a, _ = seq.decode(gpu_ctx)
    rest = [(t - half, k, v) for t, k, v in ev if t >= half]
    seq.run_script(n - half, rest)
    pcm_b, _ = seq.decode(gpu_ctx)
    pcm = np.concatenate([pcm_a, pcm_b])
    gold = GOLD["%s/%s" % (case[0], script)]
    if reference_available():
        want, _, _ = G.ref_run(Reference(), G.build(case), G.VOLUME, n, ev)
        bad = np.nonzero((pcm != want).any(axis=1))[0]
        assert bad.size == 0, "first differing ticks: %s" % bad[:8]
    assert "%016x" % fnv1a64(pcm.tobytes()) == gold["pcm_fnv1a64"]


@pytest.mark.parametrize("case,script", PAIRS, ids=IDS)
def test_going_back_inside_a_plan_replays_to_the_same_state(case, script):
    """dcs_seq_rewind keeps a snapshot every 64 ticks and runs the ticks between the snapshot and the tick asked for again.
    A caller that plans 97 ticks ahead and is taken back by every event (the way DCSDecoderHIP's pump is) ends up with the
    same bytes for the host at the same ticks, none of them twice, and the same fatal state as the tick-by-tick run --
    which equal the reference's (seq_golden.json)."""
    n, ev = romkit.SCRIPTS[script]
    ev = sorted(ev, key=lambda x: x[0])
    rs, seq = make_sequencer(case)
    seq.set_rewindable(True)
    planned, e = 0, 0
    while planned < n:
        while e < len(ev) and ev[e][0] <= planned:
            seq.event(ev[e][1], ev[e][2])
            e += 1
        seq.plan(min(97, n - planned))
        planned = min(n, planned + 97)
        if e < len(ev) and ev[e][0] < planned:
            seq.rewind(ev[e][0])
            planned = ev[e][0]
            assert seq.pending_ticks == planned
    gold = GOLD["%s/%s" % (case[0], script)]
    assert seq.pending_ticks == n
    assert [list(x) for x in seq.host_bytes()] == gold["host_bytes"]
    assert seq.fatal == gold["fatal"]


def test_plan_ahead_stops_two_ticks_into_silence():
    """dcs_seq_plan_ahead: a stream of 30 frames on a stand-alone sequencer; asked for 500 ticks it plans the 30, the tick that
    carries the last overlap tail out and one of silence; asked again it plans one tick at a time; IsStreamPlaying is
    answered for any tick of the plan without going back"""
    from util import make_stream
    L = D.load_library()
    s = make_stream(D.FMT_94_T1_S3, 30, seed=77)
    h = L.dcs_seq_create_standalone(D.OS94)
    try:
        seq = D.Sequencer.__new__(D.Sequencer)
        seq.L, seq.rs, seq.h = L, None, h
        assert L.dcs_seq_load_audio_stream_mem(h, 0, s, len(s), 0x64) == 0
        assert seq.plan_ahead(500) == 32
        assert [seq.stream_playing_at(k, 0) for k in (0, 1, 29, 30, 32)] == [True, True, True, False, False]
        assert seq.plan_ahead(500) == 1 and seq.pending_ticks == 33
        assert L.dcs_seq_load_audio_stream_mem(h, 1, s, len(s), 0x60) == 0
        assert seq.plan_ahead(7) == 7 and seq.plan_ahead(500) == 25
        assert seq.stream_playing_at(33 + 29, 1) and not seq.stream_playing_at(33 + 30, 1)
    finally:
        seq.h = None
        L.dcs_seq_destroy(h)


def test_long_stream_is_planned_while_it_is_being_walked():
    """a stream of more than 192 frames is walked by the sequencer's background thread: dcs_seq_plan_ahead hands out the ticks whose
    records are there (at least one per call, waiting for it if need be), never runs ahead of the walker, and ends two ticks
    behind the stream like with a stream walked at once; a second long stream loaded meanwhile waits for the first walk"""
    from util import make_stream
    L = D.load_library()
    for fmt, os_ in ((D.FMT_94_T1_S3, D.OS95), (D.FMT_93B_T1, D.OS93B), (D.FMT_93A_T1, D.OS93A)):
        s = make_stream(fmt, 1500, seed=78 + fmt, profile=5)
        s2 = make_stream(fmt, 900, seed=178 + fmt, profile=1)
        h = L.dcs_seq_create_standalone(os_)
        seq = D.Sequencer.__new__(D.Sequencer)
        seq.L, seq.rs, seq.h = L, None, h
        try:
            assert L.dcs_seq_set_rewindable(h, 1) == 0
            assert L.dcs_seq_load_audio_stream_mem(h, 0, s, len(s), 0x64) == 0
            total, calls = 0, 0
            while total < 700:
                n = seq.plan_ahead(64 if calls == 0 else 200)       # (at most 899 ticks then, whatever the walker's pace)
                assert n >= 1
                total += n
                calls += 1
            assert seq.stream_playing_at(total, 0)
            # going back inside the plan and forward again, while the walk may still be on
            seq.rewind(total - 300)
            assert seq.pending_ticks == total - 300
            total -= 300
            assert L.dcs_seq_load_audio_stream_mem(h, 1, s2, len(s2), 0x60) == 0      # (waits for the first stream's walk)
            while True:
                n = seq.plan_ahead(4096)
                total += n
                if not seq.stream_playing_at(total, 0) and not seq.stream_playing_at(total, 1):
                    break
            # channel 0 ends behind tick 1500; channel 1, loaded at tick `at`, 900 ticks later; then two ticks of silence
            assert seq.pending_ticks == total
            assert seq.stream_playing_at(1499, 0) and not seq.stream_playing_at(1500, 0)
        finally:
            seq.h = None
            L.dcs_seq_destroy(h)
