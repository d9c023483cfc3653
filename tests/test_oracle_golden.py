"""oracle/dcs_oracle.c against the committed golden vectors (produced by the compiled reference,
tests/golden/make_golden.py) -- this is what pins the oracle on the GPU box, where /root/reference
does not exist."""
import json
import os

import numpy as np
import pytest

from dcsexplorer_amd import workloads
from oracle.dcs_oracle import fnv1a64

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden():
    meta = json.load(open(os.path.join(GOLD, "dcs_golden_hashes.json")))
    arrays = np.load(os.path.join(GOLD, "dcs_golden.npz"))
    return meta, arrays


def case_streams(case, arrays):
    name = case["name"]
    if case["streams"] == 1:
        return [arrays[name + "/stream"].tobytes()]
    return [arrays["%s/stream%d" % (name, c)].tobytes() for c in range(case["streams"])]


def test_oracle_reproduces_every_golden_case(oracle, golden):
    meta, arrays = golden
    assert len(meta["cases"]) >= 36
    for case in meta["cases"]:
        pcm = oracle.decode(case["os"], case["volume"], case_streams(case, arrays), case["levels"], case["frames_out"])
        assert np.array_equal(pcm, arrays[case["name"] + "/pcm"]), case["name"]


def test_config_1_as_the_survey_words_it(oracle, golden):
    """BASELINE.json configs[0] / SURVEY.md 8(d) Config 1: ONE OS93a Type-0 stream, 64 frames, seed 0x93010001, volume 255, level
    0x64 -- the plumbing case, no GPU: the CPU restatement against the PCM the compiled reference produced for it (committed by
    tests/golden/make_golden.py; the stream itself is regenerated from the seed here and must be the committed one)"""
    import dcsexplorer_amd as D
    from util import make_stream
    meta, arrays = golden
    case = [c for c in meta["cases"] if c["name"] == "CONFIG-1"][0]
    assert (case["os"], case["volume"], case["levels"], case["frames_out"]) == (D.OS93A, 255, [0x64], 66)
    s = make_stream(D.FMT_93_T0, 64, seed=0x93010001, profile=6, nbands=12)
    assert s == arrays["CONFIG-1/stream"].tobytes() and ((s[0] << 8) | s[1]) == 64
    idx, info = D.index_stream(D.OS93A, s)
    assert info.format == D.FMT_93_T0 and info.nValidFrames == 64
    pcm = oracle.decode(D.OS93A, 255, [s], [0x64], 66)
    want = arrays["CONFIG-1/pcm"]
    assert np.array_equal(pcm, want)
    assert want[:64].any(axis=1).all() and want[64].any() and not want[65].any()     # 64 frames of sound, the taper frame, silence


def test_survey_appendix_d_sample_values(golden):
    """the sample values printed in SURVEY.md Appendix D (measured by the surveyor on the reference)"""
    meta, arrays = golden
    kats = [c for c in meta["cases"] if c["name"].startswith("KAT-")]
    assert len(kats) == 7
    for c in kats:
        pcm = arrays[c["name"] + "/pcm"]
        assert list(pcm[0, :8]) == c["survey_head"]
        assert list(pcm[0, 16:20]) == c["survey_mid"]
        assert list(pcm[0, 236:240]) == c["survey_tail"]
        assert not pcm[3].any()                     # frame 3 is silence
        assert not pcm[2, 16:].any()                # frame 2 is the taper: only the 16 overlap samples


@pytest.mark.parametrize("wl", ["dcs93_4096", "dcs94_65536", "mixed_16384", "survey3_65536"])
def test_oracle_matches_reference_hashes_of_full_workloads(oracle, golden, wl):
    """full-size seeded workloads: per-stream FNV-1a of the oracle's PCM == the reference's"""
    meta, _ = golden
    want = meta["workloads"][wl]
    streams = workloads.WORKLOADS[wl]()
    assert len(streams) == want["streams"]
    got = []
    for os_, s, vol, lvl in streams:
        nf = (s[0] << 8) | s[1]
        got.append("%016x" % fnv1a64(oracle.decode(os_, vol, [s], [lvl], nf).tobytes()))
    assert got == want["stream_hashes"]


@pytest.fixture(scope="module")
def encoder_golden():
    return (json.load(open(os.path.join(GOLD, "encoder_golden.json"))), np.load(os.path.join(GOLD, "encoder_golden.npz")))


def test_oracle_on_streams_made_by_the_reference_encoder(oracle, encoder_golden):
    """24 recordings made by the reference's own encoder (tests/golden/make_encoder_golden.py): the oracle's PCM equals
    the unmodified reference decoder's -- sample for sample for the first variant, by hash for the others"""
    meta, arrays = encoder_golden
    assert len(meta["cases"]) == 24
    assert sorted(set(c["name"].split("-v")[0] for c in meta["cases"])) == ["ENC-93a-T0", "ENC-93b-T0", "ENC-93b-T1", "ENC-94-T0", "ENC-94-T1s0", "ENC-94-T1s3"]
    for c in meta["cases"]:
        s = arrays[c["name"] + "/stream"].tobytes()
        pcm = oracle.decode(c["os"], c["volume"], [s], c["levels"], c["frames_out"])
        assert "%016x" % oracle.fnv1a64(pcm) == c["pcm_fnv1a64"], c["name"]
        if c["name"] + "/pcm" in arrays:
            assert np.array_equal(pcm, arrays[c["name"] + "/pcm"]), c["name"]


def test_oracle_matches_reference_hashes_of_the_realistic_workload(oracle, encoder_golden):
    meta, _ = encoder_golden
    want = meta["workloads"]["realistic_65536"]
    streams = workloads.WORKLOADS["realistic_65536"]()
    assert len(streams) == want["streams"]
    got = ["%016x" % oracle.fnv1a64(oracle.decode(os_, vol, [s], [lvl], (s[0] << 8) | s[1])) for os_, s, vol, lvl in streams]
    assert got == want["stream_hashes"]


def test_rank_goldens_cover_eight_ranks_and_rank_0_is_the_workload_golden():
    """tests/golden/rank_golden_hashes.json (make_rank_golden.py): what bench.py --gpus N holds every rank's PCM against.
    Rank 0's streams are the workload itself, so its hashes must be the committed workload hashes; ranks differ."""
    import json, os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rg = json.load(open(os.path.join(G, "rank_golden_hashes.json")))
    wl = json.load(open(os.path.join(G, "dcs_golden_hashes.json")))["workloads"]
    wl.update(json.load(open(os.path.join(G, "encoder_golden.json")))["workloads"])
    assert rg["ranks"] == 8
    for name in ("survey3_65536", "dcs94_65536", "dcs93_4096", "mixed_16384", "realistic_65536"):
        per_rank = rg["workloads"][name]["rank_stream_hashes"]
        assert len(per_rank) == 8 and per_rank[0] == wl[name]["stream_hashes"]
        assert all(len(r) == len(per_rank[0]) for r in per_rank)
        if name != "realistic_65536":                       # (24 recordings replicated: ranks repeat recordings at other volumes)
            assert len({h for r in per_rank for h in r}) == 8 * len(per_rank[0])


def test_rank_goldens_equal_the_oracle_on_a_sample_of_every_rank(oracle):
    """the restatement reproduces the reference's hashes for streams of ranks 1..7 too (CPU, a few streams per rank)"""
    import json, os
    from dcsexplorer_amd import sharding
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rg = json.load(open(os.path.join(G, "rank_golden_hashes.json")))
    for name in ("survey3_65536", "dcs93_4096", "mixed_16384"):
        for r in (1, 4, 7):
            streams = sharding.rank_streams(name, r)
            for k in (0, len(streams) // 2 + 1, len(streams) - 1):
                os_, data, vol, lvl = streams[k]
                nf = (data[0] << 8) | data[1]
                assert "%016x" % oracle.fnv1a64(oracle.decode(os_, vol, [data], [lvl], nf)) == rg["workloads"][name]["rank_stream_hashes"][r][k]


def test_full_corpus_golden_is_the_reduced_corpus_grown():
    """corpus_golden_full.json: SURVEY 8(d) Config 5 at its stated size; same recipe and seeds as the reduced corpus, so
    stream k of title t of the reduced corpus is stream k of title t of the full one"""
    import json, os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    full = json.load(open(os.path.join(G, "corpus_golden_full.json")))
    small = json.load(open(os.path.join(G, "corpus_golden.json")))
    assert full["streams"] == 29 * 600 == len(full["stream_hashes"]) and full["frames"] == 17667184
    for t in range(29):
        assert full["stream_hashes"][t * 600:t * 600 + 20] == small["stream_hashes"][t * 20:(t + 1) * 20]
