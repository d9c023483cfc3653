#!/usr/bin/env python3
"""Seeded fuzz of the decode path on a GPU box: lists of random streams (every layout, 1..18 bands, strided from a random
band, six symbol profiles, some with flipped payload bits or cut short) decoded with 4, 8 and 16 frames per wavefront
and held against the oracle, the device's index pass held against the host's; every fourth seed also a multi-channel mix of 2..6 streams on one decoder, every eighth the list through dcs_pipeline.  argv[1]: seconds to run (default 120), argv[2]: first seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dcsexplorer_amd as D
from oracle.dcs_oracle import Oracle
from util import ALL_FORMATS, FORMAT_NAMES, make_stream, os_for, corrupt, splitmix
from mixer_ref import build_mix_batch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = Oracle()
ctx = D.Context(0)
t0 = t_said = time.time(); lists = frames = mixes = piped = large = 0; seed = seed0
pipes = [ctx.pipeline(3, index_on_device=m >= 1, pack_on_device=m >= 2, plan_on_device=m == 3) for m in range(4)]
by_fmt = {f: 0 for f in ALL_FORMATS}
while time.time() - t0 < budget:
    g = splitmix(0xF022 + seed)
    streams, want = [], []
    for k in range(24):
        fmt = ALL_FORMATS[next(g) % 6]
        nfr = 3 + next(g) % 70
        nb_max = 18 if fmt == D.FMT_93A_T1 else 16
        nbands = nb_max if next(g) % 3 else 1 + next(g) % nb_max
        stride_from = 16 if next(g) % 2 else next(g) % 16
        if fmt == D.FMT_93_T0 and stride_from < 16:
            nbands = min(nbands, 12)                  # (Type 0 strided bands span 32 slots)
        s = make_stream(fmt, nfr, seed=(seed << 8) + k, profile=next(g) % 6, stride_from=stride_from, nbands=nbands)
        r = next(g) % 8
        if r == 0 and len(s) > 24:
            s = corrupt(s, next(g) & 0xFFFF, nflips=1 + next(g) % 4) + bytes(256)
        elif r == 1 and len(s) > 40:
            s = s[:18 + (len(s) - 18) * (1 + next(g) % 3) // 4]       # cut short in the middle of a frame
        os_ = os_for(fmt, next(g) & 1)
        vol, lvl = 128 + next(g) % 128, 0x20 + next(g) % 0x60
        streams.append((os_, s, vol, lvl))
        by_fmt[fmt] += 1
    try:
        ref = [orc.decode(os_, vol, [s], [lvl], ((s[0] << 8) | s[1]) + 1) for os_, s, vol, lvl in streams]
    except Exception as e:
        print("seed %d: oracle refused a stream (%s); skipped" % (seed, e)); seed += 1; continue
    want = np.concatenate(ref)
    # the index pass on the device (one wavefront per stream) against the host walk: same records, same StreamInfo
    host = D.index_streams(streams)
    dev = ctx.index_streams_gpu(streams)
    for k, ((hr, hi), (dr, di)) in enumerate(zip(host, dev)):
        if hr.tobytes() != dr.tobytes() or bytes(hi) != bytes(di):
            print("MISMATCH (index pass) seed %d stream %d" % (seed, k)); sys.exit(1)
    for fpw in (4, 8, 16):
        ctx.set_frames_per_wave(fpw)
        try:
            pcm, err, _ = ctx.decode_streams(streams, extra_frames=1)
        except D.DcsError as e:
            print("seed %d fpw %d: library error %s" % (seed, fpw, e)); sys.exit(1)
        if pcm.shape != want.shape or not np.array_equal(pcm, want):
            bad = np.argwhere(pcm != want) if pcm.shape == want.shape else None
            print("MISMATCH seed %d fpw %d: %s" % (seed, fpw, "shape" if bad is None else "%d samples in %d frames, first frame %d" % (len(bad), len(set(bad[:, 0])), bad[0][0])))
            sys.exit(1)
    lists += 1; frames += want.shape[0]
    # every eighth seed the list also goes through dcs_pipeline in its four modes (index pass on the host pool / on the
    # device / index pass and packer on the device / planner too), in flight twice: once with the taper frame, once without
    # (then a stream's last frame and the next one's first sit side by side in a chunk)
    if seed % 8 == 0:
        ctx.set_frames_per_wave(0)
        ends = np.cumsum([r.shape[0] for r in ref]) - 1
        bare = np.delete(want, ends, axis=0)
        for mode, pipe in enumerate(pipes):
            pipe.submit(streams, extra_frames=1); pipe.submit(streams, extra_frames=0)
            for w in (want, bare):
                pcm, err, first, _, _ = pipe.collect()
                if pcm.shape != w.shape or not np.array_equal(pcm, w):
                    print("MISMATCH (pipeline mode %d) seed %d" % (mode, seed)); sys.exit(1)
        piped += 1
    # every fourth seed also a multi-channel mix: 2..6 streams of one OS version on the channels of one decoder (the
    # further sources of a frame take the kernel's general path, whose deal of the bands is made in the kernel)
    if seed % 4 == 0:
        os_ = [D.OS93A, D.OS93B, D.OS94, D.OS95][next(g) % 4]
        fmts = {D.OS93A: [D.FMT_93_T0, D.FMT_93A_T1], D.OS93B: [D.FMT_93_T0, D.FMT_93B_T1]}.get(os_, [D.FMT_94_T0, D.FMT_94_T1_S0, D.FMT_94_T1_S3])
        nch = 2 + next(g) % 5
        chans, levels = [], []
        for c in range(nch):
            fmt = fmts[next(g) % len(fmts)]
            nb_max = 18 if fmt == D.FMT_93A_T1 else 16
            nbands = nb_max if next(g) % 3 else 1 + next(g) % nb_max
            stride_from = 16 if next(g) % 2 else next(g) % 16
            if fmt == D.FMT_93_T0 and stride_from < 16:
                nbands = min(nbands, 12)
            chans.append(make_stream(fmt, 4 + next(g) % 40, seed=(seed << 8) + 100 + c, profile=next(g) % 6, stride_from=stride_from, nbands=nbands))
            levels.append(0x20 + next(g) % 0x60)
        vol = 128 + next(g) % 128
        n_out = max((c[0] << 8) | c[1] for c in chans) + 2
        b = build_mix_batch(os_, vol, chans, levels, n_out)
        want = orc.decode(os_, vol, chans, levels, n_out)
        for fpw in (4, 8, 16):
            ctx.set_frames_per_wave(fpw)
            pcm, err = ctx.decode_batch(b["blob"], b["srcs"], b["jobs"])
            if not np.array_equal(pcm, want):
                bad = np.argwhere(pcm != want)
                print("MISMATCH (mix of %d) seed %d fpw %d: %d samples in %d frames, first frame %d" % (nch, seed, fpw, len(bad), len(set(bad[:, 0])), bad[0][0]))
                sys.exit(1)
        mixes += 1; frames += n_out
    # every 64th seed a LARGE list (40 streams of 820..900 frames: over 32 768 frames), which dcs_decode_streams cuts into
    # parts that go through the context's own pipeline -- with the parts' index walk, planner and packer on the device, and
    # behind the host pool's index pass -- against the oracle
    if seed % 64 == 0:
        ctx.set_frames_per_wave(0)
        big, bwant = [], []
        for k in range(40):
            fmt = ALL_FORMATS[next(g) % 6]
            s = make_stream(fmt, 820 + next(g) % 81, seed=(seed << 8) + 200 + k, profile=next(g) % 4)
            if next(g) % 16 == 0:
                s = corrupt(s, next(g) & 0xFFFF, nflips=2) + bytes(256)
            elif next(g) % 16 == 1:
                s = s[:18 + (len(s) - 18) // 2]
            os_ = os_for(fmt, next(g) & 1)
            big.append((os_, s, 128 + next(g) % 128, 0x20 + next(g) % 0x60))
        try:
            bwant = np.concatenate([orc.decode(o, v, [s], [l], ((s[0] << 8) | s[1]) + 2) for o, s, v, l in big])
        except Exception as e:
            bwant = None
        if bwant is not None:
            for on_device in (2, 1, 0):
                ctx.set_large_list_path(on_device)
                pcm, err, _ = ctx.decode_streams(big, extra_frames=2)
                if pcm.shape != bwant.shape or not np.array_equal(pcm, bwant):
                    print("MISMATCH (large list, parts %s) seed %d" % (("behind the host index", "on the device", "walk shared by host and device")[on_device], seed)); sys.exit(1)
            ctx.set_large_list_path(2)
            large += 1
    seed += 1
    if time.time() - t_said > 30:               # (a line now and then: a silent command is taken to be hung)
        t_said = time.time()
        print("  ... %d lists, %d mixes, %.0f s" % (lists, mixes, t_said - t0), flush=True)
for pipe in pipes:
    pipe.close()
print("fuzz: %d lists (%d of them also through the pipeline's four modes), %d large lists through dcs_decode_streams' three paths and %d multi-channel mixes (%d frames x 3 kernel variants) in %.0f s, seeds %d..%d, all bit-exact; streams by layout: %s" %
      (lists, piped, large, mixes, frames, time.time() - t0, seed0, seed - 1, {FORMAT_NAMES[f]: n for f, n in by_fmt.items()}))
