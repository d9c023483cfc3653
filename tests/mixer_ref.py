"""Test helper: builds a multi-source job list for `nch` streams loaded at tick 0 on channels 0..n-1 of
a fresh decoder, using only the library's public parameter functions (dcs_frame_scale etc.).  It
restates the channel bookkeeping of MainLoop / DecodeStream (DCSDecoderNative.cpp:89-306, :1546-1589)
for streams without track programs."""
import numpy as np

import dcsexplorer_amd as D


def build_mix_batch(os_, volume, streams, levels, frames_out):
    nch = len(streams)
    blob = bytearray()
    idxs, infos, offs = [], [], []
    for s in streams:
        idx, info = D.index_stream(os_, s)
        while len(blob) & 3:
            blob.append(0)
        offs.append(len(blob))
        blob += s + bytes(64)
        idxs.append(idx)
        infos.append(info)
    vol_mult = D.volume_multiplier(volume)
    mix = [0x7FFF] * nch                               # Channel::mixingMultiplier initialiser
    steady = [D.mixing_multiplier(os_, lv << 6, 0xFF) for lv in levels]
    pos = [0] * nch
    active = [True] * nch
    srcs, jobs = [], []
    for f in range(frames_out):
        act = np.array([1 if a else 0 for a in active], dtype=np.uint8)
        vs, scaled = D.frame_scale(vol_mult, np.array(mix, dtype=np.uint16), act)
        first = len(srcs)
        n = 0
        for c in range(nch):
            if not active[c]:
                continue
            fi = idxs[c][pos[c]]
            sd = np.zeros(1, dtype=D.SRC_DTYPE)
            sd["streamOff"] = offs[c]; sd["idx"] = fi
            sd["mixMul"] = scaled[c]; sd["format"] = infos[c].format; sd["hdrLen"] = infos[c].hdrLen
            srcs.append(sd)
            n += 1
            pos[c] += 1
            if pos[c] >= infos[c].nValidFrames:
                active[c] = False
        jb = np.zeros(1, dtype=D.JOB_DTYPE)
        jb["firstSrc"] = first; jb["nSrc"] = n; jb["volShift"] = vs
        jb["xform"] = D.XFORM_93 if os_ in (D.OS93A, D.OS93B) else D.XFORM_94
        jb["prev"] = D.PREV_NONE if f == 0 else f - 1
        jobs.append(jb)
        mix = list(steady)
    return dict(blob=bytes(blob), srcs=np.concatenate(srcs), jobs=np.concatenate(jobs))
