"""shared helpers for the test-suite"""
import numpy as np

import dcsexplorer_amd as D
from dcsexplorer_amd.api import format_os

ALL_FORMATS = [D.FMT_93_T0, D.FMT_93B_T1, D.FMT_93A_T1, D.FMT_94_T0, D.FMT_94_T1_S0, D.FMT_94_T1_S3]
FORMAT_NAMES = {D.FMT_93_T0: "93-T0", D.FMT_93B_T1: "93b-T1", D.FMT_93A_T1: "93a-T1", D.FMT_94_T0: "94-T0",
                D.FMT_94_T1_S0: "94-T1s0", D.FMT_94_T1_S3: "94-T1s3"}


def make_stream(fmt, nframes, seed, profile=0, stride_from=16, nbands=None):
    if nbands is None:
        nbands = 18 if fmt == D.FMT_93A_T1 else 16
        if fmt == D.FMT_93_T0 and stride_from < 16:
            nbands = 12
    return D.synth_stream(fmt, nframes, seed, nbands=nbands, stride_from=stride_from, profile=profile)


def os_for(fmt, variant=0):
    return format_os(fmt, prefer_95=bool(variant & 1), prefer_93a=bool(variant & 1))


def oracle_pcm(checker, os_, stream, volume, level, extra=0):
    nframes = (stream[0] << 8) | stream[1]
    return checker.decode(os_, volume, [stream], [level], nframes + extra)


def splitmix(seed):
    x = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def corrupt(stream, seed, nflips=3, protect=18):
    """flip a few payload bits (keeps the frame count and header intact)"""
    b = bytearray(stream)
    g = splitmix(seed)
    for _ in range(nflips):
        pos = protect + next(g) % max(1, (len(b) - protect))
        b[pos] ^= 1 << (next(g) & 7)
    return bytes(b)
