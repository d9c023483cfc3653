"""ctypes bindings for the CPU checkers -- TEST INFRASTRUCTURE ONLY.

Two libraries with identical entry-point signatures:

  * ``oracle/libdcs_oracle.so``      -- our plain-C restatement (``Oracle()``), prefix ``orc_``
  * ``oracle/_ref/libdcsref.so``     -- the unmodified reference decoder compiled by
                                        ``oracle/Makefile`` (``Reference()``), prefix ``ref_``

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg import this module;
the product package (dcsexplorer_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

OS93A, OS93B, OS94, OS95 = 0, 1, 2, 3


class Probe(ctypes.Structure):
    _fields_ = [("active", ctypes.c_int32), ("bitOff", ctypes.c_int32),
                ("mixMul", ctypes.c_uint16), ("volMult", ctypes.c_uint16),
                ("bandType", ctypes.c_uint16 * 16)]


def _u8buf(b):
    b = bytes(b)
    arr = (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(b if b else b"\0")
    return arr, len(b)


class _Checker:
    def __init__(self, path, prefix):
        self.path = path
        self.lib = ctypes.CDLL(path)
        self.prefix = prefix
        L = self.lib
        f = getattr(L, prefix + "decode")
        f.restype = ctypes.c_int
        f = getattr(L, prefix + "volume_multiplier")
        f.restype = ctypes.c_uint16
        f.argtypes = [ctypes.c_int]
        f = getattr(L, prefix + "mixing_multiplier")
        f.restype = ctypes.c_uint16
        f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def decode(self, os_, volume, streams, levels, nframes_out, probes=False):
        """streams: list of bytes (channels 0..n-1); returns int16 [nframes_out, 240] (+ probes)"""
        n = len(streams)
        keep = [_u8buf(s) for s in streams]
        ptrs = (ctypes.POINTER(ctypes.c_uint8) * n)(
            *[ctypes.cast(a, ctypes.POINTER(ctypes.c_uint8)) for a, _ in keep])
        lens = (ctypes.c_size_t * n)(*[l for _, l in keep])
        lv = (ctypes.c_int * n)(*levels)
        pcm = np.zeros((nframes_out, 240), dtype=np.int16)
        pr = (Probe * nframes_out)() if probes else None
        r = self._fn("decode")(ctypes.c_int(os_), ctypes.c_int(volume), ctypes.c_int(n), ptrs, lens, lv,
                               ctypes.c_int(nframes_out), pcm.ctypes.data_as(ctypes.c_void_p),
                               pr if probes else None)
        if r != 0:
            raise RuntimeError("%sdecode failed: %d" % (self.prefix, r))
        return (pcm, pr) if probes else pcm

    def decode_sequence(self, os_, volume, streams, levels, extra_frames=2):
        """streams played one after the other through ONE decoder (the --extract-streams loop);
        returns int16 [sum(nFrames + extra), 240]"""
        n = len(streams)
        keep = [_u8buf(s) for s in streams]
        ptrs = (ctypes.POINTER(ctypes.c_uint8) * n)(
            *[ctypes.cast(a, ctypes.POINTER(ctypes.c_uint8)) for a, _ in keep])
        lens = (ctypes.c_size_t * n)(*[l for _, l in keep])
        lv = (ctypes.c_int * n)(*levels)
        total = sum(((bytes(s)[0] << 8) | bytes(s)[1]) + extra_frames for s in streams)
        pcm = np.zeros((total, 240), dtype=np.int16)
        f = self._fn("decode_sequence")
        f.restype = ctypes.c_int
        r = f(ctypes.c_int(os_), ctypes.c_int(volume), ctypes.c_int(n), ptrs, lens, lv, ctypes.c_int(extra_frames),
              pcm.ctypes.data_as(ctypes.c_void_p))
        if r != 0:
            raise RuntimeError("%sdecode_sequence failed: %d" % (self.prefix, r))
        return pcm

    def stream_info(self, os_, stream):
        a, n = _u8buf(stream)
        nf, nb, ft, fs = (ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int())
        hdr = (ctypes.c_uint8 * 16)()
        self._fn("stream_info")(ctypes.c_int(os_), a, ctypes.c_size_t(n), ctypes.byref(nf), ctypes.byref(nb),
                                ctypes.byref(ft), ctypes.byref(fs), hdr)
        return dict(nFrames=nf.value, nBytes=nb.value, formatType=ft.value, formatSubType=fs.value,
                    header=bytes(hdr))

    def transform(self, os_, fb512, vol_shift, overlap16):
        fb = np.ascontiguousarray(fb512, dtype=np.uint16).copy()
        ov = np.ascontiguousarray(overlap16, dtype=np.uint16).copy()
        assert fb.size == 512 and ov.size == 16
        pcm = np.zeros(240, dtype=np.int16)
        self._fn("transform")(ctypes.c_int(os_), fb.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(vol_shift),
                              ov.ctypes.data_as(ctypes.c_void_p), pcm.ctypes.data_as(ctypes.c_void_p))
        return pcm, ov, fb

    def decompress(self, os_, stream, mix_mul, nframes):
        a, n = _u8buf(stream)
        out = np.zeros((nframes, 512), dtype=np.uint16)
        bit_offs = np.zeros(nframes, dtype=np.int32)
        band_types = np.zeros((nframes, 16), dtype=np.uint16)
        stops = np.zeros(nframes, dtype=np.int32)
        self._fn("decompress")(ctypes.c_int(os_), a, ctypes.c_size_t(n), ctypes.c_uint16(mix_mul),
                               ctypes.c_int(nframes), out.ctypes.data_as(ctypes.c_void_p),
                               bit_offs.ctypes.data_as(ctypes.c_void_p),
                               band_types.ctypes.data_as(ctypes.c_void_p),
                               stops.ctypes.data_as(ctypes.c_void_p))
        return out, bit_offs, band_types, stops

    def decode_many(self, streams, repeat, nthreads):
        """streams: list of (os, bytes, volume, level); decodes each from a fresh decoder `repeat` times on
        `nthreads` host threads inside the C library; returns frames decoded"""
        n = len(streams)
        keep = [_u8buf(s[1]) for s in streams]
        ptrs = (ctypes.POINTER(ctypes.c_uint8) * n)(
            *[ctypes.cast(a, ctypes.POINTER(ctypes.c_uint8)) for a, _ in keep])
        lens = (ctypes.c_size_t * n)(*[l for _, l in keep])
        os_ = (ctypes.c_int * n)(*[s[0] for s in streams])
        vol = (ctypes.c_int * n)(*[s[2] for s in streams])
        lvl = (ctypes.c_int * n)(*[s[3] for s in streams])
        f = self._fn("decode_many")
        f.restype = ctypes.c_longlong
        return int(f(os_, vol, lvl, ptrs, lens, ctypes.c_int(n), ctypes.c_int(repeat), ctypes.c_int(nthreads)))

    def volume_multiplier(self, vol):
        return self._fn("volume_multiplier")(vol)

    def mixing_multiplier(self, os_, level_sum, channel_volume=0xFF):
        return self._fn("mixing_multiplier")(os_, level_sum, channel_volume)


class Oracle(_Checker):
    def __init__(self, build=True):
        path = os.path.join(HERE, "libdcs_oracle.so")
        if build and (not os.path.exists(path) or
                      os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "dcs_oracle.c"))):
            subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
        super().__init__(path, "orc_")
        self.lib.orc_fnv1a64.restype = ctypes.c_uint64
        self.lib.orc_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_size_t]

    def frame_params(self, os_, volume, level, nframes):
        mm = np.zeros(nframes, dtype=np.uint16)
        vs = np.zeros(nframes, dtype=np.uint8)
        self.lib.orc_frame_params(ctypes.c_int(os_), ctypes.c_int(volume), ctypes.c_int(level),
                                  ctypes.c_int(nframes), mm.ctypes.data_as(ctypes.c_void_p),
                                  vs.ctypes.data_as(ctypes.c_void_p))
        return mm, vs

    def fnv1a64(self, arr):
        a = np.ascontiguousarray(arr)
        return int(self.lib.orc_fnv1a64(a.ctypes.data_as(ctypes.c_void_p), a.nbytes))


def reference_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libdcsref.so"))


class Reference(_Checker):
    """the compiled, unmodified reference (present only where oracle/_ref was built or shipped)"""
    def __init__(self):
        super().__init__(os.path.join(HERE, "_ref", "libdcsref.so"), "ref_")


def fnv1a64(data):
    h = 0xcbf29ce484222325
    for x in bytes(data):
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h
