"""Synthetic DCS sound ROM sets for the ROM-ingestion tests: U2 with a boot JUMP, signature,
catalog (ROM table with sizes / bank selects / checksums, track index pointer, track count), a track index,
track programs that use every opcode, and streams from the library's seeded stream writer spread over
U2..U4.  No real ROM exists in this tree; the layout follows the field descriptions in
DCSDecoder.cpp:26-66, :207-234, :300-345 and DCSDecoder.h:362-478."""
import io
import struct
import zipfile

import dcsexplorer_amd as D
from util import splitmix

HW93, HW95 = 2, 3


def _u16(v): return struct.pack(">H", v & 0xFFFF)
def _u24(v): return struct.pack(">I", v & 0xFFFFFF)[1:]


def checksum(img):
    return ((sum(img[0::2]) & 0xFF) << 8) | (sum(img[1::2]) & 0xFF)


class RomSet:
    def __init__(self, hw, os_, catalog_ofs, seed, n_streams=9, frames=6, version_code=True, signature="Synthetic Pinball (c) 1994 test"):
        self.hw, self.os, self.catalog_ofs = hw, os_, catalog_ofs
        g = splitmix(seed)
        sizes = {2: 0x80000, 3: 0x100000, 4: 0x80000}
        img = {c: bytearray(b"\xFF" * s) for c, s in sizes.items()}
        u2 = img[2]
        u2[0:4] = bytes([0x18, 0x01, 0x0F, 0xFF])                   # JUMP at the reset vector
        sig = signature.encode() + b"\0"
        u2[4:4 + len(sig)] = sig
        for c in (3, 4):                                            # data ROM signature: "S<n> ... mm/dd/yy"
            s = ("S%d Synthetic Sound ROM 01/0%d/94" % (c, c)).encode() + b"\0"
            img[c][0:len(s)] = s
        if version_code:                                            # code patterns CheckROMs looks for (:372-440)
            def ops(ofs, words):
                for k, w in enumerate(words):
                    u2[ofs + 4 * k: ofs + 4 * k + 3] = struct.pack(">I", w)[1:]
            if hw == HW95:
                ops(0x2000 + 0x320 * 4, [0x40104E, 0x0F16F8, 0x93300E, 0x18123F, 0x40104E, 0x0F1608, 0x0F16F8, 0x93300E, 0x18456F])
            elif os_ in (D.OS93A, D.OS93B):
                ops(0x1000 + 0x140 * 4, [0x380026, 0x3C1005, 0x0C00C0])
                if os_ == D.OS93A:
                    ops(0x2000 + 0x210 * 4, [0x47FFF2, 0x47C946])

        def linear(chip, off):
            return ((chip - 2) << (21 if hw == HW95 else 20)) | off

        # streams
        fmts = [f for f in range(6) if D.format_os(f) == os_ or D.format_os(f, prefer_95=True) == os_ or D.format_os(f, prefer_93a=True) == os_]
        self.streams = {}
        cursor = {2: 0x10000, 3: 0x100, 4: 0x100}
        for i in range(n_streams):
            chip = (2, 3, 4)[i % 3]
            data = D.synth_stream(fmts[i % len(fmts)], frames + i, seed=seed * 100 + i, profile=i % 4)
            off = cursor[chip]
            img[chip][off:off + len(data)] = data
            cursor[chip] = (off + len(data) + 0x40) & ~0xF
            self.streams[linear(chip, off)] = data
        addrs = list(self.streams)

        # track programs
        tracks = []
        def step(delay, opcode, operands=b""):
            return _u16(delay) + bytes([opcode]) + operands
        def play(ch, addr, rep=1): return bytes([ch]) + _u24(addr) + bytes([rep])
        op04 = bytes([0x55, 0x00, 0x10]) if os_ == D.OS93A else bytes([0x55])
        tracks.append(bytes([1, 0]) + step(0, 0x02, b"\0") + step(0, 0x00))                              # stop all
        tracks.append(bytes([1, 0]) + step(0, 0x07, bytes([0, 0x70])) + step(0, 0x01, play(0, addrs[0], 0)) + step(0xFFFF, 0x0D))
        tracks.append(bytes([1, 1]) + step(2, 0x0A, bytes([1, 0x50, 0, 8])) + step(0, 0x08, bytes([1, 5])) + step(1, 0x01, play(1, addrs[1]))
                      + step(0, 0x0E, bytes([3])) + step(4, 0x09, bytes([1, 2])) + step(0, 0x01, play(1, addrs[2], 2))
                      + step(0, 0x0E, bytes([2])) + step(1, 0x0D) + step(0, 0x0F) + step(0, 0x0F) + step(0, 0x03, _u16(7)) + step(0, 0x00))
        tracks.append(None)                                                                              # unpopulated
        tracks.append(bytes([2, 2]) + _u16(0x0102))                                                      # deferred
        tracks.append(bytes([3, 3]) + bytes([7, 1]))                                                     # deferred indirect
        tracks.append(bytes([1, 2]) + step(0, 0x04, op04) + step(0, 0x05, bytes([2])) + step(0, 0x06, bytes([7, 3]))
                      + step(0, 0x0B, bytes([2, 4, 0, 2])) + step(0, 0x0C, bytes([2, 9, 0, 3])) + step(3, 0x01, play(2, addrs[3]))
                      + step(0, 0x01, play(2, addrs[1])) + step(0, 0x10, bytes([1, 2])) + step(0, 0x11, bytes([1, 2, 3, 4]))
                      + step(0, 0x12, bytes([5, 6, 7, 8])) + step(0, 0x0F) + step(0, 0x00))
        tracks.append(bytes([1, 4]) + step(0, 0x0E, bytes([0])) + step(5, 0x01, play(4, addrs[4], 0)) + step(0, 0x0F) + step(0, 0x00))
        tracks.append(bytes([1, 5]) + step(0, 0x07, bytes([3, 0x40])) + step(0, 0x01, play(3, addrs[5])) + step(0, 0x33, b"\1\2"))   # invalid opcode
        tracks.append(bytes([1, 9]) + step(0, 0x00))                                                     # bad channel
        for i in range(6, n_streams):
            lvl = 0x30 + (next(g) % 0x50)
            tracks.append(bytes([1, i % 6]) + step(0, 0x07, bytes([i % 6, lvl])) + step(next(g) % 4, 0x01, play(i % 6, addrs[i])) + step(0, 0x00))
        self.n_tracks = len(tracks)

        # lay the programs out in U2 (and one in U3), build the index
        index = bytearray()
        p2, p3 = 0x8000, 0x90000
        for k, t in enumerate(tracks):
            if t is None:
                index += b"\xFF\xFF\xFF"
                continue
            if k == 2:
                img[3][p3:p3 + len(t)] = t; index += _u24(linear(3, p3)); p3 += len(t) + 3
            else:
                u2[p2:p2 + len(t)] = t; index += _u24(linear(2, p2)); p2 += len(t) + 1
        index_ofs = 0x7000
        u2[index_ofs:index_ofs + len(index)] = index

        # catalog
        c = catalog_ofs
        u2[c + 0x40:c + 0x43] = _u24(index_ofs)
        u2[c + 0x43:c + 0x46] = _u24(index_ofs + 0x800)
        u2[c + 0x46:c + 0x48] = _u16(len(tracks))
        def sel(chip): return ((chip - 2) << (1 if hw == HW95 else 0)) << 8
        table = bytearray()
        for chip in (2, 3, 4):
            table += _u16(sizes[chip] // 4096) + _u16(sel(chip)) + _u16(0 if chip == 2 else checksum(img[chip]))
        table += _u16(0)
        u2[c:c + len(table)] = table
        # U2's own checksum must come out as zero (:207-234, :300-345): two fix-up bytes
        u2[0x7FFFE] = u2[0x7FFFF] = 0
        ck = checksum(u2)
        u2[0x7FFFE] = (-(ck >> 8)) & 0xFF
        u2[0x7FFFF] = (-(ck & 0xFF)) & 0xFF
        assert checksum(u2) == 0
        self.images = {c: bytes(b) for c, b in img.items()}

    def zip_bytes(self, names=None, extra=None, compress=True):
        names = names or {2: "synth_u2.rom", 3: "synth_s3.rom", 4: "snd_u4_v1.bin"}
        buf = io.BytesIO()
        with zipfile.ZipFile(buf, "w", zipfile.ZIP_DEFLATED if compress else zipfile.ZIP_STORED) as z:
            for name, data in (extra or {}).items():
                z.writestr(name, data)
            for chip in sorted(self.images, reverse=True):          # not in chip order
                z.writestr(names[chip], self.images[chip])
        return buf.getvalue()


def load(rs):
    """-> a DcsRomSet handle (ctypes void pointer) holding rs's images"""
    import ctypes
    L = D.load_library()
    h = ctypes.c_void_p(L.dcs_romset_create())
    for chip, data in rs.images.items():
        assert L.dcs_romset_add_rom(h, chip, data, len(data)) == 0
    return h


def damage(rs, seed):
    """seed >= 0: random byte changes inside the track-program area of U2 (pointers and catalog stay intact, so
    the reference does not read outside its images); seed < 0: one changed byte in U3 (checksum failure)"""
    import copy
    out = copy.copy(rs)
    imgs = {c: bytearray(b) for c, b in rs.images.items()}
    if seed < 0:
        imgs[3][0x5000] ^= 0x5A
    else:
        g = splitmix(seed)
        for _ in range(12):
            pos = 0x8000 + next(g) % 0x120
            imgs[2][pos] = next(g) & 0xFF
        # keep U2's checksum at zero so that the catalog is still found
        imgs[2][0x7FFFE] = imgs[2][0x7FFFF] = 0
        ck = checksum(imgs[2])
        imgs[2][0x7FFFE] = (-(ck >> 8)) & 0xFF
        imgs[2][0x7FFFF] = (-(ck & 0xFF)) & 0xFF
    out.images = {c: bytes(b) for c, b in imgs.items()}
    return out
