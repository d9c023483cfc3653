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
    def __init__(self, hw, os_, catalog_ofs, seed, n_streams=9, frames=6, version_code=True, signature="Synthetic Pinball (c) 1994 test", nominal=0x0104):
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
                v = 0x400000 | (nominal << 4) | 0xE                     # SR0 = <major:minor>
                ops(0x2000 + 0x320 * 4, [v, 0x0F16F8, 0x93300E, 0x18123F, v, 0x0F1608, 0x0F16F8, 0x93300E, 0x18456F])
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


class SeqRomSet(RomSet):
    """a ROM set whose track programs exercise the sequencer: looping and repeated streams, fades, nested and
    endless loops, queued / deferred / deferred-indirect tracks, variables, data-port bytes and host event
    timers, a program that plays on another channel, a stop of the own channel, an invalid opcode and an
    invalid track type"""

    def __init__(self, hw, os_, catalog_ofs, seed, version_code=True, nominal=0x0104):
        RomSet.__init__(self, hw, os_, catalog_ofs, seed, n_streams=12, frames=5, version_code=version_code, nominal=nominal)
        imgs = {c: bytearray(b) for c, b in self.images.items()}
        u2 = imgs[2]
        A = list(self.streams)

        def step(delay, opcode, operands=b""):
            return _u16(delay) + bytes([opcode]) + operands
        def play(ch, addr, rep=1): return bytes([ch]) + _u24(addr) + bytes([rep])
        setvar = (lambda v, x: b"") if os_ in (D.OS93A, D.OS93B) else (lambda v, x: bytes([v, x]))
        def port(b, cnt=0): return bytes([b, cnt >> 8, cnt & 0xFF]) if os_ == D.OS93A else bytes([b])
        end = step(0, 0x00)
        T = []
        T.append(bytes([1, 0]) + b"".join(step(0, 0x02, bytes([c])) for c in range(1, 6)) + step(0, 0x02, b"\0"))     # 0: stops all, itself last
        T.append(bytes([1, 0]) + step(0, 0x07, bytes([0, 0x70])) + step(0, 0x01, play(0, A[0], 0)) + step(0xFFFF, 0x0D))   # 1
        T.append(bytes([1, 1]) + step(0, 0x0A, bytes([1, 0x60, 0, 8])) + step(0, 0x01, play(1, A[1], 2)) + step(5, 0x08, bytes([1, 6]))
                 + step(0, 0x0E, bytes([3])) + step(4, 0x09, bytes([1, 2])) + step(0, 0x01, play(1, A[2])) + step(0, 0x0F)
                 + step(2, 0x03, _u16(7)) + end)                                                                       # 2
        T.append(bytes([2, 2]) + _u16(8))                                                                              # 3: deferred -> track 8
        T.append(bytes([3, 3]) + bytes([7, 1]))                                                                        # 4: deferred indirect: table 1 [var 7]
        T.append(bytes([1, 2]) + step(0, 0x06, setvar(7, 2)) + step(1, 0x05, bytes([2])) + step(0, 0x05, bytes([3]))
                 + step(0, 0x05, bytes([4])) + step(0, 0x07, bytes([2, 0x58])) + step(0, 0x01, play(2, A[3])) + step(9, 0x0D) + end)   # 5
        T.append(bytes([1, 4]) + step(0, 0x04, port(0x69, 3)) + step(0, 0x07, bytes([4, 0x64])) + step(0, 0x01, play(4, A[4], 0))
                 + step(6, 0x0C, bytes([4, 0x20, 0, 5])) + step(12, 0x04, port(0x6A)) + step(0xFFFF, 0x00))             # 6
        T.append(bytes([1, 5]) + step(0, 0x07, bytes([5, 0x50])) + step(0, 0x01, play(5, A[5], 3)) + step(2, 0x0B, bytes([5, 0x10, 0, 6]))
                 + step(0, 0x10, bytes([5, 9])) + step(0, 0x11, bytes([5, 3, 0, 4])) + step(0, 0x12, bytes([5, 1, 0, 0])) + step(14, 0x0D) + end)  # 7
        T.append(bytes([1, 2]) + step(0, 0x07, bytes([2, 0x68])) + step(0, 0x01, play(2, A[6])) + step(0xFFFF, 0x0D))  # 8
        for k in range(3):                                                                                            # 9, 10, 11
            T.append(bytes([1, 3]) + step(0, 0x07, bytes([3, 0x48 + 8 * k])) + step(0, 0x01, play(3, A[7 + k])) + step(0xFFFF, 0x0D))
        T.append(bytes([1, 1]) + step(0, 0x07, bytes([0, 0x30])) + step(0, 0x01, play(0, A[10])) + step(3, 0x07, bytes([1, 0x66]))
                 + step(0, 0x01, play(1, A[11])) + step(0xFFFF, 0x0D))                                                 # 12: plays on channel 0 from channel 1
        T.append(bytes([1, 6]) + step(0, 0x07, bytes([6, 0x64])) + step(0, 0x33, b"\1\2"))                              # 13: invalid opcode, no delay: fatal
        T.append(bytes([4, 0]) + _u16(0))                                                                              # 14: invalid track type
        T.append(bytes([1, 6]) + step(0, 0x0E, bytes([2])) + step(1, 0x07, bytes([6, 0x40])) + step(0, 0x0E, bytes([0]))
                 + step(2, 0x01, play(6, A[2])) + step(0, 0x0F) + step(0, 0x0F) + end)                                  # 15: endless inner loop
        T.append(bytes([1, 7]) + step(0, 0x07, bytes([7, 0x5C])) + step(0, 0x01, play(7, A[3], 2)) + step(0xFFFF, 0x0D))  # 16
        T.append(bytes([1, 0]) + step(2, 0x02, b"\0") + step(0, 0x01, play(0, A[4])) + end)                             # 17: stops itself
        T.append(bytes([1, 6]) + step(0, 0x07, bytes([6, 0x64])) + step(0, 0x01, play(6, A[5], 0)) + step(1, 0x33, b"\1\2"))   # 18: invalid opcode behind a delay: a reset per tick, never fatal
        self.n_tracks = len(T)
        index = bytearray()
        p2 = 0x8000
        u2[0x8000:0x9000] = b"\xFF" * 0x1000
        for t in T:
            u2[p2:p2 + len(t)] = t
            index += _u24(p2 if hw != HW95 else p2)          # U2: chip select 0, linear address = offset
            p2 += len(t) + 2
        u2[0x7000:0x7000 + len(index)] = index
        # deferred-indirect table index at catalog + 0x43 -> table 1 = tracks 9, 10, 11
        di_index, table1 = 0x7800, 0x7900
        u2[catalog_ofs + 0x43:catalog_ofs + 0x46] = _u24(di_index)
        u2[di_index:di_index + 6] = _u24(0x7A00) + _u24(table1)
        u2[table1:table1 + 6] = _u16(9) + _u16(10) + _u16(11)
        u2[catalog_ofs + 0x46:catalog_ofs + 0x48] = _u16(len(T))
        u2[0x7FFFE] = u2[0x7FFFF] = 0
        ck = checksum(u2)
        u2[0x7FFFE] = (-(ck >> 8)) & 0xFF
        u2[0x7FFFF] = (-(ck & 0xFF)) & 0xFF
        self.images = {c: bytes(b) for c, b in imgs.items()}


# event scripts for the sequencer tests: (tick, kind, value); kind 0 data-port byte, 1 track command,
# 2 master volume, 3 ClearTracks
def port_cmd(tick, track):
    return [(tick, 0, track >> 8), (tick, 0, track & 0xFF)]


SCRIPTS = {
    "main": (150, [(0, 1, 1)] + port_cmd(3, 2) + [(10, 1, 3), (10, 1, 4)] + port_cmd(12, 5) + [(30, 1, 6), (40, 1, 7)]
             + [(50, 0, 0x7F)] + port_cmd(70, 12)                                # a lone first byte times out after 13 ticks
             + [(80, 0, 0x55), (80, 0, 0xAA), (80, 0, 0xB0), (80, 0, 0x4F)]      # master volume 0xB0 through the data port
             + [(84, 0, 0x55), (84, 0, 0xAC), (84, 0, 0x90), (84, 0, 0x6F)]      # channel 1 volume 0x90
             + [(90, 0, 0x55), (90, 0, 0xC2), (91, 0, 0x55), (91, 0, 0xC3)]      # version query
             + [(95, 1, 16), (100, 1, 15), (110, 1, 17), (120, 1, 0), (125, 1, 9999), (126, 0, 0x81), (126, 0, 0x00), (130, 1, 1), (140, 3, 0)]),
    "fatal-opcode": (24, [(0, 1, 1), (5, 1, 13)]),
    "reset-every-tick": (24, [(0, 1, 1), (5, 1, 18), (15, 1, 2)]),
    "invalid-track-type": (12, [(0, 1, 2), (4, 1, 14)]),
    "volume-and-clear": (40, [(0, 2, 0x40), (0, 1, 2), (8, 2, 0xFF), (15, 3, 0), (20, 1, 12), (30, 2, 0)]),
    # only what a caller that holds a plain DCSDecoder* can do (DCSExplorer.cpp:457-488): data-port bytes and
    # SetMasterVolume.  Track commands arrive as byte pairs, like from the WPC board.
    "port-only": (140, port_cmd(0, 1) + port_cmd(3, 2) + port_cmd(10, 3) + port_cmd(10, 4) + port_cmd(12, 5) + port_cmd(30, 6)
                  + port_cmd(40, 7) + [(50, 0, 0x7F)] + port_cmd(70, 12)
                  + [(80, 0, 0x55), (80, 0, 0xAA), (80, 0, 0xB0), (80, 0, 0x4F)]
                  + [(84, 0, 0x55), (84, 0, 0xAC), (84, 0, 0x90), (84, 0, 0x6F)]
                  + [(90, 0, 0x55), (90, 0, 0xC2), (91, 0, 0x55), (91, 0, 0xC3)]
                  + port_cmd(95, 16) + port_cmd(100, 15) + [(105, 2, 0x60)] + port_cmd(110, 17) + port_cmd(120, 0)
                  + port_cmd(125, 9999) + [(126, 0, 0x81), (126, 0, 0x00)] + port_cmd(130, 1) + [(135, 2, 0xE0)]),
}


def zip_recognition_archive(seed):
    """a seeded archive of made-up members for the zip member recognition tests: names with several digits, version numbers,
    upper and lower case; images that start with well-formed, malformed or missing signatures, a JUMP or not.
    -> (members [(name, bytes)], zip base name, zip bytes), or None when the draw gave two members one name"""
    g = splitmix(0x21F0 + seed)
    likely = seed >= 400            # seeds 400..: sets that mostly DO load -- a JUMP image named for chip 2 first, signatures that
    n_members = 2 + next(g) % 9     # mostly carry a digit of their member's name -- so that the U3..U9 rules are what decides
    members = []
    for k in range(n_members):
        stem = ["snd", "s", "u", "rom", "cc", "afm_s", "ng_u", "v1_", "l"][next(g) % 9]
        d1, d2 = next(g) % 10, next(g) % 10
        if likely and k == 0:
            d1 = 2
        name = "%s%d%s%d.%s" % (stem, d1, ["", "_", "v", "S", "-u"][next(g) % 5], d2, ["rom", "bin", "l1", "1_0"][next(g) % 4])
        if next(g) % 4 == 0:
            name = name.upper()
        # (a power of two, as AddROM demands; distinct sizes tell the members apart; 0x2000 is what an absent chip reads as)
        size = [0x100, 0x200, 0x400, 0x800, 0x1000, 0x4000, 0x8000, 0x10000, 0x20000, 0x40000, 0x80000][len(members)]
        img = bytearray(b"\xFF" * size)
        kind = next(g) % 8
        d = next(g) % 10
        if likely:
            kind = 0 if k == 0 else (1 if next(g) % 4 else kind)
            if next(g) % 3:
                d = (d1, d2)[next(g) % 2]
        if kind == 0:
            img[0:4] = bytes([0x18 + next(g) % 4, next(g) % 256, 0x0F | (next(g) % 16) << 4, 0])     # a JUMP
        elif kind in (1, 2, 3):
            text = "%s%s%d %s %02d/%02d/%02d" % ("SU"[next(g) % 2], ["", "ND ", "-"][next(g) % 3], d, ["Sound", "v1.0 L-%d" % (next(g) % 10), ""][next(g) % 3],
                                                  next(g) % 13, next(g) % 32, next(g) % 100)
            img[0:len(text) + 1] = text.encode() + b"\0"
        elif kind == 4:
            text = "S%d no date here" % d
            img[0:len(text) + 1] = text.encode() + b"\0"
        elif kind == 5:
            text = "X%d Sound 01/02/94" % d
            img[0:len(text) + 1] = text.encode() + b"\0"
        members.append((name, bytes(img)))
    if len({len(m[1]) for m in members}) != len(members) or len({m[0].lower() for m in members}) != len(members):
        return None
    zip_base = ["cc_13.zip", "CC_1x.zip", "afm_113b.zip", "sttng_l7.zip", "ccx.zip"][next(g) % 5]
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w", zipfile.ZIP_DEFLATED) as z:
        for name, data in members:
            z.writestr(name, data)
    return members, zip_base, buf.getvalue()
