// dcs_rom.cpp -- ROM ingestion on the input side of the decode path: from a set of sound ROM images (or a
// PinMame .zip of them) to the list of audio streams the batch decoder is fed with.
//
// Mirrors, with the same results on the same images:
//   DCSDecoder::AddROM / FindCatalog / GetSignature        DCSDecoder.cpp:26-66, :92-121, :207-234
//   DCSDecoder::CheckROMs (checksums, board and OS version)  DCSDecoder.cpp:236-495
//   DCSDecoder::MakeROMPointer (93 / 95 addressing)        DCSDecoder.cpp:68-76
//   DCSDecoder::GetTrackInfo / DecompileTrackProgram       DCSDecoder.cpp:672-905, :907-1160
//   DCSDecoder::ListStreams                                DCSDecoder.cpp:1248-1293
//   the stream loop of ExtractTracksOrStreams (which stream, at which mixing level)   DCSExplorer.cpp:1742-1810
//   DCSDecoder::LoadROMFromZipFile (file recognition heuristics)   DCSDecoderZipLoader.cpp:60-207
// Unlike the reference every read is bounds-checked: a pointer that leaves its ROM image reads 0xFF, the
// value the hardware returns for unpopulated ROM space (DCSDecoder.cpp:262-281).
#include "dcs_common.h"
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <regex>
#include <set>
#include <string>
#include <vector>

#include "dcs_rom.h"

namespace {

uint32_t be16(const std::vector<uint8_t> &d, size_t o) { return o + 1 < d.size() ? (static_cast<uint32_t>(d[o]) << 8) | d[o + 1] : 0xFFFFu; }
uint32_t be24(const std::vector<uint8_t> &d, size_t o) { return o + 2 < d.size() ? (static_cast<uint32_t>(d[o]) << 16) | (static_cast<uint32_t>(d[o + 1]) << 8) | d[o + 2] : 0xFFFFFFu; }

bool isJump(const uint8_t *p) { return (p[0] & 0xFC) == 0x18 && (p[2] & 0x0F) == 0x0F; }     // ADSP-2105 JUMP

// checksum of a ROM as the catalog stores it (:652-669): low bytes of the sums of the even- and odd-offset bytes
uint32_t romChecksum(const std::vector<uint8_t> &d)
{
    uint32_t even = 0, odd = 0;
    for (size_t i = 0 ; i + 1 < d.size() ; i += 2) { even += d[i]; odd += d[i + 1]; }
    return ((even << 8) & 0xFF00u) | (odd & 0xFFu);
}

uint32_t findCatalog(const std::vector<uint8_t> &u2)
{
    static const uint32_t offsets[] = { 0x3000, 0x4000, 0x6000 };
    for (uint32_t ofs : offsets)
    {
        if (ofs + 6 > u2.size())
            continue;
        const uint32_t size = be16(u2, ofs) * 4096u, chipSel = be16(u2, ofs + 2) >> 8, ck = be16(u2, ofs + 4);
        if (chipSel == 0 && ck == 0 && size == u2.size())
            return ofs;
    }
    return 0;
}

// A sequence of 24-bit ADSP-2105 opcodes (stored big-endian in 4-byte units) with wildcard nibbles.  Pattern
// syntax as the reference's SearchForOpcodes (:1763-1900): hex digits match, '*' matches anything, a letter
// matches anything and collects the nibble into the variable of that name.
struct OpPattern { uint32_t value, mask; uint32_t varMask; };
int searchOpcodes(const char *pattern, const std::vector<uint8_t> &rom, size_t startByte, size_t nBytes, char var, uint32_t *varOut)
{
    std::vector<OpPattern> ops;
    for (const char *p = pattern ; *p != 0 ; )
    {
        while (*p == ' ') ++p;
        if (*p == 0) break;
        OpPattern op{ 0, 0, 0 };
        for (int i = 0 ; i < 6 && *p != 0 && *p != ' ' ; ++i, ++p)
        {
            const char c = *p;
            op.value <<= 4; op.mask <<= 4; op.varMask <<= 4;
            if (isxdigit(static_cast<unsigned char>(c)))
            {
                op.value |= static_cast<uint32_t>(c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10);
                op.mask |= 0xF;
            }
            else if (c == var)
                op.varMask |= 0xF;
        }
        ops.push_back(op);
    }
    if (ops.empty() || startByte + nBytes > rom.size())
        nBytes = rom.size() > startByte ? rom.size() - startByte : 0;
    const size_t nOps = nBytes / 4;
    for (size_t i = 0 ; i + ops.size() <= nOps ; ++i)
    {
        bool ok = true;
        for (size_t k = 0 ; k < ops.size() && ok ; ++k)
            ok = (be24(rom, startByte + (i + k) * 4) & ops[k].mask) == ops[k].value;
        if (!ok)
            continue;
        if (varOut != nullptr)
            for (size_t k = 0 ; k < ops.size() ; ++k)
                if (ops[k].varMask != 0)
                {
                    // the first run of the variable: its nibbles, right-aligned
                    uint32_t m = ops[k].varMask, v = be24(rom, startByte + (i + k) * 4) & m;
                    while ((m & 1) == 0) { m >>= 1; v >>= 1; }
                    *varOut = v;
                    break;
                }
        return static_cast<int>(i * 4);
    }
    return -1;
}

// operand bytes after the 3-byte (delay, opcode) prefix as DecompileTrackProgram reads them (:956-1130); -1 = invalid
int operandBytes(int opcode, int os)
{
    switch (opcode)
    {
    case 0x00: case 0x0D: case 0x0F: return 0;
    case 0x01: return 5;
    case 0x02: case 0x05: case 0x0E: return 1;
    case 0x03: case 0x06: case 0x07: case 0x08: case 0x09: case 0x10: return 2;
    case 0x04: return os == DCS_OS93A ? 3 : 1;
    case 0x0A: case 0x0B: case 0x0C: case 0x11: case 0x12: return 4;
    default: return -1;
    }
}

const size_t kMaxProgramSteps = 1 << 16;        // a track program that has not ended by then is garbage

}   // namespace

extern "C" DcsRomSet *dcs_romset_create(void)
{
    DcsRomSet *rs = new (std::nothrow) DcsRomSet;
    if (rs != nullptr)
        rs->missing.assign(0x2000, 0xFF);
    return rs;
}

extern "C" void dcs_romset_destroy(DcsRomSet *rs) { delete rs; }

extern "C" const char *dcs_romset_last_error(const DcsRomSet *rs) { return rs != nullptr ? rs->lastError.c_str() : ""; }

extern "C" DcsStatus dcs_romset_add_rom(DcsRomSet *rs, int chip, const uint8_t *data, size_t size)
{
    if (rs == nullptr || data == nullptr || chip < 2 || chip > 9)
        return DCS_ERR_INVALID_ARG;
    if (size == 0)
        return DCS_OK;                              // ignored, as in AddROM (:33-34)
    if ((size & (size - 1)) != 0)
    {
        rs->lastError = "ROM size is not a power of two";      // MakeROMPointer masks offsets with size - 1
        return DCS_ERR_INVALID_ARG;
    }
    DcsRomImage &r = rs->rom[chip - 2];
    r.data.assign(data, data + size);
    r.present = true;
    if (chip == 2)
    {
        rs->catalogOfs = findCatalog(r.data);
        rs->trackIndex = 0; rs->indirectIndex = 0; rs->nTracks = 0;
        if (rs->catalogOfs != 0)
        {
            rs->trackIndex = be24(r.data, rs->catalogOfs + 0x40);
            rs->indirectIndex = be24(r.data, rs->catalogOfs + 0x43);
            rs->nTracks = be16(r.data, rs->catalogOfs + 0x46);
        }
    }
    return DCS_OK;
}

extern "C" DcsStatus dcs_romset_set_version(DcsRomSet *rs, int hw, int os)
{
    if (rs == nullptr || (hw != DCS_HW_DCS93 && hw != DCS_HW_DCS95) || os < DCS_OS93A || os > DCS_OS95)
        return DCS_ERR_INVALID_ARG;
    rs->hw = hw;
    rs->os = os;
    return DCS_OK;
}

// CheckROMs (:236-495): validates the images against the catalog's size/checksum table and infers the board
// and OS version from where the catalog sits and from code patterns of the decoder program in U2.
extern "C" DcsStatus dcs_romset_check(DcsRomSet *rs, DcsRomCheck *out)
{
    if (rs == nullptr || out == nullptr)
        return DCS_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    out->status = 2;
    out->hw = DCS_HW_INVALID;
    out->os = -1;
    out->catalogOffset = rs->catalogOfs;
    out->nTracks = rs->nTracks;
    const std::vector<uint8_t> &u2 = rs->rom[0].data;
    if (!rs->rom[0].present || u2.size() < 0x7000)
        return DCS_OK;

    // signature (:97-121): JUMP at 0, printable text from offset 4 up to a NUL within 120 bytes
    if (isJump(u2.data()))
    {
        size_t len = 0;
        while (len < 120 && u2[4 + len] >= 32 && u2[4 + len] < 127) ++len;
        if (u2[4 + len] == 0)
            memcpy(out->signature, &u2[4], len);
    }

    uint32_t checksum[8] = { 0 };
    int nPopulated = 0;
    for (int i = 0 ; i < 8 ; ++i)
        if (rs->rom[i].present)
        {
            checksum[i] = romChecksum(rs->rom[i].data);
            ++nPopulated;
        }

    static const uint32_t offsets[] = { 0x3000, 0x4000, 0x6000 };
    for (uint32_t ofs : offsets)
    {
        int nInTable = 0, nValidated = 0, firstFailed = -1;
        for (int entry = 0 ; entry < 9 ; ++entry)
        {
            const size_t e = ofs + static_cast<size_t>(entry) * 6;
            const uint32_t size = be16(u2, e) * 4096u;
            uint32_t chipSel = be16(u2, e + 2) >> 8;
            const uint32_t ck = be16(u2, e + 4);
            if (size == 0)
                break;
            ++nInTable;
            if (ofs == 0x6000)
                chipSel >>= 1;                      // DCS-95: smaller banking window, select shifted once more
            if (chipSel < 8 && rs->rom[chipSel].present && rs->rom[chipSel].data.size() == size && checksum[chipSel] == ck)
                ++nValidated;
            else
            {
                firstFailed = entry;
                break;
            }
        }
        if (nValidated == 0)
            continue;

        if (ofs == 0x6000)
        {
            out->hw = DCS_HW_DCS95;
            out->os = DCS_OS95;
            // version query handler of the 1996+ software: SR0 = $01xx ... (:372-389)
            uint32_t v = 0;
            if (searchOpcodes("4vvvvE 0F16F8 93300E 18***F 4wwwwE 0F1608 0F16F8 93300E 18***F", u2, 0x2000 + 0x300 * 4, 0x180 * 4, 'v', &v) >= 0)
                out->nominalVersion = v & 0xFFFFu;
        }
        else
        {
            out->hw = DCS_HW_DCS93;
            out->os = DCS_OS94;
            if (searchOpcodes("380026 3C1005 0C00C0", u2, 0x1000 + 0x100 * 4, 0x180 * 4, 0, nullptr) >= 0)
            {
                out->os = DCS_OS93B;
                if (searchOpcodes("47FFF2 47C946", u2, 0x2000 + 0x200 * 4, 0x100 * 4, 0, nullptr) >= 0)
                    out->os = DCS_OS93A;
            }
        }
        rs->hw = out->hw;
        rs->os = out->os;
        rs->nominalVersion = out->nominalVersion;
        out->status = (nValidated == nPopulated && nPopulated == nInTable) ? 1 : firstFailed + 2;
        return DCS_OK;
    }
    return DCS_OK;
}

extern "C" DcsStatus dcs_romset_pointer(const DcsRomSet *rs, uint32_t linear, const uint8_t **p, size_t *avail, int *chip)
{
    if (rs == nullptr || p == nullptr || avail == nullptr)
        return DCS_ERR_INVALID_ARG;
    const DcsRomCursor c = rs->at(linear);
    *p = c.rom->data() + c.pos;
    *avail = c.rom->size() - c.pos;
    if (chip != nullptr)
        *chip = 2 + static_cast<int>((linear >> (rs->hw == DCS_HW_DCS95 ? 21 : 20)) & 7);
    return DCS_OK;
}

// bytes from `p` to the end of the ROM image of this set that contains it (0: p points into none of them)
extern "C" size_t dcs_romset_bytes_behind(const DcsRomSet *rs, const uint8_t *p)
{
    if (rs == nullptr || p == nullptr)
        return 0;
    for (int c = 0 ; c < 8 ; ++c)
    {
        const std::vector<uint8_t> &img = rs->image(c);
        if (!img.empty() && p >= img.data() && p < img.data() + img.size())
            return static_cast<size_t>(img.data() + img.size() - p);
    }
    return 0;
}

extern "C" uint32_t dcs_romset_num_tracks(const DcsRomSet *rs) { return rs != nullptr ? rs->nTracks : 0; }

extern "C" DcsStatus dcs_romset_track_info(const DcsRomSet *rs, uint32_t track, DcsTrackInfo *ti)
{
    if (rs == nullptr || ti == nullptr)
        return DCS_ERR_INVALID_ARG;
    memset(ti, 0, sizeof(*ti));
    ti->deferCode = 0xFFFF;
    if (track >= rs->nTracks || !rs->rom[0].present)
        return DCS_ERR_BAD_STREAM;
    const uint32_t addr = be24(rs->rom[0].data, rs->trackIndex + static_cast<size_t>(track) * 3);
    if ((addr & 0x00FF0000u) == 0x00FF0000u)
        return DCS_ERR_BAD_STREAM;                  // not populated
    DcsRomCursor p = rs->at(addr);
    const uint32_t type = p.u8(), ch = p.u8();
    if (ch > 7)
        return DCS_ERR_BAD_STREAM;
    bool done = false;
    uint32_t deferCode = 0xFFFF;
    if (type == 2 || type == 3)
    {
        deferCode = p.u16();
        done = true;
    }
    else if (type != 1)
        return DCS_ERR_BAD_STREAM;

    // running time of the program in frames (:733-893): wait counters, nested loops, looping streams
    struct Level { uint64_t programTime = 0, loopingStreamTime = 0; uint32_t nLoops = 1; bool looping = false; };
    std::vector<Level> stack(1);
    for (size_t step = 0 ; !done && step < kMaxProgramSteps ; ++step)
    {
        const uint32_t counter = p.u16(), opcode = p.u8();
        if (counter == 0xFFFF)
        {
            stack.back().looping = true;
            stack.back().programTime += stack.back().loopingStreamTime;
            break;
        }
        stack.back().programTime += counter;
        switch (opcode)
        {
        case 0x00: done = true; break;
        case 0x01:
            {
                p.u8();
                DcsRomCursor stream = rs->at(p.u24());
                const uint32_t repeat = p.u8();
                const uint32_t streamTime = stream.u16();
                stack.back().loopingStreamTime = repeat == 0 ? streamTime : 0;
            }
            break;
        case 0x0E:
            stack.emplace_back();
            if ((stack.back().nLoops = p.u8()) == 0)
                stack.back().looping = true;
            break;
        case 0x0F:
            if (stack.size() > 1)
            {
                const Level lv = stack.back();
                stack.pop_back();
                stack.back().programTime += (lv.looping ? 1 : lv.nLoops) * lv.programTime;
                if (lv.looping)
                {
                    stack.back().looping = true;
                    done = true;
                }
            }
            break;
        case 0x0D: break;
        case 0x02: case 0x05: p.skip(1); break;
        case 0x03: case 0x06: case 0x07: case 0x08: case 0x09: case 0x11: case 0x12: p.skip(2); break;
        case 0x0A: case 0x0B: case 0x0C: p.skip(4); break;
        case 0x04: p.skip(rs->os == DCS_OS93A ? 3 : 1); break;
        default: break;                             // the time scan ignores what it does not know (:872-893)
        }
    }
    while (stack.size() > 1)
    {
        const Level lv = stack.back();
        stack.pop_back();
        stack.back().programTime += (lv.nLoops == 0 ? 1 : lv.nLoops) * lv.programTime;
        if (lv.looping)
            stack.back().looping = true;
    }
    ti->address = addr;
    ti->channel = static_cast<int32_t>(ch);
    ti->type = static_cast<int32_t>(type);
    ti->deferCode = static_cast<int32_t>(deferCode);
    ti->time = static_cast<uint32_t>(stack.back().programTime);
    ti->looping = stack.back().looping ? 1 : 0;
    return DCS_OK;
}

static DcsStatus decompile(const DcsRomSet *rs, uint32_t track, std::vector<DcsTrackOp> &v)
{
    v.clear();
    DcsTrackInfo ti;
    if (dcs_romset_track_info(rs, track, &ti) != DCS_OK || ti.type != 1)
        return DCS_OK;                              // no byte-code program: empty list, as the reference
    DcsRomCursor p = rs->at(ti.address);
    const size_t start = p.pos;
    p.skip(2);
    std::vector<int> loops;
    for (bool done = false ; !done && v.size() < kMaxProgramSteps ; )
    {
        DcsTrackOp op;
        memset(&op, 0, sizeof(op));
        op.nestingLevel = static_cast<int32_t>(loops.size());
        op.loopParent = loops.empty() ? -1 : loops.back();
        op.offset = static_cast<int32_t>(p.pos - start);
        op.delayCount = static_cast<uint16_t>(p.u16());
        if (op.delayCount == 0xFFFF)
            done = true;
        op.opcode = static_cast<uint8_t>(p.u8());
        const int n = operandBytes(op.opcode, rs->os);
        if (n < 0 || op.opcode == 0x00)
            done = true;
        op.nOperandBytes = static_cast<uint8_t>(n < 0 ? 0 : n);
        for (int i = 0 ; i < op.nOperandBytes ; ++i)
        {
            const uint32_t b = p.u8();
            if (i < 8) op.operandBytes[i] = static_cast<uint8_t>(b);
        }
        v.push_back(op);
        if (op.opcode == 0x0E)
            loops.push_back(static_cast<int>(v.size()));        // (sic) one past the push instruction (:1088)
        else if (op.opcode == 0x0F && !loops.empty())
            loops.pop_back();
    }
    return DCS_OK;
}

extern "C" DcsStatus dcs_romset_decompile(const DcsRomSet *rs, uint32_t track, DcsTrackOp *ops, uint32_t cap, uint32_t *nOut)
{
    if (rs == nullptr || nOut == nullptr || (ops == nullptr && cap != 0))
        return DCS_ERR_INVALID_ARG;
    std::vector<DcsTrackOp> v;
    decompile(rs, track, v);
    *nOut = static_cast<uint32_t>(v.size());
    if (cap < v.size())
        return cap == 0 ? DCS_OK : DCS_ERR_CAPACITY;
    if (!v.empty())
        memcpy(ops, v.data(), sizeof(DcsTrackOp) * v.size());
    return DCS_OK;
}

extern "C" DcsStatus dcs_romset_list_streams(const DcsRomSet *rs, uint32_t *addrs, uint32_t cap, uint32_t *nOut)
{
    if (rs == nullptr || nOut == nullptr || (addrs == nullptr && cap != 0))
        return DCS_ERR_INVALID_ARG;
    std::set<uint32_t> streams;
    std::vector<DcsTrackOp> v;
    for (uint32_t t = 0 ; t < rs->nTracks ; ++t)
    {
        decompile(rs, t, v);
        for (const DcsTrackOp &op : v)
            if (op.opcode == 0x01)
                streams.insert((static_cast<uint32_t>(op.operandBytes[1]) << 16) | (static_cast<uint32_t>(op.operandBytes[2]) << 8) | op.operandBytes[3]);
    }
    *nOut = static_cast<uint32_t>(streams.size());
    if (cap < streams.size())
        return cap == 0 ? DCS_OK : DCS_ERR_CAPACITY;
    uint32_t i = 0;
    for (uint32_t a : streams)
        addrs[i++] = a;
    return DCS_OK;
}

// Which streams `--extract-streams` extracts, in its order, and the mixing level it plays each at
// (DCSExplorer.cpp:1742-1810): per track, follow the level opcodes (default 0x64 per channel), take every
// Play opcode's stream the first time its address is seen.
extern "C" DcsStatus dcs_romset_extract_plan(const DcsRomSet *rs, DcsExtractItem *items, uint32_t cap, uint32_t *nOut)
{
    if (rs == nullptr || nOut == nullptr || (items == nullptr && cap != 0))
        return DCS_ERR_INVALID_ARG;
    std::vector<DcsExtractItem> plan;
    std::set<uint32_t> seen;
    std::vector<DcsTrackOp> v;
    for (uint32_t t = 0 ; t < rs->nTracks ; ++t)
    {
        decompile(rs, t, v);
        int level[8] = { 0x64, 0x64, 0x64, 0x64, 0x64, 0x64, 0x64, 0x64 };
        uint32_t streamNum = 0;
        for (const DcsTrackOp &op : v)
        {
            const int ch = op.operandBytes[0] & 7;
            if (op.opcode == 0x07 || op.opcode == 0x0A)
                level[ch] = op.operandBytes[1];
            else if (op.opcode == 0x08 || op.opcode == 0x0B)
                level[ch] += op.operandBytes[1];
            else if (op.opcode == 0x09)                     // (0x0C, the timed decrease, is not followed there: :1778)
                level[ch] -= op.operandBytes[1];
            else if (op.opcode == 0x01)
            {
                const uint32_t addr = (static_cast<uint32_t>(op.operandBytes[1]) << 16) | (static_cast<uint32_t>(op.operandBytes[2]) << 8) | op.operandBytes[3];
                if (seen.insert(addr).second)
                    plan.push_back(DcsExtractItem{ t, ++streamNum, addr, level[ch] });
            }
        }
    }
    *nOut = static_cast<uint32_t>(plan.size());
    if (cap < plan.size())
        return cap == 0 ? DCS_OK : DCS_ERR_CAPACITY;
    if (!plan.empty())
        memcpy(items, plan.data(), sizeof(DcsExtractItem) * plan.size());
    return DCS_OK;
}

// Which tracks `--extract-tracks` extracts and how many frames of each (DCSExplorer.cpp:1735-1925): type-1 tracks with at
// least one Play opcode in their program; ExtractToWAV takes the running time as a uint16_t and adds two (:1667-1672)
extern "C" DcsStatus dcs_romset_extract_tracks_plan(const DcsRomSet *rs, DcsExtractTrack *items, uint32_t cap, uint32_t *nOut)
{
    if (rs == nullptr || nOut == nullptr || (items == nullptr && cap != 0))
        return DCS_ERR_INVALID_ARG;
    std::vector<DcsExtractTrack> plan;
    std::vector<DcsTrackOp> v;
    for (uint32_t t = 0 ; t < rs->nTracks ; ++t)
    {
        DcsTrackInfo ti;
        if (dcs_romset_track_info(rs, t, &ti) != DCS_OK || ti.type != 1)
            continue;
        decompile(rs, t, v);
        bool plays = false;
        for (const DcsTrackOp &op : v)
            plays = plays || op.opcode == 0x01;
        if (plays)
            plan.push_back(DcsExtractTrack{ t, static_cast<uint32_t>(static_cast<uint16_t>(static_cast<uint16_t>(ti.time) + 2)) });
    }
    *nOut = static_cast<uint32_t>(plan.size());
    if (cap < plan.size())
        return cap == 0 ? DCS_OK : DCS_ERR_CAPACITY;
    if (!plan.empty())
        memcpy(items, plan.data(), sizeof(DcsExtractTrack) * plan.size());
    return DCS_OK;
}

// The plan as input of dcs_decode_stream_sequence / dcs_decode_streams: each stream where it lies in its ROM image
extern "C" DcsStatus dcs_romset_stream_refs(const DcsRomSet *rs, const DcsExtractItem *items, uint32_t n, int volume, DcsStreamRef *refs)
{
    if (rs == nullptr || items == nullptr || refs == nullptr || rs->os < 0)
        return DCS_ERR_INVALID_ARG;
    for (uint32_t i = 0 ; i < n ; ++i)
    {
        const DcsRomCursor c = rs->at(items[i].address);
        refs[i].data = c.rom->data() + c.pos;
        refs[i].len = c.rom->size() - c.pos;
        refs[i].os = rs->os;
        refs[i].volume = volume;
        refs[i].level = items[i].level;
        refs[i].channelVolume = 0xFF;
    }
    return DCS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// PinMame ROM .zip: a minimal reader (central directory, stored and deflated members; raw inflate by zlib)
// and the reference's heuristics for telling which member is which chip (DCSDecoderZipLoader.cpp:106-203).
// ---------------------------------------------------------------------------------------------------------
namespace {

struct ZipMember { std::string name; std::vector<uint8_t> data; int chip = -1; };

uint32_t le16(const uint8_t *p) { return p[0] | (p[1] << 8); }
uint32_t le32(const uint8_t *p) { return le16(p) | (le16(p + 2) << 16); }

bool readZip(const std::vector<uint8_t> &z, std::vector<ZipMember> &out, std::string &err)
{
    if (z.size() < 22) { err = "not a zip file"; return false; }
    // end-of-central-directory record: scan back over a possible comment
    size_t eocd = std::string::npos;
    for (size_t i = z.size() - 22 ; ; --i)
    {
        if (le32(&z[i]) == 0x06054b50) { eocd = i; break; }
        if (i == 0 || z.size() - i > 22 + 65535) break;
    }
    if (eocd == std::string::npos) { err = "zip end record not found"; return false; }
    const uint32_t nEntries = le16(&z[eocd + 10]);
    size_t cd = le32(&z[eocd + 16]);
    for (uint32_t e = 0 ; e < nEntries ; ++e)
    {
        if (cd + 46 > z.size() || le32(&z[cd]) != 0x02014b50) { err = "bad zip directory entry"; return false; }
        const uint32_t method = le16(&z[cd + 10]), csize = le32(&z[cd + 20]), usize = le32(&z[cd + 24]);
        const uint32_t nlen = le16(&z[cd + 28]), xlen = le16(&z[cd + 30]), clen = le16(&z[cd + 32]);
        const size_t lho = le32(&z[cd + 42]);
        if (cd + 46 + nlen > z.size()) { err = "bad zip directory entry"; return false; }
        std::string name(reinterpret_cast<const char *>(&z[cd + 46]), nlen);
        cd += 46 + static_cast<size_t>(nlen) + xlen + clen;
        if (!name.empty() && name.back() == '/')
            continue;                               // a directory
        if (lho + 30 > z.size() || le32(&z[lho]) != 0x04034b50) { err = "bad zip local header"; return false; }
        const size_t dataOfs = lho + 30 + le16(&z[lho + 26]) + le16(&z[lho + 28]);
        if (dataOfs + csize > z.size() || usize > (64u << 20)) { err = "zip member out of range"; return false; }
        ZipMember m;
        m.name = name;
        m.data.resize(usize);
        if (method == 0)
        {
            if (csize != usize) { err = "bad stored zip member"; return false; }
            memcpy(m.data.data(), &z[dataOfs], usize);
        }
        else if (method == 8)
        {
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) { err = "zlib init failed"; return false; }
            zs.next_in = const_cast<Bytef *>(&z[dataOfs]);
            zs.avail_in = csize;
            zs.next_out = m.data.data();
            zs.avail_out = usize;
            const int r = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (r != Z_STREAM_END || zs.total_out != usize) { err = "error uncompressing " + name; return false; }
        }
        else { err = "unsupported zip compression method in " + name; return false; }
        out.push_back(std::move(m));
    }
    return true;
}

}   // namespace

extern "C" DcsStatus dcs_romset_load_zip_memory(DcsRomSet *rs, const uint8_t *zip, size_t len, const char *zipBaseName,
                                                const char *explicitU2)
{
    if (rs == nullptr || zip == nullptr)
        return DCS_ERR_INVALID_ARG;
    std::vector<ZipMember> members;
    if (!readZip(std::vector<uint8_t>(zip, zip + len), members, rs->lastError))
        return DCS_ERR_BAD_STREAM;

    // U2: starts with a JUMP and has a '2' in its name, or is named explicitly (:123-139)
    ZipMember *u2 = nullptr;
    for (ZipMember &m : members)
    {
        const bool named = explicitU2 != nullptr && strcasecmp(m.name.c_str(), explicitU2) == 0;
        if ((m.data.size() >= 3 && isJump(m.data.data()) && m.name.find('2') != std::string::npos) || named)
        {
            u2 = &m;
            m.chip = 2;
            break;
        }
    }
    if (u2 == nullptr)
    {
        rs->lastError = "no file could be identified as ROM U2";
        return DCS_ERR_BAD_STREAM;
    }
    DcsStatus st = dcs_romset_add_rom(rs, 2, u2->data.data(), u2->data.size());
    if (st != DCS_OK)
        return st;

    // U3..U9: the digit in the file name AND in the image's own signature "[SU]<n> ... mm/dd/yy" (:160-200);
    // Cactus Canyon's U7 calls itself U6 (:180-184)
    const std::regex sig("[SU]([^\\d]*)(\\d).*?\\s+\\d\\d/\\d\\d/\\d\\d");
    const bool cactusCanyon = zipBaseName != nullptr && std::regex_match(zipBaseName, std::regex("^cc_\\d.*", std::regex_constants::icase));
    for (int n = 3 ; n <= 9 ; ++n)
    {
        const char digit = static_cast<char>('0' + n);
        for (ZipMember &m : members)
        {
            if (m.chip >= 0 || m.name.find(digit) == std::string::npos)
                continue;
            // the signature is the NUL-terminated text at the start of the image
            size_t slen = 0;
            while (slen < m.data.size() && slen < 256 && m.data[slen] != 0) ++slen;
            const std::string text(reinterpret_cast<const char *>(m.data.data()), slen);
            std::smatch mr;
            const bool isMatch = slen < m.data.size() && slen < 256 && std::regex_match(text, mr, sig);
            const char sigDigit = isMatch ? mr[2].str()[0] : 0;
            bool load = sigDigit == digit;
            if (cactusCanyon && isMatch && digit == '7' && sigDigit == '6')
                load = true;
            if (load)
            {
                st = dcs_romset_add_rom(rs, n, m.data.data(), m.data.size());
                if (st != DCS_OK)
                    return st;
                m.chip = n;
                break;
            }
        }
    }
    return DCS_OK;
}

extern "C" DcsStatus dcs_romset_load_zip(DcsRomSet *rs, const char *path, const char *explicitU2)
{
    if (rs == nullptr || path == nullptr)
        return DCS_ERR_INVALID_ARG;
    FILE *fp = fopen(path, "rb");
    if (fp == nullptr)
    {
        rs->lastError = std::string("cannot open ") + path;
        return DCS_ERR_INVALID_ARG;
    }
    std::vector<uint8_t> z;
    uint8_t buf[65536];
    for (size_t n ; (n = fread(buf, 1, sizeof(buf), fp)) != 0 ; )
        z.insert(z.end(), buf, buf + n);
    fclose(fp);
    const char *base = strrchr(path, '/');
    return dcs_romset_load_zip_memory(rs, z.data(), z.size(), base != nullptr ? base + 1 : path, explicitU2);
}
