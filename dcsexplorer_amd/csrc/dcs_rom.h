// dcs_rom.h -- internal: the ROM set behind the dcs_romset_* entry points (dcs_rom.cpp), shared with the
// track-program sequencer (dcs_sequencer.cpp).
#pragma once
#include "dcs_common.h"
#include <string>
#include <vector>

struct DcsRomImage
{
    std::vector<uint8_t> data;
    bool present = false;
};

// a position inside one ROM image; reads past the image give 0xFF
struct DcsRomCursor
{
    const std::vector<uint8_t> *rom = nullptr;
    size_t pos = 0;
    bool isNull() const { return rom == nullptr; }
    void clear() { rom = nullptr; pos = 0; }
    uint32_t u8() { const uint32_t v = (rom != nullptr && pos < rom->size()) ? (*rom)[pos] : 0xFFu; ++pos; return v; }
    uint32_t u16() { const uint32_t h = u8(); return (h << 8) | u8(); }
    uint32_t u24() { const uint32_t h = u16(); return (h << 8) | u8(); }
    void skip(long n) { pos = static_cast<size_t>(static_cast<long>(pos) + n); }
};

struct DcsRomSet
{
    DcsRomImage rom[8];                 // U2..U9
    std::vector<uint8_t> missing;       // stand-in for unpopulated chips: 8 KB of 0xFF (DCSDecoder.cpp:262-281)
    uint32_t catalogOfs = 0;
    uint32_t trackIndex = 0;            // offset in U2 of the 3-byte-per-track index
    uint32_t indirectIndex = 0;         // offset in U2 of the deferred-indirect table index (catalog + 0x43)
    uint32_t nTracks = 0;
    int hw = DCS_HW_UNKNOWN;
    int os = -1;                        // DcsOsVersion, -1 = not known
    uint32_t nominalVersion = 0;
    std::string lastError;

    const std::vector<uint8_t> &image(int chipSelect) const
    {
        return rom[chipSelect & 7].present ? rom[chipSelect & 7].data : missing;
    }
    // MakeROMPointer (DCSDecoder.cpp:68-76): chip select from bits 21.. (DCS-95) or 20.. (DCS-93), offset masked
    // to the ROM size
    DcsRomCursor at(uint32_t linear) const
    {
        const int cs = static_cast<int>((linear >> (hw == DCS_HW_DCS95 ? 21 : 20)) & 7);
        const std::vector<uint8_t> &img = image(cs);
        return DcsRomCursor{ &img, static_cast<size_t>(linear & static_cast<uint32_t>(img.size() - 1)) };
    }
    // a U24 stored in U2 at `ofs` (the catalog's own tables are addressed this way, not through MakeROMPointer)
    uint32_t u2U24(size_t ofs) const
    {
        const std::vector<uint8_t> &d = rom[0].data;
        return ofs + 2 < d.size() ? (static_cast<uint32_t>(d[ofs]) << 16) | (static_cast<uint32_t>(d[ofs + 1]) << 8) | d[ofs + 2] : 0xFFFFFFu;
    }
};
