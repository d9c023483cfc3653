// dcs_params.cpp -- the per-frame numbers the sequencer hands to the hot path: the master volume
// multiplier, a channel's mixing multiplier, and the frame's shared fixed-point scale (volShift).
// Host arithmetic, restating DCSDecoderNative::SetMasterVolume (DCSDecoderNative.cpp:3250-3282),
// UpdateMixingLevels (:3072-3121) and the scale block of MainLoop (:227-269).
#include "dcs_common.h"

namespace {

// redundant-sign-bit count of a 32-bit value, negated: what the ADSP-2105 EXP instruction yields
// for a double-word operand (CalcExp32, :3447-3459).  -31 for 0.
int expOf(uint32_t x)
{
    const uint32_t sign = x >> 31;
    int e = 0;
    while (e > -31 && ((x >> 30) & 1) == sign)
    {
        x <<= 1;
        --e;
    }
    // a negative value keeps shifting while bit 30 is set; all-ones would never stop in the
    // reference, but the operand here is a sum of non-negative products shifted right by two
    return e;
}

// 1.15 x 1.15 -> 1.15, truncating, operands known to be non-negative
inline uint32_t mulQ15(uint32_t a, uint32_t b) { return ((a * b) >> 15) & 0xFFFF; }

}   // namespace

extern "C" uint16_t dcs_volume_multiplier(int volume)
{
    // 0.5 * 0.981201^(255 - vol) by square-and-multiply over the ZERO bits of vol (:3266-3275)
    if (volume == 0)
        return 0;
    uint32_t bits = static_cast<uint16_t>(volume);
    uint32_t acc = 0x3FFF, step = 0x7D98;
    for (int i = 0 ; i < 8 ; ++i, bits >>= 1)
    {
        if ((bits & 1) == 0)
            acc = mulQ15(acc, step);
        step = mulQ15(step, step);
    }
    return static_cast<uint16_t>(acc << 1);
}

extern "C" uint16_t dcs_mixing_multiplier(DcsOsVersion os, int levelSum, int channelVolume)
{
    // clamp, take the high bits as an 8-bit attenuation exponent, 0.9733^(255 - exp) (:3080-3120)
    if (levelSum > 8191) levelSum = 8191;
    if (levelSum < -8191) levelSum = -8191;
    const uint32_t exp = static_cast<uint16_t>(((levelSum >> 6) & 0x3FF) + 0x80);
    uint32_t acc = (os == DCS_OS93A) ? 0x7FFFu : static_cast<uint16_t>(channelVolume << 7);
    uint32_t step = 0x7C94;
    for (int j = 0 ; j < 8 ; ++j)
    {
        if (((exp >> j) & 1) == 0)
            acc = mulQ15(acc, step);
        step = mulQ15(step, step);
    }
    return static_cast<uint16_t>(acc << 1);
}

// The scale block of MainLoop (:227-269) in its general form: channel i counts towards the sum when
// counted[i] is set and is multiplied by vol[i] (the master multiplier, or 0x7FFE for a channel with the
// "maximum mixing level" override, :232-236, :262-266).
int dcsFrameScaleV(const uint16_t *vol, uint16_t *mixMul, const uint8_t *counted, int nch)
{
    // sum of (mixing multiplier x master multiplier) over the channels with a stream, in 4.28 after
    // the >>2; its exponent picks the shift that keeps the mixed spectrum inside 1.15 (:227-260)
    uint64_t sum = 0;
    for (int i = 0 ; i < nch ; ++i)
        if (counted == nullptr || counted[i])
            sum += static_cast<uint64_t>(mixMul[i]) * vol[i];
    sum >>= 2;
    int shift = -(expOf(static_cast<uint32_t>(sum)) + 3);
    shift = shift < 0 ? 0 : shift > 8 ? 8 : shift;
    for (int i = 0 ; i < nch ; ++i)
    {
        const uint64_t m = (static_cast<uint64_t>(mixMul[i]) * vol[i]) << 1;       // :264-269
        mixMul[i] = static_cast<uint16_t>((m << shift) >> 16);
    }
    return shift;
}

extern "C" int dcs_frame_scale(uint16_t volMult, uint16_t *mixMul, const uint8_t *active, int nch)
{
    uint16_t vol[DCS_MAX_CHANNELS];
    if (nch < 0 || nch > DCS_MAX_CHANNELS)
        return 0;
    for (int i = 0 ; i < nch ; ++i)
        vol[i] = volMult;
    return dcsFrameScaleV(vol, mixMul, active, nch);
}

extern "C" DcsStatus dcs_stream_params(DcsOsVersion os, int volume, int level, int channelVolume,
                                       uint32_t nFrames, uint16_t *mixMulScaled, uint8_t *volShift)
{
    // a fresh decoder: Channel::mixingMultiplier initialiser (DCSDecoderNative.h:514)
    return dcs_stream_params_from(os, volume, level, channelVolume, 0x7FFF, nFrames, mixMulScaled, volShift);
}

extern "C" DcsStatus dcs_stream_params_from(DcsOsVersion os, int volume, int level, int channelVolume,
                                            uint16_t firstMixMul, uint32_t nFrames,
                                            uint16_t *mixMulScaled, uint8_t *volShift)
{
    if (mixMulScaled == nullptr || volShift == nullptr)
        return DCS_ERR_INVALID_ARG;
    const uint16_t volMult = dcs_volume_multiplier(volume);
    const uint16_t steady = dcs_mixing_multiplier(os, level * 64, channelVolume);      // (level byte << 6; levels may be negative)
    uint16_t mm = firstMixMul;      // what the previous tick's UpdateMixingLevels left in the channel
    for (uint32_t f = 0 ; f < nFrames ; ++f)
    {
        uint16_t scaled = mm;
        volShift[f] = static_cast<uint8_t>(dcs_frame_scale(volMult, &scaled, nullptr, 1));
        mixMulScaled[f] = scaled;
        mm = steady;                // recomputed by UpdateMixingLevels at the end of every tick (:281)
    }
    return DCS_OK;
}
