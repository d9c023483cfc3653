/* oracle/dcs_oracle.c -- TEST INFRASTRUCTURE ONLY (see dcs_oracle.h).
 *
 * CPU restatement, in plain C, of the reference's frame-decode path.  Every
 * function names the reference lines it follows (relative to
 * /root/reference/DCSDecoder/).  Written from the behaviour, not transcribed:
 * all arithmetic is done in wrapping uint32 (the reference keeps a 64-bit MR
 * but only ever reads bits 0..31 of it), the stream is a bounds-checked byte
 * array (bytes past the end read as 0), and malformed input has defined
 * results (ORC_ERR_FATAL) where the reference has undefined behaviour.
 *
 * Parity status: PINNED against oracle/_ref (the compiled reference) and the
 * committed golden vectors -- see dcs_oracle.h.
 */
#include "dcs_oracle.h"
#include "dcs_oracle_tables.h"
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

/* ------------------------------------------------------------------------
 * L0: ADSP-2105 arithmetic (DCSDecoderNative.h:822-906, .cpp:3447-3580)
 */
static inline int32_t s16(uint32_t v) { return (int32_t)(int16_t)(uint16_t)v; }

/* SaturateInt16 (DCSDecoderNative.h:826) */
static inline uint16_t sat16(int32_t v)
{
    return (uint16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
}

/* (a*b)<<1 as the low 32 bits of MR; a,b signed 1.15 (.cpp:3556-3567) */
static inline uint32_t prod_ss(uint32_t a, uint32_t b)
{
    return (uint32_t)(s16(a) * s16(b)) << 1;
}

/* signed x unsigned (.cpp:3569-3580) */
static inline uint32_t prod_su(uint32_t a, uint32_t b)
{
    return (uint32_t)(s16(a) * (int32_t)(b & 0xFFFF)) << 1;
}

static inline uint16_t mr1(uint32_t mr) { return (uint16_t)(mr >> 16); }

/* RoundMultiplyResult (.cpp:3503-3514): add 0x8000; if the LAST PRODUCT's low
 * word is exactly 0x8000, force bit 16 of the result to zero */
static inline uint32_t round_mr(uint32_t mr, uint32_t lastProd)
{
    mr += 0x8000u;
    if ((lastProd & 0xFFFFu) == 0x8000u)
        mr &= ~0x10000u;
    return mr;
}

/* MultiplyAndRound (.cpp:3526-3538) */
static inline uint32_t mul_round(uint32_t a, uint32_t b)
{
    uint32_t p = prod_ss(a, b);
    return round_mr(p, p);
}

/* MultiplyRoundAdd / MultiplyRoundSub (.cpp:3540-3554) */
static inline uint16_t mra(uint32_t mr, uint32_t a, uint32_t b)
{
    uint32_t p = prod_ss(a, b);
    return mr1(round_mr(mr + p, p));
}
static inline uint16_t mrs(uint32_t mr, uint32_t a, uint32_t b)
{
    uint32_t p = prod_ss(a, b);
    return mr1(round_mr(mr - p, p));
}

/* CalcExp32 (.cpp:3447-3459): number of redundant sign bits, negated */
static int calc_exp32(uint32_t x)
{
    int res = 0;
    if (x & 0x80000000u)
    {
        while (x & 0x40000000u) { --res; x <<= 1; }
    }
    else
    {
        while (res > -31 && !(x & 0x40000000u)) { --res; x <<= 1; }
    }
    return res;
}

/* BitShiftSigned32 (.cpp:3486-3501): by>0 left, by<0 arithmetic right */
static uint32_t shift_signed32(uint32_t v, int by)
{
    if (by >= 0)
        return v << by;
    by = -by;
    if ((int32_t)v >= 0)
        return v >> by;
    return by < 32 ? ((v >> by) | (0xFFFFFFFFu << (32 - by))) : 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------------
 * Stream container + MSB-first bit reader
 * (InitChannelStream .cpp:1433-1463, InitStreamPlayback :1595-1641,
 *  ROMBitPointer DCSDecoderNative.h:229-289)
 */
typedef struct
{
    const uint8_t *data;
    size_t len;
    int os;
    int nFrames;
    int hdrLen;
    size_t payOff;          /* first payload byte = 2 + hdrLen */

    size_t p;               /* ROMBitPointer.p : next byte to fetch */
    uint32_t buf;           /* ROMBitPointer.buf */
    int nBits;              /* ROMBitPointer.nBits */

    uint8_t header[16];     /* AudioStream::header (copy made at stream start) */
    uint16_t bandType[16];  /* AudioStream::bandTypeBuf */
    int frameCounter;
    int loopCounter;
    int stop;               /* Channel::stop */
    int fatal;
} Stream;

static inline uint32_t byte_at(const Stream *s, size_t i)
{
    return i < s->len ? s->data[i] : 0u;
}

/* Peek: refill while nBits <= n (DCSDecoderNative.h:267-279) */
static inline uint32_t br_peek(Stream *s, int n)
{
    while (s->nBits <= n)
    {
        s->buf |= byte_at(s, s->p++) << (24 - s->nBits);
        s->nBits += 8;
    }
    return s->buf >> (32 - n);
}
static inline uint32_t br_get(Stream *s, int n)
{
    uint32_t r = br_peek(s, n);
    s->nBits -= n;
    s->buf <<= n;
    return r;
}
static inline int32_t br_get_signed(Stream *s, int n)
{
    uint32_t r = br_get(s, n);
    if (r & (1u << (n - 1)))
        r |= 0xFFFFFFFFu << n;
    return (int32_t)r;
}
static inline int32_t br_bitpos(const Stream *s)
{
    return (int32_t)((s->p - s->payOff) * 8) - s->nBits;
}

/* InitChannelStream (.cpp:1433-1463) */
static void stream_open(Stream *s, int os, const uint8_t *data, size_t len)
{
    memset(s, 0, sizeof(*s));
    s->data = data;
    s->len = len;
    s->os = os;
    s->nFrames = (int)((byte_at(s, 0) << 8) | byte_at(s, 1));
    s->frameCounter = s->nFrames;
    s->hdrLen = (os == ORC_OS93A && (byte_at(s, 2) & 0x80)) ? 1 : 16;
    s->payOff = 2 + (size_t)s->hdrLen;
    s->p = s->payOff;
}

/* InitStreamPlayback (.cpp:1595-1641) */
static void stream_start(Stream *s)
{
    int i = 0;
    for ( ; i < s->hdrLen ; ++i)
        s->header[i] = (uint8_t)byte_at(s, 2 + (size_t)i);
    for ( ; i < 16 ; ++i)
        s->header[i] = 0;
    memset(s->bandType, 0, sizeof(s->bandType));
}

/* ------------------------------------------------------------------------
 * Mix-accumulate (the 32-bit "splice" MAC, .cpp:2244-2250, :2434-2443):
 * low word = scaled sample, high word = accumulator, add (int16)scaled * (uint16)mixMul
 */
static inline void mix_add(uint16_t *fb, int idx, uint32_t scaled16, uint32_t mixMul)
{
    if (idx < 0x200)
    {
        uint32_t acc = ((uint32_t)fb[idx] << 16) | (scaled16 & 0xFFFFu);
        acc += (uint32_t)(s16(scaled16) * (int32_t)mixMul);
        fb[idx] = (uint16_t)(acc >> 16);
    }
}

/* scale code 'eeeemm' -> 1.15 factor (.cpp:1978-1979, :2337-2343) */
static inline uint32_t scale_factor(int code)
{
    return (uint32_t)orc_scale_mant[code & 3] >> (15 - ((code >> 2) & 15));
}

/* DC fix-up (.cpp:2255-2257, :2609-2611) */
static inline void dc_fixup(uint16_t *fb, uint16_t saved1)
{
    uint16_t delta = sat16(s16(fb[1]) - s16(saved1));
    fb[0] = sat16(s16(delta) + s16(fb[0]));
    fb[1] = saved1;
}

/* ------------------------------------------------------------------------
 * a2: DecoderImpl94x::DecompressFrame (.cpp:1679-2261)
 */
static void decompress94(Stream *s, uint32_t mixMul, uint16_t *fb)
{
    const uint8_t *hdr = s->header;
    const uint16_t saved1 = fb[1];
    const int type = hdr[0] >> 7;
    const int sub = ((hdr[1] & 0x80) >> 6) | ((hdr[2] & 0x80) >> 7);

    /* pre-adjust from the PREVIOUS frame's band types (:1744-1773).  Only Type 1
     * consumes it; there the codes are table indices 0..15. */
    int preAdj[3] = { 0, 0, 0 };
    if (type == 1)
    {
        const uint8_t *map = (sub == 0) ? orc94_preadj_sub0 : orc94_preadj_sub3;
        for (int i = 0 ; i < 3 ; ++i)
        {
            if (s->bandType[i] > 15) { s->fatal = s->stop = 1; goto done; }
            preAdj[i] = map[s->bandType[i]];
        }
    }

    /* frame header: one delta per populated band, bit-serial tree walk (:1780-1834) */
    for (int i = 0 ; i < 16 && (hdr[i] & 0x7F) != 0x7F ; ++i)
    {
        int node = 0;
        do
        {
            node += br_get(s, 1) ? orc94_hdr_tree[node] : 1;
        }
        while (!(orc94_hdr_tree[node] & 0x8000));
        s->bandType[i] = (uint16_t)(s->bandType[i] + (orc94_hdr_tree[node] & 0xFF) - 0x2E);
    }

    int outIdx = 1;
    int valid = 1;
    for (int band = 0 ; band < 16 ; ++band)
    {
        int hb = hdr[band] & 0x7F;
        if (hb == 0x7F)
            break;

        int count = orc94_band_count[band];
        int inc = 1;
        if (hb & 0x40) { inc = 2; count /= 2; }

        int code = s->bandType[band];
        if (code == 0)
        {
            outIdx += count;            /* note: the HALVED count (:1886) */
            continue;
        }

        int scaleCode = hb;
        if (type == 1)
        {
            if (code > 15) { s->fatal = s->stop = 1; break; }
            const uint8_t *x = band < 3 ? orc94_xlat_b02 : band < 6 ? orc94_xlat_b35 : orc94_xlat_b6f;
            if (band < 3)
                hb += preAdj[band];
            scaleCode = hb + x[2*code + 1];
            code = x[2*code];
        }
        uint32_t scale = scale_factor(scaleCode);

        uint16_t cur[32];
        memset(cur, 0, sizeof(cur));
        if (code == 0)
        {
            valid = 0; s->stop = 1;                     /* :1985-1991 */
        }
        else if (code <= 6)
        {
            static const uint16_t *const books[6] = {
                orc94_cb1, orc94_cb2, orc94_cb3, orc94_cb4, orc94_cb5, orc94_cb6 };
            const uint16_t *book = books[code - 1];
            const int maxBits = orc94_cb_maxbits[code];
            const int ref = 1 << (code - 1);
            int n = 0;
            for (int i = count ; i != 0 ; --i)
            {
                uint32_t e = book[br_peek(s, maxBits)];
                br_get(s, (int)(e >> 8));
                int val = (int)(e & 0xFF);
                if (val & 0x80)
                {
                    if (i >= 2) { cur[n++] = 0; cur[n++] = 0; --i; }
                    else { valid = 0; s->stop = 1; i = 1; }     /* :2213-2218 */
                }
                else
                    cur[n++] = (uint16_t)(val - ref);
            }
        }
        else
        {
            if (code > 16) { s->fatal = s->stop = 1; break; }
            for (int i = 0 ; i < count ; ++i)
                cur[i] = (uint16_t)br_get_signed(s, code);
        }

        if (!valid)
            memset(cur, 0, sizeof(cur));                /* :2238-2239 */

        for (int i = 0 ; i < count ; ++i, outIdx += inc)
            mix_add(fb, outIdx, (uint32_t)(s16(cur[i]) * (int32_t)scale), mixMul);
    }

done:
    dc_fixup(fb, saved1);
}

/* ------------------------------------------------------------------------
 * ReadHuff93 (.cpp:2618-2684)
 */
static int read_huff93(Stream *s, int *subType)
{
    uint32_t ele = orc93_type_tree[0];
    do
    {
        uint32_t idx = br_get(s, 1) ? (ele >> 8) : (ele & 0xFF);
        ele = orc93_type_tree[idx];
    }
    while (!(ele & 0x8000));

    int val = (int)(ele & 0x3F);
    if (val < 0x1E)
        return val - 0x0F;
    *subType = (*subType != 0) ? 0 : 1;
    return val - 0x2E;
}

/* ------------------------------------------------------------------------
 * a3: DecoderImpl93::DecompressFrame (.cpp:2293-2615)
 */
static void decompress93(Stream *s, uint32_t mixMul, uint16_t *fb)
{
    const uint16_t saved1 = fb[1];
    const int type = (int)(byte_at(s, 2) >> 7);      /* header re-read from the stream (:2298, :2308) */
    int subType = (type == 1) ? 0 : 2;
    int first = 1;
    uint16_t prv = 0, prvDelta = 0;
    int reuse = 0;
    int code = 0;
    int outIdx = 1;

    for (int band = 0 ; band < 16 ; ++band)
    {
        int hb = (int)(byte_at(s, 2 + (size_t)band) & 0x7F);
        if (hb == 0x7F)
            break;

        uint32_t scale = scale_factor(hb);
        int strideCode = hb >> 6;

        int nSamples, inc, fixup, stride;
        if (type == 0)
        {
            nSamples = 16;
            if (strideCode == 0) { inc = 1; fixup = 0; stride = 16; }
            else { ++outIdx; inc = 2; fixup = -1; stride = 31; }
        }
        else
        {
            fixup = 0;
            if (strideCode == 0) { inc = 1; nSamples = stride = first ? 15 : 16; }
            else { inc = 2; nSamples = stride = 8; }
        }

        if (reuse)
            reuse = br_get(s, 1) != 0;
        if (!reuse)
        {
            if (type == 0)
            {
                if (br_get(s, 1))
                {
                    /* one bit picks +1 / -1 mod 3 (:2410-2413) */
                    subType = br_get(s, 1) ? (subType + 1) % 3 : (subType + 2) % 3;
                }
                code = (int)br_get(s, 4);
            }
            else
            {
                s->bandType[band] = (uint16_t)(s->bandType[band] + read_huff93(s, &subType));
                code = s->bandType[band];
            }
        }

        if (code == 0)
        {
            reuse = 1;
            if (subType == 0)
            {
                outIdx += stride;
                prv = 0; prvDelta = 0;
            }
            else if (subType == 1)
            {
                /* repeat previous input; the low product word is carried from
                 * iteration to iteration, not reloaded (:2513-2534) */
                uint32_t low = (uint32_t)(s16(prv) * (int32_t)scale) & 0xFFFFu;
                int32_t mulLow = s16(low);
                for (int i = 0 ; i < nSamples ; ++i, outIdx += inc)
                {
                    if (outIdx < 0x200)
                    {
                        uint32_t acc = ((uint32_t)fb[outIdx] << 16) | low;
                        acc += (uint32_t)(mulLow * (int32_t)mixMul);
                        fb[outIdx] = (uint16_t)(acc >> 16);
                        low = acc & 0xFFFFu;
                    }
                }
                prvDelta = 0;
                outIdx += fixup;
            }
            else
            {
                for (int i = 0 ; i < nSamples ; ++i, outIdx += inc)
                {
                    prv = (uint16_t)(prv + prvDelta);
                    mix_add(fb, outIdx, (uint32_t)(s16(prv) * (int32_t)scale), mixMul);
                }
                outIdx += fixup;
            }
        }
        else
        {
            int width = code + (type == 0 ? 1 : 0);
            if (width > 16) { s->fatal = s->stop = 1; break; }

            uint16_t in[16];
            for (int i = 0 ; i < nSamples ; ++i)
                in[i] = (uint16_t)br_get_signed(s, width);

            for (int i = 0 ; i < nSamples ; ++i, outIdx += inc)
            {
                if (subType == 0)
                {
                    mix_add(fb, outIdx, (uint32_t)(s16(in[i]) * (int32_t)scale), mixMul);
                }
                else
                {
                    if (subType == 1) prvDelta = in[i];
                    else prvDelta = (uint16_t)(prvDelta + in[i]);
                    prv = (uint16_t)(prv + prvDelta);
                    mix_add(fb, outIdx, (uint32_t)(s16(prv) * (int32_t)scale), mixMul);
                }
            }
            if (subType == 0)
            {
                prv = in[nSamples - 1];
                prvDelta = (uint16_t)(prv - in[nSamples - 2]);
            }
            outIdx += fixup;
        }
        first = 0;
    }

    dc_fixup(fb, saved1);
}

/* ------------------------------------------------------------------------
 * a4: DecoderImpl93a::DecompressFrame, Type 1 streams (.cpp:2831-3032)
 */
static void decompress93a(Stream *s, uint32_t mixMul, uint16_t *fb)
{
    const uint32_t hb = byte_at(s, 2);
    if (!(hb & 0x80))
    {
        decompress93(s, mixMul, fb);
        return;
    }

    int prvScale = 0x1A;
    const uint16_t *bbBook = &orc93a_bandbits_cb[(hb & 0x60)];    /* (sel>>1) entries x 2 words */
    const int numBands = (int)(hb & 0x1F);
    int outIdx = 0;

    for (int band = 0 ; band < numBands ; ++band)
    {
        if (band >= 18) { s->fatal = s->stop = 1; break; }
        int numInputs = orc93a_inputs_per_band[band];

        uint32_t pk = br_peek(s, 4);
        int bandBits = bbBook[2*pk];
        br_get(s, bbBook[2*pk + 1]);
        if (bandBits == 0xFFFF)
            break;

        if (bandBits == 0)
        {
            outIdx += numInputs * 2;
            continue;
        }

        /* two-level scale codebook: (value, nBits, subTable) (:2932-2970) */
        const uint16_t *e = &orc93a_scale_cb[3 * br_peek(s, 4)];
        br_get(s, e[1]);
        if (e[0] == 0xFFFF)
        {
            e = &orc93a_scale_cb[3 * (e[2] + br_peek(s, 4))];
            br_get(s, e[1] - 4);
        }

        int scaleCode = prvScale + (int)e[0] - 1 + bandBits * 2;
        if (scaleCode > 0x39)
            scaleCode -= 0x36;
        prvScale = scaleCode - bandBits * 2;

        uint32_t sf = 0x8000;
        for (int i = 0 ; i < (scaleCode & 3) ; ++i)
            sf = (sf * 0x9838u) >> 15;
        sf <<= (scaleCode >> 2);
        sf = ((sf >> 16) * mixMul) >> 15;                   /* then truncated to 16 bits (:2995, :3011) */

        const uint16_t *pairBase = &orc93a_pair[2 << bandBits];
        for (int i = 0 ; i < numInputs ; ++i)
        {
            const uint16_t *pair = pairBase + 2 * br_get(s, bandBits);
            for (int k = 0 ; k < 2 ; ++k, ++outIdx)
                fb[outIdx] = mra((uint32_t)fb[outIdx] << 16, pair[k], sf & 0xFFFFu);
        }
    }
}

static void decompress_frame(Stream *s, uint32_t mixMul, uint16_t *fb)
{
    switch (s->os)
    {
    case ORC_OS93A: decompress93a(s, mixMul, fb); break;
    case ORC_OS93B: decompress93(s, mixMul, fb); break;
    default:        decompress94(s, mixMul, fb); break;     /* Initialize() :3147-3160 */
    }
}

/* ------------------------------------------------------------------------
 * bit reversal over 9 bits (table at .cpp:320-353; computed here)
 */
static inline int bitrev9(int v)
{
    int r = 0;
    for (int i = 0 ; i < 9 ; ++i)
        r |= ((v >> i) & 1) << (8 - i);
    return r;
}

/* complex rotate used by both transforms (.cpp:500-506, :761-765):
 * re = a.re*c - a.im*s ; im = a.im*c + a.re*s, first term truncating, second rounded */
static inline void rotate(uint32_t are, uint32_t aim, uint32_t c, uint32_t sn, int32_t *tre, int32_t *tim)
{
    *tre = s16(mrs(prod_ss(are, c), aim, sn));
    *tim = s16(mra(prod_ss(aim, c), are, sn));
}

/* overlap-add of one sample (.cpp:545-554, :797-801): both terms signed x unsigned */
static inline uint16_t overlap_mix(uint32_t x, uint32_t cx, uint32_t o, uint32_t co)
{
    return mr1(prod_su(x, cx) + prod_su(o, co) + 0x8000u);
}

/* ------------------------------------------------------------------------
 * a5: DecoderImpl94x::TransformFrame (.cpp:397-576)
 */
static void transform94(uint16_t *f, int volShift, uint16_t *ovl, int16_t *pcm)
{
    /* pre-pass 1 (:403-418): MulSS(x, 0x8000) is a wrapping negate */
    f[0x80] = mr1(prod_ss(f[0x80], 0x8000));
    f[0x81] = mr1(prod_ss((uint32_t)(-s16(f[0x81])), 0x8000));
    for (int i = 0 ; i < 64 ; ++i)
    {
        uint16_t *a = f + 2*i, *b = f + 0x100 - 2*i;
        int32_t x0 = s16(a[0]), x1 = s16(a[1]), y0 = s16(b[0]), y1 = s16(b[1]);
        a[0] = mr1(prod_ss(sat16(x0 + y0), 0x8000));
        b[0] = mr1(prod_ss(sat16(x0 - y0), 0x8000));
        a[1] = mr1(prod_ss(sat16(x1 - y1), 0x8000));
        b[1] = mr1(prod_ss(sat16(x1 + y1), 0x8000));
    }

    /* pre-pass 2 (:420-456) */
    for (int i = 0 ; i < 64 ; ++i)
    {
        uint16_t *a = f + 2*i, *b = f + 0x100 - 2*i;
        uint32_t c0 = orc_fft_coef[bitrev9(2 + 4*i)];
        uint32_t c1 = orc_fft_coef[bitrev9(4*i)];
        int32_t x0 = s16(a[0]), x1 = s16(a[1]);
        uint32_t n0 = b[0], n1 = b[1];
        int32_t p0 = s16(mrs(prod_ss(n1, c1), n0, c0));
        int32_t p1 = s16(mra(prod_ss(n1, c0), n0, c1));
        a[0] = sat16(p1 + x0);
        a[1] = sat16(p0 + x1);
        b[0] = sat16(x0 - p1);
        b[1] = sat16(p0 - x1);
    }

    /* pre-pass 3 (:458-471) */
    for (int i = 0 ; i < 128 ; ++i)
    {
        int32_t x = s16(f[i]), y = s16(f[0x80 + i]);
        f[i] = sat16(x + y);
        f[0x80 + i] = sat16(x - y);
    }

    /* six saturating radix-2 stages (:480-524) */
    for (int stage = 0, nPart = 2, size = 0x40 ; stage < 6 ; ++stage, nPart *= 2, size /= 2)
    {
        for (int part = 0 ; part < nPart ; ++part)
        {
            uint32_t sn = orc_fft_coef[part], cs = orc_fft_coef[0x80 + part];
            uint16_t *u = f + 2*part*size, *a = u + size;
            for (int j = 0 ; j < size ; j += 2)
            {
                int32_t tre, tim;
                rotate(a[j], a[j+1], cs, sn, &tre, &tim);
                int32_t ure = s16(u[j]), uim = s16(u[j+1]);
                u[j]   = sat16(ure - tre);
                u[j+1] = sat16(uim - tim);
                a[j]   = sat16(ure + tre);
                a[j+1] = sat16(uim + tim);
            }
        }
    }

    /* volume shift (:532-534) */
    for (int i = 0 ; i < 0x100 ; ++i)
        f[i] = (uint16_t)(s16(f[i]) >> volShift);

    /* overlap the first 16 samples, in place at bit-reversed positions (:538-555) */
    for (int i = 0 ; i < 16 ; ++i)
    {
        int bi = bitrev9(i & ~1) + (i & 1);
        f[bi] = overlap_mix(f[bi], orc_overlap_coef[i], ovl[i], orc_overlap_coef[15 - i]);
    }

    /* gather 240 outputs + 16 carried (:559-575) */
    for (int i = 0 ; i < 256 ; ++i)
    {
        uint16_t v = f[bitrev9(i & ~1) + (i & 1)];
        if (i < 240) pcm[i] = (int16_t)v;
        else ovl[i - 240] = v;
    }
}

/* ------------------------------------------------------------------------
 * a6: DecoderImpl93::TransformFrame (.cpp:614-813)
 */
static void transform93(uint16_t *f, int volShift, uint16_t *ovl, int16_t *pcm)
{
    /* DC magnitude sqrt(f0^2 + f1^2) by polynomial (:635-710) */
    uint32_t ar = f[0];
    int neg = s16(ar) < 0;
    if (neg)
        ar = (uint16_t)(-s16(ar));
    uint32_t sr = prod_ss(f[1], f[1]) + prod_ss(ar, ar);
    int exponent = calc_exp32(sr);          /* Normalize32 (:3470-3483): exponent in [-31, 0] */
    if (exponent < 0)
        sr <<= -exponent;
    ar = mr1(sr);
    if (ar != 0)
    {
        uint32_t mr = 0x0D490000u;
        mr += (uint32_t)(0x5D1D * s16(ar)) << 1;
        uint32_t mf = mr1(mul_round(ar, ar));
        mr += (uint32_t)(-22035 * s16(mf)) << 1;
        mf = mr1(mul_round(ar, mf));
        mr += (uint32_t)(0x46D6 * s16(mf)) << 1;
        mf = mr1(mul_round(ar, mf));
        mr += (uint32_t)(-8790 * s16(mf)) << 1;
        mf = mr1(mul_round(ar, mf));
        mr += (uint32_t)(0x072D * s16(mf)) << 1;
        if (exponent & 1)
        {
            mr = mul_round(mr1(mr), 0x5A82);
            exponent += 1;
        }
        exponent = exponent / 2 + 1;        /* C truncation toward zero */
        ar = mr1(shift_signed32(mr, exponent));
        if (neg)
            ar = (uint16_t)(-s16(ar));
    }
    f[0] = f[0x100] = (uint16_t)ar;
    f[1] = f[0x101] = 0;

    /* expand 256 -> 512 words, wrapping adds (:714-732) */
    for (int i = 0 ; i < 64 ; ++i)
    {
        uint16_t *i0 = f + 2 + 2*i, *i1 = f + 0xFE - 2*i, *i2 = f + 0x102 + 2*i, *i3 = f + 0x1FE - 2*i;
        int32_t xr = s16(i0[0]), xi = s16(i0[1]), yr = s16(i1[0]), yi = s16(i1[1]);
        i0[0] = i1[0] = (uint16_t)(xr + yr);
        i2[0] = (uint16_t)(xr - yr);
        i3[0] = (uint16_t)(yr - xr);
        i2[1] = i3[1] = (uint16_t)(xi + yi);
        i0[1] = (uint16_t)(xi - yi);
        i1[1] = (uint16_t)(yi - xi);
    }

    /* seven wrapping radix-2 stages over 256 complex points (:742-778) */
    for (int stage = 0, nPart = 2, size = 0x80 ; stage < 7 ; ++stage, nPart *= 2, size /= 2)
    {
        for (int part = 0 ; part < nPart ; ++part)
        {
            uint32_t sn = orc_fft_coef[part], cs = orc_fft_coef[0x80 + part];
            uint16_t *u = f + 2*part*size, *a = u + size;
            for (int j = 0 ; j < size ; j += 2)
            {
                int32_t tre, tim;
                rotate(a[j], a[j+1], cs, sn, &tre, &tim);
                int32_t ure = s16(u[j]), uim = s16(u[j+1]);
                u[j]   = (uint16_t)(ure - tre);
                u[j+1] = (uint16_t)(uim - tim);
                a[j]   = (uint16_t)(tre + ure);
                a[j+1] = (uint16_t)(tim + uim);
            }
        }
    }

    /* gather in time order + volume shift (:782-785), overlap (:789-802), emit, carry (:805-812) */
    uint16_t t[256];
    for (int i = 0 ; i < 256 ; ++i)
        t[i] = (uint16_t)(s16(f[bitrev9(i)]) >> volShift);
    for (int i = 0 ; i < 256 ; ++i)
        f[2*i + 1] = t[i];          /* the reference leaves them at the odd words */
    for (int i = 0 ; i < 16 ; ++i)
        pcm[i] = (int16_t)overlap_mix(ovl[i], orc_overlap_coef[15 - i], t[i], orc_overlap_coef[i]);
    for (int i = 16 ; i < 240 ; ++i)
        pcm[i] = (int16_t)t[i];
    for (int i = 0 ; i < 16 ; ++i)
        ovl[i] = t[240 + i];
}

static void transform_frame(int os, uint16_t *f, int volShift, uint16_t *ovl, int16_t *pcm)
{
    if (os == ORC_OS93A || os == ORC_OS93B)
        transform93(f, volShift, ovl, pcm);
    else
        transform94(f, volShift, ovl, pcm);
}

/* ------------------------------------------------------------------------
 * a8: volume / mixing parameters
 */

/* SetMasterVolume (.cpp:3250-3282) */
uint16_t orc_volume_multiplier(int vol)
{
    if (vol == 0)
        return 0;
    uint32_t s = (uint16_t)vol;         /* the loop uses the UNCLAMPED value, 8 bits of it */
    uint32_t x = 0x3FFF, y = 0x7D98;
    for (int i = 0 ; i < 8 ; ++i)
    {
        if (!(s & 1))
            x = ((x * y) >> 15) & 0xFFFF;
        y = ((y * y) >> 15) & 0xFFFF;
        s >>= 1;
    }
    return (uint16_t)(x << 1);
}

/* the multiplier half of UpdateMixingLevels (.cpp:3072-3121) */
uint16_t orc_mixing_multiplier(int os, int levelSum, int channelVolume)
{
    if (levelSum > 8191) levelSum = 8191;
    else if (levelSum < -8191) levelSum = -8191;
    uint32_t mixerExp = (uint16_t)(((levelSum >> 6) & 0x3FF) + 0x80);
    uint32_t mult = (os == ORC_OS93A) ? 0x7FFFu : (uint16_t)(channelVolume << 7);
    uint32_t prod = 0x7C94;
    for (int j = 0, bit = 1 ; j < 8 ; ++j, bit <<= 1)
    {
        if (!(mixerExp & (uint32_t)bit))
            mult = (uint16_t)((mult * prod) >> 15);
        prod = (uint16_t)((prod * prod) >> 15);
    }
    return (uint16_t)(mult << 1);
}

/* ------------------------------------------------------------------------
 * Player: MainLoop (.cpp:89-306) + DecodeStream (:1546-1589) for streams
 * loaded with LoadAudioStream (:1387-1431); no track programs.
 */
typedef struct
{
    Stream st;
    int active;             /* !playbackBitPtr.IsNull() */
    uint16_t mixMul;        /* Channel::mixingMultiplier, 0x7FFF at construction (DCSDecoderNative.h:514) */
    int level;              /* mixer[ch].curLevel */
    int channelVolume;
} Chan;

typedef struct
{
    int os;
    uint16_t volMult;
    Chan ch[8];
    uint16_t fb[0x200];
    uint16_t ovl[16];
} Player;

static void player_init(Player *pl, int os, int volume)
{
    memset(pl, 0, sizeof(*pl));
    pl->os = os;
    pl->volMult = orc_volume_multiplier(volume);
    for (int i = 0 ; i < 8 ; ++i)
    {
        pl->ch[i].mixMul = 0x7FFF;
        pl->ch[i].channelVolume = 0xFF;
    }
}

static int player_load(Player *pl, int c, const uint8_t *data, size_t len, int level)
{
    Chan *ch = &pl->ch[c];
    stream_open(&ch->st, pl->os, data, len);
    if (ch->st.nFrames == 0)
        return -3;          /* the reference would play 65536 garbage frames; unsupported here */
    ch->st.loopCounter = 1;
    ch->active = 1;
    ch->level = level << 6;
    return 0;
}

/* returns volShift used */
static int player_tick(Player *pl, int16_t *pcm, uint16_t *mixMulUsed)
{
    memset(pl->fb, 0, sizeof(pl->fb));

    /* forced-stop sweep (:95-116) */
    for (int c = 0 ; c < 8 ; ++c)
    {
        Chan *ch = &pl->ch[c];
        if (ch->st.stop)
        {
            ch->st.stop = 0;
            if (ch->active)
            {
                ch->active = 0;
                ch->level = 0;          /* ResetMixingLevels(ch) */
            }
        }
    }

    /* shared fixed-point scale (:227-269) */
    uint64_t sum = 0;
    for (int c = 0 ; c < 8 ; ++c)
        if (pl->ch[c].active)
            sum += (uint64_t)pl->ch[c].mixMul * pl->volMult;
    sum >>= 2;
    int volShift = -(calc_exp32((uint32_t)sum) + 3);
    volShift = volShift < 0 ? 0 : volShift > 8 ? 8 : volShift;
    for (int c = 0 ; c < 8 ; ++c)
    {
        uint64_t m = ((uint64_t)pl->ch[c].mixMul * pl->volMult) << 1;
        pl->ch[c].mixMul = (uint16_t)((m << volShift) >> 16);
    }
    if (mixMulUsed)
        *mixMulUsed = pl->ch[0].mixMul;

    /* DecodeStream per channel (:1546-1589) */
    for (int c = 0 ; c < 8 ; ++c)
    {
        Chan *ch = &pl->ch[c];
        Stream *s = &ch->st;
        if (!ch->active)
            continue;
        if (s->p == s->payOff)
            stream_start(s);
        decompress_frame(s, ch->mixMul, pl->fb);
        s->frameCounter = (s->frameCounter - 1) & 0xFFFF;
        if (s->frameCounter != 0)
            continue;
        s->frameCounter = s->nFrames;
        s->p = s->payOff; s->buf = 0; s->nBits = 0;
        if (s->loopCounter == 0)
            continue;
        if (--s->loopCounter != 0)
            continue;
        ch->active = 0;
    }

    transform_frame(pl->os, pl->fb, volShift, pl->ovl, pcm);

    /* UpdateMixingLevels (:3042-3121): no fades here, levels are constant */
    for (int c = 0 ; c < 8 ; ++c)
        pl->ch[c].mixMul = orc_mixing_multiplier(pl->os, pl->ch[c].level, pl->ch[c].channelVolume);

    return volShift;
}

int orc_decode(int os, int volume, int nch,
    const uint8_t *const *streams, const size_t *lens, const int *levels,
    int nFramesOut, int16_t *pcm, OrcProbe *probes)
{
    if (nch < 1 || nch > 8)
        return -1;
    Player *plp = (Player *)malloc(sizeof(Player));    /* heap: re-entrant, callable from several threads */
    if (!plp)
        return -4;
#define pl (*plp)
    player_init(&pl, os, volume);
    for (int c = 0 ; c < nch ; ++c)
    {
        int r = player_load(&pl, c, streams[c], lens[c], levels[c]);
        if (r != 0)
        {
            free(plp);
            return r;
        }
    }
    for (int f = 0 ; f < nFramesOut ; ++f)
    {
        if (probes)
        {
            Chan *ch = &pl.ch[0];
            OrcProbe *pr = &probes[f];
            pr->active = ch->active;
            pr->bitOff = ch->active ? br_bitpos(&ch->st) : -1;
            pr->mixMul = ch->mixMul;
            pr->volMult = pl.volMult;
            int atStart = ch->active && ch->st.p == ch->st.payOff;
            for (int i = 0 ; i < 16 ; ++i)
                pr->bandType[i] = atStart ? 0 : ch->st.bandType[i];
        }
        player_tick(&pl, pcm + (size_t)f * 240, NULL);
    }
#undef pl
    free(plp);
    return 0;
}

/* The stream loop of DCSExplorer --extract-streams (DCSExplorer.cpp:1628-1907) on one decoder: same contract as
 * ref_decode_sequence in ref_driver.cpp.  ClearTracks (:1466-1473) clears every channel's audio stream. */
int orc_decode_sequence(int os, int volume, int n, const uint8_t *const *streams, const size_t *lens,
    const int *levels, int extraFrames, int16_t *pcm)
{
    Player *pl = (Player *)malloc(sizeof(Player));
    if (!pl)
        return -4;
    player_init(pl, os, volume);
    for (int i = 0 ; i < n ; ++i)
    {
        int r = player_load(pl, 0, streams[i], lens[i], levels[i]);
        if (r != 0)
        {
            free(pl);
            return r;
        }
        const int nFrames = (int)pl->ch[0].st.nFrames + extraFrames;
        for (int frame = 0 ; frame < nFrames ; ++frame)
        {
            player_tick(pl, pcm, NULL);
            pcm += 240;
            if (frame + 2 >= nFrames)
                for (int c = 0 ; c < 8 ; ++c)
                    pl->ch[c].active = 0;
        }
    }
    free(pl);
    return 0;
}

int orc_frame_params(int os, int volume, int level, int nFrames,
    uint16_t *mixMulScaled, uint8_t *volShiftOut)
{
    /* a one-frame dummy stream is enough: the parameters depend only on
     * (os, volume, level) and on whether the channel is active */
    uint16_t volMult = orc_volume_multiplier(volume);
    uint16_t mixMul = 0x7FFF;
    for (int f = 0 ; f < nFrames ; ++f)
    {
        uint64_t sum = ((uint64_t)mixMul * volMult) >> 2;
        int vs = -(calc_exp32((uint32_t)sum) + 3);
        vs = vs < 0 ? 0 : vs > 8 ? 8 : vs;
        uint64_t m = ((uint64_t)mixMul * volMult) << 1;
        mixMulScaled[f] = (uint16_t)((m << vs) >> 16);
        volShiftOut[f] = (uint8_t)vs;
        mixMul = orc_mixing_multiplier(os, level << 6, 0xFF);
    }
    return 0;
}

int orc_stream_info(int os, const uint8_t *stream, size_t len,
    int *nFrames, int *nBytes, int *formatType, int *formatSubType, uint8_t *header16)
{
    Stream s;
    stream_open(&s, os, stream, len);
    stream_start(&s);
    for (int i = 0 ; i < s.nFrames ; ++i)
    {
        uint16_t fb[0x200];
        memset(fb, 0, sizeof(fb));
        decompress_frame(&s, 0x7FFF, fb);
    }
    *nFrames = s.nFrames;
    *nBytes = (int)s.p;                             /* includes the reader's look-ahead (:1509) */
    *formatType = s.header[0] >> 7;
    *formatSubType = 0;
    if (os == ORC_OS94 || os == ORC_OS95)           /* sic: both terms test header[1] (:1517) */
        *formatSubType = ((s.header[1] & 0x80) >> 6) | ((s.header[1] & 0x80) >> 7);
    memset(header16, 0, 16);
    memcpy(header16, s.header, (os == ORC_OS93A && *formatType == 1) ? 1 : 16);
    return 0;
}

int orc_transform(int os, uint16_t *frameBuf512, int volShift, uint16_t *overlap16, int16_t *pcm240)
{
    transform_frame(os, frameBuf512, volShift, overlap16, pcm240);
    return 0;
}

int orc_decompress(int os, const uint8_t *stream, size_t len, uint16_t mixMul,
    int nFrames, uint16_t *out, int32_t *bitOffs, uint16_t *bandTypes, int32_t *stopFlags)
{
    Stream s;
    stream_open(&s, os, stream, len);
    stream_start(&s);
    for (int f = 0 ; f < nFrames ; ++f)
    {
        uint16_t *fb = out + (size_t)f * 0x200;
        bitOffs[f] = br_bitpos(&s);
        s.stop = 0;
        memset(fb, 0, 0x200 * sizeof(uint16_t));
        decompress_frame(&s, mixMul, fb);
        memcpy(bandTypes + (size_t)f * 16, s.bandType, sizeof(s.bandType));
        stopFlags[f] = s.stop | (s.fatal << 1);
    }
    return 0;
}

/* CPU-baseline helper, same contract as ref_decode_many in ref_driver.cpp */
typedef struct
{
    const int *os, *volume, *level;
    const uint8_t *const *streams;
    const size_t *lens;
    int n, repeat, nThreads, tid;
    long long frames;
} ManyArg;

static void *many_worker(void *p)
{
    ManyArg *a = (ManyArg *)p;
    int16_t *pcm = NULL;
    size_t cap = 0;
    for (int rep = 0 ; rep < a->repeat ; ++rep)
        for (int i = a->tid ; i < a->n ; i += a->nThreads)
        {
            const int nf = (a->streams[i][0] << 8) | a->streams[i][1];
            if ((size_t)nf * 240 > cap)
            {
                cap = (size_t)nf * 240;
                pcm = (int16_t *)realloc(pcm, cap * sizeof(int16_t));
            }
            const uint8_t *sp = a->streams[i];
            orc_decode(a->os[i], a->volume[i], 1, &sp, &a->lens[i], &a->level[i], nf, pcm, NULL);
            a->frames += nf;
        }
    free(pcm);
    return NULL;
}

long long orc_decode_many(const int *os, const int *volume, const int *level,
    const uint8_t *const *streams, const size_t *lens, int n, int repeat, int nThreads)
{
    if (nThreads < 1) nThreads = 1;
    if (nThreads > 256) nThreads = 256;
    pthread_t th[256];
    ManyArg args[256];
    for (int t = 0 ; t < nThreads ; ++t)
    {
        args[t] = (ManyArg){ os, volume, level, streams, lens, n, repeat, nThreads, t, 0 };
        pthread_create(&th[t], NULL, many_worker, &args[t]);
    }
    long long total = 0;
    for (int t = 0 ; t < nThreads ; ++t)
    {
        pthread_join(th[t], NULL);
        total += args[t].frames;
    }
    return total;
}

uint64_t orc_fnv1a64(const void *data, size_t n)
{
    const uint8_t *p = (const uint8_t *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0 ; i < n ; ++i)
        h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}
