/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the continuous (P-channel, non-burst) path of AeroL::Decode(bits, soft=true)
 * (JAERO/aerol.cpp:1124-2039) for 600 / 1200 / 10500 bps, with the pieces it is made of:
 *   PreambleDetector / PreambleDetectorPhaseInvariant   JAERO/aerol.cpp:717-809
 *   AeroLInterleaver::deinterleave_ba                    JAERO/aerol.cpp:603-625
 *   JConvolutionalCodec::Decode_Continuous               JAERO/jconvolutionalcodec.cpp:151-201 (viterbi_oracle.c)
 *   DelayLine, AeroLScrambler, AeroLcrc16                JAERO/aerol.h:453-481,394-440,283-392
 *   setSettings constants                                JAERO/aerol.cpp:990-1072
 * Everything after the CRC check of a signal unit (message-type names, ISU/ACARS reassembly, plane database) is text
 * formatting / control plane and is not restated; the data-carrier-detect bookkeeping that feeds back into the unique-word
 * gating (datacd, datacdcountdown) is.  The reference also lowers datacdcountdown from a 1 s wall-clock QTimer
 * (AeroL::updateDCD, aerol.cpp:1109-1122); a batch run has no event loop, so neither this file nor the _ref driver ticks it.
 * Members the reference never initialises (realimag, muw, lastframeinfo: aerol.h:956,975,990) start at 0, as in the _ref
 * driver's zeroed storage.  Pinned against the unmodified AeroL built into oracle/_ref (tests/test_aerol_oracle.py).
 *
 * 8400 bps C channel: AeroL::DecodeC (JAERO/aerol.cpp:2187-2502) with OQPSKPreambleDetectorAndAmbiguityCorrection
 * (:811-900), deinterleave_ba(block, 4), PuncturedCode::depunture_soft_block (:2505-2518), the 2714-bit delay line / scrambler,
 * the three sub-band signal units and the 300 voice bytes of a frame.  Pinned the same way (the _ref driver prints what DecodeC
 * hands to Voicesignal).  One thing cannot be pinned: on the codec's FIRST call JConvolutionalCodec::Decode_Continuous returns bits
 * 25..2741 of a buffer of which libcorrect wrote 2736, the rest being whatever QByteArray::resize left there
 * (jconvolutionalcodec.cpp:165-191); they end up as the three low bits of the last voice byte of the second frame handed out.
 * Here they are decoded zeros.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "aerol_oracle.h"
#include "viterbi_oracle.h"

typedef struct { char *p; size_t len, cap; } gb;
static void gb_push(gb *g, const void *src, size_t n)
{
    if (g->len + n > g->cap)
    {
        size_t nc = g->cap ? g->cap * 2 : 4096;
        while (nc < g->len + n) nc *= 2;
        g->p = (char *)realloc(g->p, nc); g->cap = nc;
    }
    memcpy(g->p + g->len, src, n); g->len += n;
}
static long gb_take(gb *g, void *dst, size_t elsz, long capels)
{
    long have = (long)(g->len / elsz), n = have < capels ? have : capels;
    memcpy(dst, g->p, (size_t)n * elsz);
    memmove(g->p, g->p + (size_t)n * elsz, g->len - (size_t)n * elsz);
    g->len -= (size_t)n * elsz;
    return n;
}

typedef struct { int preamble[64], buffer[64], len, tollerence, inverted; } pdet_t;
static void pdet_set(pdet_t *d, uint64_t bits, int len) /* setPreamble(quint64,int) :730-743 */
{
    d->len = len;
    for (int i = len - 1, k = 0; i >= 0; i--, k++) d->preamble[k] = (int)((bits >> i) & 1);
    memset(d->buffer, 0, sizeof(d->buffer));
}
static int pdet_update_exact(pdet_t *d, int val) /* PreambleDetector::Update :744-750 */
{
    for (int i = 0; i < d->len - 1; i++) d->buffer[i] = d->buffer[i + 1];
    d->buffer[d->len - 1] = val;
    if (!memcmp(d->buffer, d->preamble, sizeof(int) * (size_t)d->len)) { memset(d->buffer, 0, sizeof(d->buffer)); return 1; }
    return 0;
}
static int pdet_update_pi(pdet_t *d, int val) /* PreambleDetectorPhaseInvariant::Update :781-804 */
{
    int xorsum = 0;
    for (int i = 0; i < d->len - 1; i++) { d->buffer[i] = d->buffer[i + 1]; xorsum += d->buffer[i] ^ d->preamble[i]; }
    xorsum += val ^ d->preamble[d->len - 1];
    d->buffer[d->len - 1] = val;
    if (xorsum >= (d->len - d->tollerence)) { d->inverted = 1; return 1; }
    if (xorsum <= d->tollerence) { d->inverted = 0; return 1; }
    return 0;
}

static uint16_t crc16_bytes(const unsigned char *bytes, int n) /* AeroLcrc16::calcusingbytes aerol.h:333-360 */
{
    uint16_t crc = 0xFFFF;
    for (int i = 0; i < n; i++)
    {
        int message_byte = (signed char)bytes[i]; /* message_byte=bytes[i] with char bytes: sign-extends, only the low 8 bits are used */
        for (int k = 0; k < 8; k++)
        {
            int message_bit = message_byte & 1;
            message_byte >>= 1;
            int crc_bit = crc & 1;
            crc >>= 1;
            if (crc_bit ^ message_bit) crc = crc ^ 0x8408;
        }
    }
    return (uint16_t)~crc;
}

struct jo_aerol
{
    int ifb, useingOQPSK, N, blocksz, dl2_len;
    int NumberOfBits, BitsInHeader, TotalNumberOfBits;
    int cntr, datacd, datacdcountdown, gotsync_last, realimag, blockcnt, muw;
    uint16_t frameinfo, lastframeinfo;
    int formatid, supfrmaker, framecounter1, framecounter2;
    pdet_t pd_exact, pd_imag, pd_real;
    int *block;
    jo_codec *codec;
    int *dl2; int dl2_ptr, dl2_sz;
    unsigned char scr[5000]; int scr_pos;
    unsigned char infofield[4096]; int ninfo;
    int depermute[64];
    long nbits_total, nframes;
    gb sus, events;
    /* burst mode (R/T channel packets): RTChannelDeleaveFECScram aerol.h:554-873 */
    int burstmode;
    int *rt_block; int rt_blockptr, rt_last, rt_scr_pos, rt_numberofsus, rt_targetSUSize, rt_targetBlocks;
    pdet_t pd_msk; /* mskBurstDetector */
    jo_codec *rt_codec;
    long npackets;
    gb packets;
    /* 8400 bps C channel: DecodeC aerol.cpp:2187-2502 */
    struct { int p1[64], p2[64], b1[64], b2[64], len, tol, inverted; } cpd[2]; /* [0] = preambledetectorreal, [1] = preambledetectorimag */
    int cindex;
    unsigned char cdel[16 * 256 + 256]; int ncdel;
    gb voice;
};
#define RT_BLOCKSZ (64 * 95)
enum { RT_OK_R = 3, RT_OK_T = 5, RT_BAD = 0, RT_TEST_FAILED = 32, RT_NOTHING = 8, RT_FULL = 16 };

static void ev(jo_aerol *a, long idx, int kind, long value)
{
    int64_t row[3] = {idx, kind, value};
    gb_push(&a->events, row, sizeof(row));
}

jo_aerol *jo_aerol_create(int fb)
{
    jo_aerol *a = (jo_aerol *)calloc(1, sizeof(*a));
    /* ctor :904-977 */
    a->cntr = 1000000000; a->blockcnt = -1;
    a->codec = jo_codec_create(24);
    pdet_set(&a->pd_exact, 3780831379ULL, 32);
    pdet_set(&a->pd_imag, 3780831379ULL, 32);
    pdet_set(&a->pd_real, 3780831379ULL, 32);
    for (int i = 0; i < 64; i++) a->depermute[i] = (i * 27) % 64; /* AeroLInterleaver ctor :523-537 */
    { /* AeroLScrambler ctor aerol.h:397-420 */
        int state[15] = {1, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1};
        for (int k = 0; k < 5000; k++)
        {
            int val0 = state[0] ^ state[14];
            a->scr[k] = (unsigned char)val0;
            for (int i = 14; i > 0; i--) state[i] = state[i - 1];
            state[0] = val0;
        }
    }
    /* setSettings(fb,false) :990-1072: non-burst tolerances 0 */
    a->ifb = fb;
    switch (fb)
    {
    case 600: a->N = 6; a->dl2_len = 576 - 6; a->NumberOfBits = 1152; a->BitsInHeader = 16; a->TotalNumberOfBits = 16 + 1152 + 32; a->useingOQPSK = 0; break;
    case 1200: a->N = 9; a->dl2_len = 576 - 6; a->NumberOfBits = 1152; a->BitsInHeader = 16; a->TotalNumberOfBits = 16 + 1152 + 32; a->useingOQPSK = 0; break;
    case 8400: /* :1033-1043 */
        a->N = 4; a->dl2_len = 2714 - 6; a->NumberOfBits = 4096; a->BitsInHeader = 0; a->TotalNumberOfBits = 4096; a->useingOQPSK = 1;
        for (int q = 0; q < 2; q++) /* ctor :953-954 + setSettings non-burst tolerance 6 :1008-1009 */
        {
            a->cpd[q].len = 52; a->cpd[q].tol = 6;
            for (int i = 51, k = 0; i >= 0; i--, k++)
            {
                a->cpd[q].p1[k] = (int)((216866263330005ULL >> i) & 1);
                a->cpd[q].p2[k] = (int)((3012071630031408ULL >> i) & 1);
            }
        }
        a->ncdel = 16 * 4 * 64; /* deleaveredBlock.resize(16*4*64) */
        break;
    default: a->N = 78; a->dl2_len = 4992 - 6; a->NumberOfBits = 4992; a->BitsInHeader = 16 + 178; a->TotalNumberOfBits = 16 + 178 + 4992 + 64; a->useingOQPSK = 1; a->ifb = 10500; break;
    }
    a->blocksz = a->N * 64;
    a->block = (int *)calloc((size_t)a->blocksz, sizeof(int));
    a->dl2_sz = a->dl2_len + 1;
    a->dl2 = (int *)calloc((size_t)a->dl2_sz, sizeof(int));
    ev(a, 0, 0, 0); /* emit DataCarrierDetect(false) in the ctor */
    return a;
}
/* AeroL(parent) + setSettings(fb, true): the R/T-channel (burst) decoder for 10500 bps.  The phase-invariant unique-word detectors get
 * tolerance 4, the frame countdown becomes one second of bits (aerol.cpp:996-1003,1062-1070). */
jo_aerol *jo_aerol_create_burst(int fb)
{
    if (fb != 10500 && fb != 1200 && fb != 600) return NULL;
    jo_aerol *a = jo_aerol_create(fb);
    a->burstmode = 1;
    a->pd_imag.tollerence = 4; a->pd_real.tollerence = 4;
    pdet_set(&a->pd_msk, 3780831379ULL, 32); a->pd_msk.tollerence = 4; /* :963,1002 */
    a->TotalNumberOfBits = a->useingOQPSK ? a->ifb : a->ifb * 3;    /* 1 s / 3 s countdown :1062-1070 */
    a->rt_block = (int *)calloc(RT_BLOCKSZ, sizeof(int));
    a->rt_codec = jo_codec_create(24);
    a->rt_last = RT_NOTHING;
    return a;
}
long jo_aerol_take_packets(jo_aerol *a, int32_t *dst, long cap) { return gb_take(&a->packets, dst, 16 * sizeof(int32_t), cap); }

void jo_aerol_destroy(jo_aerol *a)
{
    if (a) free(a->voice.p);
    if (!a) return;
    free(a->rt_block); if (a->rt_codec) jo_codec_destroy(a->rt_codec); free(a->packets.p);
    free(a->block); free(a->dl2); jo_codec_destroy(a->codec); free(a->sus.p); free(a->events.p); free(a);
}
int jo_aerol_dcd(jo_aerol *a) { return a->datacd; }

/* AeroL::updateDCD (aerol.cpp:1109-1122); the reference calls it from a 1 s wall-clock QTimer, tests call it at chosen points of
 * the stream.  The DataCarrierDetect(false) it may emit is logged with the index of the next soft bit. */
int jo_aerol_tick_dcd(jo_aerol *a)
{
    if (a->datacdcountdown > 0) a->datacdcountdown -= 3;
    else if (a->datacdcountdown < 0) a->datacdcountdown = 0;
    if (a->datacd && !a->datacdcountdown)
    {
        a->datacd = 0;
        ev(a, a->nbits_total, 0, 0);
    }
    return a->datacd;
}
long jo_aerol_take_sus(jo_aerol *a, int32_t *dst, long cap) { return gb_take(&a->sus, dst, 16 * sizeof(int32_t), cap); }
long jo_aerol_take_events(jo_aerol *a, int64_t *dst, long cap) { return gb_take(&a->events, dst, 3 * sizeof(int64_t), cap); }
long jo_aerol_take_voice(jo_aerol *a, unsigned char *dst, long cap) { return gb_take(&a->voice, dst, 304, cap); }

static void block_done(jo_aerol *a, long bitidx) /* aerol.cpp:1553-1600 */
{
    a->blockcnt++;
    /* leaver.deinterleave_ba(block, 0): matrix_ba[k] = block[depermute[i]*N + j], k = j*64 + i */
    unsigned char *deleaved = (unsigned char *)malloc((size_t)a->blocksz);
    int k = 0;
    for (int j = 0; j < a->N; j++)
        for (int i = 0; i < 64; i++) deleaved[k++] = (unsigned char)a->block[a->depermute[i] * a->N + j];
    unsigned char *bits = (unsigned char *)malloc((size_t)a->blocksz);
    int nb = jo_decode_continuous(a->codec, deleaved, a->blocksz, bits);
    for (int h = 0; h < nb; h++)
    {
        /* dl2.update :1560 (DelayLine aerol.h:468-476) */
        a->dl2[a->dl2_ptr] = bits[h];
        a->dl2_ptr++; a->dl2_ptr %= a->dl2_sz;
        int v = a->dl2[a->dl2_ptr];
        /* scrambler.update :1563 */
        v ^= a->scr[a->scr_pos < 5000 ? a->scr_pos : 4999];
        a->scr_pos++;
        bits[h] = (unsigned char)v;
    }
    /* pack :1566-1578 */
    {
        int charptr = 0; unsigned char ch = 0;
        for (int h = 0; h < nb; h++)
        {
            ch |= bits[h] * 128;
            charptr++; charptr %= 8;
            if (charptr == 0) { if (a->ninfo < (int)sizeof(a->infofield)) a->infofield[a->ninfo++] = ch; ch = 0; }
            else ch >>= 1;
        }
    }
    free(deleaved); free(bits);
    if ((a->cntr - a->BitsInHeader) == (a->NumberOfBits - 1)) /* frame is done :1580 */
    {
        for (int kk = 0; kk < a->ninfo / 12; kk++)
        {
            const unsigned char *su = a->infofield + kk * 12;
            uint16_t crc_calc = crc16_bytes(su, 10);
            uint16_t crc_rec = (uint16_t)((su[11] << 8) | su[10]);
            if ((!crc_rec) && (crc_calc != crc_rec))
            {
                int tsum = 0;
                for (int ii = 0; ii < 10; ii++) tsum += su[ii];
                if (tsum == 0) crc_calc = 0;
            }
            if (crc_calc == crc_rec) { if (a->datacdcountdown < 12) a->datacdcountdown += 2; }
            else { if (a->datacdcountdown > 0) a->datacdcountdown -= 3; }
            if (!a->datacd && a->datacdcountdown > 2) { a->datacd = 1; ev(a, bitidx, 0, 1); }
            int32_t row[16];
            row[0] = (int32_t)a->nframes; row[1] = kk;
            for (int j = 0; j < 12; j++) row[2 + j] = su[j];
            row[14] = (crc_calc == crc_rec); row[15] = a->frameinfo;
            gb_push(&a->sus, row, sizeof(row));
        }
        a->nframes++;
    }
}

static int crc16_bits_check(const unsigned char *bits, int numberofbits) /* AeroLcrc16::calcusingbitsandcheck aerol.h:287-315 */
{
    uint16_t crc_rec = 0;
    for (int i = numberofbits - 1; i >= numberofbits - 16; i--) { crc_rec <<= 1; crc_rec |= bits[i]; }
    numberofbits -= 16;
    uint16_t crc = 0xFFFF;
    for (int i = 0; i < numberofbits; i++)
    {
        int crc_bit = crc & 1;
        crc >>= 1;
        if (crc_bit ^ bits[i]) crc = crc ^ 0x8408;
    }
    crc = (uint16_t)~crc;
    return crc_rec == crc;
}
static int rt_reset(jo_aerol *a) /* resetblockptr aerol.h:591-602 */
{
    a->rt_blockptr = 0;
    if (a->rt_last == RT_TEST_FAILED) { a->rt_last = RT_NOTHING; return RT_BAD; }
    a->rt_last = RT_NOTHING;
    return RT_NOTHING;
}
/* a decoded packet goes out as rows [packet, chunk, 12 bytes (zero padded), total bytes, type]; type 1 = R, 2 = T */
static void rt_emit(jo_aerol *a, const unsigned char *deconvol, int ndeconvol, int chop, int type)
{
    unsigned char info[1024];
    int ninfo = 0, charptr = 0; unsigned char ch = 0;
    for (int h = 0; h < ndeconvol; h++) /* packintobytes aerol.h:603-628 */
    {
        ch |= deconvol[h] * 128;
        charptr++; charptr %= 8;
        if (charptr == 0) { info[ninfo++] = ch; ch = 0; }
        else ch >>= 1;
    }
    ninfo -= chop; /* infofield.chop(1) for T packets */
    for (int c = 0; c * 12 < ninfo; c++)
    {
        int32_t row[16];
        row[0] = (int32_t)a->npackets; row[1] = c;
        for (int j = 0; j < 12; j++) row[2 + j] = (c * 12 + j < ninfo) ? info[c * 12 + j] : 0;
        row[14] = ninfo; row[15] = type;
        gb_push(&a->packets, row, sizeof(row));
    }
    a->npackets++;
}
static int rt_update(jo_aerol *a, int bit) /* RTChannelDeleaveFECScram::update aerol.h:785-873 */
{
    if (a->rt_blockptr >= RT_BLOCKSZ) return RT_FULL;
    a->rt_block[a->rt_blockptr] = bit;
    a->rt_blockptr++;
    if (((a->rt_blockptr - (64 * 5)) % (64 * 3)) != 0) return RT_NOTHING; /* C '%': also true for negative multiples -- none occur below 64*5 except 128 */
    const int blockptr = a->rt_blockptr, cols = blockptr / 64;
    /* deinterleave_ba(block, cols) aerol.cpp:603-625 */
    unsigned char *del = (unsigned char *)malloc((size_t)blockptr);
    int k = 0;
    for (int j = 0; j < cols; j++)
        for (int i = 0; i < 64; i++) del[k++] = (unsigned char)a->rt_block[a->depermute[i] * cols + j];
    unsigned char *dec = (unsigned char *)malloc((size_t)blockptr);
    const int nd = jo_decode_soft(a->rt_codec, del, blockptr, dec); /* Decode_soft(delBlock, blockptr) */
    a->rt_scr_pos = 0; /* scrambler.reset(); scrambler.update(deconvol) */
    for (int h = 0; h < nd; h++) dec[h] ^= a->scr[a->rt_scr_pos < 5000 ? a->rt_scr_pos : 4999], a->rt_scr_pos++;
    int result;
    if (blockptr == 64 * 5)
    {
        if (!crc16_bits_check(dec, 8 * 19)) { a->rt_last = RT_TEST_FAILED; result = RT_TEST_FAILED; }
        else { rt_emit(a, dec, nd, 0, 1); a->rt_blockptr = RT_BLOCKSZ; a->rt_last = RT_OK_R; result = RT_OK_R; }
    }
    else
    {
        int ok = crc16_bits_check(dec, 8 * 6);
        if (ok)
        {
            a->rt_numberofsus = 1 + (blockptr - (64 * 5)) / (64 * 3);
            for (int i = 0; i < a->rt_numberofsus && ok; i++) ok = crc16_bits_check(dec + (8 * 6) + (8 * 12) * i, 8 * 12);
        }
        if (!ok)
        {
            if (blockptr >= RT_BLOCKSZ) { a->rt_last = RT_BAD; result = RT_BAD; }
            else { a->rt_last = RT_TEST_FAILED; result = RT_TEST_FAILED; }
        }
        else { rt_emit(a, dec, nd, 1, 2); a->rt_blockptr = RT_BLOCKSZ; a->rt_last = RT_OK_T; result = RT_OK_T; }
    }
    free(del); free(dec);
    return result;
}

static int rt_update_msk(jo_aerol *a, int bit) /* RTChannelDeleaveFECScram::updateMSK aerol.h:631-782 */
{
    if (a->rt_blockptr >= RT_BLOCKSZ) return RT_FULL;
    a->rt_block[a->rt_blockptr] = bit;
    a->rt_blockptr++;
    const int blockptr = a->rt_blockptr, nb = blockptr / 64;
    if (!((((blockptr - (64 * 5)) % (64 * 3)) == 0) && (nb == 5 || nb == a->rt_targetBlocks || nb == 11 || nb == 50))) return RT_NOTHING;
    /* deinterleaveMSK_ba(block, blocks) aerol.cpp:671-711: 5 columns, then groups of 3 */
    unsigned char *del = (unsigned char *)malloc((size_t)blockptr);
    int k = 0;
    for (int j = 0; j < 5; j++)
        for (int i = 0; i < 64; i++) del[k++] = (unsigned char)a->rt_block[a->depermute[i] * 5 + j];
    for (int proc = 5; k < nb * 64; proc += 3)
        for (int j = 0; j < 3; j++)
            for (int i = 0; i < 64; i++) del[k++] = (unsigned char)a->rt_block[64 * proc + a->depermute[i] * 3 + j];
    unsigned char *dec = (unsigned char *)malloc((size_t)blockptr);
    const int nd = jo_decode_soft(a->rt_codec, del, blockptr, dec);
    a->rt_scr_pos = 0;
    for (int h = 0; h < nd; h++) dec[h] ^= a->scr[a->rt_scr_pos < 5000 ? a->rt_scr_pos : 4999], a->rt_scr_pos++;
    int result = RT_NOTHING;
    if (blockptr == 64 * 5)
    {
        a->rt_targetSUSize = 0; a->rt_targetBlocks = 0;
        if (crc16_bits_check(dec, 8 * 19)) { rt_emit(a, dec, nd, 0, 1); a->rt_blockptr = RT_BLOCKSZ; a->rt_last = RT_OK_R; result = RT_OK_R; }
        else result = RT_NOTHING; /* lastpacketstate untouched */
    }
    else if (!crc16_bits_check(dec, 8 * 6)) { a->rt_last = RT_BAD; result = RT_BAD; }
    else if (nb == 11)
    {
        /* the signal unit after the initial one tells how many there are :709-729 */
        const unsigned char *isu = dec + (8 * 6) + (8 * 12) * 1;
        int bin = 2 + (isu[0] * 1 + isu[1] * 2 + isu[2] * 4 + isu[3] * 8 + isu[4] * 16 + isu[5] * 32);
        a->rt_targetSUSize = bin;
        if (a->rt_targetSUSize >= 16) a->rt_targetSUSize = a->rt_targetSUSize / 2 + 1;
        a->rt_targetBlocks = ((a->rt_targetSUSize + 1) * 3) + 2;
        result = RT_NOTHING;
    }
    else if (nb == a->rt_targetBlocks)
    {
        /* the per-unit CRCs are counted but cannot fail the packet (ok <= targetSUSize always) :732-766 */
        rt_emit(a, dec, nd, 1, 2);
        a->rt_numberofsus = a->rt_targetSUSize;
        a->rt_blockptr = RT_BLOCKSZ; a->rt_last = RT_OK_T; result = RT_OK_T;
    }
    free(del); free(dec);
    return result;
}

/* OQPSKPreambleDetectorAndAmbiguityCorrection::Update aerol.cpp:848-895 (q: 0 real, 1 imag) */
static int cpd_update(jo_aerol *a, int q, int val)
{
    const int n = a->cpd[q].len, tol = a->cpd[q].tol;
    int *b1 = a->cpd[q].b1, *b2 = a->cpd[q].b2;
    int xorsum = 0;
    for (int i = 0; i < n - 1; i++) { b1[i] = b1[i + 1]; xorsum += b1[i] ^ a->cpd[q].p1[i]; }
    xorsum += val ^ a->cpd[q].p1[n - 1];
    b1[n - 1] = val;
    if (xorsum >= (n - tol)) { a->cpd[q].inverted = 1; return 1; }
    if (xorsum <= tol) { a->cpd[q].inverted = 0; return 1; }
    xorsum = 0;
    for (int i = 0; i < n - 1; i++) { b2[i] = b2[i + 1]; xorsum += b2[i] ^ a->cpd[q].p2[i]; }
    xorsum += val ^ a->cpd[q].p2[n - 1];
    b2[n - 1] = val;
    if (xorsum >= (n - tol)) { a->cpd[q].inverted = 1; return 1; }
    if (xorsum <= tol) { a->cpd[q].inverted = 0; return 1; }
    return 0;
}

/* end of a C-channel frame aerol.cpp:2318-2490: depuncture, Viterbi, delay line, scrambler, 3 sub-band signal units, 300 voice bytes */
static void c_frame_done(jo_aerol *a, long bitidx)
{
    /* puncturedCode.depunture_soft_block(deleaveredBlock, depuncturedBlock, 4, true) :2505-2518 (the last source byte is not used) */
    unsigned char *dep = (unsigned char *)malloc((size_t)a->ncdel * 2 + 16);
    int nd = 0, ptr = 0;
    for (int i = 0; i < a->ncdel - 1; i++)
    {
        ptr++;
        dep[nd++] = a->cdel[i];
        if (ptr >= 4 - 1) dep[nd++] = 128;
        ptr %= (4 - 1);
    }
    unsigned char *bits = (unsigned char *)calloc((size_t)nd + 3000, 1);
    int nb = jo_decode_continuous(a->codec, dep, nd, bits);
    (void)nb; /* deconvol.resize(2714): drops the trailing dummy bits (or zero-extends a short first frame) */
    for (int h = 0; h < 2714; h++)
    {
        int v = (h < nb) ? bits[h] : 0;
        a->dl2[a->dl2_ptr] = v;
        a->dl2_ptr++; a->dl2_ptr %= a->dl2_sz;
        v = a->dl2[a->dl2_ptr];
        v ^= a->scr[a->scr_pos < 5000 ? a->scr_pos : 4999];
        a->scr_pos++;
        bits[h] = (unsigned char)v;
    }
    int charptr = 0; unsigned char ch = 0;
    unsigned char info[16]; int ninfo = 0, kk = 0;
    for (int y = 0; y < 24; y++)
    {
        const int offset = y * (1 + 96 + 12);
        for (int h = offset + 97; h < offset + 109; h++)
        {
            ch |= bits[h] * 128;
            charptr++; charptr %= 8;
            if (charptr == 0) { info[ninfo++] = ch; ch = 0; }
            else ch >>= 1;
        }
        if (ninfo == 12)
        {
            uint16_t crc_calc = crc16_bytes(info, 10);
            uint16_t crc_rec = (uint16_t)((info[11] << 8) | info[10]);
            if (crc_calc == crc_rec) { if (a->datacdcountdown < 12) a->datacdcountdown += 2; }
            else { if (a->datacdcountdown > 0) a->datacdcountdown -= 5; }
            if (!a->datacd && a->datacdcountdown > 2) { a->datacd = 1; ev(a, bitidx, 0, 1); }
            int32_t row[16];
            row[0] = (int32_t)a->nframes; row[1] = kk++;
            for (int j = 0; j < 12; j++) row[2 + j] = info[j];
            row[14] = (crc_calc == crc_rec); row[15] = 0;
            gb_push(&a->sus, row, sizeof(row));
            ninfo = 0;
        }
    }
    /* voice data :2454-2478 (ch / charptr carry on from above; both are 0 again after 288 bits) */
    unsigned char row[304];
    memset(row, 0, sizeof(row));
    { const uint32_t f = (uint32_t)a->nframes; memcpy(row, &f, 4); }
    int nv = 0, bitsin = 0;
    for (int h = 1; h < 2714; h++)
    {
        ch |= bits[h] * 128;
        charptr++; charptr %= 8;
        if (charptr == 0) { if (nv < 300) row[4 + nv++] = ch; ch = 0; }
        else ch >>= 1;
        bitsin++;
        if (bitsin == 96) { bitsin = 0; h += 13; }
    }
    gb_push(&a->voice, row, sizeof(row));
    a->nframes++;
    free(dep); free(bits);
}

/* DecodeC aerol.cpp:2187-2502 */
static void c_write(jo_aerol *a, const int16_t *sb, long n)
{
    for (long i = 0; i < n; i++)
    {
        const long bitidx = a->nbits_total + i;
        int bit = (((unsigned char)sb[i]) >= 128) ? 1 : 0;
        unsigned short soft_bit = (unsigned short)sb[i];
        int gotsync = 0;
        a->realimag++; a->realimag %= 2;
        const int q = a->realimag ? 0 : 1; /* realimag != 0: preambledetectorreal */
        if (a->cntr > a->NumberOfBits - 112 || a->cntr <= 0)
        {
            gotsync = cpd_update(a, q, bit);
            if (!a->gotsync_last) { a->gotsync_last = gotsync; gotsync = 0; }
            else a->gotsync_last = 0;
        }
        else { gotsync = 0; a->gotsync_last = 0; }
        if (a->cpd[q].inverted)
        {
            bit = 1 - bit;
            if (soft_bit > 128) soft_bit = 255 - soft_bit;
            else if (soft_bit < 128) soft_bit = 255 - soft_bit;
        }
        if (gotsync)
        {
            a->cntr = -1; a->cindex = -1;
            a->ncdel = 0;
            a->scr_pos = 0;
            ev(a, bitidx, 2, 1);
        }
        else
        {
            if (a->cntr < 1000000000) a->cntr++;
            if (a->cntr <= a->NumberOfBits - 1) { a->cindex++; a->block[a->cindex] = soft_bit; }
            if (a->cindex == 255)
            {
                /* leaver.deinterleave_ba(block, 4) appended to deleaveredBlock :2306-2316 */
                for (int j = 0; j < 4; j++)
                    for (int k = 0; k < 64; k++)
                        if (a->ncdel < (int)sizeof(a->cdel)) a->cdel[a->ncdel++] = (unsigned char)a->block[a->depermute[k] * 4 + j];
                a->cindex = -1;
            }
            if (a->cntr == a->NumberOfBits - 1) { c_frame_done(a, bitidx); a->cindex = -1; }
        }
    }
    a->nbits_total += n;
}

void jo_aerol_write(jo_aerol *a, const int16_t *sb, long n)
{
    if (a->ifb == 8400) { c_write(a, sb, n); return; } /* processDemodulatedSoftBits :2084-2087 */
    /* Decode(): decodedbytes.clear() etc. are text; the loop :1131-2027 */
    for (long i = 0; i < n; i++)
    {
        const long bitidx = a->nbits_total + i;
        int bit = (((unsigned char)sb[i]) >= 128) ? 1 : 0;
        unsigned short soft_bit = (unsigned short)sb[i];
        if (sb[i] < 0) { a->muw = 0; continue; }
        if (a->muw < 100000) a->muw++;
        int gotsync;
        if (a->useingOQPSK)
        {
            a->realimag++; a->realimag %= 2;
            pdet_t *pd = a->realimag ? &a->pd_imag : &a->pd_real;
            if (a->cntr > a->NumberOfBits - 68 || a->cntr <= 0 || !a->datacd)
            {
                gotsync = pdet_update_pi(pd, bit);
                if (!a->gotsync_last) { a->gotsync_last = gotsync; gotsync = 0; }
                else a->gotsync_last = 0;
            }
            else { gotsync = 0; a->gotsync_last = 0; }
            /* burst mode: the unique word has to come about 80 soft bits after the demodulator's start-of-burst marker :1192-1200 */
            if (gotsync && a->burstmode && a->ifb == 10500 && abs(a->muw - 80) > 150) gotsync = 0;
            if (pd->inverted)
            {
                bit = 1 - bit;
                if (soft_bit > 128) soft_bit = 255 - soft_bit;
                else if (soft_bit < 128) soft_bit = 255 - soft_bit;
            }
        }
        else if (a->burstmode) /* 600/1200 bps bursts: phase-invariant detector, the word has to come within 250 soft bits of the marker :1235-1267 */
        {
            const int inverted = a->pd_msk.inverted;
            gotsync = pdet_update_pi(&a->pd_msk, bit);
            if (a->muw > 250 && gotsync) { a->pd_msk.inverted = inverted; gotsync = 0; }
            if (a->pd_msk.inverted)
            {
                bit = 1 - bit;
                if (soft_bit > 128) soft_bit = 255 - soft_bit;
                else if (soft_bit < 128) soft_bit = 255 - soft_bit;
            }
        }
        else gotsync = pdet_update_exact(&a->pd_exact, bit);

        if (a->cntr < 1000000000) a->cntr++;
        if (a->cntr < 16)
        {
            if (a->cntr == 0)
            {
                a->frameinfo = (uint16_t)bit; a->ninfo = 0;
                if (a->burstmode) /* R and T channels have no header: a dummy one :1281-1294 */
                {
                    a->formatid = 1; a->supfrmaker = 0; a->framecounter1 = 0; a->framecounter2 = 0;
                    a->cntr = 16;
                    if (rt_reset(a) == RT_BAD) ev(a, bitidx, 3, 0); /* " Bad R/T Packet" */
                }
            }
            else { a->frameinfo <<= 1; a->frameinfo |= (uint16_t)bit; }
        }
        if (a->cntr == 15)
        {
            uint16_t tval = a->frameinfo;
            a->frameinfo = a->lastframeinfo;
            a->lastframeinfo = tval;
            a->formatid = (a->frameinfo >> 12) & 0xF; a->supfrmaker = (a->frameinfo >> 8) & 0xF;
            a->framecounter1 = (a->frameinfo >> 4) & 0xF; a->framecounter2 = a->frameinfo & 0xF;
        }
        if (a->cntr >= 16 && a->burstmode)
        {
            if ((a->useingOQPSK ? rt_update(a, soft_bit) : rt_update_msk(a, soft_bit)) == RT_BAD) ev(a, bitidx, 3, 0); /* :1531 " Bad R/T Packet" */
        }
        else if (a->cntr >= 16)
        {
            if (a->cntr == 16) a->blockcnt = -1;
            int idx = (a->cntr - a->BitsInHeader) % a->blocksz;
            if (idx < 0) idx = 0;
            a->block[idx] = soft_bit;
            if (idx == (a->blocksz - 1)) block_done(a, bitidx);
        }
        if (gotsync)
        {
            if (!a->burstmode && a->cntr + 1 != a->TotalNumberOfBits) ev(a, bitidx, 1, a->cntr + 1); /* "Error short frame!!!" (isudata.reset()) */
            a->cntr = -1;
            a->datacd = 1; a->datacdcountdown = 12;
            ev(a, bitidx, 0, 1);
            ev(a, bitidx, 2, 0);
            a->scr_pos = 0;
        }
        if (a->cntr + 1 == a->TotalNumberOfBits)
        {
            a->scr_pos = 0; a->cntr = -1;
            if (a->burstmode) /* end of signal :2018-2027: stop, carrier detect low, and the REST OF THIS GROUP of soft bits is dropped */
            {
                a->cntr = 1000000000;
                a->datacd = 0; a->datacdcountdown = 0;
                ev(a, bitidx, 0, 0);
                break;
            }
        }
    }
    a->nbits_total += n;
}
