// k_aerol_burst.h -- AeroL in burst mode: the R / T channel packet search of SURVEY.md section 8 row f2 (10500 bps).
//   AeroL::Decode with burstmode = true        JAERO/aerol.cpp:1124-1345,1985-2030 (unique word with tolerance 4 and the
//                                              "about 80 soft bits after the start-of-burst marker" rule, dummy header, end of signal)
//   RTChannelDeleaveFECScram::update           JAERO/aerol.h:785-873 (block fill, trial decodes at 2, 5, 8, .. 95 interleaver columns)
//   AeroLInterleaver::deinterleave_ba(.,cols)  JAERO/aerol.cpp:603-625;  JConvolutionalCodec::Decode_soft jconvolutionalcodec.cpp:90-119
//   AeroLcrc16::calcusingbitsandcheck          JAERO/aerol.h:287-315
// One channel per lane walks its soft bits until the block reaches a length at which the reference tries to decode it; the trial
// (deinterleave -> k_viterbi with that channel's length -> descramble -> CRCs -> R packet / T packet / keep collecting) runs for all
// such channels at once, then the lanes go on: rounds, as in the continuous pipeline.  Tuned for large banks in round 5: trials one
// block per lane (k_viterbi_lanes with per-lane lengths), eight decoded bits per access in the post kernel, eight soft entries per load and
// whole groups of inert entries at once in the bit walk.  The reference drops the rest of the demodulator's current group of soft bits when the one-second frame
// countdown ends; the groups are re-derived from the stream with the demodulator's rule (a marker is one entry, soft bits come in
// pairs, a group is complete at >= 32 entries after a pair; JAERO/burstoqpskdemodulator.cpp:546-585).
#pragma once
#include "k_aerol.h"

#define RT_BLOCKSZ (64 * 95)
enum { RT_NOTHING = 8, RT_TEST_FAILED = 32, RT_BAD = 0, RT_OK_R = 3, RT_OK_T = 5 };
enum // burst-only fields, stored in slots the continuous pipeline uses for its own block bookkeeping
{
    BI_RT_BLOCKPTR = AI_BLOCKCNT, BI_RT_LAST = AI_NINFO, BI_GRP_CNT = AI_SCR_POS, BI_GRP_PAIR = AI_DL2_PTR, BI_GRP_SKIP = AI_VBLOCKS,
    BI_TRIAL_LEN = AI_BULK_LEN, BI_RESUME_GEND = AI_BULK_SRC, BI_NPACKETS = AI_NFRAMES, BI_TARGET_BLOCKS = AI_BULK_DST, BI_TARGET_SUS = AI_BULK_FLAGS
};

// WIDE: the rows of soft are 16-byte aligned (base and stride): entries are taken eight at a time
template <bool WIDE>
__global__ __launch_bounds__(64) void k_aerolb_bits(const AGeom g, const APtrs p, const int16_t *__restrict__ soft, const int *__restrict__ counts, int stride)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= g.nch) return;
    int cntr = ALD(AI_CNTR), datacd = ALD(AI_DATACD), dcdcount = ALD(AI_DCDCOUNT), gotsync_last = ALD(AI_GOTSYNC_LAST), realimag = ALD(AI_REALIMAG);
    int muw = ALD(AI_MUW), inv_imag = ALD(AI_INV_IMAG), inv_real = ALD(AI_INV_REAL), ev_cnt = ALD(AI_EV_CNT), overflow = ALD(AI_OVERFLOW);
    unsigned pd_imag = (unsigned)ALD(AI_PD_IMAG), pd_real = (unsigned)ALD(AI_PD_REAL);
    int blockptr = ALD(BI_RT_BLOCKPTR), rt_last = ALD(BI_RT_LAST), gcnt = ALD(BI_GRP_CNT), pair = ALD(BI_GRP_PAIR), skip = ALD(BI_GRP_SKIP);
    int pos = ALD(AI_IN_POS), resume = ALD(AI_RESUME);
    const int target_blocks = ALD(BI_TARGET_BLOCKS); // 600 / 1200 bps T packets: set by k_aerolb_post at 11 blocks
    const long long nbits0 = ((long long)(unsigned)ALD(AI_NBITS_LO)) | ((long long)ALD(AI_NBITS_HI) << 32);
    const int n = counts[ch];
    const int16_t *sb = soft + (size_t)ch * stride;
    uint8_t *blk = p.rx + (size_t)ch * RT_BLOCKSZ;
    int has_trial = 0, gotsync = 0, gend = 0;
    // block bytes leave eight at a time (one store per lane and bit is a write request per lane and bit); a word begun in an earlier
    // launch is taken up again from the row, a word unfinished at the end of this one is written as far as it goes
    unsigned long long bacc = 0;
    if ((blockptr & 7) && blockptr < RT_BLOCKSZ) bacc = *(const unsigned long long *)(blk + (blockptr & ~7)) & ((1ull << (8 * (blockptr & 7))) - 1ull);

    auto part_b = [&](long long bitidx) { // aerol.cpp:1985-2030
        if (gotsync)
        {
            cntr = -1;
            datacd = 1; dcdcount = 12;
            aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 0, 1);
            aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 2, 0);
        }
        if (cntr + 1 == g.TotalNumberOfBits)
        {
            // end of signal: stop, data carrier detect low, and the rest of this group of soft bits is dropped
            cntr = 1000000000;
            datacd = 0; dcdcount = 0;
            aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 0, 0);
            skip = 1;
        }
        gotsync = 0;
        if (gend) { gcnt = 0; skip = 0; }
    };
    if (resume)
    {
        gotsync = ALD(AI_RESUME_GOTSYNC); gend = ALD(BI_RESUME_GEND);
        part_b(nbits0 + pos);
        pos++;
        resume = 0;
    }
    // Soft entries in aligned groups of eight (WIDE: one 16-byte load per lane and group).  Every lane walks its own row, so a 2-byte load
    // per bit is a cache-line request per lane and bit and -- worse -- a vmcnt(0) per bit, which on this target also waits for the block
    // store of the bit before: ~1 400 cycles per bit and wavefront (1.8 ms for the 3 600 entries behind a burst).  The wait belongs inside
    // the group switch, once per eight bits: left to the compiler it lands behind the join.
    int4 cur = {0, 0, 0, 0};
    int curg = -1;
    auto fetch = [&](int q) -> int {
        if (!WIDE) return sb[q];
        const int gq = q >> 3;
        if (gq != curg)
        {
            cur = ((const int4 *)sb)[gq];
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
            curg = gq;
        }
        const int k = q & 7;
        const unsigned w = (unsigned)((k & 4) ? ((k & 2) ? cur.w : cur.z) : ((k & 2) ? cur.y : cur.x));
        return (int)(short)(w >> ((k & 1) * 16));
    };
    while (pos < n)
    {
        if (WIDE && g.oqpsk)
        {
            // Between the unique word and the point where the countdown re-arms the detectors (pre-increment cntr in [1, NumberOfBits -
            // 68], carrier detected) a soft bit changes nothing but the counters -- pair and group count, muw, cntr; realimag and the
            // detectors' registers come out of eight such bits as they went in -- and, while the collector runs, the block.  Aligned
            // groups of eight entries without a start-of-burst marker are taken in one go: behind a packet (collector stopped, blockptr =
            // RT_BLOCKSZ; the demodulator emits ~3 600 entries there) and inside one as long as no trial length (a multiple of 64) lies
            // within the eight.  Anything else, bit by bit below.
            while (!skip && datacd && cntr >= 1 && cntr + 7 <= g.NumberOfBits - 68 && (pos & 7) == 0 && pos + 8 <= n)
            {
                const bool collecting = blockptr < RT_BLOCKSZ;
                if (collecting && (cntr < 16 || (blockptr & 63) + 8 >= 64)) break;
                cur = ((const int4 *)sb)[pos >> 3];
                __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
                curg = pos >> 3;
                if ((cur.x | cur.y | cur.z | cur.w) & (int)0x80008000u) break; // a marker: bit by bit
                if (collecting)
                {
                    const unsigned e[4] = {(unsigned)cur.x, (unsigned)cur.y, (unsigned)cur.z, (unsigned)cur.w};
                    unsigned long long bytes8 = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++)
                    {
                        const int inv = ((realimag ^ k ^ 1) & 1) ? inv_imag : inv_real; // realimag toggles before it is looked at
                        unsigned sbk = (e[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                        if (inv && sbk != 128u) sbk = 255u - sbk;
                        bytes8 |= (unsigned long long)(sbk & 0xFFu) << (8 * k);
                    }
                    const int sh = 8 * (blockptr & 7);
                    *(unsigned long long *)(blk + (blockptr & ~7)) = bacc | (bytes8 << sh);
                    bacc = sh ? (bytes8 >> (64 - sh)) : 0ull;
                    blockptr += 8;
#ifdef AEROLB_EMUL_COUNT
                    g_aerolb_fast_fill_groups++;
#endif
                }
#pragma unroll
                for (int k = 0; k < 8; k++)
                {
                    pair ^= 1;
                    gcnt++;
                    gend = (pair == 0 && gcnt >= 32) ? 1 : 0;
                    if (gend) gcnt = 0;
                }
                muw = min(100000, muw + 8);
                cntr += 8;
                gotsync = 0; gotsync_last = 0;
                pos += 8;
#ifdef AEROLB_EMUL_COUNT
                g_aerolb_fast_groups++;
#endif
            }
            if (pos >= n) break;
        }
        const int v = fetch(pos);
        const long long bitidx = nbits0 + pos;
        const bool isneg = v < 0;
        if (!isneg) pair ^= 1;
        gcnt++;
        gend = (!isneg && pair == 0 && gcnt >= 32) ? 1 : 0;
        if (skip || isneg)
        {
            if (isneg && !skip) muw = 0; // start-of-burst marker (aerol.cpp:1146-1152)
            if (gend) { gcnt = 0; skip = 0; }
            pos++;
            continue;
        }
        int bit = (((unsigned)v & 0xFFu) >= 128u) ? 1 : 0;
        unsigned soft_bit = (unsigned)v & 0xFFFFu;
        if (muw < 100000) muw++;
        int inverted;
        if (!g.oqpsk)
        {
            // 600 / 1200 bps: one phase-invariant detector (tolerance 4) on every bit; a word later than 250 soft bits after the
            // marker is refused and leaves the inversion as it was (aerol.cpp:1235-1267).  State in the imag-arm slots.
            const int inv_before = inv_imag;
            pd_imag = (pd_imag << 1) | (unsigned)bit;
            const int xorsum = __popc(pd_imag ^ AEROL_UW);
            gotsync = 0;
            if (xorsum >= 32 - 4) { inv_imag = 1; gotsync = 1; }
            else if (xorsum <= 4) { inv_imag = 0; gotsync = 1; }
            if (muw > 250 && gotsync) { inv_imag = inv_before; gotsync = 0; }
            inverted = inv_imag;
        }
        else
        {
        realimag ^= 1;
        unsigned pd = realimag ? pd_imag : pd_real;
        inverted = realimag ? inv_imag : inv_real;
        if (cntr > g.NumberOfBits - 68 || cntr <= 0 || !datacd)
        {
            // PreambleDetectorPhaseInvariant::Update with tolerance 4 (aerol.cpp:781-804, 996-1003)
            pd = (pd << 1) | (unsigned)bit;
            const int xorsum = __popc(pd ^ AEROL_UW);
            gotsync = 0;
            if (xorsum >= 32 - 4) { inverted = 1; gotsync = 1; }
            else if (xorsum <= 4) { inverted = 0; gotsync = 1; }
            if (!gotsync_last) { gotsync_last = gotsync; gotsync = 0; }
            else gotsync_last = 0;
        }
        else { gotsync = 0; gotsync_last = 0; }
        if (realimag) { pd_imag = pd; inv_imag = inverted; }
        else { pd_real = pd; inv_real = inverted; }
        if (gotsync && abs(muw - 80) > 150) gotsync = 0; // the unique word comes ~80 soft bits after the marker (:1192-1200)
        }
        if (inverted)
        {
            bit = 1 - bit;
            if (soft_bit != 128u) soft_bit = 255u - soft_bit;
        }
        if (cntr < 1000000000) cntr++;
        if (cntr == 0)
        {
            // R and T channels have no header: dummy one, and the packet collector starts over (:1281-1294; resetblockptr aerol.h:591)
            cntr = 16;
            blockptr = 0; bacc = 0;
            if (rt_last == RT_TEST_FAILED) aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 3, 0); // " Bad R/T Packet"
            rt_last = RT_NOTHING;
        }
        bool trial = false;
        if (cntr >= 16 && blockptr < RT_BLOCKSZ)
        {
            bacc |= (unsigned long long)(soft_bit & 0xFFu) << (8 * (blockptr & 7));
            if ((blockptr & 7) == 7) { *(unsigned long long *)(blk + (blockptr & ~7)) = bacc; bacc = 0; }
            blockptr++;
            // ((blockptr - 64*5) % (64*3)) == 0 in C: also -192, i.e. two columns
            trial = (blockptr == 128) || (blockptr >= 320 && ((blockptr - 320) % 192) == 0);
            if (!g.oqpsk) // updateMSK (aerol.h:649): only at 5, 11, 50 and the announced number of 64-bit blocks
            {
                const int nb = blockptr / 64;
                trial = trial && (nb == 5 || nb == target_blocks || nb == 11 || nb == 50);
            }
        }
        if (trial)
        {
            has_trial = 1; resume = 1;
            break; // the verdict on this length (k_aerolb_post) comes before the rest of this soft bit
        }
        part_b(bitidx);
        pos++;
    }
    if ((blockptr & 7) && blockptr < RT_BLOCKSZ) *(unsigned long long *)(blk + (blockptr & ~7)) = bacc;
    ALD(AI_CNTR) = cntr; ALD(AI_DATACD) = datacd; ALD(AI_DCDCOUNT) = dcdcount; ALD(AI_GOTSYNC_LAST) = gotsync_last; ALD(AI_REALIMAG) = realimag;
    ALD(AI_MUW) = muw; ALD(AI_INV_IMAG) = inv_imag; ALD(AI_INV_REAL) = inv_real; ALD(AI_EV_CNT) = ev_cnt; ALD(AI_OVERFLOW) = overflow;
    ALD(AI_PD_IMAG) = (int)pd_imag; ALD(AI_PD_REAL) = (int)pd_real;
    ALD(BI_RT_BLOCKPTR) = blockptr; ALD(BI_RT_LAST) = rt_last; ALD(BI_GRP_CNT) = gcnt; ALD(BI_GRP_PAIR) = pair; ALD(BI_GRP_SKIP) = skip;
    ALD(AI_IN_POS) = pos; ALD(AI_RESUME) = resume; ALD(AI_RESUME_GOTSYNC) = gotsync; ALD(BI_RESUME_GEND) = gend;
    ALD(AI_HAS_BLOCK) = has_trial; ALD(BI_TRIAL_LEN) = has_trial ? blockptr : 0;
}

// deinterleave_ba(block, cols) of the channels that reached a trial length: out[j*64 + i] = block[((i*27) % 64) * cols + j]
__global__ __launch_bounds__(256) void k_aerolb_deint(const AGeom g, const APtrs p)
{
    __shared__ __attribute__((aligned(16))) uint8_t blk[4][RT_BLOCKSZ];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + w;
    if (ch >= g.nch) return;
    if (!ALD(AI_HAS_BLOCK)) return; // wave-uniform
    const int len = ALD(BI_TRIAL_LEN), cols = len / 64;
    const uint8_t *src = p.rx + (size_t)ch * RT_BLOCKSZ;
    // (round 6) sixteen bytes per lane and request in (len is a multiple of 64, the rows of rx and blk are 16-byte aligned), a word per lane out: it was
    // one byte per lane either way, 2 x len / 64 dependent rounds per trial
    for (int q = lane * 16; q < len; q += 1024) *(uint4 *)(blk[w] + q) = *(const uint4 *)(src + q);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // this wavefront's LDS writes before its LDS reads (one wavefront per channel: no barrier)
    uint8_t *dst = p.deint + (size_t)ch * RT_BLOCKSZ;
    const int perm = (lane * 27) & 63;
    if (g.oqpsk)
    {
        // out[j * 64 + i] = block[((i * 27) % 64) * cols + j]: lane -> (four consecutive i, one j of four): 256 contiguous bytes per store instruction
        const int i4 = (lane & 15) * 4, jo = lane >> 4;
        const int p0 = ((i4 + 0) * 27) & 63, p1 = ((i4 + 1) * 27) & 63, p2 = ((i4 + 2) * 27) & 63, p3 = ((i4 + 3) * 27) & 63;
        for (int j = jo; j < cols; j += 4)
        {
            const unsigned v = (unsigned)blk[w][p0 * cols + j] | ((unsigned)blk[w][p1 * cols + j] << 8) | ((unsigned)blk[w][p2 * cols + j] << 16) |
                               ((unsigned)blk[w][p3 * cols + j] << 24);
            *(unsigned *)(dst + j * 64 + i4) = v;
        }
    }
    else
    {
        // deinterleaveMSK_ba (aerol.cpp:671-711): the first five columns are one 64 x 5 block, every following three a 64 x 3 block
        for (int j = 0; j < 5 && j < cols; j++) dst[j * 64 + lane] = blk[w][perm * 5 + j];
        for (int proc = 5; proc + 3 <= cols; proc += 3)
            for (int j = 0; j < 3; j++) dst[(proc + j) * 64 + lane] = blk[w][64 * proc + perm * 3 + j];
    }
}

// decoded bits lie one per byte (0 / 1); eight of them from one 8-byte word: bit k of the result = byte k (the first bit is the LSB)
__device__ __forceinline__ unsigned aerolb_pack8(unsigned long long w)
{
    return (unsigned)(((w & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
}

// calcusingbitsandcheck; bits 8-byte aligned, numberofbits a multiple of 8 (48, 96 and 152 here): one load per eight bits
__device__ __forceinline__ bool aerolb_crc_bits(const uint8_t *bits, int numberofbits)
{
    const unsigned long long *b8 = (const unsigned long long *)bits;
    const int nw = numberofbits / 8;
    const unsigned crc_rec = aerolb_pack8(b8[nw - 2]) | (aerolb_pack8(b8[nw - 1]) << 8); // bit j = bits[numberofbits - 16 + j]
    unsigned crc = 0xFFFFu;
    for (int h = 0; h < nw - 2; h++)
    {
        const unsigned v = aerolb_pack8(b8[h]);
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const unsigned crc_bit = crc & 1u;
            crc >>= 1;
            if (crc_bit ^ ((v >> k) & 1u)) crc ^= 0x8408u;
        }
    }
    crc = (~crc) & 0xFFFFu;
    return crc_rec == crc;
}

// the verdict on a trial length: descramble, CRCs, emit the packet (rows [packet, chunk, 12 bytes, total bytes, type]) or keep collecting
__global__ __launch_bounds__(64) void k_aerolb_post(const AGeom g, const APtrs p)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= g.nch) return;
    if (!ALD(AI_HAS_BLOCK)) return;
    const int blockptr = ALD(BI_TRIAL_LEN), nd = blockptr / 2;
    uint8_t *dec = p.vbits + (size_t)ch * (RT_BLOCKSZ / 2);
    // the decoder leaves the last K-1 bits of its zero-initialised output untouched; scrambler.reset(); scrambler.update(deconvol)
    // (nd <= 3040 < 5000).  Eight bits per access: rows are 3040 bytes apart and nd is a multiple of 32
    {
        unsigned long long *d8 = (unsigned long long *)dec;
        const unsigned long long *s8 = (const unsigned long long *)p.scr;
        const int nw = nd / 8;
        for (int h = 0; h < nw; h++)
        {
            unsigned long long w = d8[h];
            if (h == nw - 1) w &= 0xFFFFull; // bytes nd-6 .. nd-1
            d8[h] = w ^ s8[h];
        }
    }
    int result, type = 0, chop = 0;
    bool keep_last = false;
    if (!g.oqpsk)
    {
        // RTChannelDeleaveFECScram::updateMSK (aerol.h:655-782)
        const int nb = blockptr / 64;
        result = RT_NOTHING;
        keep_last = true; // "Nothing" leaves lastpacketstate alone
        if (nb == 5)
        {
            ALD(BI_TARGET_SUS) = 0; ALD(BI_TARGET_BLOCKS) = 0;
            if (aerolb_crc_bits(dec, 8 * 19)) { result = RT_OK_R; type = 1; keep_last = false; }
        }
        else if (!aerolb_crc_bits(dec, 8 * 6)) { result = RT_BAD; keep_last = false; }
        else if (nb == 11)
        {
            // the signal unit after the initial one announces how many there are (:709-729)
            const uint8_t *isu = dec + (8 * 6) + (8 * 12) * 1;
            int tsus = 2 + (int)(aerolb_pack8(*(const unsigned long long *)isu) & 63u); // isu[0] + 2 isu[1] + .. + 32 isu[5]
            if (tsus >= 16) tsus = tsus / 2 + 1;
            ALD(BI_TARGET_SUS) = tsus; ALD(BI_TARGET_BLOCKS) = ((tsus + 1) * 3) + 2;
        }
        else if (nb == ALD(BI_TARGET_BLOCKS)) { result = RT_OK_T; type = 2; chop = 1; keep_last = false; } // the per-unit CRCs cannot fail it (:732-766)
    }
    else if (blockptr == 64 * 5)
    {
        if (!aerolb_crc_bits(dec, 8 * 19)) result = RT_TEST_FAILED;
        else { result = RT_OK_R; type = 1; }
    }
    else
    {
        bool ok = aerolb_crc_bits(dec, 8 * 6);
        if (ok)
        {
            const int numberofsus = 1 + (blockptr - (64 * 5)) / (64 * 3); // C division: 0 at two columns
            for (int i = 0; i < numberofsus && ok; i++) ok = aerolb_crc_bits(dec + (8 * 6) + (8 * 12) * i, 8 * 12);
        }
        if (!ok) result = (blockptr >= RT_BLOCKSZ) ? RT_BAD : RT_TEST_FAILED;
        else { result = RT_OK_T; type = 2; chop = 1; }
    }
    int ev_cnt = ALD(AI_EV_CNT), overflow = ALD(AI_OVERFLOW);
    if (type)
    {
        const int ninfo = nd / 8 - chop; // packintobytes (+ infofield.chop(1) for T packets)
        int row_cnt = ALD(AI_SU_CNT);
        const int npk = ALD(BI_NPACKETS);
        for (int c = 0; c * 12 < ninfo; c++)
        {
            if (row_cnt < g.su_cap)
            {
                int32_t *row = p.sus + ((size_t)ch * g.su_cap + row_cnt) * 16;
                row[0] = npk; row[1] = c;
                // the twelve loads first, then the stores: interleaved, every load waited for the store in front of it (vmcnt retires in order)
                unsigned long long w8[12];
#pragma unroll
                for (int j = 0; j < 12; j++) w8[j] = (c * 12 + j < ninfo) ? ((const unsigned long long *)dec)[c * 12 + j] : 0ull;
#pragma unroll
                for (int j = 0; j < 12; j++) row[2 + j] = (c * 12 + j < ninfo) ? (int)aerolb_pack8(w8[j]) : 0; // first bit of a byte is its LSB
                row[14] = ninfo; row[15] = type;
                row_cnt++;
            }
            else overflow |= 1;
        }
        ALD(AI_SU_CNT) = row_cnt; ALD(BI_NPACKETS) = npk + 1;
        ALD(BI_RT_BLOCKPTR) = RT_BLOCKSZ; // stop further testing
    }
    if (result == RT_BAD)
    {
        const long long bitidx = (((long long)(unsigned)ALD(AI_NBITS_LO)) | ((long long)ALD(AI_NBITS_HI) << 32)) + ALD(AI_IN_POS);
        aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 3, 0); // " Bad R/T Packet" (aerol.cpp:1531)
    }
    if (!keep_last) ALD(BI_RT_LAST) = result;
    ALD(AI_EV_CNT) = ev_cnt; ALD(AI_OVERFLOW) = overflow;
}
