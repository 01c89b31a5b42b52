// k_aerol.h -- the Aero-L bit pipeline around the Viterbi decoder, continuous P-channel path of
// AeroL::Decode(bits, soft=true) (JAERO/aerol.cpp:1124-2039) for 600 / 1200 / 10500 bps, one channel per lane:
//   k_aerol_bits : unique-word detection + I/Q ambiguity (PreambleDetector / PreambleDetectorPhaseInvariant, aerol.cpp:744-804),
//                  frame counter and header (:1274-1322), block fill straight into DEINTERLEAVED order (AeroLInterleaver::
//                  deinterleave_ba, :603-625, folded into the store address), runs until the channel has a full block
//   k_viterbi    : JConvolutionalCodec::Decode_Continuous for the channels that completed a block (k_viterbi.h)
//   k_aerol_post : DelayLine dl2 (:1560), AeroLScrambler (:1563), byte packing (:1566-1578), and at the end of a frame the CRC-16
//                  of every 12-byte signal unit with the data-carrier-detect bookkeeping (:1583-1600)
// The three run in rounds (at most one block per channel per round) because the reference finishes a block -- including its DCD
// update, which gates unique-word detection -- before it looks at the next soft bit.  Everything after the CRC check (message
// names, ISU/ACARS reassembly, plane database) is text / control plane and stays on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AEROL_UW 0xE15AE893u
enum
{
    AI_CNTR, AI_DATACD, AI_DCDCOUNT, AI_GOTSYNC_LAST, AI_REALIMAG, AI_BLOCKCNT, AI_MUW, AI_FRAMEINFO, AI_LASTFRAMEINFO,
    AI_PD_EXACT, AI_PD_IMAG, AI_PD_REAL, AI_INV_IMAG, AI_INV_REAL, AI_SCR_POS, AI_DL2_PTR, AI_NINFO, AI_NFRAMES,
    AI_IN_POS, AI_RESUME, AI_RESUME_GOTSYNC, AI_HAS_BLOCK, AI_VBLOCKS, AI_SU_CNT, AI_EV_CNT, AI_OVERFLOW, AI_NBITS_LO, AI_NBITS_HI,
    AI_NFIELDS
};
struct AGeom
{
    int nch, nchp, fb, oqpsk, N, blocksz, dl2_sz, NumberOfBits, BitsInHeader, TotalNumberOfBits, su_cap, ev_cap, info_cap;
};
struct APtrs
{
    int *I;              // [AI_NFIELDS][nchp]
    uint8_t *deint;      // [nchp][blocksz]   deinterleaved soft bytes of the block being filled
    uint8_t *vbits;      // [nchp][blocksz/2] decoded bits of the last completed block
    uint8_t *overlap;    // [nchp][64]        Decode_Continuous overlap state (k_viterbi convention)
    uint8_t *dl2;        // [nchp][dl2_sz]
    uint8_t *info;       // [nchp][info_cap]
    int32_t *sus;        // [nchp][su_cap][16]
    long long *events;   // [nchp][ev_cap][3]
    const uint8_t *scr;  // [5000] scrambler sequence
};
#define ALD(f) (p.I[(size_t)(f) * g.nchp + ch])

__device__ __forceinline__ void aerol_event(const AGeom &g, const APtrs &p, int ch, int &ev_cnt, int &overflow, long long idx, int kind, long long value)
{
    if (ev_cnt < g.ev_cap)
    {
        long long *e = p.events + ((size_t)ch * g.ev_cap + ev_cnt) * 3;
        e[0] = idx; e[1] = kind; e[2] = value;
        ev_cnt++;
    }
    else overflow |= 2;
}

__global__ void k_aerol_bits(const AGeom g, const APtrs p, const int16_t *__restrict__ soft, const int *__restrict__ counts, int stride)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= g.nch) return;
    int cntr = ALD(AI_CNTR), datacd = ALD(AI_DATACD), gotsync_last = ALD(AI_GOTSYNC_LAST), realimag = ALD(AI_REALIMAG);
    int blockcnt = ALD(AI_BLOCKCNT), muw = ALD(AI_MUW), ninfo = ALD(AI_NINFO);
    unsigned frameinfo = (unsigned)ALD(AI_FRAMEINFO), lastframeinfo = (unsigned)ALD(AI_LASTFRAMEINFO);
    unsigned pd_exact = (unsigned)ALD(AI_PD_EXACT), pd_imag = (unsigned)ALD(AI_PD_IMAG), pd_real = (unsigned)ALD(AI_PD_REAL);
    int inv_imag = ALD(AI_INV_IMAG), inv_real = ALD(AI_INV_REAL), scr_pos = ALD(AI_SCR_POS), dcdcount = ALD(AI_DCDCOUNT);
    int pos = ALD(AI_IN_POS), resume = ALD(AI_RESUME), ev_cnt = ALD(AI_EV_CNT), overflow = ALD(AI_OVERFLOW);
    const long long nbits0 = ((long long)(unsigned)ALD(AI_NBITS_LO)) | ((long long)ALD(AI_NBITS_HI) << 32);
    const int n = counts[ch];
    const int16_t *sb = soft + (size_t)ch * stride;
    uint8_t *deint = p.deint + (size_t)ch * g.blocksz;
    int has_block = 0;
    int gotsync = resume ? ALD(AI_RESUME_GOTSYNC) : 0;

    while (pos < n || resume)
    {
        const long long bitidx = nbits0 + pos;
        if (!resume)
        {
            const int v = sb[pos];
            if (v < 0) { muw = 0; pos++; continue; } // start-of-burst marker (aerol.cpp:1146-1152)
            int bit = (((unsigned)v & 0xFFu) >= 128u) ? 1 : 0;
            unsigned soft_bit = (unsigned)v & 0xFFFFu;
            if (muw < 100000) muw++;
            if (g.oqpsk)
            {
                realimag++; realimag %= 2;
                unsigned &pd = realimag ? pd_imag : pd_real;
                int &inverted = realimag ? inv_imag : inv_real;
                if (cntr > g.NumberOfBits - 68 || cntr <= 0 || !datacd)
                {
                    // PreambleDetectorPhaseInvariant::Update, tolerance 0 outside burst mode (aerol.cpp:781-804, 1009-1016)
                    pd = (pd << 1) | (unsigned)bit;
                    const int xorsum = __popc(pd ^ AEROL_UW);
                    gotsync = 0;
                    if (xorsum >= 32) { inverted = 1; gotsync = 1; }
                    else if (xorsum <= 0) { inverted = 0; gotsync = 1; }
                    if (!gotsync_last) { gotsync_last = gotsync; gotsync = 0; }
                    else gotsync_last = 0;
                }
                else { gotsync = 0; gotsync_last = 0; }
                if (inverted)
                {
                    bit = 1 - bit;
                    if (soft_bit > 128) soft_bit = 255 - soft_bit;
                    else if (soft_bit < 128) soft_bit = 255 - soft_bit;
                }
            }
            else
            {
                // PreambleDetector::Update (aerol.cpp:744-750): exact match, buffer cleared on a hit
                pd_exact = (pd_exact << 1) | (unsigned)bit;
                gotsync = 0;
                if (pd_exact == AEROL_UW) { pd_exact = 0; gotsync = 1; }
            }
            if (cntr < 1000000000) cntr++;
            if (cntr < 16)
            {
                if (cntr == 0) { frameinfo = (unsigned)bit; ninfo = 0; }
                else { frameinfo = ((frameinfo << 1) | (unsigned)bit) & 0xFFFFu; }
            }
            if (cntr == 15)
            {
                const unsigned tval = frameinfo;
                frameinfo = lastframeinfo;
                lastframeinfo = tval;
            }
            if (cntr >= 16)
            {
                if (cntr == 16) blockcnt = -1;
                int idx = (cntr - g.BitsInHeader) % g.blocksz;
                if (idx < 0) idx = 0;
                // deinterleave_ba: out[j*64 + i] = block[((i*27)%64)*N + j]  <=>  block[idx] goes to (idx%N)*64 + ((idx/N)*19)%64
                deint[(idx % g.N) * 64 + (((idx / g.N) * 19) & 63)] = (uint8_t)soft_bit;
                if (idx == g.blocksz - 1)
                {
                    blockcnt++;
                    has_block = 1;
                    resume = 1; // the rest of this soft bit (gotsync / frame-length handling) runs after the block's Viterbi + post pass
                    break;
                }
            }
        }
        resume = 0;
        if (gotsync)
        {
            if (cntr + 1 != g.TotalNumberOfBits) aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 1, cntr + 1);
            cntr = -1;
            datacd = 1; dcdcount = 12;
            aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 0, 1);
            aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 2, 0);
            scr_pos = 0;
        }
        if (cntr + 1 == g.TotalNumberOfBits) { scr_pos = 0; cntr = -1; }
        gotsync = 0;
        pos++;
    }
    ALD(AI_CNTR) = cntr; ALD(AI_DATACD) = datacd; ALD(AI_GOTSYNC_LAST) = gotsync_last; ALD(AI_REALIMAG) = realimag;
    ALD(AI_BLOCKCNT) = blockcnt; ALD(AI_MUW) = muw; ALD(AI_NINFO) = ninfo;
    ALD(AI_FRAMEINFO) = (int)frameinfo; ALD(AI_LASTFRAMEINFO) = (int)lastframeinfo;
    ALD(AI_PD_EXACT) = (int)pd_exact; ALD(AI_PD_IMAG) = (int)pd_imag; ALD(AI_PD_REAL) = (int)pd_real;
    ALD(AI_INV_IMAG) = inv_imag; ALD(AI_INV_REAL) = inv_real; ALD(AI_SCR_POS) = scr_pos; ALD(AI_DCDCOUNT) = dcdcount;
    ALD(AI_IN_POS) = pos; ALD(AI_RESUME) = resume; ALD(AI_RESUME_GOTSYNC) = gotsync; ALD(AI_HAS_BLOCK) = has_block;
    ALD(AI_EV_CNT) = ev_cnt; ALD(AI_OVERFLOW) = overflow;
}

__device__ __forceinline__ unsigned aerol_crc16(const uint8_t *bytes, int n) // AeroLcrc16::calcusingbytes (aerol.h:333-360)
{
    unsigned crc = 0xFFFFu;
    for (int i = 0; i < n; i++)
    {
        unsigned mb = bytes[i];
        for (int k = 0; k < 8; k++)
        {
            const unsigned message_bit = mb & 1u;
            mb >>= 1;
            const unsigned crc_bit = crc & 1u;
            crc >>= 1;
            if (crc_bit ^ message_bit) crc ^= 0x8408u;
        }
    }
    return (~crc) & 0xFFFFu;
}

__global__ void k_aerol_post(const AGeom g, const APtrs p)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= g.nch) return;
    if (!ALD(AI_HAS_BLOCK)) return;
    int dl2_ptr = ALD(AI_DL2_PTR), scr_pos = ALD(AI_SCR_POS), ninfo = ALD(AI_NINFO);
    const int cntr = ALD(AI_CNTR);
    uint8_t *dl2 = p.dl2 + (size_t)ch * g.dl2_sz;
    uint8_t *info = p.info + (size_t)ch * g.info_cap;
    const uint8_t *vb = p.vbits + (size_t)ch * (g.blocksz / 2);
    // Decode_Continuous returns blocksz/2 bits except for the first block of a stream, whose overlap buffer is still empty
    // (jconvolutionalcodec.cpp:194: mid(paddinglength+1, n/2) of the (n+24)/2 decoded bits)
    const int vblocks = ALD(AI_VBLOCKS);
    const int nb = vblocks ? g.blocksz / 2 : (g.blocksz + 24) / 2 - 25;
    ALD(AI_VBLOCKS) = vblocks + 1;
    int charptr = 0;
    unsigned chv = 0;
    for (int h = 0; h < nb; h++)
    {
        dl2[dl2_ptr] = vb[h];
        dl2_ptr++; if (dl2_ptr >= g.dl2_sz) dl2_ptr = 0;
        unsigned v = dl2[dl2_ptr];
        v ^= p.scr[scr_pos < 5000 ? scr_pos : 4999];
        scr_pos++;
        chv |= v * 128u;
        charptr++; charptr %= 8;
        if (charptr == 0) { if (ninfo < g.info_cap) info[ninfo++] = (uint8_t)chv; chv = 0; }
        else chv >>= 1;
    }
    ALD(AI_DL2_PTR) = dl2_ptr; ALD(AI_SCR_POS) = scr_pos; ALD(AI_NINFO) = ninfo;
    if ((cntr - g.BitsInHeader) == (g.NumberOfBits - 1))
    {
        int datacd = ALD(AI_DATACD), dcdcount = ALD(AI_DCDCOUNT), su_cnt = ALD(AI_SU_CNT), ev_cnt = ALD(AI_EV_CNT), overflow = ALD(AI_OVERFLOW);
        const int nframes = ALD(AI_NFRAMES);
        const int frameinfo = ALD(AI_FRAMEINFO);
        const long long bitidx = (((long long)(unsigned)ALD(AI_NBITS_LO)) | ((long long)ALD(AI_NBITS_HI) << 32)) + ALD(AI_IN_POS);
        for (int kk = 0; kk < ninfo / 12; kk++)
        {
            const uint8_t *su = info + kk * 12;
            unsigned crc_calc = aerol_crc16(su, 10);
            const unsigned crc_rec = ((unsigned)su[11] << 8) | su[10];
            if ((!crc_rec) && (crc_calc != crc_rec))
            {
                int tsum = 0;
                for (int ii = 0; ii < 10; ii++) tsum += su[ii];
                if (tsum == 0) crc_calc = 0; // some SUs are just zeros
            }
            if (crc_calc == crc_rec) { if (dcdcount < 12) dcdcount += 2; }
            else { if (dcdcount > 0) dcdcount -= 3; }
            if (!datacd && dcdcount > 2) { datacd = 1; aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 0, 1); }
            if (su_cnt < g.su_cap)
            {
                int32_t *row = p.sus + ((size_t)ch * g.su_cap + su_cnt) * 16;
                row[0] = nframes; row[1] = kk;
                for (int j = 0; j < 12; j++) row[2 + j] = su[j];
                row[14] = (crc_calc == crc_rec); row[15] = frameinfo;
                su_cnt++;
            }
            else overflow |= 1;
        }
        ALD(AI_DATACD) = datacd; ALD(AI_DCDCOUNT) = dcdcount; ALD(AI_SU_CNT) = su_cnt; ALD(AI_EV_CNT) = ev_cnt; ALD(AI_OVERFLOW) = overflow;
        ALD(AI_NFRAMES) = nframes + 1;
    }
}

// after a write: input positions back to 0, absolute bit counter advanced by this write's counts
__global__ void k_aerol_end_write(const AGeom g, const APtrs p, const int *__restrict__ counts)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= g.nch) return;
    long long nb = (((long long)(unsigned)ALD(AI_NBITS_LO)) | ((long long)ALD(AI_NBITS_HI) << 32)) + counts[ch];
    ALD(AI_NBITS_LO) = (int)(unsigned)(nb & 0xFFFFFFFFll); ALD(AI_NBITS_HI) = (int)(nb >> 32);
    ALD(AI_IN_POS) = 0;
}
