// aerolc.h -- Aero-L C-channel bit pipeline (8400 bps): AeroL::DecodeC (JAERO/aerol.cpp:2187-2502) for a bank of channels.
// SURVEY 8 row f4, second half.  Included by jaero_hip.hip after aerol_host.h; reached through jaero_aerol_create(fb = 8400).
//
// Written from the oracle restatement (oracle/aerol_oracle.c, c_write / c_frame_done), which is pinned against the unmodified AeroL.
// Golden + banks of 5 / 70 / 65 536 channels in both Viterbi layouts are green on the GPU (tests/test_gpu_aerol_c.py, test_gpu_scale_aerol.py);
// the bank logic also runs on the CPU (tests/host_emul/aerolc_emul.cpp).  One lane per channel walks the soft bits around the unique word;
// the body of a frame is jumped over and copied by a whole wavefront (k_aerolc_bulk, round 5).
//
// A frame = 104 unique-word bits (two 52-bit words, one per arm, OQPSKPreambleDetectorAndAmbiguityCorrection :811-900, tolerance 6)
// + 4096 channel bits = 16 interleaver blocks of 64 x 4 -> deinterleaved and depunctured (rate 3/4, every 4th coded bit an
// erasure, the last channel bit unused: 5460 soft symbols) -> continuous K=7 Viterbi (k_viterbi with the 62-byte overlap of
// JConvolutionalCodec::Decode_Continuous) -> 2714 bits -> 2708-bit delay line -> scrambler -> three 12-byte sub-band signal units
// (CRC-16) and 300 voice bytes.  As in the P-channel pipeline a write is processed in rounds: k_aerolc_bits walks every channel
// up to the end of its next frame (the soft bits land directly at their deinterleaved + depunctured positions), the Viterbi decodes
// the channels that completed one, k_aerolc_post finishes those frames, and the walk resumes.
#pragma once

#define CC_NSOFT 5460  // depunctured soft symbols per frame
#define CC_PITCH 5472  // distance between the rows of CcParams::dep: CC_NSOFT rounded up to 16 bytes, so that k_viterbi_lanes reads aligned 16-byte groups
#define CC_NBITS 2714  // decoded bits kept per frame
#define CC_PREV_PITCH 2720 // distance between the rows of CPtrs::dl2 (the previous frame's CC_NBITS decoded bits: what DelayLine(2714 - 6) holds), 16-byte aligned
#define CC_FRAME 4096  // channel bits per frame after the unique word
#define CC_MASK52 ((1ull << 52) - 1ull)
#define CC_UW1 216866263330005ull
#define CC_UW2 3012071630031408ull

enum
{
    CI_CNTR, CI_REALIMAG, CI_GSLAST, CI_INV_REAL, CI_INV_IMAG, CI_DATACD, CI_DCDCOUNT, CI_POS, CI_HAS_BLOCK, CI_NFRAMES,
    CI_SU_CNT, CI_V_CNT, CI_EV_CNT, CI_OVERFLOW, CI_DL2_PTR, CI_NBITS_LO, CI_NBITS_HI,
    // the stretches k_aerolc_bits jumped over in this round, for k_aerolc_bulk: input position of the first soft bit, its (post-increment) cntr,
    // length, and the arm / inversion state in front of it (bit 0: realimag, bit 1: inverted real arm, bit 2: inverted imaginary arm)
    CI_BULK_N, CI_BULK0_SRC, CI_BULK0_CNTR, CI_BULK0_LEN, CI_BULK0_FLAGS, CI_BULK1_SRC, CI_BULK1_CNTR, CI_BULK1_LEN, CI_BULK1_FLAGS,
    CI_NFIELDS
};
#define CC_MINRUN 32 // shorter stretches are walked

struct CGeom
{
    int nch, nchp, su_cap, v_cap, ev_cap;
};
struct CPtrs
{
    int *I;                      // [CI_NFIELDS][nchp]
    unsigned long long *B;       // [4][nchp] detector shift registers: real b1, real b2, imag b1, imag b2
    uint8_t *dep;                // [nchp][CC_PITCH] (CC_NSOFT used) deinterleaved + depunctured soft symbols of the frame being received
    uint8_t *vbits;              // [nchp][CC_NSOFT / 2] Viterbi output, one byte per bit
    uint8_t *overlap;            // [nchp][64] Decode_Continuous overlap (byte 62 = length)
    uint8_t *dl2;                // [nchp][CC_PREV_PITCH] delay line = the decoded bits of the frame finished before (k_aerolc_post)
    const uint8_t *scr;          // [5000] scrambler sequence
    const unsigned long long *scrf; // [25][2] the same by primary field: bit i of the pair y = scr[109 y + 1 + i], i < 108 (k_aerolc_post)
    int32_t *sus;                // [nchp][su_cap][16]  rows [frame, k, 12 bytes, crc_ok, 0]
    uint8_t *voice;              // [nchp][v_cap][304]  rows: uint32 frame, 300 voice bytes
    long long *events;           // [nchp][ev_cap][3]   rows [soft-bit index, kind (0 DCD, 2 sync), value]
};
#define CLD(f) p.I[(size_t)(f) * g.nchp + ch]

// OQPSKPreambleDetectorAndAmbiguityCorrection::Update (aerol.cpp:848-895) on 52-bit shift registers (oldest bit in bit 51)
__device__ __forceinline__ int cc_detect(unsigned long long &b1, unsigned long long &b2, int val, int &inverted)
{
    b1 = ((b1 << 1) | (unsigned long long)val) & CC_MASK52;
    int x = __popcll(b1 ^ CC_UW1);
    if (x >= 52 - 6) { inverted = 1; return 1; }
    if (x <= 6) { inverted = 0; return 1; }
    b2 = ((b2 << 1) | (unsigned long long)val) & CC_MASK52; // only reached when the first word did not match, as in the reference
    x = __popcll(b2 ^ CC_UW2);
    if (x >= 52 - 6) { inverted = 1; return 1; }
    if (x <= 6) { inverted = 0; return 1; }
    return 0;
}

__device__ __forceinline__ void cc_event(const CGeom &g, const CPtrs &p, int ch, int &ev_cnt, int &overflow, long long idx, int kind, long long value)
{
    if (ev_cnt < g.ev_cap)
    {
        long long *e = p.events + ((size_t)ch * g.ev_cap + ev_cnt) * 3;
        e[0] = idx; e[1] = kind; e[2] = value;
        ev_cnt++;
    }
    else overflow |= 2;
}

// received index cntr (post-increment, 0 .. CC_FRAME - 1) -> position in the deinterleaved + depunctured frame buffer, or -1 (the last
// source byte is never used, :2509): interleaver block / row / column -> deinterleaved source index -> depunctured position
__device__ __forceinline__ int cc_dep_index(int cntr)
{
    const int blk = cntr >> 8, r = cntr & 255;
    const int i = ((r >> 2) * 19) & 63; // inverse of the row permutation (i * 27) % 64
    const int src = blk * 256 + (r & 3) * 64 + i;
    return src < CC_FRAME - 1 ? src + src / 3 : -1;
}

// Lane = channel: DecodeC's loop over the soft bits (:2201-2316) up to the end of the next frame.  For the soft bits whose pre-increment
// cntr lies in [1, CC_FRAME - 112] DecodeC runs no unique-word detection: it toggles the arm, inverts by the arm's flag, counts and stores
// -- 3984 of a frame's 4200 soft bits.  Such a stretch is not walked: the lane notes it (at most two per round) and k_aerolc_bulk copies
// it with a whole wavefront, coalesced.  A third stretch in one round (two false unique words inside detection windows) ends the lane's
// round early; the host runs enough rounds for that (aerolc_write).
__global__ __launch_bounds__(64) void k_aerolc_bits(const CGeom g, const CPtrs p, const int16_t *__restrict__ soft, const int *__restrict__ counts, int stride)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= g.nch) return;
    int pos = CLD(CI_POS);
    const int n = counts[ch];
    int cntr = CLD(CI_CNTR), realimag = CLD(CI_REALIMAG), gslast = CLD(CI_GSLAST);
    int inv[2] = {CLD(CI_INV_REAL), CLD(CI_INV_IMAG)};
    int ev_cnt = CLD(CI_EV_CNT), overflow = CLD(CI_OVERFLOW);
    unsigned long long b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) b[k] = p.B[(size_t)k * g.nchp + ch];
    const long long base = ((long long)(unsigned)CLD(CI_NBITS_LO)) | ((long long)CLD(CI_NBITS_HI) << 32);
    const int16_t *s = soft + (size_t)ch * stride;
    uint8_t *dep = p.dep + (size_t)ch * CC_PITCH;
    int has = 0, nbulk = 0, yield = 0;
    while (pos < n && !has && !yield)
    {
        if (cntr >= 1 && cntr <= CC_FRAME - 112)
        {
            const int room = (CC_FRAME - 112) - cntr + 1, left = n - pos;
            const int len = room < left ? room : left;
            if (len >= CC_MINRUN)
            {
                if (nbulk == 2) { yield = 1; continue; }
                const int f0 = CI_BULK0_SRC + 4 * nbulk;
                p.I[(size_t)(f0 + 0) * g.nchp + ch] = pos;
                p.I[(size_t)(f0 + 1) * g.nchp + ch] = cntr + 1;
                p.I[(size_t)(f0 + 2) * g.nchp + ch] = len;
                p.I[(size_t)(f0 + 3) * g.nchp + ch] = (realimag & 1) | (inv[0] ? 2 : 0) | (inv[1] ? 4 : 0);
                nbulk++;
                pos += len; cntr += len; realimag = (realimag + len) & 1; gslast = 0;
                continue;
            }
        }
        // Eight soft bits per request (round 6): one 2-byte load per bit meant one wait per bit for everything in flight -- vmcnt retires in order, the
        // frame-buffer stores included (~2 us per walked bit; a frame has ~216 of them).  The bits are then taken one by one exactly as before; a lane
        // leaves the group early where the loop above would have done something else first (end of the write, a finished frame, a stretch to jump).
        short v8[8];
        if (n - pos >= 8) __builtin_memcpy(v8, s + pos, 16);
        else
        {
#pragma unroll
            for (int k = 0; k < 8; k++) v8[k] = (pos + k < n) ? s[pos + k] : (short)0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            if (k > 0)
            {
                if (pos >= n || has) break;
                if (cntr >= 1 && cntr <= CC_FRAME - 112)
                {
                    const int room = (CC_FRAME - 112) - cntr + 1, left = n - pos;
                    if ((room < left ? room : left) >= CC_MINRUN) break;
                }
            }
            const int sv = v8[k];
            int bit = (((unsigned char)sv) >= 128) ? 1 : 0;
            unsigned soft_bit = (unsigned)(unsigned short)sv;
            int gotsync = 0;
            realimag++; realimag %= 2;
            const int q = realimag ? 0 : 1; // realimag != 0: preambledetectorreal
            if (cntr > CC_FRAME - 112 || cntr <= 0)
            {
                gotsync = q == 0 ? cc_detect(b[0], b[1], bit, inv[0]) : cc_detect(b[2], b[3], bit, inv[1]);
                if (!gslast) { gslast = gotsync; gotsync = 0; }
                else gslast = 0;
            }
            else { gotsync = 0; gslast = 0; }
            if (inv[q])
            {
                bit = 1 - bit;
                if (soft_bit > 128) soft_bit = 255 - soft_bit;
                else if (soft_bit < 128) soft_bit = 255 - soft_bit;
            }
            if (gotsync)
            {
                cntr = -1; // index = -1, deleaveredBlock / depuncturedBlock emptied, scrambler reset: implicit (positions are absolute)
                cc_event(g, p, ch, ev_cnt, overflow, base + pos, 2, 1);
            }
            else
            {
                if (cntr < 1000000000) cntr++;
                if (cntr <= CC_FRAME - 1)
                {
                    const int di = cc_dep_index(cntr);
                    if (di >= 0) dep[di] = (uint8_t)soft_bit;
                }
                if (cntr == CC_FRAME - 1) has = 1;
            }
            pos++;
        }
    }
    CLD(CI_POS) = pos; CLD(CI_HAS_BLOCK) = has; CLD(CI_BULK_N) = nbulk;
    CLD(CI_CNTR) = cntr; CLD(CI_REALIMAG) = realimag; CLD(CI_GSLAST) = gslast;
    CLD(CI_INV_REAL) = inv[0]; CLD(CI_INV_IMAG) = inv[1];
    CLD(CI_EV_CNT) = ev_cnt; CLD(CI_OVERFLOW) = overflow;
#pragma unroll
    for (int k = 0; k < 4; k++) p.B[(size_t)k * g.nchp + ch] = b[k];
}

// depunctured position's source index (0 .. CC_FRAME - 2) -> received index cntr: the inverse of cc_dep_index's first half (19 * 27 = 1 mod 64)
__device__ __forceinline__ int cc_cntr_of_src(int src)
{
    const int w = src & 255;
    return (src & ~255) + ((((w & 63) * 27) & 63) << 2) + (w >> 6);
}

// One wavefront per channel: the (at most two) stretches k_aerolc_bits jumped over in this round, in stream order: a later stretch overwrites an
// earlier one's positions (a frame abandoned for a new unique word).  Both in ONE launch since round 6 (the second launch was ~65 000
// workgroups that returned at once, every round): the wavefront finishes stretch 0 -- its stores made visible to the wavefront and retired -- before
// it starts stretch 1.  The walked soft bits of the same round never share a position with them (their cntr values lie outside [2, CC_FRAME - 111]).
// Round 6, second half: driven by the OUTPUT.  Source order meant one byte store per soft bit, the 64 lanes of every store instruction in 64 different
// 64-byte lines of the frame buffer (the interleaver spreads neighbours 85 positions apart): 4096 partial-line writes per frame, 0.97 ms per 65 536-channel
// step.  Now the stretch is staged in LDS (16-byte loads, coalesced), and a lane forms four consecutive bytes of the frame buffer -- three soft bits
// looked up in LDS through the inverse permutation and the erasure byte every fourth position holds (128, never anything else: aerolc_create) -- and
// stores them as one word, 256 contiguous bytes per store instruction; words with a position outside the stretch (the 112 positions of the unique-word
// window, and stretches cut by a write boundary) store their bytes singly.
// phase: -1 = stage, then emit (the device); 0 / 1 = one of the two (tests/host_emul runs a workgroup's threads one after the other: every thread stages
// before any emits); konly likewise >= 0 only there.
// Four channels per workgroup, a wavefront each (65 536 one-wavefront workgroups cost ~130 us to dispatch even when every one of them returns at once, and two
// of a write's three rounds are like that); a wavefront only touches its own LDS row, whose accesses the LDS takes in program order: a compiler fence, no barrier.
#ifndef AEROLC_KERNELS_ONLY
#define CC_BULK_SYNC() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront")
#define CC_BULK_PASS 2048
#else
#define CC_BULK_PASS 4096 // (tests/host_emul: one pass, because there every thread stages before any emits)
#endif
__global__ __launch_bounds__(256) void k_aerolc_bulk(const CGeom g, const CPtrs p, const int16_t *__restrict__ soft, int stride, int konly, int phase)
{
    __shared__ __attribute__((aligned(16))) int16_t rows[4][CC_BULK_PASS + 8];
    const int ch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ch >= g.nch) return;
    int16_t *row = rows[threadIdx.x >> 6];
    const int nk = CLD(CI_BULK_N); // wave-uniform
    const int16_t *s = soft + (size_t)ch * stride;
    uint8_t *dep = p.dep + (size_t)ch * CC_PITCH;
    const int tid = threadIdx.x & 63;
    for (int k = 0; k < nk && k < 2; k++)
    {
        if (konly >= 0 && k != konly) continue;
        if (k > 0 && konly < 0)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
            CC_BULK_SYNC(); // every lane has read what it needs of stretch 0 from LDS
        }
        const int f0 = CI_BULK0_SRC + 4 * k;
        const int src_k = p.I[(size_t)(f0 + 0) * g.nchp + ch], c0_k = p.I[(size_t)(f0 + 1) * g.nchp + ch];
        const int len_k = p.I[(size_t)(f0 + 2) * g.nchp + ch], fl_k = p.I[(size_t)(f0 + 3) * g.nchp + ch];
        // In passes of CC_BULK_PASS received positions, cut at multiples of 256 (the interleaver's blocks): 4 KB of LDS per wavefront instead of 8, so that
        // twice as many of these short-lived wavefronts -- a few dependent round trips each -- are resident (the kernel is bound by their latency, not by bytes
        // or instructions).  A word whose three sources fall into two passes is written byte by byte by both.
        for (int pa = (c0_k > 0 ? c0_k : 0) / CC_BULK_PASS * CC_BULK_PASS; pa < c0_k + len_k; pa += CC_BULK_PASS)
        {
        const int c0 = c0_k > pa ? c0_k : pa;
        const int c1 = (c0_k + len_k) < (pa + CC_BULK_PASS) ? (c0_k + len_k) : (pa + CC_BULK_PASS);
        const int len = c1 - c0, src = src_k + (c0 - c0_k), fl = fl_k ^ ((c0 - c0_k) & 1);
        if (pa > (c0_k > 0 ? c0_k : 0) / CC_BULK_PASS * CC_BULK_PASS) CC_BULK_SYNC(); // the previous pass has read its entries
        if (phase != 1)
        {
            for (int j8 = tid * 8; j8 < len; j8 += 64 * 8)
            {
                if (j8 + 8 <= len) __builtin_memcpy(row + j8, s + src + j8, 16);
                else for (int j = j8; j < len; j++) row[j] = s[src + j];
            }
        }
        if (phase < 0) CC_BULK_SYNC();
        if (phase != 0)
        {
            // the words that can hold positions of [c0, c0 + len): the interleaver permutes inside blocks of 256 received bits = 256 source indices = words
            // 256 b / 3 .. (256 b + 255) / 3, so only the words of the blocks the stretch touches are looked at (a write boundary cuts a frame body into two
            // stretches, copied in two rounds: looking at all 1365 words in both made this kernel instruction-bound, 0.25 ms per round)
            const int m_lo = (((c0 > 0 ? c0 : 0) >> 8) << 8) / 3;
            int m_hi = ((((c0 + len - 1) >> 8) << 8) + 255) / 3 + 1;
            if (m_hi > CC_NSOFT / 4) m_hi = CC_NSOFT / 4;
            for (int m = m_lo + tid; m < m_hi; m += 64)
            {
                unsigned word = 128u << 24;
                int inside = 0;
                unsigned char bytes[3];
#pragma unroll
                for (int e = 0; e < 3; e++)
                {
                    const int j = cc_cntr_of_src(3 * m + e) - c0;
                    const bool in = j >= 0 && j < len;
                    unsigned soft_bit = 0;
                    if (in)
                    {
                        soft_bit = (unsigned)(unsigned short)row[j];
                        const int realimag = ((fl & 1) + j + 1) & 1;         // toggled before the bit is used (:2207)
                        const int inverted = realimag ? (fl & 2) : (fl & 4); // realimag != 0: the real arm's detector and flag
                        if (inverted) { if (soft_bit != 128) soft_bit = 255 - soft_bit; }
                        inside++;
                    }
                    bytes[e] = (unsigned char)soft_bit;
                    word |= (soft_bit & 255u) << (8 * e);
                }
                if (inside == 3) *(unsigned *)(dep + 4 * m) = word;
                else if (inside > 0)
                {
#pragma unroll
                    for (int e = 0; e < 3; e++)
                    {
                        const int j = cc_cntr_of_src(3 * m + e) - c0;
                        if (j >= 0 && j < len) dep[4 * m + e] = bytes[e];
                    }
                }
            }
        }
        } // passes
    }
}

// lane = channel: the end of a frame (:2318-2490) for the channels whose frame the Viterbi just decoded
__global__ __launch_bounds__(64) void k_aerolc_post(const CGeom g, const CPtrs p)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= g.nch) return;
    if (!CLD(CI_HAS_BLOCK)) return;
    const uint8_t *vb = p.vbits + (size_t)ch * (CC_NSOFT / 2);
    int datacd = CLD(CI_DATACD), dcdcount = CLD(CI_DCDCOUNT);
    int ev_cnt = CLD(CI_EV_CNT), overflow = CLD(CI_OVERFLOW), su_cnt = CLD(CI_SU_CNT), v_cnt = CLD(CI_V_CNT);
    const int nframes = CLD(CI_NFRAMES);
    const long long bitidx = (((long long)(unsigned)CLD(CI_NBITS_LO)) | ((long long)CLD(CI_NBITS_HI) << 32)) + CLD(CI_POS) - 1;
    uint8_t *vrow = nullptr;
    if (v_cnt < g.v_cap)
    {
        vrow = p.voice + ((size_t)ch * g.v_cap + v_cnt) * 304;
        *(unsigned *)vrow = (unsigned)nframes;
        v_cnt++;
    }
    else overflow |= 4;
    // The 2714 bits of a frame are 25 primary fields of 109: bit 0 unused, 96 voice bits = 12 bytes, 12 sub-band bits (fields 0..23; the 25th ends
    // after its sub-band bit 0, unused); a sub-band unit is 96 sub-band bits = fields 8 k .. 8 k + 7.  DecodeC goes bit by bit: delay line (:2330), scrambler
    // (:2333), then the two extractions (:2343-2358, :2454-2478), each a shift register with a modulo-8 counter.  Here one field per iteration, its structure
    // known at compile time -- the bit-by-bit form of this kernel, unrolled by 64 bits with every branch of the reference inside, was 150 KB of code
    // (the instruction cache two CUs share holds 64) and ran at ~600 cycles per bit.
    // The delay line (DelayLine(2714 - 6)) hands back, for bit h of a frame, the bit pushed 2708 steps earlier.  Only this kernel pushes, 2714 bits per
    // finished frame, so that bit is bit h + 6 of the frame finished before this one (h < 2708) or bit h - 2708 of this frame: the line IS the previous
    // frame's decoded bits.  Until round 6 it was a byte ring [slot][channel] with a byte load and a byte store per bit; now p.dl2 keeps the previous
    // frame's 2714 bytes per channel (positions the codec's first call does not produce stay what they were, as in vbits), read 16 bytes at a time and
    // replaced by this frame's after the last field.
    uint8_t *prev = p.dl2 + (size_t)ch * CC_PREV_PITCH;
    // four bytes that are 0 or 1 -> their four bits (byte 0 -> bit 0): the products' partial terms meet in no bit, so nothing carries into bits 28..31
    auto nib = [](unsigned w) __attribute__((always_inline)) { return (w * 0x10204080u) >> 28; };
    unsigned long long sb_lo = 0, sb_hi = 0; // the sub-band unit being collected, first bit in bit 0 of sb_lo
    int kk = 0;
#pragma unroll 1
    for (int y = 0; y < 25; y++)
    {
        const int hb = 109 * y;
        // d[i] = the delayed bits of this field from its bit 1 on (i = 0 .. 107), four to a word: frame bit hb + 1 + i <- previous frame's bit hb + 7 + i
        unsigned d[27];
#pragma unroll
        for (int k = 0; k < 6; k++) __builtin_memcpy(d + 4 * k, prev + hb + 7 + 16 * k, 16);
        __builtin_memcpy(d + 24, prev + hb + 7 + 96, 12); // (field 24: reads up to byte 2730 of a 2720-byte row; the rows are followed by 64 spare bytes)
        if (y == 24)
        {
            // frame bits 2708 .. 2712 (field bits 92 .. 96 = d bytes 91 .. 95) come from this frame's bits 0 .. 4
            unsigned char c5[8];
#pragma unroll
            for (int k = 0; k < 5; k++) c5[k] = vb[k];
            d[22] = (d[22] & 0x00FFFFFFu) | ((unsigned)c5[0] << 24);
            d[23] = (unsigned)c5[1] | ((unsigned)c5[2] << 8) | ((unsigned)c5[3] << 16) | ((unsigned)c5[4] << 24);
        }
        const unsigned long long s0 = p.scrf[2 * y], s1 = p.scrf[2 * y + 1]; // the scrambler's bits for field bits 1 .. 108, bit i of the pair = field bit 1 + i
        unsigned vw[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
        {
            unsigned w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) w |= (nib(d[8 * q + 2 * b]) | (nib(d[8 * q + 2 * b + 1]) << 4)) << (8 * b);
            const unsigned sw = (q == 0) ? (unsigned)s0 : (q == 1 ? (unsigned)(s0 >> 32) : (unsigned)s1);
            vw[q] = w ^ sw;
        }
        if (vrow)
        {
            unsigned *vp = (unsigned *)(vrow + 4 + 12 * y);
            vp[0] = vw[0]; vp[1] = vw[1]; vp[2] = vw[2];
        }
        if (y < 24)
        {
            const unsigned sb12 = (nib(d[24]) | (nib(d[25]) << 4) | (nib(d[26]) << 8)) ^ ((unsigned)(s1 >> 32) & 0xFFFu);
            const int pos = 12 * (y & 7); // wave-uniform
            if (pos < 64) sb_lo |= (unsigned long long)sb12 << pos;
            if (pos > 52) sb_hi |= pos >= 64 ? ((unsigned long long)sb12 << (pos - 64)) : ((unsigned long long)sb12 >> (64 - pos));
            if ((y & 7) == 7)
            {
                unsigned char info[12];
#pragma unroll
                for (int j = 0; j < 8; j++) info[j] = (unsigned char)(sb_lo >> (8 * j));
#pragma unroll
                for (int j = 0; j < 4; j++) info[8 + j] = (unsigned char)(sb_hi >> (8 * j));
                const unsigned crc_calc = aerol_crc16(info, 10);
                const unsigned crc_rec = ((unsigned)info[11] << 8) | info[10];
                if (crc_calc == crc_rec) { if (dcdcount < 12) dcdcount += 2; }
                else { if (dcdcount > 0) dcdcount -= 5; }
                if (!datacd && dcdcount > 2) { datacd = 1; cc_event(g, p, ch, ev_cnt, overflow, bitidx, 0, 1); }
                if (su_cnt < g.su_cap)
                {
                    int32_t *row = p.sus + ((size_t)ch * g.su_cap + su_cnt) * 16;
                    row[0] = nframes; row[1] = kk;
                    for (int j = 0; j < 12; j++) row[2 + j] = info[j];
                    row[14] = (crc_calc == crc_rec); row[15] = 0;
                    su_cnt++;
                }
                else overflow |= 1;
                kk++;
                sb_lo = 0; sb_hi = 0;
            }
        }
    }
    // this frame's bits become the delay line's content
#pragma unroll 2
    for (int k = 0; k < CC_PREV_PITCH; k += 80)
    {
        unsigned long long t[10];
#pragma unroll
        for (int q = 0; q < 5; q++) __builtin_memcpy(&t[2 * q], vb + k + 16 * q, 16);
#pragma unroll
        for (int q = 0; q < 5; q++) __builtin_memcpy(prev + k + 16 * q, &t[2 * q], 16);
    }
    CLD(CI_DATACD) = datacd; CLD(CI_DCDCOUNT) = dcdcount;
    CLD(CI_EV_CNT) = ev_cnt; CLD(CI_OVERFLOW) = overflow; CLD(CI_SU_CNT) = su_cnt; CLD(CI_V_CNT) = v_cnt;
    CLD(CI_NFRAMES) = nframes + 1;
    CLD(CI_HAS_BLOCK) = 0;
}

__global__ void k_aerolc_end_write(const CGeom g, const CPtrs p, const int *__restrict__ counts)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= g.nch) return;
    long long nb = (((long long)(unsigned)CLD(CI_NBITS_LO)) | ((long long)CLD(CI_NBITS_HI) << 32)) + counts[ch];
    CLD(CI_NBITS_LO) = (int)(unsigned)(nb & 0xFFFFFFFFll); CLD(CI_NBITS_HI) = (int)(nb >> 32);
    CLD(CI_POS) = 0;
}

// AeroL::updateDCD (aerol.cpp:1109-1122)
__global__ void k_aerolc_tick_dcd(const CGeom g, const CPtrs p, int *dcd_out)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= g.nch) return;
    int dcdcount = CLD(CI_DCDCOUNT), datacd = CLD(CI_DATACD), ev_cnt = CLD(CI_EV_CNT), overflow = CLD(CI_OVERFLOW);
    if (dcdcount > 0) dcdcount -= 3;
    else { if (dcdcount < 0) dcdcount = 0; }
    if (datacd && !dcdcount)
    {
        datacd = 0;
        const long long bitidx = ((long long)(unsigned)CLD(CI_NBITS_LO)) | ((long long)CLD(CI_NBITS_HI) << 32);
        cc_event(g, p, ch, ev_cnt, overflow, bitidx, 0, 0);
    }
    CLD(CI_DCDCOUNT) = dcdcount; CLD(CI_DATACD) = datacd; CLD(CI_EV_CNT) = ev_cnt; CLD(CI_OVERFLOW) = overflow;
    if (dcd_out) dcd_out[ch] = datacd;
}

#ifndef AEROLC_KERNELS_ONLY // tests/host_emul/aerolc_emul.cpp compiles the kernels above as host functions
// ------------------------------------------------------------------------------------------------ host side
struct aerolc_state
{
    CGeom g{};
    CPtrs p{};
    unsigned long long *d_vhist = nullptr; // k_viterbi_lanes history scratch (banks large enough for the lane layout)
};

static int aerolc_create(jaero_aerol_ctx *c, int nchannels, int su_capacity)
{
    aerolc_state *cs = new aerolc_state();
    c->cmode = cs;
    CGeom &g = cs->g;
    g.nch = nchannels; g.nchp = (nchannels + 63) / 64 * 64;
    g.su_cap = su_capacity > 0 ? su_capacity : 3 * 64; // 64 frames between reads
    g.v_cap = (g.su_cap + 2) / 3;
    g.ev_cap = 256;
    int rc;
#define CA(ptr, count) do { if ((rc = aalloc(c, &(ptr), (size_t)(count)))) return rc; } while (0)
    CA(cs->p.I, (size_t)CI_NFIELDS * g.nchp);
    CA(cs->p.B, (size_t)4 * g.nchp);
    CA(cs->p.dep, (size_t)g.nchp * CC_PITCH);
    CA(cs->p.vbits, (size_t)g.nchp * (CC_NSOFT / 2));
    CA(cs->p.overlap, (size_t)g.nchp * 64);
    CA(cs->p.dl2, (size_t)CC_PREV_PITCH * g.nchp + 64);
    CA(cs->p.sus, (size_t)g.nchp * g.su_cap * 16);
    CA(cs->p.voice, (size_t)g.nchp * g.v_cap * 304);
    CA(cs->p.events, (size_t)g.nchp * g.ev_cap * 3);
    if (viterbi_use_lanes(g.nch, CC_NSOFT, 24)) CA(cs->d_vhist, viterbi_hist_bytes(g.nch) / sizeof(unsigned long long));
    uint8_t *d_scr = nullptr;
    CA(d_scr, 5000);
    unsigned long long *d_scrf = nullptr;
    CA(d_scrf, 50);
#undef CA
    cs->p.scr = d_scr;
    {
        std::vector<uint8_t> scr(5000);
        int state[15] = {1, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1};
        for (int k = 0; k < 5000; k++)
        {
            const int val0 = state[0] ^ state[14];
            scr[k] = (uint8_t)val0;
            for (int i = 14; i > 0; i--) state[i] = state[i - 1];
            state[0] = val0;
        }
        HIPCHK(hipMemcpy(d_scr, scr.data(), 5000, hipMemcpyHostToDevice));
        std::vector<unsigned long long> scrf(50, 0ull);
        for (int y = 0; y < 25; y++)
            for (int i = 0; i < 108; i++) scrf[2 * y + i / 64] |= (unsigned long long)(scr[109 * y + 1 + i] & 1) << (i % 64);
        HIPCHK(hipMemcpy(d_scrf, scrf.data(), scrf.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        cs->p.scrf = d_scrf;
        // the depunctured buffer: every 4th symbol an erasure, for good (the walk only writes the other three)
        std::vector<uint8_t> dep((size_t)g.nchp * CC_PITCH, 0);
        for (size_t k = 0; k < dep.size(); k++) if ((k % CC_PITCH) % 4 == 3) dep[k] = 128;
        HIPCHK(hipMemcpy(cs->p.dep, dep.data(), dep.size(), hipMemcpyHostToDevice));
        std::vector<int> I((size_t)CI_NFIELDS * g.nchp, 0);
        for (int ch = 0; ch < g.nchp; ch++)
        {
            I[(size_t)CI_CNTR * g.nchp + ch] = 1000000000; // AeroL constructor (aerol.cpp:907)
            I[(size_t)CI_EV_CNT * g.nchp + ch] = 1;        // row 0 = [0, DCD, 0]: DataCarrierDetect(false) emitted by the constructor
        }
        HIPCHK(hipMemcpy(cs->p.I, I.data(), I.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    return 0;
}

static int aerolc_write(jaero_aerol_ctx *c, const int16_t *dsoft, const int *dcounts, int stride, int max_count, hipStream_t st)
{
    aerolc_state *cs = (aerolc_state *)c->cmode;
    const CGeom &g = cs->g;
    // a round finishes at most one frame per channel.  Frame ends are at least 4098 soft bits apart: the detector window reopens at
    // cntr > CC_FRAME - 112, so a (false or real) unique word can fire two bits after a completed frame and the next frame ends
    // CC_FRAME bits after that -- not CC_FRAME + 104 as in a clean stream.
    // A lane's round also ends when it meets a third jumpable stretch (k_aerolc_bits): it has then consumed at least one whole frame body
    // (CC_FRAME - 112 soft bits), so the bound below covers that too.
    const int rounds = max_count / (CC_FRAME - 112) + 2;
    const int *valid = cs->p.I + (size_t)CI_HAS_BLOCK * g.nchp;
    const dim3 grid(g.nchp / 64), block(64);
    for (int r = 0; r < rounds; r++)
    {
        aprof_begin(c, 0, st);
        hipLaunchKernelGGL(k_aerolc_bits, grid, block, 0, st, g, cs->p, dsoft, dcounts, stride);
        hipLaunchKernelGGL(k_aerolc_bulk, dim3((g.nch + 3) / 4), dim3(256), 0, st, g, cs->p, dsoft, stride, -1, -1); // both stretches of a round, in order
        aprof_end(c, st);
        aprof_begin(c, 1, st);
        // one block per wavefront for small banks, one per lane (k_viterbi_lanes) from 16 384 channels on, as the P-channel pipeline
        viterbi_launch(st, (const uint8_t *)cs->p.dep, CC_NSOFT, (const uint8_t *)cs->p.overlap, 24, cs->p.vbits, CC_NSOFT / 2, 25, CC_NSOFT / 2, g.nch, valid,
                       cs->d_vhist, 0, 0, 0, CC_PITCH);
        hipLaunchKernelGGL(k_viterbi_overlap_update, dim3(g.nch), dim3(64), 0, st, (const uint8_t *)cs->p.dep, CC_NSOFT, cs->p.overlap, g.nch, valid, 0, CC_PITCH);
        aprof_end(c, st);
        aprof_begin(c, 2, st);
        hipLaunchKernelGGL(k_aerolc_post, grid, block, 0, st, g, cs->p);
        aprof_end(c, st);
    }
    hipLaunchKernelGGL(k_aerolc_end_write, grid, block, 0, st, g, cs->p, dcounts);
    HIPCHK(hipGetLastError());
    return 0;
}

// drains rows of one channel: copies min(count, caprows) rows, keeps the rest (as aerol_read_rows does for the other modes)
static int aerolc_read(jaero_aerol_ctx *c, int ch, void *rows, int caprows, int *nrows, int cnt_field, const void *base, int cap, size_t rowbytes, int ovbit)
{
    aerolc_state *cs = (aerolc_state *)c->cmode;
    const CGeom &g = cs->g;
    if (ch < 0 || ch >= g.nch || caprows < 0 || !nrows) return fail(JAERO_EINVAL, "jaero_aerol read: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    int cnt = 0;
    int *dcnt = cs->p.I + (size_t)cnt_field * g.nchp + ch;
    HIPCHK(hipMemcpy(&cnt, dcnt, sizeof(int), hipMemcpyDeviceToHost));
    const int take = cnt < caprows ? cnt : caprows;
    const char *src = (const char *)base + (size_t)ch * cap * rowbytes;
    if (take > 0 && rows) HIPCHK(hipMemcpy(rows, src, rowbytes * take, hipMemcpyDeviceToHost));
    const int left = cnt - take;
    if (left > 0)
    {
        std::vector<char> tmp(rowbytes * left);
        HIPCHK(hipMemcpy(tmp.data(), src + rowbytes * take, tmp.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy((void *)src, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(dcnt, &left, sizeof(int), hipMemcpyHostToDevice));
    *nrows = take;
    // rows the kernels had to drop because the caller fell behind (CI_OVERFLOW: 1 signal units, 2 events, 4 voice frames):
    // reported once, then cleared, as aerol_read_rows does for the P and R/T banks
    int ov = 0;
    int *dov = cs->p.I + (size_t)CI_OVERFLOW * g.nchp + ch;
    HIPCHK(hipMemcpy(&ov, dov, sizeof(int), hipMemcpyDeviceToHost));
    if (ov & ovbit)
    {
        const int z = ov & ~ovbit;
        HIPCHK(hipMemcpy(dov, &z, sizeof(int), hipMemcpyHostToDevice));
        return fail(JAERO_EOVERFLOW, "Aero-L C-channel %d overflowed an output buffer (flag %d); rows were dropped", ch, ovbit);
    }
    return 0;
}

extern "C" int jaero_aerol_read_voice(jaero_aerol_ctx *c, int ch, uint8_t *rows, int caprows, int *nrows)
{
    if (!c || !c->cmode) return fail(JAERO_EINVAL, "jaero_aerol_read_voice: not a C-channel (fb = 8400) bank");
    aerolc_state *cs = (aerolc_state *)c->cmode;
    return aerolc_read(c, ch, rows, caprows, nrows, CI_V_CNT, cs->p.voice, cs->g.v_cap, 304, 4);
}
#endif // AEROLC_KERNELS_ONLY
