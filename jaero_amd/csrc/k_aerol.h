// k_aerol.h -- the Aero-L bit pipeline around the Viterbi decoder, continuous P-channel path of
// AeroL::Decode(bits, soft=true) (JAERO/aerol.cpp:1124-2039) for 600 / 1200 / 10500 bps, one channel per lane:
//   k_aerol_scan : once per write, does a channel's input hold a start-of-burst marker (those channels go bit by bit throughout)
//   k_aerol_bits : unique-word detection + I/Q ambiguity (PreambleDetector / PreambleDetectorPhaseInvariant, aerol.cpp:744-804),
//                  frame counter and header (:1274-1322), block fill in received order; runs until the channel has a full block;
//                  jumps over the body of a frame while the channel is locked
//   k_aerol_bulk : copies the jumped-over soft bits into the block, one wavefront per channel
//   k_aerol_deint: AeroLInterleaver::deinterleave_ba (:603-625) of the completed blocks, one wavefront per channel through LDS
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
    AI_IN_POS, AI_RESUME, AI_RESUME_GOTSYNC, AI_HAS_BLOCK, AI_VBLOCKS, AI_SU_CNT, AI_EV_CNT, AI_OVERFLOW, AI_NBITS_LO, AI_NBITS_HI, AI_ACC, AI_ACCBAD,
    AI_MARKER, AI_BULK_LEN, AI_BULK_SRC, AI_BULK_DST, AI_BULK_FLAGS, AI_BULK2_LEN, AI_BULK2_SRC, AI_BULK2_DST, AI_BULK2_FLAGS, AI_NMATCH, AI_MATCH0, AI_MATCH1, AI_MATCH2, AI_MATCH3,
    AI_NFIELDS
};
struct AGeom
{
    int nch, nchp, fb, oqpsk, N, blocksz, dl2_sz, NumberOfBits, BitsInHeader, TotalNumberOfBits, su_cap, ev_cap, info_cap, idx_sat, tiled, burst, dl2_words, packed;
};
struct APtrs
{
    int *I;              // [AI_NFIELDS][nchp]
    uint8_t *rx;         // [nchp][blocksz]   soft bytes of the block being filled, received order
    uint8_t *deint;      // [nchp][blocksz]   the last completed block, deinterleaved
    uint8_t *vbits;      // [nchp][blocksz/2] decoded bits of the last completed block
    uint8_t *overlap;    // [nchp][64]        Decode_Continuous overlap state (k_viterbi convention)
    uint8_t *dl2;        // [nchp][dl2_sz]
    uint8_t *info;       // [nchp][info_cap]
    int32_t *sus;        // [nchp][su_cap][16]
    long long *events;   // [nchp][ev_cap][3]
    const uint8_t *scr;  // [5000] scrambler sequence
    const unsigned *scrw; // the same, 32 per word (bit k of the sequence = bit k&31 of word k>>5), one spare word
    unsigned *dl2w;      // [nchp][dl2_words] the delay line as a bit ring (large banks: k_aerol_post_packed)
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

// One soft bit of AeroL::Decode for one channel, in two parts: everything up to and including the block store (part A; returns true
// when the store completed an interleaver block -- the rest of this soft bit then waits for the block's Viterbi + post pass), and
// the unique-word / frame-length handling after it (part B).
struct ABitState
{
    int cntr, datacd, gotsync_last, realimag, blockcnt, muw, ninfo, inv_imag, inv_real, scr_pos, dcdcount, ev_cnt, overflow;
    unsigned frameinfo, lastframeinfo, pd_exact, pd_imag, pd_real, acc;
    int accbad;
};
__device__ __forceinline__ bool aerol_bit_a(const AGeom &g, ABitState &s, int v, int &gotsync, uint8_t *rx)
{
    int bit = (((unsigned)v & 0xFFu) >= 128u) ? 1 : 0;
    unsigned soft_bit = (unsigned)v & 0xFFFFu;
    if (s.muw < 100000) s.muw++;
    if (g.oqpsk)
    {
        s.realimag ^= 1;
        unsigned pd = s.realimag ? s.pd_imag : s.pd_real;
        int inverted = s.realimag ? s.inv_imag : s.inv_real;
        if (s.cntr > g.NumberOfBits - 68 || s.cntr <= 0 || !s.datacd)
        {
            // PreambleDetectorPhaseInvariant::Update, tolerance 0 outside burst mode (aerol.cpp:781-804, 1009-1016)
            pd = (pd << 1) | (unsigned)bit;
            const int xorsum = __popc(pd ^ AEROL_UW);
            gotsync = 0;
            if (xorsum >= 32) { inverted = 1; gotsync = 1; }
            else if (xorsum <= 0) { inverted = 0; gotsync = 1; }
            if (!s.gotsync_last) { s.gotsync_last = gotsync; gotsync = 0; }
            else s.gotsync_last = 0;
        }
        else { gotsync = 0; s.gotsync_last = 0; }
        if (s.realimag) { s.pd_imag = pd; s.inv_imag = inverted; }
        else { s.pd_real = pd; s.inv_real = inverted; }
        if (inverted)
        {
            bit = 1 - bit;
            if (soft_bit != 128u) soft_bit = 255u - soft_bit;
        }
    }
    else
    {
        // PreambleDetector::Update (aerol.cpp:744-750): exact match, buffer cleared on a hit
        s.pd_exact = (s.pd_exact << 1) | (unsigned)bit;
        gotsync = 0;
        if (s.pd_exact == AEROL_UW) { s.pd_exact = 0; gotsync = 1; }
    }
    if (s.cntr < 1000000000) s.cntr++;
    if (s.cntr < 16)
    {
        if (s.cntr == 0) { s.frameinfo = (unsigned)bit; s.ninfo = 0; }
        else { s.frameinfo = ((s.frameinfo << 1) | (unsigned)bit) & 0xFFFFu; }
    }
    if (s.cntr == 15)
    {
        const unsigned tval = s.frameinfo;
        s.frameinfo = s.lastframeinfo;
        s.lastframeinfo = tval;
    }
    if (s.cntr >= 16)
    {
        if (s.cntr == 16) s.blockcnt = -1;
        // idx = (cntr - BitsInHeader) % blocksz, clamped at 0 (aerol.cpp:1330-1332).  cntr never exceeds TotalNumberOfBits except for
        // its saturated start value, whose remainder the host precomputed.
        int idx;
        if (s.cntr >= 1000000000) idx = g.idx_sat;
        else
        {
            idx = s.cntr - g.BitsInHeader;
            if (idx < 0) idx = 0;
            if (idx >= g.blocksz) idx -= g.blocksz;
            if (idx >= g.blocksz) idx -= g.blocksz;
            if (idx >= g.blocksz) idx -= g.blocksz;
        }
        // the block is kept in RECEIVED order, four soft bytes per store (idx runs 0, 1, 2, .. within a block, so the three bytes
        // before an idx with (idx & 3) == 3 are the block's idx-3 .. idx-1); k_aerol_deint reorders completed blocks
        s.acc = (s.acc >> 8) | (soft_bit << 24);
        if ((idx & 3) == 0) s.accbad = 0;                     // a fresh group of four starts here
        if (s.accbad) rx[idx] = (uint8_t)soft_bit;            // group entered in the middle (after a bulk run): byte by byte
        else if ((idx & 3) == 3) *(unsigned *)(rx + (idx - 3)) = s.acc;
        if (idx == g.blocksz - 1)
        {
            s.blockcnt++;
            return true;
        }
    }
    return false;
}
__device__ __forceinline__ void aerol_bit_b(const AGeom &g, const APtrs &p, int ch, ABitState &s, int &gotsync, long long bitidx)
{
    if (gotsync)
    {
        if (s.cntr + 1 != g.TotalNumberOfBits) aerol_event(g, p, ch, s.ev_cnt, s.overflow, bitidx, 1, s.cntr + 1);
        s.cntr = -1;
        s.datacd = 1; s.dcdcount = 12;
        aerol_event(g, p, ch, s.ev_cnt, s.overflow, bitidx, 0, 1);
        aerol_event(g, p, ch, s.ev_cnt, s.overflow, bitidx, 2, 0);
        s.scr_pos = 0;
    }
    if (s.cntr + 1 == g.TotalNumberOfBits) { s.scr_pos = 0; s.cntr = -1; }
    gotsync = 0;
}

// Lane = channel, every lane at its own input position.  A locked 10.5 kbps channel (data carrier detected, no start-of-burst
// marker in this write -- k_aerol_scan) spends 4909 of the 5250 soft bits of a frame where AeroL::Decode does nothing but count,
// toggle the I/Q arm and copy the (possibly inverted) soft bit into the block: pre-increment cntr in [16, NumberOfBits-68]
// (aerol.cpp:1256-1345: no unique-word detection, header done, block not complete).  Such a stretch is not walked: the lane jumps
// over it and leaves a descriptor for k_aerol_bulk, which copies it with a whole wavefront.  Everything else (unique-word windows,
// headers, unlocked channels, 600/1200 bps, writes with markers) goes bit by bit.
#define AEROL_MINRUN 32
#ifndef AEROL_WINDOW_JUMP
#define AEROL_WINDOW_JUMP 1
#endif
#ifndef AEROL_WINDOW_FLUSH
#define AEROL_WINDOW_FLUSH 64 // bits of a detector window walked before its stretch may be jumped (the shift registers' 32 bits per arm)
#endif
#define AEROL_NMATCH 4 // potential unique-word positions k_aerol_scan records per channel and write
template <bool BULK>
__global__ __launch_bounds__(64) void k_aerol_bits(const AGeom g, const APtrs p, const int16_t *__restrict__ soft, const int *__restrict__ counts, int stride)
{
    const int lane = threadIdx.x;
    const int ch0 = blockIdx.x * 64 + lane;
    const bool valid = ch0 < g.nch;
    const int ch = valid ? ch0 : 0;
    ABitState s;
    s.cntr = ALD(AI_CNTR); s.datacd = ALD(AI_DATACD); s.gotsync_last = ALD(AI_GOTSYNC_LAST); s.realimag = ALD(AI_REALIMAG);
    s.blockcnt = ALD(AI_BLOCKCNT); s.muw = ALD(AI_MUW); s.ninfo = ALD(AI_NINFO);
    s.frameinfo = (unsigned)ALD(AI_FRAMEINFO); s.lastframeinfo = (unsigned)ALD(AI_LASTFRAMEINFO);
    s.pd_exact = (unsigned)ALD(AI_PD_EXACT); s.pd_imag = (unsigned)ALD(AI_PD_IMAG); s.pd_real = (unsigned)ALD(AI_PD_REAL);
    s.inv_imag = ALD(AI_INV_IMAG); s.inv_real = ALD(AI_INV_REAL); s.scr_pos = ALD(AI_SCR_POS); s.dcdcount = ALD(AI_DCDCOUNT);
    s.ev_cnt = ALD(AI_EV_CNT); s.overflow = ALD(AI_OVERFLOW); s.acc = (unsigned)ALD(AI_ACC); s.accbad = ALD(AI_ACCBAD);
    int pos = ALD(AI_IN_POS), resume = ALD(AI_RESUME);
    const long long nbits0 = ((long long)(unsigned)ALD(AI_NBITS_LO)) | ((long long)ALD(AI_NBITS_HI) << 32);
    const int n = valid ? counts[ch] : 0;
    const int16_t *sb = soft + (size_t)ch * stride;
    uint8_t *rx = p.rx + (size_t)ch * g.blocksz;
    const bool may_bulk = BULK && !ALD(AI_MARKER);
    int has_block = 0, gotsync = 0, bulk_len = 0, bulk_src = 0, bulk_dst = 0, bulk_flags = 0;
    int bulk2_len = 0, bulk2_src = 0, bulk2_dst = 0, bulk2_flags = 0, njump = 0; // a round has TWO descriptor slots (round 6, below)
    if (resume && valid)
    {
        // second half of the soft bit whose block store completed a block in the previous round
        gotsync = ALD(AI_RESUME_GOTSYNC);
        aerol_bit_b(g, p, ch, s, gotsync, nbits0 + pos);
        pos++;
        resume = 0;
    }
    bool live = valid && pos < n;
    const int zone_hi = g.NumberOfBits - 68; // last pre-increment cntr without unique-word detection
    const int block_last = g.BitsInHeader + g.blocksz - 2; // pre-increment cntr of the bit that completes the (only) block of a frame
    const int scan_hi = 32768; // k_aerol_scan covers this many positions of a write
    // first position >= pos at which the detector of an unlocked channel could fire; -1: none in this write
    const int nmatch = BULK ? ALD(AI_NMATCH) : 0;
    int match[AEROL_NMATCH];
#pragma unroll
    for (int k = 0; k < AEROL_NMATCH; k++) match[k] = (BULK && k < nmatch) ? p.I[(size_t)(AI_MATCH0 + k) * g.nchp + ch] : 0x7fffffff;
    int next_match = -1;
    // Soft bits eight per request (round 6): with one 2-byte load per walked bit -- even requested a bit ahead -- every bit waited for everything in
    // flight, block stores included (vmcnt retires in order): ~2 us per walked bit.  The eight values sit in two 64-bit registers; a bit is a shift.
    unsigned long long sw_lo = 0, sw_hi = 0;
    int sw_pos = -0x40000000; // position of the first of the eight
    auto soft_at = [&](int q) __attribute__((always_inline)) -> int {
        if (q < sw_pos || q >= sw_pos + 8)
        {
            sw_pos = q;
            if (q + 8 <= stride) { unsigned long long t[2]; __builtin_memcpy(t, sb + q, 16); sw_lo = t[0]; sw_hi = t[1]; }
            else
            {
                sw_lo = 0; sw_hi = 0;
                for (int k = 0; k < 8 && q + k < stride; k++)
                {
                    const unsigned long long v = (unsigned long long)(unsigned short)sb[q + k];
                    if (k < 4) sw_lo |= v << (16 * k); else sw_hi |= v << (16 * (k - 4));
                }
            }
        }
        const int k = q - sw_pos;
        return (int)(short)(unsigned short)((k < 4 ? sw_lo : sw_hi) >> (16 * (k & 3)));
    };
    while (__any(live))
    {
        if (live)
        {
            // next potential unique-word position at or after pos (beyond the AEROL_NMATCH-th known one: walk bit by bit)
            next_match = -1;
            {
                bool found = false;
#pragma unroll
                for (int k = 0; k < AEROL_NMATCH; k++)
                    if (!found && match[k] != 0x7fffffff && match[k] >= pos) { next_match = match[k]; found = true; }
                if (!found && nmatch > AEROL_NMATCH) next_match = pos; // more hits than recorded: no jumps past the last known one
            }
            bool jumped = false;
            // Two kinds of stretch in which AeroL::Decode only counts, toggles the arm and copies the soft bit:
            //  locked   (data carrier): pre-increment cntr in [16, NumberOfBits-68] -- no unique-word detection, header done
            //  unlocked (no carrier, detector on every bit): up to the bit before the next position at which the detector could
            //           fire (k_aerol_scan) and before the bit that completes the block; cntr >= 16 (or still saturated)
            // Round 6: a LOCKED channel is walked bit by bit for 341 of a frame's 5250 soft bits -- the detector is on from pre-increment cntr
            // NumberOfBits - 67 = 4925 on, 325 bits before the frame ends, because NumberOfBits (4992) counts the block's bits and cntr also the
            // header's and the dummy bits' 194 (aerol.cpp:1256-1264).  While the detector is on a locked channel does exactly what an unlocked one
            // does, so the unlocked jump applies -- once the detector's two shift registers hold nothing but bits of this window: they are not
            // shifted while the detector is off, so its first 64 decisions (32 per arm) see stale bits mixed with fresh ones and are walked as the
            // reference takes them; k_aerol_scan's positions assume registers that have seen the last 32 bits of their arm.  Walked per frame:
            // those 64, the bit that completes the block, the unique word's 64, the header's 16.
            int L = 0;
            bool unlocked_jump = false;
            if (may_bulk && njump < 2 && s.cntr >= 16)
            {
                if (s.datacd && s.cntr <= zone_hi) L = min(zone_hi - s.cntr + 1, n - pos);
                else if ((!s.datacd || (AEROL_WINDOW_JUMP && s.cntr >= zone_hi + AEROL_WINDOW_FLUSH + 1 && s.cntr < 1000000000)) && pos >= 64 && pos < scan_hi)
                {
                    int lim = min(n, scan_hi) - pos;
                    if (next_match >= pos) lim = min(lim, next_match - pos);
                    if (s.cntr < 1000000000) lim = min(lim, block_last - s.cntr); // pre-increment cntr of the completing bit = block_last
                    L = lim;
                    unlocked_jump = true;
                }
            }
            if (L >= AEROL_MINRUN)
            {
                // L soft bits with pre-increment cntr = c .. c+L-1: bit i has arm parity realimag ^ ((i+1)&1) and goes to block
                // index max(0, c+1+i - BitsInHeader); the dummy bits in front of the block all land on index 0 and are overwritten
                // by the first real one, so only i >= i0 is copied.  (Saturated cntr: every bit lands on one index; nothing to copy.)
                const int c = s.cntr;
                const bool sat = c >= 1000000000;
                // soft bytes of a group of four that the bit-by-bit path has collected but not stored yet
                if (!sat && !s.accbad)
                {
                    const int idx_next = c + 1 - g.BitsInHeader, k = idx_next & 3;
                    if (idx_next > 0)
                        for (int j = 0; j < k; j++) rx[idx_next - k + j] = (uint8_t)(s.acc >> (8 * (4 - k + j)));
                }
                const int i0 = sat ? L : max(0, g.BitsInHeader - (c + 1));
                if (i0 < L)
                {
                    const int d_src = pos + i0, d_dst = c + 1 + i0 - g.BitsInHeader, d_len = L - i0;
                    const int d_flags = ((s.realimag ^ ((i0 + 1) & 1)) & 1) | (s.inv_imag ? 2 : 0) | (s.inv_real ? 4 : 0);
                    if (njump == 0) { bulk_src = d_src; bulk_dst = d_dst; bulk_len = d_len; bulk_flags = d_flags; }
                    else { bulk2_src = d_src; bulk2_dst = d_dst; bulk2_len = d_len; bulk2_flags = d_flags; }
                }
                njump++; // (a jump with nothing to copy takes a slot too: two jumps per round)
                if (unlocked_jump)
                {
                    // the detector ran over every jumped bit: bring both arms' shift registers up to date from the last 64 of them
                    const int m = min(L, 64);
                    int r = s.realimag ^ ((L - m) & 1);
                    for (int i = L - m; i < L; i++)
                    {
                        r ^= 1;
                        const unsigned bitv = (((unsigned)soft_at(pos + i) & 0xFFu) >= 128u) ? 1u : 0u; // (eight per request)
                        if (r) s.pd_imag = (s.pd_imag << 1) | bitv;
                        else s.pd_real = (s.pd_real << 1) | bitv;
                    }
                }
                if (!sat) s.cntr += L;
                s.realimag ^= (L & 1);
                s.muw = min(100000, s.muw + L);
                s.gotsync_last = 0;
                gotsync = 0;
                s.accbad = 1; // the per-bit path may resume inside a group of four
                pos += L;
                jumped = true;
            }
            if (!jumped)
            {
                const int v = soft_at(pos);
                if (v < 0) { s.muw = 0; pos++; } // start-of-burst marker (aerol.cpp:1146-1152)
                else if (aerol_bit_a(g, s, v, gotsync, rx)) { has_block = 1; resume = 1; live = false; }
                else
                {
                    aerol_bit_b(g, p, ch, s, gotsync, nbits0 + pos);
                    pos++;
                    // The frame counter has just restarted (a unique word, or the frame length ran out): from here on block indices begin again at 0.  A
                    // stretch jumped EARLIER in this round is copied by k_aerol_bulk AFTER this kernel -- it would overwrite what the bits that follow
                    // store at the same indices.  So the lane copies its pending stretches itself, now, in stream order (a false unique word inside a
                    // detector window, or a channel acquiring lock behind a jumped stretch: once per event, not per frame -- a frame's own unique word
                    // comes in the round after its block was completed, with nothing pending), and both descriptor slots are free again.
                    if (BULK && s.cntr == -1 && njump > 0)
                    {
                        auto copy_now = [&](int src0, int dst0, int len, int flags) {
                            const int inv_first = (flags & 1) ? (flags >> 1) & 1 : (flags >> 2) & 1;
                            const int inv_second = (flags & 1) ? (flags >> 2) & 1 : (flags >> 1) & 1;
                            for (int k = 0; k < len; k++)
                            {
                                unsigned sbit = (unsigned)(int)sb[src0 + k] & 0xFFFFu;
                                if (((k & 1) ? inv_second : inv_first) && sbit != 128u) sbit = 255u - sbit;
                                rx[dst0 + k] = (uint8_t)(sbit & 255u);
                            }
                        };
                        if (bulk_len > 0) copy_now(bulk_src, bulk_dst, bulk_len, bulk_flags);
                        if (bulk2_len > 0) copy_now(bulk2_src, bulk2_dst, bulk2_len, bulk2_flags);
                        bulk_len = 0; bulk2_len = 0; njump = 0;
                    }
                }
            }
            if (pos >= n) live = false;
        }
    }
    if (!valid) return;
    ALD(AI_CNTR) = s.cntr; ALD(AI_DATACD) = s.datacd; ALD(AI_GOTSYNC_LAST) = s.gotsync_last; ALD(AI_REALIMAG) = s.realimag;
    ALD(AI_BLOCKCNT) = s.blockcnt; ALD(AI_MUW) = s.muw; ALD(AI_NINFO) = s.ninfo;
    ALD(AI_FRAMEINFO) = (int)s.frameinfo; ALD(AI_LASTFRAMEINFO) = (int)s.lastframeinfo;
    ALD(AI_PD_EXACT) = (int)s.pd_exact; ALD(AI_PD_IMAG) = (int)s.pd_imag; ALD(AI_PD_REAL) = (int)s.pd_real;
    ALD(AI_INV_IMAG) = s.inv_imag; ALD(AI_INV_REAL) = s.inv_real; ALD(AI_SCR_POS) = s.scr_pos; ALD(AI_DCDCOUNT) = s.dcdcount;
    ALD(AI_IN_POS) = pos; ALD(AI_RESUME) = resume; ALD(AI_RESUME_GOTSYNC) = gotsync; ALD(AI_HAS_BLOCK) = has_block;
    ALD(AI_EV_CNT) = s.ev_cnt; ALD(AI_OVERFLOW) = s.overflow; ALD(AI_ACC) = (int)s.acc; ALD(AI_ACCBAD) = s.accbad;
    ALD(AI_BULK_LEN) = bulk_len > 0 ? bulk_len : 0; ALD(AI_BULK_SRC) = bulk_src; ALD(AI_BULK_DST) = bulk_dst; ALD(AI_BULK_FLAGS) = bulk_flags;
    ALD(AI_BULK2_LEN) = bulk2_len > 0 ? bulk2_len : 0; ALD(AI_BULK2_SRC) = bulk2_src; ALD(AI_BULK2_DST) = bulk2_dst; ALD(AI_BULK2_FLAGS) = bulk2_flags;
}

// Per write and channel, one wavefront per channel: (1) does the input hold a start-of-burst marker (a negative soft value); (2) the
// first positions q >= 62 at which the unique-word detector of q's arm WOULD see the word or its complement if it had been running
// over the last 32 soft bits of that arm (q, q-2, .., q-62) -- which is the case for a channel without data carrier, whose detector
// runs on every bit.  k_aerol_bits uses them to jump over the stretches of such a channel in which nothing can happen.
// Hard bits are packed per position parity into LDS; a window of 32 consecutive bits of one parity, oldest in bit 0, equals the
// detector's shift register bit-reversed, so it is compared with the bit-reversed word.
__global__ __launch_bounds__(256) void k_aerol_scan(const AGeom g, const APtrs p, const int16_t *__restrict__ soft, const int *__restrict__ counts, int stride)
{
    __shared__ unsigned par[4][2][512 + 2]; // [wave][position parity][32 positions of that parity per word]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + w;
    if (ch >= g.nch) return;
    const int nall = counts[ch];
    const int n = min(nall, 32768);
    const int16_t *sb = soft + (size_t)ch * stride;
    // a channel without data carrier consults the positions everywhere, a locked one inside its detector windows (round 6: k_aerol_bits jumps those
    // too); until then only unlocked channels were scanned
    const bool want = g.oqpsk != 0;
    int neg = 0;
    for (int q = n + lane; q < nall; q += 64) neg |= (sb[q] < 0) ? 1 : 0; // beyond the scanned range: markers only
    // Eight positions per lane and request (round 6; one 2-byte load per lane and 64 positions before: 82 dependent rounds of load, ballot, LDS store per
    // frame, 0.53 ms per 65 536-channel write against 0.19 for the bytes at HBM speed).  A lane's eight hard bits are four of either parity = a nibble of
    // each parity stream at bit 4 (lane & 7) of word base / 64 + lane / 8: eight lanes OR their nibbles together, the first of them stores the word.
    for (int base = 0; base < n; base += 512)
    {
        const int q0 = base + 8 * lane;
        short v8[8];
        if (q0 + 8 <= n) __builtin_memcpy(v8, sb + q0, 16);
        else
        {
#pragma unroll
            for (int k = 0; k < 8; k++) v8[k] = (q0 + k < n) ? sb[q0 + k] : (short)0;
        }
        unsigned ev = 0, od = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            neg |= (v8[k] < 0) ? 1 : 0;
            const unsigned h = ((((unsigned)(int)v8[k]) & 0xFFu) >= 128u && q0 + k < n) ? 1u : 0u;
            if (k & 1) od |= h << (k >> 1); else ev |= h << (k >> 1);
        }
        if (!want) continue;
        const int sh = 4 * (lane & 7);
        unsigned long long x = ((unsigned long long)(od << sh) << 32) | (unsigned long long)(ev << sh);
        x |= __shfl_xor(x, 1);
        x |= __shfl_xor(x, 2);
        x |= __shfl_xor(x, 4);
        if ((lane & 7) == 0)
        {
            par[w][0][(base >> 6) + (lane >> 3)] = (unsigned)x;
            par[w][1][(base >> 6) + (lane >> 3)] = (unsigned)(x >> 32);
        }
    }
    const int any = __any(neg) ? 1 : 0;
    int nmatch = 0;
    if (want)
    {
        const unsigned ruw = __brev(AEROL_UW);
        for (int base = 0; base < n; base += 64)
        {
            const int q = base + lane;
            bool hit = false;
            if (q >= 62 && q < n)
            {
                const int j = q >> 1; // index within the parity stream; window = stream bits j-31 .. j
                const unsigned *pw = par[w][q & 1];
                const int lo = j - 31;
                const unsigned long long two = (unsigned long long)pw[lo >> 5] | ((unsigned long long)pw[(lo >> 5) + 1] << 32);
                const unsigned win = (unsigned)(two >> (lo & 31));
                hit = (win == ruw) || (win == ~ruw);
            }
            const unsigned long long hm = __ballot(hit);
            if (hm)
            {
                if (hit)
                {
                    const int r = nmatch + __popcll(hm & ((1ull << lane) - 1ull));
                    if (r < AEROL_NMATCH) p.I[(size_t)(AI_MATCH0 + r) * g.nchp + ch] = q;
                }
                nmatch += __popcll(hm);
            }
        }
    }
    if (lane == 0) { ALD(AI_MARKER) = any; ALD(AI_NMATCH) = nmatch; }
}

// the stretches k_aerol_bits jumped over: rx[dst + i] = soft bit (src + i), inverted per arm, i < len.  One wavefront per channel,
// four soft bits per lane and step.
__global__ __launch_bounds__(256) void k_aerol_bulk(const AGeom g, const APtrs p, const int16_t *__restrict__ soft, int stride)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + w;
    if (ch >= g.nch) return;
    // two stretches per channel and round (k_aerol_bits: the frame body and the detector window behind it); they never overlap
    for (int d = 0; d < 2; d++)
    {
    const int len = d ? ALD(AI_BULK2_LEN) : ALD(AI_BULK_LEN);
    if (len <= 0) continue; // wave-uniform
    const int flags = d ? ALD(AI_BULK2_FLAGS) : ALD(AI_BULK_FLAGS);
    const int16_t *src = soft + (size_t)ch * stride + (d ? ALD(AI_BULK2_SRC) : ALD(AI_BULK_SRC));
    uint8_t *dst = p.rx + (size_t)ch * g.blocksz + (d ? ALD(AI_BULK2_DST) : ALD(AI_BULK_DST));
    const int inv_first = (flags & 1) ? (flags >> 1) & 1 : (flags >> 2) & 1;   // arm of bit 0: imag if parity 1
    const int inv_second = (flags & 1) ? (flags >> 2) & 1 : (flags >> 1) & 1;
    auto conv = [](int v, int inv) -> unsigned {
        unsigned sbit = (unsigned)v & 0xFFFFu;
        if (inv && sbit != 128u) sbit = 255u - sbit;
        return sbit & 255u;
    };
    const int n4 = len & ~3;
    for (int i = lane * 4; i < n4; i += 256)
    {
        short q[4];
        __builtin_memcpy(q, src + i, 8);
        const unsigned o = conv(q[0], inv_first) | (conv(q[1], inv_second) << 8) | (conv(q[2], inv_first) << 16) | (conv(q[3], inv_second) << 24);
        __builtin_memcpy(dst + i, &o, 4);
    }
    const int i = n4 + lane;
    if (i < len) dst[i] = (uint8_t)conv(src[i], (i & 1) ? inv_second : inv_first);
    }
}

// AeroLInterleaver::deinterleave_ba (aerol.cpp:603-625) for the channels that completed a block this round, one wavefront per
// channel: out[j*64 + i] = block[((i*27) % 64) * N + j].  The block goes through LDS (coalesced 16-byte reads of the received-order
// row, 4-byte coalesced writes of the deinterleaved row).
__global__ __launch_bounds__(256) void k_aerol_deint(const AGeom g, const APtrs p)
{
    __shared__ uint8_t blk[4][4992 + 16];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + w;
    if (ch >= g.nch) return;
    if (!ALD(AI_HAS_BLOCK)) return; // wave-uniform
    const uint4 *src = (const uint4 *)(p.rx + (size_t)ch * g.blocksz);
    uint4 *b16 = (uint4 *)blk[w];
    for (int q = lane; q < g.blocksz / 16; q += 64) b16[q] = src[q];
    // (one wavefront per block buffer: LDS accesses of a wave are in order, no barrier needed)
    // row-major [channel][blocksz] for k_viterbi, or k_viterbi_lanes' tiled layout [wavefront][16-byte group][lane][16]
    unsigned *dst = (unsigned *)(g.tiled ? p.deint + (size_t)(ch >> 6) * 64 * g.blocksz + (ch & 63) * 16 : p.deint + (size_t)ch * g.blocksz);
    const int gmul = g.tiled ? 64 : 1;
    const int i0 = (lane & 15) * 4;
    const int r0 = ((i0 * 27) & 63) * g.N, r1 = (((i0 + 1) * 27) & 63) * g.N, r2 = (((i0 + 2) * 27) & 63) * g.N, r3 = (((i0 + 3) * 27) & 63) * g.N;
    const uint8_t *bb = blk[w];
    for (int j = lane >> 4; j < g.N; j += 4)
    {
        const unsigned v = (unsigned)bb[r0 + j] | ((unsigned)bb[r1 + j] << 8) | ((unsigned)bb[r2 + j] << 16) | ((unsigned)bb[r3 + j] << 24);
        dst[((j * 4 + ((lane & 15) >> 2)) * gmul) * 4 + (lane & 3)] = v; // byte j*64 + 4*(lane&15): group j*4 + (lane&15)/4, dword lane&3
    }
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

// end of a frame: CRC-16 of every 12-byte signal unit, data-carrier-detect bookkeeping, output rows (aerol.cpp:1583-1600)
__device__ __forceinline__ void aerol_frame_end(const AGeom &g, const APtrs &p, int ch)
{
    uint8_t *info = p.info + (size_t)ch * g.info_cap;
    const int ninfo = ALD(AI_NINFO);
    int datacd = ALD(AI_DATACD), dcdcount = ALD(AI_DCDCOUNT), su_cnt = ALD(AI_SU_CNT), ev_cnt = ALD(AI_EV_CNT), overflow = ALD(AI_OVERFLOW);
    const int nframes = ALD(AI_NFRAMES);
    const int frameinfo = ALD(AI_FRAMEINFO);
    const long long bitidx = (((long long)(unsigned)ALD(AI_NBITS_LO)) | ((long long)ALD(AI_NBITS_HI) << 32)) + ALD(AI_IN_POS);
    for (int kk = 0; kk < ninfo / 12; kk++)
    {
        // the unit's 12 bytes once, into registers: read through the pointer, every byte of the row below was loaded again and waited for on its own --
        // behind the row's previous store, since vmcnt retires in order -- 312 round trips per frame, most of k_aerol_post_packed's 0.45 ms (round 6)
        uint8_t su[12];
        __builtin_memcpy(su, info + kk * 12, 12);
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
    // DelayLine::update (aerol.h: write at the pointer, advance, read at the new pointer) + descrambler + LSB-first byte packing.
    // A block is shorter than the delay line, so every bit read in this pass predates it: 16 decoded bits per step -- one 16-byte
    // read of the line (the bits leaving it), one 16-byte write (the bits entering), 16 scrambler bytes, two packed bytes out.
    // Steps that would wrap the line, run off the scrambler table or the info buffer go bit by bit.
    int h = 0;
    for (; h + 16 <= nb; h += 16)
    {
        if (dl2_ptr + 17 <= g.dl2_sz && scr_pos + 16 <= 5000 && ninfo + 2 <= g.info_cap)
        {
            uint4 in4, old4, sc4;
            __builtin_memcpy(&in4, vb + h, 16);
            __builtin_memcpy(&old4, dl2 + dl2_ptr + 1, 16);
            __builtin_memcpy(&sc4, p.scr + scr_pos, 16);
            __builtin_memcpy(dl2 + dl2_ptr, &in4, 16);
            dl2_ptr += 16; if (dl2_ptr >= g.dl2_sz) dl2_ptr = 0;
            scr_pos += 16;
            const unsigned x0 = old4.x ^ sc4.x, x1 = old4.y ^ sc4.y, x2 = old4.z ^ sc4.z, x3 = old4.w ^ sc4.w;
            // four 0/1 bytes -> a nibble, first byte in bit 0
            const unsigned b0 = ((x0 * 0x01020408u) >> 24) & 15u, b1 = ((x1 * 0x01020408u) >> 24) & 15u;
            const unsigned b2 = ((x2 * 0x01020408u) >> 24) & 15u, b3 = ((x3 * 0x01020408u) >> 24) & 15u;
            info[ninfo] = (uint8_t)(b0 | (b1 << 4));
            info[ninfo + 1] = (uint8_t)(b2 | (b3 << 4));
            ninfo += 2;
        }
        else
        {
            unsigned chv = 0;
            for (int q = 0; q < 16; q++)
            {
                dl2[dl2_ptr] = vb[h + q];
                dl2_ptr++; if (dl2_ptr >= g.dl2_sz) dl2_ptr = 0;
                unsigned v = dl2[dl2_ptr];
                v ^= p.scr[scr_pos < 5000 ? scr_pos : 4999];
                scr_pos++;
                chv |= v * 128u;
                if ((q & 7) == 7) { if (ninfo < g.info_cap) info[ninfo++] = (uint8_t)chv; chv = 0; }
                else chv >>= 1;
            }
        }
    }
    int charptr = 0;
    unsigned chv = 0;
    for (; h < nb; h++)
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
    if ((cntr - g.BitsInHeader) == (g.NumberOfBits - 1)) aerol_frame_end(g, p, ch);
}

// ---- the same pass on packed bits (large banks: k_viterbi_lanes writes one bit per decoded bit) ----
// bits [q, q+m) of a little-endian bit array, m <= 32, no wrap (one spare word behind the array)
__device__ __forceinline__ unsigned abits_get(const unsigned *w, int q, int m)
{
    const unsigned long long two = (unsigned long long)w[q >> 5] | ((unsigned long long)w[(q >> 5) + 1] << 32);
    const unsigned v = (unsigned)(two >> (q & 31));
    return m >= 32 ? v : (v & ((1u << m) - 1u));
}
__device__ __forceinline__ void abits_put(unsigned *w, int q, int m, unsigned val)
{
    const unsigned long long mask = (m >= 32 ? 0xFFFFFFFFull : ((1ull << m) - 1ull)) << (q & 31);
    const unsigned long long v = ((unsigned long long)val << (q & 31)) & mask;
    unsigned long long two = (unsigned long long)w[q >> 5] | ((unsigned long long)w[(q >> 5) + 1] << 32);
    two = (two & ~mask) | v;
    w[q >> 5] = (unsigned)two;
    w[(q >> 5) + 1] = (unsigned)(two >> 32);
}
// DelayLine + scrambler + byte packing, 32 decoded bits per step: the bits leaving the line are the line's content from the read
// position on (a block is shorter than the line), the block's bits go in behind; packed LSB-first bits ARE the information bytes.
__global__ void k_aerol_post_packed(const AGeom g, const APtrs p)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= g.nch) return;
    if (!ALD(AI_HAS_BLOCK)) return;
    int dl2_ptr = ALD(AI_DL2_PTR), scr_pos = ALD(AI_SCR_POS), ninfo = ALD(AI_NINFO);
    const int cntr = ALD(AI_CNTR);
    unsigned *ring = p.dl2w + (size_t)ch * g.dl2_words;
    uint8_t *info = p.info + (size_t)ch * g.info_cap;
    const unsigned *vbw = (const unsigned *)(p.vbits + (size_t)ch * (g.blocksz / 2));
    const int vblocks = ALD(AI_VBLOCKS);
    const int nb = vblocks ? g.blocksz / 2 : (g.blocksz + 24) / 2 - 25; // see k_aerol_post
    ALD(AI_VBLOCKS) = vblocks + 1;
    const int sz = g.dl2_sz;
    auto ring_get = [&](int q, int m) -> unsigned { // m bits from ring position q, wrapping at sz
        if (q + m <= sz) return abits_get(ring, q, m);
        const int m1 = sz - q;
        return abits_get(ring, q, m1) | (abits_get(ring, 0, m - m1) << m1);
    };
    auto ring_put = [&](int q, int m, unsigned v) {
        if (q + m <= sz) { abits_put(ring, q, m, v); return; }
        const int m1 = sz - q;
        abits_put(ring, q, m1, v);
        abits_put(ring, 0, m - m1, v >> m1);
    };
    const int nbytes = nb / 8; // bits beyond the last whole byte are dropped (charptr starts at 0 in every block)
    // Round 6: a block is shorter than the line (the comment above), so nothing this call reads was written by this call: ALL the reads first, eight steps
    // per wait, then the block's bits go into the ring as whole words (bit-granular only at the two ends and at the wrap).  Step by step it was two
    // round trips per 32 bits -- the read, then the read-modify-write of the put -- 156 per 10.5 kbps frame.
    if (nb + 64 < sz && scr_pos + nb <= 5000)
    {
        const int ptr0 = dl2_ptr, scr0 = scr_pos;
        for (int h0 = 0; h0 < nb; h0 += 256)
        {
            unsigned outw[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const int h = h0 + 32 * k;
                outw[k] = 0;
                if (h < nb)
                {
                    const int m = min(32, nb - h);
                    int rq = ptr0 + 1 + h; if (rq >= sz) rq -= sz;
                    outw[k] = ring_get(rq, m) ^ abits_get(p.scrw, scr0 + h, m);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const int h = h0 + 32 * k;
                if (h >= nb) break;
                const int nby = min(4, nbytes - (h >> 3)); // whole bytes of this step
                for (int b = 0; b < nby; b++)
                    if (ninfo < g.info_cap) info[ninfo++] = (uint8_t)(outw[k] >> (8 * b));
            }
        }
        // the block's bits behind the pointer: ring bits [ptr0, ptr0 + nb), wrapping at sz
        auto put_seg = [&](int q, int sbit, int len) {
            while (len > 0)
            {
                const int off = q & 31;
                if (off == 0 && len >= 256)
                {
                    unsigned sw[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) sw[k] = vbw[(sbit >> 5) + k]; // (one spare word behind vbits' rows: abits_get reads the same)
                    const int sh = sbit & 31;
#pragma unroll
                    for (int k = 0; k < 8; k++) ring[(q >> 5) + k] = sh ? ((sw[k] >> sh) | (sw[k + 1] << (32 - sh))) : sw[k];
                    q += 256; sbit += 256; len -= 256;
                    continue;
                }
                const int m = min(32 - off, len);
                const unsigned v = abits_get(vbw, sbit, m);
                if (m == 32) ring[q >> 5] = v;
                else abits_put(ring, q, m, v);
                q += m; sbit += m; len -= m;
            }
        };
        const int seg1 = min(nb, sz - ptr0);
        put_seg(ptr0, 0, seg1);
        if (seg1 < nb) put_seg(0, seg1, nb - seg1);
        dl2_ptr = ptr0 + nb; if (dl2_ptr >= sz) dl2_ptr -= sz;
        scr_pos = scr0 + nb;
    }
    else
    for (int h = 0; h < nb; h += 32)
    {
        const int m = min(32, nb - h);
        const unsigned in = vbw[h >> 5] & (m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u));
        // update(): write at the pointer, advance, read at the new pointer -- for bit i of this step: write (ptr+i), read (ptr+i+1)
        int rq = dl2_ptr + 1; if (rq >= sz) rq -= sz;
        unsigned outw = ring_get(rq, m);
        ring_put(dl2_ptr, m, in);
        dl2_ptr += m; if (dl2_ptr >= sz) dl2_ptr -= sz;
        // descramble (scr_pos + m <= 5000 in every frame the reference can produce; the clamp of the table index is kept for the rest)
        if (scr_pos + m <= 5000) outw ^= abits_get(p.scrw, scr_pos, m);
        else
            for (int i = 0; i < m; i++) outw ^= (unsigned)p.scr[scr_pos + i < 5000 ? scr_pos + i : 4999] << i;
        scr_pos += m;
        const int nby = min(4, nbytes - (h >> 3)); // whole bytes of this step
        for (int b = 0; b < nby; b++)
            if (ninfo < g.info_cap) info[ninfo++] = (uint8_t)(outw >> (8 * b));
    }
    ALD(AI_DL2_PTR) = dl2_ptr; ALD(AI_SCR_POS) = scr_pos; ALD(AI_NINFO) = ninfo;
    if ((cntr - g.BitsInHeader) == (g.NumberOfBits - 1)) aerol_frame_end(g, p, ch);
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
