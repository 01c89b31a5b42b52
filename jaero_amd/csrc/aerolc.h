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
#define CC_DL2 2709    // DelayLine(2714 - 6): ring of length + 1
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
    uint8_t *dl2;                // [CC_DL2][nchp] delay line
    const uint8_t *scr;          // [5000] scrambler sequence
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
        const int sv = s[pos];
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
    CLD(CI_POS) = pos; CLD(CI_HAS_BLOCK) = has; CLD(CI_BULK_N) = nbulk;
    CLD(CI_CNTR) = cntr; CLD(CI_REALIMAG) = realimag; CLD(CI_GSLAST) = gslast;
    CLD(CI_INV_REAL) = inv[0]; CLD(CI_INV_IMAG) = inv[1];
    CLD(CI_EV_CNT) = ev_cnt; CLD(CI_OVERFLOW) = overflow;
#pragma unroll
    for (int k = 0; k < 4; k++) p.B[(size_t)k * g.nchp + ch] = b[k];
}

// One wavefront per channel: the (at most two) stretches k_aerolc_bits jumped over in this round, in stream order: a later stretch overwrites an
// earlier one's positions (a frame abandoned for a new unique word).  Both in ONE launch since round 6 (the second launch was ~65 000
// workgroups that returned at once, every round): the wavefront finishes stretch 0 -- its stores made visible to the wavefront and retired -- before
// it starts stretch 1.  The walked soft bits of the same round never share a position with them (their cntr values lie outside [2, CC_FRAME - 111]).
__global__ __launch_bounds__(64) void k_aerolc_bulk(const CGeom g, const CPtrs p, const int16_t *__restrict__ soft, int stride, int konly)
{
    // konly < 0: both stretches (the product's one launch); 0 / 1: that stretch alone (tests/host_emul runs a workgroup's threads one after the other, where
    // only a launch boundary orders the stretches)
    const int ch = blockIdx.x;
    if (ch >= g.nch) return;
    const int nk = CLD(CI_BULK_N); // wave-uniform
    const int16_t *s = soft + (size_t)ch * stride;
    uint8_t *dep = p.dep + (size_t)ch * CC_PITCH;
    for (int k = 0; k < nk && k < 2; k++)
    {
    if (konly >= 0 && k != konly) continue;
    if (k > 0 && konly < 0)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
    }
    const int f0 = CI_BULK0_SRC + 4 * k;
    const int src = p.I[(size_t)(f0 + 0) * g.nchp + ch], c0 = p.I[(size_t)(f0 + 1) * g.nchp + ch];
    const int len = p.I[(size_t)(f0 + 2) * g.nchp + ch], fl = p.I[(size_t)(f0 + 3) * g.nchp + ch];
    for (int j = threadIdx.x; j < len; j += blockDim.x)
    {
        const int sv = s[src + j];
        unsigned soft_bit = (unsigned)(unsigned short)sv;
        const int realimag = ((fl & 1) + j + 1) & 1;         // toggled before the bit is used (:2207)
        const int inverted = realimag ? (fl & 2) : (fl & 4); // realimag != 0: the real arm's detector and flag
        if (inverted) { if (soft_bit != 128) soft_bit = 255 - soft_bit; }
        const int di = cc_dep_index(c0 + j);
        if (di >= 0) dep[di] = (uint8_t)soft_bit;
    }
    }
}

// lane = channel: the end of a frame (:2318-2490) for the channels whose frame the Viterbi just decoded
__global__ __launch_bounds__(64) void k_aerolc_post(const CGeom g, const CPtrs p)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= g.nch) return;
    if (!CLD(CI_HAS_BLOCK)) return;
    const uint8_t *vb = p.vbits + (size_t)ch * (CC_NSOFT / 2);
    int dl2_ptr = CLD(CI_DL2_PTR), datacd = CLD(CI_DATACD), dcdcount = CLD(CI_DCDCOUNT);
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
    // one pass over the 2714 bits: delay line (:2330), scrambler (:2333), then both extractions read the same descrambled bit h
    unsigned char info[12]; int ninfo = 0, sch = 0, scharptr = 0, kk = 0; // sub-band units (:2343-2358)
    int vch = 0, vcharptr = 0, nv = 0;                                      // voice bytes (:2454-2478)
    // CB bits per batch (round 5): a bit's step through the delay line is a store at the ring pointer and a load one slot further -- the oldest
    // entry, written 2708 steps ago -- so within a batch the loads touch slots ptr + 1 .. ptr + CB and the stores slots ptr .. ptr + CB - 1:
    // issued loads first, then stores, every load still sees what the bit-by-bit order shows it (load k reads the slot store k + 1 overwrites), and
    // the round trips to memory overlap instead of following one another (0.83 us per bit before: 2.25 ms per 65 536-channel step; batches of
    // 8: 1.24 ms, of 16: 1.13 ms).
    constexpr int CB = 16;
    for (int h0 = 0; h0 < CC_NBITS; h0 += CB)
    {
        const int nb = (CC_NBITS - h0) < CB ? (CC_NBITS - h0) : CB;
        int vin[CB], vold[CB];
#pragma unroll
        for (int k = 0; k < CB; k++) vin[k] = (k < nb) ? vb[h0 + k] : 0; // positions the first call of the codec does not produce stay 0 (buffer zeroed at create)
#pragma unroll
        for (int k = 0; k < CB; k++)
        {
            int q = dl2_ptr + k + 1; if (q >= CC_DL2) q -= CC_DL2;
            vold[k] = (k < nb) ? (int)p.dl2[(size_t)q * g.nchp + ch] : 0;
        }
#pragma unroll
        for (int k = 0; k < CB; k++)
        {
            int q = dl2_ptr + k; if (q >= CC_DL2) q -= CC_DL2;
            if (k < nb) p.dl2[(size_t)q * g.nchp + ch] = (uint8_t)vin[k];
        }
        dl2_ptr += nb; if (dl2_ptr >= CC_DL2) dl2_ptr -= CC_DL2;
#pragma unroll
        for (int k = 0; k < CB; k++)
        {
        if (k >= nb) break;
        const int h = h0 + k;
        int v = vold[k];
        v ^= p.scr[h];
        const int y = h / 109, o = h - y * 109; // primary field y: bit 0, 96 voice bits (1..96), 12 sub-band bits (97..108)
        if (o >= 1 && o <= 96)
        {
            vch |= v * 128;
            vcharptr++; vcharptr %= 8;
            if (vcharptr == 0) { if (vrow && nv < 300) vrow[4 + nv] = (uint8_t)vch; nv++; vch = 0; }
            else vch >>= 1;
        }
        else if (o >= 97 && y < 24)
        {
            sch |= v * 128;
            scharptr++; scharptr %= 8;
            if (scharptr == 0) { info[ninfo++] = (unsigned char)sch; sch = 0; }
            else sch >>= 1;
            if (o == 108 && ninfo == 12)
            {
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
                ninfo = 0;
            }
        }
        }
    }
    CLD(CI_DL2_PTR) = dl2_ptr; CLD(CI_DATACD) = datacd; CLD(CI_DCDCOUNT) = dcdcount;
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
    CA(cs->p.dl2, (size_t)CC_DL2 * g.nchp);
    CA(cs->p.sus, (size_t)g.nchp * g.su_cap * 16);
    CA(cs->p.voice, (size_t)g.nchp * g.v_cap * 304);
    CA(cs->p.events, (size_t)g.nchp * g.ev_cap * 3);
    if (viterbi_use_lanes(g.nch, CC_NSOFT, 24)) CA(cs->d_vhist, viterbi_hist_bytes(g.nch) / sizeof(unsigned long long));
    uint8_t *d_scr = nullptr;
    CA(d_scr, 5000);
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
        hipLaunchKernelGGL(k_aerolc_bulk, dim3(g.nch), block, 0, st, g, cs->p, dsoft, stride, -1); // both stretches of a round, in order
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
