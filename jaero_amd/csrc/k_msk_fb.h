// k_msk_fb.h -- sample-loop kernel for the continuous MSK demodulator with an 80-tap matched filter (1200 bps at 48 kHz, 600 bps at
// 24 kHz), front / back wavefront pairs.
//
// Same arithmetic as k_msk.h (MskDemodulator::writeData's per-sample loop, JAERO/mskdemodulator.cpp:319-485), one channel per lane, the
// per-sample work of 64 channels shared by two wavefronts exactly as in k_oqpsk_fb.h:
//
//   F ("front"):  PCM -> double, coarse ring fill (mixer_center), mix with the carrier NCO value the back half hands over, half-sine
//                 matched filter (history in LDS + registers), MSKEbNoMeasure, AGC + clip.  Owns the PCM / AGC / EbNo / coarse-ring
//                 streams and the filter history.
//   B ("back"):   the SPS-sample delayed arm, |pt_msk| -> resonator -> quadrature delay -> symbol PLL weighted by 1 - |tanh(err)|, and at
//                 symbol instants the carrier loop, residual rotation, MSE, soft differential decode, soft bits; the carrier NCO.  Owns
//                 the two delay lines, the symbol-rate windows and the outputs.
//
// The matched filter's output for sample n+1 does not contain x[n+1] (FIR::FIRUpdateAndProcess excludes the newest sample,
// DSP.cpp:292-304) and x[n] = mixer2(n) * pcm[n] is known once the back half has finished sample n-1: while B runs sample n, F forms
// x[n], pushes it and produces the AGC'd, clipped sample n+1.  One LDS-only barrier per sample, double-buffered mailboxes.
// Only the front half needs LDS.  Banks of at most two channel groups per CU run ONE pair per workgroup (the halves on different SIMDs,
// 512 registers each): 36 of the 80 history entries of each arm in LDS, 44 in the front half's registers.  Full banks run FOUR pairs per
// workgroup (a pair shares a SIMD, 256 registers per wavefront, 40 064 B of LDS per pair = 160 256 B per CU); there the front half's
// 44-deep tail does not fit, so the history is split three ways (MFB4_*, below).  Measured on an MI355X: 256 channels 82 -> 113
// Msamples/s, 16 384 channels 4.07 -> 4.94 Gsamples/s, 65 536 channels 9.2 -> 10.6 Gsamples/s (sample loop 15.9 -> 12.1 ms per step); the
// first four-pair version, with the whole tail in the front half, spilled 458 registers and ran at 6.6.  The 160-tap filter (600 bps
// at 48 kHz) keeps k_msk_samples.
#pragma once
#include <type_traits>
#include "jaero_device.h"
#include "k_oqpsk_fb.h" // fb_barrier, fb_wt_next, jd_div_const

#define MFB_LDSN 36  // one pair per workgroup (small banks): 36 history entries of each arm in LDS, 44 in registers
// ... of which the MFB1_TB oldest in the BACK half's (round 5): in a small bank the halves sit on different SIMDs and the front half is the long pole
// (its 320 filter instructions: 1.5 of a sample's 2.5 us in the phase trace, the back half waiting 1.1 us at the barrier), so the back half
// starts every filter sum as it does in the four-pair kernel
#ifndef MFB1_TB
#define MFB1_TB 32 // 24 / 32 / 36 / 40 measured: 256-channel bank 117.0 / 121.3 / 121.0 / 119.5 Msamples/s (103.4 with the whole tail in the front half)
#endif
// Four pairs per workgroup (full banks): a wavefront has 256 registers at two per SIMD, the front half's 44-deep tail (176 registers)
// does not fit.  Then the OLDEST MFB4_TB entries of each arm live in the BACK half's registers (it has room), which starts every filter
// output -- the sum runs oldest first -- and hands the partial sum to the front half; the front half keeps 80 - 32 - MFB4_TB entries in registers
// and 32 in LDS and continues the same sum.  Same operations in the same order as the single accumulator chain.
// (Round 5: 18, not 22 -- with 22 the back half reloaded ten spilled values inside its sample loop, every reload an s_waitcnt vmcnt(0) in
// front of the ring entries requested ahead; with 18 neither loop touches scratch, what is left spilled is the state prologue / epilogue.
// Round 6, with the filter in four versions over ring blocks: 18 / 14 / 10 = 14.1 / 15.0 / 18.9 ms per step -- 18 stays.)
#define MFB4_LDSN 32
#ifndef MFB4_TB
#define MFB4_TB 18
#endif
// 160 taps (600 bps at 48 kHz; round 3): TWO pairs per workgroup, every wavefront alone on its SIMD (512 registers): 72 entries of each arm
// in LDS (81 664 B per pair, 163 328 B per CU), 28 in the front half's registers, the 60 oldest in the back half's (36 / 52 until round 5; round 3 measured splits 36/44/52/60: 5.87, 6.02, 6.03 Gsamples/s; all still spill 230-300 registers).  k_msk_samples<160,78>
// (two wavefronts per CU, 82 entries in 256 registers) spilled ~750 registers; one wavefront per CU with the whole history in LDS and no
// scratch is slower still (3.4 against 2.3 Gsamples/s at 65 536 channels): occupancy, not the scratch traffic, decides there.
#ifndef MFB2_LDSN
#define MFB2_LDSN 72
#endif
#ifndef MFB_LAZY_K
#define MFB_LAZY_K 2 // the back half's register tail moves every MFB_LAZY_K-th sample, by MFB_LAZY_K places (see mfb_back): 1 / 2 / 4 / 8 measured at 600 bps =
                     // 28.7 / 28.4 / 30.6 / 31.4 ms per step -- every further version of the 60-term sum is 3 KB more code in a loop that already fills
                     // the instruction cache two CUs share (front half 8 versions x ~600 instructions + back half ~2 600), and that costs more than the moves
#endif
#ifndef MFB2_TB
#define MFB2_TB 64 // round 5, with the filter op for op: 40 / 52 / 60 / 68 measured 5 621 / 5 998 / 6 145 / 6 130 Msamples/s at 65 536 channels (60 until round 6);
                   // round 6, with ring blocks of 4 and the tail moved every 2nd sample: 52 / 56 / 60 / 64 / 68 = 28.9 / 28.9 / 27.9 / 27.3 / 28.2 ms per step
#endif

struct MfbLds
{
    double *lre, *lim, *ltap; // [LDSN][64], [LDSN][64], [FIRN]
    double *data;             // [2][2][64]  F -> B: sre, sim of a sample
    int *idx;                 // [2][64]     B -> F: table index of mixer2 for a sample
    double *oldx;             // [2][2][64]  F -> B (TB > 0): the history entry that reaches the back half's tail two samples on
    double *acc;              // [2][2][64]  B -> F (TB > 0): the filter sum over the back half's entries for a sample
};
template <int FIRN, int LDSN, int TB>
constexpr int mfb_pair_doubles() { return 2 * LDSN * 64 + FIRN + 2 * 2 * 64 + 64 + (TB > 0 ? 2 * 2 * 2 * 64 : 0); }

// jd_fir_eval (jaero_device.h) continuing a sum: taps T0 .. FIRN-1 over the TAILN register entries (oldest first) and the LDSN LDS
// entries, starting from (are0, aim0) = the sum over taps 0 .. T0-1
template <int FIRN, int LDSN, int D, int T0, int TAILA>
__device__ __forceinline__ void mfb_fir_continue(const double *lre, const double *lim, const double *ltap, const double (&tre)[TAILA],
                                                 const double (&tim)[TAILA], int fir_slot, int lane, double are0, double aim0, double &ore, double &oim)
{
    constexpr int NT = FIRN - T0, TAILN = NT - LDSN;
    double pr[D], pi[D], pt[D];
    int slot = fir_slot;
    auto fetch = [&](int s, int q) {
        pt[q] = ltap[T0 + s];
        if (s >= TAILN)
        {
            pr[q] = lre[slot * 64 + lane];
            pi[q] = lim[slot * 64 + lane];
            slot++;
            if (slot >= LDSN) slot = 0;
        }
    };
#pragma unroll
    for (int s = 0; s < D; s++) fetch(s, s);
    __builtin_amdgcn_sched_barrier(0);
    double are = are0, aim = aim0;
#pragma unroll
    for (int s = 0; s < NT; s++)
    {
        const int q = s % D;
        const double xr = (s < TAILN) ? tre[(TAILN - 1 - s) < 0 ? 0 : (TAILN - 1 - s)] : pr[q];
        const double xi = (s < TAILN) ? tim[(TAILN - 1 - s) < 0 ? 0 : (TAILN - 1 - s)] : pi[q];
        are = are + pt[q] * xr;
        aim = aim + pt[q] * xi;
        asm volatile("" : "+v"(are), "+v"(aim)); // keeps the software pipeline as written (see jd_fir_eval)
        if (s + D < NT) fetch(s + D, q);
        __builtin_amdgcn_sched_barrier(0);
    }
    ore = are; oim = aim;
}

// mfb_fir_continue without per-term address arithmetic (round 6).  A wavefront that is alone on its SIMD issues at most one instruction of ANY kind
// per four cycles, so the sample loops are bound by their instruction COUNT, scalar bookkeeping included (profiles/r6_msk600_trace.md): as written
// above a term of the LDS part costs ~15 instructions, six of them useful (2 mul, 2 add, one ds_read2st64_b64 for both arms, one tap read) -- the rest
// forms the ring address (fir_slot + k) mod LDSN (five scalar and two vector instructions) and moves the tap's constant address into a register.
// Here the ring is seen as NB = LDSN / BS blocks of BS slots and fir_slot = a BS + BV: entry k of the LDS part lies in block ((a + m) mod NB) at slot
// r of it, m = (BV + k) / BS and r = (BV + k) % BS -- both compile-time constants once BV is (a switch over the BS values of BV picks the version, as
// k_oqpsk_fb's filter does over all its 36 ring positions).  The NB + 1 block addresses (per lane) are formed once per sample; every read is then
// `base register + immediate offset`, and so are the tap reads (tapz: a zero the compiler cannot see through).  Same terms, same order, same sums.
#ifndef MFB_BS_MAX
#define MFB_BS_MAX 4 // ring blocks of 4 slots = 4 versions of the front half's sum: 8 / 4 / 2 measured: 600 bps 28.4 / 27.9 / 29.7 ms, 1200 bps (four pairs) 15.2 / 14.1 / 16.3 ms
                     // per step -- half the code for nine more block addresses per sample (profiles/r6_msk600_trace.md)
#endif
template <int LDSN> constexpr int mfb_bs() { return (LDSN % 8 == 0 && MFB_BS_MAX >= 8) ? 8 : (MFB_BS_MAX >= 4 ? 4 : 2); }
template <int FIRN, int LDSN, int D, int T0, int BV, int TAILA, int NBA>
__device__ __forceinline__ void mfb_fir_continue_v(const double *lre, const double *ltap, const double (&tre)[TAILA], const double (&tim)[TAILA],
                                                   const int (&blk)[NBA], int tapz, double are0, double aim0, double &ore, double &oim)
{
    constexpr int NT = FIRN - T0, TAILN = NT - LDSN, BS = mfb_bs<LDSN>();
    static_assert(LDSN % BS == 0 && NBA == LDSN / BS + 1, "ring blocks");
    double pr[D], pi[D], pt[D];
    auto fetch = [&](int s, int q) __attribute__((always_inline)) {
        pt[q] = ltap[tapz + T0 + s];
        if (s >= TAILN)
        {
            const int k = s - TAILN, m = (BV + k) / BS, r = (BV + k) % BS;
            pr[q] = lre[blk[m] + r * 64];
            pi[q] = lre[blk[m] + r * 64 + LDSN * 64]; // the other arm's ring follows this one (MfbLds)
        }
    };
#pragma unroll
    for (int s = 0; s < D; s++) fetch(s, s);
    __builtin_amdgcn_sched_barrier(0);
    double are = are0, aim = aim0;
#pragma unroll
    for (int s = 0; s < NT; s++)
    {
        const int q = s % D;
        const double xr = (s < TAILN) ? tre[(TAILN - 1 - s) < 0 ? 0 : (TAILN - 1 - s)] : pr[q];
        const double xi = (s < TAILN) ? tim[(TAILN - 1 - s) < 0 ? 0 : (TAILN - 1 - s)] : pi[q];
        are = are + pt[q] * xr;
        aim = aim + pt[q] * xi;
        asm volatile("" : "+v"(are), "+v"(aim)); // keeps the software pipeline as written (see jd_fir_eval)
        if (s + D < NT) fetch(s + D, q);
        __builtin_amdgcn_sched_barrier(0);
    }
    ore = are; oim = aim;
}

// ------------------------------------------------------------------------------------------------------------------ front half
template <int FIRN, int LDSN, bool EBNO, int TB>
__device__ __forceinline__ void mfb_front(const JGeom &g, const JPtrs &p, const MfbLds &L, const int16_t *__restrict__ pcm, int pcm_stride, int n,
                                          int skip_a_first, int only_a_last, int fir_slot0, int grp, int lane)
{
    constexpr int TAILN = FIRN - LDSN - TB; // this half's register tail; the TB oldest entries are the back half's
    double tre[TAILN], tim[TAILN]; // tre[j] = x_re[n-LDSN-j] once x[n] has been pushed
    const int ch = grp * 64 + lane;
    const int nchp = g.nchp;
    const bool live = ch < g.nch;
    const double2 *__restrict__ cis = p.cis;
    const int nB = n - (only_a_last ? 1 : 0); // samples whose B-part runs in this launch

    double mc_ptr = LDF(S_MC_PTR), mc_step = LDF(S_MC_STEP);
    double agc_sum = LDF(S_AGC_SUM);
    double eb_esum = LDF(S_EB_ESUM), eb_e2sum = LDF(S_EB_E2SUM), eb_ebno = LDF(S_EB_EBNO);
    int agc_pos = LDI(I_AGC_POS), bb_ptr = LDI(I_BB_PTR), coarse_cnt = LDI(I_COARSE_CNT);
    const int flags = LDI(I_FLAGS);
    const int nfft_mask = g.nfft - 1;
    double2 *__restrict__ bbring = p.bbring + (size_t)ch * g.nfft;
    // ONE ring of |sig2| values (JPtrs::win, jaero_device.h): the AGC's moving-average buffer, the EbNo meter's E buffer, and -- squared -- its
    // E2 buffer; written once per sample at agc_pos, read at the two window lengths behind it
    double *__restrict__ win = p.win + (size_t)grp * g.win_len * 64 + lane;
    auto wslot = [&](int pos, int lag) { const int q = pos - lag; return q < 0 ? q + g.win_len : q; };

    // coarse ring fill, four entries at a time as one complete 64-byte sector (see k_oqpsk_fb.h)
    double2 cq1 = make_double2(0.0, 0.0), cq2 = cq1, cq3 = cq1;
    int cq_n = 0;
    auto ring_fill = [&](const double2 v) __attribute__((always_inline)) {
        if ((bb_ptr & 3) == 3)
        {
            double2 *dst = bbring + bb_ptr;
            if (cq_n >= 3) dst[-3] = cq3;
            if (cq_n >= 2) dst[-2] = cq2;
            if (cq_n >= 1) dst[-1] = cq1;
            dst[0] = v;
            cq_n = 0;
        }
        else
        {
            cq3 = cq2; cq2 = cq1; cq1 = v;
            cq_n++;
        }
        bb_ptr = (bb_ptr + 1) & nfft_mask;
    };
    auto ring_flush = [&]() __attribute__((always_inline)) {
        double2 *dst = bbring + bb_ptr;
        if (cq_n >= 3) dst[-3] = cq3;
        if (cq_n >= 2) dst[-2] = cq2;
        if (cq_n >= 1) dst[-1] = cq1;
        cq_n = 0;
    };

    double *lre = L.lre, *lim = L.lim, *ltap = L.ltap;
    {
        const double *fs = p.firsave + (size_t)grp * 2 * FIRN * 64 + lane;
        for (int k = 0; k < LDSN; k++)
        {
            lre[k * 64 + lane] = fs[(size_t)k * 64];
            lim[k * 64 + lane] = fs[(size_t)(FIRN + k) * 64];
        }
#pragma unroll
        for (int j = 0; j < TAILN; j++)
        {
            tre[j] = fs[(size_t)(LDSN + j) * 64];
            tim[j] = fs[(size_t)(FIRN + LDSN + j) * 64];
        }
        for (int k = lane; k < FIRN; k += 64) ltap[k] = p.taps2[k];
    }
    int fir_slot = fir_slot0; // wave-uniform: LDS slot holding the oldest LDS entry, overwritten by the next input
    int tapz;                 // a zero in a vector register: tap reads become `register + immediate offset` (mfb_fir_continue_v)
    asm volatile("v_mov_b32 %0, 0" : "=v"(tapz));
    // rel = index of the sample (within this launch) whose output is formed: its partial sum from the back half sits in acc[rel & 1]
    auto fir_eval = [&](int rel, double &ore, double &oim) __attribute__((always_inline)) {
        if constexpr (TB > 0)
        {
            const double *a = L.acc + (rel & 1) * 2 * 64 + lane;
            const double a0 = a[0], a1 = a[64];
            constexpr int BS = mfb_bs<LDSN>(), NB = LDSN / BS;
            int blk[NB + 1];
            const int ab = fir_slot / BS; // wave-uniform
#pragma unroll
            for (int m = 0; m <= NB; m++)
            {
                int t = ab + m;
                if (t >= NB) t -= NB;
                if (t >= NB) t -= NB;
                blk[m] = t * BS * 64 + lane;
            }
#define MFB_V(B) case B: if constexpr (B < BS) mfb_fir_continue_v<FIRN, LDSN, 8, TB, B>(lre, ltap, tre, tim, blk, tapz, a0, a1, ore, oim); break;
            switch (fir_slot % BS)
            {
                MFB_V(0) MFB_V(1) MFB_V(2) MFB_V(3)
            default:
                if constexpr (BS == 8)
                {
                    switch (fir_slot % BS) { MFB_V(4) MFB_V(5) MFB_V(6) MFB_V(7) default: break; }
                }
                break;
            }
#undef MFB_V
        }
        else jd_fir_eval<FIRN, LDSN, 8>(lre, lim, ltap, tre, tim, fir_slot, lane, ore, oim);
    };

    const double agc_len_d = (double)g.agc_len, eb_len_d = (double)g.ebno_len;

    // MSKEbNoMeasure::Update (DSP.cpp:493-505), AGC + clip (mskdemodulator.cpp:378-382) for one sample; hands {sre, sim} to the back half
    auto front_sample = [&](double sre, double sim, double agc_old, double e_old, int j, int buf) __attribute__((always_inline)) {
        const double dabval = sqrt(sre * sre + sim * sim);
        if (EBNO)
        {
            const double sq = dabval * dabval;
            const double e2_old = e_old * e_old; // E2's buffer holds fabs(sig * sig) of the same samples (MovingAverage::Update, DSP.cpp:408-416)
            eb_e2sum = eb_e2sum - e2_old; eb_e2sum = eb_e2sum + fabs(sq);
            eb_esum = eb_esum - e_old; eb_esum = eb_esum + fabs(dabval);
            if (j >= n - JD_EBNO_TAIL) // wave-uniform; see JD_EBNO_TAIL
            {
                const double e2val = eb_e2sum / eb_len_d, mean = eb_esum / eb_len_d;
                const double var = e2val - (mean * mean);
                const double alpha = sqrt(2.0) / mean;
                double tebno = 10.0 * (log10(2.0) - log10(((var * alpha * alpha) - 0.0085))) - 5.0;
                if (isnan(tebno)) tebno = 50;
                if (tebno > 50.0) tebno = 50;
                eb_ebno = eb_ebno * 0.8 + 0.2 * tebno;
            }
        }
        {
            double *ap = win + (size_t)agc_pos * 64;
            agc_sum = agc_sum - agc_old;
            agc_sum = agc_sum + fabs(dabval);
            *ap = fabs(dabval); // the one store: the EbNo meter above pushed the same value
            agc_pos++; if (agc_pos >= g.win_len) agc_pos = 0;
        }
        double gain = jd_div(1.414213562, fmax(agc_sum / agc_len_d, 0.000001));
        gain = fmax(gain, 0.000001);
        sre *= gain; sim *= gain;
        const double abval = sqrt(sre * sre + sim * sim);
        if (abval > 2.84) { const double k = jd_div(2.84, abval); sre = k * sre; sim = k * sim; }
        double *d = L.data + buf * 2 * 64 + lane;
        d[0] = sre; d[64] = sim;
    };

    auto ring_pos_next = [](int pos, int len) { pos++; return pos >= len ? 0 : pos; };
    double r1_agc = win[(size_t)wslot(agc_pos, g.agc_len) * 64]; // the entries leaving the two windows at the next sample to be fronted
    double r1_e = 0;
    if (EBNO) r1_e = win[(size_t)wslot(agc_pos, g.ebno_len) * 64];
    short nx_pcm = (live && n > 0) ? pcm[ch] : (short)0;
    double2 nx_cc = cis[jd_cisidx(mc_ptr)];

    if constexpr (TB > 0) fb_barrier(); // the back half has published its partial sums for samples 0 and 1
    // prologue: sample 0's filter output comes from the saved history
    if (nB > 0)
    {
        double y_re, y_im;
        fir_eval(0, y_re, y_im);
        front_sample(y_re, y_im, r1_agc, r1_e, 0, 0);
        r1_agc = win[(size_t)wslot(agc_pos, g.agc_len) * 64];
        if (EBNO) r1_e = win[(size_t)wslot(agc_pos, g.ebno_len) * 64];
    }
    fb_barrier();

    FB_TRACE_DECL; // phase trace build (k_oqpsk_fb.h): empty in the product
    for (int i = 0; i < nB; i++)
    {
        FB_TRACE(0); // the barrier
        // the carrier NCO's table value for sample i (index handed over by the back half): an L2 hit; the register half of the history
        // shifts meanwhile
        const int m2i = L.idx[(i & 1) * 64 + lane];
        const double2 c_m2 = cis[m2i];
#pragma unroll
        for (int j = TAILN - 1; j > 0; j--) { tre[j] = tre[j - 1]; tim[j] = tim[j - 1]; }
        double *hre = lre + fir_slot * 64 + lane, *him = lim + fir_slot * 64 + lane;
        tre[0] = *hre;
        tim[0] = *him;
        const short s = nx_pcm;
        const double dval = ((double)s) / 32768.0;
        const double2 cc = nx_cc;
        const bool do_fill = !(i == 0 && skip_a_first) && ((coarse_cnt >= g.Fs_int) || !(flags & JF_CPUREDUCE));
        __builtin_amdgcn_sched_barrier(0);
        {
            const double cre = c_m2.x * dval, cim = c_m2.y * dval; // x[n], mixed with the carrier phase this sample started with
            *hre = cre;
            *him = cim;
            fir_slot++;
            if (fir_slot >= LDSN) fir_slot = 0;
        }
        if constexpr (TB > 0)
        {
            // the entry that leaves this tail two samples on is the newest one of the partial sum the back half forms during the next sample
            double *o = L.oldx + ((i + 1) & 1) * 2 * 64 + lane;
            o[0] = tre[TAILN - 2]; o[64] = tim[TAILN - 2];
        }
        __builtin_amdgcn_sched_barrier(0);
        FB_TRACE(1); // index read, gather, mix, push, hand-over of the entry that leaves the tail
        if (do_fill) ring_fill(make_double2(cc.x * dval, cc.y * dval)); // :350-355
        coarse_cnt++;                                                   // :368
        fb_wt_next(mc_ptr, mc_step);
        if (i + 1 < n)
        {
            nx_pcm = live ? pcm[(size_t)(i + 1) * pcm_stride + ch] : (short)0;
            nx_cc = cis[jd_cisidx(mc_ptr)];
        }
        double r2_agc = 0, r2_e = 0;
        if (i + 2 < nB)
        {
            const int wn = ring_pos_next(agc_pos, g.win_len);
            r2_agc = win[(size_t)wslot(wn, g.agc_len) * 64];
            if (EBNO) r2_e = win[(size_t)wslot(wn, g.ebno_len) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        FB_TRACE(2); // coarse ring, oscillator, row requests
        if (i + 1 < nB)
        {
            double y_re, y_im;
            fir_eval(i + 1, y_re, y_im);
            FB_TRACE(3); // the matched filter (this half's share)
            front_sample(y_re, y_im, r1_agc, r1_e, i + 1, (i + 1) & 1);
            r1_agc = r2_agc; r1_e = r2_e;
            FB_TRACE(4); // EbNo sums, AGC, clip, mailbox
        }
        fb_barrier();
    }
    FB_TRACE_FLUSH(0, nB);
    if (only_a_last) // the coarse estimate runs now; the next launch resumes with this sample's B-part
    {
        const double dval = ((double)nx_pcm) / 32768.0;
        const bool do_fill = !(nB == 0 && skip_a_first) && ((coarse_cnt >= g.Fs_int) || !(flags & JF_CPUREDUCE));
        if (do_fill) ring_fill(make_double2(nx_cc.x * dval, nx_cc.y * dval));
    }
    ring_flush();

    LDF(S_MC_PTR) = mc_ptr; LDF(S_MC_STEP) = mc_step;
    LDF(S_AGC_SUM) = agc_sum;
    LDF(S_EB_ESUM) = eb_esum; LDF(S_EB_E2SUM) = eb_e2sum; LDF(S_EB_EBNO) = eb_ebno;
    LDI(I_AGC_POS) = agc_pos; LDI(I_BB_PTR) = bb_ptr; LDI(I_COARSE_CNT) = coarse_cnt;
    {
        double *fs = p.firsave + (size_t)grp * 2 * FIRN * 64 + lane;
        for (int k = 0; k < LDSN; k++)
        {
            fs[(size_t)k * 64] = lre[k * 64 + lane];
            fs[(size_t)(FIRN + k) * 64] = lim[k * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < TAILN; j++)
        {
            fs[(size_t)(LDSN + j) * 64] = tre[j];
            fs[(size_t)(FIRN + LDSN + j) * 64] = tim[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- back half
template <bool CAPSYM, int FIRN, int LDSN, int TB>
__device__ __forceinline__ void mfb_back(const JGeom &g, const JPtrs &p, const MfbLds &L, int n, int only_a_last, int dly_slot0, int d8_slot0,
                                         int grp, int lane)
{
    constexpr int TF = FIRN - LDSN - TB;
    // TB > 0: the TB oldest history entries of each arm live in this half's registers.  They do not move one place per sample (60 entries of two arms:
    // ~450 register moves a sample at 600 bps, most of them in and out of accumulation registers -- profiles/r6_msk600_trace.md): entry j (0 = the
    // newest) is at tbr[tb_s + j], a new entry goes in BELOW the others (tb_s - 1), and only when tb_s is 0 do all move up, by KL places at once.
    // tb_s is wave-uniform, the sample loop switches over it, every register index below is a compile-time one.
    constexpr int KL = TB > 0 ? MFB_LAZY_K : 1;
    constexpr int TBA = TB > 0 ? TB + KL - 1 : 1;
    double tbr[TBA], tbi[TBA];
    int tb_s = KL - 1;
    double acc_prev_re = 0, acc_prev_im = 0, acc_last_re = 0, acc_last_im = 0, xin_re = 0, xin_im = 0;
    const int ch = grp * 64 + lane;
    const int nchp = g.nchp;
    const double2 *__restrict__ cis = p.cis;
    const int nB = n - (only_a_last ? 1 : 0);

    double m2_ptr = LDF(S_M2_PTR), m2_step = LDF(S_M2_STEP), m2_freq = LDF(S_M2_FREQ);
    double st_ptr = LDF(S_ST_PTR), st_step = LDF(S_ST_STEP), st_last = LDF(S_ST_LAST);
    double res_x1 = LDF(S_RES_X1), res_x2 = LDF(S_RES_X2), res_y1 = LDF(S_RES_Y1), res_y2 = LDF(S_RES_Y2);
    double marg_sum = LDF(S_MARG_SUM), msema_sum = LDF(S_MSEMA_SUM), mse = LDF(S_MSE);
    double diff_last = LDF(S_DIFF_LAST);
    int marg_pos = LDI(I_MARG_POS), dt_pos = LDI(I_DT_POS), msema_pos = LDI(I_MSEMA_POS);
    const int flags = LDI(I_FLAGS);
    const bool dcd = flags & JF_DCD;
    int soft_cnt = LDI(I_SOFT_CNT), sym_cnt = LDI(I_SYM_CNT), overflow = LDI(I_OVERFLOW);

    const JdAtanLane atl = jd_atan_lane_table(lane); // jd_atan2's table, one entry per lane (every lane of the wavefront runs the sample loop)
    const double samplerate = g.Fs;
    double *__restrict__ marg_ring = p.marg + (size_t)ch * g.marg_len;
    double2 *__restrict__ dt_ring = p.dt + (size_t)ch * g.dt_len;
    double *__restrict__ msema_ring = p.msema + (size_t)ch * g.msema_len;
    int16_t *__restrict__ soft = p.soft + (size_t)ch * g.soft_cap;
    const int dly_len = g.sps + 1, d8_len = g.sps2 + 1;
    double2 *__restrict__ dly_ring = p.dly + (size_t)grp * dly_len * 64 + lane;
    double *__restrict__ d8_ring = p.dly8 + (size_t)grp * d8_len * 64 + lane;
    int dly_slot = dly_slot0, d8_slot = d8_slot0; // wave-uniform ring phases

    auto ring_next = [](int pos, int len) { pos++; return pos >= len ? 0 : pos; };
    // sum over this half's entries, oldest first: taps[t] <-> tb[TB - 1 - t]
    int tapz; // a zero in a vector register: the tap reads below become `register + immediate offset` instead of a v_mov of a constant address each (mfb_fir_continue_v)
    asm volatile("v_mov_b32 %0, 0" : "=v"(tapz));
    auto tail_sum = [&](auto sv, double &ore, double &oim) __attribute__((always_inline)) {
        constexpr int SV = decltype(sv)::value;
        double are = 0, aim = 0;
#pragma unroll
        for (int t = 0; t < TB; t++)
        {
            const double tp = L.ltap[tapz + t];
            are = are + tp * tbr[SV + TB - 1 - t];
            aim = aim + tp * tbi[SV + TB - 1 - t];
        }
        ore = are; oim = aim;
    };
    // a new entry joins the tail (the oldest drops out), then the sum over it; SV = tb_s before
    auto tail_push_sum = [&](auto sv, double xr, double xi, double &ore, double &oim) __attribute__((always_inline)) {
        constexpr int SV = decltype(sv)::value;
        constexpr int SN = SV == 0 ? KL - 1 : SV - 1;
        if constexpr (SV == 0)
        {
#pragma unroll
            for (int j = TB - 2; j >= 0; j--) { tbr[j + KL] = tbr[j]; tbi[j + KL] = tbi[j]; }
        }
        tbr[SN] = xr; tbi[SN] = xi;
        tail_sum(std::integral_constant<int, SN>{}, ore, oim);
        tb_s = SN;
    };
    if constexpr (TB > 0)
    {
        // saved history of the group: [0, LDSN) the LDS ring, [LDSN, LDSN + TF) the front half's tail, [LDSN + TF, FIRN) this tail; the
        // partial sum for sample 0 was formed two samples ago (the launch before): kept in the state; the one for sample 1 is the sum
        // over the tail as saved; the entry arriving during sample 0 is what the front half would have handed over: its tail entry TF - 2
        const double *fs = p.firsave + (size_t)grp * 2 * FIRN * 64 + lane;
#pragma unroll
        for (int j = 0; j < TB; j++) { tbr[KL - 1 + j] = fs[(size_t)(LDSN + TF + j) * 64]; tbi[KL - 1 + j] = fs[(size_t)(FIRN + LDSN + TF + j) * 64]; }
        xin_re = fs[(size_t)(LDSN + TF - 2) * 64]; xin_im = fs[(size_t)(FIRN + LDSN + TF - 2) * 64];
        acc_prev_re = LDF(S_MFB_A0_RE); acc_prev_im = LDF(S_MFB_A0_IM);
        // the front half fills its LDS copy of the taps before its first barrier, this half reads them behind it: use the bank's table here
        {
            double are = 0, aim = 0;
#pragma unroll
            for (int t = 0; t < TB; t++)
            {
                const double tp = p.taps2[t];
                are = are + tp * tbr[KL - 1 + TB - 1 - t];
                aim = aim + tp * tbi[KL - 1 + TB - 1 - t];
            }
            acc_last_re = are; acc_last_im = aim;
        }
        double *a = L.acc + lane;
        a[0] = acc_prev_re; a[64] = acc_prev_im;             // sample 0
        a[128] = acc_last_re; a[128 + 64] = acc_last_im;     // sample 1
    }
    // The output half of a symbol -- marg / dt / rotation by the averaged error / MSE / soft bits (mskdemodulator.cpp:428-469) -- feeds nothing
    // back into the loops, so it is queued at the instant and run for all lanes together every MFB_DEFER samples (round 4; k_oqpsk_fb has
    // done this since round 2).  With 40 samples per symbol some lane of 64 is at an instant in 80 % of the samples: run in place, the
    // ~700 instructions of this half (two window divisions, cos, sin, the differential decode) ran in 80 % of them for one lane in 64.  A lane
    // has at most one symbol queued (instants are ~SPS >= 20 samples apart, the queue empties every 16); one about to queue a second goes first.
    // The entries leaving the three windows are requested one sample after the instant and used when the queue empties.
    constexpr int MFB_DEFER = 32; // < 2 * Fs / fb = 40 samples between a lane's symbols (80 at 600 bps): the one-deep queue never overflows (16 until round 5)
    bool pend = false, need_px = false;
    double pd_ec = 0, pd_re = 0, pd_im = 0, px_marg = 0, px_ms = 0;
    double2 px_dt = make_double2(0.0, 0.0);
    auto request_px = [&]() __attribute__((always_inline)) {
        px_marg = marg_ring[marg_pos];
        int dn = dt_pos + 1; if (dn >= g.dt_len) dn = 0;
        px_dt = dt_ring[dn]; // dt_len = SPS/2 + 1 > 1
        px_ms = msema_ring[msema_pos];
        need_px = false;
    };
    auto output_half = [&]() __attribute__((always_inline)) {
        const double ct_ec = pd_ec;
        double q_re = pd_re, q_im = pd_im;
        {
            const double v = ct_ec / 2.0;
            double *mp = marg_ring + marg_pos;
            marg_sum = marg_sum - px_marg; marg_sum = marg_sum + v; *mp = v;
            marg_pos++; if (marg_pos >= g.marg_len) marg_pos = 0;
        }
        const double marg_val = marg_sum / ((double)g.marg_len);
        {
            dt_ring[dt_pos] = make_double2(q_re, q_im);
            dt_pos++; if (dt_pos >= g.dt_len) dt_pos = 0;
            q_re = px_dt.x; q_im = px_dt.y;
        }
        {
            const double cr = cos(marg_val), sr = sin(marg_val);
            const double nr = q_re * cr - q_im * sr;
            const double ni = q_re * sr + q_im * cr;
            q_re = nr; q_im = ni;
        }
        {
            const double tda = (fabs(q_re * 0.75) - 1.0), tdb = (fabs(q_im * 0.75) - 1.0);
            const double e = (tda * tda) + (tdb * tdb);
            double *ep = msema_ring + msema_pos;
            msema_sum = msema_sum - px_ms; msema_sum = msema_sum + fabs(e); *ep = fabs(e);
            msema_pos++; if (msema_pos >= g.msema_len) msema_pos = 0;
            mse = msema_sum / ((double)g.msema_len);
        }
        if (CAPSYM)
        {
            if (sym_cnt < g.sym_cap)
            {
                double *sp = p.sym + ((size_t)ch * g.sym_cap + sym_cnt) * 3;
                sp[0] = q_re; sp[1] = q_im; sp[2] = mse;
                sym_cnt++;
            }
            else overflow |= 2;
        }
        // soft differential decode + demap (:450-469, DSP.cpp:531-563)
        int b0, b1;
        {
            double soft_in = q_im, r;
            if (soft_in < 0 && diff_last < 0) r = diff_last;
            else if (soft_in > 0 && diff_last > 0) r = -diff_last;
            else r = fabs(diff_last);
            diff_last = soft_in;
            b0 = jd_softbit((r) * 127.0 + 128.0);
            soft_in = q_re;
            if (soft_in < 0 && diff_last < 0) r = diff_last;
            else if (soft_in > 0 && diff_last > 0) r = -diff_last;
            else r = fabs(diff_last);
            diff_last = soft_in;
            r = -r;
            b1 = jd_softbit((r) * 127.0 + 128.0);
        }
        if (soft_cnt + 2 <= g.soft_cap)
        {
            soft[soft_cnt] = (int16_t)b0;
            soft[soft_cnt + 1] = (int16_t)b1;
            soft_cnt += 2;
        }
        else overflow |= 1;
        pend = false;
    };
    // mailbox: the table index of mixer2 for sample 0
    L.idx[lane] = jd_cisidx(m2_ptr);
    double2 nx_cst = cis[jd_cisidx(st_ptr)];
    double2 nx_ptd = dly_ring[(size_t)ring_next(dly_slot, dly_len) * 64]; // slot read after this sample's write to dly_slot
    double nx_d8 = d8_ring[(size_t)ring_next(d8_slot, d8_len) * 64];
    if constexpr (TB > 0) fb_barrier(); // partial sums of samples 0 and 1 published; the front half forms sample 0 now
    fb_barrier();

    FB_TRACE_DECL;
    for (int i = 0; i < nB; i++)
    {
        FB_TRACE(0); // the barrier
        const double2 c_st = nx_cst;
        const double2 ptd = nx_ptd;
        const double d8out = nx_d8;
        const double *d = L.data + (i & 1) * 2 * 64 + lane;
        const double sre = d[0], sim = d[64];
        if constexpr (TB > 0)
        {
            // the entry the front half announced during the previous sample joins this tail; the sum over it is what the front half
            // starts from when it forms sample i + 2 (during sample i + 1)
            if (i > 0) { const double *o = L.oldx + (i & 1) * 2 * 64 + lane; xin_re = o[0]; xin_im = o[64]; }
            acc_prev_re = acc_last_re; acc_prev_im = acc_last_im;
            switch (tb_s)
            {
// (the marker keeps the optimiser from sinking the cases' common tails into one block that indexes the registers through a phi, i.e. through scratch memory)
#define MFB_TC(V) case V: if constexpr (V < KL) { tail_push_sum(std::integral_constant<int, V>{}, xin_re, xin_im, acc_last_re, acc_last_im); asm volatile("; tail phase %0" ::"n"(V)); } break;
                MFB_TC(0) MFB_TC(1) MFB_TC(2) MFB_TC(3) MFB_TC(4) MFB_TC(5) MFB_TC(6) MFB_TC(7)
#undef MFB_TC
            }
            double *a = L.acc + (i & 1) * 2 * 64 + lane; // (i + 2) & 1
            a[0] = acc_last_re; a[64] = acc_last_im;
        }
        if (i + 1 < nB)
        {
            nx_ptd = dly_ring[(size_t)ring_next(ring_next(dly_slot, dly_len), dly_len) * 64]; // dly_len, d8_len >= 3
            nx_d8 = d8_ring[(size_t)ring_next(ring_next(d8_slot, d8_len), d8_len) * 64];
        }

        FB_TRACE(1); // mailbox read, this half's share of the matched filter (the TB oldest terms), ring requests
        // pt_d = delayedsmpl.update_dont_touch(sig2) (:384): SPS-sample delay on a ring of SPS+1
        {
            dly_ring[(size_t)dly_slot * 64] = make_double2(sre, sim);
            dly_slot++; if (dly_slot >= dly_len) dly_slot = 0; // ptd = the entry at the new dly_slot, requested one iteration ago
        }
        double q_re = sre, q_im = ptd.y; // pt_msk

        // symbol timing (:387-405)
        double st_eta;
        {
            const double x0 = jd_hypot(q_re, q_im);
            double y = 0;
            y += res_x2 * g.res_b2; y += res_x1 * g.res_b1; y += x0 * g.res_b0;
            y -= res_y2 * g.res_a2; y -= res_y1 * g.res_a1;
            res_x2 = res_x1; res_x1 = x0; res_y2 = res_y1; res_y1 = y;
            st_eta = y;
        }
        {
            // Delay<double>(SPS/2): integer delay, weighting 0 -> returns x[n-SPS/2] (d8out, requested one iteration ago)
            d8_ring[(size_t)d8_slot * 64] = st_eta;
            d8_slot++; if (d8_slot >= d8_len) d8_slot = 0;
        }
        {
            const double2 so = c_st;
            const double m_re = st_eta, m_im = -d8out;
            const double o_re = so.x * m_re - so.y * m_im;
            const double o_im = so.x * m_im + so.y * m_re;
            const double st_angle_error = jd_atan2(o_im, o_re, atl);
            const double weighting = fabs(jd_tanh(st_angle_error));
            if (!dcd) jd_wt_advance_fraction(st_ptr, -(1.0 - weighting) * st_angle_error * (0.05 / 360.0));
            else jd_wt_advance_fraction(st_ptr, -(1.0 - weighting) * st_angle_error * (0.003 / 360.0));
        }

        FB_TRACE(2); // symbol timing: hypot, resonator, atan2, tanh, oscillator nudge
        if (need_px) request_px(); // for the symbol queued in the previous sample
        double frac;
        const bool inst = jd_wt_passed(st_last, st_ptr, st_step, g.ee, frac);
        if (pend && (inst || (i & (MFB_DEFER - 1)) == 0)) output_half();
        FB_TRACE(5); // requests, instant test, queued output halves
        if (inst)
        {
            // carrier tracking (:411-426): the half of the symbol that feeds back
            const double ct_xt = jd_tanh(sim) * sre;
            const double ct_xt_d = jd_tanh(ptd.x) * ptd.y;
            double ct_ec = ct_xt_d - ct_xt;
            if (ct_ec > M_PI) ct_ec = M_PI;
            if (ct_ec < -M_PI) ct_ec = -M_PI;
            if (ct_ec > M_PI_2) ct_ec = M_PI_2;
            if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
            double carrier_aggression = 12.0 * g.correctionfactor;
            if (dcd) carrier_aggression = 8.0 * g.correctionfactor;
            jd_wt_inc_phase_deg(m2_ptr, carrier_aggression * 1.0 * ct_ec);
            jd_wt_setfreq(m2_freq, m2_step, (carrier_aggression * 0.01 * ct_ec) + m2_freq, samplerate);
            pend = true; need_px = true; pd_ec = ct_ec; pd_re = q_re; pd_im = q_im;
        }

        // advance the NCOs (:480-483) and hand the next sample's carrier table index to the front half
        jd_wt_next(m2_ptr, m2_step);
        L.idx[((i + 1) & 1) * 64 + lane] = jd_cisidx(m2_ptr);
        if (st_step < 0) st_step = 0;
        st_last = st_ptr;
        st_ptr += st_step;
        while (((int)st_ptr) >= JD_WTSIZE) st_ptr -= JD_WTSIZE;
        nx_cst = cis[jd_cisidx(st_ptr)]; // the symbol NCO's table value for the next sample: in flight across the barrier
        FB_TRACE(4); // instant block (two tanh, carrier phase and frequency), oscillators, hand-back
        fb_barrier();
    }
    FB_TRACE_FLUSH(1, nB);
    if (need_px) request_px();
    if (pend) output_half();

    LDF(S_M2_PTR) = m2_ptr; LDF(S_M2_STEP) = m2_step; LDF(S_M2_FREQ) = m2_freq;
    LDF(S_ST_PTR) = st_ptr; LDF(S_ST_STEP) = st_step; LDF(S_ST_LAST) = st_last;
    LDF(S_RES_X1) = res_x1; LDF(S_RES_X2) = res_x2; LDF(S_RES_Y1) = res_y1; LDF(S_RES_Y2) = res_y2;
    LDF(S_MARG_SUM) = marg_sum; LDF(S_MSEMA_SUM) = msema_sum; LDF(S_MSE) = mse;
    LDF(S_DIFF_LAST) = diff_last;
    LDI(I_MARG_POS) = marg_pos; LDI(I_DT_POS) = dt_pos; LDI(I_MSEMA_POS) = msema_pos;
    LDI(I_SOFT_CNT) = soft_cnt; LDI(I_SYM_CNT) = sym_cnt; LDI(I_OVERFLOW) = overflow;
    if constexpr (TB > 0)
    {
        // next launch: its sample 0 starts from the sum formed for sample nB (acc_prev after nB steps), its tail is this one
        LDF(S_MFB_A0_RE) = acc_prev_re; LDF(S_MFB_A0_IM) = acc_prev_im;
        double *fs = p.firsave + (size_t)grp * 2 * FIRN * 64 + lane;
        switch (tb_s)
        {
#define MFB_TS(V) case V: if constexpr (V < KL) { _Pragma("unroll") for (int j = 0; j < TB; j++) { fs[(size_t)(LDSN + TF + j) * 64] = tbr[V + j]; fs[(size_t)(FIRN + LDSN + TF + j) * 64] = tbi[V + j]; } } break;
            MFB_TS(0) MFB_TS(1) MFB_TS(2) MFB_TS(3) MFB_TS(4) MFB_TS(5) MFB_TS(6) MFB_TS(7)
#undef MFB_TS
        }
    }
}

// PAIRS front/back pairs per workgroup: waves 0..PAIRS-1 are the front halves of channel groups blockIdx.x*PAIRS + w, waves
// PAIRS..2*PAIRS-1 the back halves of the same groups.  A pair whose group lies beyond the bank only keeps the barrier count.
template <int FIRN, int LDSN, bool EBNO, bool CAPSYM, int PAIRS, int TB = 0>
__global__ __launch_bounds__(PAIRS * 128) void k_msk_fb(const JGeom g, const JPtrs p, const int16_t *__restrict__ pcm, int pcm_stride, int n,
                                                        int skip_a_first, int only_a_last, int fir_slot0, int dly_slot0, int d8_slot0)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const bool back = wave >= PAIRS;
    const int pair = back ? wave - PAIRS : wave;
    const int grp = blockIdx.x * PAIRS + pair;
    double *base = lds + (size_t)pair * mfb_pair_doubles<FIRN, LDSN, TB>();
    MfbLds L;
    L.lre = base; L.lim = base + LDSN * 64; L.ltap = base + 2 * LDSN * 64;
    L.data = L.ltap + FIRN;
    L.idx = (int *)(L.data + 2 * 2 * 64);
    L.oldx = L.data + 2 * 2 * 64 + 64;
    L.acc = L.oldx + 2 * 2 * 64;
    if (grp >= g.ngroups)
    {
        const int nB = n - (only_a_last ? 1 : 0);
        for (int i = 0; i <= nB + (TB > 0 ? 1 : 0); i++) fb_barrier();
        return;
    }
    if (back) mfb_back<CAPSYM, FIRN, LDSN, TB>(g, p, L, n, only_a_last, dly_slot0, d8_slot0, grp, lane);
    else mfb_front<FIRN, LDSN, EBNO, TB>(g, p, L, pcm, pcm_stride, n, skip_a_first, only_a_last, fir_slot0, grp, lane);
}
