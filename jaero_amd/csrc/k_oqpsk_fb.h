// k_oqpsk_fb.h -- sample-loop kernel for the continuous OQPSK demodulator (10.5 kbps; 8400 bps with PRE8400), front / back wavefront pairs.
//
// Same arithmetic as k_oqpsk.h (OqpskDemodulator::writeData's per-sample loop, JAERO/oqpskdemodulator.cpp:388-605, fb > 8400), one
// channel per lane, but the per-sample work of 64 channels is shared by TWO wavefronts that run concurrently:
//
//   F ("front"):  K1 PCM -> double, K3 coarse ring fill (mixer_center), K2 mix with the carrier NCO value the back half hands over,
//                 K6 RRC matched filter (history in LDS + registers), K7 EbNo meter, K8 AGC + clip.   Owns every HBM stream
//                 (PCM, AGC / EbNo window rows, coarse ring) and the filter history; needs few registers besides the filter's.
//   B ("back"):   K9 symbol timing (delays, resonator, atan2, symbol NCO), K10 sample instant + interpolation, K11 carrier loop,
//                 K12 residual rotation, K13 MSE, K14 soft bits, the carrier NCO.   Owns the symbol-rate rings and the outputs;
//                 needs registers and no LDS.
//
// Why this split works: the matched filter's output for sample n+1 does not contain x[n+1] (FIR::FIRUpdateAndProcess excludes the
// newest sample, DSP.cpp:292-304), and x[n] = mixer2(n) * pcm[n] is known as soon as the back half has finished sample n-1.  So while
// B runs sample n, F forms x[n], pushes it and produces the AGC'd, clipped sample n+1.  One s_barrier per sample, two mailboxes in
// LDS (double buffered): B -> F the carrier table index of the next sample, F -> B {sre, sim, |.|} of the next sample.
//
// Why it pays: a wavefront issues one instruction every ~4 cycles; the single-wavefront kernel is ~1700 instructions per sample, of which
// fewer than half are fp64 VALU work.  It fills a SIMD's register file (512) and 40 KiB of LDS, so nothing else can run beside it.
// Split, each half fits 256 registers, only F needs LDS, and a 512-thread workgroup (four pairs) puts one F and one B wavefront on
// every SIMD (waves w and w+4 of a workgroup share a SIMD): the halves' instruction streams interleave, and the serial chain of
// a channel is spread over two instruction streams.  Small banks use one pair per workgroup (two SIMDs per 64 channels).
//
// Second change against k_oqpsk.h: the OUTPUT half of the symbol block (averages, residual rotation, MSE, soft bits -- nothing
// of it feeds back into the signal path) is queued per lane and run for all lanes together every FB_DEFER samples: with channels
// that are not symbol-synchronous some lane is at a symbol instant in nearly every sample, and the whole block used to run each
// time for ~5 % of the lanes.  The feedback half (tanh detector, loop filter, carrier NCO) still runs at the instant.
// Third: divisions by constants are done with the constant's reciprocal and two fma corrections (jd_div_const), bit-identical to
// the IEEE quotient; fmod(x, 360) takes the exact shortcut for |x| < 720.
#pragma once
#include "jaero_device.h"

// The queue is ONE symbol deep, so the batch interval must not exceed the distance between two symbols of a lane (48 000 / 5 250 = 9.14
// samples at 10.5 kbps, 11.4 at 8400 bps): with 16 (rounds 2-4) a lane's next symbol arrived before the batch in 43 % of the cases, took the
// "a lane about to queue a second one goes first" path below, and with 64 unsynchronised lanes that path -- the whole output half for one or two
// lanes -- ran in 95 % of the samples (the phase trace of round 5 found it: 1.03 us of the back half's 2.94 per sample).
#define FB_DEFER 8

// Phase trace of the sample loop (scripts/gpu_r5.sh trace): the trace build (make -C jaero_amd/csrc trace) reads the 100 MHz
// clock at five points of a sample in either half, every wavefront alike (so that no wavefront waits for a slower, traced one), and one pair
// adds its sums to g_fb_trace at the end of the launch; jaero_destroy prints them (a JSON line on stderr).  In the product build the macros are empty.
#ifdef FB_TRACE_BUILD
__device__ unsigned long long g_fb_trace[2][10];
#define FB_TRACE_DECL unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tr_prev = wall_clock64()
#define FB_TRACE(k)                                                                                                                                           \
    do {                                                                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                                                    \
        const unsigned long long t_ = wall_clock64();                                                                                                         \
        tr_acc[k] += t_ - tr_prev; tr_prev = t_;                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                                                    \
    } while (0)
#define FB_TRACE_FLUSH(which, nsamp)                                                                                                                          \
    do {                                                                                                                                                      \
        if (lane == 0 && grp == (g.ngroups > 600 ? 597 : 0))                                                                                                  \
        {                                                                                                                                                     \
            for (int k_ = 0; k_ < 8; k_++) atomicAdd(&g_fb_trace[which][k_], tr_acc[k_]);                                                                    \
            atomicAdd(&g_fb_trace[which][8], (unsigned long long)(nsamp));                                                                                    \
        }                                                                                                                                                     \
    } while (0)
#else
#define FB_TRACE_DECL
#define FB_TRACE(k)
#define FB_TRACE_FLUSH(which, nsamp)
#endif
#ifndef FB_SOLO_D
#define FB_SOLO_D 6 // LDS reads in flight in the one-pair kernel's filter (A/B: 12 measured in round 5, profiles/r5_small_bank.md)
#endif
#define FB_LDSN 36 // filter history slots in LDS: 36 KiB + 3.5 KiB of mailboxes = 40 448 B per pair, four pairs per CU (161 792 of 163 840 B)

// x / d for a positive constant d with rd = 1.0 / d (correctly rounded): q = x*rd is within an ulp, two Newton corrections through exact
// fma residuals give the correctly rounded quotient (Markstein); the sign of a zero result is x's.  Checked against x / d on 2e9
// random, near-multiple and near-midpoint operands per constant (scripts/div_const_check.c): no difference.
__device__ __forceinline__ double jd_div_const(double x, double d, double rd)
{
    double q = x * rd;
    double r = fma(-d, q, x);
    q = fma(r, rd, q);
    r = fma(-d, q, x);
    q = fma(r, rd, q);
    return copysign(q, x);
}
// WaveTable::SetFreq(double) with the division by the (constant) sample rate done by jd_div_const
__device__ __forceinline__ void fb_wt_setfreq(double &freq, double &step, double f, double samplerate, double r_samplerate)
{
    freq = f;
    if (freq < 0) freq = 0;
    step = jd_div_const((freq) * ((double)JD_WTSIZE), samplerate, r_samplerate);
}
// WaveTable::WTnextFrame (DSP.cpp:70-77): one step of an oscillator.  ptr < WTSIZE before, step < WTSIZE (a frequency below the sample
// rate), so the reference's `while ((int)ptr >= WTSIZE) ptr -= WTSIZE` runs at most once; written as a select plus a loop that is
// never entered it costs a handful of instructions instead of a divergent loop (same result for any ptr, step).
__device__ __forceinline__ void fb_wt_next(double &ptr, double &step)
{
    if (step < 0) step = 0;
    ptr += step;
    if (((int)ptr) >= JD_WTSIZE)
    {
        ptr -= JD_WTSIZE;
        while (((int)ptr) >= JD_WTSIZE) ptr -= JD_WTSIZE;
    }
}
// fmod(x, 360.0): exact by definition, so any exact evaluation gives the same bits; |x| < 720 covers every value the carrier loop
// produces (360 * ptr / 19999 + a clamped error), the general case falls back to the library
__device__ __forceinline__ double fb_fmod360(double x)
{
    const double ax = fabs(x);
    if (ax < 360.0) return x;
    if (ax < 720.0) return copysign(ax - 360.0, x);
    return fmod(x, 360.0);
}

struct FbLds
{
    double *lre, *lim;        // [LDSN][64], [LDSN][64]   (the taps are scalar kernel arguments: JTaps28)
    double *data;             // [2][3][64]  F -> B: sre, sim, abval of a sample
    int *idx;                 // [2][64]     B -> F: table index of mixer2 for a sample
};
template <int LDSN>
constexpr int fb_pair_doubles() { return 2 * LDSN * 64 + 2 * 3 * 64 + 64; }

// one LDS-only barrier per sample; waiting for the partner half alone through sequence words in LDS measured slower (13.9 against 12.8 ms)
#define FB_SYNC(L) fb_barrier()
__device__ __forceinline__ void fb_barrier()
{
    // LDS traffic of this wavefront done, then the workgroup barrier.  NOT __syncthreads(): that also drains vmcnt, i.e. every
    // HBM row requested ahead for the next sample.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------------------ front half
// PRE8400 (fb == 8400, k_pre8400.h): there is no matched filter in the loop -- the sample is the prefiltered complex value times the
// carrier NCO's value OF THE SAME SAMPLE (oqpskdemodulator.cpp:436-448), so the front half cannot run ahead of the back half: the two
// take turns (two barriers per sample).  What the split still buys there: the back half's code (queued output half, exact rewrites)
// instead of the single-wavefront kernel's, and the A-part of a sample (coarse ring fill, next inputs) under the back half's work.
template <int FIRN, int LDSN, bool EBNO, bool PRE8400, bool SOLO>
__device__ __forceinline__ void fb_front(const JGeom &g, const JPtrs &p, FbLds &L, const int16_t *__restrict__ pcm, int pcm_stride, int n,
                                         int skip_a_first, int only_a_last, int fir_slot0, int grp, int lane, const JTaps28 &tp,
                                         const double2 *__restrict__ prefilt, int pf_rows)
{
    constexpr int TAILN = FIRN - LDSN;
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
    int agc_hold = LDI(I_AGC_HOLD); // samples for which the AGC's buffer (re-created by setSettings: zeros) still returns zeros while the meter's keeps its values
    const int flags = LDI(I_FLAGS);
    const int nfft_mask = g.nfft - 1;
    double2 *__restrict__ bbring = p.bbring + (size_t)ch * g.nfft;
    // ONE ring of |sig2| values (JPtrs::win, jaero_device.h): the AGC's moving-average buffer, the EbNo meter's E buffer, and -- squared -- its
    // E2 buffer; written once per sample at agc_pos, read at the two window lengths behind it
    double *__restrict__ win = p.win + (size_t)grp * g.win_len * 64 + lane;
    auto wslot = [&](int pos, int lag) { const int q = pos - lag; return q < 0 ? q + g.win_len : q; };

    // Coarse ring fill, four entries at a time: a 16-byte store into the per-channel ring is a quarter of a 64-byte sector; issued one per
    // sample (3.5 us apart) every one of them cost the L2 a sector fill from HBM plus a sector write (measured: 23 GB written and
    // 15 GB of extra reads per 4096-sample launch for 4.3 GB of ring entries).  The last three entries wait in registers and go out with
    // the fourth, back to back, as one complete sector; what is left at the end of the launch (at most three) goes out singly.
    double2 cq1 = make_double2(0.0, 0.0), cq2 = cq1, cq3 = cq1;
    int cq_n = 0; // entries waiting (they are the ring positions just below bb_ptr)
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

    double *lre = L.lre, *lim = L.lim;
    if constexpr (!PRE8400)
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
    }
    int fir_slot = fir_slot0; // wave-uniform: LDS slot holding the oldest LDS entry, overwritten by the next input

    const double agc_len_d = (double)g.agc_len, eb_len_d = (double)g.ebno_len;
    const double r_agc_len = 1.0 / agc_len_d, r_eb_len = 1.0 / eb_len_d;

    // K7 + K8 for one sample: EbNo meter, AGC, clip; hands {sre, sim, abval} to the back half through mailbox `buf`
    // (oqpskdemodulator.cpp:458-470, DSP.cpp:729-744, :370-379); agc_old / e_old / e2_old = the rows leaving the windows
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
                const double e2val = jd_div_const(eb_e2sum, eb_len_d, r_eb_len), mean = jd_div_const(eb_esum, eb_len_d, r_eb_len);
                const double meansq = mean * mean;
                double var = e2val - (mean * mean);
                var -= (0.024709 * meansq);
                double mvr = (((g.Fs * meansq / (2.0 * g.fb * var))) * 0.13743);
                if (mvr < 0.000000001) mvr = 0.000000001;
                double tebno = 10.0 * log10(mvr);
                if (isnan(tebno)) tebno = 50;
                if (tebno > 50.0) tebno = 50;
                if (tebno < 0.0) tebno = 0;
                eb_ebno = eb_ebno * 0.8 + 0.2 * tebno;
            }
        }
        {
            if (agc_hold > 0) { agc_old = 0.0; agc_hold--; }
            double *ap = win + (size_t)agc_pos * 64;
            agc_sum = agc_sum - agc_old;
            agc_sum = agc_sum + fabs(dabval);
            *ap = fabs(dabval); // the one store: the EbNo meter above pushed the same value
            agc_pos++; if (agc_pos >= g.win_len) agc_pos = 0;
        }
        double gain = jd_div(1.414213562, fmax(jd_div_const(agc_sum, agc_len_d, r_agc_len), 0.000001)); // (jd_libm.h: the quotient's bits with 8 instructions for 11)
        gain = fmax(gain, 0.000001);
        sre *= gain; sim *= gain;
        const double abval = jd_hypot(sre, sim);
        if (abval > 2.84) { const double k = jd_div(2.84, abval); sre = k * sre; sim = k * sim; }
        double *d = L.data + buf * 3 * 64 + lane;
        d[0] = sre; d[64] = sim; d[128] = abval;
    };

    // Memory order matters here: vmcnt retires in order, so the wait for the carrier table value (an L2 hit, needed at once) also
    // waits for every older request.  All slow requests of a step -- the HBM rows leaving the AGC / EbNo windows TWO samples on, the
    // next PCM value, the coarse ring store -- are therefore issued right behind that wait, a whole step before the next one.
    auto ring_pos_next = [](int pos, int len) { pos++; return pos >= len ? 0 : pos; };
    double r1_agc = win[(size_t)wslot(agc_pos, g.agc_len) * 64]; // the entries leaving the two windows at the next sample to be fronted
    double r1_e = 0;
    if (EBNO) r1_e = win[(size_t)wslot(agc_pos, g.ebno_len) * 64];
    short nx_pcm = (live && n > 0) ? pcm[ch] : (short)0;
    double2 nx_cc = cis[jd_cisidx(mc_ptr)];

    if constexpr (PRE8400)
    {
        double2 nx_pf = (nB > 0) ? prefilt[JD_G4(0, ch, pf_rows)] : make_double2(0.0, 0.0); // (prefilt already points at this launch's first row)
        FB_SYNC(L); // the back half has published the carrier table index of sample 0
        for (int i = 0; i < nB; i++)
        {
            // sig2 = mixer2.WTCISValue() * cval_prefiltered[i], then EbNo, AGC, clip -> mailbox: the back half waits for this
            const int m2i = L.idx[(i & 1) * 64 + lane];
            const double2 c_m2 = cis[m2i];
            const double2 pf = nx_pf;
            const double sre = c_m2.x * pf.x - c_m2.y * pf.y, sim = c_m2.x * pf.y + c_m2.y * pf.x;
            front_sample(sre, sim, r1_agc, r1_e, i, i & 1);
            FB_SYNC(L);
            // under the back half's sample i: this sample's coarse ring entry (K3, :410-415) and the next sample's inputs
            const double dval = ((double)nx_pcm) / 32768.0;
            const double2 cc = nx_cc;
            const bool do_fill = !(i == 0 && skip_a_first) && ((coarse_cnt >= g.Fs_int) || !(flags & JF_CPUREDUCE));
            if (do_fill) ring_fill(make_double2(cc.x * dval, cc.y * dval));
            coarse_cnt++; // :431
            fb_wt_next(mc_ptr, mc_step);
            if (i + 1 < n)
            {
                nx_pcm = live ? pcm[(size_t)(i + 1) * pcm_stride + ch] : (short)0;
                nx_cc = cis[jd_cisidx(mc_ptr)];
                nx_pf = prefilt[JD_G4(i + 1, ch, pf_rows)];
            }
            if (i + 1 < nB)
            {
                r1_agc = win[(size_t)wslot(agc_pos, g.agc_len) * 64];
                if (EBNO) r1_e = win[(size_t)wslot(agc_pos, g.ebno_len) * 64];
            }
            FB_SYNC(L); // the back half has published the carrier table index of sample i + 1
        }
    }
    else
    {
    // prologue: sample 0's filter output comes from the saved history
    if (nB > 0)
    {
        double y_re, y_im;
        jd_fir_eval_sym<FIRN, LDSN, 6>(lre, lim, tp, tre, tim, fir_slot, lane, y_re, y_im);
        front_sample(y_re, y_im, r1_agc, r1_e, 0, 0);
        r1_agc = win[(size_t)wslot(agc_pos, g.agc_len) * 64];
        if (EBNO) r1_e = win[(size_t)wslot(agc_pos, g.ebno_len) * 64];
    }
    FB_SYNC(L);

    FB_TRACE_DECL;
    for (int i = 0; i < nB; i++)
    {
        FB_TRACE(0); // the barrier
        // the carrier NCO's table value for sample i (index handed over by the back half): an L2 hit a few hundred ns away; the
        // register half of the history shifts meanwhile.  (Summing 54 of the 55 filter terms of the next output under that latency
        // -- jd_fir_partial_static -- was measured and is SLOWER, 15.3 against 13.7 ms per step: on the shared SIMD the front half's wait
        // is where the back half gets the VALU, and a front half that computes through it only collides with the back half's densest
        // stretch.)
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
        // SOLO (one pair per workgroup, this wavefront alone on its SIMD): the 54 older terms of the next output under the gather's latency
        // (on a SIMD shared with the back half the same thing is slower -- the comment above -- so the four-pair kernel does not do it)
        double y_re = 0, y_im = 0;
        if constexpr (SOLO)
            if (i + 1 < nB) jd_fir_eval_sym_static_but_last<FIRN, LDSN, FB_SOLO_D>(lre, lim, tp, tre, tim, fir_slot, lane, y_re, y_im);
        const double cre = c_m2.x * dval, cim = c_m2.y * dval;
        {
            *hre = cre;
            *him = cim;
            fir_slot++;
            if (fir_slot >= LDSN) fir_slot = 0;
        }
        __builtin_amdgcn_sched_barrier(0);
        FB_TRACE(1); // mailbox read, gather of the carrier table value, mix, push
        if (do_fill) ring_fill(make_double2(cc.x * dval, cc.y * dval));
        coarse_cnt++; // :431
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
        FB_TRACE(2); // coarse ring entry, oscillator step, requests for the next samples' rows
        if (i + 1 < nB)
        {
            if constexpr (SOLO) { y_re = y_re + tp.t[0] * cre; y_im = y_im + tp.t[0] * cim; } // tap[54] == tap[0] (bitwise symmetric), x[n] from registers
            else jd_fir_eval_sym_static<FIRN, LDSN, 6>(lre, lim, tp, tre, tim, fir_slot, lane, y_re, y_im);
            FB_TRACE(3); // the matched filter
            front_sample(y_re, y_im, r1_agc, r1_e, i + 1, (i + 1) & 1);
            r1_agc = r2_agc; r1_e = r2_e;
            FB_TRACE(4); // EbNo sums, AGC, clip, mailbox write
        }
        FB_SYNC(L);
    }
    FB_TRACE_FLUSH(0, nB);
    } // !PRE8400
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
    LDI(I_AGC_HOLD) = agc_hold;
    if constexpr (!PRE8400)
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
template <bool CAPSYM, bool PRE8400>
__device__ __forceinline__ void fb_back(const JGeom &g, const JPtrs &p, FbLds &L, int n, int only_a_last, int grp, int lane)
{
    const int ch = grp * 64 + lane;
    const int nchp = g.nchp;
    const double2 *__restrict__ cis = p.cis;
    const int nB = n - (only_a_last ? 1 : 0);

    double m2_ptr = LDF(S_M2_PTR), m2_step = LDF(S_M2_STEP), m2_freq = LDF(S_M2_FREQ);
    double st_ptr = LDF(S_ST_PTR), st_step = LDF(S_ST_STEP), st_freq = LDF(S_ST_FREQ), st_last = LDF(S_ST_LAST);
    double d1 = LDF(S_D1);
    double d41_1 = LDF(S_D41_1), d41_2 = LDF(S_D41_2), d41_3 = LDF(S_D41_3);
    double d42_1 = LDF(S_D42_1), d42_2 = LDF(S_D42_2), d42_3 = LDF(S_D42_3);
    double d8_1 = LDF(S_D8_1), d8_2 = LDF(S_D8_2);
    double res_x1 = LDF(S_RES_X1), res_x2 = LDF(S_RES_X2), res_y1 = LDF(S_RES_Y1), res_y2 = LDF(S_RES_Y2);
    double lf_x1 = LDF(S_LF_X1), lf_x2 = LDF(S_LF_X2), lf_y1 = LDF(S_LF_Y1), lf_y2 = LDF(S_LF_Y2);
    double sig2l_re = LDF(S_SIG2L_RE), sig2l_im = LDF(S_SIG2L_IM), ptd_re = LDF(S_PTD_RE), ptd_im = LDF(S_PTD_IM);
    double ptd_th = jd_tanh(ptd_re); // kept beside ptd_re inside a launch (see the instant block); formed again here rather than stored
    double marg_sum = LDF(S_MARG_SUM), pm_sum = LDF(S_PM_SUM), msema_sum = LDF(S_MSEMA_SUM), mse = LDF(S_MSE);
    const double thresh = LDF(S_THRESH);
    int marg_pos = LDI(I_MARG_POS), dt_pos = LDI(I_DT_POS), pm_pos = LDI(I_PM_POS), msema_pos = LDI(I_MSEMA_POS);
    int yui = LDI(I_YUI), sig2l_init = LDI(I_SIG2L_INIT);
    int soft_cnt = LDI(I_SOFT_CNT), sym_cnt = LDI(I_SYM_CNT), overflow = LDI(I_OVERFLOW);

    const JdAtanLane atl = jd_atan_lane_table(lane); // jd_atan2's table, one entry per lane (every lane of the wavefront runs the loop below)
    const double samplerate = g.Fs; // WaveTable::samplerate after SetFreq(freq,(int)Fs)
    const double r_samplerate = 1.0 / samplerate;
    const double wtsize_d = (double)JD_WTSIZE, r_wtsize = 1.0 / wtsize_d, r_360 = 1.0 / 360.0;
    const double marg_len_d = (double)g.marg_len, pm_len_d = (double)g.pm_len, msema_len_d = (double)g.msema_len;
    const double r_marg_len = 1.0 / marg_len_d, r_pm_len = 1.0 / pm_len_d, r_msema_len = 1.0 / msema_len_d;
    // Symbol-rate windows: marg (MovingAverage(800)), dt (DelayThing(400)), pointmean and msema (MovingAverage(400) each) all advance
    // once per symbol pair from the same start, so ONE ring of 800 records {ct_ec, q_re, q_im, |q|, e} serves the four: the entry
    // leaving marg's window is in the record about to be overwritten, those leaving the other three in the record written 400 symbols
    // ago.  A symbol then costs two 64-byte record reads and one full-sector write; as four per-channel arrays of 8-byte entries it
    // was four sector fills and four partial writes (measured: ~15 GB of reads and ~4 GB of writes per launch for 0.6 GB of entries).
    double *__restrict__ symrec = p.symrec + (size_t)ch * JD_SYMREC_LEN * 8;
    const double w4 = g.w4, w4c = 1.0 - g.w4, w8 = g.w8, w8c = 1.0 - g.w8;

    // the output half of a symbol, queued at the instant: marg->UpdateSigned(ct_ec) .. soft bits (oqpskdemodulator.cpp:534-595).
    // pd_* = what it needs from the instant; px_* = the ring entries leaving the four windows, requested when the symbol is queued.
    // The ring entries are HBM misses; they are requested at the top of the NEXT sample, right behind the wait for the symbol
    // NCO's table value (vmcnt retires in order: requested at the instant they would sit in front of that wait one sample later).
    bool pend = false, need_px = false;
    double pd_ec = 0, pd_re = 0, pd_im = 0;
    double px_marg = 0, px_pm = 0, px_ms = 0;
    double2 px_dt = make_double2(0.0, 0.0);
    auto queue_symbol = [&](double ct_ec, double q_re, double q_im) {
        pend = true; need_px = true; pd_ec = ct_ec; pd_re = q_re; pd_im = q_im;
    };
    auto request_px = [&]() {
        px_marg = symrec[marg_pos * 8]; // written 800 symbols ago
        int o = marg_pos + JD_SYMREC_LEN / 2; if (o >= JD_SYMREC_LEN) o -= JD_SYMREC_LEN;
        const double2 *r = (const double2 *)(symrec + o * 8); // written 400 symbols ago: {ct_ec, q_re}, {q_im, |q|}, {e, -}
        const double2 r0 = r[0], r1 = r[1], r2 = r[2];
        px_dt = make_double2(r0.y, r1.x); // = what dt.update returns
        px_pm = r1.y;
        px_ms = r2.x;
        need_px = false;
    };
    auto output_half = [&]() {
        const double ct_ec = pd_ec;
        double q_re = pd_re, q_im = pd_im;
        // marg->UpdateSigned(ct_ec)
        marg_sum = marg_sum - px_marg; marg_sum = marg_sum + ct_ec;
        const double marg_val = jd_div_const(marg_sum, marg_len_d, r_marg_len);
        // dt.update(pt_qpsk)
        const double in_re = q_re, in_im = q_im;
        q_re = px_dt.x; q_im = px_dt.y;
        {
            double sr, cr;
            sincos(marg_val, &sr, &cr); // the same two values as cos() and sin() (one argument reduction, the same kernels)
            const double nr = q_re * cr - q_im * sr;
            const double ni = q_re * sr + q_im * cr;
            q_re = nr; q_im = ni;
        }
        // MSEcalc::Update (DSP.cpp:451-463)
        double av_w, e_w;
        {
            const double av = jd_hypot(q_re, q_im);
            pm_sum = pm_sum - px_pm; pm_sum = pm_sum + fabs(av); av_w = fabs(av);
            double mu = jd_div_const(pm_sum, pm_len_d, r_pm_len);
            if (mu < 0.000001) mu = 0.000001;
            const double s2 = sqrt(2.0);
            const double t_re = jd_div(s2 * q_re, mu), t_im = jd_div(s2 * q_im, mu);
            const double tda = (fabs(t_re) - 1.0), tdb = (fabs(t_im) - 1.0);
            const double e = (tda * tda) + (tdb * tdb);
            msema_sum = msema_sum - px_ms; msema_sum = msema_sum + fabs(e); e_w = fabs(e);
            mse = jd_div_const(msema_sum, msema_len_d, r_msema_len);
        }
        // this symbol's record: one complete 64-byte sector
        {
            double2 *w = (double2 *)(symrec + marg_pos * 8);
            w[0] = make_double2(ct_ec, in_re);
            w[1] = make_double2(in_im, av_w);
            w[2] = make_double2(e_w, 0.0);
            w[3] = make_double2(0.0, 0.0);
            marg_pos++; if (marg_pos >= JD_SYMREC_LEN) marg_pos = 0;
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
        if (mse < thresh)
        {
            const int b0 = jd_softbit(0.75 * q_im * 127.0 + 128.0);
            const int b1 = jd_softbit(0.75 * q_re * 127.0 + 128.0);
            if (soft_cnt + 2 <= g.soft_cap)
            {
                int16_t *sp = p.soft + (size_t)ch * g.soft_cap + soft_cnt;
                sp[0] = (int16_t)b0;
                sp[1] = (int16_t)b1;
                soft_cnt += 2;
            }
            else overflow |= 1;
        }
        pend = false;
    };

    // mailbox: the table index of mixer2 for sample 0
    L.idx[lane] = jd_cisidx(m2_ptr);
    double2 nx_cst = cis[jd_cisidx(st_ptr)];
    FB_SYNC(L);

    double m2fsum = 0; // PRE8400: mixer2_freq_sum of this launch (:447,607)
    FB_TRACE_DECL;
    for (int i = 0; i < nB; i++)
    {
        if constexpr (!PRE8400) FB_TRACE(0); // the barrier
        if constexpr (PRE8400)
        {
            FB_SYNC(L); // the front half has formed this sample with the table index published one barrier ago
            m2fsum += m2_freq;
        }
        const double2 c_st = nx_cst; // requested at the end of the previous sample
        const double *d = L.data + (i & 1) * 3 * 64 + lane;
        double sre = d[0], sim = d[64];
        const double abval = d[128];

        // ---- K9 symbol timing (:473-484) ----
        const double ab2 = abval * abval;
        const double st_diff = d1 - ab2; d1 = ab2;
        const double st_d1out = w4 * d41_2 + w4c * d41_3; d41_3 = d41_2; d41_2 = d41_1; d41_1 = st_diff;
        const double st_d2out = w4 * d42_2 + w4c * d42_3; d42_3 = d42_2; d42_2 = d42_1; d42_1 = st_d1out;
        double st_eta = (st_d2out - st_diff) * st_d1out;
        {
            double y = 0;
            y += res_x2 * g.res_b2; y += res_x1 * g.res_b1; y += st_eta * g.res_b0;
            y -= res_y2 * g.res_a2; y -= res_y1 * g.res_a1;
            res_x2 = res_x1; res_x1 = st_eta; res_y2 = res_y1; res_y1 = y;
            st_eta = y;
        }
        const double d8out = w8 * d8_1 + w8c * d8_2; d8_2 = d8_1; d8_1 = st_eta;
        {
            const double2 so = c_st;
            const double m_re = st_eta, m_im = -d8out;
            const double o_re = so.x * m_re - so.y * m_im;
            const double o_im = so.x * m_im + so.y * m_re;
            const double st_angle_error = jd_atan2(o_im, o_re, atl);
            fb_wt_setfreq(st_freq, st_step, (-st_angle_error * 0.00000001) + st_freq, samplerate, r_samplerate);
            jd_wt_advance_fraction(st_ptr, jd_div_const(-st_angle_error * 0.01, 360.0, r_360));
            if (st_freq < (g.stref_freq - 0.1)) fb_wt_setfreq(st_freq, st_step, (g.stref_freq - 0.1), samplerate, r_samplerate);
            if (st_freq > (g.stref_freq + 0.1)) fb_wt_setfreq(st_freq, st_step, (g.stref_freq + 0.1), samplerate, r_samplerate);
        }
        if constexpr (!PRE8400) FB_TRACE(1); // mailbox read, symbol timing: delays, resonator, atan2, oscillator nudges
        if (need_px) request_px(); // for the symbol queued in the previous sample
        if constexpr (!PRE8400) FB_TRACE(5); // (finer split of the third row: the record requests)

        // ---- K10..K14 at symbol instants (:487-595) ----
        if (!sig2l_init) { sig2l_re = sre; sig2l_im = sim; sig2l_init = 1; }
        double frac;
        const bool inst = jd_wt_passed(st_last, st_ptr, st_step, g.ee, frac);
        const bool full = inst && (yui == 0); // yui flips to 1 at this instant: the instant that closes a symbol pair
        if constexpr (!PRE8400) FB_TRACE(6); // (the instant test)
        // queued output halves: all lanes together every FB_DEFER samples; a lane about to queue a second one goes first
        if (pend && (full || (i & (FB_DEFER - 1)) == 0)) output_half();
        if constexpr (!PRE8400) FB_TRACE(2); // record requests, instant test, queued output halves (every 16th sample)
        if (inst)
        {
            const double pt_last = frac, pt_this = 1.0 - pt_last;
            const double pt_re = pt_this * sre + pt_last * sig2l_re;
            const double pt_im = pt_this * sim + pt_last * sig2l_im;
            yui++; yui %= 2;
            // The carrier detector needs tanh(pt_im) of this instant and tanh(ptd_re) of the instant half a symbol earlier (:509-512).  With
            // some lanes of the wavefront at either kind of instant in nearly every sample, both branches run every sample: ONE evaluation
            // here serves both -- the lanes at the earlier kind of instant take the tanh of the value they are about to keep as ptd_re and
            // keep it with it.  Same argument, same function: the same bits as evaluating it half a symbol later (~100 instructions per
            // sample less in the back half).
            const double th = jd_tanh(yui ? pt_im : pt_re);
            if (!yui) { ptd_re = pt_re; ptd_im = pt_im; ptd_th = th; }
            else
            {
                const double ct_xt = th * pt_re;
                const double ct_xt_d = ptd_th * ptd_im;
                double ct_ec = ct_xt_d - ct_xt;
                if (ct_ec > M_PI) ct_ec = M_PI;
                if (ct_ec < -M_PI) ct_ec = -M_PI;
                double lf_y;
                {
                    double y = 0;
                    y += lf_x2 * g.lf_b2; y += lf_x1 * g.lf_b1; y += ct_ec * g.lf_b0;
                    y -= lf_y2 * g.lf_a2; y -= lf_y1 * g.lf_a1;
                    lf_x2 = lf_x1; lf_x1 = ct_ec; lf_y2 = lf_y1; lf_y1 = y;
                    lf_y = y;
                }
                if constexpr (!PRE8400)
                {
                    ct_ec = lf_y;
                    if (ct_ec > M_PI_2) ct_ec = M_PI_2;
                    if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
                }
                // 8400 "works better with faster phase agility" (:526-532): the raw error moves the phase, the filtered one the frequency
                // mixer2.IncresePhaseDeg(1.0*ct_ec) (DSP.cpp:169-180)
                {
                    double phase_deg = 1.0 * ct_ec;
                    phase_deg += jd_div_const(360.0 * m2_ptr, wtsize_d, r_wtsize);
                    phase_deg = fb_fmod360(phase_deg);
                    while (phase_deg < 0) phase_deg += 360.0;
                    m2_ptr = jd_div_const(phase_deg, 360.0, r_360) * wtsize_d;
                }
                if constexpr (PRE8400) fb_wt_setfreq(m2_freq, m2_step, (0.5 * 0.01 * lf_y) + m2_freq, samplerate, r_samplerate);
                else fb_wt_setfreq(m2_freq, m2_step, (0.01 * ct_ec) + m2_freq, samplerate, r_samplerate);
                queue_symbol(ct_ec, pt_re, ptd_im);
            }
        }
        sig2l_re = sre; sig2l_im = sim;
        if constexpr (!PRE8400) FB_TRACE(3); // the instant block: interpolation, tanh, loop filter, carrier phase and frequency

        // ---- advance the NCOs (:600-603) and hand the next sample's carrier table index to the front half ----
        fb_wt_next(m2_ptr, m2_step);
        L.idx[((i + 1) & 1) * 64 + lane] = jd_cisidx(m2_ptr);
        if (st_step < 0) st_step = 0;
        st_last = st_ptr;
        st_ptr += st_step;
        if (((int)st_ptr) >= JD_WTSIZE)
        {
            st_ptr -= JD_WTSIZE;
            while (((int)st_ptr) >= JD_WTSIZE) st_ptr -= JD_WTSIZE;
        }
        nx_cst = cis[jd_cisidx(st_ptr)]; // the symbol NCO's table value for the next sample: in flight across the barrier
        if constexpr (!PRE8400) FB_TRACE(4); // oscillators advance, hand-back of the carrier table index
        FB_SYNC(L);
    }
    if constexpr (!PRE8400) FB_TRACE_FLUSH(1, nB);
    if (need_px) request_px();
    if (pend) output_half();

    LDF(S_M2_PTR) = m2_ptr; LDF(S_M2_STEP) = m2_step; LDF(S_M2_FREQ) = m2_freq;
    LDF(S_ST_PTR) = st_ptr; LDF(S_ST_STEP) = st_step; LDF(S_ST_FREQ) = st_freq; LDF(S_ST_LAST) = st_last;
    LDF(S_D1) = d1;
    LDF(S_D41_1) = d41_1; LDF(S_D41_2) = d41_2; LDF(S_D41_3) = d41_3;
    LDF(S_D42_1) = d42_1; LDF(S_D42_2) = d42_2; LDF(S_D42_3) = d42_3;
    LDF(S_D8_1) = d8_1; LDF(S_D8_2) = d8_2;
    LDF(S_RES_X1) = res_x1; LDF(S_RES_X2) = res_x2; LDF(S_RES_Y1) = res_y1; LDF(S_RES_Y2) = res_y2;
    LDF(S_LF_X1) = lf_x1; LDF(S_LF_X2) = lf_x2; LDF(S_LF_Y1) = lf_y1; LDF(S_LF_Y2) = lf_y2;
    LDF(S_SIG2L_RE) = sig2l_re; LDF(S_SIG2L_IM) = sig2l_im; LDF(S_PTD_RE) = ptd_re; LDF(S_PTD_IM) = ptd_im;
    LDF(S_MARG_SUM) = marg_sum; LDF(S_PM_SUM) = pm_sum; LDF(S_MSEMA_SUM) = msema_sum; LDF(S_MSE) = mse;
    LDI(I_MARG_POS) = marg_pos; LDI(I_DT_POS) = dt_pos; LDI(I_PM_POS) = pm_pos; LDI(I_MSEMA_POS) = msema_pos;
    LDI(I_YUI) = yui; LDI(I_SIG2L_INIT) = sig2l_init;
    LDI(I_SOFT_CNT) = soft_cnt; LDI(I_SYM_CNT) = sym_cnt; LDI(I_OVERFLOW) = overflow;
    if constexpr (PRE8400) LDF(S_PRE_FSUM) = LDF(S_PRE_FSUM) + m2fsum;
}

// PAIRS front/back pairs per workgroup: waves 0..PAIRS-1 are the front halves of channel groups blockIdx.x*PAIRS + w, waves
// PAIRS..2*PAIRS-1 the back halves of the same groups.  A pair whose group lies beyond the bank only keeps the barrier count.
template <int FIRN, int LDSN, bool EBNO, bool CAPSYM, int PAIRS, bool PRE8400 = false>
__global__ __launch_bounds__(PAIRS * 128) void k_oqpsk_fb(const JGeom g, const JPtrs p, const int16_t *__restrict__ pcm, int pcm_stride, int n,
                                                          int skip_a_first, int only_a_last, int fir_slot0, const JTaps28 tp,
                                                          const double2 *__restrict__ prefilt, int pf_rows)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const bool back = wave >= PAIRS;
    const int pair = back ? wave - PAIRS : wave;
    const int grp = blockIdx.x * PAIRS + pair;
    double *base = lds + (size_t)pair * fb_pair_doubles<LDSN>();
    FbLds L;
    L.lre = base; L.lim = base + LDSN * 64;
    L.data = base + 2 * LDSN * 64;
    L.idx = (int *)(L.data + 2 * 3 * 64);
    if (grp >= g.ngroups)
    {
        const int nB = n - (only_a_last ? 1 : 0);
        const int nbar = PRE8400 ? 2 * nB + 1 : nB + 1;
        for (int i = 0; i < nbar; i++) fb_barrier();
        return;
    }
    if (back) fb_back<CAPSYM, PRE8400>(g, p, L, n, only_a_last, grp, lane);
    else fb_front<FIRN, LDSN, EBNO, PRE8400, (PAIRS == 1 && !PRE8400)>(g, p, L, pcm, pcm_stride, n, skip_a_first, only_a_last, fir_slot0, grp, lane, tp, prefilt, pf_rows);
}
