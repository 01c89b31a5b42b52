// k_coarse.h -- FreqOffsetEstimateSlot: what the demodulators do with a coarse-frequency estimate (acquisition countdowns, AFC,
// status log), JAERO/oqpskdemodulator.cpp:629-677 and JAERO/mskdemodulator.cpp:490-519.  Called by thread 0 of the estimate kernels
// (k_coarse2.h).  (The first estimate kernel, a four-step FFT through L2 scratch, lived here until round 2; k_coarse2<13> / k_coarse4
// replaced it and the A/B switch went with it.)
#pragma once
#include "jaero_device.h"

// Thread-0 epilogue of one estimate: emptyingcountdown (coarsefreqestimate.cpp:133-135), FreqOffsetEstimateSlot
// (oqpskdemodulator.cpp:629-677 / mskdemodulator.cpp:490-519), coarseCounter reset, status row.  In two halves so that the kernel can
// request the channel's state (thirteen global loads, one round trip) BEFORE its peak search and use it behind it: with load and use
// together the whole workgroup stood at the barrier behind this epilogue for that round trip, once per estimate.
struct CoarseSlotState
{
    int emptying, flags, countdown, countdown2, nest, log_cnt;
    double mse, thr, m2_freq, m2_step, mc_freq, mc_step, ebno;
};
// Loads through the CONSTANT address space: with a wave-uniform address they become scalar loads (lgkmcnt), so waiting for them does
// not wait for the vector loads in flight around them -- as vector loads (vmcnt retires in order) thread 0 stood behind its wavefront's
// share of the next estimate's ring prefetch, and the workgroup behind thread 0, once per estimate.  Safe here: what is read was written
// by earlier launches (the scalar cache is invalidated at kernel start) or, for other channels, by this kernel.
__device__ __forceinline__ CoarseSlotState coarse_slot_load(const JGeom &g, const JPtrs &p, int ch)
{
    const int nchp = g.nchp;
    jd_cint *I = (jd_cint *)(p.I + ch);
    jd_cdouble *S = (jd_cdouble *)(p.S + ch);
    CoarseSlotState c;
    c.emptying = I[(size_t)I_EMPTYING * nchp]; c.flags = I[(size_t)I_FLAGS * nchp]; c.countdown = I[(size_t)I_COUNTDOWN * nchp];
    c.countdown2 = I[(size_t)I_COUNTDOWN2 * nchp]; c.nest = I[(size_t)I_NEST * nchp]; c.log_cnt = I[(size_t)I_LOG_CNT * nchp];
    c.mse = S[(size_t)S_MSE * nchp]; c.thr = S[(size_t)S_THRESH * nchp];
    c.m2_freq = S[(size_t)S_M2_FREQ * nchp]; c.m2_step = S[(size_t)S_M2_STEP * nchp];
    c.mc_freq = S[(size_t)S_MC_FREQ * nchp]; c.mc_step = S[(size_t)S_MC_STEP * nchp];
    c.ebno = S[(size_t)S_EB_EBNO * nchp];
    return c;
}
// The same thirteen values through ordinary (vector) loads, for a caller that requests them long before it needs them: every lane reads
// the same addresses (one request per instruction), the values arrive in order with the loads around them (vmcnt), and no scalar load
// is in flight while LDS results are waited for (lgkmcnt counts both; scalar loads return out of order, so any LDS wait becomes a wait
// for them too -- what made the early request through coarse_slot_load slower in round 3).
__device__ __forceinline__ CoarseSlotState coarse_slot_load_v(const JGeom &g, const JPtrs &p, int ch)
{
    const int nchp = g.nchp;
    const int *I = p.I + ch;
    const double *S = p.S + ch;
    CoarseSlotState c;
    c.emptying = I[(size_t)I_EMPTYING * nchp]; c.flags = I[(size_t)I_FLAGS * nchp]; c.countdown = I[(size_t)I_COUNTDOWN * nchp];
    c.countdown2 = I[(size_t)I_COUNTDOWN2 * nchp]; c.nest = I[(size_t)I_NEST * nchp]; c.log_cnt = I[(size_t)I_LOG_CNT * nchp];
    c.mse = S[(size_t)S_MSE * nchp]; c.thr = S[(size_t)S_THRESH * nchp];
    c.m2_freq = S[(size_t)S_M2_FREQ * nchp]; c.m2_step = S[(size_t)S_M2_STEP * nchp];
    c.mc_freq = S[(size_t)S_MC_FREQ * nchp]; c.mc_step = S[(size_t)S_MC_STEP * nchp];
    c.ebno = S[(size_t)S_EB_EBNO * nchp];
    return c;
}
// Returns 1 when the AFC recentre fired (caller then performs bigchange(): y[i]=20 and zeroes the ring).
// writer = false: evaluate only (every thread of the workgroup can know the outcome without a broadcast; one of them writes)
__device__ __forceinline__ int coarse_slot_apply(const JGeom &g, const JPtrs &p, int ch, const CoarseSlotState &c, int zmaxloc, int N, double hzperbin, double lockingbw,
                                                 const bool writer = true)
{
    const int nchp = g.nchp;
    int *I = p.I + ch;
    double *S = p.S + ch;
    // the sample rate through an opaque copy: what derives from it (Fs / 2, reciprocals) is a few instructions here, but hoisted out of a
    // persistent kernel's estimate loop it is kept -- spilled -- across the whole loop, and the reload's wait (vmcnt counts in order) stands
    // behind every vector load in flight
    double Fs = g.Fs;
    asm volatile("" : "+s"(Fs));
#define CI(f) I[(size_t)(f) * nchp]
#define CS(f) S[(size_t)(f) * nchp]
    double freq_offset_est = -((double)(zmaxloc - N / 2)) * hzperbin * 0.5;
    int emptying = c.emptying;
    if (emptying > 0) { emptying--; freq_offset_est = 0; }

    const bool afc = c.flags & JF_AFC, dcd = c.flags & JF_DCD;
    const double mse = c.mse, thr = c.thr;
    double m2_freq = c.m2_freq, m2_step = c.m2_step;
    double mc_freq = c.mc_freq, mc_step = c.mc_step;
    int countdown = c.countdown;
    bool big = false;
    if (g.kind == 1) // OQPSK
    {
        int countdown2 = c.countdown2;
        if ((mse < thr) && (!dcd))
        {
            if (countdown2 > 0) countdown2--;
            else jd_wt_setfreq(m2_freq, m2_step, mc_freq + freq_offset_est, Fs);
        }
        else countdown2 = 5;
        if (writer) CI(I_COUNTDOWN2) = countdown2;
        if ((mse > thr) && (fabs(m2_freq - (mc_freq + freq_offset_est)) > 3.0))
            jd_wt_setfreq(m2_freq, m2_step, mc_freq + freq_offset_est, Fs);
        if ((afc) && (mse < thr) && (fabs(m2_freq - mc_freq) > 3.0))
        {
            if (countdown > 0) countdown--;
            else big = true;
        }
        else countdown = 4;
    }
    else // MSK
    {
        if ((mse > thr) && (fabs(m2_freq - (mc_freq + freq_offset_est)) > 0.0))
            jd_wt_setfreq(m2_freq, m2_step, mc_freq + freq_offset_est, Fs);
        if ((afc) && (dcd) && (fabs(m2_freq - mc_freq) > 2.0))
        {
            if (countdown > 0) countdown--;
            else big = true;
        }
        else countdown = 4;
    }
    if (big)
    {
        const double lbw = lockingbw; // demodulator's lockingbw == estimator's (oqpsk passes 2*bw/2)
        jd_wt_setfreq(mc_freq, mc_step, m2_freq, Fs);
        if (mc_freq < lbw / 2.0) jd_wt_setfreq(mc_freq, mc_step, lbw / 2.0, Fs);
        if (mc_freq > (Fs / 2.0 - lbw / 2.0)) jd_wt_setfreq(mc_freq, mc_step, Fs / 2.0 - lbw / 2.0, Fs);
        emptying = 4; // coarsefreqestimate->bigchange()
    }
    if (!writer) return big ? 1 : 0;
    CI(I_EMPTYING) = emptying;
    CI(I_COUNTDOWN) = countdown;
    CS(S_M2_FREQ) = m2_freq; CS(S_M2_STEP) = m2_step;
    CS(S_MC_FREQ) = mc_freq; CS(S_MC_STEP) = mc_step;
    CI(I_COARSE_CNT) = 0; // :426 coarseCounter = 0
    const int nest = c.nest;
    if (g.flags & 2u)
    {
        const int lc = c.log_cnt;
        if (lc < g.log_cap)
        {
            double *row = p.slog + ((size_t)ch * g.log_cap + lc) * 6;
            row[0] = (double)nest; row[1] = m2_freq; row[2] = mc_freq; row[3] = mse; row[4] = c.ebno;
            row[5] = (mse > thr) ? 0.0 : 1.0;
            CI(I_LOG_CNT) = lc + 1;
        }
        else CI(I_OVERFLOW) = CI(I_OVERFLOW) | 4;
    }
    CI(I_NEST) = nest + 1;
#undef CI
#undef CS
    return big ? 1 : 0;
}
__device__ __forceinline__ int coarse_slot(const JGeom &g, const JPtrs &p, int ch, int zmaxloc, int N, double hzperbin, double lockingbw)
{
    const CoarseSlotState c = coarse_slot_load(g, p, ch);
    return coarse_slot_apply(g, p, ch, c, zmaxloc, N, hzperbin, lockingbw);
}
