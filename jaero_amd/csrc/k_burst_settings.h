// k_burst_settings.h -- setSettings on LIVE channels of a burst bank (BurstOqpskDemodulator::setSettings, JAERO/burstoqpskdemodulator.cpp:202-277;
// BurstMskDemodulator::setSettings, JAERO/burstmskdemodulator.cpp:150-325), same bit rate and sample rate.  gfx950 only.
//
// What the reference's function leaves behind is a member-by-member mixture:
//   re-created, i.e. empty      agc, agc2, the EbNo meter, bt_ma1 / mav1 (moving averages), bt_d1 / bt_ma_diff / a1 (Delay<>::setdelay refills
//                               with zeros), the Hilbert filter (JFastFir::SetKernel: no history, L zeros of latency); burst MSK also the
//                               matched filters, delayt8 and the resonator
//   restarted                   the peak detector's counters (cntdown = 2 length: no trigger for that long), tridentbuffer_ptr = 0 (the buffer
//                               fills and is CHECKED once, whatever is in the air), rotator / symboltone_averotator = 1; burst MSK: cntr = 0,
//                               mse = 10, dcd = false
//   retuned                     mixer2 (burst MSK: mixer_center too) to the new centre frequency, phase kept
//   kept, pointer back at zero  d1, d2, the peak detector's three lines, burst MSK's delayedsmpl: DelayThing::setLength (DSP.h:447-453) resizes
//                               the QVector (same length: contents untouched) and sets buffer_ptr = 0
//   untouched                   startstop, mse (burst OQPSK), the oscillators' phases, the timing chain of burst OQPSK, msema, RxDataBits
// The last-but-one group is why this is not "zero the channel's columns".  A DelayThing of sz entries whose pointer stood at P holds the last sz
// inputs in storage order; with the pointer at 0 it hands them out as storage[1], storage[2] ... -- the same samples, rotated by P -- before the
// samples written behind the call come through.  Here d1, d2 and the trident buffer are fixed lags into ONE ring of AGC'd samples (burst_device.h),
// so the ring's history is rewritten instead: with A[k] the ring's last sz entries in time order, storage[(k + P) mod sz] = A[k] (induction over
// samples; a rewrite keeps it), and reads at the unchanged lag return storage[j] once A'[j] = A[(j - P) mod sz]: a rotation in place.  d2 holds
// d1's OUTPUTS, i.e. the ring's entries one d1 lag further back: the range in front of d1's, rotated by its own pointer (the two ranges share one
// entry: d1's oldest, which d1 never reads again, is d2's newest).  The peak detector's d1 / d3 (same length, same pointer) are the whole ring of
// burst-timing values; its d2 disagrees with them about the same entries but is only read by the trigger, which setSettings locks for longer
// than d2 is long.  P = (samples since the previous setSettings) mod sz: BS_RESET_AT; delayedsmpl only advances while the burst gate is open:
// BI_GCNT (k_burst_msk_fb.h).  bt_d1's zeros would have to stand in the same ring entries d1 still needs: k_burst_front<true> (BI_BT_HOLD).
// The Hilbert filter restarts = the channel's PCM history is zeroed (k_hilbert_fft is a function of the history ring alone).
#pragma once
#include "burst_device.h"
#include "jaero_device.h"
#include "k_burst_demod.h"

struct BSetVals // the reference's Settings members a burst demodulator reads
{
    double freq_center, lockingbw, signalthreshold;
};

// elements [lo, hi) of a window that starts at ring slot `base` (ring of `len` slots, `stride` elements between slots), reversed in place
template <class T>
__device__ __forceinline__ void bs_reverse(T *col, size_t stride, int base, int len, int lo, int hi)
{
    for (int a = lo, b = hi - 1; a < b; a++, b--)
    {
        int sa = base + a; if (sa >= len) sa -= len;
        int sb = base + b; if (sb >= len) sb -= len;
        const T va = col[(size_t)sa * stride], vb = col[(size_t)sb * stride];
        col[(size_t)sa * stride] = vb; col[(size_t)sb * stride] = va;
    }
}
// W'[j] = W[(j - P) mod n] for the n-entry window at `base`
template <class T>
__device__ __forceinline__ void bs_rotate(T *col, size_t stride, int base, int len, int n, int P)
{
    if (P == 0) return;
    bs_reverse(col, stride, base, len, 0, n);
    bs_reverse(col, stride, base, len, 0, P);
    bs_reverse(col, stride, base, len, P, n);
}
__device__ __forceinline__ int bs_slot(long long t, int len) { return (int)((t + 16LL * len) % len); } // t >= -16 len

// the scalar state a burst setSettings leaves behind (see the table at the top), for channel ch of bank (g, p); appends the Plottables row
__device__ __forceinline__ void bs_apply_scalars(const BGeom &g, const BPtrs &p, int ch, const BSetVals &v, long long T0)
{
    const int nchp = g.nchp;
    const bool oq = g.kind == JAERO_KIND_BURST_OQPSK_D;
    BLDF(BS_RESET_AT) = (double)T0;
    // ---- front end ----
    BLDF(BS_AGC_SUM) = 0.0; BLDF(BS_MA1_RE) = 0.0; BLDF(BS_MA1_IM) = 0.0; BLDF(BS_MAV1_SUM) = 0.0; BLDF(BS_LASTDY) = 0.0;
    BLDI(BI_CNTDOWN) = 2 * g.PL; BLDI(BI_MAXPOSCD) = -1; BLDI(BI_TRI_PTR) = 0;
    BLDI(BI_BT_HOLD) = g.bt_lag;
    // ---- demodulator ----
    double fc = v.freq_center;
    if (fc > ((g.Fs / 2.0) - (v.lockingbw / 2.0))) fc = ((g.Fs / 2.0) - (v.lockingbw / 2.0));
    double m2_freq = BLDF(BS_M2_FREQ), m2_step = BLDF(BS_M2_STEP);
    jd_wt_setfreq(m2_freq, m2_step, fc, g.Fs); // WaveTable::SetFreq: negative -> 0, phase kept
    BLDF(BS_M2_FREQ) = m2_freq; BLDF(BS_M2_STEP) = m2_step;
    BLDF(BS_THRESH) = v.signalthreshold; BLDF(BS_LOCKINGBW) = v.lockingbw;
    BLDF(BS_AGC2_SUM) = 0.0; BLDF(BS_EB_ESUM) = 0.0; BLDF(BS_EB_E2SUM) = 0.0; BLDF(BS_EB_EBNO) = 0.0;
    BLDF(BS_SAV_RE) = 1.0; BLDF(BS_SAV_IM) = 0.0; BLDF(BS_ROT_RE) = 1.0; BLDF(BS_ROT_IM) = 0.0;
    if (oq)
    {
        BLDF(BS_A1_1) = 0.0; BLDF(BS_A1_2) = 0.0; BLDF(BS_A1_3) = 0.0; BLDF(BS_A1_4) = 0.0; BLDF(BS_A1_5) = 0.0;
        BLDI(BI_INSERTPRE) = 0;
    }
    else
    {
        BLDF(BS_MC_FREQ) = m2_freq; // mixer_center.SetFreq(freq_center) (the same clamp)
        BLDF(BS_MSE) = 10.0;
        BLDI(BI_CNTR) = 0;
        BLDF(BS_RES_X1) = 0.0; BLDF(BS_RES_X2) = 0.0; BLDF(BS_RES_Y1) = 0.0; BLDF(BS_RES_Y2) = 0.0;
        BLDI(BI_FLAGS) = BLDI(BI_FLAGS) & ~JF_DCD;
        BLDI(BI_GCNT) = 0;
        // st_osc.SetFreq(fb / 2), st_osc_half.SetFreq(fb / 2): the same values unless the bit rate changed (k_burst_carry); phases kept
        double st_freq = BLDF(BS_ST_FREQ), st_step = BLDF(BS_ST_STEP);
        jd_wt_setfreq(st_freq, st_step, g.stref_freq, g.Fs);
        BLDF(BS_ST_FREQ) = st_freq; BLDF(BS_ST_STEP) = st_step;
    }
    // emit Plottables(mixer2.GetFreqHz(), ...)
    int ev_cnt = BLDI(BI_EV_CNT), overflow = BLDI(BI_OVERFLOW);
    bd_event(g, p, ch, ev_cnt, overflow, T0, BEV_FREQ, m2_freq);
    BLDI(BI_EV_CNT) = ev_cnt; BLDI(BI_OVERFLOW) = overflow;
}

// grid (ceil(nsel / 64), NY), 64 threads: lane = channel ch_lo + 64 blockIdx.x + lane; row 0 of the grid rewrites the scalar state and rotates the
// kept lines, the other rows zero the re-created ones (every ring slot of the channel's column).
__global__ __launch_bounds__(64) void k_burst_apply_settings(const BGeom g, const BPtrs p, int ch_lo, int ch_hi, const BSetVals v, long long T0)
{
    const int ch = ch_lo + blockIdx.x * 64 + threadIdx.x, nchp = g.nchp;
    if (ch >= ch_hi) return;
    const int grp = ch >> 6, lane = ch & 63;
    const bool oq = g.kind == JAERO_KIND_BURST_OQPSK_D;
    if (blockIdx.y > 0)
    {
        const int y = blockIdx.y - 1, ny = gridDim.y - 1;
        double *agc_ring = p.agc_ring + (size_t)grp * g.agc_len * 64 + lane;
        double *ma1r = p.ma1re + (size_t)grp * g.ma1_len * 64 + lane, *ma1i = p.ma1im + (size_t)grp * g.ma1_len * 64 + lane;
        double *mav1 = p.mav1 + (size_t)grp * g.mav1_len * 64 + lane;
        double *fa = p.fa + (size_t)grp * g.fa_len * 64 + lane;
        for (int s = y; s < g.agc_len; s += ny) agc_ring[(size_t)s * 64] = 0.0;
        for (int s = y; s < g.ma1_len; s += ny) { ma1r[(size_t)s * 64] = 0.0; ma1i[(size_t)s * 64] = 0.0; }
        for (int s = y; s < g.mav1_len; s += ny) mav1[(size_t)s * 64] = 0.0;
        for (int s = y; s < g.fa_len; s += ny) fa[(size_t)s * 64] = 0.0;
        for (int s = y; s < g.hist_len; s += ny) p.pcmhist[hb_idx(s, nchp, ch)] = 0;
        if (oq)
        {
            double *ebe = p.eb_e + (size_t)grp * g.win_ring * 64 + lane; // agc2's and the EbNo meter's windows: one ring (k_burst_demod.h)
            for (int s = y; s < g.win_ring; s += ny) ebe[(size_t)s * 64] = 0.0;
        }
        else
        {
            double *ebe = p.eb_e + (size_t)ch * g.win_ring;
            double *fs = p.firsave + (size_t)ch * 2 * g.fir_n;
            double *d8 = p.dly8 + (size_t)ch * g.d8_ring, *a1 = p.a1 + (size_t)ch * g.d8_len;
            for (int s = y; s < g.win_ring; s += ny) ebe[s] = 0.0;
            for (int s = y; s < 2 * g.fir_n; s += ny) fs[s] = 0.0;
            for (int s = y; s < g.d8_ring; s += ny) d8[s] = 0.0;
            for (int s = y; s < g.d8_len; s += ny) a1[s] = 0.0;
        }
        return;
    }
    // ---- the kept lines ----
    const long long age = T0 - (long long)BLDF(BS_RESET_AT);
    {
        double *cvre = p.cvre + (size_t)grp * g.cv_len * 64 + lane, *cvim = p.cvim + (size_t)grp * g.cv_len * 64 + lane;
        const int sz1 = g.D1 + 1, sz2 = g.D2 + 1;
        const int P1 = (int)(age % sz1), P2 = (int)(age % sz2);
        const int baseA = bs_slot(T0 - sz1, g.cv_len), baseB = bs_slot(T0 - g.D1 - sz2, g.cv_len);
        // d1's oldest entry = d2's newest (ring time T0 - D1 - 1): d2's rotation ends there, d1's starts from the old value and its own result
        // for that entry is never read (storage[0] is overwritten by the first sample behind the call)
        const double sv_re = cvre[(size_t)baseA * 64], sv_im = cvim[(size_t)baseA * 64];
        bs_rotate(cvre, 64, baseB, g.cv_len, sz2, P2); // d2 holds real parts only
        const double b_re = cvre[(size_t)baseA * 64];
        cvre[(size_t)baseA * 64] = sv_re;
        bs_rotate(cvre, 64, baseA, g.cv_len, sz1, P1);
        bs_rotate(cvim, 64, baseA, g.cv_len, sz1, P1);
        cvre[(size_t)baseA * 64] = b_re;
        (void)sv_im;
        // the peak detector's d1 / d3: the whole ring of burst-timing values (bt_len = 2 PL + 1 entries = their length)
        double *bt = p.bt + (size_t)grp * g.bt_len * 64 + lane;
        bs_rotate(bt, 64, bs_slot(T0, g.bt_len), g.bt_len, g.bt_len, (int)(age % g.bt_len));
    }
    if (!oq)
    {
        // delayedsmpl (DelayThing<cpx>, SamplesPerSymbol + 1 entries): the last dly_len entries in front of the write position, rotated by where
        // the reference's pointer stood
        double2 *dly = p.dly + (size_t)ch * g.dly_ring;
        const int pos = BLDI(BI_DLY_POS);
        int base = pos - g.dly_len; if (base < 0) base += g.dly_ring;
        bs_rotate(dly, 1, base, g.dly_ring, g.dly_len, BLDI(BI_GCNT));
    }
    bs_apply_scalars(g, p, ch, v, T0);
}

// setSettings with ANOTHER bit rate on a live burst MSK bank (the same BurstMskDemodulator object serves 600 and 1200 bps): every length
// changes, so a sibling bank is created for the new rate (burst_host.h) and this kernel moves over what the reference keeps.  Its scalar
// state came across as whole columns; here the DelayThings: setLength(new) resizes the QVector -- the first min(old, new) entries stay in
// STORAGE order, new ones are zero -- and puts the pointer at zero.  storage_old[j] = A_old[(j - P) mod sz_old] (A = the old line's last sz_old
// inputs in time order, P where its pointer stood); the new bank's ring gets storage_new[j] at the place a read at its own fixed lag will
// find it, j = 1 .. sz_new - 1 (see the top of this file).  Everything else the reference re-creates is the new bank's zeros.
__global__ __launch_bounds__(64) void k_burst_carry(const BGeom og, const BPtrs op, const BGeom g, const BPtrs p, const BSetVals v, long long T0)
{
    const int ch = blockIdx.x * 64 + threadIdx.x, nchp = g.nchp;
    if (ch >= g.nchp) return;
    const int grp = ch >> 6, lane = ch & 63;
    const long long age = T0 - (long long)op.S[(size_t)BS_RESET_AT * nchp + ch];
    {
        const double *ocre = op.cvre + (size_t)grp * og.cv_len * 64 + lane, *ocim = op.cvim + (size_t)grp * og.cv_len * 64 + lane;
        double *cvre = p.cvre + (size_t)grp * g.cv_len * 64 + lane, *cvim = p.cvim + (size_t)grp * g.cv_len * 64 + lane;
        // d1 (complex)
        const int so1 = og.D1 + 1, sn1 = g.D1 + 1, P1 = (int)(age % so1);
        for (int j = 1; j < sn1; j++)
        {
            double re = 0.0, im = 0.0;
            if (j < so1)
            {
                int k = j - P1; if (k < 0) k += so1;
                const int sl = bs_slot(T0 - so1 + k, og.cv_len);
                re = ocre[(size_t)sl * 64]; im = ocim[(size_t)sl * 64];
            }
            const int dl = bs_slot(T0 - sn1 + j, g.cv_len);
            cvre[(size_t)dl * 64] = re; cvim[(size_t)dl * 64] = im;
        }
        // d2 (real parts of d1's outputs): one d1 lag further back; its newest entry lands on d1's oldest, which d1 never reads again
        const int so2 = og.D2 + 1, sn2 = g.D2 + 1, P2 = (int)(age % so2);
        for (int j = 1; j < sn2; j++)
        {
            double re = 0.0;
            if (j < so2)
            {
                int k = j - P2; if (k < 0) k += so2;
                re = ocre[(size_t)bs_slot(T0 - og.D1 - so2 + k, og.cv_len) * 64];
            }
            cvre[(size_t)bs_slot(T0 - g.D1 - sn2 + j, g.cv_len) * 64] = re;
        }
        // the peak detector's d1 / d3
        const double *obt = op.bt + (size_t)grp * og.bt_len * 64 + lane;
        double *bt = p.bt + (size_t)grp * g.bt_len * 64 + lane;
        const int sob = og.bt_len, snb = g.bt_len, Pb = (int)(age % sob);
        for (int j = 1; j < snb; j++)
        {
            double x = 0.0;
            if (j < sob)
            {
                int k = j - Pb; if (k < 0) k += sob;
                x = obt[(size_t)bs_slot(T0 - sob + k, sob) * 64];
            }
            bt[(size_t)bs_slot(T0 - snb + j, snb) * 64] = x;
        }
    }
    {
        // delayedsmpl: the new bank's gated rings all start at position 0
        const double2 *odly = op.dly + (size_t)ch * og.dly_ring;
        double2 *dly = p.dly + (size_t)ch * g.dly_ring;
        const int so = og.dly_len, sn = g.dly_len, Pd = op.I[(size_t)BI_GCNT * nchp + ch], opos = op.I[(size_t)BI_DLY_POS * nchp + ch];
        for (int j = 1; j < sn; j++)
        {
            double2 x = make_double2(0.0, 0.0);
            if (j < so)
            {
                int k = j - Pd; if (k < 0) k += so;
                int sl = opos - so + k; if (sl < 0) sl += og.dly_ring;
                x = odly[sl];
            }
            int dl = j - sn; if (dl < 0) dl += g.dly_ring;
            dly[dl] = x;
        }
        BLDI(BI_FIR_POS) = 0; BLDI(BI_AGC2_POS) = 0; BLDI(BI_EB_POS) = 0; BLDI(BI_DLY_POS) = 0; BLDI(BI_D8_POS) = 0; BLDI(BI_A1_POS) = 0;
    }
    bs_apply_scalars(g, p, ch, v, T0);
}
