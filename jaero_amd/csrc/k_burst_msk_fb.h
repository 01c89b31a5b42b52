// k_burst_msk_fb.h -- the burst MSK tracking chain as a front / back wavefront pair (the MSK half).
//
// Same function as the single-wavefront kernel k_burst_msk_demod of rounds 1-2 (JAERO/burstmskdemodulator.cpp:524-745), which it replaces.  Measured
// there (round 3, a timing-only build with the filter's evaluation compiled out; the switches left with commit history, the numbers are in DESIGN 10): the half-sine matched filter -- 80 or 160 taps, its history advancing
// per channel only while that channel's gate is open -- is 44 % of the kernel (44.6 -> 25.1 ms per 4096-sample launch of 65 536 channels),
// on a chip where that kernel keeps two of a CU's four SIMDs idle (80 KiB of filter history per wavefront, two wavefronts per CU).
//   front half (wavefront 0): the input ring, the mix with mixer2's table entry, the push into the history, the filter.  The reference's
//     FIR::FIRUpdateAndProcess returns the sum over the FIRN samples BEFORE the one being pushed, so the output sample i needs is complete
//     once sample i - 1 is in: it is formed while the back half still tracks sample i - 1.
//   back half (wavefront 1): everything else, in the order of the single-wavefront kernel -- trident verdict, sample counters, gate,
//     symbol-tone PLL, rotators, EbNo, AGC2, delays, resonator, timing, symbol instants, soft bits.
//   per sample, ONE workgroup barrier: before it the back half publishes {gate, mixer2's table entry, input gain} of sample i (mixer2 is only
//     retuned by a trident verdict, so its table entry for the next sample is requested a whole sample ahead) and
//     the front half has published the filter output for sample i; behind it the front half pushes sample i (where the gate is open) and
//     forms the output for the next gated sample while the back half tracks sample i.  Mailboxes are double-buffered on i & 1.
// LDS per pair: 2 x LDSN x 64 doubles of history + 5.5 KiB of mailboxes + 16 KiB of write-combining cells (round 4, below): LDSN = 48 of 80
// taps (two pairs per CU: all four SIMDs busy), 128 of 160 (one pair per CU); the FIRN - LDSN = 32 oldest entries of each arm sit in the
// front half's registers, shifted under the gate's exec mask.
// Round 4, the back half's window stores.  The EbNo, AGC2 and delay windows advance once per GATED sample of their own channel, so their
// positions are per lane and their rings [channel][entry]; one 8-byte store per ring and gated sample, into a 64-byte sector that the
// channel comes back to a sample later -- with 65 536 channels' sectors in flight L2 has evicted it by then, and every store cost a partial
// sector write: 61 GB per busy launch for 7 GB of entries (profiles/pmc_summary_burst_msk.json, round 3).  Now a lane's entries of the
// current cell of eight go to LDS ([ring][entry & 7][lane]) and leave as one complete sector when the cell is full.  All five windows
// (EbNo e and e^2, AGC2, delayt8, delayedsmpl) advance together, start together and have sizes that are multiples of eight (the two delay
// rings are rounded up to whole cells: a delay line needs "the entry written D pushes ago", not a ring of exactly D + 1), so one phase
// serves them all and a cell never wraps.  Entries read back (the ones leaving a window) are at least 19 pushes old: long flushed.
// Results are bit-identical to k_burst_msk_demod's (same operations in the same order); every burst-MSK bank test runs on this kernel.
// Later in round 4: the EbNo meter's E window, its E2 window and AGC2's window were three rings (24 B written and 24 B read per gated sample).
// ebnomeasure->Update(std::abs(sig2)) and agc2->Update(std::abs(sig2)) (burstmskdemodulator.cpp:634,643) push the SAME value in the same
// samples and both start from empty buffers at the same moments (constructor, setSettings), and E2's buffer holds the squares of E's
// (MovingAverage::Update stores fabs(sig), DSP.cpp:408-416).  ONE ring of max(agc2_len, eb_len) entries serves all three: 8 B written,
// 16 B read (the entries leaving the two windows), the square formed again from the entry -- the same doubles, a third of the traffic.
#pragma once
#include "k_burst_demod.h"

struct BmskMail
{
    double *out; // [2][2][64] filter output (re, im) for the sample of that parity
    double *in;  // [2][3][64] mixer2's table entry (re, im) for the sample and vol_gain
    int *gate;   // [2][64]    1 = the sample is pushed (the channel's gate is open)
    double *wc;  // [3 or 4][8][64] the back half's write-combining cells
    double *d8;  // [d8_len][64] delayt8's ring where it fits (1200 bps: 21 entries), behind the cells
};
#define BMSK_FB_MAIL_BYTES (2 * 2 * 64 * 8 + 2 * 3 * 64 * 8 + 2 * 64 * 4) // 5632
#define BMSK_FB_LDSN_80 48
#define BMSK_FB_LDSN_160 128
// cells: |sig2| (the EbNo / AGC2 window ring), delayedsmpl re / im, and delayt8 where its ring stays in HBM (600 bps: 41 entries do not fit beside
// 128 KiB of filter history); at 1200 bps delayt8's 21 entries per lane live in LDS for the launch (10.5 KiB) and cost HBM nothing per sample
#define BMSK_FB_WC_BYTES(d8lds) (((d8lds) ? 3 : 4) * 8 * 64 * 8)
#define BMSK_FB_D8_BYTES(d8lds, d8_len) ((d8lds) ? (d8_len) * 64 * 8 : 0)

template <int FIRN, int LDSN>
__device__ __forceinline__ void bmsk_front(const BGeom &g, const BPtrs &p, double *lre, double *lim, const BmskMail &M, int n, long long n0, int grp, int lane)
{
    constexpr int TAILN = FIRN - LDSN, TAILA = TAILN > 0 ? TAILN : 1;
    double tre[TAILA], tim[TAILA]; // tre[j] = x_re[newest - LDSN - j]
    const int ch = grp * 64 + lane, nchp = g.nchp;
    jd_cdouble *taps = (jd_cdouble *)p.taps2; // this bank's own half-sine taps through the constant address space: scalar loads
    int fir_pos = BLDI(BI_FIR_POS);
    const double *__restrict__ cvre = p.cvre + (size_t)grp * g.cv_len * 64 + lane;
    {
        const double *fs = p.firsave + (size_t)ch * 2 * FIRN; // [0, LDSN): the LDS ring's slots, [LDSN, FIRN): the register tail
        for (int k = 0; k < LDSN; k++) { lre[k * 64 + lane] = fs[k]; lim[k * 64 + lane] = fs[FIRN + k]; }
#pragma unroll
        for (int j = 0; j < TAILN; j++) { tre[j] = fs[LDSN + j]; tim[j] = fs[FIRN + LDSN + j]; }
    }
    int s_val = (int)((n0 - g.D1 - g.D2 + 8LL * g.cv_len) % g.cv_len);
    double nx_val = cvre[(size_t)s_val * 64];
    // output from x[n-FIRN .. n-1], taps[t] <-> x[n-FIRN+t], oldest first: the register tail, then the LDS ring from this lane's oldest slot.
    // The ring position is per lane (the history advances only while that channel's gate is open): byte offsets with a compare-free wrap
    // (a + 512 or a + 512 - ring bytes, whichever is smaller as unsigned), eight entries of each arm requested one batch ahead of the
    // sixteen fmas that consume them.
    auto evaluate = [&](double &sre, double &sim) __attribute__((always_inline)) {
        sre = 0; sim = 0;
#pragma unroll
        for (int t = 0; t < TAILN; t++)
        {
            const double tp = taps[t];
            sre = fma(tp, tre[TAILN - 1 - t], sre);
            sim = fma(tp, tim[TAILN - 1 - t], sim);
        }
        constexpr unsigned RING = (unsigned)LDSN * 512u;
        constexpr int NB = LDSN / 8;
        static_assert(LDSN % 8 == 0, "the LDS part of the history is read in batches of eight");
        unsigned a = ((unsigned)fir_pos * 64u + (unsigned)lane) * 8u;
        const char *bre = (const char *)lre, *bim = (const char *)lim;
        double xr[2][8], xi[2][8];
        auto fetch = [&](int w) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                xr[w][u] = *(const double *)(bre + a);
                xi[w][u] = *(const double *)(bim + a);
                const unsigned a2 = a + 512u;
                a = min(a2, a2 - RING);
            }
        };
        fetch(0);
#pragma unroll
        for (int b = 0; b < NB; b++)
        {
            if (b + 1 < NB) fetch((b + 1) & 1);
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                const double tp = taps[TAILN + 8 * b + u];
                sre = fma(tp, xr[b & 1][u], sre);
                sim = fma(tp, xi[b & 1][u], sim);
            }
        }
    };
    double osre, osim; // the filter output over the history as it stands
    evaluate(osre, osim);
    M.out[lane] = osre; M.out[64 + lane] = osim; // for sample 0: the history as the previous launch left it
    for (int i = 0; i < n; i++)
    {
        fb_barrier();
        const int gate = M.gate[(i & 1) * 64 + lane];
        const double *in = M.in + (i & 1) * 192 + lane;
        const double cx = in[0], cy = in[64], vg = in[128];
        const double val = nx_val;
        s_val++; if (s_val >= g.cv_len) s_val = 0;
        if (i + 1 < n) nx_val = cvre[(size_t)s_val * 64];
        if (gate)
        {
            const double cre = (cx * val) * vg, cim = (cy * val) * vg;
            // push x[n]: the oldest LDS entry moves into the register tail (under the gate's exec mask)
            if constexpr (TAILN > 0)
            {
#pragma unroll
                for (int j = TAILN - 1; j > 0; j--) { tre[j] = tre[j - 1]; tim[j] = tim[j - 1]; }
                tre[0] = lre[fir_pos * 64 + lane]; tim[0] = lim[fir_pos * 64 + lane];
            }
            lre[fir_pos * 64 + lane] = cre; lim[fir_pos * 64 + lane] = cim;
            fir_pos++; if (fir_pos >= LDSN) fir_pos = 0;
        }
        if (i + 1 < n)
        {
            // no channel of this wavefront pushed: every history is what it was, and so is every output (between bursts: most samples)
            if (__builtin_amdgcn_ballot_w64(gate != 0) != 0ull) evaluate(osre, osim);
            M.out[((i + 1) & 1) * 128 + lane] = osre; M.out[((i + 1) & 1) * 128 + 64 + lane] = osim;
        }
    }
    BLDI(BI_FIR_POS) = fir_pos;
    {
        double *fs = p.firsave + (size_t)ch * 2 * FIRN;
        for (int k = 0; k < LDSN; k++) { fs[k] = lre[k * 64 + lane]; fs[FIRN + k] = lim[k * 64 + lane]; }
#pragma unroll
        for (int j = 0; j < TAILN; j++) { fs[LDSN + j] = tre[j]; fs[FIRN + LDSN + j] = tim[j]; }
    }
}

template <bool CAPSYM, bool D8LDS>
__device__ __forceinline__ void bmsk_back(const BGeom &g, const BPtrs &p, const BmskMail &M, int n, long long n0, int first_of_write, int grp, int lane)
{
    const int ch = grp * 64 + lane, nchp = g.nchp;
    const double2 *__restrict__ cis = p.cis;
    const double SPS = g.SPS, samplerate = g.Fs;

    double m2_ptr = BLDF(BS_M2_PTR), m2_step = BLDF(BS_M2_STEP), m2_freq = BLDF(BS_M2_FREQ), mc_freq = BLDF(BS_MC_FREQ);
    double st_ptr = BLDF(BS_ST_PTR), st_last = BLDF(BS_ST_LAST), sth_ptr = BLDF(BS_STQ_PTR), vol_gain = BLDF(BS_VOL_GAIN);
    const double st_step = BLDF(BS_ST_STEP);
    double str_re = BLDF(BS_STR_RE), str_im = BLDF(BS_STR_IM), sav_re = BLDF(BS_SAV_RE), sav_im = BLDF(BS_SAV_IM);
    double rot_re = BLDF(BS_ROT_RE), rot_im = BLDF(BS_ROT_IM), rot_freq = BLDF(BS_ROT_FREQ);
    double rfs, rfc; // cis(rot_freq), formed where rot_freq changes instead of in every gated sample (k_burst_demod.h)
    sincos(rot_freq, &rfs, &rfc);
    double agc2_sum = BLDF(BS_AGC2_SUM), eb_esum = BLDF(BS_EB_ESUM), eb_e2sum = BLDF(BS_EB_E2SUM), eb_ebno = BLDF(BS_EB_EBNO);
    double res_x1 = BLDF(BS_RES_X1), res_x2 = BLDF(BS_RES_X2), res_y1 = BLDF(BS_RES_Y1), res_y2 = BLDF(BS_RES_Y2);
    double msema_sum = BLDF(BS_MSEMA_SUM), mse = BLDF(BS_MSE), diff_last = BLDF(BS_DIFF_LAST);
    const double thresh = BLDF(BS_THRESH), lockingbw = BLDF(BS_LOCKINGBW);

    int startstop = BLDI(BI_STARTSTOP), cntr = BLDI(BI_CNTR), msema_pos = BLDI(BI_MSEMA_POS), nrx = BLDI(BI_NRX);
    int eb_pos = BLDI(BI_EB_POS), dly_pos = BLDI(BI_DLY_POS), d8_pos = BLDI(BI_D8_POS), a1_pos = BLDI(BI_A1_POS); // eb_pos: write position in the window ring
    const int eb_pos0 = eb_pos;
    int soft_cnt = BLDI(BI_SOFT_CNT), sym_cnt = BLDI(BI_SYM_CNT), ev_cnt = BLDI(BI_EV_CNT), overflow = BLDI(BI_OVERFLOW);
    const int flags = BLDI(BI_FLAGS);
    const int ev_pos = BLDI(BI_EV_POS);
    const bool dcd = flags & JF_DCD, afc = flags & JF_AFC;
    const bool trace = (g.flags & 8u) != 0;
    (void)first_of_write;

    double *ebe_ring = p.eb_e + (size_t)ch * g.win_ring;
    double2 *dly_ring = p.dly + (size_t)ch * g.dly_ring;
    double *d8_ring = p.dly8 + (size_t)ch * g.d8_ring;
    // write-combining cells: this lane's column of [ring][entry & 7][lane]
    double *wc = M.wc + lane;
    enum { WC_E = 0, WC_DX = 1, WC_DY = 2, WC_D8 = 3 };
    double *d8l = M.d8 + lane; // D8LDS: entry k of this lane's delayt8 ring at d8l[k * 64]; d8_pos counts modulo d8_len there (no cells)
    auto wc_at = [&](int ring, int k) -> double & { return wc[(ring * 8 + k) * 64]; };
    // the cell being filled holds the entries pushed since its start: back from HBM (the previous launch wrote them out entry by entry)
    {
        const int ph = eb_pos & 7; // = dly_pos & 7 (= d8_pos & 7 where delayt8 goes through cells): the rings advance together
        if constexpr (D8LDS) for (int k = 0; k < g.d8_len; k++) d8l[k * 64] = d8_ring[k];
        for (int k = 0; k < ph; k++)
        {
            wc_at(WC_E, k) = ebe_ring[eb_pos - ph + k];
            if constexpr (!D8LDS) wc_at(WC_D8, k) = d8_ring[d8_pos - ph + k];
            const double2 v = dly_ring[dly_pos - ph + k];
            wc_at(WC_DX, k) = v.x; wc_at(WC_DY, k) = v.y;
        }
    }
    auto wc_flush4 = [&](int ring, double *dst) __attribute__((always_inline)) { // eight entries of one ring: one 64-byte sector
        double2 *d2 = (double2 *)dst;
#pragma unroll
        for (int k = 0; k < 4; k++) d2[k] = make_double2(wc_at(ring, 2 * k), wc_at(ring, 2 * k + 1));
    };
    double *a1_ring = p.a1 + (size_t)ch * g.d8_len;
    double *msema_ring = p.msema + (size_t)ch * g.msema_len;
    int16_t *__restrict__ soft = p.soft + (size_t)ch * g.soft_cap;
    const double agc2_len_d = (double)g.agc2_len, eb_len_d = (double)g.eb_len;

    int c2_idx = jd_cisidx(m2_ptr);
    double2 nx_c2 = cis[c2_idx];
    for (int i = 0; i < n; i++)
    {
        const long long sample = n0 + i;
        // ---- trident verdict (:524-568) ----
        if (i == ev_pos)
        {
            const TriResult tr = p.tri[ch];
            const bool ok = tr.ok && !(dcd) && !(cntr > 0 && cntr < (500 * SPS));
            if (trace) bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_TRIDENT, ok ? tr.metric : -tr.metric);
            if (ok)
            {
                vol_gain = tr.vol_gain;
                bd_set_phase_deg(m2_ptr, tr.phase_deg);
                jd_wt_setfreq(m2_freq, m2_step, tr.freq, samplerate);
                // CenterFreqChangedSlot(freq) (:327-343)
                {
                    double fc = tr.freq;
                    if (fc < (0.75 * g.fb)) fc = 0.75 * g.fb;
                    if (fc > (g.Fs / 2.0 - 0.75 * g.fb)) fc = g.Fs / 2.0 - 0.75 * g.fb;
                    mc_freq = fc; if (mc_freq < 0) mc_freq = 0;
                    if (afc) jd_wt_setfreq(m2_freq, m2_step, mc_freq, samplerate);
                    if ((m2_freq - mc_freq) > (lockingbw / 2.0)) jd_wt_setfreq(m2_freq, m2_step, mc_freq + (lockingbw / 2.0), samplerate);
                    if ((m2_freq - mc_freq) < (-lockingbw / 2.0)) jd_wt_setfreq(m2_freq, m2_step, mc_freq - (lockingbw / 2.0), samplerate);
                    bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_FREQ, m2_freq);
                }
                bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_FREQ, m2_freq);
                startstop = g.startstopstart;
                cntr = 0;
                bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_SIGNAL, 1.0);
                soft_cnt -= nrx; nrx = 0; // RxDataBits.clear()
                if (soft_cnt < g.soft_cap) { soft[soft_cnt++] = (int16_t)-1; nrx = 1; } else overflow |= 1;
                mse = 0;
                for (int k = 0; k < g.msema_len; k++) msema_ring[k] = 0;
                msema_pos = 0; msema_sum = 0;
                sav_re = 1; sav_im = 0; str_re = 1; str_im = 0;
                rot_re = 1; rot_im = 0; rot_freq = 0; rfs = 0.0; rfc = 1.0;
                res_x1 = res_x2 = res_y1 = res_y2 = 0;
                bd_set_phase_deg(st_ptr, 0);
                bd_set_phase_deg(sth_ptr, 0);
            }
        }
        // ---- sample counting (:571-598) ----
        if (startstop > 0)
        {
            if (cntr >= (g.startProcessing * SPS)) startstop--;
            if (cntr < 1000000) cntr++;
            if (mse < thresh) startstop = g.startstopstart;
        }
        if (startstop == 0)
        {
            startstop--;
            bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_SIGNAL, 0.0);
            cntr = 0;
            mse = 1;
        }
        // ---- hand-over: this sample's gate, carrier table index and input gain to the front half; its filter output for this sample back ----
        const bool gate = startstop > 0 || mse < thresh;
        {
            // mixer2's table entry for this sample was requested a sample ago; a verdict that retuned mixer2 above asks again
            const double2 c2 = (jd_cisidx(m2_ptr) == c2_idx) ? nx_c2 : cis[jd_cisidx(m2_ptr)];
            double *in = M.in + (i & 1) * 192 + lane;
            in[0] = c2.x; in[64] = c2.y; in[128] = vol_gain;
            M.gate[(i & 1) * 64 + lane] = gate ? 1 : 0;
        }
        fb_barrier();
        {
            // mixer2 advances at the end of a gated sample and only there (:736-740): its entry for the next sample can be requested now
            double m2n = m2_ptr;
            if (gate) jd_wt_next(m2n, m2_step);
            c2_idx = jd_cisidx(m2n);
            nx_c2 = cis[c2_idx];
        }
        if (gate)
        {
            // window entries this sample replaces / reads: requested now, consumed behind the filter
            // the delay rings are whole cells (g.dly_ring >= dly_len, g.d8_ring >= d8_len): "the oldest entry of a ring of dly_len" is the one
            // written dly_len - 1 pushes ago, "the one behind it" dly_len - 2 ago
            auto back = [](int pos, int lag, int ring) { const int q = pos - lag; return q < 0 ? q + ring : q; };
            // the entries leaving the EbNo and the AGC2 window: written eb_len / agc2_len gated samples ago (one of the two is the slot this
            // sample overwrites); at least 5120 pushes old: long flushed
            const double e_old = ebe_ring[back(eb_pos, g.eb_len, g.win_ring)], agc2_old = ebe_ring[back(eb_pos, g.agc2_len, g.win_ring)];
            const double e2_old = e_old * e_old;
            const double2 ptd_pre = dly_ring[back(dly_pos, g.dly_len - 1, g.dly_ring)];
            double d8_a, d8_b; // the two oldest entries of delayt8's d8_len: written d8_len - 2 and d8_len - 1 pushes ago
            if constexpr (D8LDS) { d8_a = d8l[back(d8_pos, g.d8_len - 2, g.d8_len) * 64]; d8_b = d8l[back(d8_pos, g.d8_len - 1, g.d8_len) * 64]; }
            else { d8_a = d8_ring[back(d8_pos, g.d8_len - 2, g.d8_ring)]; d8_b = d8_ring[back(d8_pos, g.d8_len - 1, g.d8_ring)]; }
            const int ph = eb_pos & 7;
            const double st_ptr_top = st_ptr;
            const double2 so_pre = cis[jd_cisidx(st_ptr)]; // the symbol oscillator's table entry: valid unless the preamble block below moves st_ptr
            double sre = M.out[(i & 1) * 128 + lane], sim = M.out[(i & 1) * 128 + 64 + lane]; // formed while the previous sample was tracked
            if (cntr > (g.startProcessing * SPS) && cntr < g.endRotation)
            {
                double t_re = sre, t_im = sim;
                bd_cmul(t_re, t_im, str_re, str_im);
                bd_cmul(t_re, t_im, 0.0, 1.0);
                const double er = jd_tanh(t_im) * (t_re);
                double sn, cs;
                sincos(er * 0.5, &sn, &cs);
                bd_cmul(str_re, str_im, cs, sn);
                sav_re = sav_re * 0.999 + 0.001 * str_re; sav_im = sav_im * 0.999 + 0.001 * str_im;
                // a1.update(): integer delay SPS/2 -> weighting 0: the value written d8_len-1 updates ago
                a1_ring[a1_pos] = t_re;
                a1_pos++; if (a1_pos >= g.d8_len) a1_pos = 0;
                t_im = 0.0 * a1_ring[(a1_pos + 1 >= g.d8_len) ? 0 : a1_pos + 1] + 1.0 * a1_ring[a1_pos];
                double progress = (double)cntr - (SPS * (g.startProcessing));
                const double goal = g.endRotation - (SPS * g.startProcessing);
                progress = progress / goal;
                const double2 cq = cis[jd_cisidx(sth_ptr)];
                const double e_re = cq.x * t_re - cq.y * (-t_im), e_im = cq.x * (-t_im) + cq.y * t_re;
                double st_err = atan2(e_im, e_re);
                st_err *= 0.5 * (1.0 - progress * progress);
                jd_wt_advance_fraction(sth_ptr, -(1.0 / (2.0 * M_PI)) * st_err * 0.05);
                bd_set_phase_deg(st_ptr, (360.0 * sth_ptr / ((double)JD_WTSIZE)) + (360.0 * (1.0 - g.ee)));
            }
            bd_cmul(sre, sim, sav_re, sav_im);
            bd_cmul(rot_re, rot_im, rfc, rfs);
            bd_cmul(sre, sim, rot_re, rot_im);
            const double sabs = hypot(sre, sim);
            {
                const double sq = sabs * sabs;
                eb_e2sum = eb_e2sum - e2_old; eb_e2sum = eb_e2sum + fabs(sq);
                eb_esum = eb_esum - e_old; eb_esum = eb_esum + fabs(sabs); wc_at(WC_E, ph) = fabs(sabs);
                // the value is observable once per burst (the emission below), at the end of a launch (status) and where the gate closes;
                // its IIR forgets a term after k samples as 0.8^k, so the arithmetic runs only in the JD_EBNO_TAIL samples before those
                const int to_emit = (g.endRotation + (int)(200 * SPS)) - cntr;
                if (i >= n - JD_EBNO_TAIL || (to_emit >= 0 && to_emit < JD_EBNO_TAIL) || startstop <= JD_EBNO_TAIL)
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
            if (cntr == g.endRotation + (200 * SPS)) bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_EBNO, eb_ebno);
            {
                agc2_sum = agc2_sum - agc2_old; agc2_sum = agc2_sum + fabs(sabs); // (the value is in the ring: the EbNo meter pushed it)
                double gain = 1.414213562 / fmax(agc2_sum / agc2_len_d, 0.000001);
                gain = fmax(gain, 0.000001);
                sre *= gain; sim *= gain;
            }
            const double abval = hypot(sre, sim);
            if (abval > 2.84) { const double k = (2.84 / abval); sre = k * sre; sim = k * sim; }
            // delayedsmpl.update_dont_touch(sig2)
            wc_at(WC_DX, ph) = sre; wc_at(WC_DY, ph) = sim;
            const double2 ptd = ptd_pre; // the oldest entry, not the one just written (dly_len >= 2)
            const double pm_re = sre, pm_im = ptd.y;
            double st_eta = hypot(pm_re, pm_im);
            {
                double y = 0;
                y += res_x2 * g.res_b2; y += res_x1 * g.res_b1; y += st_eta * g.res_b0;
                y -= res_y2 * g.res_a2; y -= res_y1 * g.res_a1;
                res_x2 = res_x1; res_x1 = st_eta; res_y2 = res_y1; res_y1 = y;
                st_eta = y;
            }
            // delayt8.update(st_eta): integer delay SPS/2
            if constexpr (D8LDS) d8l[d8_pos * 64] = st_eta; else wc_at(WC_D8, ph) = st_eta;
            const double d8out = 0.0 * d8_a + 1.0 * d8_b; // the two oldest entries of a ring of d8_len: older than the entry just written (d8_len >= 3)
            // the five windows' entries of this gated sample are in their cell; a full cell leaves as whole sectors, then all advance
            if (ph == 7)
            {
                wc_flush4(WC_E, ebe_ring + (eb_pos - 7));
                if constexpr (!D8LDS) wc_flush4(WC_D8, d8_ring + (d8_pos - 7));
                double2 *dd = dly_ring + (dly_pos - 7);
#pragma unroll
                for (int k = 0; k < 8; k++) dd[k] = make_double2(wc_at(WC_DX, k), wc_at(WC_DY, k));
            }
            eb_pos++; if (eb_pos >= g.win_ring) eb_pos = 0;
            dly_pos++; if (dly_pos >= g.dly_ring) dly_pos = 0;
            d8_pos++; if (d8_pos >= (D8LDS ? g.d8_len : g.d8_ring)) d8_pos = 0;
            {
                double2 so = so_pre;
                if (st_ptr != st_ptr_top) so = cis[jd_cisidx(st_ptr)];
                const double m_re = st_eta, m_im = -d8out;
                const double o_re = so.x * m_re - so.y * m_im, o_im = so.x * m_im + so.y * m_re;
                const double st_angle_error = atan2(o_im, o_re);
                if (cntr > g.endRotation) jd_wt_advance_fraction(st_ptr, -st_angle_error * 0.002 / 360.0);
            }
            double frac;
            if (jd_wt_passed(st_last, st_ptr, st_step, g.ee, frac))
            {
                const double ct_xt = jd_tanh(sim) * sre;
                const double ct_xt_d = jd_tanh(ptd.x) * ptd.y;
                double ct_ec = ct_xt_d - ct_xt;
                if (ct_ec > M_PI) ct_ec = M_PI;
                if (ct_ec < -M_PI) ct_ec = -M_PI;
                if (ct_ec > M_PI_2) ct_ec = M_PI_2;
                if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
                if (cntr > (g.startProcessing * SPS))
                {
                    double sn, cs;
                    sincos(ct_ec * 0.25, &sn, &cs);
                    bd_cmul(rot_re, rot_im, cs, sn);
                    if (cntr > g.endRotation) { rot_freq = rot_freq + ct_ec * 0.0001; sincos(rot_freq, &rfs, &rfc); }
                    const double tda = (fabs((pm_re * 0.75)) - 1.0), tdb = (fabs((pm_im * 0.75)) - 1.0);
                    const double e = (tda * tda) + (tdb * tdb);
                    double *mp = msema_ring + msema_pos;
                    msema_sum = msema_sum - *mp; msema_sum = msema_sum + fabs(e); *mp = fabs(e);
                    msema_pos++; if (msema_pos >= g.msema_len) msema_pos = 0;
                    mse = msema_sum / ((double)g.msema_len);
                }
                if (CAPSYM)
                {
                    if (sym_cnt < g.sym_cap) { double *sp = p.sym + ((size_t)ch * g.sym_cap + sym_cnt) * 3; sp[0] = pm_re; sp[1] = pm_im; sp[2] = mse; sym_cnt++; }
                    else overflow |= 2;
                }
                // DiffDecode::UpdateSoft x2 (DSP.cpp:531-563)
                double imagin, realv;
                {
                    const double sf = pm_im;
                    if (sf < 0 && diff_last < 0) imagin = diff_last;
                    else if (sf > 0 && diff_last > 0) imagin = -diff_last;
                    else imagin = fabs(diff_last);
                    diff_last = sf;
                }
                {
                    const double sf = pm_re;
                    if (sf < 0 && diff_last < 0) realv = diff_last;
                    else if (sf > 0 && diff_last > 0) realv = -diff_last;
                    else realv = fabs(diff_last);
                    diff_last = sf;
                }
                realv = -realv;
                const int b0 = jd_softbit((imagin) * 127.0 + 128.0);
                const int b1 = jd_softbit((realv) * 127.0 + 128.0);
                if (soft_cnt + 2 <= g.soft_cap) { soft[soft_cnt] = (int16_t)b0; soft[soft_cnt + 1] = (int16_t)b1; soft_cnt += 2; nrx += 2; }
                else overflow |= 1;
                if (nrx >= 12) nrx = 0;
            }
            // st_osc / st_osc_half / mixer2 WTnextFrame (:736-740)
            st_last = st_ptr;
            st_ptr += st_step;
            while (((int)st_ptr) >= JD_WTSIZE) st_ptr -= JD_WTSIZE;
            sth_ptr += st_step;
            while (((int)sth_ptr) >= JD_WTSIZE) sth_ptr -= JD_WTSIZE;
            jd_wt_next(m2_ptr, m2_step);
        }
    }
    {
        // the cell being filled goes out entry by entry (once per launch): the next launch reads it back
        const int ph = eb_pos & 7;
        for (int k = 0; k < ph; k++)
        {
            ebe_ring[eb_pos - ph + k] = wc_at(WC_E, k);
            if constexpr (!D8LDS) d8_ring[d8_pos - ph + k] = wc_at(WC_D8, k);
            dly_ring[dly_pos - ph + k] = make_double2(wc_at(WC_DX, k), wc_at(WC_DY, k));
        }
    }
    BLDF(BS_M2_PTR) = m2_ptr; BLDF(BS_M2_STEP) = m2_step; BLDF(BS_M2_FREQ) = m2_freq; BLDF(BS_MC_FREQ) = mc_freq;
    BLDF(BS_ST_PTR) = st_ptr; BLDF(BS_ST_LAST) = st_last; BLDF(BS_STQ_PTR) = sth_ptr; BLDF(BS_VOL_GAIN) = vol_gain;
    BLDF(BS_STR_RE) = str_re; BLDF(BS_STR_IM) = str_im; BLDF(BS_SAV_RE) = sav_re; BLDF(BS_SAV_IM) = sav_im;
    BLDF(BS_ROT_RE) = rot_re; BLDF(BS_ROT_IM) = rot_im; BLDF(BS_ROT_FREQ) = rot_freq;
    BLDF(BS_AGC2_SUM) = agc2_sum; BLDF(BS_EB_ESUM) = eb_esum; BLDF(BS_EB_E2SUM) = eb_e2sum; BLDF(BS_EB_EBNO) = eb_ebno;
    BLDF(BS_RES_X1) = res_x1; BLDF(BS_RES_X2) = res_x2; BLDF(BS_RES_Y1) = res_y1; BLDF(BS_RES_Y2) = res_y2;
    BLDF(BS_MSEMA_SUM) = msema_sum; BLDF(BS_MSE) = mse; BLDF(BS_DIFF_LAST) = diff_last;
    BLDI(BI_STARTSTOP) = startstop; BLDI(BI_CNTR) = cntr; BLDI(BI_MSEMA_POS) = msema_pos; BLDI(BI_NRX) = nrx;
    if constexpr (D8LDS) for (int k = 0; k < g.d8_len; k++) d8_ring[k] = d8l[k * 64];
    BLDI(BI_EB_POS) = eb_pos; BLDI(BI_DLY_POS) = dly_pos; BLDI(BI_D8_POS) = d8_pos; BLDI(BI_A1_POS) = a1_pos;
    {
        // where the reference's delayedsmpl.buffer_ptr stands (k_burst_apply_settings needs it; DelayThing::setLength keeps the contents and
        // restarts the pointer): the gated samples of this launch = how far the window ring's write position moved (a launch is shorter than that ring)
        int adv = eb_pos - eb_pos0;
        if (adv < 0) adv += g.win_ring;
        BLDI(BI_GCNT) = (BLDI(BI_GCNT) + adv) % g.dly_len;
    }
    BLDI(BI_SOFT_CNT) = soft_cnt; BLDI(BI_SYM_CNT) = sym_cnt; BLDI(BI_EV_CNT) = ev_cnt; BLDI(BI_OVERFLOW) = overflow;
}

template <bool CAPSYM, int FIRN, int LDSN>
__global__ __launch_bounds__(128) void k_burst_msk_fb(const BGeom g, const BPtrs p, int n, long long n0, int first_of_write)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lre = lds, *lim = lds + LDSN * 64;
    BmskMail M;
    M.out = lds + 2 * LDSN * 64;
    M.in = M.out + 256;
    M.gate = (int *)(M.in + 384);
    constexpr bool D8LDS = FIRN == 80;
    M.wc = (double *)((char *)M.out + BMSK_FB_MAIL_BYTES);
    M.d8 = (double *)((char *)M.wc + BMSK_FB_WC_BYTES(D8LDS));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, grp = blockIdx.x;
    if (n <= 0) return;
    if (wave == 0) bmsk_front<FIRN, LDSN>(g, p, lre, lim, M, n, n0, grp, lane);
    else bmsk_back<CAPSYM, D8LDS>(g, p, M, n, n0, first_of_write, grp, lane);
}
