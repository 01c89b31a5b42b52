// k_burst_oqpsk_fb.h -- the burst OQPSK tracking chain (BurstOqpskDemodulator::writeDataSlot, JAERO/burstoqpskdemodulator.cpp:488-733) as a front / back
// wavefront pair (round 6).
//
// Why a split pays here although round 3 priced it at 10 %: that estimate took the kernel for a latency-bound recurrence and compiled the matched filter
// out.  The listing says otherwise (profiles/r6_burst_split.md): the single-wavefront kernel k_burst_oqpsk_demod executed ~3 000 instructions per sample
// of which 960 were fp64 arithmetic -- 256 re-materialised constants, 228 moves to and from accumulation registers (it held 256 + 248 registers), 107
// scalar-register spills, 280 exec-mask instructions, 180 instructions of ring addressing for the filter -- and a wavefront that is alone on its SIMD
// issues at most one instruction of any kind per four cycles: 5.8 us per sample.  Two wavefronts share the instruction count, and each keeps its
// working set in 256 registers.
//
//   F ("front"):  val_to_demod from the ring k_burst_front filled, the trident verdict's carrier frequency / phase / volume, mix with mixer2, the
//                 55-tap RRC (history in LDS + registers, the 28 distinct taps as scalar operands, no address arithmetic: k_oqpsk_fb's filter), mixer2's
//                 oscillator.  NOTHING of it depends on the back half (in the burst demodulator the carrier is tracked by a rotator BEHIND the filter,
//                 mixer2 only changes at a trident verdict), so the front half needs no mailbox from the back half at all.
//   B ("back"):   everything behind the filter: burst timing, the preamble's symbol tone, carrier rotator, EbNo meter, AGC, symbol timer, sample
//                 instants, soft bits, emissions.  The chain of k_burst_oqpsk_demod minus the filter.
// One LDS-only barrier per sample (fb_barrier), F one sample ahead, {sre, sim} through a double-buffered mailbox.
// Arithmetic: the front half's filter is the continuous kernels' (one multiplication and one addition per tap, each rounded, as the reference's
// build executes it) -- the front half has the time; the back half keeps the device library's hypot / atan2 (on the reference's
// own off-air recording neither choice moves a soft byte or a soft symbol beyond 1e-9, profiles/r6_burst_recording_ab.json).
#pragma once
#include "k_burst_demod.h"

#define BFB_LDSN 36 // filter history slots in LDS: 36 KiB + 2 KiB of mailbox = 38 912 B per pair, four pairs per CU (155 648 of 163 840 B)
constexpr int bfb_pair_doubles() { return 2 * BFB_LDSN * 64 + 2 * 2 * 64; }

// ------------------------------------------------------------------------------------------------------------------ front half
template <int FIRN, int LDSN>
__device__ __forceinline__ void bfb_front(const BGeom &g, const BPtrs &p, double *lre, double *lim, double *mail, int n, long long n0, int grp, int lane,
                                          const JTaps28 &tp)
{
    constexpr int TAILN = FIRN - LDSN;
    double tre[TAILN], tim[TAILN]; // tre[j] = x_re[n-LDSN-j] once x[n] has been pushed
    const int ch = grp * 64 + lane, nchp = g.nchp;
    const double2 *__restrict__ cis = p.cis;
    const double samplerate = g.Fs;
    double m2_ptr = BLDF(BS_M2_PTR), m2_step = BLDF(BS_M2_STEP), m2_freq = BLDF(BS_M2_FREQ), vol_gain = BLDF(BS_VOL_GAIN);
    const int ev_pos = BLDI(BI_EV_POS);
    const double *__restrict__ cvre = p.cvre + (size_t)grp * g.cv_len * 64 + lane;
    {
        const double *fs = p.firsave + (size_t)grp * 2 * FIRN * 64 + lane;
        for (int k = 0; k < LDSN; k++) { lre[k * 64 + lane] = fs[(size_t)k * 64]; lim[k * 64 + lane] = fs[(size_t)(FIRN + k) * 64]; }
#pragma unroll
        for (int j = 0; j < TAILN; j++) { tre[j] = fs[(size_t)(LDSN + j) * 64]; tim[j] = fs[(size_t)(FIRN + LDSN + j) * 64]; }
    }
    int fir_slot = (int)(n0 % LDSN); // wave-uniform: LDS slot holding the oldest LDS entry, overwritten by the next input
    int s_val = (int)((n0 - g.D1 - g.D2 + 8LL * g.cv_len) % g.cv_len);
    double nx_val = cvre[(size_t)s_val * 64];

    // sample i: trident verdict (:488-515, the carrier oscillator's part), mix + rrc (:517-521), oscillator step (:727); -> mailbox i & 1
    auto produce = [&](int i, bool from_saved_history) __attribute__((always_inline)) {
        const double val = nx_val;
        s_val++; if (s_val >= g.cv_len) s_val = 0;
        if (i + 1 < n) nx_val = cvre[(size_t)s_val * 64];
        if (i == ev_pos)
        {
            const TriResult tr = p.tri[ch];
            if (tr.ok)
            {
                jd_wt_setfreq(m2_freq, m2_step, tr.freq, samplerate);
                bd_set_phase_deg(m2_ptr, tr.phase_deg);
                vol_gain = tr.vol_gain;
            }
        }
        const double2 c2 = cis[jd_cisidx(m2_ptr)];
        const double xin = (vol_gain * val);
        const double cre = c2.x * xin, cim = c2.y * xin;
        double sre, sim;
        // output from x[n-FIRN .. n-1] (FIR::FIRUpdateAndProcess excludes the sample being pushed): taps[i] <-> x[n-FIRN+i]
        if (from_saved_history) jd_fir_eval_sym<FIRN, LDSN, 6>(lre, lim, tp, tre, tim, fir_slot, lane, sre, sim);
        else jd_fir_eval_sym_static<FIRN, LDSN, 6>(lre, lim, tp, tre, tim, fir_slot, lane, sre, sim);
        // push x[n]: the oldest LDS entry moves into the register tail
#pragma unroll
        for (int j = TAILN - 1; j > 0; j--) { tre[j] = tre[j - 1]; tim[j] = tim[j - 1]; }
        tre[0] = lre[fir_slot * 64 + lane]; tim[0] = lim[fir_slot * 64 + lane];
        lre[fir_slot * 64 + lane] = cre; lim[fir_slot * 64 + lane] = cim;
        fir_slot++; if (fir_slot >= LDSN) fir_slot = 0;
        double *d = mail + (i & 1) * 2 * 64 + lane;
        d[0] = sre; d[64] = sim;
        jd_wt_next(m2_ptr, m2_step);
    };
    if (n > 0) produce(0, true);
    fb_barrier();
    for (int i = 0; i < n; i++)
    {
        if (i + 1 < n) produce(i + 1, false);
        fb_barrier();
    }
    BLDF(BS_M2_PTR) = m2_ptr; BLDF(BS_M2_STEP) = m2_step; BLDF(BS_M2_FREQ) = m2_freq; BLDF(BS_VOL_GAIN) = vol_gain;
    {
        double *fs = p.firsave + (size_t)grp * 2 * FIRN * 64 + lane;
        for (int k = 0; k < LDSN; k++) { fs[(size_t)k * 64] = lre[k * 64 + lane]; fs[(size_t)(FIRN + k) * 64] = lim[k * 64 + lane]; }
#pragma unroll
        for (int j = 0; j < TAILN; j++) { fs[(size_t)(LDSN + j) * 64] = tre[j]; fs[(size_t)(FIRN + LDSN + j) * 64] = tim[j]; }
    }
}

// ------------------------------------------------------------------------------------------------------------------- back half
template <bool CAPSYM>
__device__ __forceinline__ void bfb_back(const BGeom &g, const BPtrs &p, const double *mail, int n, long long n0, int first_of_write, int grp, int lane)
{
    const int ch = grp * 64 + lane, nchp = g.nchp;
    const double2 *__restrict__ cis = p.cis;
    const double SPS = g.SPS, samplerate = g.Fs;
#ifdef JD_BURST_EXACT
    const JdAtanLane bd_atl = jd_atan_lane_table(lane);
#endif

    double st_ptr = BLDF(BS_ST_PTR), st_step = BLDF(BS_ST_STEP), st_freq = BLDF(BS_ST_FREQ), st_last = BLDF(BS_ST_LAST);
    double stq_ptr = BLDF(BS_STQ_PTR);
    double str_re = BLDF(BS_STR_RE), str_im = BLDF(BS_STR_IM), sav_re = BLDF(BS_SAV_RE), sav_im = BLDF(BS_SAV_IM);
    double rot_re = BLDF(BS_ROT_RE), rot_im = BLDF(BS_ROT_IM), rot_freq = BLDF(BS_ROT_FREQ);
    // cis(rot_freq), formed where rot_freq changes (verdict, symbol instants) instead of in every sample: the same function of the same
    // argument, off the per-sample chain (the instant block runs in nearly every sample anyway, and there this sincos overlaps the other one)
    double rfs, rfc;
    sincos(rot_freq, &rfs, &rfc);
    double a1_1 = BLDF(BS_A1_1), a1_2 = BLDF(BS_A1_2), a1_3 = BLDF(BS_A1_3), a1_4 = BLDF(BS_A1_4), a1_5 = BLDF(BS_A1_5);
    double agc2_sum = BLDF(BS_AGC2_SUM), eb_esum = BLDF(BS_EB_ESUM), eb_e2sum = BLDF(BS_EB_E2SUM), eb_ebno = BLDF(BS_EB_EBNO);
    double d1 = BLDF(BS_D1), d41_1 = BLDF(BS_D41_1), d41_2 = BLDF(BS_D41_2), d41_3 = BLDF(BS_D41_3);
    double d42_1 = BLDF(BS_D42_1), d42_2 = BLDF(BS_D42_2), d42_3 = BLDF(BS_D42_3), d8_1 = BLDF(BS_D8_1), d8_2 = BLDF(BS_D8_2);
    double res_x1 = BLDF(BS_RES_X1), res_x2 = BLDF(BS_RES_X2), res_y1 = BLDF(BS_RES_Y1), res_y2 = BLDF(BS_RES_Y2);
    double sig2l_re = BLDF(BS_SIG2L_RE), sig2l_im = BLDF(BS_SIG2L_IM), ptd_re = BLDF(BS_PTD_RE), ptd_im = BLDF(BS_PTD_IM);
    double ptd_th = jd_tanh(ptd_re); // kept beside ptd_re inside a launch; formed again here rather than stored
    double msema_sum = BLDF(BS_MSEMA_SUM), mse = BLDF(BS_MSE), lastmse = BLDF(BS_LASTMSE);
    const double thresh = BLDF(BS_THRESH);
    if (first_of_write) lastmse = mse; // double lastmse=mse at the top of writeDataSlot (:318)

    int startstop = BLDI(BI_STARTSTOP), cntr = BLDI(BI_CNTR), yui = BLDI(BI_YUI), insertpre = BLDI(BI_INSERTPRE);
    int msema_pos = BLDI(BI_MSEMA_POS), nrx = BLDI(BI_NRX);
    int soft_cnt = BLDI(BI_SOFT_CNT), sym_cnt = BLDI(BI_SYM_CNT), ev_cnt = BLDI(BI_EV_CNT), overflow = BLDI(BI_OVERFLOW);
    const int flags = BLDI(BI_FLAGS);
    const int ev_pos = BLDI(BI_EV_POS);
    const bool trace = (g.flags & 8u) != 0;

    // ONE ring of |sig2|: ebnomeasure->Update(sig2abs) and agc2->Update(sig2abs) (:570,:576) are fed the same value in the same samples and both
    // start from empty windows at the same moments (constructor, setSettings), so AGC2's moving-average buffer is the newest agc2_len
    // entries of E's, and E2's entries are E's squared (MovingAverage::Update stores fabs(sig), DSP.cpp:408-416: fabs(x)^2 == fabs(x x))
    double *ebe_ring = p.eb_e + (size_t)grp * g.win_ring * 64 + lane;
    double *msema_ring = p.msema + (size_t)ch * g.msema_len;
    int16_t *__restrict__ soft = p.soft + (size_t)ch * g.soft_cap;

    int s_eb = (int)(n0 % g.win_ring);
    int s_agc2 = s_eb - g.agc2_len; if (s_agc2 < 0) s_agc2 += g.win_ring; // the entry written agc2_len samples ago (slot s_eb itself holds the one written win_ring ago)
    int s_e = s_eb - g.eb_len; if (s_e < 0) s_e += g.win_ring;             // the entry written eb_len samples ago
    // wave-uniform constants come from the geometry (scalar registers / literals): formed here they would each hold a vector register pair for the whole launch
    const double w4 = g.w4, w4c = g.w4c, w8 = g.w8, w8c = g.w8c, a1w = g.a1_w, a1wc = g.a1_wc;
    const double agc2_len_d = g.agc2_len_d, eb_len_d = g.eb_len_d;
    // divisions by these constants: reciprocal + two fma corrections (jd_div_const, bit-identical to the quotient)
    const double r_agc2_len = g.r_agc2_len, r_samplerate = g.r_Fs, r_360 = 1.0 / 360.0, wtsize_d = (double)JD_WTSIZE, r_wtsize = 1.0 / wtsize_d;
    const double msema_len_d = g.msema_len_d, r_msema_len = g.r_msema_len;

    // ring entries of sample i+1 are requested at the top of iteration i (all slots are wave-uniform and data independent)
    double nx_agc2 = ebe_ring[(size_t)s_agc2 * 64];
    double nx_e = ebe_ring[(size_t)s_e * 64];
    fb_barrier(); // the front half has formed sample 0
    for (int i = 0; i < n; i++)
    {
        const long long sample = n0 + i;
        const double agc2_old = nx_agc2, e_old = nx_e, e2_old = nx_e * nx_e;
        if (i + 1 < n)
        {
            int sa = s_agc2 + 1; if (sa >= g.win_ring) sa = 0;
            int se = s_e + 1; if (se >= g.win_ring) se = 0;
            nx_agc2 = ebe_ring[(size_t)sa * 64];
            nx_e = ebe_ring[(size_t)se * 64];
        }
        // ---- trident verdict for this sample (:488-515) ----
        if (i == ev_pos)
        {
            const TriResult tr = p.tri[ch];
            if (trace) bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_TRIDENT, tr.ok ? tr.metric : -tr.metric);
            if (tr.ok)
            {
                // (mixer2's frequency / phase and vol_gain: the front half applies them at this sample; the emission carries the frequency as
                // WaveTable::SetFreq leaves it)
                bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_FREQ, tr.freq < 0 ? 0.0 : tr.freq);
                jd_wt_setfreq(st_freq, st_step, g.stref_freq, samplerate);
                bd_set_phase_deg(st_ptr, 0);
                res_x1 = res_x2 = res_y1 = res_y2 = 0;
                startstop = g.startstopstart;
                cntr = 0;
                rot_re = 1; rot_im = 0;
                insertpre = 1;
                rot_freq = 0; rfs = 0.0; rfc = 1.0; // = sincos(0)
                sav_re = 1; sav_im = 0;
                bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_SIGNAL, 1.0);
                mse = 0;
                for (int k = 0; k < g.msema_len; k++) msema_ring[k] = 0;
                msema_pos = 0; msema_sum = 0;
            }
        }
        // ---- mix + rrc (:517-521): the front half's; its output for this sample is in the mailbox ----
        const double *md = mail + (i & 1) * 2 * 64 + lane;
        double sre = md[0], sim = md[64];
        // the symbol oscillator's table entry, needed by the timing loop far below: requested here (valid unless the preamble block moves st_ptr in between)
        const double st_ptr_top = st_ptr;
        const double2 so_pre = cis[jd_cisidx(st_ptr)];
        // ---- sample counting and signal time-out (:523-544) ----
        if (startstop > 0)
        {
            startstop--;
            if (cntr < 1000000) cntr++;
            if (mse < 0.75) startstop = g.startstopstart;
        }
        if (startstop == 0) { startstop--; bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_SIGNAL, 0.0); }
        if ((cntr > ((256 - 10) * SPS)) && insertpre)
        {
            if (soft_cnt < g.soft_cap) { soft[soft_cnt++] = (int16_t)-1; nrx++; } else overflow |= 1;
            insertpre = 0;
        }
        // ---- symbol tone in the preamble (:547-566) ----
        if ((cntr > SPS * (128 + 10)) && (cntr < ((256 - 10) * SPS)))
        {
            const double progress = (((double)cntr) - (SPS * (128 + 10))) / (((256 - 10) * SPS) - (SPS * (128 + 10)));
            double t_re = sre, t_im = sim;
            bd_cmul(t_re, t_im, str_re, str_im);
            bd_cmul(t_re, t_im, 0.0, 1.0);
            const double er = jd_tanh(t_im) * (t_re);
            double sn, cs;
            sincos(er * 0.01, &sn, &cs);
            bd_cmul(str_re, str_im, cs, sn);
            sav_re = sav_re * 0.95 + 0.05 * str_re; sav_im = sav_im * 0.95 + 0.05 * str_im;
            // a1.update(symboltone_pt.real()): Delay<double>(SPS/2): older = x[k-5], newer = x[k-4]
            const double a1out = a1w * a1_4 + a1wc * a1_5;
            a1_5 = a1_4; a1_4 = a1_3; a1_3 = a1_2; a1_2 = a1_1; a1_1 = t_re;
            t_im = a1out;
            const double2 cq = cis[jd_cisidx(stq_ptr)];
            const double e_re = cq.x * t_re - cq.y * (-t_im), e_im = cq.x * (-t_im) + cq.y * t_re;
            double st_err = BD_ATAN2(e_im, e_re);
            st_err *= 1.5 * (1.0 - progress * progress);
            jd_wt_advance_fraction(stq_ptr, -(1.0 / (2.0 * M_PI)) * st_err * 0.1);
            bd_set_phase_deg(st_ptr, jd_div_const(360.0 * stq_ptr, wtsize_d, r_wtsize) * 4.0 + (360.0 * g.ee));
        }
        // ---- carrier phase correction, EbNo, AGC, clip (:570-590) ----
        bd_cmul(sre, sim, sav_re, sav_im);
        bd_cmul(rot_re, rot_im, rfc, rfs);
        bd_cmul(sre, sim, rot_re, rot_im);
        const double sig2abs = BD_HYPOT(sre, sim);
        {
            const double sq = sig2abs * sig2abs;
            double *ep = ebe_ring + (size_t)s_eb * 64;
            eb_e2sum = eb_e2sum - e2_old; eb_e2sum = eb_e2sum + fabs(sq);
            eb_esum = eb_esum - e_old; eb_esum = eb_esum + fabs(sig2abs); *ep = fabs(sig2abs); // the one store: AGC2 below pushes the same value
            s_eb++; if (s_eb >= g.win_ring) s_eb = 0;
            s_e++; if (s_e >= g.win_ring) s_e = 0;
            // The meter's value is observable at the end of a launch (status) and once per burst, at cntr == 384 symbols (:581); its
            // IIR forgets a term after k samples as 0.8^k, so the divide/log10 runs only in the JD_EBNO_TAIL samples before either.
            const double to_emit = ((128.0 + 128.0 + 128.0) * SPS) - (double)cntr;
            if (i >= n - JD_EBNO_TAIL || (to_emit > -1.0 && to_emit < (double)JD_EBNO_TAIL))
            {
                const double e2val = eb_e2sum / eb_len_d, mean = eb_esum / eb_len_d;
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
        if (fabs(cntr - ((128.0 + 128.0 + 128.0) * SPS)) < 0.5) bd_event(g, p, ch, ev_cnt, overflow, sample, BEV_EBNO, eb_ebno);
        {
            agc2_sum = agc2_sum - agc2_old; agc2_sum = agc2_sum + fabs(sig2abs);
            s_agc2++; if (s_agc2 >= g.win_ring) s_agc2 = 0;
            double gain = 1.414213562 / fmax(jd_div_const(agc2_sum, agc2_len_d, r_agc2_len), 0.000001);
            gain = fmax(gain, 0.000001);
            sre *= gain; sim *= gain;
        }
        const double abval = BD_HYPOT(sre, sim);
        if (abval > 2.84) { const double k = (2.84 / abval); sre = k * sre; sim = k * sim; }

        // ---- symbol timer (:592-612) ----
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
            if (cntr > SPS * (128 + 128)) st_eta = y;
        }
        const double d8out = w8 * d8_1 + w8c * d8_2; d8_2 = d8_1; d8_1 = st_eta;
        {
            double2 so = so_pre;
            if (st_ptr != st_ptr_top) so = cis[jd_cisidx(st_ptr)];
            const double m_re = st_eta, m_im = -d8out;
            const double o_re = so.x * m_re - so.y * m_im, o_im = so.x * m_im + so.y * m_re;
            const double st_angle_error = BD_ATAN2(o_im, o_re);
            if (cntr > SPS * (128 + 64))
            {
                fb_wt_setfreq(st_freq, st_step, (-st_angle_error * 0.00000001) + st_freq, samplerate, r_samplerate);
                jd_wt_advance_fraction(st_ptr, jd_div_const(-st_angle_error * 0.01, 360.0, r_360));
            }
            if (st_freq < (g.stref_freq - 0.1)) fb_wt_setfreq(st_freq, st_step, (g.stref_freq - 0.1), samplerate, r_samplerate);
            if (st_freq > (g.stref_freq + 0.1)) fb_wt_setfreq(st_freq, st_step, (g.stref_freq + 0.1), samplerate, r_samplerate);
        }
        // ---- sample times (:615-724) ----
        double frac;
        if (jd_wt_passed(st_last, st_ptr, st_step, g.ee, frac))
        {
            const double pt_last = frac, pt_this = 1.0 - pt_last;
            const double pt_re = pt_this * sre + pt_last * sig2l_re, pt_im = pt_this * sim + pt_last * sig2l_im;
            const double twospeed = -4.0 * (jd_div_const(fb_fmod360(jd_div_const(360.0 * stq_ptr, wtsize_d, r_wtsize) * 2.0 + (360.0 * g.ee * 0.5)), 360.0, r_360) - (0.34046 + 0.4111 * g.ee));
            const bool even = !(twospeed < 0);
            yui++; yui %= 2;
            if (cntr < ((128 + 128) * SPS))
            {
                if ((even && yui == 1) || (!even && yui == 0)) { yui++; yui %= 2; }
            }
            // one tanh per instant for both kinds of instant (k_oqpsk_fb.h, round 4): the earlier instant keeps tanh(ptd_re) beside ptd_re
            const double th = jd_tanh(yui ? pt_im : pt_re);
            if (!yui) { ptd_re = pt_re; ptd_im = pt_im; ptd_th = th; }
            else
            {
                const double q_re = pt_re, q_im = ptd_im;
                const double ct_xt = th * pt_re;
                const double ct_xt_d = ptd_th * ptd_im;
                double ct_ec = ct_xt_d - ct_xt;
                if (ct_ec > M_PI) ct_ec = M_PI;
                if (ct_ec < -M_PI) ct_ec = -M_PI;
                if (ct_ec > M_PI_2) ct_ec = M_PI_2;
                if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
                if (cntr > ((128 + 10) * SPS))
                {
                    double sn, cs;
                    sincos(ct_ec * 0.1, &sn, &cs);
                    bd_cmul(rot_re, rot_im, cs, sn);
                    rot_freq = rot_freq + ct_ec * 0.0001;
                    sincos(rot_freq, &rfs, &rfc);
                    const double tda = (fabs(q_re) - 1.0), tdb = (fabs(q_im) - 1.0);
                    const double e = (tda * tda) + (tdb * tdb);
                    double *mp = msema_ring + msema_pos;
                    msema_sum = msema_sum - *mp; msema_sum = msema_sum + fabs(e); *mp = fabs(e);
                    msema_pos++; if (msema_pos >= g.msema_len) msema_pos = 0;
                    mse = jd_div_const(msema_sum, msema_len_d, r_msema_len);
                }
                if (startstop > 0)
                {
                    if (CAPSYM)
                    {
                        if (sym_cnt < g.sym_cap) { double *sp = p.sym + ((size_t)ch * g.sym_cap + sym_cnt) * 3; sp[0] = q_re; sp[1] = q_im; sp[2] = mse; sym_cnt++; }
                        else overflow |= 2;
                    }
                    const int b0 = jd_softbit(0.75 * q_im * 127.0 + 128.0);
                    const int b1 = jd_softbit(0.75 * q_re * 127.0 + 128.0);
                    if (soft_cnt + 2 <= g.soft_cap) { soft[soft_cnt] = (int16_t)b0; soft[soft_cnt + 1] = (int16_t)b1; soft_cnt += 2; nrx += 2; }
                    else overflow |= 1;
                    if (nrx >= 32)
                    {
                        // emit unless squelched (:708-715); a squelched group is dropped
                        if (!(!(flags & JF_SQL) || mse < thresh || lastmse < thresh)) soft_cnt -= nrx;
                        nrx = 0;
                    }
                }
            }
        }
        sig2l_re = sre; sig2l_im = sim;
        // ---- advance the oscillators (:727-730) ----
        if (st_step < 0) st_step = 0;
        st_last = st_ptr;
        st_ptr += st_step;
        while (((int)st_ptr) >= JD_WTSIZE) st_ptr -= JD_WTSIZE;
        stq_ptr += g.stq_step;
        while (((int)stq_ptr) >= JD_WTSIZE) stq_ptr -= JD_WTSIZE;
        fb_barrier();
    }

    BLDF(BS_ST_PTR) = st_ptr; BLDF(BS_ST_STEP) = st_step; BLDF(BS_ST_FREQ) = st_freq; BLDF(BS_ST_LAST) = st_last;
    BLDF(BS_STQ_PTR) = stq_ptr;
    BLDF(BS_STR_RE) = str_re; BLDF(BS_STR_IM) = str_im; BLDF(BS_SAV_RE) = sav_re; BLDF(BS_SAV_IM) = sav_im;
    BLDF(BS_ROT_RE) = rot_re; BLDF(BS_ROT_IM) = rot_im; BLDF(BS_ROT_FREQ) = rot_freq;
    BLDF(BS_A1_1) = a1_1; BLDF(BS_A1_2) = a1_2; BLDF(BS_A1_3) = a1_3; BLDF(BS_A1_4) = a1_4; BLDF(BS_A1_5) = a1_5;
    BLDF(BS_AGC2_SUM) = agc2_sum; BLDF(BS_EB_ESUM) = eb_esum; BLDF(BS_EB_E2SUM) = eb_e2sum; BLDF(BS_EB_EBNO) = eb_ebno;
    BLDF(BS_D1) = d1; BLDF(BS_D41_1) = d41_1; BLDF(BS_D41_2) = d41_2; BLDF(BS_D41_3) = d41_3;
    BLDF(BS_D42_1) = d42_1; BLDF(BS_D42_2) = d42_2; BLDF(BS_D42_3) = d42_3; BLDF(BS_D8_1) = d8_1; BLDF(BS_D8_2) = d8_2;
    BLDF(BS_RES_X1) = res_x1; BLDF(BS_RES_X2) = res_x2; BLDF(BS_RES_Y1) = res_y1; BLDF(BS_RES_Y2) = res_y2;
    BLDF(BS_SIG2L_RE) = sig2l_re; BLDF(BS_SIG2L_IM) = sig2l_im; BLDF(BS_PTD_RE) = ptd_re; BLDF(BS_PTD_IM) = ptd_im;
    BLDF(BS_MSEMA_SUM) = msema_sum; BLDF(BS_MSE) = mse; BLDF(BS_LASTMSE) = lastmse;
    BLDI(BI_STARTSTOP) = startstop; BLDI(BI_CNTR) = cntr; BLDI(BI_YUI) = yui; BLDI(BI_INSERTPRE) = insertpre;
    BLDI(BI_MSEMA_POS) = msema_pos; BLDI(BI_NRX) = nrx;
    BLDI(BI_SOFT_CNT) = soft_cnt; BLDI(BI_SYM_CNT) = sym_cnt; BLDI(BI_EV_CNT) = ev_cnt; BLDI(BI_OVERFLOW) = overflow;
}

// PAIRS front/back pairs per workgroup: waves 0..PAIRS-1 are the front halves of channel groups blockIdx.x*PAIRS + w, waves PAIRS..2*PAIRS-1 the back
// halves of the same groups (waves w and w + PAIRS of a four-pair workgroup share a SIMD).  A pair whose group lies beyond the bank only keeps the
// barrier count.
template <bool CAPSYM, int PAIRS>
__global__ __launch_bounds__(PAIRS * 128) void k_burst_oqpsk_fb(const BGeom g, const BPtrs p, int n, long long n0, int first_of_write, const JTaps28 tp)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const bool back = wave >= PAIRS;
    const int pair = back ? wave - PAIRS : wave;
    const int grp = blockIdx.x * PAIRS + pair;
    double *base = lds + (size_t)pair * bfb_pair_doubles();
    double *lre = base, *lim = base + BFB_LDSN * 64, *mail = base + 2 * BFB_LDSN * 64;
    if (grp >= g.ngroups)
    {
        for (int i = 0; i <= n; i++) fb_barrier();
        return;
    }
    if (back) bfb_back<CAPSYM>(g, p, mail, n, n0, first_of_write, grp, lane);
    else bfb_front<55, BFB_LDSN>(g, p, lre, lim, mail, n, n0, grp, lane, tp);
}
