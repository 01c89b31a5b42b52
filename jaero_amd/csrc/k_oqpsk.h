// k_oqpsk.h -- sample-loop kernel for the continuous 10.5 kbps OQPSK demodulator.
//
// Re-implements OqpskDemodulator::writeData's per-sample loop (JAERO/oqpskdemodulator.cpp:388-605, fb>8400 branch)
// for 64 channels per wavefront, one channel per lane.  Per-sample stages (SURVEY.md 2.1 numbering):
//   K1 PCM->double, K3 coarse ring fill, K2 NCO mix, K6 RRC matched filter (history in LDS + VGPR tail, taps via scalar loads),
//   K7 EbNo meter (optional), K8 AGC + clip, K9 symbol-timing detector + PLL, K10 sample instant + interpolation,
//   K11 carrier loop, K12 residual rotation, K13 MSE lock detector, K14 soft-bit demap.
// The coarse-frequency estimate (K4/K5) runs in k_coarse.h between two launches of this kernel, at exactly the
// sample where the reference calls it (host splits jaero_write at those samples: `only_a_last`/`skip_a_first`).
#pragma once
#include "jaero_device.h"

#define LDF(f) (p.S[(size_t)(f) * nchp + ch])
#define LDI(f) (p.I[(size_t)(f) * nchp + ch])

// Matched-filter history: the LDSN most recent mixed samples of each arm live in LDS ([slot][lane], wave-uniform slot),
// the FIRN-LDSN oldest ones in a VGPR shift register.  LDSN = 40 makes the ring 40 KiB per wavefront, so four
// wavefronts (one per SIMD) fit a CU's 160 KiB.  The filter output for sample n+1 depends only on inputs up to sample n
// (FIR::FIRUpdateAndProcess excludes the newest sample, DSP.cpp:292-304), so it is evaluated one iteration ahead of the
// serial AGC/timing/carrier chain and overlaps with it.
// PRE8400 (fb == 8400, see k_pre8400.h): the sample is the prefiltered complex value of k_pre8400_fir mixed with
// mixer2 instead of PCM x mixer2 through the 55-tap filter (oqpskdemodulator.cpp:434-448), the carrier loop takes the <= 8400 branch
// (:526-532) and mixer2's frequency is summed over the write for the prefilter's oscillator (:447,607).  Every difference is under
// `if constexpr`, so the other instantiations compile to what they were.
template <int FIRN, int LDSN, bool EBNO, bool CAPSYM, bool PRE8400>
__device__ __forceinline__ void oqpsk_samples_body(const JGeom g, const JPtrs p, const int16_t *__restrict__ pcm, int pcm_stride, int n,
                                                   int skip_a_first, int only_a_last, int fir_slot0, const double2 *__restrict__ prefilt)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lre = lds;              // [LDSN][64]
    double *lim = lds + LDSN * 64;  // [LDSN][64]
    constexpr int TAILN = FIRN - LDSN;
    double tre[TAILN], tim[TAILN];  // tre[j] = x_re[n-LDSN-j] once x[n] has been pushed

    const int lane = threadIdx.x;
    const int grp = blockIdx.x;
    const int ch = grp * 64 + lane;
    const int nchp = g.nchp;
    const bool live = ch < g.nch;
    const double2 *__restrict__ cis = p.cis;
    const double *__restrict__ taps = p.taps2; // this bank's own taps (they depend on fb: alpha 1.0 at 10500, 0.6 at 8400), read once into LDS

    // ---- load state ----
    double m2_ptr = LDF(S_M2_PTR), m2_step = LDF(S_M2_STEP), m2_freq = LDF(S_M2_FREQ);
    double mc_ptr = LDF(S_MC_PTR), mc_step = LDF(S_MC_STEP);
    double st_ptr = LDF(S_ST_PTR), st_step = LDF(S_ST_STEP), st_freq = LDF(S_ST_FREQ), st_last = LDF(S_ST_LAST);
    double agc_sum = LDF(S_AGC_SUM);
    double eb_esum = LDF(S_EB_ESUM), eb_e2sum = LDF(S_EB_E2SUM), eb_ebno = LDF(S_EB_EBNO);
    double d1 = LDF(S_D1);
    double d41_1 = LDF(S_D41_1), d41_2 = LDF(S_D41_2), d41_3 = LDF(S_D41_3);
    double d42_1 = LDF(S_D42_1), d42_2 = LDF(S_D42_2), d42_3 = LDF(S_D42_3);
    double d8_1 = LDF(S_D8_1), d8_2 = LDF(S_D8_2);
    double res_x1 = LDF(S_RES_X1), res_x2 = LDF(S_RES_X2), res_y1 = LDF(S_RES_Y1), res_y2 = LDF(S_RES_Y2);
    double lf_x1 = LDF(S_LF_X1), lf_x2 = LDF(S_LF_X2), lf_y1 = LDF(S_LF_Y1), lf_y2 = LDF(S_LF_Y2);
    double sig2l_re = LDF(S_SIG2L_RE), sig2l_im = LDF(S_SIG2L_IM), ptd_re = LDF(S_PTD_RE), ptd_im = LDF(S_PTD_IM);
    double marg_sum = LDF(S_MARG_SUM), pm_sum = LDF(S_PM_SUM), msema_sum = LDF(S_MSEMA_SUM), mse = LDF(S_MSE);
    const double thresh = LDF(S_THRESH);

    int agc_pos = LDI(I_AGC_POS), eb_pos = LDI(I_EB_POS), bb_ptr = LDI(I_BB_PTR), coarse_cnt = LDI(I_COARSE_CNT);
    int marg_pos = LDI(I_MARG_POS), dt_pos = LDI(I_DT_POS), pm_pos = LDI(I_PM_POS), msema_pos = LDI(I_MSEMA_POS);
    int yui = LDI(I_YUI), sig2l_init = LDI(I_SIG2L_INIT);
    const int flags = LDI(I_FLAGS);
    int soft_cnt = LDI(I_SOFT_CNT), sym_cnt = LDI(I_SYM_CNT), overflow = LDI(I_OVERFLOW);

    const double samplerate = g.Fs; // WaveTable::samplerate after SetFreq(freq,(int)Fs)
    const int nfft_mask = g.nfft - 1;
    double2 *__restrict__ bbring = p.bbring + (size_t)ch * g.nfft;
    double *__restrict__ agc_ring = p.agc_ring + (size_t)grp * g.agc_len * 64 + lane;
    double *__restrict__ ebe_ring = p.eb_e + (size_t)grp * g.ebno_len * 64 + lane;
    double *__restrict__ ebe2_ring = p.eb_e2 + (size_t)grp * g.ebno_len * 64 + lane;
    double *__restrict__ marg_ring = p.marg + (size_t)ch * g.marg_len;
    double2 *__restrict__ dt_ring = p.dt + (size_t)ch * g.dt_len;
    double *__restrict__ pm_ring = p.pm + (size_t)ch * g.pm_len;
    double *__restrict__ msema_ring = p.msema + (size_t)ch * g.msema_len;

    // ---- matched-filter history -> LDS + registers (each lane owns one LDS column: no barrier needed) ----
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

    // filter output for the current sample from the history x[n-FIRN .. n-1]: taps[i] <-> x[n-FIRN+i] (jd_fir_eval)
    double *ltap = lds + 2 * LDSN * 64; // [FIRN] this wavefront's copy of the taps
    if (lane < FIRN) ltap[lane] = taps[lane];
    auto fir_eval = [&](double &ore, double &oim) { jd_fir_eval<FIRN, LDSN, 8>(lre, lim, ltap, tre, tim, fir_slot, lane, ore, oim); };
    double ycur_re = 0, ycur_im = 0;
    if constexpr (!PRE8400) fir_eval(ycur_re, ycur_im);
    double m2fsum = 0; // PRE8400: mixer2_freq_sum of this launch
    double2 nx_pf = make_double2(0.0, 0.0);
    if constexpr (PRE8400) nx_pf = prefilt[ch];

    const double w4 = g.w4, w4c = 1.0 - g.w4, w8 = g.w8, w8c = 1.0 - g.w8;
    const double agc_len_d = (double)g.agc_len, eb_len_d = (double)g.ebno_len;

    // Inputs of sample i+1 (PCM and the ring rows that leave the AGC / EbNo windows) are requested at the top of
    // iteration i and consumed one iteration later, so their HBM latency overlaps a whole sample of arithmetic.
    short nx_pcm = (live && n > 0) ? pcm[ch] : (short)0;
    double nx_agc = agc_ring[(size_t)agc_pos * 64];
    double nx_e = 0, nx_e2 = 0;
    if (EBNO) { nx_e = ebe_ring[(size_t)eb_pos * 64]; nx_e2 = ebe2_ring[(size_t)eb_pos * 64]; }
    // entries leaving the symbol-rate windows at this lane's next full symbol (see the symbol block)
    double sx_marg = marg_ring[marg_pos], sx_pm = pm_ring[pm_pos], sx_ms = msema_ring[msema_pos];
    double2 sx_dt;
    {
        int dn = dt_pos + 1; if (dn >= g.dt_len) dn = 0;
        sx_dt = dt_ring[dn];
    }

    for (int i = 0; i < n; i++)
    {
        const short s = nx_pcm;
        const double dval = ((double)s) / 32768.0;
        const double agc_old = nx_agc, e_old = nx_e, e2_old = nx_e2;
        // requests for this iteration's table look-ups (L2 hits); the next iteration's streams (HBM misses) are requested further
        // down, after c_st has been consumed: vmcnt retires in order, and the wait the compiler puts in front of c_st's use
        // also waited for every younger request, i.e. for those misses, ~450 instructions after they were issued
        const double2 c_m2 = cis[jd_cisidx(m2_ptr)];
        const double2 c_st = cis[jd_cisidx(st_ptr)];

        // ---- K3: coarse-frequency ring fill (oqpskdemodulator.cpp:410-415): decided here, stored further down (before the push)
        // so that cc, like c_m2 and c_st, is loaded and consumed within one iteration: a loaded value carried into the next
        // iteration gets copied between registers at the loop latch, and the s_waitcnt vmcnt(0) in front of that copy drained
        // every request in flight once per sample ----
        const double2 cc = cis[jd_cisidx(mc_ptr)];
        const bool do_fill = !(i == 0 && skip_a_first) && ((coarse_cnt >= g.Fs_int) || !(flags & JF_CPUREDUCE));
        if (i == n - 1 && only_a_last) // the coarse estimate runs now, then the next launch resumes here
        {
            if (do_fill)
            {
                bbring[bb_ptr] = make_double2(cc.x * dval, cc.y * dval);
                bb_ptr = (bb_ptr + 1) & nfft_mask;
            }
            break;
        }
        coarse_cnt++;                          // :431

        // ---- K2 mix + K6 matched filter (:453-456, DSP.cpp:292-304) ----
        // this sample's filter output was evaluated one iteration ago; push x[n] and evaluate the next one now
        double sre = ycur_re, sim = ycur_im;
        if constexpr (PRE8400)
        {
            // sig2 = mixer2.WTCISValue() * cval_prefiltered[i]
            const double2 pf = nx_pf;
            sre = c_m2.x * pf.x - c_m2.y * pf.y;
            sim = c_m2.x * pf.y + c_m2.y * pf.x;
            m2fsum += m2_freq;
            if (i + 1 < n) nx_pf = prefilt[(size_t)(i + 1) * nchp + ch];
        }

        // ---- K7 EbNo (DSP.cpp:729-744) ----
        const double dabval = sqrt(sre * sre + sim * sim);
        if (EBNO)
        {
            const double sq = dabval * dabval;
            double *e2p = ebe2_ring + (size_t)eb_pos * 64;
            double *ep = ebe_ring + (size_t)eb_pos * 64;
            eb_e2sum = eb_e2sum - e2_old; eb_e2sum = eb_e2sum + fabs(sq); *e2p = fabs(sq);
            eb_esum = eb_esum - e_old; eb_esum = eb_esum + fabs(dabval); *ep = fabs(dabval);
            eb_pos++; if (eb_pos >= g.ebno_len) eb_pos = 0;
            if (i >= n - JD_EBNO_TAIL) // wave-uniform; see JD_EBNO_TAIL
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

        // ---- K8 AGC + clip (DSP.cpp:370-379, :466-470) ----
        {
            double *ap = agc_ring + (size_t)agc_pos * 64;
            agc_sum = agc_sum - agc_old;
            agc_sum = agc_sum + fabs(dabval);
            *ap = fabs(dabval);
            agc_pos++; if (agc_pos >= g.agc_len) agc_pos = 0;
        }
        double gain = 1.414213562 / fmax(agc_sum / agc_len_d, 0.000001);
        gain = fmax(gain, 0.000001);
        sre *= gain; sim *= gain;
        const double abval = hypot(sre, sim);
        if (abval > 2.84) { const double k = (2.84 / abval); sre = k * sre; sim = k * sim; }

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
            const double st_angle_error = atan2(o_im, o_re);
            jd_wt_setfreq(st_freq, st_step, (-st_angle_error * 0.00000001) + st_freq, samplerate);
            jd_wt_advance_fraction(st_ptr, -st_angle_error * 0.01 / 360.0);
            if (st_freq < (g.stref_freq - 0.1)) jd_wt_setfreq(st_freq, st_step, (g.stref_freq - 0.1), samplerate);
            if (st_freq > (g.stref_freq + 0.1)) jd_wt_setfreq(st_freq, st_step, (g.stref_freq + 0.1), samplerate);
        }

        // ---- next iteration's streams (see the top of the loop); agc_pos / eb_pos already point at the next sample's rows ----
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < n)
        {
            nx_pcm = live ? pcm[(size_t)(i + 1) * pcm_stride + ch] : (short)0;
            nx_agc = agc_ring[(size_t)agc_pos * 64];
            if (EBNO)
            {
                nx_e = ebe_ring[(size_t)eb_pos * 64];
                nx_e2 = ebe2_ring[(size_t)eb_pos * 64];
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- K10..K14 at symbol instants (:487-595) ----
        if (!sig2l_init) { sig2l_re = sre; sig2l_im = sim; sig2l_init = 1; }
        double frac;
        if (jd_wt_passed(st_last, st_ptr, st_step, g.ee, frac))
        {
            const double pt_last = frac, pt_this = 1.0 - pt_last;
            const double pt_re = pt_this * sre + pt_last * sig2l_re;
            const double pt_im = pt_this * sim + pt_last * sig2l_im;
            yui++; yui %= 2;
            // The four symbol-rate rings are per-channel arrays in HBM (every access a miss).  The entries that leave their windows
            // at this symbol were requested at this lane's previous symbol (sx_*): requested here for this symbol, ~230 instructions
            // did not cover the miss, and with some lane at a symbol instant in nearly
            // every iteration the whole wavefront waited for it once per sample.
            const double marg_old = sx_marg, pm_old = sx_pm, ms_old = sx_ms;
            const double2 dt_old = sx_dt;
            if (yui)
            {
                // next symbol's leaving entries: the slots after the ones this symbol rewrites (all four rings are longer than 2:
                // 800 / 401 / 400 / 400), requested before this block's ~1000 instructions so that they have landed when the
                // loop latch copies them between registers (the s_waitcnt vmcnt(0) in front of that copy is the one wait left)
                int q = marg_pos + 1; if (q >= g.marg_len) q = 0;
                sx_marg = marg_ring[q];
                q = dt_pos + 1; if (q >= g.dt_len) q = 0;
                q = q + 1; if (q >= g.dt_len) q = 0;
                sx_dt = dt_ring[q];
                q = pm_pos + 1; if (q >= g.pm_len) q = 0;
                sx_pm = pm_ring[q];
                q = msema_pos + 1; if (q >= g.msema_len) q = 0;
                sx_ms = msema_ring[q];
            }
            if (!yui) { ptd_re = pt_re; ptd_im = pt_im; }
            else
            {
                double q_re = pt_re, q_im = ptd_im;
                const double ct_xt = tanh(pt_im) * pt_re;
                const double ct_xt_d = tanh(ptd_re) * ptd_im;
                double ct_ec = ct_xt_d - ct_xt;
                if (ct_ec > M_PI) ct_ec = M_PI;
                if (ct_ec < -M_PI) ct_ec = -M_PI;
                if constexpr (!PRE8400)
                {
                    {
                        double y = 0;
                        y += lf_x2 * g.lf_b2; y += lf_x1 * g.lf_b1; y += ct_ec * g.lf_b0;
                        y -= lf_y2 * g.lf_a2; y -= lf_y1 * g.lf_a1;
                        lf_x2 = lf_x1; lf_x1 = ct_ec; lf_y2 = lf_y1; lf_y1 = y;
                        ct_ec = y;
                    }
                    if (ct_ec > M_PI_2) ct_ec = M_PI_2;
                    if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
                    jd_wt_inc_phase_deg(m2_ptr, 1.0 * ct_ec);
                    jd_wt_setfreq(m2_freq, m2_step, (0.01 * ct_ec) + m2_freq, samplerate);
                }
                else
                {
                    // 8400 works better with faster phase agility (:526-532): the raw error moves the phase, the filtered one the frequency
                    double y = 0;
                    y += lf_x2 * g.lf_b2; y += lf_x1 * g.lf_b1; y += ct_ec * g.lf_b0;
                    y -= lf_y2 * g.lf_a2; y -= lf_y1 * g.lf_a1;
                    lf_x2 = lf_x1; lf_x1 = ct_ec; lf_y2 = lf_y1; lf_y1 = y;
                    jd_wt_inc_phase_deg(m2_ptr, 1.0 * ct_ec);
                    jd_wt_setfreq(m2_freq, m2_step, (0.5 * 0.01 * y) + m2_freq, samplerate);
                }

                // marg->UpdateSigned(ct_ec)
                {
                    double *mp = marg_ring + marg_pos;
                    marg_sum = marg_sum - marg_old; marg_sum = marg_sum + ct_ec; *mp = ct_ec;
                    marg_pos++; if (marg_pos >= g.marg_len) marg_pos = 0;
                }
                const double marg_val = marg_sum / ((double)g.marg_len);
                // dt.update(pt_qpsk)
                {
                    dt_ring[dt_pos] = make_double2(q_re, q_im);
                    dt_pos++; if (dt_pos >= g.dt_len) dt_pos = 0;
                    q_re = dt_old.x; q_im = dt_old.y; // = dt_ring[dt_pos]: dt_len > 1, so not the entry just written
                }
                {
                    const double cr = cos(marg_val), sr = sin(marg_val);
                    const double nr = q_re * cr - q_im * sr;
                    const double ni = q_re * sr + q_im * cr;
                    q_re = nr; q_im = ni;
                }
                // MSEcalc::Update (DSP.cpp:451-463)
                {
                    const double av = hypot(q_re, q_im);
                    double *pp = pm_ring + pm_pos;
                    pm_sum = pm_sum - pm_old; pm_sum = pm_sum + fabs(av); *pp = fabs(av);
                    pm_pos++; if (pm_pos >= g.pm_len) pm_pos = 0;
                    double mu = pm_sum / ((double)g.pm_len);
                    if (mu < 0.000001) mu = 0.000001;
                    const double s2 = sqrt(2.0);
                    const double t_re = (s2 * q_re) / mu, t_im = (s2 * q_im) / mu;
                    const double tda = (fabs(t_re) - 1.0), tdb = (fabs(t_im) - 1.0);
                    const double e = (tda * tda) + (tdb * tdb);
                    double *ep = msema_ring + msema_pos;
                    msema_sum = msema_sum - ms_old; msema_sum = msema_sum + fabs(e); *ep = fabs(e);
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
                if (mse < thresh)
                {
                    const int b0 = jd_softbit(0.75 * q_im * 127.0 + 128.0);
                    const int b1 = jd_softbit(0.75 * q_re * 127.0 + 128.0);
                    if (soft_cnt + 2 <= g.soft_cap)
                    {
                        // address rebuilt from an opaque lane index here: kept live across the loop it is spilled, and the
                        // reload sits in this block behind an s_waitcnt vmcnt(0) that drains every prefetch in flight
                        int lx = lane;
                        asm volatile("" : "+v"(lx));
                        int16_t *sp = p.soft + (size_t)(grp * 64 + lx) * g.soft_cap + soft_cnt;
                        sp[0] = (int16_t)b0;
                        sp[1] = (int16_t)b1;
                        soft_cnt += 2;
                    }
                    else overflow |= 1;
                }
            }
        }
        sig2l_re = sre; sig2l_im = sim;

        if (do_fill)
        {
            bbring[bb_ptr] = make_double2(cc.x * dval, cc.y * dval);
            bb_ptr = (bb_ptr + 1) & nfft_mask;
        }

        // ---- push x[n] (mixed with the carrier phase this sample started with) and evaluate the filter for n+1 ----
        if constexpr (!PRE8400)
        {
            const double cre = c_m2.x * dval, cim = c_m2.y * dval;
#pragma unroll
            for (int j = TAILN - 1; j > 0; j--) { tre[j] = tre[j - 1]; tim[j] = tim[j - 1]; }
            tre[0] = lre[fir_slot * 64 + lane];
            tim[0] = lim[fir_slot * 64 + lane];
            lre[fir_slot * 64 + lane] = cre;
            lim[fir_slot * 64 + lane] = cim;
            fir_slot++;
            if (fir_slot >= LDSN) fir_slot = 0;
            fir_eval(ycur_re, ycur_im);
        }

        // ---- advance the NCOs (:600-603) ----
        jd_wt_next(m2_ptr, m2_step);
        jd_wt_next(mc_ptr, mc_step);
        if (st_step < 0) st_step = 0;
        st_last = st_ptr;
        st_ptr += st_step;
        while (((int)st_ptr) >= JD_WTSIZE) st_ptr -= JD_WTSIZE;
    }

    // ---- store state ----
    LDF(S_M2_PTR) = m2_ptr; LDF(S_M2_STEP) = m2_step; LDF(S_M2_FREQ) = m2_freq;
    LDF(S_MC_PTR) = mc_ptr; LDF(S_MC_STEP) = mc_step;
    LDF(S_ST_PTR) = st_ptr; LDF(S_ST_STEP) = st_step; LDF(S_ST_FREQ) = st_freq; LDF(S_ST_LAST) = st_last;
    LDF(S_AGC_SUM) = agc_sum;
    LDF(S_EB_ESUM) = eb_esum; LDF(S_EB_E2SUM) = eb_e2sum; LDF(S_EB_EBNO) = eb_ebno;
    LDF(S_D1) = d1;
    LDF(S_D41_1) = d41_1; LDF(S_D41_2) = d41_2; LDF(S_D41_3) = d41_3;
    LDF(S_D42_1) = d42_1; LDF(S_D42_2) = d42_2; LDF(S_D42_3) = d42_3;
    LDF(S_D8_1) = d8_1; LDF(S_D8_2) = d8_2;
    LDF(S_RES_X1) = res_x1; LDF(S_RES_X2) = res_x2; LDF(S_RES_Y1) = res_y1; LDF(S_RES_Y2) = res_y2;
    LDF(S_LF_X1) = lf_x1; LDF(S_LF_X2) = lf_x2; LDF(S_LF_Y1) = lf_y1; LDF(S_LF_Y2) = lf_y2;
    LDF(S_SIG2L_RE) = sig2l_re; LDF(S_SIG2L_IM) = sig2l_im; LDF(S_PTD_RE) = ptd_re; LDF(S_PTD_IM) = ptd_im;
    LDF(S_MARG_SUM) = marg_sum; LDF(S_PM_SUM) = pm_sum; LDF(S_MSEMA_SUM) = msema_sum; LDF(S_MSE) = mse;
    if constexpr (PRE8400) LDF(S_PRE_FSUM) = LDF(S_PRE_FSUM) + m2fsum;
    LDI(I_AGC_POS) = agc_pos; LDI(I_EB_POS) = eb_pos; LDI(I_BB_PTR) = bb_ptr; LDI(I_COARSE_CNT) = coarse_cnt;
    LDI(I_MARG_POS) = marg_pos; LDI(I_DT_POS) = dt_pos; LDI(I_PM_POS) = pm_pos; LDI(I_MSEMA_POS) = msema_pos;
    LDI(I_YUI) = yui; LDI(I_SIG2L_INIT) = sig2l_init;
    LDI(I_SOFT_CNT) = soft_cnt; LDI(I_SYM_CNT) = sym_cnt; LDI(I_OVERFLOW) = overflow;
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

template <int FIRN, int LDSN, bool EBNO, bool CAPSYM>
__global__ __launch_bounds__(64) void k_oqpsk_samples(const JGeom g, const JPtrs p, const int16_t *__restrict__ pcm,
                                                      int pcm_stride, int n, int skip_a_first, int only_a_last, int fir_slot0)
{
    oqpsk_samples_body<FIRN, LDSN, EBNO, CAPSYM, false>(g, p, pcm, pcm_stride, n, skip_a_first, only_a_last, fir_slot0, nullptr);
}

// fb == 8400: `prefilt` = this segment's rows of JPre::out
template <int FIRN, int LDSN, bool EBNO, bool CAPSYM>
__global__ __launch_bounds__(64) void k_oqpsk_samples_8400(const JGeom g, const JPtrs p, const int16_t *__restrict__ pcm, int pcm_stride, int n,
                                                           int skip_a_first, int only_a_last, int fir_slot0, const double2 *__restrict__ prefilt)
{
    oqpsk_samples_body<FIRN, LDSN, EBNO, CAPSYM, true>(g, p, pcm, pcm_stride, n, skip_a_first, only_a_last, fir_slot0, prefilt);
}

