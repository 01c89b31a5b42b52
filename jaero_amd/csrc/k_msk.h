// k_msk.h -- sample-loop kernel for the continuous 600/1200 bps MSK demodulator.
//
// Re-implements MskDemodulator::writeData's per-sample loop (JAERO/mskdemodulator.cpp:319-485) for 64 channels per
// wavefront, one channel per lane: coarse ring fill, NCO mix, half-sine matched filter (2*SPS taps),
// MSKEbNoMeasure (optional), AGC + clip, SPS-sample delayed arm, |pt_msk| -> resonator -> quadrature delay -> symbol
// PLL weighted by 1-|tanh(err)|, and at symbol instants the decision-directed carrier loop, residual rotation, MSE,
// soft differential decode (DiffDecode::UpdateSoft) and soft-bit demap.
//
// Layout of the matched-filter history (as k_oqpsk_samples): the newest LDSN inputs in an LDS ring, the older
// FIRN - LDSN in registers as a shift register.  1200 bps: 80 = 39 + 41 -> 39.6 KiB of LDS per wavefront with its copy of
// the taps, four wavefronts per CU (one per SIMD); 600 bps: 160 = 78 + 82 -> two per CU.  The filter output of a sample does not contain that
// sample (the reference evaluates, then inserts), so it is evaluated one iteration ahead, and everything a sample reads
// from HBM (PCM, the rows leaving the AGC / EbNo windows, the two delay lines) is requested one iteration ahead too:
// written in reference order, every one of those reads was waited for on the spot, ~8 memory round trips per sample.
#pragma once
#include "jaero_device.h"

template <int FIRN, int LDSN, bool EBNO, bool CAPSYM>
__global__ __launch_bounds__(64) void k_msk_samples(const JGeom g, const JPtrs p, const int16_t *__restrict__ pcm, int pcm_stride,
                                                    int n, int skip_a_first, int only_a_last, int fir_slot0, int dly_slot0,
                                                    int d8_slot0)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lre = lds;             // [LDSN][64]
    double *lim = lds + LDSN * 64; // [LDSN][64]
    constexpr int TAILN = FIRN - LDSN;
    constexpr int TAILA = TAILN > 0 ? TAILN : 1;
    double tre[TAILA], tim[TAILA]; // tre[j] = x_re[n-LDSN-j] once x[n] has been pushed

    const int lane = threadIdx.x;
    const int grp = blockIdx.x;
    const int ch = grp * 64 + lane;
    const int nchp = g.nchp;
    const bool live = ch < g.nch;
    const double2 *__restrict__ cis = p.cis;
    const double *__restrict__ taps = p.taps2; // this bank's own taps, read once into LDS

    double m2_ptr = LDF(S_M2_PTR), m2_step = LDF(S_M2_STEP), m2_freq = LDF(S_M2_FREQ);
    double mc_ptr = LDF(S_MC_PTR), mc_step = LDF(S_MC_STEP);
    double st_ptr = LDF(S_ST_PTR), st_step = LDF(S_ST_STEP), st_last = LDF(S_ST_LAST);
    double agc_sum = LDF(S_AGC_SUM);
    double eb_esum = LDF(S_EB_ESUM), eb_e2sum = LDF(S_EB_E2SUM), eb_ebno = LDF(S_EB_EBNO);
    double res_x1 = LDF(S_RES_X1), res_x2 = LDF(S_RES_X2), res_y1 = LDF(S_RES_Y1), res_y2 = LDF(S_RES_Y2);
    double marg_sum = LDF(S_MARG_SUM), msema_sum = LDF(S_MSEMA_SUM), mse = LDF(S_MSE);
    double diff_last = LDF(S_DIFF_LAST);

    int agc_pos = LDI(I_AGC_POS), bb_ptr = LDI(I_BB_PTR), coarse_cnt = LDI(I_COARSE_CNT);
    int marg_pos = LDI(I_MARG_POS), dt_pos = LDI(I_DT_POS), msema_pos = LDI(I_MSEMA_POS);
    const int flags = LDI(I_FLAGS);
    const bool dcd = flags & JF_DCD;
    int soft_cnt = LDI(I_SOFT_CNT), sym_cnt = LDI(I_SYM_CNT), overflow = LDI(I_OVERFLOW);

    const JdAtanLane atl = jd_atan_lane_table(lane); // jd_atan2's table, one entry per lane (every lane of the wavefront runs the sample loop)
    const double samplerate = g.Fs;
    const int nfft_mask = g.nfft - 1;
    double2 *__restrict__ bbring = p.bbring + (size_t)ch * g.nfft;
    // ONE ring of |sig2| values: the AGC's buffer, the EbNo meter's E buffer and, squared, its E2 buffer (k_msk_fb.h)
    double *__restrict__ win = p.win + (size_t)grp * g.win_len * 64 + lane;
    auto wslot = [&](int pos, int lag) { const int q = pos - lag; return q < 0 ? q + g.win_len : q; };
    double *__restrict__ marg_ring = p.marg + (size_t)ch * g.marg_len;
    double2 *__restrict__ dt_ring = p.dt + (size_t)ch * g.dt_len;
    double *__restrict__ msema_ring = p.msema + (size_t)ch * g.msema_len;
    int16_t *__restrict__ soft = p.soft + (size_t)ch * g.soft_cap;
    const int dly_len = g.sps + 1, d8_len = g.sps2 + 1;
    double2 *__restrict__ dly_ring = p.dly + (size_t)grp * dly_len * 64 + lane;
    double *__restrict__ d8_ring = p.dly8 + (size_t)grp * d8_len * 64 + lane;

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
    int fir_slot = fir_slot0, dly_slot = dly_slot0, d8_slot = d8_slot0; // wave-uniform ring phases
    const double agc_len_d = (double)g.agc_len, eb_len_d = (double)g.ebno_len;

    // filter output for the current sample from the history x[n-FIRN .. n-1]: taps[i] <-> x[n-FIRN+i], oldest first (jd_fir_eval)
    double *ltap = lds + 2 * LDSN * 64; // [FIRN] this wavefront's copy of the taps
    for (int k = lane; k < FIRN; k += 64) ltap[k] = taps[k];
    auto fir_eval = [&](double &ore, double &oim) { jd_fir_eval<FIRN, LDSN, 8>(lre, lim, ltap, tre, tim, fir_slot, lane, ore, oim); };
    double ycur_re, ycur_im;
    fir_eval(ycur_re, ycur_im);

    // inputs of sample i+1 are requested at the top of iteration i (see the file header)
    auto ring_next = [](int pos, int len) { pos++; return pos >= len ? 0 : pos; };
    short nx_pcm = (live && n > 0) ? pcm[ch] : (short)0;
    double nx_agc = win[(size_t)wslot(agc_pos, g.agc_len) * 64];
    double nx_e = 0, nx_e2 = 0;
    if (EBNO) { nx_e = win[(size_t)wslot(agc_pos, g.ebno_len) * 64]; nx_e2 = nx_e * nx_e; }
    double2 nx_cc = cis[jd_cisidx(mc_ptr)];
    double2 nx_ptd = dly_ring[(size_t)ring_next(dly_slot, dly_len) * 64]; // slot read after this sample's write to dly_slot
    double nx_d8 = d8_ring[(size_t)ring_next(d8_slot, d8_len) * 64];

    for (int i = 0; i < n; i++)
    {
        const short s = nx_pcm;
        const double dval = ((double)s) / 32768.0;
        const double agc_old = nx_agc, e_old = nx_e, e2_old = nx_e2, d8out = nx_d8;
        const double2 ptd = nx_ptd;
        // this iteration's table look-ups and next iteration's streams, all independent of the chain below
        const double2 c2 = cis[jd_cisidx(m2_ptr)];
        const double2 c_st = cis[jd_cisidx(st_ptr)];
        if (i + 1 < n)
        {
            nx_pcm = live ? pcm[(size_t)(i + 1) * pcm_stride + ch] : (short)0;
            const int wn = ring_next(agc_pos, g.win_len);
            nx_agc = win[(size_t)wslot(wn, g.agc_len) * 64];
            if (EBNO) { nx_e = win[(size_t)wslot(wn, g.ebno_len) * 64]; nx_e2 = nx_e * nx_e; }
            nx_ptd = dly_ring[(size_t)ring_next(ring_next(dly_slot, dly_len), dly_len) * 64]; // dly_len, d8_len >= 3
            nx_d8 = d8_ring[(size_t)ring_next(ring_next(d8_slot, d8_len), d8_len) * 64];
        }

        // coarse-frequency ring fill (mskdemodulator.cpp:350-355)
        const double2 cc = nx_cc; // table entry of mixer_center for this sample, requested one iteration ago
        {
            double mcn = mc_ptr, mcs = mc_step;
            jd_wt_next(mcn, mcs); // nothing but WTnextFrame moves mixer_center inside a launch
            nx_cc = cis[jd_cisidx(mcn)];
        }
        if (!(i == 0 && skip_a_first))
        {
            const bool fill = (coarse_cnt >= g.Fs_int) || !(flags & JF_CPUREDUCE);
            if (fill)
            {
                bbring[bb_ptr] = make_double2(cc.x * dval, cc.y * dval);
                bb_ptr = (bb_ptr + 1) & nfft_mask;
            }
        }
        if (i == n - 1 && only_a_last) break;
        coarse_cnt++; // :368

        // mix + matched filter (:369-370): this sample's output was evaluated one iteration ago; x[n] is pushed at the end
        double sre = ycur_re, sim = ycur_im;
        const double dabval = sqrt(sre * sre + sim * sim);

        // MSKEbNoMeasure::Update (DSP.cpp:493-505)
        if (EBNO)
        {
            const double sq = dabval * dabval;
            eb_e2sum = eb_e2sum - e2_old; eb_e2sum = eb_e2sum + fabs(sq);
            eb_esum = eb_esum - e_old; eb_esum = eb_esum + fabs(dabval);
            if (i >= n - JD_EBNO_TAIL) // wave-uniform; see JD_EBNO_TAIL
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

        // AGC + clip (:378-382)
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

        double frac;
        if (jd_wt_passed(st_last, st_ptr, st_step, g.ee, frac))
        {
            // entries leaving the three symbol-rate windows (per-channel arrays in HBM), requested together ahead of their use
            const double marg_old = marg_ring[marg_pos];
            int dn = dt_pos + 1; if (dn >= g.dt_len) dn = 0;
            const double2 dt_old = dt_ring[dn]; // dt_len = SPS/2 + 1 > 1
            const double ms_old = msema_ring[msema_pos];
            // carrier tracking (:411-426)
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

            {
                const double v = ct_ec / 2.0;
                double *mp = marg_ring + marg_pos;
                marg_sum = marg_sum - marg_old; marg_sum = marg_sum + v; *mp = v;
                marg_pos++; if (marg_pos >= g.marg_len) marg_pos = 0;
            }
            const double marg_val = marg_sum / ((double)g.marg_len);
            {
                dt_ring[dt_pos] = make_double2(q_re, q_im);
                dt_pos++; if (dt_pos >= g.dt_len) dt_pos = 0;
                q_re = dt_old.x; q_im = dt_old.y;
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
        }

        // push x[n] (mixed with the carrier phase this sample started with) and evaluate the filter for n+1
        {
            const double cre = c2.x * dval, cim = c2.y * dval;
#pragma unroll
            for (int j = TAILN - 1; j > 0; j--) { tre[j] = tre[j - 1]; tim[j] = tim[j - 1]; }
            if (TAILN > 0)
            {
                tre[0] = lre[fir_slot * 64 + lane];
                tim[0] = lim[fir_slot * 64 + lane];
            }
            lre[fir_slot * 64 + lane] = cre;
            lim[fir_slot * 64 + lane] = cim;
            fir_slot++;
            if (fir_slot >= LDSN) fir_slot = 0;
            fir_eval(ycur_re, ycur_im);
        }

        jd_wt_next(m2_ptr, m2_step);
        jd_wt_next(mc_ptr, mc_step);
        if (st_step < 0) st_step = 0;
        st_last = st_ptr;
        st_ptr += st_step;
        while (((int)st_ptr) >= JD_WTSIZE) st_ptr -= JD_WTSIZE;
    }

    LDF(S_M2_PTR) = m2_ptr; LDF(S_M2_STEP) = m2_step; LDF(S_M2_FREQ) = m2_freq;
    LDF(S_MC_PTR) = mc_ptr; LDF(S_MC_STEP) = mc_step;
    LDF(S_ST_PTR) = st_ptr; LDF(S_ST_STEP) = st_step; LDF(S_ST_LAST) = st_last;
    LDF(S_AGC_SUM) = agc_sum;
    LDF(S_EB_ESUM) = eb_esum; LDF(S_EB_E2SUM) = eb_e2sum; LDF(S_EB_EBNO) = eb_ebno;
    LDF(S_RES_X1) = res_x1; LDF(S_RES_X2) = res_x2; LDF(S_RES_Y1) = res_y1; LDF(S_RES_Y2) = res_y2;
    LDF(S_MARG_SUM) = marg_sum; LDF(S_MSEMA_SUM) = msema_sum; LDF(S_MSE) = mse;
    LDF(S_DIFF_LAST) = diff_last;
    LDI(I_AGC_POS) = agc_pos; LDI(I_BB_PTR) = bb_ptr; LDI(I_COARSE_CNT) = coarse_cnt;
    LDI(I_MARG_POS) = marg_pos; LDI(I_DT_POS) = dt_pos; LDI(I_MSEMA_POS) = msema_pos;
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
