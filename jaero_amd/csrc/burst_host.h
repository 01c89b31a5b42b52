// burst_host.h -- host side of the burst banks (included by jaero_hip.hip after jaero_ctx / dalloc / fail are defined).
// Geometry = what BurstOqpskDemodulator::setSettings / BurstMskDemodulator::setSettings compute
// (JAERO/burstoqpskdemodulator.cpp:202-277, JAERO/burstmskdemodulator.cpp:150-325); initial scalar state = their constructors.
#pragma once

static int prof_begin(jaero_ctx *c, int which, hipStream_t st);
static void prof_end(jaero_ctx *c, int idx, hipStream_t st);

static int host_qround(double d) { return d >= 0.0 ? (int)(d + 0.5) : (int)(d - (double)((int)(d - 1)) + 0.5) + (int)(d - 1); }

static void burst_fill_geometry(BGeom &g, const jaero_settings &s, int nch, unsigned flags, int max_write)
{
    memset(&g, 0, sizeof g);
    g.kind = s.kind; g.nch = nch; g.nchp = (nch + 63) / 64 * 64; g.ngroups = g.nchp / 64;
    g.Fs = s.Fs; g.fb = s.fb; g.flags = flags;
    g.hil_ntaps = 2048;
    g.hil_lat = 4 * 2048 - 2048 + 1; // JFastFir: nfft - K + 1 with the inferred default nfft = 4 * 2^ceil(log2 K)
    g.agc_len = (int)round(1 * s.Fs);
    if (s.kind == JAERO_KIND_BURST_OQPSK)
    {
        const double SPS = 2.0 * s.Fs / s.fb;
        g.SPS = SPS;
        g.agc2_len = (int)round((SPS * 64.0 / s.Fs) * s.Fs);
        g.bt_lag = (int)ceil(1.0 * SPS); g.bt_w = delay_weight(1.0 * SPS);
        g.ma1_len = (int)round((double)host_qround(128.0 * SPS));
        g.mav1_len = (int)(SPS * 128);
        g.fa_lag = (int)ceil(SPS * 128); g.fa_w = delay_weight(SPS * 128); g.fa_len = g.fa_lag + 1;
        g.D1 = (int)(SPS * 128.0 * 2.5 - 190);
        g.tri_sz = host_qround((256.0 + 16.0 + 16.0) * SPS);
        g.D2 = g.tri_sz;
        g.PL = (int)(SPS * 128.0 / 2.0); g.pd_thr = 0.2;
        g.nb = host_qround(128.0 * SPS); g.nt = g.nb;
        g.a1_lag = (int)ceil(SPS / 2.0); g.a1_w = delay_weight(SPS / 2.0);
        g.ee = 0.4;
        g.eb_len = (int)(SPS * (256.0));
        g.msema_len = 128;
        g.startstopstart = (int)(SPS * (1050));
        g.w4 = delay_weight(SPS / 4.0); g.w8 = delay_weight(SPS / 8.0);
        g.res_b0 = 0.0048847995518126464; g.res_b1 = 0; g.res_b2 = -0.0048847995518126464;
        g.res_a1 = -0.3882746897971619; g.res_a2 = 0.99023040089637471;
        g.stref_freq = s.fb;
        g.stq_step = (s.fb / 4.0) * ((double)JD_WTSIZE) / ((float)(double)(int)s.Fs);
        g.fir_n = 55;
        g.maxseg = 2048; // <= tri_sz: at most one trident event per channel per segment
    }
    else
    {
        const double SPS = (int)(s.Fs / s.fb);
        g.SPS = SPS;
        g.agc2_len = (int)round((SPS * 128.0 / s.Fs) * s.Fs);
        g.eb_len = (int)(0.15 * s.Fs);
        g.msema_len = 75;
        g.bt_lag = (int)ceil(1.0 * SPS); g.bt_w = delay_weight(1.0 * SPS);
        if (s.fb >= 1200)
        {
            g.ma1_len = (int)round((double)host_qround(126.0 * SPS));
            g.mav1_len = (int)(SPS * 126);
            g.fa_lag = (int)ceil(SPS * 126); g.fa_w = delay_weight(SPS * 126);
            g.PL = (int)(SPS * 126.0 / 2.0); g.pd_thr = 0.1;
            g.tri_sz = host_qround((200.0) * SPS);
            g.D1 = (int)(((int)289 * SPS) + 20);
            g.D2 = (int)(host_qround(72 + 120.0) * SPS);
            g.startstopstart = (int)(SPS * (500));
            g.endRotation = (int)((120 + 37) * SPS);
            g.res_a1 = -1.993312819378528; g.res_a2 = 0.999476538254407;
            g.res_b0 = 2.617308727964618e-04; g.res_b1 = 0; g.res_b2 = -2.617308727964618e-04;
            g.ee = 0.025;
            g.startProcessing = 120;
            g.nb = host_qround(126 * SPS); g.nt = host_qround(74 * SPS);
        }
        else
        {
            g.mav1_len = (int)(SPS * 150);
            g.fa_lag = (int)ceil(SPS * 150); g.fa_w = delay_weight(SPS * 150);
            g.ma1_len = (int)round((double)host_qround(150.0 * SPS));
            g.PL = (int)(SPS * 150.0 / 2.0); g.pd_thr = 0.2;
            g.tri_sz = host_qround((224) * SPS);
            g.D1 = (int)(((int)397 * SPS) + 20);
            g.D2 = host_qround((72 + 150.0) * SPS);
            g.startstopstart = (int)(SPS * (500));
            g.res_a1 = -1.991228154418550; g.res_a2 = 0.997385427096603;
            g.res_b0 = 0.001307286451699; g.res_b1 = 0; g.res_b2 = -0.001307286451699;
            g.ee = 0.015;
            g.startProcessing = 150;
            g.endRotation = (int)((g.startProcessing + 56) * SPS);
            g.nb = host_qround(150 * SPS); g.nt = host_qround(74 * SPS);
        }
        g.fa_len = g.fa_lag + 1;
        g.a1_lag = (int)(SPS / 2); g.a1_w = 0.0;
        g.d8_len = (int)(SPS / 2) + 1; g.dly_len = (int)SPS + 1;
        g.d8_ring = (g.d8_len + 7) / 8 * 8; g.dly_ring = (g.dly_len + 7) / 8 * 8;
        if ((int)SPS == 40) g.d8_ring = g.d8_len; // 1200 bps: delayt8's ring lives in LDS during a launch (k_burst_msk_fb.h), its HBM copy is exactly d8_len entries
        g.stref_freq = s.fb / 2.0;
        g.stq_step = (s.fb / 2.0) * ((double)JD_WTSIZE) / ((float)(double)(int)s.Fs);
        g.fir_n = 2 * (int)SPS;
        g.maxseg = 4096; // <= tri_sz (8000 / 17920)
    }
    if (g.maxseg > max_write) g.maxseg = (max_write + 15) / 16 * 16;
    g.bt_len = 2 * g.PL + 1;
    g.win_ring = g.agc2_len > g.eb_len ? g.agc2_len : g.eb_len;
    g.cv_len = g.D1 + (g.D2 > g.tri_sz ? g.D2 : g.tri_sz) + g.maxseg + 64;
    // whole cells of four samples (k_hilbert); k_hilbert_fft's first block starts up to 2047 samples before the segment and looks
    // hil_lat + 2048 samples further back
    g.hist_len = (g.hil_lat + 2 * g.hil_ntaps + max_write + 64 + 3) & ~3;
}

static int burst_create(jaero_ctx *c, const std::vector<jaero_settings> &sets, const hipDeviceProp_t &prop, int softbit_capacity)
{
    const jaero_settings &s0 = sets[0];
    const int nch = (int)sets.size();
    int rc = 0;
    c->burst = true;
    burst_fill_geometry(c->bg, s0, nch, c->flags, c->max_write);
    BGeom &g = c->bg;
    BPtrs &p = c->bp;
    if (softbit_capacity <= 0) softbit_capacity = (int)ceil(2.0 * c->max_write * g.fb / g.Fs) + 128;
    g.soft_cap = (softbit_capacity + 7) & ~7; // 16-byte aligned rows: the burst-mode Aero-L bank reads them in place eight entries per load (k_aerolb_bits<true>)
    g.sym_cap = (c->flags & JAERO_FLAG_CAPTURE_SYMBOLS) ? g.soft_cap / 2 + 8 : 0;
    g.ev_cap = (c->flags & JAERO_FLAG_TRACE) ? 4096 : 256;
    const int nchp = g.nchp, ng = g.ngroups;
    const bool oq = g.kind == JAERO_KIND_BURST_OQPSK;
#define DA(ptr, count) do { if ((rc = dalloc(c, &(ptr), (size_t)(count)))) return rc; } while (0)
    DA(p.S, (size_t)BS_NFIELDS * nchp);
    DA(p.I, (size_t)BI_NFIELDS * nchp);
    DA(p.pcmhist, (size_t)g.hist_len * nchp);
    DA(p.him, (size_t)ng * g.maxseg * 64);
    DA(p.agc_ring, (size_t)ng * g.agc_len * 64);
    DA(p.cvre, (size_t)ng * g.cv_len * 64); DA(p.cvim, (size_t)ng * g.cv_len * 64);
    DA(p.ma1re, (size_t)ng * g.ma1_len * 64); DA(p.ma1im, (size_t)ng * g.ma1_len * 64);
    DA(p.mav1, (size_t)ng * g.mav1_len * 64);
    DA(p.fa, (size_t)ng * g.fa_len * 64);
    DA(p.bt, (size_t)ng * g.bt_len * 64);
    DA(p.ev_list, nchp); DA(p.ev_count, 4); DA(p.ev_mask, ng);
    DA(p.tri, nchp);
    DA(p.eb_e, (size_t)nchp * g.win_ring);
    DA(p.firsave, (size_t)nchp * 2 * g.fir_n);
    if (!oq) { DA(p.dly, (size_t)nchp * g.dly_ring); DA(p.dly8, (size_t)nchp * g.d8_ring); DA(p.a1, (size_t)nchp * g.d8_len); }
    DA(p.msema, (size_t)nchp * g.msema_len);
    DA(p.soft, (size_t)nchp * g.soft_cap);
    if (g.sym_cap) DA(p.sym, (size_t)nchp * g.sym_cap * 3);
    DA(p.evlog, (size_t)nchp * g.ev_cap * 3);
    DA(c->d_pcm_raw, (size_t)c->max_write * nch);
    DA(c->d_status, nchp);
    c->tri_grid = 2 * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256); // two 256-thread workgroups per CU
    if (c->tri_grid > nchp) c->tri_grid = nchp;
    if (!oq && ((g.agc2_len | g.eb_len) & 7))
        return fail(JAERO_ENOTSUP, "burst MSK at fb %g / Fs %g: the AGC2 / EbNo windows (%d, %d entries) are not whole cells of eight", g.fb, g.Fs, g.agc2_len, g.eb_len);
    if (oq && ((int)floor((0.25 * g.fb) / (g.Fs / (double)TRI_N) + 0.5)) % 4 != 0)
        return fail(JAERO_ENOTSUP, "burst OQPSK at fb %g / Fs %g: k_trident searches one residue class of bins at a time and needs round(fb / 4 / hzperbin) to be a multiple of 4", g.fb, g.Fs);
    c->tri_lds = TRI_XCH * (int)sizeof(double); // wg_fft13_e32's exchange buffer (k_trident; one residue class of trident differences shares it)
    double2 *d_cis = nullptr, *d_tw = nullptr, *d_tw15 = nullptr;
    double *d_taps = nullptr, *d_hil = nullptr;
    DA(d_cis, JD_WTSIZE); DA(d_tw, TRI_H); DA(d_tw15, TRI_H); DA(d_taps, 2 * g.fir_n); DA(d_hil, g.hil_ntaps / 4);
#undef DA
    p.cis = d_cis; p.tw14 = d_tw; p.tw15 = d_tw15; p.taps2 = d_taps; p.hil_taps = d_hil;
    {
        std::vector<double2> cis(JD_WTSIZE);
        for (int i = 0; i < JD_WTSIZE; i++)
        {
            cis[i].y = (sin(2 * M_PI * ((double)i) / JD_WTSIZE));
            cis[i].x = (sin(M_PI_2 + 2 * M_PI * ((double)i) / JD_WTSIZE));
        }
        HIPCHK(hipMemcpy(d_cis, cis.data(), sizeof(double2) * JD_WTSIZE, hipMemcpyHostToDevice));
        std::vector<double2> tw(TRI_H);
        for (int i = 0; i < 8192; i++) { double a = -2.0 * M_PI * ((double)i) / 8192.0; tw[i].x = cos(a); tw[i].y = sin(a); }
        HIPCHK(hipMemcpy(d_tw, tw.data(), sizeof(double2) * 8192, hipMemcpyHostToDevice));
        for (int i = 0; i < TRI_H; i++) { double a = -2.0 * M_PI * ((double)i) / ((double)TRI_N); tw[i].x = cos(a); tw[i].y = sin(a); }
        HIPCHK(hipMemcpy(d_tw15, tw.data(), sizeof(double2) * TRI_H, hipMemcpyHostToDevice));
        std::vector<double> taps;
        if (oq) taps = rrc_design(1.0, 55, g.Fs, g.fb / 2.0);
        else
        {
            taps.resize(g.fir_n);
            for (int i = 0; i < g.fir_n; i++) taps[i] = sin(M_PI * i / (2.0 * g.SPS)) / (2.0 * g.SPS);
        }
        std::vector<double> t2(2 * g.fir_n);
        for (int i = 0; i < 2 * g.fir_n; i++) t2[i] = taps[i % g.fir_n];
        HIPCHK(hipMemcpy(d_taps, t2.data(), sizeof(double) * t2.size(), hipMemcpyHostToDevice));
        // QJHilbertFilter::setSize (JAERO/DSP.cpp:760-787): imaginary part of the odd taps
        const int N = g.hil_ntaps;
        std::vector<double> hil(N / 4);
        for (int j = 0; j < N / 4; j++)
        {
            const int i = 2 * j + 1;
            hil[j] = (2.0 / ((double)N)) / (tan(M_PI * (((double)i) / ((double)N) - 0.5)));
        }
        HIPCHK(hipMemcpy(d_hil, hil.data(), sizeof(double) * hil.size(), hipMemcpyHostToDevice));
        // the same taps at their positions k = 1, 3, ... 2047 for the overlap-save form (k_hilbert_fft)
        std::vector<double> gk(N, 0.0);
        for (int k = 1; k < N; k += 2) gk[k] = (2.0 / ((double)N)) / (tan(M_PI * (((double)k) / ((double)N) - 0.5)));
        double2 *dH = nullptr, *dtw = nullptr;
        if ((rc = fft4096_tables(gk, &dH, &dtw, (const void *)k_hilbert_fft))) return rc;
        c->allocs.push_back(dH); c->allocs.push_back(dtw);
        p.hilH = dH; p.tw12 = dtw;
        if (g.hil_ntaps != 2048) return fail(JAERO_ENOTSUP, "the overlap-save Hilbert kernel is built for QJHilbertFilter's 2048 taps");
    }
    // scalar state
    {
        std::vector<double> S((size_t)BS_NFIELDS * nchp, 0.0);
        std::vector<int> I((size_t)BI_NFIELDS * nchp, 0);
        std::vector<double> ev((size_t)nchp * g.ev_cap * 3, 0.0);
        c->settings.resize(nchp);
        for (int ch = 0; ch < nchp; ch++)
        {
            const jaero_settings &s = sets[ch < nch ? ch : 0];
            c->settings[ch] = s;
            auto SS = [&](int f) -> double & { return S[(size_t)f * nchp + ch]; };
            auto II = [&](int f) -> int & { return I[(size_t)f * nchp + ch]; };
            double fc = s.freq_center;
            if (fc > ((s.Fs / 2.0) - (s.lockingbw / 2.0))) fc = ((s.Fs / 2.0) - (s.lockingbw / 2.0));
            if (fc < 0) fc = 0;
            SS(BS_M2_FREQ) = fc; SS(BS_M2_STEP) = fc * ((double)JD_WTSIZE) / ((float)(double)(int)s.Fs); SS(BS_MC_FREQ) = fc;
            SS(BS_ST_FREQ) = g.stref_freq; SS(BS_ST_STEP) = g.stref_freq * ((double)JD_WTSIZE) / ((float)(double)(int)s.Fs);
            SS(BS_VOL_GAIN) = 1; SS(BS_STR_RE) = 1; SS(BS_SAV_RE) = 1; SS(BS_ROT_RE) = 1;
            SS(BS_MSE) = oq ? 100.0 : 10.0; SS(BS_LASTMSE) = SS(BS_MSE);
            SS(BS_THRESH) = s.signalthreshold; SS(BS_LOCKINGBW) = s.lockingbw; SS(BS_DIFF_LAST) = -1.0;
            II(BI_CNTDOWN) = 2 * g.PL; II(BI_MAXPOSCD) = -1; II(BI_TRI_PTR) = 0; II(BI_EV_POS) = -1;
            II(BI_STARTSTOP) = -1;
            // emit Plottables(...) at the end of setSettings
            ev[((size_t)ch * g.ev_cap) * 3 + 0] = 0; ev[((size_t)ch * g.ev_cap) * 3 + 1] = BEV_FREQ; ev[((size_t)ch * g.ev_cap) * 3 + 2] = fc;
            II(BI_EV_CNT) = 1;
        }
        HIPCHK(hipMemcpy(p.S, S.data(), S.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(p.I, I.data(), I.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(p.evlog, ev.data(), ev.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    c->o_nch = nch; c->o_nchp = nchp; c->o_soft_cap = g.soft_cap; c->o_sym_cap = g.sym_cap;
    c->o_soft = p.soft; c->o_sym = p.sym;
    c->o_soft_cnt = p.I + (size_t)BI_SOFT_CNT * nchp; c->o_sym_cnt = p.I + (size_t)BI_SYM_CNT * nchp;
    c->o_overflow = p.I + (size_t)BI_OVERFLOW * nchp; c->o_flags = p.I + (size_t)BI_FLAGS * nchp;
    c->o_nrx = p.I + (size_t)BI_NRX * nchp;
    c->m.nch = nch; c->m.nchp = nchp;
    c->m.flags.assign(nchp, 0);
    // burst OQPSK: BD_LDSN (36) of its 55 history slots + the taps in LDS; burst MSK: 39 of 80 (1200 bps) or all 160 (600 bps) slots
    const int lds = oq ? (2 * BD_LDSN * 64 + 64) * (int)sizeof(double) : 2 * (g.fir_n == 80 ? BMSK_FB_LDSN_80 : BMSK_FB_LDSN_160) * 64 * (int)sizeof(double) + BMSK_FB_MAIL_BYTES + BMSK_FB_WC_BYTES(g.fir_n == 80) + BMSK_FB_D8_BYTES(g.fir_n == 80, g.d8_len);
    if (oq)
    {
        HIPCHK(hipFuncSetAttribute((const void *)k_burst_oqpsk_demod<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void *)k_burst_oqpsk_demod<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    else
    {
        if (g.fir_n != 80 && g.fir_n != 160) return fail(JAERO_ENOTSUP, "burst MSK matched filter of %d taps has no kernel", g.fir_n);
        // front / back wavefront pairs (k_burst_msk_fb.h): 72 of 80 (two pairs per CU) or 152 of 160 history slots in LDS + 4 KiB of mailboxes
        HIPCHK(hipFuncSetAttribute((const void *)k_burst_msk_fb<false, 80, BMSK_FB_LDSN_80>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void *)k_burst_msk_fb<true, 80, BMSK_FB_LDSN_80>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void *)k_burst_msk_fb<false, 160, BMSK_FB_LDSN_160>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void *)k_burst_msk_fb<true, 160, BMSK_FB_LDSN_160>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    HIPCHK(hipFuncSetAttribute((const void *)k_trident<true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->tri_lds));
    HIPCHK(hipFuncSetAttribute((const void *)k_trident<false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->tri_lds));
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

static int burst_write(jaero_ctx *c, const int16_t *pcm, int nsamples, int layout, int is_device_ptr, hipStream_t st)
{
    const BGeom &g = c->bg;
    const BPtrs &p = c->bp;
    const int nch = g.nch, nchp = g.nchp;
    const int16_t *dsrc = pcm;
    if (!is_device_ptr)
    {
        HIPCHK(hipMemcpyAsync(c->d_pcm_raw, pcm, sizeof(int16_t) * (size_t)nch * nsamples, hipMemcpyHostToDevice, st));
        dsrc = c->d_pcm_raw;
    }
    // new samples -> PCM history ring (the Hilbert FIR only ever reads the ring)
    {
        const int pi = prof_begin(c, 2, st);
        const int slot0 = (int)(c->nsamples_total % g.hist_len);
        if (layout == JAERO_PCM_FRAME_MAJOR)
            hipLaunchKernelGGL(k_hist_push_frames, dim3((nchp + 255) / 256, nsamples), dim3(256), 0, st, dsrc, nch, nch, p.pcmhist, nchp, g.hist_len, slot0, nsamples);
        else
            hipLaunchKernelGGL(k_hist_push_chmajor, dim3(nchp / 64, (nsamples + 63) / 64), dim3(256), 0, st, dsrc, nch, nsamples, p.pcmhist, nchp, g.hist_len, slot0);
        LAUNCHCHK("the burst history push");
        prof_end(c, pi, st);
    }
    const bool cs = (c->flags & JAERO_FLAG_CAPTURE_SYMBOLS) != 0;
    const int lds = g.kind == JAERO_KIND_BURST_OQPSK ? (2 * 39 * 64 + 64) * (int)sizeof(double) : 2 * (g.fir_n == 80 ? BMSK_FB_LDSN_80 : BMSK_FB_LDSN_160) * 64 * (int)sizeof(double) + BMSK_FB_MAIL_BYTES + BMSK_FB_WC_BYTES(g.fir_n == 80) + BMSK_FB_D8_BYTES(g.fir_n == 80, g.d8_len);
    int first = 1;
    c->poisoned = true; // the history push above is idempotent (same slots if the write is repeated); from here on state advances
    for (int pos = 0; pos < nsamples;)
    {
        const int n = (nsamples - pos) < g.maxseg ? (nsamples - pos) : g.maxseg;
        const long long n0 = c->nsamples_total;
        int pi = prof_begin(c, 3, st);
        hipLaunchKernelGGL(k_hilbert_fft, dim3(g.nchp / 8, (int)(((n0 + n - 1) >> 11) - (n0 >> 11) + 1)), dim3(PF_THREADS), 4 * 2 * PRE_L * (int)sizeof(double), st, g, p, n, n0);
        LAUNCHCHK("k_hilbert_fft");
        prof_end(c, pi, st);
        pi = prof_begin(c, 4, st);
        // bt_hold_left is an UPPER BOUND of every lane's BI_BT_HOLD (the per-lane, per-sample counter the kernel obeys): each setSettings sets
        // both to bt_lag at the same moment, the lane's falls by one per sample, this one by n per segment of n samples -- so while any lane
        // still holds, the HOLD instantiation runs (for all lanes: those whose counter is 0 compute what <false> computes).  A write that
        // poisons the bank leaves the counter alone: a poisoned bank accepts nothing but jaero_destroy.
        if (c->bt_hold_left > 0)
        {
            hipLaunchKernelGGL(k_burst_front<true>, dim3(g.ngroups), dim3(64), 0, st, g, p, n, n0);
            c->bt_hold_left -= n;
        }
        else hipLaunchKernelGGL(k_burst_front<false>, dim3(g.ngroups), dim3(64), 0, st, g, p, n, n0);
        LAUNCHCHK("k_burst_front");
        hipLaunchKernelGGL(k_ev_compact, dim3(1), dim3(1024), 0, st, (const unsigned long long *)p.ev_mask, g.ngroups, p.ev_list, p.ev_count);
        LAUNCHCHK("k_ev_compact");
        prof_end(c, pi, st);
        pi = prof_begin(c, 1, st);
        if (g.kind == JAERO_KIND_BURST_OQPSK) hipLaunchKernelGGL(k_trident<true>, dim3(c->tri_grid), dim3(TRI_THREADS), c->tri_lds, st, g, p, n0);
        else hipLaunchKernelGGL(k_trident<false>, dim3(c->tri_grid), dim3(TRI_THREADS), c->tri_lds, st, g, p, n0);
        LAUNCHCHK("k_trident");
        prof_end(c, pi, st);
        pi = prof_begin(c, 0, st);
        if (g.kind == JAERO_KIND_BURST_OQPSK)
        {
            if (cs) hipLaunchKernelGGL((k_burst_oqpsk_demod<true>), dim3(g.ngroups), dim3(64), lds, st, g, p, n, n0, first);
            else hipLaunchKernelGGL((k_burst_oqpsk_demod<false>), dim3(g.ngroups), dim3(64), lds, st, g, p, n, n0, first);
        }
        else
        {
            if (g.fir_n == 80)
            {
                if (cs) hipLaunchKernelGGL((k_burst_msk_fb<true, 80, BMSK_FB_LDSN_80>), dim3(g.ngroups), dim3(128), lds, st, g, p, n, n0, first);
                else hipLaunchKernelGGL((k_burst_msk_fb<false, 80, BMSK_FB_LDSN_80>), dim3(g.ngroups), dim3(128), lds, st, g, p, n, n0, first);
            }
            else
            {
                if (cs) hipLaunchKernelGGL((k_burst_msk_fb<true, 160, BMSK_FB_LDSN_160>), dim3(g.ngroups), dim3(128), lds, st, g, p, n, n0, first);
                else hipLaunchKernelGGL((k_burst_msk_fb<false, 160, BMSK_FB_LDSN_160>), dim3(g.ngroups), dim3(128), lds, st, g, p, n, n0, first);
            }
        }
        LAUNCHCHK("the burst demodulator");
        prof_end(c, pi, st);
        first = 0;
        c->nsamples_total += n;
        pos += n;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// jaero_set_settings with another bit rate on a burst MSK bank (BurstMskDemodulator::setSettings with another fb on the live object,
// burstmskdemodulator.cpp:150-325): a sibling bank for the new rate takes the old one's place behind the handle; the scalar state comes
// across as whole columns, k_burst_carry moves the DelayThings' contents in storage order and applies what setSettings re-creates, msema and
// the outputs not read yet are copied.  Control plane: allocates and synchronises.
static void prof_collect(jaero_ctx *c);
static int burst_rebank(jaero_ctx *c, const jaero_settings *s)
{
    if (c->poisoned) return fail(JAERO_EHIP, "jaero_set_settings: a launch inside an earlier jaero_write failed; this bank's state cannot be carried over");
    const BGeom og = c->bg;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    jaero_ctx *n = nullptr;
    int rc = jaero_create(c->device, og.nch, s, 0, c->flags, c->max_write, c->soft_cap_req, &n);
    if (rc) return rc;
    const BGeom &ng = n->bg;
    const int nchp = og.nchp;
    auto fin = [&](int code) { jaero_destroy(n); return code; };
#define BCP(dst, src, bytes) do { if (hipMemcpy((dst), (src), (bytes), hipMemcpyDeviceToDevice) != hipSuccess) return fin(fail(JAERO_EHIP, "jaero_set_settings: carry-over copy failed")); } while (0)
#define BCP2(dst, dpitch, src, spitch, width, rows) do { if (hipMemcpy2D((dst), (dpitch), (src), (spitch), (width), (rows), hipMemcpyDeviceToDevice) != hipSuccess) return fin(fail(JAERO_EHIP, "jaero_set_settings: carry-over copy failed")); } while (0)
    {
        // outputs not read yet: soft bits (RxDataBits' pending tail included), captured symbols, event rows
        std::vector<int> cnt(2 * (size_t)nchp);
        static_assert(BI_SYM_CNT == BI_SOFT_CNT + 1, "output counters are consecutive columns");
        if (hipMemcpy(cnt.data(), c->bp.I + (size_t)BI_SOFT_CNT * nchp, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost) != hipSuccess) return fin(fail(JAERO_EHIP, "jaero_set_settings: reading the output counters failed"));
        int mx[2] = {0, 0};
        for (int k = 0; k < 2; k++) for (int ch = 0; ch < og.nch; ch++) mx[k] = cnt[(size_t)k * nchp + ch] > mx[k] ? cnt[(size_t)k * nchp + ch] : mx[k];
        if (mx[0] > ng.soft_cap || mx[1] > ng.sym_cap)
            return fin(fail(JAERO_EINVAL, "jaero_set_settings: unread outputs (%d soft bits, %d symbols) exceed the new bank's buffers; read them first", mx[0], mx[1]));
        if (mx[0]) BCP2(n->bp.soft, sizeof(int16_t) * ng.soft_cap, c->bp.soft, sizeof(int16_t) * og.soft_cap, sizeof(int16_t) * mx[0], (size_t)nchp);
        if (mx[1]) BCP2(n->bp.sym, sizeof(double) * 3 * ng.sym_cap, c->bp.sym, sizeof(double) * 3 * og.sym_cap, sizeof(double) * 3 * mx[1], (size_t)nchp);
        BCP(n->bp.evlog, c->bp.evlog, sizeof(double) * (size_t)nchp * og.ev_cap * 3);
    }
    BCP(n->bp.S, c->bp.S, sizeof(double) * (size_t)BS_NFIELDS * nchp);
    BCP(n->bp.I, c->bp.I, sizeof(int) * (size_t)BI_NFIELDS * nchp);
    BCP(n->bp.msema, c->bp.msema, sizeof(double) * (size_t)nchp * og.msema_len); // msema is made once, in the constructor
    BSetVals v;
    v.freq_center = s->freq_center; v.lockingbw = s->lockingbw; v.signalthreshold = s->signalthreshold;
    hipLaunchKernelGGL(k_burst_carry, dim3(nchp / 64), dim3(64), 0, 0, og, c->bp, ng, n->bp, v, (long long)c->nsamples_total);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(0) != hipSuccess) return fin(fail(JAERO_EHIP, "jaero_set_settings: carry-over failed"));
    n->nsamples_total = c->nsamples_total;
    n->m.flags = c->m.flags;
    for (int ch = 0; ch < nchp; ch++) n->m.flags[ch] &= ~JF_DCD;
    n->bt_hold_left = ng.bt_lag;
    if (c->prof) prof_collect(c);
    n->prof = c->prof;
    for (size_t k = 0; k < sizeof(c->slots) / sizeof(c->slots[0]); k++) n->slots[k] = c->slots[k];
#undef BCP
#undef BCP2
    std::swap(*c, *n);
    jaero_destroy(n); // the old bank
    return 0;
}

// jaero_set_settings on live channels [lo, hi) of a burst bank: what BurstOqpskDemodulator::setSettings / BurstMskDemodulator::setSettings do to an
// object that has been running (k_burst_settings.h), on the bank's stream like every other call.  A bank's bit rate and sample rate are fixed
// (ring lengths, kernel instantiations): another fb / Fs is another bank.
static int burst_set_settings(jaero_ctx *c, int channel, const jaero_settings *s)
{
    const BGeom &g = c->bg;
    if (s->kind != g.kind) return fail(JAERO_EINVAL, "jaero_set_settings: the kind of a bank is fixed (another demodulator class in the reference); create a new bank");
    if (s->fb != g.fb || s->Fs != g.Fs)
    {
        if (g.kind != JAERO_KIND_BURST_MSK || s->Fs != g.Fs)
            return fail(JAERO_ENOTSUP, "jaero_set_settings on a burst bank: only burst MSK changes its bit rate (600 / 1200 bps at %g Hz); asked for fb %g / Fs %g", g.Fs, s->fb, s->Fs);
        if (channel >= 0 && g.nch > 1)
            return fail(JAERO_EINVAL, "jaero_set_settings: fb is shared by the channels of a bank; change it for the whole bank (channel = -1)");
        return burst_rebank(c, s);
    }
    HIPCHK(hipSetDevice(c->device));
    const int lo = channel < 0 ? 0 : channel, hi = channel < 0 ? g.nch : channel + 1;
    BSetVals v;
    v.freq_center = s->freq_center; v.lockingbw = s->lockingbw; v.signalthreshold = s->signalthreshold;
    hipLaunchKernelGGL(k_burst_apply_settings, dim3((hi - lo + 63) / 64, 65), dim3(64), 0, c->last_stream, g, c->bp, lo, hi, v, (long long)c->nsamples_total);
    HIPCHK(hipGetLastError());
    for (int ch = lo; ch < hi; ch++)
    {
        c->settings[ch] = *s;
        if (g.kind == JAERO_KIND_BURST_MSK) c->m.flags[ch] &= ~JF_DCD; // dcd = false at the end of BurstMskDemodulator::setSettings
    }
    c->bt_hold_left = g.bt_lag;
    return 0;
}
