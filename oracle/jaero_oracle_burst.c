/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Part of the jaero_oracle.c translation unit (#included at its end so that it
 * shares the static DSP primitives restated there).
 *
 * Plain-C restatement of the reference's BURST demodulators, one function per
 * reference function, citing file:line relative to /root/reference/:
 *   BurstOqpskDemodulator  ctor/setSettings/writeDataSlot   (JAERO/burstoqpskdemodulator.cpp:4-131,202-277,315-737)
 *   BurstMskDemodulator    ctor/setSettings/writeData/CenterFreqChangedSlot
 *                                                             (JAERO/burstmskdemodulator.cpp:9-84,150-325,371-754,327-343)
 *   QJHilbertFilter (JAERO/DSP.cpp:754-794) on top of JFastFir; PeakDetector (JAERO/DSP.h:491-576);
 *   Delay<T> / DelayThing<T> / TMovingAverage<T> (JAERO/DSP.h:341-379,439-486,145-199); FFTrWrapper (JAERO/fftrwrapper.cpp:19-27).
 * JFastFir belongs to jontio/JFFT, which the reference links but does not vendor (unpinned HEAD clone,
 * ci-linux-build.sh:152-162).  Its observable behaviour is pinned by the reference's own golden vectors
 * (JAERO/tests/jfastfir_tests.cpp: causal convolution delayed by nfft-K+1 samples); the block overlap-add below is the
 * same one oracle/ref/shim/jfft.h gives the unmodified reference build, so that this file is BIT-IDENTICAL to
 * oracle/_ref/jaero_ref (tests/test_oracle_vs_ref.py).  The default nfft of the 1-argument SetKernel (4 * 2^ceil(log2 K))
 * is an inference: it only shifts burst outputs in time.
 *
 * Members the reference leaves uninitialised (BurstOqpskDemodulator::rotator_freq, burstoqpskdemodulator.h:186) are 0
 * here; the _ref driver constructs the object in zeroed storage for the same reason.
 * GUI-only work (spectrum ring, scatter points, PeakVolume, QElapsedTimer) is omitted.
 */

enum { JO_EV_SIGNAL = 0, JO_EV_EBNO = 1, JO_EV_FREQ = 2, JO_EV_PEAK = 3, JO_EV_TRIDENT = 4 };

/* ------------------------------------------------------------------ dynamic-size Delay<T> (DSP.h:341-379) */
typedef struct { double *buff; int size, buffptr; double fractdelay; } ddelay_t;
static void ddelay_set(ddelay_t *d, double fractdelay)
{
    d->fractdelay = fractdelay;
    d->size = (int)ceil(fractdelay) + 1;
    free(d->buff);
    d->buff = (double *)calloc((size_t)d->size, sizeof(double));
    d->buffptr = 0;
}
static double ddelay_update(ddelay_t *d, double sig)
{
    d->buff[d->buffptr] = sig;
    double dptr = ((double)d->buffptr) - d->fractdelay;
    d->buffptr++; d->buffptr %= d->size;
    while (floor(dptr) < 0) dptr += ((double)d->size);
    int iptr = (int)floor(dptr);
    double weighting = dptr - ((double)iptr);
    double older = d->buff[iptr];
    iptr++; iptr %= d->size;
    double newer = d->buff[iptr];
    return (weighting * newer + (1.0 - weighting) * older);
}
typedef struct { cpx *buff; int size, buffptr; double fractdelay; } cdelay_t;
static void cdelay_set(cdelay_t *d, double fractdelay)
{
    d->fractdelay = fractdelay;
    d->size = (int)ceil(fractdelay) + 1;
    free(d->buff);
    d->buff = (cpx *)calloc((size_t)d->size, sizeof(cpx));
    d->buffptr = 0;
}
static cpx cdelay_update(cdelay_t *d, cpx sig)
{
    d->buff[d->buffptr] = sig;
    double dptr = ((double)d->buffptr) - d->fractdelay;
    d->buffptr++; d->buffptr %= d->size;
    while (floor(dptr) < 0) dptr += ((double)d->size);
    int iptr = (int)floor(dptr);
    double weighting = dptr - ((double)iptr);
    cpx older = d->buff[iptr];
    iptr++; iptr %= d->size;
    cpx newer = d->buff[iptr];
    cpx r; /* weighting*newer+(1.0-weighting)*older : double*complex scales both parts */
    r.re = weighting * newer.re + (1.0 - weighting) * older.re;
    r.im = weighting * newer.im + (1.0 - weighting) * older.im;
    return r;
}

/* ------------------------------------------------------------------ DelayThing<double> (DSP.h:439-486) */
typedef struct { double *buffer; int ptr, sz; } dthing_t;
static void dthing_set_length(dthing_t *d, int length)
{
    length++;
    double *nb = (double *)calloc((size_t)length, sizeof(double));
    if (d->buffer) { int keep = d->sz < length ? d->sz : length; memcpy(nb, d->buffer, sizeof(double) * (size_t)keep); free(d->buffer); }
    d->buffer = nb; d->ptr = 0; d->sz = length;
}
static void dthing_update(dthing_t *d, double *data)
{
    d->buffer[d->ptr] = *data;
    d->ptr++; d->ptr %= d->sz;
    *data = d->buffer[d->ptr];
}
static double dthing_update_dont_touch(dthing_t *d, double data)
{
    d->buffer[d->ptr] = data;
    d->ptr++; d->ptr %= d->sz;
    return d->buffer[d->ptr];
}
static int dthing_findmaxpos(dthing_t *d, double *maxval) /* :465-479 */
{
    int maxpos = 0;
    *maxval = d->buffer[d->ptr];
    for (int i = 0; i < d->sz; i++)
    {
        if (d->buffer[d->ptr] > *maxval) { *maxval = d->buffer[d->ptr]; maxpos = i; }
        d->ptr++; d->ptr %= d->sz;
    }
    return maxpos;
}

/* ------------------------------------------------------------------ TMovingAverage<complex> (DSP.h:145-199) */
typedef struct { int sz, ptr; cpx sum, val; cpx *buf; } cma_t;
static void cma_set_length(cma_t *m, int number)
{
    free(m->buf);
    m->sz = (int)round(number);
    m->buf = (cpx *)calloc((size_t)m->sz, sizeof(cpx));
    m->sum.re = m->sum.im = 0; m->val.re = m->val.im = 0; m->ptr = 0;
}
static cpx cma_update_signed(cma_t *m, cpx sig)
{
    m->sum.re = m->sum.re - m->buf[m->ptr].re; m->sum.im = m->sum.im - m->buf[m->ptr].im;
    m->sum.re = m->sum.re + sig.re; m->sum.im = m->sum.im + sig.im;
    m->buf[m->ptr] = sig;
    m->ptr++; m->ptr %= m->sz;
    m->val.re = m->sum.re / ((double)m->sz); m->val.im = m->sum.im / ((double)m->sz);
    return m->val;
}
static void ma_zero(ma_t *m) /* MovingAverage::Zero DSP.cpp:400-406 */
{
    for (int i = 0; i < m->sz; i++) m->buf[i] = 0;
    m->ptr = 0; m->val = 0; m->sum = 0;
}

/* ------------------------------------------------------------------ PeakDetector (DSP.h:491-576) */
typedef struct
{
    dthing_t d1, d2, d3;
    double lastdy, threshold, maxval;
    int cntdown, maxcntdown, maxpos, maxposcntdown;
} peakdet_t;
static void peakdet_set(peakdet_t *p, int length, double threshold) /* setSettings(int,double) :503-514 */
{
    dthing_set_length(&p->d1, length * 2);
    dthing_set_length(&p->d2, length);
    p->lastdy = 0;
    p->maxcntdown = 2 * length;
    p->cntdown = p->maxcntdown;
    p->threshold = threshold;
    p->maxposcntdown = -1;
    dthing_set_length(&p->d3, 2 * length);
}
static int peakdet_update(peakdet_t *p, double *val) /* :528-560 */
{
    double val2 = dthing_update_dont_touch(&p->d3, *val);
    double dy = *val - dthing_update_dont_touch(&p->d1, *val);
    double tmp = *val; dthing_update(&p->d2, &tmp); /* d2.update(val) overwrites val with the delayed sample ... */
    *val = tmp;                                      /* ... which is what the comparison below then sees (DSP.h:533-534) */
    if ((!p->cntdown) && (*val > p->threshold) && ((p->lastdy >= 0 && dy < 0)))
    {
        p->cntdown = p->maxcntdown;
        p->maxval = 0;
        p->maxpos = dthing_findmaxpos(&p->d3, &p->maxval);
        p->maxposcntdown = p->maxpos;
    }
    if (p->cntdown > 0) p->cntdown--;
    p->lastdy = dy;
    *val = val2;
    if (!p->maxposcntdown) { p->maxposcntdown--; return 1; }
    if (p->maxposcntdown > 0) p->maxposcntdown--;
    return 0;
}

/* ------------------------------------------------------------------ QJHilbertFilter on JFastFir (fastfir_t: jaero_oracle.c) */
/* QJHilbertFilter::setSize (DSP.cpp:760-787) + JFastFir::SetKernel(kernel) default nfft */
static void hilbert_set_size(fastfir_t *f, int N)
{
    N = (int)pow(2.0, (ceil(log2(N))));
    cpx *kernel = (cpx *)calloc((size_t)N, sizeof(cpx));
    for (int i = 0; i < N; i++)
    {
        if (i == N / 2) { kernel[i].re = -1; kernel[i].im = 0; continue; }
        if ((i % 2) == 0) { kernel[i].re = 0; kernel[i].im = 0; continue; }
        kernel[i].re = 0;
        kernel[i].im = (2.0 / ((double)N)) / (tan(M_PI * (((double)i) / ((double)N) - 0.5)));
    }
    int p = 1;
    while (p < N) p <<= 1;
    fastfir_set_kernel(f, kernel, N, 4 * p);
    free(kernel);
}
/* the Hilbert kernel itself, for the tests that pin the HIP FIR against it */
int jo_hilbert_kernel(int N, double *re_im)
{
    N = (int)pow(2.0, (ceil(log2(N))));
    for (int i = 0; i < N; i++)
    {
        double re = 0, im = 0;
        if (i == N / 2) re = -1;
        else if ((i % 2) != 0) im = (2.0 / ((double)N)) / (tan(M_PI * (((double)i) / ((double)N) - 0.5)));
        re_im[2 * i] = re; re_im[2 * i + 1] = im;
    }
    return N;
}

/* ------------------------------------------------------------------ the burst demodulator object */
struct jo_burst
{
    int kind;
    int afc, sql, cpuReduce, dcd;
    double Fs, freq_center, lockingbw, fb, signalthreshold, SamplesPerSymbol;
    long nsamples_total; /* samples consumed so far (event time stamps) */
    /* front end */
    fastfir_t hfir; cpx *hfirbuff; long hfircap;
    agc_t *agc, *agc2;
    delaything_t d1; dthing_t d2;
    cdelay_t bt_d1; ddelay_t bt_ma_diff;
    cma_t bt_ma1; ma_t *mav1;
    peakdet_t pdet;
    double *tridentbuffer; int tridentbuffer_ptr, tridentbuffer_sz;
    fftplan *fftr; int N; cpx *out_base, *out_top; double *out_abs_diff, *in;
    /* demod */
    wavetable mixer2, mixer_center, st_osc, st_osc_ref, st_osc_quarter, st_osc_half;
    fir_t *fir_re, *fir_im;
    delay_t delays, delayt41, delayt42, delayt8, a1;
    iir_t st_iir_resonator, ct_iir_loopfilter;
    double ee; cpx symboltone_averotator, symboltone_rotator, rotator; double carrier_rotation_est, rotator_freq;
    ebno_t ebno;
    double mse; ma_t *msema, *pointmean;
    delaything_t delayedsmpl;
    double diff_lastsoftstate;
    int startstopstart, startstop, cntr, yui, insertpreamble, pointbuff_ptr;
    int startProcessing, endRotation;
    cpx pt_d, sig2_last; double vol_gain;
    /* RxDataBits */
    short rx[96]; int nrx;
    gbuf soft, events, symbols;
    int capture_symbols, trace;
};
typedef struct jo_burst jo_burst;

static void burst_event(jo_burst *d, long sample, int kind, double value)
{
    double row[3] = {(double)sample, (double)kind, value};
    gpush(&d->events, row, sizeof(row));
}
static void burst_emit_soft(jo_burst *d) { gpush(&d->soft, d->rx, sizeof(short) * (size_t)d->nrx); d->nrx = 0; }

/* FFTrWrapper<double>::transform(real in, complex out) (fftrwrapper.cpp:19-27): full complex FFT of the real input;
 * bins above N/2 are zeroed (never read by the callers) */
static void fftr_transform(jo_burst *d, const double *in, cpx *out)
{
    for (int i = 0; i < d->N; i++) { out[i].re = in[i]; out[i].im = 0.0; }
    fft_run(d->fftr, out, 0);
    for (int i = d->N / 2 + 1; i < d->N; i++) { out[i].re = 0; out[i].im = 0; }
}

/* ---------------- burst OQPSK ---------------- */
static void boqpsk_set_settings(jo_burst *d, const jo_settings *s) /* burstoqpskdemodulator.cpp:202-277 */
{
    d->Fs = s->Fs; d->lockingbw = s->lockingbw; d->fb = s->fb;
    d->freq_center = s->freq_center;
    if (d->freq_center > ((d->Fs / 2.0) - (d->lockingbw / 2.0))) d->freq_center = ((d->Fs / 2.0) - (d->lockingbw / 2.0));
    d->signalthreshold = s->signalthreshold;
    d->SamplesPerSymbol = 2.0 * d->Fs / d->fb;
    const double SPS = d->SamplesPerSymbol;
    wt_setfreq_sr(&d->mixer2, d->freq_center, (int)d->Fs);
    agc_free(d->agc); d->agc = agc_new(1, d->Fs);
    agc_free(d->agc2); d->agc2 = agc_new(SPS * 64.0 / d->Fs, d->Fs);
    hilbert_set_size(&d->hfir, 2048);
    d->pointbuff_ptr = 0;
    cdelay_set(&d->bt_d1, 1.0 * SPS);
    cma_set_length(&d->bt_ma1, qRound_(128.0 * SPS));
    ma_free(d->mav1); d->mav1 = ma_new((int)(SPS * 128));
    ddelay_set(&d->bt_ma_diff, SPS * 128);
    delaything_set_length(&d->d1, (int)(SPS * 128.0 * 2.5 - 190));
    if (d->fftr) fft_free(d->fftr);
    d->N = 4096 * 4 * 2;
    d->fftr = fft_plan(d->N);
    d->tridentbuffer_sz = qRound_((256.0 + 16.0 + 16.0) * SPS);
    d->tridentbuffer = (double *)realloc(d->tridentbuffer, sizeof(double) * (size_t)d->tridentbuffer_sz);
    d->tridentbuffer_ptr = 0; /* QVector::resize keeps old contents; they are overwritten before they are read */
    dthing_set_length(&d->d2, d->tridentbuffer_sz);
    d->in = (double *)realloc(d->in, sizeof(double) * (size_t)d->N);
    d->out_base = (cpx *)realloc(d->out_base, sizeof(cpx) * (size_t)d->N);
    d->out_top = (cpx *)realloc(d->out_top, sizeof(cpx) * (size_t)d->N);
    d->out_abs_diff = (double *)realloc(d->out_abs_diff, sizeof(double) * (size_t)(d->N / 2));
    peakdet_set(&d->pdet, (int)(SPS * 128.0 / 2.0), 0.2);
    delay_set(&d->a1, SPS / 2.0);
    d->ee = 0.4;
    d->symboltone_averotator.re = 1; d->symboltone_averotator.im = 0;
    d->carrier_rotation_est = 0;
    ma_free(d->ebno.E); ma_free(d->ebno.E2);
    d->ebno.E = ma_new((int)(SPS * (256.0))); d->ebno.E2 = ma_new((int)(SPS * (256.0))); d->ebno.Fs = d->Fs; d->ebno.fb = d->fb;
    d->ebno.EbNo = 0; /* uninitialised in the reference (DSP.cpp:715-721) */
    d->rotator.re = 1; d->rotator.im = 0;
    d->startstopstart = (int)(SPS * (1050));
    d->insertpreamble = 0;
    burst_event(d, d->nsamples_total, JO_EV_FREQ, d->mixer2.freq);
}
static void boqpsk_ctor(jo_burst *d) /* burstoqpskdemodulator.cpp:4-131 */
{
    d->mse = 100;
    d->insertpreamble = 0;
    d->Fs = 48000; d->fb = 10500;
    d->SamplesPerSymbol = 2.0 * d->Fs / d->fb;
    wt_init(&d->mixer2); wt_init(&d->st_osc); wt_init(&d->st_osc_ref); wt_init(&d->st_osc_quarter);
    double pts[64];
    int np = jo_rrc_design(1, 55, d->Fs, d->fb / 2.0, pts);
    d->fir_re = fir_new(np); d->fir_im = fir_new(np);
    for (int i = 0; i < np; i++) { d->fir_re->points[i] = pts[i]; d->fir_im->points[i] = pts[i]; }
    const double SPS = d->SamplesPerSymbol;
    delay_set(&d->delays, 1); delay_set(&d->delayt41, SPS / 4.0); delay_set(&d->delayt42, SPS / 4.0); delay_set(&d->delayt8, SPS / 8.0);
    /* 75 Hz resonator :67-74 */
    set_resonator(&d->st_iir_resonator, 0.0048847995518126464, 0, -0.0048847995518126464, 1, -0.3882746897971619, 0.99023040089637471);
    iir_init(&d->st_iir_resonator);
    wt_setfreq_sr(&d->st_osc, d->fb, (int)d->Fs);
    wt_setfreq_sr(&d->st_osc_ref, d->fb, (int)d->Fs);
    wt_setfreq_sr(&d->st_osc_quarter, d->fb / 4.0, (int)d->Fs);
    set_resonator(&d->ct_iir_loopfilter, 0.0010275610653672064, 0.0020551221307344128, 0.0010275610653672064, 1, -1.9207386815577139, 0.92509247310306331);
    iir_init(&d->ct_iir_loopfilter);
    d->msema = ma_new(128);
    d->pt_d.re = d->pt_d.im = 0; d->yui = 0; d->sig2_last.re = d->sig2_last.im = 0;
    d->symboltone_rotator.re = 1; d->symboltone_rotator.im = 0;
    d->startstop = -1; d->vol_gain = 1; d->cntr = 0;
    d->rotator_freq = 0; /* uninitialised in the reference */
}

static void boqpsk_trident_check(jo_burst *d, long sample) /* burstoqpskdemodulator.cpp:412-515 */
{
    const double SPS = d->SamplesPerSymbol;
    const int nb = qRound_(128.0 * SPS);
    const int N = d->N;
    for (int k = 0; k < N; k++) d->in[k] = (k < nb) ? d->tridentbuffer[k] : 0;
    fftr_transform(d, d->in, d->out_base);
    for (int k = 0; k < N; k++) d->in[k] = (k < nb && nb + k < d->tridentbuffer_sz) ? d->tridentbuffer[nb + k] : 0;
    fftr_transform(d, d->in, d->out_top);
    for (int i = 0; i < N / 2; i++) d->out_abs_diff[i] = (hypot(d->out_top[i].re, d->out_top[i].im) - hypot(d->out_base[i].re, d->out_base[i].im));
    double hzperbin = d->Fs / ((double)N);
    double binpeakspacing = (0.25 * d->fb) / hzperbin;
    int b = qRound_(binpeakspacing);
    int firstbin = b;
    int lstbin = N / 2 - b;
    double maxval = d->out_abs_diff[firstbin - b] + d->out_abs_diff[firstbin + b] - d->out_abs_diff[firstbin];
    double maxvalbin = firstbin;
    for (int i = firstbin; i < lstbin; i++)
    {
        double testval = d->out_abs_diff[i - b] + d->out_abs_diff[i + b] - d->out_abs_diff[i];
        if (testval > maxval) { maxval = testval; maxvalbin = i; }
    }
    /* minval2 search (:447-456) has no effect on the outcome */
    double minval = hypot(d->out_base[0].re, d->out_base[0].im);
    double minvalbin = 0;
    for (int i = 0; i < N / 2; i++)
    {
        double a = hypot(d->out_base[i].re, d->out_base[i].im);
        if (a > minval) { minval = a; minvalbin = i; }
    }
    int ok = (maxval > 500.0) && (fabs((((double)(maxvalbin - minvalbin))) * hzperbin) < 20.0);
    if (d->trace) burst_event(d, sample, JO_EV_TRIDENT, ok ? maxval : -maxval);
    if (ok)
    {
        double carrierphase = atan2(d->out_base[(int)minvalbin].im, d->out_base[(int)minvalbin].re) - (M_PI / 4.0);
        wt_setfreq(&d->mixer2, hzperbin * minvalbin);
        wt_set_phase_deg(&d->mixer2, (180.0 / M_PI) * carrierphase);
        burst_event(d, sample, JO_EV_FREQ, d->mixer2.freq);
        d->vol_gain = 1.4142 * 500.0 / minval;
        d->pointbuff_ptr = -128 - 128;
        wt_setfreq(&d->st_osc, d->st_osc_ref.freq);
        wt_set_phase_deg(&d->st_osc, 0);
        wt_set_phase_deg(&d->st_osc_ref, 0);
        iir_init(&d->st_iir_resonator);
        iir_init(&d->ct_iir_loopfilter);
        d->startstop = d->startstopstart;
        d->cntr = 0;
        d->rotator.re = 1; d->rotator.im = 0;
        d->insertpreamble = 1;
        d->rotator_freq = 0;
        d->symboltone_averotator.re = 1; d->symboltone_averotator.im = 0;
        d->carrier_rotation_est = 0;
        burst_event(d, sample, JO_EV_SIGNAL, 1.0);
        d->mse = 0;
        ma_zero(d->msema);
    }
}

static cpx cexp_i(double x) /* std::exp(imag*x): real part of the argument is +-0 -> exp()=1 */
{
    cpx r; r.re = cos(x); r.im = sin(x); return r;
}

static void burst_front_end(jo_burst *d, cpx *cvalp, cpx *cval_d_out, double *val_to_demod_out, int *fire_out)
{
    /* burstoqpskdemodulator.cpp:372-396 == burstmskdemodulator.cpp:413-435 */
    cpx cval = *cvalp;
    agc_update(d->agc, hypot(cval.re, cval.im));
    cval = cscale(cval, d->agc->val);
    cpx cval_d = delaything_update_dont_touch(&d->d1, cval);
    double val_to_demod = dthing_update_dont_touch(&d->d2, cval_d.re);
    cpx dl = cdelay_update(&d->bt_d1, cval);
    cpx cj; cj.re = dl.re; cj.im = -dl.im;
    cpx m = cma_update_signed(&d->bt_ma1, cmul(cval, cj));
    double fastarm = hypot(m.re, m.im);
    fastarm = ma_update_signed(d->mav1, fastarm);
    fastarm -= ddelay_update(&d->bt_ma_diff, fastarm);
    if (fastarm < 0) fastarm = 0;
    double bt_sig = fastarm * fastarm;
    if (bt_sig > 500) bt_sig = 500;
    *fire_out = peakdet_update(&d->pdet, &bt_sig);
    *cvalp = cval; *cval_d_out = cval_d; *val_to_demod_out = val_to_demod;
}

static void boqpsk_write(jo_burst *d, const int16_t *ptr, long n) /* burstoqpskdemodulator.cpp:315-737 (mono) */
{
    double lastmse = d->mse;
    if (n > d->hfircap) { d->hfircap = n; d->hfirbuff = (cpx *)realloc(d->hfirbuff, sizeof(cpx) * (size_t)n); }
    for (long i = 0; i < n; i++) { d->hfirbuff[i].re = ((double)ptr[i]) / 32768.0; d->hfirbuff[i].im = 0; }
    fastfir_update(&d->hfir, d->hfirbuff, n);
    const double SPS = d->SamplesPerSymbol;
    const cpx imag = {0, 1};
    for (long i = 0; i < n; i++)
    {
        const long sample = d->nsamples_total + i;
        cpx cval = d->hfirbuff[i], cval_d; double val_to_demod; int fire;
        burst_front_end(d, &cval, &cval_d, &val_to_demod, &fire);
        if (fire) { d->tridentbuffer_ptr = 0; if (d->trace) burst_event(d, sample, JO_EV_PEAK, 0); }
        if (d->tridentbuffer_ptr < d->tridentbuffer_sz) { d->tridentbuffer[d->tridentbuffer_ptr] = cval_d.re; d->tridentbuffer_ptr++; }
        else if (d->tridentbuffer_ptr == d->tridentbuffer_sz) { d->tridentbuffer_ptr++; boqpsk_trident_check(d, sample); }

        /* mix + rrc :517-521 */
        cpx cval_dd = cscale(wt_cis(&d->mixer2), (d->vol_gain * val_to_demod));
        cpx sig2;
        sig2.re = fir_update_and_process(d->fir_re, cval_dd.re);
        sig2.im = fir_update_and_process(d->fir_im, cval_dd.im);

        /* sample counting :523-544 */
        if (d->startstop > 0)
        {
            d->startstop--;
            if (d->cntr < 1000000) d->cntr++;
            if (d->mse < 0.75) d->startstop = d->startstopstart;
        }
        if (d->startstop == 0) { d->startstop--; burst_event(d, sample, JO_EV_SIGNAL, 0.0); }
        if ((d->cntr > ((256 - 10) * SPS)) && d->insertpreamble) { d->rx[d->nrx++] = -1; d->insertpreamble = 0; }

        /* symbol tone in preamble :547-566 */
        if ((d->cntr > SPS * (128 + 10)) && (d->cntr < ((256 - 10) * SPS)))
        {
            double progress = (((double)d->cntr) - (SPS * (128 + 10))) / (((256 - 10) * SPS) - (SPS * (128 + 10)));
            cpx symboltone_pt = cmul(cmul(sig2, d->symboltone_rotator), imag);
            double er = tanh(symboltone_pt.im) * (symboltone_pt.re);
            d->symboltone_rotator = cmul(d->symboltone_rotator, cexp_i(er * 0.01));
            d->symboltone_averotator.re = d->symboltone_averotator.re * 0.95 + 0.05 * d->symboltone_rotator.re;
            d->symboltone_averotator.im = d->symboltone_averotator.im * 0.95 + 0.05 * d->symboltone_rotator.im;
            symboltone_pt.im = delay_update(&d->a1, symboltone_pt.re);
            d->carrier_rotation_est = atan2(d->symboltone_averotator.im, d->symboltone_averotator.re);
            cpx cj; cj.re = symboltone_pt.re; cj.im = -symboltone_pt.im;
            cpx t = cmul(wt_cis(&d->st_osc_quarter), cj);
            double st_err = atan2(t.im, t.re);
            st_err *= 1.5 * (1.0 - progress * progress);
            wt_advance_fraction_of_wave(&d->st_osc_quarter, -(1.0 / (2.0 * M_PI)) * st_err * 0.1);
            wt_set_phase_deg(&d->st_osc, (360.0 * d->st_osc_quarter.WTptr / ((double)WTSIZE)) * 4.0 + (360.0 * d->ee));
        }

        /* correct carrier phase :570-573 */
        sig2 = cmul(sig2, d->symboltone_averotator);
        d->rotator = cmul(d->rotator, cexp_i(d->rotator_freq));
        sig2 = cmul(sig2, d->rotator);
        double sig2abs = hypot(sig2.re, sig2.im);
        oqpsk_ebno_update(&d->ebno, sig2abs);
        if (fabs(d->cntr - ((128.0 + 128.0 + 128.0) * SPS)) < 0.5) burst_event(d, sample, JO_EV_EBNO, d->ebno.EbNo);
        sig2 = cscale(sig2, agc_update(d->agc2, sig2abs));
        double abval = hypot(sig2.re, sig2.im);
        if (abval > 2.84) sig2 = cscale(sig2, (2.84 / abval)); /* (2.84/abval)*sig2 */

        /* symbol timer :592-612 */
        double st_diff = delay_update(&d->delays, abval * abval) - (abval * abval);
        double st_d1out = delay_update(&d->delayt41, st_diff);
        double st_d2out = delay_update(&d->delayt42, st_d1out);
        double st_eta = (st_d2out - st_diff) * st_d1out;
        iir_update(&d->st_iir_resonator, st_eta);
        if (d->cntr > SPS * (128 + 128)) st_eta = d->st_iir_resonator.y;
        cpx st_m1; st_m1.re = st_eta; st_m1.im = -delay_update(&d->delayt8, st_eta);
        cpx st_out = cmul(wt_cis(&d->st_osc), st_m1);
        double st_angle_error = atan2(st_out.im, st_out.re);
        if (d->cntr > SPS * (128 + 64))
        {
            wt_increase_freq(&d->st_osc, -st_angle_error * 0.00000001);
            wt_advance_fraction_of_wave(&d->st_osc, -st_angle_error * 0.01 / 360.0);
        }
        if (d->st_osc.freq < (d->st_osc_ref.freq - 0.1)) wt_setfreq(&d->st_osc, (d->st_osc_ref.freq - 0.1));
        if (d->st_osc.freq > (d->st_osc_ref.freq + 0.1)) wt_setfreq(&d->st_osc, (d->st_osc_ref.freq + 0.1));

        /* sample times :615-724 */
        if (wt_if_have_passed_point(&d->st_osc, d->ee))
        {
            double pt_last = d->st_osc.FractionOfSampleItPassesBy;
            double pt_this = 1.0 - pt_last;
            cpx pt;
            pt.re = pt_this * sig2.re + pt_last * d->sig2_last.re;
            pt.im = pt_this * sig2.im + pt_last * d->sig2_last.im;
            double twospeed = -4.0 * ((fmod((360.0 * d->st_osc_quarter.WTptr / ((double)WTSIZE)) * 2.0 + (360.0 * d->ee * 0.5), 360.0) / 360.0) - (0.34046 + 0.4111 * d->ee));
            int even = 1;
            if (twospeed < 0) even = 0;
            d->yui++; d->yui %= 2;
            if (d->cntr < ((128 + 128) * SPS))
            {
                if ((even && d->yui == 1) || (!even && d->yui == 0)) { d->yui++; d->yui %= 2; }
            }
            if (!d->yui) d->pt_d = pt;
            else
            {
                cpx pt_qpsk; pt_qpsk.re = pt.re; pt_qpsk.im = d->pt_d.im;
                double ct_xt = tanh(pt.im) * pt.re;
                double ct_xt_d = tanh(d->pt_d.re) * d->pt_d.im;
                double ct_ec = ct_xt_d - ct_xt;
                if (ct_ec > M_PI) ct_ec = M_PI;
                if (ct_ec < -M_PI) ct_ec = -M_PI;
                if (ct_ec > M_PI_2) ct_ec = M_PI_2;
                if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
                if (d->cntr > ((128 + 10) * SPS))
                {
                    d->rotator = cmul(d->rotator, cexp_i(ct_ec * 0.1));
                    d->rotator_freq = d->rotator_freq + ct_ec * 0.0001;
                }
                /* pointbuff bookkeeping is GUI only */
                if (d->cntr > ((128 + 10) * SPS))
                {
                    double tda = (fabs(pt_qpsk.re) - 1.0);
                    double tdb = (fabs(pt_qpsk.im) - 1.0);
                    d->mse = ma_update(d->msema, (tda * tda) + (tdb * tdb));
                }
                if (d->startstop > 0)
                {
                    if (d->capture_symbols) { double row[3] = {pt_qpsk.re, pt_qpsk.im, d->mse}; gpush(&d->symbols, row, sizeof(row)); }
                    int ibit = qRound_(0.75 * pt_qpsk.im * 127.0 + 128.0);
                    if (ibit > 255) ibit = 255;
                    if (ibit < 0) ibit = 0;
                    d->rx[d->nrx++] = (short)(unsigned char)ibit;
                    ibit = qRound_(0.75 * pt_qpsk.re * 127.0 + 128.0);
                    if (ibit > 255) ibit = 255;
                    if (ibit < 0) ibit = 0;
                    d->rx[d->nrx++] = (short)(unsigned char)ibit;
                    if (d->nrx >= 32)
                    {
                        if (!d->sql || d->mse < d->signalthreshold || lastmse < d->signalthreshold) burst_emit_soft(d);
                        d->nrx = 0;
                    }
                }
            }
        }
        d->sig2_last = sig2;
        wt_next(&d->mixer2);
        wt_next(&d->st_osc);
        wt_next(&d->st_osc_ref);
        wt_next(&d->st_osc_quarter);
    }
    d->nsamples_total += n;
}

/* ---------------- burst MSK ---------------- */
static void bmsk_make_filters(jo_burst *d)
{
    int ntaps = (int)(2 * d->SamplesPerSymbol);
    fir_free(d->fir_re); fir_free(d->fir_im);
    d->fir_re = fir_new(ntaps); d->fir_im = fir_new(ntaps);
    for (int i = 0; i < 2 * d->SamplesPerSymbol; i++)
    {
        double v = sin(M_PI * i / (2.0 * d->SamplesPerSymbol)) / (2.0 * d->SamplesPerSymbol);
        if (i >= 0 && i < ntaps) { d->fir_re->points[i] = v; d->fir_im->points[i] = v; }
    }
}
static void bmsk_center_freq_changed(jo_burst *d, double freq_center, long sample) /* burstmskdemodulator.cpp:327-343 */
{
    if (freq_center < (0.75 * d->fb)) freq_center = 0.75 * d->fb;
    if (freq_center > (d->Fs / 2.0 - 0.75 * d->fb)) freq_center = d->Fs / 2.0 - 0.75 * d->fb;
    wt_setfreq_sr(&d->mixer_center, freq_center, (int)d->Fs);
    if (d->afc) wt_setfreq(&d->mixer2, d->mixer_center.freq);
    if ((d->mixer2.freq - d->mixer_center.freq) > (d->lockingbw / 2.0)) wt_setfreq(&d->mixer2, d->mixer_center.freq + (d->lockingbw / 2.0));
    if ((d->mixer2.freq - d->mixer_center.freq) < (-d->lockingbw / 2.0)) wt_setfreq(&d->mixer2, d->mixer_center.freq - (d->lockingbw / 2.0));
    burst_event(d, sample, JO_EV_FREQ, d->mixer2.freq);
}
/* CenterFreqChangedSlot as a caller reaches it (JAERO/mainwindow.cpp:415 wires the spectrum display to it): BurstMskDemodulator acts
 * (burstmskdemodulator.cpp:327-342), BurstOqpskDemodulator's slot is empty (burstoqpskdemodulator.cpp:284-289) */
void jo_burst_center_freq_changed(jo_burst *d, double freq_center)
{
    if (d->kind == JO_KIND_BURST_MSK) bmsk_center_freq_changed(d, freq_center, d->nsamples_total);
}
static void bmsk_set_settings(jo_burst *d, const jo_settings *s) /* burstmskdemodulator.cpp:150-325 */
{
    d->Fs = s->Fs; d->lockingbw = s->lockingbw; d->fb = s->fb;
    if (d->fb > d->Fs) d->fb = d->Fs;
    d->freq_center = s->freq_center;
    if (d->freq_center > ((d->Fs / 2.0) - (d->lockingbw / 2.0))) d->freq_center = ((d->Fs / 2.0) - (d->lockingbw / 2.0));
    d->signalthreshold = s->signalthreshold;
    d->SamplesPerSymbol = (int)(d->Fs / d->fb);
    const double SPS = d->SamplesPerSymbol;
    wt_setfreq_sr(&d->mixer_center, d->freq_center, (int)d->Fs);
    wt_setfreq_sr(&d->mixer2, d->freq_center, (int)d->Fs);
    bmsk_make_filters(d);
    agc_free(d->agc); d->agc = agc_new(1, d->Fs);
    agc_free(d->agc2); d->agc2 = NULL;
    ma_free(d->ebno.E); ma_free(d->ebno.E2);
    d->ebno.E = ma_new((int)(0.15 * d->Fs)); d->ebno.E2 = ma_new((int)(0.15 * d->Fs)); d->ebno.EbNo = 0; /* uninitialised in ref */
    hilbert_set_size(&d->hfir, 2048);
    d->pointbuff_ptr = 0;
    d->mse = 10.0;
    burst_event(d, d->nsamples_total, JO_EV_FREQ, d->mixer2.freq);
    delay_set(&d->a1, SPS / 2);
    d->symboltone_averotator.re = 1; d->symboltone_averotator.im = 0;
    d->rotator.re = 1; d->rotator.im = 0;
    d->cntr = 0;
    d->N = 4096 * 4 * 2;
    if (d->fftr) fft_free(d->fftr);
    d->fftr = fft_plan(d->N);
    if (d->fb >= 1200)
    {
        cdelay_set(&d->bt_d1, 1.0 * SPS);
        cma_set_length(&d->bt_ma1, qRound_(126.0 * SPS));
        ma_free(d->mav1); d->mav1 = ma_new((int)(SPS * 126));
        ddelay_set(&d->bt_ma_diff, SPS * 126);
        peakdet_set(&d->pdet, (int)(SPS * 126.0 / 2.0), 0.1);
        d->tridentbuffer_sz = qRound_((200.0) * SPS);
        delaything_set_length(&d->d1, (int)(((int)289 * SPS) + 20));
        dthing_set_length(&d->d2, (int)(qRound_(72 + 120.0) * SPS));
        d->startstopstart = (int)(SPS * (500));
        d->endRotation = (int)((120 + 37) * SPS);
        set_resonator(&d->st_iir_resonator, 2.617308727964618e-04, 0, -2.617308727964618e-04, 1, -1.993312819378528, 0.999476538254407);
        d->ee = 0.025;
        iir_init(&d->st_iir_resonator);
        d->startProcessing = 120;
    }
    else
    {
        ma_free(d->mav1); d->mav1 = ma_new((int)(SPS * 150));
        ddelay_set(&d->bt_ma_diff, SPS * 150);
        cdelay_set(&d->bt_d1, 1.0 * SPS);
        cma_set_length(&d->bt_ma1, qRound_(150.0 * SPS));
        peakdet_set(&d->pdet, (int)(SPS * 150.0 / 2.0), 0.2);
        d->tridentbuffer_sz = qRound_((224) * SPS);
        delaything_set_length(&d->d1, (int)(((int)397 * SPS) + 20));
        dthing_set_length(&d->d2, qRound_((72 + 150.0) * SPS));
        d->startstopstart = (int)(SPS * (500));
        set_resonator(&d->st_iir_resonator, 0.001307286451699, 0, -0.001307286451699, 1, -1.991228154418550, 0.997385427096603);
        iir_init(&d->st_iir_resonator);
        d->ee = 0.015;
        d->startProcessing = 150;
        d->endRotation = (int)((d->startProcessing + 56) * SPS);
    }
    d->agc2 = agc_new(SPS * 128.0 / d->Fs, d->Fs);
    delay_set(&d->delayt8, (SPS) / 2.0);
    d->tridentbuffer = (double *)realloc(d->tridentbuffer, sizeof(double) * (size_t)d->tridentbuffer_sz);
    d->tridentbuffer_ptr = 0;
    d->in = (double *)realloc(d->in, sizeof(double) * (size_t)d->N);
    d->out_base = (cpx *)realloc(d->out_base, sizeof(cpx) * (size_t)d->N);
    d->out_top = (cpx *)realloc(d->out_top, sizeof(cpx) * (size_t)d->N);
    wt_setfreq_sr(&d->st_osc, d->fb / 2.0, (int)d->Fs);
    wt_setfreq_sr(&d->st_osc_half, d->fb / 2.0, (int)d->Fs);
    d->dcd = 0;
    delaything_set_length(&d->delayedsmpl, (int)SPS);
}
static void bmsk_ctor(jo_burst *d) /* burstmskdemodulator.cpp:9-84 */
{
    d->afc = 1;
    d->Fs = 48000; d->lockingbw = 1800; d->fb = 1200; d->signalthreshold = 0.6;
    d->SamplesPerSymbol = d->Fs / d->fb;
    bmsk_make_filters(d);
    wt_init(&d->mixer_center); wt_init(&d->mixer2); wt_init(&d->st_osc); wt_init(&d->st_osc_half);
    wt_setfreq_sr(&d->mixer_center, 1000, (int)d->Fs);
    wt_setfreq_sr(&d->mixer2, 1000, (int)d->Fs);
    d->mse = 10.0;
    d->startstop = -1;
    wt_setfreq_sr(&d->st_osc, d->fb / 2, (int)d->Fs);
    wt_setfreq_sr(&d->st_osc_half, d->fb / 2.0, (int)d->Fs);
    d->pt_d.re = d->pt_d.im = 0;
    d->msema = ma_new(75);
    d->pointmean = ma_new(100);
    d->dcd = 0;
    d->diff_lastsoftstate = -1; /* DiffDecode ctor DSP.cpp:509-515 */
    d->symboltone_rotator.re = 1; d->symboltone_rotator.im = 0; /* uninitialised until the first burst; only read inside a burst */
    d->vol_gain = 1; d->rotator_freq = 0;                         /* same */
}

static void bmsk_trident_check(jo_burst *d, long sample) /* burstmskdemodulator.cpp:444-569 */
{
    const double SPS = d->SamplesPerSymbol;
    int size_base = 126, size_top = 74;
    if (d->fb < 1200) { size_base = 150; size_top = 74; }
    const int N = d->N;
    const int nb = qRound_(size_base * SPS), nt = qRound_(size_top * SPS);
    for (int k = 0; k < N; k++) d->in[k] = (k < nb) ? d->tridentbuffer[k] : 0;
    fftr_transform(d, d->in, d->out_base);
    for (int k = 0; k < N; k++) d->in[k] = (k < nt && nb + k < d->tridentbuffer_sz) ? d->tridentbuffer[nb + k] : 0;
    fftr_transform(d, d->in, d->out_top);
    double hzperbin = d->Fs / ((double)N);
    int peakspacingbins = qRound_((0.5 * d->fb) / hzperbin);
    int minvalbin = 0;
    double minval = 0;
    for (int i = 0; i < N / 2; i++)
    {
        double a = hypot(d->out_base[i].re, d->out_base[i].im);
        if (a > minval) { minval = a; minvalbin = i; }
    }
    double maxtop = 0, maxtophigh = 0;
    int maxtoppos = 0, maxtopposhigh = 0;
    for (int i = 0; i < N / 2; i++)
    {
        if (i > 50)
        {
            double a = hypot(d->out_top[i].re, d->out_top[i].im);
            if ((i < minvalbin - (peakspacingbins / 2)) && a > maxtop) { maxtop = a; maxtoppos = i; }
            if ((i > minvalbin + (peakspacingbins / 2)) && a > maxtophigh) { maxtophigh = a; maxtopposhigh = i; }
        }
    }
    int distfrompeak = abs(maxtoppos - minvalbin);
    int ok = (minval > 500.0 && abs(distfrompeak - peakspacingbins) < abs(peakspacingbins / 20) && !(d->dcd) && !(d->cntr > 0 && d->cntr < (500 * SPS)));
    if (d->trace) burst_event(d, sample, JO_EV_TRIDENT, ok ? minval : -minval);
    if (ok)
    {
        d->vol_gain = 1.4142 * (500.0 / (minval / 3));
        double carrierphase = atan2(d->out_base[minvalbin].im, d->out_base[minvalbin].re) - (M_PI / 4.0);
        wt_set_phase_deg(&d->mixer2, (180.0 / M_PI) * carrierphase);
        wt_setfreq(&d->mixer2, ((maxtopposhigh + maxtoppos) / 2) * hzperbin);
        bmsk_center_freq_changed(d, ((maxtopposhigh + maxtoppos) / 2) * hzperbin, sample);
        burst_event(d, sample, JO_EV_FREQ, d->mixer2.freq);
        ma_zero(d->pointmean);
        d->pointbuff_ptr = 0;
        d->startstop = d->startstopstart;
        d->cntr = 0;
        burst_event(d, sample, JO_EV_SIGNAL, 1.0);
        d->nrx = 0;
        d->rx[d->nrx++] = -1;
        d->mse = 0;
        ma_zero(d->msema);
        d->symboltone_averotator.re = 1; d->symboltone_averotator.im = 0;
        d->symboltone_rotator.re = 1; d->symboltone_rotator.im = 0;
        d->rotator.re = 1; d->rotator.im = 0;
        d->rotator_freq = 0;
        d->carrier_rotation_est = 0;
        iir_init(&d->st_iir_resonator);
        wt_set_phase_deg(&d->st_osc, 0);
        wt_set_phase_deg(&d->st_osc_half, 0);
    }
}

static void bmsk_write(jo_burst *d, const int16_t *ptr, long n) /* burstmskdemodulator.cpp:371-754 */
{
    if (n > d->hfircap) { d->hfircap = n; d->hfirbuff = (cpx *)realloc(d->hfirbuff, sizeof(cpx) * (size_t)n); }
    for (long i = 0; i < n; i++) { d->hfirbuff[i].re = ((double)ptr[i]) / 32768.0; d->hfirbuff[i].im = 0; }
    fastfir_update(&d->hfir, d->hfirbuff, n);
    const double SPS = d->SamplesPerSymbol;
    const cpx imag = {0, 1};
    for (long i = 0; i < n; i++)
    {
        const long sample = d->nsamples_total + i;
        cpx cval = d->hfirbuff[i], cval_d; double val_to_demod; int fire;
        burst_front_end(d, &cval, &cval_d, &val_to_demod, &fire);
        if (fire) { d->tridentbuffer_ptr = 0; if (d->trace) burst_event(d, sample, JO_EV_PEAK, 0); }
        if (d->tridentbuffer_ptr < d->tridentbuffer_sz) { d->tridentbuffer[d->tridentbuffer_ptr] = cval_d.re; d->tridentbuffer_ptr++; }
        else if (d->tridentbuffer_ptr == d->tridentbuffer_sz) { d->tridentbuffer_ptr++; bmsk_trident_check(d, sample); }

        /* sample counting :572-598 */
        if (d->startstop > 0)
        {
            if (d->cntr >= (d->startProcessing * SPS)) d->startstop--;
            if (d->cntr < 1000000) d->cntr++;
            if (d->mse < d->signalthreshold) d->startstop = d->startstopstart;
        }
        if (d->startstop == 0)
        {
            d->startstop--;
            burst_event(d, sample, JO_EV_SIGNAL, 0.0);
            d->cntr = 0;
            d->mse = 1;
        }
        if (d->startstop > 0 || d->mse < d->signalthreshold)
        {
            /* mixer2.WTCISValue()*(val_to_demod)*vol_gain : (complex*double)*double */
            cval = cscale(cscale(wt_cis(&d->mixer2), val_to_demod), d->vol_gain);
            cpx sig2;
            sig2.re = fir_update_and_process(d->fir_re, cval.re);
            sig2.im = fir_update_and_process(d->fir_im, cval.im);
            if (d->cntr > (d->startProcessing * SPS) && d->cntr < d->endRotation)
            {
                cpx symboltone_pt = cmul(cmul(sig2, d->symboltone_rotator), imag);
                double er = tanh(symboltone_pt.im) * (symboltone_pt.re);
                d->symboltone_rotator = cmul(d->symboltone_rotator, cexp_i(er * 0.5));
                d->symboltone_averotator.re = d->symboltone_averotator.re * 0.999 + 0.001 * d->symboltone_rotator.re;
                d->symboltone_averotator.im = d->symboltone_averotator.im * 0.999 + 0.001 * d->symboltone_rotator.im;
                symboltone_pt.im = delay_update(&d->a1, symboltone_pt.re);
                double progress = (double)d->cntr - (SPS * (d->startProcessing));
                double goal = d->endRotation - (SPS * d->startProcessing);
                progress = progress / goal;
                cpx cj; cj.re = symboltone_pt.re; cj.im = -symboltone_pt.im;
                cpx t = cmul(wt_cis(&d->st_osc_half), cj);
                double st_err = atan2(t.im, t.re);
                st_err *= 0.5 * (1.0 - progress * progress);
                wt_advance_fraction_of_wave(&d->st_osc_half, -(1.0 / (2.0 * M_PI)) * st_err * 0.05);
                wt_set_phase_deg(&d->st_osc, (360.0 * d->st_osc_half.WTptr / ((double)WTSIZE)) + (360.0 * (1.0 - d->ee)));
            }
            sig2 = cmul(sig2, d->symboltone_averotator);
            d->rotator = cmul(d->rotator, cexp_i(d->rotator_freq));
            sig2 = cmul(sig2, d->rotator);
            msk_ebno_update(&d->ebno, hypot(sig2.re, sig2.im));
            if (d->cntr == d->endRotation + (200 * SPS)) burst_event(d, sample, JO_EV_EBNO, d->ebno.EbNo);
            sig2 = cscale(sig2, agc_update(d->agc2, hypot(sig2.re, sig2.im)));
            double abval = hypot(sig2.re, sig2.im);
            if (abval > 2.84) sig2 = cscale(sig2, (2.84 / abval));
            cpx pt_d = delaything_update_dont_touch(&d->delayedsmpl, sig2);
            cpx pt_msk; pt_msk.re = sig2.re; pt_msk.im = pt_d.im;
            double st_eta = hypot(pt_msk.re, pt_msk.im);
            st_eta = iir_update(&d->st_iir_resonator, st_eta);
            cpx st_m1; st_m1.re = st_eta; st_m1.im = -delay_update(&d->delayt8, st_eta);
            cpx st_out = cmul(wt_cis(&d->st_osc), st_m1);
            double st_angle_error = atan2(st_out.im, st_out.re);
            if (d->cntr > d->endRotation) wt_advance_fraction_of_wave(&d->st_osc, -st_angle_error * 0.002 / 360.0);
            if (wt_if_have_passed_point(&d->st_osc, d->ee))
            {
                double ct_xt = tanh(sig2.im) * sig2.re;
                double ct_xt_d = tanh(pt_d.re) * pt_d.im;
                double ct_ec = ct_xt_d - ct_xt;
                if (ct_ec > M_PI) ct_ec = M_PI;
                if (ct_ec < -M_PI) ct_ec = -M_PI;
                if (ct_ec > M_PI_2) ct_ec = M_PI_2;
                if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
                if (d->cntr > (d->startProcessing * SPS))
                {
                    d->rotator = cmul(d->rotator, cexp_i(ct_ec * 0.25));
                    if (d->cntr > d->endRotation) d->rotator_freq = d->rotator_freq + ct_ec * 0.0001;
                }
                if (d->cntr > (d->startProcessing * SPS))
                {
                    double tda = (fabs((pt_msk.re * 0.75)) - 1.0);
                    double tdb = (fabs((pt_msk.im * 0.75)) - 1.0);
                    d->mse = ma_update(d->msema, (tda * tda) + (tdb * tdb));
                }
                if (d->capture_symbols) { double row[3] = {pt_msk.re, pt_msk.im, d->mse}; gpush(&d->symbols, row, sizeof(row)); }
                double imagin = diffdecode_update_soft(&d->diff_lastsoftstate, pt_msk.im);
                int ibit = qRound_((imagin) * 127.0 + 128.0);
                if (ibit > 255) ibit = 255;
                if (ibit < 0) ibit = 0;
                d->rx[d->nrx++] = (short)(unsigned char)ibit;
                double real = diffdecode_update_soft(&d->diff_lastsoftstate, pt_msk.re);
                real = -real;
                ibit = qRound_((real) * 127.0 + 128.0);
                if (ibit > 255) ibit = 255;
                if (ibit < 0) ibit = 0;
                d->rx[d->nrx++] = (short)(unsigned char)ibit;
                if (d->nrx >= 12) burst_emit_soft(d);
            }
            wt_next(&d->st_osc);
            wt_next(&d->st_osc_half);
            wt_next(&d->mixer2);
            wt_next(&d->mixer_center);
        }
    }
    d->nsamples_total += n;
}

/* ---------------- public API ---------------- */
jo_burst *jo_burst_create(const jo_settings *s)
{
    trig_init();
    jo_burst *d = (jo_burst *)calloc(1, sizeof(jo_burst));
    d->kind = s->kind;
    if (d->kind == JO_KIND_BURST_OQPSK) { boqpsk_ctor(d); boqpsk_set_settings(d, s); }
    else { bmsk_ctor(d); bmsk_set_settings(d, s); }
    return d;
}
void jo_burst_destroy(jo_burst *d)
{
    if (!d) return;
    fastfir_free(&d->hfir); free(d->hfirbuff);
    agc_free(d->agc); agc_free(d->agc2);
    free(d->d1.buffer); free(d->d2.buffer); free(d->bt_d1.buff); free(d->bt_ma_diff.buff); free(d->bt_ma1.buf); ma_free(d->mav1);
    free(d->pdet.d1.buffer); free(d->pdet.d2.buffer); free(d->pdet.d3.buffer);
    free(d->tridentbuffer); if (d->fftr) fft_free(d->fftr); free(d->out_base); free(d->out_top); free(d->out_abs_diff); free(d->in);
    fir_free(d->fir_re); fir_free(d->fir_im);
    ma_free(d->ebno.E); ma_free(d->ebno.E2); ma_free(d->msema); ma_free(d->pointmean);
    free(d->delayedsmpl.buffer);
    free(d->soft.p); free(d->events.p); free(d->symbols.p);
    free(d);
}
/* setSettings on the LIVE object (a user pressing OK in the settings dialog): the same functions the constructor path runs -- new AGCs, EbNo
 * meter and moving averages, Hilbert filter / peak detector / trident fill restarted, and DelayThing::setLength (DSP.h:447-453) keeping the
 * old CONTENTS of d1 / d2 / the peak detector's lines (delayedsmpl for burst MSK) with the pointer back at zero.  The bit rate and sample rate
 * of a live object may change too (all lengths follow). */
void jo_burst_set_settings(jo_burst *d, const jo_settings *s)
{
    if (d->kind == JO_KIND_BURST_OQPSK) boqpsk_set_settings(d, s); else bmsk_set_settings(d, s);
}
void jo_burst_set_flags(jo_burst *d, int afc, int sql, int cpu_reduce) { d->afc = afc; d->sql = sql; d->cpuReduce = cpu_reduce; }
void jo_burst_set_dcd(jo_burst *d, int dcd) { d->dcd = dcd; }
void jo_burst_trace(jo_burst *d, int on) { d->trace = on; }
long jo_burst_write(jo_burst *d, const int16_t *pcm, long n)
{
    if (n <= 0) return 0;
    if (d->kind == JO_KIND_BURST_OQPSK) boqpsk_write(d, pcm, n); else bmsk_write(d, pcm, n);
    return 2 * n;
}
long jo_burst_take_soft(jo_burst *d, int16_t *dst, long cap) { return gtake(&d->soft, dst, sizeof(int16_t), cap); }
long jo_burst_take_events(jo_burst *d, double *dst, long caprows) { return gtake(&d->events, dst, 3 * sizeof(double), caprows); }
void jo_burst_capture_symbols(jo_burst *d, int on) { d->capture_symbols = on; }
long jo_burst_take_symbols(jo_burst *d, double *dst, long caprows) { return gtake(&d->symbols, dst, 3 * sizeof(double), caprows); }
int jo_burst_pending_soft(jo_burst *d) { return d->nrx; }
double jo_burst_get_mse(jo_burst *d) { return d->mse; }
double jo_burst_get_freq_est(jo_burst *d) { return d->mixer2.freq; }
/* streaming Hilbert filter alone (what hfir.update does to a real input), for the tests that pin the HIP FIR */
typedef struct jo_hilbert { fastfir_t f; } jo_hilbert;
jo_hilbert *jo_hilbert_create(int N) { jo_hilbert *h = (jo_hilbert *)calloc(1, sizeof(jo_hilbert)); hilbert_set_size(&h->f, N); return h; }
void jo_hilbert_destroy(jo_hilbert *h) { if (!h) return; fastfir_free(&h->f); free(h); }
int jo_hilbert_latency(jo_hilbert *h) { return h->f.L; }
void jo_hilbert_update(jo_hilbert *h, const int16_t *pcm, long n, double *out_re_im)
{
    cpx *o = (cpx *)out_re_im;
    for (long i = 0; i < n; i++) { o[i].re = ((double)pcm[i]) / 32768.0; o[i].im = 0; }
    fastfir_update(&h->f, o, n);
}
