/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's continuous demodulator hot path, one function per reference
 * function, each citing the file:line (relative to /root/reference/) it follows:
 *   OqpskDemodulator  ctor/setSettings/writeData/FreqOffsetEstimateSlot/CenterFreqChangedSlot
 *                     (JAERO/oqpskdemodulator.cpp:8-117,175-289,334-627,629-677,291-310)
 *   MskDemodulator    the same five (JAERO/mskdemodulator.cpp:9-84,135-263,313-488,490-519,265-282)
 *   CoarseFreqEstimate (JAERO/coarsefreqestimate.cpp:5-137) and FFTWrapper (JAERO/fftwrapper.cpp:19-35)
 *   DSP primitives    (JAERO/DSP.h, JAERO/DSP.cpp)
 * Arithmetic is restated operation for operation (same order, no FMA contraction, glibc libm), so that it is
 * BIT-IDENTICAL to the unmodified reference built as oracle/_ref/jaero_ref on the same machine; that identity
 * is what tests/test_oracle_vs_ref.py checks (and the committed tests/golden/ vectors were produced by _ref).
 * The FFT is the same radix-2 as oracle/ref/shim/jfft.h because JFFT itself is not in the reference tree
 * (its conventions are pinned by JAERO/tests/fft*wrapper_tests.cpp golden vectors).
 *
 * Function-local statics of the reference (maxval, sig2_last, yui, pt_d, slowdown, countdown, countdown2:
 * JAERO/oqpskdemodulator.cpp:393,487,496,498,540,641,652; JAERO/mskdemodulator.cpp:434,493) are per-object
 * state here, which is what one-reference-process-per-channel computes.
 * GUI-only work (spectrum ring, scatter points, PeakVolume, QElapsedTimer gating) is omitted: it does not feed
 * back into the signal path.  The 8400 bps C-channel branches (fb==8400: JFastFir prefilter, centre-weighted coarse window,
 * alpha-0.6 pulse, its own resonator and carrier-loop gains) are restated too; JFastFir itself is JFFT's
 * (not in the reference tree): it is restated as oracle/ref/shim/jfft.h builds it, overlap-add with block nfft-K+1, the
 * behaviour JAERO/tests/jfastfir_tests.cpp pins with exactly this kernel (RRC 0.6, 2049 taps, nfft 4096).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  Do not add -ffast-math / -march=native.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "jaero_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif

typedef struct { double re, im; } cpx;

static inline cpx cmul(cpx a, cpx b) { cpx r; r.re = a.re * b.re - a.im * b.im; r.im = a.re * b.im + a.im * b.re; return r; }
static inline cpx cscale(cpx a, double s) { cpx r; r.re = a.re * s; r.im = a.im * s; return r; }

/* Qt 5.9 qglobal.h:525-526 */
static inline int qRound_(double d) { return d >= 0.0 ? (int)(d + 0.5) : (int)(d - (double)((int)(d - 1)) + 0.5) + (int)(d - 1); }

/* ------------------------------------------------------------------ TrigLookUp (DSP.cpp:11-30) */
#define WTSIZE 19999
#define WTSIZE_3_4 (3.0 * WTSIZE / 4.0)
#define WTSIZE_1_4 (1.0 * WTSIZE / 4.0)
static cpx CISWT[WTSIZE];
static int cis_ready = 0;
static void trig_init(void)
{
    if (cis_ready) return;
    for (int i = 0; i < WTSIZE; i++)
    {
        double s = sin(2 * M_PI * ((double)i) / WTSIZE);
        double c = sin(M_PI_2 + 2 * M_PI * ((double)i) / WTSIZE);
        CISWT[i].re = c;
        CISWT[i].im = s;
    }
    cis_ready = 1;
}
void jo_cis_table(double *dst)
{
    trig_init();
    for (int i = 0; i < WTSIZE; i++) { dst[2 * i] = CISWT[i].re; dst[2 * i + 1] = CISWT[i].im; }
}

/* ------------------------------------------------------------------ WaveTable (DSP.h:40-81, DSP.cpp:32-265) */
typedef struct
{
    double WTptr, WTstep, freq, samplerate, last_WTptr, FractionOfSampleItPassesBy;
} wavetable;

static void wt_init(wavetable *w) /* DSP.cpp:32-40 */
{
    w->last_WTptr = 0; w->samplerate = 48000; w->freq = 1000;
    w->WTstep = (1000.0) * WTSIZE / (48000);
    w->WTptr = 0; w->FractionOfSampleItPassesBy = 0.0;
}
static void wt_setfreq_sr(wavetable *w, double freq, int samplerate) /* DSP.cpp:142-149 */
{
    w->freq = freq;
    w->samplerate = samplerate;
    if (w->freq < 0) w->freq = 0;
    w->WTstep = (w->freq) * ((double)WTSIZE) / ((float)w->samplerate);
    while (((int)w->WTptr) >= WTSIZE) w->WTptr -= WTSIZE;
}
static void wt_setfreq(wavetable *w, double freq) /* DSP.cpp:151-156 */
{
    w->freq = freq;
    if (w->freq < 0) w->freq = 0;
    w->WTstep = (w->freq) * ((double)WTSIZE) / w->samplerate;
}
static void wt_next(wavetable *w) /* DSP.cpp:70-77 */
{
    if (w->WTstep < 0) w->WTstep = 0;
    w->last_WTptr = w->WTptr;
    w->WTptr += w->WTstep;
    while (((int)w->WTptr) >= WTSIZE) w->WTptr -= WTSIZE;
}
static cpx wt_cis(const wavetable *w) /* DSP.cpp:79-85 */
{
    int tint = (int)w->WTptr;
    if (tint >= WTSIZE) tint = 0;
    if (tint < 0) tint = WTSIZE - 1;
    return CISWT[tint];
}
static cpx wt_cis_conj(const wavetable *w) /* DSP.cpp:87-93 */
{
    cpx c = wt_cis(w);
    c.im = -c.im;
    return c;
}
static void wt_increase_freq(wavetable *w, double freq_hz) /* DSP.cpp:163-167 */
{
    freq_hz += w->freq;
    wt_setfreq(w, freq_hz);
}
static void wt_set_phase_deg(wavetable *w, double phase_deg) /* DSP.cpp:175-180 */
{
    phase_deg = fmod(phase_deg, 360.0);
    while (phase_deg < 0) phase_deg += 360.0;
    w->WTptr = (phase_deg / 360.0) * ((double)WTSIZE);
}
static double wt_get_phase_deg(const wavetable *w) { return (360.0 * w->WTptr / ((double)WTSIZE)); } /* DSP.cpp:192-195 */
static void wt_increase_phase_deg(wavetable *w, double phase_deg) /* DSP.cpp:169-173 */
{
    phase_deg += (360.0 * w->WTptr / ((double)WTSIZE));
    wt_set_phase_deg(w, phase_deg);
}
static void wt_advance_fraction_of_wave(wavetable *w, double f) /* DSP.h:56 */
{
    w->WTptr += f * WTSIZE;
    while (w->WTptr >= WTSIZE) w->WTptr -= WTSIZE;
    while (w->WTptr < 0) w->WTptr += WTSIZE;
}
static int wt_if_have_passed_point(wavetable *w, double FractionOfWave) /* DSP.cpp:222-238 */
{
    double t_last_WTptr = w->last_WTptr;
    double t_WTptr = w->WTptr;
    double pt = (FractionOfWave * WTSIZE);
    t_last_WTptr -= pt;
    t_WTptr -= pt;
    if (t_last_WTptr < 0.0) t_last_WTptr += WTSIZE;
    if (t_WTptr < 0.0) t_WTptr += WTSIZE;
    if ((t_last_WTptr > WTSIZE_3_4) && (t_WTptr < WTSIZE_1_4))
    {
        w->FractionOfSampleItPassesBy = t_WTptr / w->WTstep;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ FIR (DSP.cpp:271-304) */
typedef struct { double *points, *buff; int NumberOfPoints, buffsize, ptr; } fir_t;
static fir_t *fir_new(int n)
{
    fir_t *f = (fir_t *)calloc(1, sizeof(fir_t));
    f->NumberOfPoints = n; f->buffsize = n + 1;
    f->points = (double *)calloc((size_t)n, sizeof(double));
    f->buff = (double *)calloc((size_t)n + 1, sizeof(double));
    f->ptr = 0;
    return f;
}
static void fir_free(fir_t *f) { if (!f) return; free(f->points); free(f->buff); free(f); }
static double fir_update_and_process(fir_t *f, double sig) /* DSP.cpp:292-304 */
{
    f->buff[f->ptr] = sig;
    f->ptr++; if (f->ptr >= f->buffsize) f->ptr = 0;
    int tptr = f->ptr;
    double outsum = 0;
    for (int i = 0; i < f->NumberOfPoints; i++)
    {
        outsum += f->points[i] * f->buff[tptr];
        tptr++; if (tptr >= f->buffsize) tptr = 0;
    }
    return outsum;
}

/* ------------------------------------------------------------------ AGC (DSP.cpp:357-379) */
typedef struct { int sz, ptr; double sum, val; double *buf; } agc_t;
static agc_t *agc_new(double secs, double Fs)
{
    agc_t *a = (agc_t *)calloc(1, sizeof(agc_t));
    a->sz = (int)round(secs * Fs);
    a->buf = (double *)calloc((size_t)a->sz, sizeof(double));
    return a;
}
static void agc_free(agc_t *a) { if (!a) return; free(a->buf); free(a); }
static double agc_update(agc_t *a, double sig)
{
    a->sum = a->sum - a->buf[a->ptr];
    a->sum = a->sum + fabs(sig);
    a->buf[a->ptr] = fabs(sig);
    a->ptr++; a->ptr %= a->sz;
    a->val = 1.414213562 / fmax(a->sum / ((double)a->sz), 0.000001);
    a->val = fmax(a->val, 0.000001);
    return a->val;
}

/* ------------------------------------------------------------------ MovingAverage (DSP.cpp:388-426) */
typedef struct { int sz, ptr; double sum, val; double *buf; } ma_t;
static ma_t *ma_new(int n)
{
    ma_t *m = (ma_t *)calloc(1, sizeof(ma_t));
    m->sz = n;
    m->buf = (double *)calloc((size_t)n, sizeof(double));
    return m;
}
static void ma_free(ma_t *m) { if (!m) return; free(m->buf); free(m); }
static double ma_update(ma_t *m, double sig) /* :408-416 */
{
    m->sum = m->sum - m->buf[m->ptr];
    m->sum = m->sum + fabs(sig);
    m->buf[m->ptr] = fabs(sig);
    m->ptr++; m->ptr %= m->sz;
    m->val = m->sum / ((double)m->sz);
    return m->val;
}
static double ma_update_signed(ma_t *m, double sig) /* :418-426 */
{
    m->sum = m->sum - m->buf[m->ptr];
    m->sum = m->sum + (sig);
    m->buf[m->ptr] = (sig);
    m->ptr++; m->ptr %= m->sz;
    m->val = m->sum / ((double)m->sz);
    return m->val;
}

/* ------------------------------------------------------------------ MSEcalc (DSP.cpp:434-463) */
typedef struct { ma_t *pointmean, *msema; double mse; } msecalc_t;
static double msecalc_update(msecalc_t *m, cpx pt)
{
    ma_update(m->pointmean, hypot(pt.re, pt.im)); /* std::abs(complex) */
    double mu = m->pointmean->val;
    if (mu < 0.000001) mu = 0.000001;
    double s2 = sqrt(2.0);
    cpx t; t.re = (s2 * pt.re) / mu; t.im = (s2 * pt.im) / mu;
    double tda = (fabs(t.re) - 1.0);
    double tdb = (fabs(t.im) - 1.0);
    m->mse = ma_update(m->msema, (tda * tda) + (tdb * tdb));
    return m->mse;
}

/* ------------------------------------------------------------------ EbNo meters (DSP.cpp:487-505,715-744) */
typedef struct { ma_t *E, *E2; double EbNo, Var, Mean, Fs, fb; } ebno_t;
static double oqpsk_ebno_update(ebno_t *e, double sig) /* :729-744 */
{
    ma_update(e->E2, sig * sig);
    e->Mean = ma_update(e->E, sig);
    double MeanSquared = e->Mean * e->Mean;
    e->Var = (e->E2->val) - (e->E->val * e->E->val);
    e->Var -= (0.024709 * MeanSquared);
    double mvr = (((e->Fs * MeanSquared / (2.0 * e->fb * e->Var))) * 0.13743);
    if (mvr < 0.000000001) mvr = 0.000000001;
    double tebno = 10.0 * log10(mvr);
    if (isnan(tebno)) tebno = 50;
    if (tebno > 50.0) tebno = 50;
    if (tebno < 0.0) tebno = 0;
    e->EbNo = e->EbNo * 0.8 + 0.2 * tebno;
    return e->EbNo;
}
static double msk_ebno_update(ebno_t *e, double sig) /* :493-505 */
{
    ma_update(e->E2, sig * sig);
    e->Mean = ma_update(e->E, sig);
    e->Var = (e->E2->val) - (e->E->val * e->E->val);
    double alpha = sqrt(2.0) / e->Mean;
    double tebno = 10.0 * (log10(2.0) - log10(((e->Var * alpha * alpha) - 0.0085))) - 5.0;
    if (isnan(tebno)) tebno = 50;
    if (tebno > 50.0) tebno = 50;
    e->EbNo = e->EbNo * 0.8 + 0.2 * tebno;
    return e->EbNo;
}

/* ------------------------------------------------------------------ DiffDecode::UpdateSoft (DSP.cpp:531-563) */
static double diffdecode_update_soft(double *lastsoftstate, double soft)
{
    double retval = 0;
    if (soft < 0 && *lastsoftstate < 0) { retval = *lastsoftstate; *lastsoftstate = soft; }
    else if (soft > 0 && *lastsoftstate > 0) { retval = -*lastsoftstate; *lastsoftstate = soft; }
    else { retval = fabs(*lastsoftstate); *lastsoftstate = soft; }
    return retval;
}

/* ------------------------------------------------------------------ Delay<double> (DSP.h:341-379) */
typedef struct { double buff[64]; int size, buffptr; double fractdelay; } delay_t;
static void delay_set(delay_t *d, double fractdelay)
{
    d->fractdelay = fractdelay;
    d->size = (int)ceil(fractdelay) + 1;
    memset(d->buff, 0, sizeof(d->buff));
    d->buffptr = 0;
}
static double delay_update(delay_t *d, double sig)
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

/* ------------------------------------------------------------------ IIR (DSP.cpp:634-709), a,b of size 3 */
typedef struct { double a[3], b[3], bx[3], by[2]; int px, py; double y; } iir_t;
static void iir_init(iir_t *f) { f->px = 0; f->py = 0; memset(f->bx, 0, sizeof(f->bx)); memset(f->by, 0, sizeof(f->by)); }
static double iir_update(iir_t *f, double sig)
{
    if (f->px >= 3) f->px = 0;
    if (f->py >= 2) f->py = 0;
    f->bx[f->px] = sig;
    f->px++; f->px %= 3;
    f->y = 0;
    for (int i = 2; i >= 0; i--) { f->y += f->bx[f->px] * f->b[i]; f->px++; f->px %= 3; }
    for (int i = 2; i >= 1; i--) { f->y -= f->by[f->py] * f->a[i]; f->py++; f->py %= 2; }
    f->y /= f->a[0];
    f->by[f->py] = f->y;
    f->py++; f->py %= 2;
    return f->y;
}

/* ------------------------------------------------------------------ DelayThing<cpx_type> (DSP.h:439-486) */
typedef struct { cpx *buffer; int ptr, sz; } delaything_t;
static void delaything_set_length(delaything_t *d, int length)
{
    length++;
    /* QVector::resize keeps old contents and zero-fills new elements */
    cpx *nb = (cpx *)calloc((size_t)length, sizeof(cpx));
    if (d->buffer) { int keep = d->sz < length ? d->sz : length; memcpy(nb, d->buffer, sizeof(cpx) * (size_t)keep); free(d->buffer); }
    d->buffer = nb; d->ptr = 0; d->sz = length;
}
static void delaything_update(delaything_t *d, cpx *data)
{
    d->buffer[d->ptr] = *data;
    d->ptr++; d->ptr %= d->sz;
    *data = d->buffer[d->ptr];
}
static cpx delaything_update_dont_touch(delaything_t *d, cpx data)
{
    d->buffer[d->ptr] = data;
    d->ptr++; d->ptr %= d->sz;
    return d->buffer[d->ptr];
}

/* ------------------------------------------------------------------ RootRaisedCosine::design (DSP.h:316-338) */
int jo_rrc_design(double alpha, int firsize, double samplerate, double symbol_freq, double *Points)
{
    if ((firsize % 2) == 0) firsize += 1;
    double T = (samplerate) / (symbol_freq);
    double fi;
    for (int i = 0; i < firsize; i++)
    {
        if (i == ((firsize - 1) / 2)) Points[i] = (4.0 * alpha + M_PI - M_PI * alpha) / (M_PI * sqrt(T));
        else
        {
            fi = (((double)i) - ((double)(firsize - 1)) / 2.0);
            if (fabs(1.0 - pow(4.0 * alpha * fi / T, 2)) < 0.0000000001)
                Points[i] = (alpha * ((M_PI - 2.0) * cos(M_PI / (4.0 * alpha)) + (M_PI + 2.0) * sin(M_PI / (4.0 * alpha))) / (M_PI * sqrt(2.0 * T)));
            else
                Points[i] = (4.0 * alpha / (M_PI * sqrt(T)) * (cos((1.0 + alpha) * M_PI * fi / T) + T / (4.0 * alpha * fi) * sin((1.0 - alpha) * M_PI * fi / T)) / (1.0 - pow(4.0 * alpha * fi / T, 2)));
        }
    }
    return firsize;
}

/* ------------------------------------------------------------------ FFT (same algorithm as oracle/ref/shim/jfft.h) */
typedef struct { int n; cpx *tw; int *rev; } fftplan;
static fftplan *fft_plan(int n)
{
    fftplan *p = (fftplan *)calloc(1, sizeof(fftplan));
    p->n = n;
    p->tw = (cpx *)calloc((size_t)(n / 2 > 0 ? n / 2 : 1), sizeof(cpx));
    for (int i = 0; i < n / 2; i++)
    {
        double a = -2.0 * M_PI * ((double)i) / ((double)n);
        p->tw[i].re = cos(a); p->tw[i].im = sin(a);
    }
    p->rev = (int *)calloc((size_t)n, sizeof(int));
    int bits = 0;
    while ((1 << bits) < n) bits++;
    for (int i = 0; i < n; i++)
    {
        int r = 0;
        for (int b = 0; b < bits; b++) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        p->rev[i] = r;
    }
    return p;
}
static void fft_free(fftplan *p) { if (!p) return; free(p->tw); free(p->rev); free(p); }
static void fft_run(const fftplan *p, cpx *x, int inverse)
{
    const int n = p->n;
    for (int i = 0; i < n; i++) { int r = p->rev[i]; if (r > i) { cpx t = x[i]; x[i] = x[r]; x[r] = t; } }
    for (int len = 2; len <= n; len <<= 1)
    {
        int half = len >> 1, step = n / len;
        for (int base = 0; base < n; base += len)
            for (int j = 0; j < half; j++)
            {
                cpx w = p->tw[j * step];
                if (inverse) w.im = -w.im;
                cpx a = x[base + j], b = x[base + j + half], t;
                t.re = b.re * w.re - b.im * w.im;
                t.im = b.re * w.im + b.im * w.re;
                x[base + j].re = a.re + t.re; x[base + j].im = a.im + t.im;
                x[base + j + half].re = a.re - t.re; x[base + j + half].im = a.im - t.im;
            }
    }
}
/* FFTWrapper<double>::transform with kissfft_scaling=true (fftwrapper.cpp:19-35): forward unnormalised; inverse =
 * JFFT ifft (1/N) then multiplied by N again */
static void fftwrapper_transform(const fftplan *p, const cpx *in, cpx *out, int inverse)
{
    for (int i = 0; i < p->n; i++) out[i] = in[i];
    if (inverse)
    {
        fft_run(p, out, 1);
        double s = 1.0 / ((double)p->n);
        for (int i = 0; i < p->n; i++) { out[i].re *= s; out[i].im *= s; }
        for (int i = 0; i < p->n; i++) { out[i].re *= (double)p->n; out[i].im *= (double)p->n; }
    }
    else fft_run(p, out, 0);
}
void jo_fft(double *re_im, int n, int inverse)
{
    fftplan *p = fft_plan(n);
    cpx *tmp = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    fftwrapper_transform(p, (const cpx *)re_im, tmp, inverse);
    memcpy(re_im, tmp, sizeof(cpx) * (size_t)n);
    free(tmp);
    fft_free(p);
}

/* ------------------------------------------------------------------ CoarseFreqEstimate (coarsefreqestimate.cpp) */
struct jo_coarse
{
    fftplan *plan;
    cpx *out, *in;
    double *y, *z, *window;
    double nfft, Fs, hzperbin, lockingbw, fb, freq_offset_est;
    int startbin, stopbin, expectedpeakbin, emptyingcountdown;
    int ynalloc;
};
static void coarse_set_settings(jo_coarse *c, int power, double lockingbw, double fb, double Fs) /* :39-76 */
{
    c->lockingbw = lockingbw; c->fb = fb; c->Fs = Fs;
    if (c->plan) fft_free(c->plan);
    c->nfft = pow(2, power);
    int n = (int)c->nfft;
    c->plan = fft_plan(n);
    c->hzperbin = Fs / ((double)c->nfft);
    c->out = (cpx *)realloc(c->out, sizeof(cpx) * (size_t)n);
    c->in = (cpx *)realloc(c->in, sizeof(cpx) * (size_t)n);
    /* y.resize(nfft): keeps old values, new elements are zero */
    double *ny = (double *)calloc((size_t)n, sizeof(double));
    if (c->y) { int keep = c->ynalloc < n ? c->ynalloc : n; memcpy(ny, c->y, sizeof(double) * (size_t)keep); free(c->y); }
    c->y = ny; c->ynalloc = n;
    c->z = (double *)realloc(c->z, sizeof(double) * (size_t)n);
    c->startbin = (int)fmax(round(lockingbw / c->hzperbin), 1.0);
    c->stopbin = (int)(c->nfft - c->startbin);
    c->expectedpeakbin = (int)round(fb / (2.0 * c->hzperbin));
    /* raised-cos window favouring the middle (:61-74), used by the fb==8400 branch of ProcessBasebandData */
    c->window = (double *)realloc(c->window, sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) c->window[i] = 0;
    c->window[0] = 1;
    for (int i = 1; i <= c->startbin; i++)
    {
        double val = cos(M_PI_2 * ((double)i) / ((double)c->startbin));
        val *= val;
        if ((n - i) < 0) break;
        if (i >= n) break;
        c->window[n - i] = val;
        c->window[i] = val;
    }
}
jo_coarse *jo_coarse_create(int power, double lockingbw, double fb, double Fs)
{
    jo_coarse *c = (jo_coarse *)calloc(1, sizeof(jo_coarse));
    /* ctor :5-37 (power 13 defaults; only emptyingcountdown=1 and the zeroed y survive setSettings) */
    c->emptyingcountdown = 1;
    coarse_set_settings(c, power, lockingbw, fb, Fs);
    return c;
}
void jo_coarse_destroy(jo_coarse *c)
{
    if (!c) return;
    fft_free(c->plan); free(c->out); free(c->in); free(c->y); free(c->z); free(c->window); free(c);
}
void jo_coarse_bigchange(jo_coarse *c) /* :84-88 */
{
    c->emptyingcountdown = 4;
    for (int i = 0; i < (int)c->nfft; i++) c->y[i] = 20;
}
void jo_coarse_get_y(jo_coarse *c, double *y) { memcpy(y, c->y, sizeof(double) * (size_t)c->nfft); }
/* ProcessBasebandData :90-137; returns the value passed to emit FreqOffsetEstimate */
static double coarse_process(jo_coarse *c, const cpx *data)
{
    int n = (int)c->nfft;
    fftwrapper_transform(c->plan, data, c->out, 0);
    if (c->fb != 8400) { for (int i = c->startbin; i <= c->stopbin; i++) { c->out[i].re = 0; c->out[i].im = 0; } } /* :99 */
    else { for (int i = 0; i < n; i++) c->out[i] = cscale(c->out[i], c->window[i]); }                               /* :100 */
    fftwrapper_transform(c->plan, c->out, c->in, 1);
    for (int i = 0; i < n; i++) c->in[i] = cmul(c->in[i], c->in[i]);
    fftwrapper_transform(c->plan, c->in, c->out, 0);
    for (int i = 0; i < n / 2; i++) { cpx t = c->out[i + n / 2]; c->out[i + n / 2] = c->out[i]; c->out[i] = t; }
    for (int i = 0; i < n; i++) c->y[i] = c->y[i] * 0.9 + 0.1 * 10 * log10(fmax(hypot(c->out[i].re, c->out[i].im), 1));
    double zmax = 0;
    int zmaxloc = (int)(c->nfft / 2);
    int i0 = (int)round((-c->lockingbw / c->hzperbin) + ((double)(c->nfft / 2)));
    int i1 = (int)round((c->lockingbw / c->hzperbin) + ((double)(c->nfft / 2)));
    for (int i = i0; i < i1; i++)
    {
        if ((i < 0) || (i >= n)) continue;
        double val = 0;
        for (int j = -1; j <= 1; j++)
        {
            if (((i - c->expectedpeakbin - j) < 0) || ((i + c->expectedpeakbin + j) >= n)) continue;
            val += (c->y[i - c->expectedpeakbin - j] + c->y[i + c->expectedpeakbin + j]);
        }
        c->z[i] = val;
        if (c->z[i] > zmax) { zmax = c->z[i]; zmaxloc = i; }
    }
    c->freq_offset_est = -((double)(zmaxloc - (int)(c->nfft / 2))) * c->hzperbin * 0.5;
    if (c->emptyingcountdown <= 0) return c->freq_offset_est;
    c->emptyingcountdown--;
    return 0;
}
double jo_coarse_process(jo_coarse *c, const double *re_im) { return coarse_process(c, (const cpx *)re_im); }

/* ------------------------------------------------------------------ growable capture buffers */
/* ------------------------------------------------------------------ JFastFir (JFFT; restated as oracle/ref/shim/jfft.h builds it) */
typedef struct
{
    fftplan *plan;
    int nfft, K, L;
    cpx *H, *inbuf, *tail, *blk;
    int nin;
    cpx *outq; long outq_len, outq_cap, outq_rd;
} fastfir_t;
static void fastfir_set_kernel(fastfir_t *f, const cpx *k, int K, int nfft)
{
    if (f->plan) fft_free(f->plan);
    free(f->H); free(f->inbuf); free(f->tail); free(f->blk); free(f->outq);
    f->K = K; f->nfft = nfft; f->L = nfft - K + 1;
    f->plan = fft_plan(nfft);
    f->H = (cpx *)calloc((size_t)nfft, sizeof(cpx));
    for (int i = 0; i < K; i++) f->H[i] = k[i];
    fft_run(f->plan, f->H, 0);
    f->inbuf = (cpx *)calloc((size_t)f->L, sizeof(cpx)); f->nin = 0;
    f->tail = (cpx *)calloc((size_t)nfft, sizeof(cpx));
    f->blk = (cpx *)calloc((size_t)nfft, sizeof(cpx));
    f->outq_cap = 4 * (long)nfft; f->outq = (cpx *)calloc((size_t)f->outq_cap, sizeof(cpx));
    f->outq_len = f->L; f->outq_rd = 0; /* L samples of latency */
}
static void fastfir_processblock(fastfir_t *f)
{
    const int nfft = f->nfft, L = f->L;
    for (int i = 0; i < nfft; i++) { f->blk[i].re = 0; f->blk[i].im = 0; }
    for (int i = 0; i < L; i++) f->blk[i] = f->inbuf[i];
    f->nin = 0;
    fft_run(f->plan, f->blk, 0);
    for (int i = 0; i < nfft; i++) f->blk[i] = cmul(f->blk[i], f->H[i]);
    fft_run(f->plan, f->blk, 1);
    double s = 1.0 / ((double)nfft);
    for (int i = 0; i < nfft; i++) { f->blk[i].re *= s; f->blk[i].im *= s; }
    for (int i = 0; i < nfft; i++) { f->tail[i].re += f->blk[i].re; f->tail[i].im += f->blk[i].im; }
    if (f->outq_len + L > f->outq_cap)
    {
        f->outq_cap = 2 * (f->outq_len + L);
        f->outq = (cpx *)realloc(f->outq, sizeof(cpx) * (size_t)f->outq_cap);
    }
    for (int i = 0; i < L; i++) f->outq[f->outq_len++] = f->tail[i];
    for (int i = 0; i + L < nfft; i++) f->tail[i] = f->tail[i + L];
    for (int i = nfft - L; i < nfft; i++) { f->tail[i].re = 0; f->tail[i].im = 0; }
}
static void fastfir_update(fastfir_t *f, cpx *data, long n)
{
    for (long i = 0; i < n; i++)
    {
        f->inbuf[f->nin++] = data[i];
        if (f->nin == f->L) fastfir_processblock(f);
        data[i] = f->outq[f->outq_rd++];
    }
    if (f->outq_rd > 0)
    {
        memmove(f->outq, f->outq + f->outq_rd, sizeof(cpx) * (size_t)(f->outq_len - f->outq_rd));
        f->outq_len -= f->outq_rd; f->outq_rd = 0;
    }
}
static void fastfir_free(fastfir_t *f)
{
    if (f->plan) fft_free(f->plan);
    free(f->H); free(f->inbuf); free(f->tail); free(f->blk); free(f->outq);
    memset(f, 0, sizeof(*f));
}
static void fastfir_set_kernel_real(fastfir_t *f, const double *pts, int K, int nfft) /* SetKernel(QVector<double>[, nfft]) */
{
    cpx *k = (cpx *)malloc(sizeof(cpx) * (size_t)K);
    for (int i = 0; i < K; i++) { k[i].re = pts[i]; k[i].im = 0.0; }
    if (nfft <= 0) { int pw = 1; while (pw < K) pw <<= 1; nfft = 4 * pw; } /* the 1-argument SetKernel */
    fastfir_set_kernel(f, k, K, nfft);
    free(k);
}

typedef struct { char *p; size_t len, cap; } gbuf;
static void gpush(gbuf *g, const void *src, size_t n)
{
    if (g->len + n > g->cap)
    {
        size_t nc = g->cap ? g->cap * 2 : 4096;
        while (nc < g->len + n) nc *= 2;
        g->p = (char *)realloc(g->p, nc); g->cap = nc;
    }
    memcpy(g->p + g->len, src, n); g->len += n;
}
static long gtake(gbuf *g, void *dst, size_t elsz, long capels)
{
    long have = (long)(g->len / elsz);
    long n = have < capels ? have : capels;
    memcpy(dst, g->p, (size_t)n * elsz);
    memmove(g->p, g->p + (size_t)n * elsz, g->len - (size_t)n * elsz);
    g->len -= (size_t)n * elsz;
    return n;
}

/* ------------------------------------------------------------------ the demodulator object */
struct jo_demod
{
    int kind;
    int afc, sql, cpuReduce, dcd;
    double Fs, freq_center, lockingbw, fb, signalthreshold, SamplesPerSymbol;
    int bbnfft, bbcycbuff_ptr, coarseCounter;
    cpx *bbcycbuff, *bbtmpbuff;
    fir_t *fir_re, *fir_im;
    delay_t delays, delayt41, delayt42, delayt8;
    iir_t st_iir_resonator, ct_iir_loopfilter;
    wavetable st_osc, st_osc_ref, mixer_center, mixer2, mixer_fir_pre;
    fastfir_t fir_pre;          /* fb==8400 prefilter */
    cpx *prefilt; long prefilt_cap;
    jo_coarse *coarse;
    double mse;
    msecalc_t msecalc;      /* OQPSK */
    ma_t *msema;            /* MSK */
    agc_t *agc;
    ebno_t ebno;
    ma_t *marg;
    delaything_t dt, delayedsmpl;
    double ee, correctionfactor;
    double diff_lastsoftstate;
    /* former function statics */
    cpx sig2_last; int sig2_last_init;
    int yui; cpx pt_d;
    int countdown, countdown2;
    /* RxDataBits */
    short rx[64]; int nrx;
    /* captures */
    gbuf soft, status, symbols;
    int capture_symbols;
    double nest;
};

static void set_resonator(iir_t *f, double b0, double b1, double b2, double a0, double a1, double a2)
{
    f->b[0] = b0; f->b[1] = b1; f->b[2] = b2; f->a[0] = a0; f->a[1] = a1; f->a[2] = a2;
}

static void emit_soft(jo_demod *d) { gpush(&d->soft, d->rx, sizeof(short) * (size_t)d->nrx); d->nrx = 0; }

static void bb_resize(jo_demod *d, int n)
{
    /* QVector::resize keeps contents / zero-fills */
    cpx *nb = (cpx *)calloc((size_t)n, sizeof(cpx));
    if (d->bbcycbuff) { int keep = d->bbnfft < n ? d->bbnfft : n; memcpy(nb, d->bbcycbuff, sizeof(cpx) * (size_t)keep); free(d->bbcycbuff); }
    d->bbcycbuff = nb;
    d->bbtmpbuff = (cpx *)realloc(d->bbtmpbuff, sizeof(cpx) * (size_t)n);
    d->bbnfft = n;
}

/* ---------------- OQPSK ---------------- */
static void oqpsk_ctor(jo_demod *d) /* oqpskdemodulator.cpp:8-117 */
{
    d->afc = 0; d->sql = 0; d->dcd = 0; d->cpuReduce = 0;
    d->mse = 100;
    d->Fs = 48000; d->lockingbw = 10500; d->freq_center = 8000; d->fb = 10500; d->signalthreshold = 0.5;
    d->SamplesPerSymbol = 2.0 * d->Fs / d->fb;
    wt_init(&d->mixer_center); wt_init(&d->mixer2); wt_init(&d->st_osc); wt_init(&d->st_osc_ref);
    wt_setfreq_sr(&d->mixer_center, d->freq_center, (int)d->Fs);
    wt_setfreq_sr(&d->mixer2, d->freq_center, (int)d->Fs);
    bb_resize(d, (int)pow(2, 14));
    d->bbcycbuff_ptr = 0;
    d->agc = agc_new(4, d->Fs);
    d->ebno.E = ma_new((int)(2 * d->Fs)); d->ebno.E2 = ma_new((int)(2 * d->Fs)); d->ebno.Fs = d->Fs; d->ebno.fb = d->fb;
    d->ebno.EbNo = 0; /* uninitialised in the reference (DSP.cpp:715-721); defined as 0 here */
    d->marg = ma_new(800);
    delaything_set_length(&d->dt, 400);
    d->msecalc.pointmean = ma_new(400); d->msecalc.msema = ma_new(400); d->msecalc.mse = 0;
    d->coarse = jo_coarse_create(14, d->lockingbw, d->fb, d->Fs);
    double pts[64];
    int np = jo_rrc_design(1, 55, d->Fs, 10500 / 2, pts);
    d->fir_re = fir_new(np); d->fir_im = fir_new(np);
    for (int i = 0; i < np; i++) { d->fir_re->points[i] = pts[i]; d->fir_im->points[i] = pts[i]; }
    double T = d->Fs / 5250.0;
    delay_set(&d->delays, 1); delay_set(&d->delayt41, T / 4.0); delay_set(&d->delayt42, T / 4.0); delay_set(&d->delayt8, T / 8.0);
    set_resonator(&d->st_iir_resonator, 0.00032714218939589035, 0, 0.00032714218939589035, 1, -0.39005299948210803, 0.99934571562120822);
    iir_init(&d->st_iir_resonator);
    wt_setfreq_sr(&d->st_osc, 10500, (int)d->Fs);
    wt_setfreq_sr(&d->st_osc_ref, 10500, (int)d->Fs);
    set_resonator(&d->ct_iir_loopfilter, 0.0010275610653672064, 0.0020551221307344128, 0.0010275610653672064, 1, -1.9207386815577139, 0.92509247310306331);
    iir_init(&d->ct_iir_loopfilter);
    d->countdown = 4; d->countdown2 = 5; d->yui = 0; d->pt_d.re = 0; d->pt_d.im = 0; d->sig2_last_init = 0;
    /* just for 8400 (:110-115) */
    {
        double *pp = (double *)malloc(sizeof(double) * 1100);
        int npp = jo_rrc_design(1, 1024, d->Fs, 10500 / 2, pp);
        fastfir_set_kernel_real(&d->fir_pre, pp, npp, 0);
        free(pp);
        wt_init(&d->mixer_fir_pre);
        wt_setfreq_sr(&d->mixer_fir_pre, d->freq_center, (int)d->Fs);
    }
}

static void oqpsk_set_settings(jo_demod *d, const jo_settings *s) /* oqpskdemodulator.cpp:175-289 */
{
    d->Fs = s->Fs;
    d->lockingbw = s->lockingbw;
    d->fb = s->fb;
    d->freq_center = s->freq_center;
    if (d->freq_center > ((d->Fs / 2.0) - (d->lockingbw / 2.0))) d->freq_center = ((d->Fs / 2.0) - (d->lockingbw / 2.0));
    d->signalthreshold = s->signalthreshold;
    d->SamplesPerSymbol = 2.0 * d->Fs / d->fb;
    bb_resize(d, (int)pow(2, s->coarsefreqest_fft_power));
    d->bbcycbuff_ptr = 0;
    coarse_set_settings(d->coarse, s->coarsefreqest_fft_power, 2.0 * d->lockingbw / 2.0, d->fb, d->Fs);
    wt_setfreq_sr(&d->mixer_center, d->freq_center, (int)d->Fs);
    wt_setfreq_sr(&d->mixer2, d->freq_center, (int)d->Fs);
    agc_free(d->agc);
    d->agc = agc_new(4, d->Fs);
    fir_free(d->fir_re); fir_free(d->fir_im);
    double pts[64];
    int np = jo_rrc_design(d->fb == 8400 ? 0.6 : 1.0, 55, d->Fs, d->fb / 2, pts); /* :209-211 */
    d->fir_re = fir_new(np); d->fir_im = fir_new(np);
    for (int i = 0; i < np; i++) { d->fir_re->points[i] = pts[i]; d->fir_im->points[i] = pts[i]; }
    double T = d->Fs / (d->fb / 2);
    delay_set(&d->delays, 1); delay_set(&d->delayt41, T / 4.0); delay_set(&d->delayt42, T / 4.0); delay_set(&d->delayt8, T / 8.0);
    if (d->fb == 8400)
    {
        /* the 10 Hz resonator that overwrites the 35 Hz one (:231-251) */
        set_resonator(&d->st_iir_resonator, 0.0012845857864470789, 0, -0.0012845857864470789, 1, -0.90681461999279889, 0.99743082842710584);
        d->ee = 0.65;
    }
    else
    {
        set_resonator(&d->st_iir_resonator, 0.00032714218939589035, 0, 0.00032714218939589035, 1, -0.39005299948210803, 0.99934571562120822);
        d->ee = 0.4;
    }
    iir_init(&d->st_iir_resonator);
    wt_setfreq_sr(&d->st_osc, d->fb, (int)d->Fs);
    wt_setfreq_sr(&d->st_osc_ref, d->fb, (int)d->Fs);
    d->ebno.Fs = d->Fs; d->ebno.fb = d->fb;
    /* prefilter kernel (:278-283); built for every rate, used at 8400 */
    {
        double *pp = (double *)malloc(sizeof(double) * 2100);
        int npp = jo_rrc_design(d->fb == 8400 ? 0.6 : 1.0, 2048, d->Fs, d->fb / 2, pp);
        fastfir_set_kernel_real(&d->fir_pre, pp, npp, 4096);
        free(pp);
    }
    d->coarseCounter = 0;
}

static void record_status(jo_demod *d)
{
    double row[6];
    row[0] = d->nest; row[1] = d->mixer2.freq; row[2] = d->mixer_center.freq; row[3] = d->mse; row[4] = d->ebno.EbNo;
    row[5] = (d->mse > d->signalthreshold) ? 0.0 : 1.0;
    gpush(&d->status, row, sizeof(row));
    d->nest += 1.0;
}

static void oqpsk_freq_offset_estimate_slot(jo_demod *d, double freq_offset_est) /* oqpskdemodulator.cpp:629-677 */
{
    /* update prefilter until we have dcd and a signal (:634-638) */
    if ((d->mse > d->signalthreshold) || (!d->dcd)) wt_setfreq_sr(&d->mixer_fir_pre, d->mixer_center.freq + freq_offset_est, (int)d->Fs);
    if ((d->mse < d->signalthreshold) && (!d->dcd))
    {
        if (d->countdown2 > 0) d->countdown2--;
        else wt_setfreq(&d->mixer2, d->mixer_center.freq + freq_offset_est);
    }
    else d->countdown2 = 5;

    if ((d->mse > d->signalthreshold) && (fabs(d->mixer2.freq - (d->mixer_center.freq + freq_offset_est)) > 3.0))
        wt_setfreq(&d->mixer2, d->mixer_center.freq + freq_offset_est);
    if ((d->afc) && (d->mse < d->signalthreshold) && (fabs(d->mixer2.freq - d->mixer_center.freq) > 3.0))
    {
        if (d->countdown > 0) d->countdown--;
        else
        {
            wt_setfreq(&d->mixer_center, d->mixer2.freq);
            if (d->mixer_center.freq < d->lockingbw / 2.0) wt_setfreq(&d->mixer_center, d->lockingbw / 2.0);
            if (d->mixer_center.freq > (d->Fs / 2.0 - d->lockingbw / 2.0)) wt_setfreq(&d->mixer_center, d->Fs / 2.0 - d->lockingbw / 2.0);
            jo_coarse_bigchange(d->coarse);
            for (int j = 0; j < d->bbnfft; j++) { d->bbcycbuff[j].re = 0; d->bbcycbuff[j].im = 0; }
        }
    }
    else d->countdown = 4;
    record_status(d);
}

void jo__msk_slot(jo_demod *d, double freq_offset_est);
static void coarse_ring_step(jo_demod *d, double dval, int is_oqpsk)
{
    /* oqpskdemodulator.cpp:410-431 == mskdemodulator.cpp:350-368 */
    if ((d->coarseCounter >= d->Fs || !d->cpuReduce))
    {
        d->bbcycbuff[d->bbcycbuff_ptr] = cscale(wt_cis(&d->mixer_center), dval);
        d->bbcycbuff_ptr++; d->bbcycbuff_ptr %= d->bbnfft;
        if (d->bbcycbuff_ptr % (d->cpuReduce ? d->bbnfft : d->bbnfft / 4) == 0)
        {
            for (int j = 0; j < d->bbnfft; j++)
            {
                d->bbtmpbuff[j] = d->bbcycbuff[d->bbcycbuff_ptr];
                d->bbcycbuff_ptr++; d->bbcycbuff_ptr %= d->bbnfft;
            }
            double est = coarse_process(d->coarse, d->bbtmpbuff);
            if (is_oqpsk) oqpsk_freq_offset_estimate_slot(d, est);
            else jo__msk_slot(d, est);
            d->coarseCounter = 0;
        }
    }
    d->coarseCounter++;
}

static void oqpsk_write(jo_demod *d, const int16_t *ptr, long n) /* oqpskdemodulator.cpp:334-627 */
{
    if (!n) return;
    /* prefilter (:343-381): down with mixer_fir_pre, JFastFir over the whole write, up again from the saved phase */
    if (d->fb == 8400)
    {
        if (d->prefilt_cap < n) { d->prefilt = (cpx *)realloc(d->prefilt, sizeof(cpx) * (size_t)n); d->prefilt_cap = n; }
        const double savedphase = wt_get_phase_deg(&d->mixer_fir_pre);
        for (long i = 0; i < n; i++)
        {
            double dval = ((double)ptr[i]) / 32768.0;
            d->prefilt[i] = cscale(wt_cis(&d->mixer_fir_pre), dval);
            wt_next(&d->mixer_fir_pre);
        }
        fastfir_update(&d->fir_pre, d->prefilt, n);
        wt_set_phase_deg(&d->mixer_fir_pre, savedphase);
        for (long i = 0; i < n; i++)
        {
            cpx cj = wt_cis_conj(&d->mixer_fir_pre);
            d->prefilt[i] = cmul(d->prefilt[i], cj);
            wt_next(&d->mixer_fir_pre);
        }
    }
    double mixer2_freq_sum = 0;
    for (long i = 0; i < n; i++)
    {
        double dval = ((double)ptr[i]) / 32768.0;
        coarse_ring_step(d, dval, 1);

        cpx sig2;
        if (d->fb == 8400)
        {
            sig2 = cmul(wt_cis(&d->mixer2), d->prefilt[i]); /* :440 */
            mixer2_freq_sum += d->mixer2.freq;              /* :447 */
        }
        else
        {
            cpx cval = cscale(wt_cis(&d->mixer2), dval);
            sig2.re = fir_update_and_process(d->fir_re, cval.re);
            sig2.im = fir_update_and_process(d->fir_im, cval.im);
        }

        double dabval = sqrt(sig2.re * sig2.re + sig2.im * sig2.im);
        oqpsk_ebno_update(&d->ebno, dabval);
        sig2 = cscale(sig2, agc_update(d->agc, dabval));

        double abval = hypot(sig2.re, sig2.im);
        if (abval > 2.84) sig2 = cscale(sig2, (2.84 / abval));

        /* symbol timer :473-484 */
        double st_diff = delay_update(&d->delays, abval * abval) - (abval * abval);
        double st_d1out = delay_update(&d->delayt41, st_diff);
        double st_d2out = delay_update(&d->delayt42, st_d1out);
        double st_eta = (st_d2out - st_diff) * st_d1out;
        st_eta = iir_update(&d->st_iir_resonator, st_eta);
        cpx st_m1; st_m1.re = st_eta; st_m1.im = -delay_update(&d->delayt8, st_eta);
        cpx st_out = cmul(wt_cis(&d->st_osc), st_m1);
        double st_angle_error = atan2(st_out.im, st_out.re);
        wt_increase_freq(&d->st_osc, -st_angle_error * 0.00000001);
        wt_advance_fraction_of_wave(&d->st_osc, -st_angle_error * 0.01 / 360.0);
        if (d->st_osc.freq < (d->st_osc_ref.freq - 0.1)) wt_setfreq(&d->st_osc, (d->st_osc_ref.freq - 0.1));
        if (d->st_osc.freq > (d->st_osc_ref.freq + 0.1)) wt_setfreq(&d->st_osc, (d->st_osc_ref.freq + 0.1));

        /* sample times :487-595 */
        if (!d->sig2_last_init) { d->sig2_last = sig2; d->sig2_last_init = 1; }
        if (wt_if_have_passed_point(&d->st_osc, d->ee))
        {
            double pt_last = d->st_osc.FractionOfSampleItPassesBy;
            double pt_this = 1.0 - pt_last;
            cpx pt;
            pt.re = pt_this * sig2.re + pt_last * d->sig2_last.re;
            pt.im = pt_this * sig2.im + pt_last * d->sig2_last.im;
            d->yui++; d->yui %= 2;
            if (!d->yui) d->pt_d = pt;
            else
            {
                cpx pt_qpsk; pt_qpsk.re = pt.re; pt_qpsk.im = d->pt_d.im;
                double ct_xt = tanh(pt.im) * pt.re;
                double ct_xt_d = tanh(d->pt_d.re) * d->pt_d.im;
                double ct_ec = ct_xt_d - ct_xt;
                if (ct_ec > M_PI) ct_ec = M_PI;
                if (ct_ec < -M_PI) ct_ec = -M_PI;
                if (d->fb > 8400) /* :518-525 */
                {
                    ct_ec = iir_update(&d->ct_iir_loopfilter, ct_ec);
                    if (ct_ec > M_PI_2) ct_ec = M_PI_2;
                    if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
                    wt_increase_phase_deg(&d->mixer2, 1.0 * ct_ec);
                    wt_increase_freq(&d->mixer2, 0.01 * ct_ec);
                }
                else /* 8400 works better with faster phase agility (:526-532) */
                {
                    wt_increase_phase_deg(&d->mixer2, 1.0 * ct_ec);
                    wt_increase_freq(&d->mixer2, 0.5 * 0.01 * iir_update(&d->ct_iir_loopfilter, ct_ec));
                }

                ma_update_signed(d->marg, ct_ec);
                delaything_update(&d->dt, &pt_qpsk);
                cpx rot; rot.re = cos(d->marg->val); rot.im = sin(d->marg->val);
                pt_qpsk = cmul(pt_qpsk, rot);

                d->mse = msecalc_update(&d->msecalc, pt_qpsk);
                if (d->capture_symbols) { double row[3] = {pt_qpsk.re, pt_qpsk.im, d->mse}; gpush(&d->symbols, row, sizeof(row)); }

                if (d->mse < d->signalthreshold)
                {
                    int ibit = qRound_(0.75 * pt_qpsk.im * 127.0 + 128.0);
                    if (ibit > 255) ibit = 255;
                    if (ibit < 0) ibit = 0;
                    d->rx[d->nrx++] = (short)(unsigned char)ibit;
                    ibit = qRound_(0.75 * pt_qpsk.re * 127.0 + 128.0);
                    if (ibit > 255) ibit = 255;
                    if (ibit < 0) ibit = 0;
                    d->rx[d->nrx++] = (short)(unsigned char)ibit;
                    if (d->nrx >= 32) emit_soft(d); /* the sql test at :585 is always true inside mse<threshold */
                }
            }
        }
        d->sig2_last = sig2;

        wt_next(&d->mixer2);
        wt_next(&d->mixer_center);
        wt_next(&d->st_osc);
        wt_next(&d->st_osc_ref);
    }
    /* update the 8400bps pre filter with better estimates of carrier (:607-608; runs at every rate) */
    wt_setfreq(&d->mixer_fir_pre, mixer2_freq_sum / ((double)n));
}

static void center_freq_changed(jo_demod *d, double freq_center) /* oqpsk :291-310, msk :265-282 */
{
    if (d->kind == JO_KIND_OQPSK)
    {
        if (d->fb != 8400) /* :293 */
        {
            if (freq_center < (0.5 * d->fb)) freq_center = 0.5 * d->fb;
            if (freq_center > (d->Fs / 2.0 - 0.5 * d->fb)) freq_center = d->Fs / 2.0 - 0.5 * d->fb;
        }
    }
    else
    {
        if (freq_center < (0.75 * d->fb)) freq_center = 0.75 * d->fb;
        if (freq_center > (d->Fs / 2.0 - 0.75 * d->fb)) freq_center = d->Fs / 2.0 - 0.75 * d->fb;
    }
    wt_setfreq_sr(&d->mixer_center, freq_center, (int)d->Fs);
    if (d->afc) wt_setfreq(&d->mixer2, d->mixer_center.freq);
    if ((d->mixer2.freq - d->mixer_center.freq) > (d->lockingbw / 2.0)) wt_setfreq(&d->mixer2, d->mixer_center.freq + (d->lockingbw / 2.0));
    if ((d->mixer2.freq - d->mixer_center.freq) < (-d->lockingbw / 2.0)) wt_setfreq(&d->mixer2, d->mixer_center.freq - (d->lockingbw / 2.0));
    for (int j = 0; j < d->bbnfft; j++) { d->bbcycbuff[j].re = 0; d->bbcycbuff[j].im = 0; }
}

/* ---------------- MSK ---------------- */
static void msk_make_filters(jo_demod *d)
{
    int ntaps = (int)(2 * d->SamplesPerSymbol); /* FIR(int) from double */
    fir_free(d->fir_re); fir_free(d->fir_im);
    d->fir_re = fir_new(ntaps); d->fir_im = fir_new(ntaps);
    for (int i = 0; i < 2 * d->SamplesPerSymbol; i++)
    {
        double v = sin(M_PI * i / (2.0 * d->SamplesPerSymbol)) / (2.0 * d->SamplesPerSymbol);
        if (i >= 0 && i < ntaps) { d->fir_re->points[i] = v; d->fir_im->points[i] = v; } /* FIRSetPoint bounds check */
    }
}
static void msk_ctor(jo_demod *d) /* mskdemodulator.cpp:9-84 */
{
    d->afc = 0; d->sql = 0; d->cpuReduce = 0;
    d->Fs = 48000; d->lockingbw = 900; d->fb = 600; d->signalthreshold = 0.5;
    d->SamplesPerSymbol = d->Fs / d->fb;
    msk_make_filters(d);
    d->agc = agc_new(1, d->Fs);
    d->ebno.E = ma_new((int)(2.0 * d->Fs)); d->ebno.E2 = ma_new((int)(2.0 * d->Fs)); d->ebno.EbNo = 0;
    wt_init(&d->mixer_center); wt_init(&d->mixer2); wt_init(&d->st_osc); wt_init(&d->st_osc_ref);
    wt_setfreq_sr(&d->mixer_center, 1000, (int)d->Fs);
    wt_setfreq_sr(&d->mixer2, 1000, (int)d->Fs);
    wt_setfreq_sr(&d->st_osc, d->fb / 2, (int)d->Fs);
    bb_resize(d, (int)pow(2, 14));
    d->bbcycbuff_ptr = 0;
    d->mse = 10.0;
    d->msema = ma_new(600);
    d->marg = ma_new(80);
    delaything_set_length(&d->dt, 40);
    delaything_set_length(&d->delayedsmpl, 12); /* DelayThing default ctor */
    delay_set(&d->delayt8, 1);                  /* Delay default ctor */
    /* IIR default ctor :634-645 */
    set_resonator(&d->st_iir_resonator, 0.00032714218939589035, 0, 0.00032714218939589035, 1, -0.39005299948210803, 0.99934571562120822);
    iir_init(&d->st_iir_resonator);
    d->coarse = jo_coarse_create(14, d->lockingbw, d->fb, d->Fs);
    d->dcd = 0;
    d->correctionfactor = 1.0;
    d->diff_lastsoftstate = -1;
    d->countdown = 4;
}
static void msk_set_settings(jo_demod *d, const jo_settings *s) /* mskdemodulator.cpp:135-263 */
{
    d->Fs = s->Fs;
    d->lockingbw = s->lockingbw;
    d->fb = s->fb;
    d->freq_center = s->freq_center;
    if (d->freq_center > ((d->Fs / 2.0) - (d->lockingbw / 2.0))) d->freq_center = ((d->Fs / 2.0) - (d->lockingbw / 2.0));
    d->signalthreshold = s->signalthreshold;
    d->SamplesPerSymbol = (int)(d->Fs / d->fb);
    bb_resize(d, (int)pow(2, s->coarsefreqest_fft_power));
    d->bbcycbuff_ptr = 0;
    coarse_set_settings(d->coarse, s->coarsefreqest_fft_power, d->lockingbw, d->fb, d->Fs);
    wt_setfreq_sr(&d->mixer_center, d->freq_center, (int)d->Fs);
    wt_setfreq_sr(&d->mixer2, d->freq_center, (int)d->Fs);
    wt_setfreq_sr(&d->st_osc, d->fb / 2, (int)d->Fs);
    msk_make_filters(d);
    agc_free(d->agc);
    d->agc = agc_new(1, d->Fs);
    ma_free(d->ebno.E); ma_free(d->ebno.E2);
    d->ebno.E = ma_new((int)(2.0 * d->Fs)); d->ebno.E2 = ma_new((int)(2.0 * d->Fs)); d->ebno.EbNo = 0; /* uninitialised in ref */
    d->mse = 10.0;
    if (d->fb >= 1200)
    {
        d->correctionfactor = 0.6;
        if (d->Fs == 48000) { set_resonator(&d->st_iir_resonator, 2.617308727964618e-04, 0, -2.617308727964618e-04, 1, -1.993312819378528, 0.999476538254407); d->ee = 0.025; }
        else { set_resonator(&d->st_iir_resonator, 5.233248111921052e-04, 0, -5.233248111921052e-04, 1, -1.974342917561558, 0.998953350377616); d->ee = 0.05; }
    }
    else
    {
        d->correctionfactor = 1.0;
        if (d->Fs == 48000) { set_resonator(&d->st_iir_resonator, 1.308825621597620e-04, 0, -1.308825621597620e-04, 1, -1.998196509168551, 0.999738234875681); d->ee = 0.025; }
        else { set_resonator(&d->st_iir_resonator, 5.233248111921052e-04, 0, -5.233248111921052e-04, 1, -1.974342917561558, 0.998953350377616); d->ee = 0.0125; }
    }
    iir_init(&d->st_iir_resonator);
    ma_free(d->marg);
    d->marg = ma_new((int)d->SamplesPerSymbol);
    delaything_set_length(&d->dt, (int)(d->SamplesPerSymbol / 2));
    delaything_set_length(&d->delayedsmpl, (int)d->SamplesPerSymbol);
    delay_set(&d->delayt8, (d->SamplesPerSymbol) / 2.0);
    d->coarseCounter = 0;
}

void jo__msk_slot(jo_demod *d, double freq_offset_est) /* mskdemodulator.cpp:490-519 */
{
    if ((d->mse > d->signalthreshold) && (fabs(d->mixer2.freq - (d->mixer_center.freq + freq_offset_est)) > 0.0))
        wt_setfreq(&d->mixer2, d->mixer_center.freq + freq_offset_est);
    if ((d->afc) && (d->dcd) && (fabs(d->mixer2.freq - d->mixer_center.freq) > 2.0))
    {
        if (d->countdown > 0) d->countdown--;
        else
        {
            wt_setfreq(&d->mixer_center, d->mixer2.freq);
            if (d->mixer_center.freq < d->lockingbw / 2.0) wt_setfreq(&d->mixer_center, d->lockingbw / 2.0);
            if (d->mixer_center.freq > (d->Fs / 2.0 - d->lockingbw / 2.0)) wt_setfreq(&d->mixer_center, d->Fs / 2.0 - d->lockingbw / 2.0);
            jo_coarse_bigchange(d->coarse);
            for (int j = 0; j < d->bbnfft; j++) { d->bbcycbuff[j].re = 0; d->bbcycbuff[j].im = 0; }
        }
    }
    else d->countdown = 4;
    record_status(d);
}

static void msk_write(jo_demod *d, const int16_t *ptr, long n) /* mskdemodulator.cpp:313-488 */
{
    for (long i = 0; i < n; i++)
    {
        double dval = ((double)ptr[i]) / 32768.0;
        coarse_ring_step(d, dval, 0);

        cpx cval = cscale(wt_cis(&d->mixer2), dval);
        cpx sig2;
        sig2.re = fir_update_and_process(d->fir_re, cval.re);
        sig2.im = fir_update_and_process(d->fir_im, cval.im);
        double dabval = sqrt(sig2.re * sig2.re + sig2.im * sig2.im);
        msk_ebno_update(&d->ebno, dabval);
        sig2 = cscale(sig2, agc_update(d->agc, dabval));
        double abval = sqrt(sig2.re * sig2.re + sig2.im * sig2.im);
        if (abval > 2.84) sig2 = cscale(sig2, (2.84 / abval));

        cpx pt_d = delaything_update_dont_touch(&d->delayedsmpl, sig2);
        cpx pt_msk; pt_msk.re = sig2.re; pt_msk.im = pt_d.im;

        double st_eta = iir_update(&d->st_iir_resonator, hypot(pt_msk.re, pt_msk.im));
        cpx st_m1; st_m1.re = st_eta; st_m1.im = -delay_update(&d->delayt8, st_eta);
        cpx st_out = cmul(wt_cis(&d->st_osc), st_m1);
        double st_angle_error = atan2(st_out.im, st_out.re);
        double weighting = fabs(tanh(st_angle_error));
        if (!d->dcd) wt_advance_fraction_of_wave(&d->st_osc, -(1.0 - weighting) * st_angle_error * (0.05 / 360.0));
        else wt_advance_fraction_of_wave(&d->st_osc, -(1.0 - weighting) * st_angle_error * (0.003 / 360.0));

        if (wt_if_have_passed_point(&d->st_osc, d->ee))
        {
            double ct_xt = tanh(sig2.im) * sig2.re;
            double ct_xt_d = tanh(pt_d.re) * pt_d.im;
            double ct_ec = ct_xt_d - ct_xt;
            if (ct_ec > M_PI) ct_ec = M_PI;
            if (ct_ec < -M_PI) ct_ec = -M_PI;
            if (ct_ec > M_PI_2) ct_ec = M_PI_2;
            if (ct_ec < -M_PI_2) ct_ec = -M_PI_2;
            double carrier_aggression = 12.0 * d->correctionfactor;
            if (d->dcd) carrier_aggression = 8.0 * d->correctionfactor;
            wt_increase_phase_deg(&d->mixer2, carrier_aggression * 1.0 * ct_ec);
            wt_increase_freq(&d->mixer2, carrier_aggression * 0.01 * ct_ec);

            ma_update_signed(d->marg, ct_ec / 2.0);
            delaything_update(&d->dt, &pt_msk);
            cpx rot; rot.re = cos(d->marg->val); rot.im = sin(d->marg->val);
            pt_msk = cmul(pt_msk, rot);

            double tda = (fabs((pt_msk).re * 0.75) - 1.0);
            double tdb = (fabs((pt_msk).im * 0.75) - 1.0);
            d->mse = ma_update(d->msema, (tda * tda) + (tdb * tdb));
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
            if (d->nrx >= 12) emit_soft(d);
        }
        wt_next(&d->mixer2);
        wt_next(&d->mixer_center);
        wt_next(&d->st_osc);
    }
}

/* ---------------- public API ---------------- */
jo_demod *jo_demod_create(const jo_settings *s)
{
    trig_init();
    jo_demod *d = (jo_demod *)calloc(1, sizeof(jo_demod));
    d->kind = s->kind;
    if (d->kind == JO_KIND_OQPSK) { oqpsk_ctor(d); oqpsk_set_settings(d, s); }
    else { msk_ctor(d); msk_set_settings(d, s); }
    return d;
}
void jo_demod_set_settings(jo_demod *d, const jo_settings *s)
{
    if (d->kind == JO_KIND_OQPSK) oqpsk_set_settings(d, s); else msk_set_settings(d, s);
}
void jo_demod_destroy(jo_demod *d)
{
    if (!d) return;
    free(d->bbcycbuff); free(d->bbtmpbuff); fir_free(d->fir_re); fir_free(d->fir_im);
    if (d->kind == JO_KIND_OQPSK) fastfir_free(&d->fir_pre);
    free(d->prefilt);
    jo_coarse_destroy(d->coarse); agc_free(d->agc); ma_free(d->ebno.E); ma_free(d->ebno.E2); ma_free(d->marg);
    ma_free(d->msema); ma_free(d->msecalc.pointmean); ma_free(d->msecalc.msema);
    free(d->dt.buffer); free(d->delayedsmpl.buffer);
    free(d->soft.p); free(d->status.p); free(d->symbols.p);
    free(d);
}
void jo_demod_set_flags(jo_demod *d, int afc, int sql, int cpu_reduce) { d->afc = afc; d->sql = sql; d->cpuReduce = cpu_reduce; }
void jo_demod_set_dcd(jo_demod *d, int dcd) { d->dcd = dcd; }
void jo_demod_center_freq_changed(jo_demod *d, double f) { center_freq_changed(d, f); }
long jo_demod_write(jo_demod *d, const int16_t *pcm, long n)
{
    if (d->kind == JO_KIND_OQPSK) oqpsk_write(d, pcm, n); else msk_write(d, pcm, n);
    return 2 * n;
}
long jo_demod_take_soft(jo_demod *d, int16_t *dst, long cap) { return gtake(&d->soft, dst, sizeof(int16_t), cap); }
long jo_demod_take_status(jo_demod *d, double *dst, long caprows) { return gtake(&d->status, dst, 6 * sizeof(double), caprows); }
void jo_demod_capture_symbols(jo_demod *d, int on) { d->capture_symbols = on; }
long jo_demod_take_symbols(jo_demod *d, double *dst, long caprows) { return gtake(&d->symbols, dst, 3 * sizeof(double), caprows); }
int jo_demod_pending_soft(jo_demod *d) { return d->nrx; }
double jo_demod_get_mse(jo_demod *d) { return d->mse; }
double jo_demod_get_freq_est(jo_demod *d) { return d->mixer2.freq; }
double jo_demod_get_freq_center(jo_demod *d) { return d->mixer_center.freq; }

/* JFastFir::SetKernel(rrc(alpha, K, Fs, fsym), nfft) + update(x) on n complex samples (re, im interleaved): the operation
 * JAERO/tests/jfastfir_tests.cpp:31-58 checks against the recorded output of JAERO v1.0.4.11 */
void jo_fastfir_run(const double *in, long n, double alpha, int K, int nfft, double Fs, double fsym, double *out)
{
    double *pts = (double *)malloc(sizeof(double) * (size_t)(K + 8));
    const int np = jo_rrc_design(alpha, K, Fs, fsym, pts);
    fastfir_t f;
    memset(&f, 0, sizeof f);
    fastfir_set_kernel_real(&f, pts, np, nfft);
    cpx *x = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    for (long i = 0; i < n; i++) { x[i].re = in[2 * i]; x[i].im = in[2 * i + 1]; }
    fastfir_update(&f, x, n);
    for (long i = 0; i < n; i++) { out[2 * i] = x[i].re; out[2 * i + 1] = x[i].im; }
    free(x); free(pts);
    fastfir_free(&f);
}

/* burst demodulators: same translation unit (shares the static primitives above) */
#include "jaero_oracle_burst.c"
