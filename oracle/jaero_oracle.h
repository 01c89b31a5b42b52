/* ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's demodulator hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product path
 * (jaero_amd/, libjaero_hip.so) must never include, link or call anything in oracle/.  See jaero_oracle.c. */
#ifndef JAERO_ORACLE_H
#define JAERO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { JO_KIND_MSK = 0, JO_KIND_OQPSK = 1, JO_KIND_BURST_MSK = 2, JO_KIND_BURST_OQPSK = 3 };

typedef struct
{
    int kind;                    /* JO_KIND_* */
    int coarsefreqest_fft_power; /* Settings::coarsefreqest_fft_power */
    double freq_center;          /* Hz */
    double lockingbw;            /* Hz */
    double fb;                   /* bps */
    double Fs;                   /* Hz */
    double signalthreshold;
} jo_settings;

typedef struct jo_demod jo_demod;

/* = constructor + setSettings(settings) + start() of OqpskDemodulator / MskDemodulator */
jo_demod *jo_demod_create(const jo_settings *s);
void jo_demod_destroy(jo_demod *d);
/* = setSettings on a live object (JAERO/oqpskdemodulator.cpp:175-289, JAERO/mskdemodulator.cpp:135-263) */
void jo_demod_set_settings(jo_demod *d, const jo_settings *s);
/* = setAFC / setSQL / setCPUReduce */
void jo_demod_set_flags(jo_demod *d, int afc, int sql, int cpu_reduce);
/* = DCDstatSlot */
void jo_demod_set_dcd(jo_demod *d, int dcd);
/* = CenterFreqChangedSlot */
void jo_demod_center_freq_changed(jo_demod *d, double freq_center);
/* = writeData(const char*, len) with len = 2*nsamples */
long jo_demod_write(jo_demod *d, const int16_t *pcm, long nsamples);

/* captured outputs (drained by the calls below) */
/* soft bits exactly as passed to processDemodulatedSoftBits, concatenated */
long jo_demod_take_soft(jo_demod *d, int16_t *dst, long cap);
/* one row of 6 doubles per FreqOffsetEstimateSlot: [n, freq_est(mixer2), freq_center(mixer_center), mse, ebno, signal] */
long jo_demod_take_status(jo_demod *d, double *dst, long caprows);
/* soft symbols: (re,im) of pt_qpsk / pt_msk after the residual rotation, one per symbol pair,
 * plus the mse value after that symbol: rows of 3 doubles.  Enabled by jo_demod_capture_symbols(d,1). */
void jo_demod_capture_symbols(jo_demod *d, int on);
long jo_demod_take_symbols(jo_demod *d, double *dst, long caprows);
/* pending soft bits not yet emitted (RxDataBits.size()) */
int jo_demod_pending_soft(jo_demod *d);
/* current values */
double jo_demod_get_mse(jo_demod *d);
double jo_demod_get_freq_est(jo_demod *d);
double jo_demod_get_freq_center(jo_demod *d);

/* stand-alone pieces for unit tests */
/* RootRaisedCosine::design (JAERO/DSP.h:316-338); returns number of points written */
int jo_rrc_design(double alpha, int firsize, double samplerate, double symbol_freq, double *points);
/* CISWT table (JAERO/DSP.cpp:11-30): 19999 (re,im) pairs */
void jo_cis_table(double *dst_re_im);
/* FFTWrapper<double>::transform semantics on top of the same radix-2 FFT as oracle/ref/shim/jfft.h */
void jo_fft(double *re_im, int n, int inverse);
/* CoarseFreqEstimate stand-alone: create/setSettings, process one nfft buffer, returns emitted estimate */
typedef struct jo_coarse jo_coarse;
jo_coarse *jo_coarse_create(int power, double lockingbw, double fb, double Fs);
void jo_coarse_destroy(jo_coarse *c);
void jo_coarse_bigchange(jo_coarse *c);
double jo_coarse_process(jo_coarse *c, const double *re_im);
void jo_coarse_get_y(jo_coarse *c, double *y);

/* ---- burst demodulators (jaero_oracle_burst.c): BurstOqpskDemodulator / BurstMskDemodulator ---- */
typedef struct jo_burst jo_burst;
/* = constructor + setAFC/SQL/CPUReduce defaults + setSettings + start(); kind = JO_KIND_BURST_* */
jo_burst *jo_burst_create(const jo_settings *s);
void jo_burst_destroy(jo_burst *d);
/* BurstOqpskDemodulator::setSettings / BurstMskDemodulator::setSettings on the live object (burstoqpskdemodulator.cpp:202-277,
 * burstmskdemodulator.cpp:150-325) */
void jo_burst_set_settings(jo_burst *d, const jo_settings *s);
void jo_burst_set_flags(jo_burst *d, int afc, int sql, int cpu_reduce);
void jo_burst_set_dcd(jo_burst *d, int dcd); /* BurstMskDemodulator::DCDstatSlot */
/* = writeData (mono) */
long jo_burst_write(jo_burst *d, const int16_t *pcm, long nsamples);
/* CenterFreqChangedSlot(freq_center): acts on burst MSK (burstmskdemodulator.cpp:327-342), empty for burst OQPSK (burstoqpskdemodulator.cpp:284-289) */
void jo_burst_center_freq_changed(jo_burst *d, double freq_center);
/* soft bits as passed to processDemodulatedSoftBits, concatenated; -1 = start-of-burst marker */
long jo_burst_take_soft(jo_burst *d, int16_t *dst, long cap);
/* rows of 3 doubles [absolute sample index, kind, value]; kind 0 SignalStatus(value), 1 EbNoMeasurmentSignal(value),
 * 2 Plottables(freq_est=value); with jo_burst_trace(d,1) also 3 = peak detector fired, 4 = trident check ran
 * (value = +metric accepted / -metric rejected) */
long jo_burst_take_events(jo_burst *d, double *dst, long caprows);
void jo_burst_trace(jo_burst *d, int on);
/* rows [re, im, mse] of pt_qpsk (burst OQPSK, while startstop>0) / pt_msk (burst MSK, every symbol instant) */
void jo_burst_capture_symbols(jo_burst *d, int on);
long jo_burst_take_symbols(jo_burst *d, double *dst, long caprows);
int jo_burst_pending_soft(jo_burst *d);
double jo_burst_get_mse(jo_burst *d);
double jo_burst_get_freq_est(jo_burst *d);
/* QJHilbertFilter pieces (JAERO/DSP.cpp:754-794) */
int jo_hilbert_kernel(int N, double *re_im);
typedef struct jo_hilbert jo_hilbert;
jo_hilbert *jo_hilbert_create(int N);
void jo_hilbert_destroy(jo_hilbert *h);
int jo_hilbert_latency(jo_hilbert *h);
void jo_hilbert_update(jo_hilbert *h, const int16_t *pcm, long n, double *out_re_im);

#ifdef __cplusplus
}
#endif
void jo_fastfir_run(const double *in, long n, double alpha, int K, int nfft, double Fs, double fsym, double *out);
#endif
