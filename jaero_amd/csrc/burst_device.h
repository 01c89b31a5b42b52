// burst_device.h -- device-side layout of a BURST demodulator bank (BurstOqpskDemodulator / BurstMskDemodulator,
// SURVEY.md section 8 row a3).  gfx950 only.
//
// The reference's burst writeData (JAERO/burstoqpskdemodulator.cpp:315-737, JAERO/burstmskdemodulator.cpp:371-754) is, per
// sample:  Hilbert fast-FIR -> AGC -> [d1 delay] -> [d2 delay] ---------------------------> mix / matched filter / tracking
//                                  \-> burst-timing detector -> peak detector -> trident FFT check --^ (retunes the demod)
// The left column never reads demodulator state, so it is a producer the tracking chain consumes through delay lines.  On the GPU:
//   k_hilbert      time-parallel 2048-tap Hilbert FIR over the PCM history ring (taps in SGPRs, register sliding windows)
//   k_burst_front  lane = channel: AGC, ONE ring of AGC'd analytic samples that serves d1, d2, bt_d1 and the trident buffer
//                  (they are all pure delays of the same signal), burst-timing moving averages, peak detector; emits at most one
//                  "trident buffer full" event per channel per segment (segments are <= tridentbuffer_sz samples)
//   k_trident      one 512-thread workgroup per event: the two zero-padded 2^15-point real FFTs as 2x2 register-resident
//                  2^14-point transforms (decimation in frequency by 2), peak searches, acceptance test
//   k_burst_*_demod lane = channel: applies the trident result at the sample the reference would, then the tracking chain
// Rings that advance once per sample for every channel are [group][slot][lane] (coalesced 512 B rows, wave-uniform slot);
// rings that advance only while a channel's burst gate is open (burst MSK) or once per symbol are [channel][slot].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JAERO_KIND_BURST_MSK_D 2
#define JAERO_KIND_BURST_OQPSK_D 3

// event kinds written to the per-channel event log (rows of 3 doubles: absolute sample index, kind, value)
#define BEV_SIGNAL 0
#define BEV_EBNO 1
#define BEV_FREQ 2
#define BEV_PEAK 3
#define BEV_TRIDENT 4

enum // double state
{
    // front end
    BS_AGC_SUM, BS_MA1_RE, BS_MA1_IM, BS_MAV1_SUM, BS_LASTDY,
    // demod
    BS_M2_PTR, BS_M2_STEP, BS_M2_FREQ, BS_MC_FREQ,
    BS_ST_PTR, BS_ST_STEP, BS_ST_FREQ, BS_ST_LAST,
    BS_STQ_PTR, // st_osc_quarter (OQPSK) / st_osc_half (MSK)
    BS_VOL_GAIN,
    BS_STR_RE, BS_STR_IM, BS_SAV_RE, BS_SAV_IM, BS_ROT_RE, BS_ROT_IM, BS_ROT_FREQ,
    BS_A1_1, BS_A1_2, BS_A1_3, BS_A1_4, BS_A1_5,
    BS_AGC2_SUM, BS_EB_ESUM, BS_EB_E2SUM, BS_EB_EBNO,
    BS_D1, BS_D41_1, BS_D41_2, BS_D41_3, BS_D42_1, BS_D42_2, BS_D42_3, BS_D8_1, BS_D8_2,
    BS_RES_X1, BS_RES_X2, BS_RES_Y1, BS_RES_Y2,
    BS_SIG2L_RE, BS_SIG2L_IM, BS_PTD_RE, BS_PTD_IM,
    BS_MSEMA_SUM, BS_MSE, BS_LASTMSE, BS_THRESH, BS_LOCKINGBW, BS_DIFF_LAST,
    BS_RESET_AT, // absolute index of the first sample behind the channel's last setSettings (0: the constructor's): where the reference's DelayThing pointers stand follows from it
    BS_NFIELDS
};
enum // int state
{
    BI_CNTDOWN, BI_MAXPOSCD, BI_TRI_PTR, BI_EV_POS,
    BI_STARTSTOP, BI_CNTR, BI_YUI, BI_INSERTPRE, BI_MSEMA_POS, BI_NRX,
    BI_FIR_POS, BI_AGC2_POS, BI_EB_POS, BI_DLY_POS, BI_D8_POS, BI_A1_POS, // burst MSK: rings that advance only while gated on
    BI_SOFT_CNT, BI_SYM_CNT, BI_EV_CNT, BI_OVERFLOW, BI_FLAGS,
    BI_BT_HOLD, // samples for which bt_d1 (a Delay<> refilled with zeros by setSettings) still returns zeros (k_burst_front<true>)
    BI_GCNT,    // burst MSK: gated samples since the last setSettings, modulo dly_len = where delayedsmpl's buffer_ptr stands in the reference
    BI_NFIELDS
};

struct TriResult // written by k_trident, read by the demod kernel at the event sample
{
    int ok;          // signal-only part of the acceptance test
    int pad;
    double freq;     // mixer2 frequency to set
    double phase_deg;// mixer2 phase to set
    double vol_gain;
    double metric;   // maxval (OQPSK) / minval (MSK), for the trace log
};

struct BGeom
{
    int kind, nch, nchp, ngroups;
    double Fs, fb, SPS;
    unsigned flags;
    // Hilbert
    int hil_ntaps, hil_lat, hist_len, maxseg;
    // front end
    int agc_len, cv_len, D1, D2, tri_sz, nb, nt;
    int ma1_len, mav1_len, fa_len, fa_lag, bt_lag, bt_len, PL;
    double fa_w, bt_w, pd_thr;
    // demod
    int fir_n, agc2_len, eb_len, msema_len, a1_lag, dly_len, d8_len;
    int win_ring; // entries of the ONE ring of |sig2| values that serves the AGC2 window and the EbNo meter's two windows: max(agc2_len, eb_len)
    int dly_ring, d8_ring; // burst MSK: sizes of the delayedsmpl / delayt8 rings, dly_len / d8_len rounded up to whole 64-byte cells of eight entries (k_burst_msk_fb.h)
    double a1_w, w4, w8, ee;
    double res_b0, res_b1, res_b2, res_a1, res_a2;
    double stref_freq, stq_step;
    int startstopstart, startProcessing, endRotation;
    int soft_cap, sym_cap, ev_cap;
};

struct BPtrs
{
    double *S; int *I;
    int16_t *pcmhist;            // [hist_len][nchp] frame-major PCM history ring
    double *him;                 // [ng][maxseg][64] imaginary part of the analytic signal of the current segment (the real part is a delayed copy of the PCM: read from pcmhist)
    double *agc_ring;            // [ng][agc_len][64]
    double *cvre, *cvim;         // [ng][cv_len][64]  AGC'd analytic samples
    double *ma1re, *ma1im;       // [ng][ma1_len][64]
    double *mav1;                // [ng][mav1_len][64]
    double *fa;                  // [ng][fa_len][64]
    double *bt;                  // [ng][bt_len][64]
    unsigned long long *ev_mask; // [ng] lanes of each group whose trident buffer filled in this segment (k_burst_front)
    int *ev_list, *ev_count;     // the same as a list of channels, ascending (k_ev_compact), and its length
    TriResult *tri;              // [nchp]
    // demod (burst OQPSK: uniform-slot rings [ng][len][64]; burst MSK: per-channel rings [nchp][len])
    double *eb_e;                // |sig2| of the last win_ring (gated) samples: agc2's window, E's window; E2's entries are its squares
    double *firsave;             // OQPSK [ng][2][fir_n][64]; MSK [nchp][2][fir_n]
    double2 *dly; double *dly8, *a1;
    double *msema;               // [nchp][msema_len]
    int16_t *soft; double *sym; double *evlog;
    const double2 *cis; const double *taps2; const double *hil_taps; // hil_taps[j] = imag(kernel[2j+1]), j < ntaps/4
    const double2 *tw14;         // W_8192^k (twiddle table of the 2^13-point transforms in k_trident)
    const double2 *tw15;         // W_32768^n, n < 16384 (pre-twiddle of the odd-bin half transforms in k_trident)
    const double2 *hilH;         // [4096] DFT of the Hilbert kernel's imaginary taps (zero-padded) / 4096      (k_hilbert_fft)
    const double2 *tw12;         // [4096] exp(-2 pi i k / 4096)                                                  (k_hilbert_fft)
};
