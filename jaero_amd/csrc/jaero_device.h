// jaero_device.h -- shared device-side definitions for libjaero_hip.so (gfx950 only).
//
// Data layout in HBM (one jaero_ctx = one bank of channels on one GPU; nchp = nch rounded up to 64):
//   * channel c lives in wave-group g = c/64, lane l = c%64; one wavefront demodulates 64 channels, one channel per
//     lane, because every stage after the matched filter is a per-sample scalar recurrence (AGC -> timing PLL ->
//     carrier loop) that cannot be spread over lanes.
//   * scalar state: SoA  S[field][nchp] (double), I[field][nchp] (int): a wave's load of one field is one 512 B row.
//   * sample-rate rings (AGC, EbNo, MSK delay lines): [group][slot][lane]   -> coalesced 512 B rows, slot advances
//     once per sample for all lanes.
//   * symbol-rate rings (marg, dt, pointmean, msema): [channel][slot]       -> per-lane positions (symbol instants
//     are not aligned between channels).
//   * coarse-frequency ring: [channel][nfft] complex double = mixer_center.WTCISValue()*dval, as the reference's
//     bbcycbuff (a packed {pcm, table index} form was tried first: its 4x-overlapped table gathers in the coarse kernel
//     cost more than the 12 extra bytes per sample).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jd_libm.h" // jd_atan2 (correctly rounded), jd_hypot (glibc 2.35's, bit for bit): what the loops feed back must round as the host's does

#define JD_WTSIZE 19999
#define JD_WAVE 64

// ---- double state fields (S) ----
enum
{
    S_M2_PTR, S_M2_STEP, S_M2_FREQ,
    S_MC_PTR, S_MC_STEP, S_MC_FREQ,
    S_ST_PTR, S_ST_STEP, S_ST_FREQ, S_ST_LAST,
    S_AGC_SUM,
    S_EB_ESUM, S_EB_E2SUM, S_EB_EBNO,
    S_D1,                       // delays: abval^2 of previous sample
    S_D41_1, S_D41_2, S_D41_3,  // delayt41 history of st_diff   (x[n-1..n-3])
    S_D42_1, S_D42_2, S_D42_3,  // delayt42 history of st_d1out
    S_D8_1, S_D8_2,             // delayt8 history of st_eta (OQPSK)
    S_RES_X1, S_RES_X2, S_RES_Y1, S_RES_Y2,
    S_LF_X1, S_LF_X2, S_LF_Y1, S_LF_Y2,
    S_SIG2L_RE, S_SIG2L_IM, S_PTD_RE, S_PTD_IM,
    S_MARG_SUM, S_PM_SUM, S_MSEMA_SUM, S_MSE,
    S_DIFF_LAST,
    S_LOCKINGBW, S_THRESH,
    S_MFB_A0_RE, S_MFB_A0_IM,          // k_msk_fb with four pairs: the back half's partial filter sum for the next launch's first sample
    S_PRE_PTR, S_PRE_STEP, S_PRE_FSUM, // fb == 8400: mixer_fir_pre phase / step, sum of mixer2's frequency over the current write
    S_PRE_PTR_NEXT, S_PRE_STEP_NEXT,   // ... as k_pre8400_mix's last stretch leaves them; k_pre8400_commit makes them current
    S_NFIELDS
};
// ---- int state fields (I) ----
enum
{
    I_AGC_POS, // write position in the window ring (JPtrs::win)
    I_AGC_HOLD, // OQPSK: samples for which the AGC (re-created by setSettings) still sees zeros leaving its window while the EbNo meter keeps its values
    I_BB_PTR, I_COARSE_CNT,
    I_MARG_POS, I_DT_POS, I_PM_POS, I_MSEMA_POS,
    I_YUI, I_SIG2L_INIT, I_FLAGS, I_COUNTDOWN, I_COUNTDOWN2, I_EMPTYING, I_NEST,
    I_SOFT_CNT, I_SYM_CNT, I_LOG_CNT, I_OVERFLOW,
    I_DLY_POS,   // MSK: shared slot of delayedsmpl / delayt8 rings
    I_NFIELDS
};
// Read-only data through the CONSTANT address space: with a wave-uniform address such a load is a scalar load (lgkmcnt) that the compiler
// may also hoist and reuse -- a load through a plain pointer held in a kernel-argument struct is neither (any store may alias it), and as
// a vector load it queues behind every other vector load in flight (vmcnt retires in order).  Only for memory this kernel does not write.
typedef const __attribute__((address_space(4))) int jd_cint;
typedef const __attribute__((address_space(4))) double jd_cdouble;
__device__ __forceinline__ int jd_sload(const int *q) { return *(jd_cint *)q; }
__device__ __forceinline__ double jd_sload(const double *q) { return *(jd_cdouble *)q; }

// state access of the continuous sample kernels (ch, nchp in scope)
// [channel group of 4][row][4] layout of the 8400 bps prefilter's per-sample arrays (k_pre8400.h: PRE_XI): element (row, channel) of an array of `rows` rows
#define JD_G4(row, ch, rows) ((((size_t)((ch) >> 2)) * (size_t)(rows) + (size_t)(row)) * 4 + (size_t)((ch) & 3))
#define LDF(f) (p.S[(size_t)(f) * nchp + ch])
#define LDI(f) (p.I[(size_t)(f) * nchp + ch])

#define JF_AFC 1
#define JF_SQL 2
#define JF_CPUREDUCE 4
#define JF_DCD 8

struct JGeom
{
    int kind, nch, nchp, ngroups;
    double Fs, fb;
    int Fs_int;
    int fir_n;      // 55 (OQPSK) / 2*SPS (MSK)
    int agc_len;    // round(4*Fs) OQPSK, round(Fs) MSK
    int ebno_len;   // 2*Fs
    int win_len;    // entries of the one ring that holds both windows' samples: max(agc_len, ebno_len) (agc_len without JAERO_FLAG_EBNO)
    int nfft, nfft_log2;
    int marg_len, dt_len, pm_len, msema_len; // dt_len = length+1 (ring size)
    double ee;
    double w4, w8;  // OQPSK fractional-delay weights (Delay<double>::update weighting)
    int sps, sps2;  // MSK: SamplesPerSymbol, SamplesPerSymbol/2
    double res_b0, res_b1, res_b2, res_a1, res_a2;
    double lf_b0, lf_b1, lf_b2, lf_a1, lf_a2;
    double stref_freq;
    double correctionfactor;
    unsigned flags;
    int soft_cap, sym_cap, log_cap;
};

struct JPtrs
{
    double *S;
    int *I;
    // [ng][win_len][64]: |sig2| of the last win_len samples.  The reference keeps three buffers of it -- the AGC's moving average, the EbNo meter's E
    // and (squared) E2 (agc->Update(dabval), ebnomeasure->Update(dabval): oqpskdemodulator.cpp:463-466, mskdemodulator.cpp:375-378; DSP.cpp:370-379,
    // 408-416, 493-505, 729-744) -- 24 B written and 24 B read per sample where 8 B written and 16 B read say the same
    double *win;
    double2 *bbring;     // [nchp][nfft] complex double, exactly the reference's bbcycbuff entries
    double *y;           // [nchp][nfft]
    double *marg;        // [nchp][marg_len]
    double2 *dt;         // [nchp][dt_len]
    double *pm;          // [nchp][pm_len]
    double *msema;       // [nchp][msema_len]
    double *firsave;     // [ng][2][fir_n][64]
    double2 *dly;        // MSK delayedsmpl ring [ng][sps+1][64]
    double *dly8;        // MSK delayt8 ring     [ng][sps2+1][64]
    int16_t *soft;       // [nchp][soft_cap]
    double *sym;         // [nchp][sym_cap][3]
    double *slog;        // [nchp][log_cap][6]
    const double2 *cis;  // [19999]
    const double *taps2; // [2*fir_n] taps repeated twice
    double *symrec;      // k_oqpsk_fb: [nchp][JD_SYMREC_LEN][8] one 64-byte record per symbol pair (see fb_back)
};
#define JD_SYMREC_LEN 800

__device__ __forceinline__ int jd_cisidx(double wtptr)
{
    int t = (int)wtptr;
    if (t >= JD_WTSIZE) t = 0;
    if (t < 0) t = JD_WTSIZE - 1;
    return t;
}
// WaveTable::WTnextFrame (JAERO/DSP.cpp:70-77) without last_WTptr bookkeeping
__device__ __forceinline__ void jd_wt_next(double &ptr, double &step)
{
    if (step < 0) step = 0;
    ptr += step;
    while (((int)ptr) >= JD_WTSIZE) ptr -= JD_WTSIZE;
}
// WaveTable::SetFreq(double) (JAERO/DSP.cpp:151-156)
__device__ __forceinline__ void jd_wt_setfreq(double &freq, double &step, double f, double samplerate)
{
    freq = f;
    if (freq < 0) freq = 0;
    step = (freq) * ((double)JD_WTSIZE) / samplerate;
}
// WaveTable::IncresePhaseDeg -> SetPhaseDeg (JAERO/DSP.cpp:169-180)
__device__ __forceinline__ void jd_wt_inc_phase_deg(double &ptr, double phase_deg)
{
    phase_deg += (360.0 * ptr / ((double)JD_WTSIZE));
    phase_deg = fmod(phase_deg, 360.0);
    while (phase_deg < 0) phase_deg += 360.0;
    ptr = (phase_deg / 360.0) * ((double)JD_WTSIZE);
}
// WaveTable::AdvanceFractionOfWave (JAERO/DSP.h:56)
__device__ __forceinline__ void jd_wt_advance_fraction(double &ptr, double f)
{
    ptr += f * JD_WTSIZE;
    while (ptr >= JD_WTSIZE) ptr -= JD_WTSIZE;
    while (ptr < 0) ptr += JD_WTSIZE;
}
// WaveTable::IfHavePassedPoint (JAERO/DSP.cpp:222-238); frac receives FractionOfSampleItPassesBy
__device__ __forceinline__ bool jd_wt_passed(double last_ptr, double ptr, double step, double fraction_of_wave, double &frac)
{
    double pt = (fraction_of_wave * JD_WTSIZE);
    double tl = last_ptr - pt;
    double tp = ptr - pt;
    if (tl < 0.0) tl += JD_WTSIZE;
    if (tp < 0.0) tp += JD_WTSIZE;
    if ((tl > (3.0 * JD_WTSIZE / 4.0)) && (tp < (1.0 * JD_WTSIZE / 4.0)))
    {
        frac = tp / step;
        return true;
    }
    return false;
}
// Qt 5.9 qRound (qglobal.h:525-526)
__device__ __forceinline__ int jd_qround(double d)
{
    return d >= 0.0 ? (int)(d + 0.5) : (int)(d - (double)((int)(d - 1)) + 0.5) + (int)(d - 1);
}
__device__ __forceinline__ int jd_softbit(double v)
{
    int ibit = jd_qround(v);
    if (ibit > 255) ibit = 255;
    if (ibit < 0) ibit = 0;
    return ibit;
}

// The EbNo meters (OQPSKEbNoMeasure / MSKEbNoMeasure, JAERO/DSP.cpp:729-744,493-505) never feed back into the signal path; their
// output EbNo = 0.8 EbNo + 0.2 t[n] forgets a term after k samples as 0.8^k (0.8^192 = 2.5e-19).  The ring sums are kept up to
// date on every sample, but the divide/log10 part is only evaluated over the last JD_EBNO_TAIL samples of a launch -- the value
// anyone can read after the launch differs from the every-sample evaluation by < 1e-16.
#define JD_EBNO_TAIL 192

// Matched-filter taps: every kernel reads its bank's own taps (JPtrs::taps2 / BPtrs::taps2, uploaded by *_create): the sample kernels
// copy them into LDS once per launch (jd_fir_eval) or take them as scalar kernel arguments (JTaps28); the burst MSK kernel reads
// p.taps2 inside its loop with a wave-uniform index (scalar loads).  There is NO process-global tap table: the values depend on
// (kind, fb, Fs) and banks of different (fb, Fs) are alive together (tests/test_gpu_scale.py::test_msk_family_banks_alive_together).

// Matched-filter evaluation for one sample of 64 channels (one per lane): sum over i of taps[i] * x[n-FIRN+i], oldest first,
// re and im chains, one multiplication and one addition per tap and chain, each rounded: FIR::FIRUpdateAndProcess (DSP.cpp:292-304) as the
// reference's x86-64 release build executes it (mulsd, addsd: checked in the binary).  Until round 5 this was one fma per tap, the last
// arithmetic difference between these kernels and the reference.
// History: the oldest TAILN = FIRN-LDSN inputs in registers (tre[j] = x[n-LDSN-1-j]), the newest LDSN in an LDS ring
// ([slot][lane], oldest at fir_slot).  The taps are read from LDS too (ltap, a wave-uniform address: a broadcast), NOT from
// the constant segment: LDS operations return in order, so the compiler can wait for exactly the read it needs (lgkmcnt(N))
// while D steps are in flight; with one scalar load pending it has to drain everything (lgkmcnt(0)).  Written as a plain
// loop, every tap was: ds_read, s_waitcnt lgkmcnt(0), 2 x v_fmac -- one LDS round trip per tap, 40 per sample, and with a
// single wavefront per SIMD nothing else to run meanwhile (SQ_WAIT_ANY was 45 % of the wave's cycles).
// FUSED (one fma per tap and chain) is for the burst demodulators only: their input has passed an FFT filter (the Hilbert transform) whose
// round-off is not the reference FFT's, and measured on a bank of bursts the op-for-op filter, glibc's hypot and the correctly rounded atan2
// change neither a soft byte nor the distribution of soft-symbol differences there, while costing 10 % of both burst workloads
// (profiles/r5_burst_ab.json).  The continuous demodulators, whose whole chain IS the reference's op for op, never fuse.
template <int FIRN, int LDSN, int D, bool FUSED = false, int TAILA = 1>
__device__ __forceinline__ void jd_fir_eval(const double *lre, const double *lim, const double *ltap, const double (&tre)[TAILA],
                                            const double (&tim)[TAILA], int fir_slot, int lane, double &ore, double &oim)
{
    constexpr int TAILN = FIRN - LDSN;
    double pr[D], pi[D], pt[D];
    int slot = fir_slot;
    auto fetch = [&](int s, int q) {
        pt[q] = ltap[s];
        if (s >= TAILN)
        {
            pr[q] = lre[slot * 64 + lane];
            pi[q] = lim[slot * 64 + lane];
            slot++;
            if (slot >= LDSN) slot = 0;
        }
    };
#pragma unroll
    for (int s = 0; s < D; s++) fetch(s, s);
    __builtin_amdgcn_sched_barrier(0);
    double are = 0, aim = 0;
#pragma unroll
    for (int s = 0; s < FIRN; s++)
    {
        const int q = s % D;
        const double xr = (s < TAILN) ? tre[(TAILN - 1 - s) < 0 ? 0 : (TAILN - 1 - s)] : pr[q];
        const double xi = (s < TAILN) ? tim[(TAILN - 1 - s) < 0 ? 0 : (TAILN - 1 - s)] : pi[q];
        if constexpr (FUSED) { are = fma(pt[q], xr, are); aim = fma(pt[q], xi, aim); }
        else { are = are + pt[q] * xr; aim = aim + pt[q] * xi; }
        // keep the software pipeline as written: the empty asm orders the two sums before the next read (without it instruction
        // selection places every pure arithmetic instruction after the last read: all 94 reads first, 260 registers of them)
        asm volatile("" : "+v"(are), "+v"(aim));
        if (s + D < FIRN) fetch(s + D, q);
        __builtin_amdgcn_sched_barrier(0);
    }
    ore = are; oim = aim;
}

// The 55 root-raised-cosine taps are bitwise symmetric (t[i] == t[54 - i]; checked by jaero_create): 28 distinct values, handed to the
// kernel by value so that they live in SGPRs and are scalar operands of the filter's fmas -- no tap reads from LDS at all.
struct JTaps28 { double t[28]; };

// jd_fir_eval with the taps as scalar operands (FIRN must be 55) and NO address arithmetic: the LDS ring position is wave-uniform
// (and the same for every wavefront of a launch, give or take a few samples), so the newest-LDSN part of the sum exists in LDSN
// versions, one per ring position, each with compile-time LDS offsets; a switch picks one.  All wavefronts of a CU are in the same one
// or two versions at any time, so the instruction cache sees little of the ~30 KiB this unrolls to.
typedef __attribute__((address_space(3))) const double jd_lds_cdouble;
template <int FIRN, int LDSN, int D, int S, int TAP0, int NQ = LDSN>
__device__ __forceinline__ void jd_fir_lds_part(jd_lds_cdouble *lre_l, jd_lds_cdouble *lim_l, const JTaps28 &tp, double &are, double &aim)
{
    constexpr int TAILN = TAP0; // tap index that meets the oldest LDS entry
    double pr[D], pi[D];
#pragma unroll
    for (int q = 0; q < D; q++) { pr[q] = lre_l[((S + q) % LDSN) * 64]; pi[q] = lim_l[((S + q) % LDSN) * 64]; }
    __builtin_amdgcn_sched_barrier(0);
    // products one step ahead of the sums: with multiplication and addition rounded separately (round 5) a step is mul, mul, add, add, and
    // written in that order each addition waited for the multiplication issued right in front of it
    auto tap_of = [&](int q) { const int s = TAILN + q; return tp.t[s <= 27 ? s : 54 - s]; };
    double mr = tap_of(0) * pr[0], mi = tap_of(0) * pi[0];
#pragma unroll
    for (int q = 0; q < NQ; q++)
    {
        double nr = 0, ni = 0;
        if (q + 1 < NQ) { nr = tap_of(q + 1) * pr[(q + 1) % D]; ni = tap_of(q + 1) * pi[(q + 1) % D]; }
        are = are + mr;
        aim = aim + mi;
        // keep the software pipeline as written (see jd_fir_eval)
        asm volatile("" : "+v"(are), "+v"(aim));
        if (q + D < NQ) { pr[q % D] = lre_l[((S + q + D) % LDSN) * 64]; pi[q % D] = lim_l[((S + q + D) % LDSN) * 64]; }
        mr = nr; mi = ni;
        __builtin_amdgcn_sched_barrier(0);
    }
}
// the same sum with run-time ring addresses (one call per launch: the first sample's output from the saved history)
template <int FIRN, int LDSN, int D, int TAILA>
__device__ __forceinline__ void jd_fir_eval_sym(const double *lre, const double *lim, const JTaps28 &tp, const double (&tre)[TAILA],
                                                const double (&tim)[TAILA], int fir_slot, int lane, double &ore, double &oim)
{
    static_assert(FIRN == 55, "symmetric-tap filter is the 55-tap RRC");
    constexpr int TAILN = FIRN - LDSN;
    double are = 0, aim = 0;
#pragma unroll
    for (int s = 0; s < TAILN; s++)
    {
        const double tap = tp.t[s <= 27 ? s : 54 - s];
        are = are + tap * tre[TAILN - 1 - s];
        aim = aim + tap * tim[TAILN - 1 - s];
    }
    int slot = fir_slot;
#pragma unroll 1
    for (int q = 0; q < LDSN; q++)
    {
        const int s = TAILN + q;
        const double tap = tp.t[s <= 27 ? s : 54 - s]; // a scalar load per step: this is the once-per-launch path
        are = are + tap * lre[slot * 64 + lane];
        aim = aim + tap * lim[slot * 64 + lane];
        slot++;
        if (slot >= LDSN) slot = 0;
    }
    ore = are; oim = aim;
}
template <int FIRN, int LDSN, int D, int TAILA>
__device__ __forceinline__ void jd_fir_eval_sym_static(const double *lre, const double *lim, const JTaps28 &tp, const double (&tre)[TAILA],
                                                       const double (&tim)[TAILA], int fir_slot, int lane, double &ore, double &oim)
{
    static_assert(FIRN == 55 && LDSN <= 40, "symmetric-tap filter is the 55-tap RRC; the switch below has 40 cases");
    constexpr int TAILN = FIRN - LDSN;
    jd_lds_cdouble *lre_l = (jd_lds_cdouble *)lre + lane, *lim_l = (jd_lds_cdouble *)lim + lane;
    double are = 0, aim = 0;
#pragma unroll
    for (int s = 0; s < TAILN; s++)
    {
        const double tap = tp.t[s <= 27 ? s : 54 - s];
        are = are + tap * tre[TAILN - 1 - s];
        aim = aim + tap * tim[TAILN - 1 - s];
    }
#define JD_FIR_CASE(S) case S: if constexpr (S < LDSN) jd_fir_lds_part<FIRN, LDSN, D, (S < LDSN ? S : 0), FIRN - LDSN>(lre_l, lim_l, tp, are, aim); break;
    switch (fir_slot)
    {
        JD_FIR_CASE(0) JD_FIR_CASE(1) JD_FIR_CASE(2) JD_FIR_CASE(3) JD_FIR_CASE(4) JD_FIR_CASE(5) JD_FIR_CASE(6) JD_FIR_CASE(7)
        JD_FIR_CASE(8) JD_FIR_CASE(9) JD_FIR_CASE(10) JD_FIR_CASE(11) JD_FIR_CASE(12) JD_FIR_CASE(13) JD_FIR_CASE(14) JD_FIR_CASE(15)
        JD_FIR_CASE(16) JD_FIR_CASE(17) JD_FIR_CASE(18) JD_FIR_CASE(19) JD_FIR_CASE(20) JD_FIR_CASE(21) JD_FIR_CASE(22) JD_FIR_CASE(23)
        JD_FIR_CASE(24) JD_FIR_CASE(25) JD_FIR_CASE(26) JD_FIR_CASE(27) JD_FIR_CASE(28) JD_FIR_CASE(29) JD_FIR_CASE(30) JD_FIR_CASE(31)
        JD_FIR_CASE(32) JD_FIR_CASE(33) JD_FIR_CASE(34) JD_FIR_CASE(35) JD_FIR_CASE(36) JD_FIR_CASE(37) JD_FIR_CASE(38) JD_FIR_CASE(39)
    default: break;
    }
#undef JD_FIR_CASE
    ore = are; oim = aim;
}

// (Round 6: the same sum in 4 or 6 versions over ring BLOCKS -- k_msk_fb.h's mfb_fir_continue_v -- instead of 36 over ring positions, a sixth of the code,
// was built for this kernel too and changes nothing: 10.44 - 10.49 against 10.43 - 10.45 ms per step.  The front half is not this loop's long pole.)
// The same sum WITHOUT its last term (the newest input): the 54 older terms do not depend on the sample being formed, so a front half that
// is alone on its SIMD (one pair per workgroup: banks of at most 2 x #CUs groups) evaluates them while the carrier oscillator's table value
// for that sample is still on its way from L2, and adds tap[54] x[n] from registers when it arrives.  Call BEFORE the new input overwrites
// LDS slot `slot_old` (which holds the oldest LDS entry; the caller has already moved it into tre[0] / tim[0]).
template <int FIRN, int LDSN, int D, int TAILA>
__device__ __forceinline__ void jd_fir_eval_sym_static_but_last(const double *lre, const double *lim, const JTaps28 &tp, const double (&tre)[TAILA],
                                                                const double (&tim)[TAILA], int slot_old, int lane, double &ore, double &oim)
{
    static_assert(FIRN == 55 && LDSN <= 40 && LDSN >= 8, "symmetric-tap filter is the 55-tap RRC; the switch below has 40 cases");
    constexpr int TAILN = FIRN - LDSN;
    jd_lds_cdouble *lre_l = (jd_lds_cdouble *)lre + lane, *lim_l = (jd_lds_cdouble *)lim + lane;
    double are = 0, aim = 0;
#pragma unroll
    for (int s = 0; s < TAILN; s++)
    {
        const double tap = tp.t[s <= 27 ? s : 54 - s];
        are = are + tap * tre[TAILN - 1 - s];
        aim = aim + tap * tim[TAILN - 1 - s];
    }
#define JD_FIR_CASE(S) case S: if constexpr (S < LDSN) jd_fir_lds_part<FIRN, LDSN, D, (S < LDSN ? (S + 1) % LDSN : 0), FIRN - LDSN, LDSN - 1>(lre_l, lim_l, tp, are, aim); break;
    switch (slot_old)
    {
        JD_FIR_CASE(0) JD_FIR_CASE(1) JD_FIR_CASE(2) JD_FIR_CASE(3) JD_FIR_CASE(4) JD_FIR_CASE(5) JD_FIR_CASE(6) JD_FIR_CASE(7)
        JD_FIR_CASE(8) JD_FIR_CASE(9) JD_FIR_CASE(10) JD_FIR_CASE(11) JD_FIR_CASE(12) JD_FIR_CASE(13) JD_FIR_CASE(14) JD_FIR_CASE(15)
        JD_FIR_CASE(16) JD_FIR_CASE(17) JD_FIR_CASE(18) JD_FIR_CASE(19) JD_FIR_CASE(20) JD_FIR_CASE(21) JD_FIR_CASE(22) JD_FIR_CASE(23)
        JD_FIR_CASE(24) JD_FIR_CASE(25) JD_FIR_CASE(26) JD_FIR_CASE(27) JD_FIR_CASE(28) JD_FIR_CASE(29) JD_FIR_CASE(30) JD_FIR_CASE(31)
        JD_FIR_CASE(32) JD_FIR_CASE(33) JD_FIR_CASE(34) JD_FIR_CASE(35) JD_FIR_CASE(36) JD_FIR_CASE(37) JD_FIR_CASE(38) JD_FIR_CASE(39)
    default: break;
    }
#undef JD_FIR_CASE
    ore = are; oim = aim;
}

// tanh as glibc 2.35 computes it (sysdeps/ieee754/dbl-64/s_tanh.c over s_expm1.c, the fdlibm algorithms with glibc's grouping of the
// expm1 polynomial), operation for operation: with correctly rounded +, *, / and no contraction the result is the reference
// libm's, bit for bit (scripts/tanh_check.c: 0 differences from the host's tanh / expm1 on 2e8 arguments) -- and it is half the
// instructions of the device library's tanh (~90 against ~180 wave instructions), which the carrier detector calls twice per symbol.
__device__ __forceinline__ double jd_with_hi(double x, int h) { return __hiloint2double(h, __double2loint(x)); }
__device__ __forceinline__ double jd_expm1(double x)
{
    const double one = 1.0, huge = 1.0e+300, tiny = 1.0e-300, o_threshold = 7.09782712893383973096e+02,
                 ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
                 Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
                 Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
    double y, hi, lo, c = 0, t, e, hxs, hfx, r1;
    int k;
    unsigned hx = (unsigned)__double2hiint(x);
    const unsigned xsb = hx & 0x80000000u;
    hx &= 0x7fffffffu;
    if (hx >= 0x4043687Au) // |x| >= 56 ln2
    {
        if (hx >= 0x40862E42u)
        {
            if (hx >= 0x7ff00000u)
            {
                if (((hx & 0xfffffu) | (unsigned)__double2loint(x)) != 0) return x + x;
                return (xsb == 0) ? x : -1.0;
            }
            if (x > o_threshold) return huge * huge;
        }
        if (xsb != 0 && x + tiny < 0.0) return tiny - one;
    }
    if (hx > 0x3fd62e42u) // |x| > 0.5 ln2
    {
        if (hx < 0x3FF0A2B2u) // |x| < 1.5 ln2
        {
            if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
            else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
        }
        else
        {
            k = (int)(invln2 * x + ((xsb == 0) ? 0.5 : -0.5));
            t = k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    }
    else if (hx < 0x3c900000u) return x;
    else k = 0;
    hfx = 0.5 * x;
    hxs = x * hfx;
    {
        const double R1 = one + hxs * Q1, h2 = hxs * hxs;
        const double R2 = Q2 + hxs * Q3, h4 = h2 * h2;
        const double R3 = Q4 + hxs * Q5;
        r1 = R1 + h2 * R2 + h4 * R3;
    }
    t = 3.0 - r1 * hfx;
    e = hxs * ((r1 - t) / (6.0 - x * t));
    if (k == 0) return x - (x * e - hxs);
    e = (x * (e - c) - c);
    e -= hxs;
    if (k == -1) return 0.5 * (x - e) - 0.5;
    if (k == 1)
    {
        if (x < -0.25) return -2.0 * (e - (x + 0.5));
        return one + 2.0 * (x - e);
    }
    if (k <= -2 || k > 56)
    {
        y = one - (e - x);
        y = jd_with_hi(y, __double2hiint(y) + (k << 20));
        return y - one;
    }
    if (k < 20)
    {
        t = jd_with_hi(one, 0x3ff00000 - (0x200000 >> k)); // 1 - 2^-k
        y = t - (e - x);
        y = jd_with_hi(y, __double2hiint(y) + (k << 20));
    }
    else
    {
        t = jd_with_hi(one, (0x3ff - k) << 20); // 2^-k
        y = x - (e + t);
        y += one;
        y = jd_with_hi(y, __double2hiint(y) + (k << 20));
    }
    return y;
}
__device__ __noinline__ double jd_tanh_full(double x)
{
    const double one = 1.0, two = 2.0, tiny = 1.0e-300;
    double t, z;
    const int jx = __double2hiint(x);
    const int ix = jx & 0x7fffffff;
    if (ix >= 0x7ff00000) return (jx >= 0) ? one / x + one : one / x - one;
    if (ix < 0x40360000) // |x| < 22
    {
        if ((ix | __double2loint(x)) == 0) return x;
        if (ix < 0x3c800000) return x * (one + x);
        // |x| >= 1: t = expm1(2|x|), z = 1 - 2/(t+2);  |x| < 1: t = expm1(-2|x|), z = -t/(t+2).  One call and one division for both
        // cases (the same operations per lane): lanes on either side of |x| = 1 -- constellation points sit there -- do not run
        // the exponential twice.
        const bool big = ix >= 0x3ff00000;
        const double ax2 = two * fabs(x);
        t = jd_expm1(big ? ax2 : -ax2);
        const double q = (big ? two : t) / (t + two);
        z = big ? one - q : -q;
    }
    else z = one - tiny;
    return (jx >= 0) ? z : -z;
}

// The same function for 2^-55 <= |x| < 6.5 (everything an AGC'd constellation point can be) as straight-line code: glibc's branches on
// the reduction index k -- k = 0, -1, <= -2 for the arguments below 1, 2..19 above -- become four short tails and a select, and the
// quick reduction for |x| < 1.5 ln 2 is the general one with t = +-1 (t * ln2_hi is exact).  Every lane performs exactly the operations
// the branchy form performs for its k, so the result is still the host libm's (scripts/tanh_check.c -DFAST: 0 differences on 3e8
// arguments).  Written with branches, the few lanes at a symbol instant sat on both sides of |x| = 1 (constellation points are there)
// and of the k boundaries: every path ran every time, and the carrier detector cost more than with the device library's tanh.
__device__ __forceinline__ double jd_tanh(double xin)
{
    const double one = 1.0, two = 2.0, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
                 Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
    const int jx = __double2hiint(xin), ix = jx & 0x7fffffff;
    if (!(ix >= 0x3c800000 && ix < 0x401a0000)) return jd_tanh_full(xin); // |x| < 2^-55, |x| >= 6.5, inf, nan: the general form
    const bool big = ix >= 0x3ff00000;
    const double ax2 = two * fabs(xin);
    double x = big ? ax2 : -ax2;
    // t = expm1(x)
    const unsigned hx = (unsigned)__double2hiint(x) & 0x7fffffffu;
    const int k = (hx > 0x3fd62e42u) ? (int)(invln2 * x + (big ? 0.5 : -0.5)) : 0;
    const double tk = (double)k;
    const double hi = x - tk * ln2_hi, lo = tk * ln2_lo;
    x = hi - lo;
    const double c = (hi - x) - lo;
    const double hfx = 0.5 * x, hxs = x * hfx;
    const double R1 = one + hxs * Q1, h2 = hxs * hxs, R2 = Q2 + hxs * Q3, h4 = h2 * h2, R3 = Q4 + hxs * Q5;
    const double r1 = R1 + h2 * R2 + h4 * R3;
    const double t3 = 3.0 - r1 * hfx;
    const double e = hxs * jd_div(r1 - t3, 6.0 - x * t3);
    const double r_k0 = x - (x * e - hxs);
    double e2 = (x * (e - c) - c);
    e2 -= hxs;
    const double r_m1 = 0.5 * (x - e2) - 0.5;
    double yn = one - (e2 - x);
    yn = jd_with_hi(yn, __double2hiint(yn) + (k << 20));
    const double r_neg = yn - one;
    const double tm = jd_with_hi(one, 0x3ff00000 - (0x200000 >> (k & 31))); // 1 - 2^-k
    double ym = tm - (e2 - x);
    ym = jd_with_hi(ym, __double2hiint(ym) + (k << 20));
    const double t = (k == 0) ? r_k0 : (k == -1) ? r_m1 : (k <= -2) ? r_neg : ym;
    const double q = jd_div(big ? two : t, t + two);
    const double z = big ? one - q : -q;
    return (jx >= 0) ? z : -z;
}
