// jd_libm.h -- atan2 and hypot for the sample loops, rounded as the host libm (glibc 2.35) rounds it as nearly as any function can be; and
// glibc 2.35's hypot restated operation for operation (bit-identical to the host's).  Both are what the product kernels call (round 5).
//
// Why: the timing loops feed atan2 of every sample back into two oscillators (JAERO/oqpskdemodulator.cpp:472-494,
// JAERO/mskdemodulator.cpp:384-408), so the closer the result is to the reference's double the closer the loops track it.  Measured
// (scripts/ubench/atan2_rates.hip, MI355X, 6.7e7 operand pairs of the timing detector's kind): the device library's atan2 differs from the
// host libm's in 26.9 % of the calls (by one ulp; 523 calls by more), its hypot in 14.5 %.  glibc 2.35's atan2 itself is NOT correctly
// rounded (within 0.503 ulp: 9.5e-4 of its results are the other neighbour -- scripts/atan2_check.c against __float128 -- and which ones
// depends on whether its ifunc picked the fma build, so "the host's bits" are not a function of the arguments alone).  The nearest a
// device function can get is therefore the correctly rounded value, and that is what this one returns in all but ~2e-7 of the calls:
//   u = min/max, c = round(64 u)/64, t = (min - c max)/(max + c min) as a double-double quotient (|t| <= 2^-7), atan(t) by
//   t - t^3/3 + t^5/5 - t^7/7 + t^9/9, atan(c) from a 64-entry table held one entry per LANE (ds_bpermute: no memory access), the
//   quadrant constant folded in BEFORE the one final rounding.  The unrounded (hi + lo) is within 0.4 x 2^-64 of the result
//   (scripts/atan2_check.c, 2e8 operand pairs against __float128), i.e. the rounded result is within 0.5002 ulp.  ~55 fp64 instructions,
//   straight-line.  (With Ziv's rounding test and a double-double second stage it was correctly rounded on all 2e8 pairs, and at 274 ns
//   per dependent call against the device library's 145 ns slower than what it replaces: the second stage runs whenever one lane of 64
//   needs it.  Not kept.)
//   Zeros, infinities, NaN, exponents outside 2^+-300: the device library's atan2, as before.
// scripts/atan2_check.c compiles THIS header for the host (the hardware hooks replaced by portable C, the reciprocal seed deliberately
// bad) and compares with the host libm and with __float128.
#pragma once
#include <stdint.h>

#ifndef JDA_HOST_CHECK
#include <hip/hip_runtime.h>
#define JDA_FN __device__ __forceinline__
#define JDA_RCP(x) __builtin_amdgcn_rcp(x)
#define JDA_SQRT(x) __builtin_sqrt(x)
#define JDA_FMA(a, b, c) __builtin_fma(a, b, c)
#define JDA_LIB_ATAN2(y, x) atan2(y, x)
#define JDA_LIB_HYPOT(x, y) hypot(x, y)
__device__ __forceinline__ uint32_t jda_hi32(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ uint32_t jda_lo32(double x) { return (uint32_t)__double2loint(x); }
__device__ __forceinline__ double jda_words(uint32_t h, uint32_t l) { return __hiloint2double((int)h, (int)l); }
#endif

#include "jd_atan2_tbl.h"

#define JDA_PIO2_HI 0x1.921fb54442d18p+0
#define JDA_PIO2_LO 0x1.1a62633145c07p-54

// the per-lane copy of the table (entry i = 1..64 in lane i - 1): {hi word, lo word of atan(i/64)'s leading double, the float tail}
struct JdAtanLane
{
    int hi_h, hi_l, lo_f;
};
#ifndef JDA_HOST_CHECK
__device__ __forceinline__ JdAtanLane jd_atan_lane_table(int lane)
{
    JdAtanLane t;
    const double h = JD_ATAN_HI[(lane & 63) + 1];
    t.hi_h = __double2hiint(h); t.hi_l = __double2loint(h);
    t.lo_f = __float_as_int(JD_ATAN_LOF[(lane & 63) + 1]);
    return t;
}
// entry i (0..64) for this lane.  A bpermute reads other lanes' registers and returns nothing for lanes the exec mask has switched off: when the
// call is made with part of the wavefront inactive (no call site does today -- padding lanes run the sample loops -- but nothing else enforces it:
// ADVICE r5), the entry comes from the table in memory instead.  The test is wave-uniform (one s_cmp, never taken in the sample loops).
__device__ __forceinline__ void jda_fetch(const JdAtanLane &T, int i, double &A_hi, double &A_lo)
{
    if (__builtin_expect(__builtin_amdgcn_read_exec() != ~0ull, 0))
    {
        A_hi = JD_ATAN_HI[i];
        A_lo = (double)JD_ATAN_LOF[i];
        return;
    }
    const int addr = (i - 1) << 2; // i = 0 reads lane 63 (address wraps) and is discarded below
    const int hh = __builtin_amdgcn_ds_bpermute(addr, T.hi_h);
    const int hl = __builtin_amdgcn_ds_bpermute(addr, T.hi_l);
    const int lf = __builtin_amdgcn_ds_bpermute(addr, T.lo_f);
    const bool z = i == 0;
    A_hi = __hiloint2double(z ? 0 : hh, z ? 0 : hl);
    A_lo = (double)__int_as_float(z ? 0 : lf);
}
#endif

template <class JDA_TBL>
JDA_FN double jd_atan2_t(double y, double x, const JDA_TBL &T)
{
    const uint32_t hx = jda_hi32(x), hy = jda_hi32(y);
    // both exponents in [2^-300, 2^300): no zero, infinity, NaN, denormal; no intermediate under- or overflow below.  The straight-line
    // part runs for EVERY lane (on whatever the out-of-range lanes hold) and the library call replaces their result at the end: the table
    // lookup is a ds_bpermute, which reads other lanes' registers and returns nothing for lanes the exec mask has switched off -- with the
    // usual early return, one padding lane of a ragged bank on the library path corrupted its neighbours' table entries.
    const uint32_t ex = (hx & 0x7fffffffu) - 0x2d300000u, ey = (hy & 0x7fffffffu) - 0x2d300000u;
    const bool special = ex >= 0x25800000u || ey >= 0x25800000u;
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const bool sw = ay > ax;
    const double mx = sw ? ay : ax, mn = sw ? ax : ay;
    const bool xneg = (int32_t)hx < 0;
    // index: u0 within 2^-13 of mn/mx is enough (c only has to be NEAR u)
    const double u0 = mn * JDA_RCP(mx);
    const double k = JDA_FMA(u0, 64.0, 0x1.8p52);
    const int i = (int)(jda_lo32(k) & 0x7fu);
    const double c = (k - 0x1.8p52) * 0.015625;
    double A_hi, A_lo;
    jda_fetch(T, i > 64 ? 64 : i, A_hi, A_lo);
    // n = mn - c mx
    const double ph = c * mx, pl = JDA_FMA(c, mx, -ph);
    const double s = mn - ph; // exact (Sterbenz; c = 0: ph = 0)
    const double n_hi = s - pl, n_lo = (s - n_hi) - pl;
    // d = mx + c mn
    const double qh = c * mn, ql = JDA_FMA(c, mn, -qh);
    const double d_hi = mx + qh, d_lo = (qh - (d_hi - mx)) + ql;
    // 1 / d_hi: two Newton steps from the hardware seed (any seed good to 2^-13 gives 2^-52)
    double rd = JDA_RCP(d_hi);
    rd = JDA_FMA(JDA_FMA(-d_hi, rd, 1.0), rd, rd);
    rd = JDA_FMA(JDA_FMA(-d_hi, rd, 1.0), rd, rd);
    const double t_hi = n_hi * rd;
    const double e = JDA_FMA(-t_hi, d_lo, JDA_FMA(-d_hi, t_hi, n_hi) + n_lo);
    const double t_lo = e * rd;
    const double t2 = t_hi * t_hi;
    const double P = JDA_FMA(t2, JDA_FMA(t2, JDA_FMA(t2, 0x1.c71c71c71c71cp-4, -0x1.2492492492492p-3), 0x1.999999999999ap-3), -0x1.5555555555555p-2);
    double lo = JDA_FMA(t_hi * t2, P, t_lo) + A_lo;
    const double s1 = A_hi + t_hi;
    lo += t_hi - (s1 - A_hi);
    // quadrant: |result| = m pi/2 + sigma (s1 + lo)
    const double m = sw ? 1.0 : (xneg ? 2.0 : 0.0);
    const double sigma = (sw != xneg) ? -1.0 : 1.0;
    const double K_hi = m * JDA_PIO2_HI, hs = sigma * s1;
    const double R = K_hi + hs;
    const double lo2 = JDA_FMA(sigma, lo, m * JDA_PIO2_LO) + (hs - (R - K_hi));
    const double z = R + lo2; // the one rounding
    double res = jda_words((jda_hi32(z) & 0x7fffffffu) | (hy & 0x80000000u), jda_lo32(z));
    if (special) res = JDA_LIB_ATAN2(y, x);
    return res;
}

JDA_FN double jd_atan2(double y, double x, const JdAtanLane &T) { return jd_atan2_t(y, x, T); }

// ---- division ----------------------------------------------------------------------------------------------------------------------
// a / b for the sample loops' variable divisors (round 6).  The compiler expands an fp64 division into
//   v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal (4 fma), q = a r, e = fma(-b, q, a), v_div_fmas (= fma(e, r, q) plus the
//   scaling), v_div_fixup            -- 11 instructions, the IEEE quotient.
// v_div_scale multiplies by a power of two only when an exponent is extreme and v_div_fixup only repairs zeros, infinities, NaN and denormals;
// for finite operands with 2^-500 <= |b| <= 2^500 and a = 0 or 2^-500 <= |a| <= 2^500 neither does anything, and the remaining EIGHT
// instructions below are the same operations on the same values: the same bits as a / b (a zero quotient comes out as +0 where IEEE gives -0
// for a = -0; every call site takes the magnitude of the quotient or subtracts it from a non-zero value).  Three instructions and two links of
// the dependency chain less per division: jd_tanh has two of them on the carrier loop's critical path in every sample.
// Checked on the device against `/`: scripts/ubench/div_check.hip, 2^33 operand pairs incl. quotients next to a rounding boundary and the
// operands of each call site's kind: no difference.  Call sites (each with |b| far inside the range): jd_tanh (b = t + 2 in [1, 4.5e5] and
// b = 6 - x t3 in [2, 6.1]), jd_hypot (b = 2 h, only taken for 2^-200 < h < 2^200), the AGC gain (b >= 1e-6), MSEcalc (mu >= 1e-6).
JDA_FN double jd_div(double a, double b)
{
    double r = JDA_RCP(b);
    r = JDA_FMA(JDA_FMA(-b, r, 1.0), r, r);
    r = JDA_FMA(JDA_FMA(-b, r, 1.0), r, r);
    const double q = a * r;
    return JDA_FMA(JDA_FMA(-b, q, a), r, q);
}

// ---- hypot -----------------------------------------------------------------------------------------------------------------------
// glibc 2.35 sysdeps/ieee754/dbl-64/e_hypot.c, the branch without a fast fma (the x86-64 baseline build), operation for operation:
// h = sqrt(ax ax + ay ay), then ONE correction step h -= (t1 + t2) / (2 h) whose t1, t2 depend on which side of 2 ay the first h fell.
// Both forms of (t1, t2) are written as straight-line code and selected (a wavefront of 64 channels has lanes on both sides in
// nearly every call); operands outside glibc's unscaled range (2^-459 < ay, ax < 2^511), infinities and NaN take the device library's
// value, patched in afterwards; ax >= ay 2^54 returns ax + ay as glibc does.  Bit-identical to the host libm on 6.4e8 operand pairs
// (scripts/atan2_check.c -DHYPOT).  This -- not atan2 -- is what decides whether the loops track the reference: DESIGN 9 item 18.
JDA_FN double jd_hypot(double x, double y)
{
    x = __builtin_fabs(x); y = __builtin_fabs(y);
    const double ax = x < y ? y : x, ay = x < y ? x : y;
    const bool unscaled = ax < 0x1p+511 && ay > 0x1p-459;
    const bool far = ax >= ay * 0x1p+54; // glibc: ax >= ay / EPS, EPS = 2^-54: the same comparison, the quotient is exact
    const double h = JDA_SQRT(ax * ax + ay * ay);
    const bool near = h <= 2.0 * ay;
    const double delta = h - (near ? ay : ax);
    const double t1n = ax * (2.0 * delta - ax), t2n = (delta - 2.0 * (ax - ay)) * delta;
    const double t1f = 2.0 * delta * (ax - 2.0 * ay), t2f = (4.0 * delta - ay) * ay + delta * delta;
    const double t1 = near ? t1n : t1f, t2 = near ? t2n : t2f;
    // (jd_div's operand range: h between 2^-200 and 2^200, i.e. everything but pathological inputs, which keep the library's division)
    const bool mid = h > 0x1p-200 && h < 0x1p+200;
    double res = h - jd_div(t1 + t2, 2.0 * h);
    if (!mid) res = h - (t1 + t2) / (2.0 * h);
    if (far) res = ax + ay;
    if (!unscaled) res = JDA_LIB_HYPOT(x, y);
    return res;
}

#if !defined(JDA_HOST_CHECK)
// A/B builds only (make -C jaero_amd/csrc ab -> gpurun_tmp/libjaero_hip_libm.so / _libatan2.so, loaded through JAERO_HIP_LIB by
// scripts/recording_full.py and bench.py): the device library's functions, as the kernels called them until round 4.  Never the product.
#if defined(JD_LIBRARY_LIBM)
#define jd_hypot(x, y) hypot(x, y)
#endif
#if defined(JD_LIBRARY_LIBM) || defined(JD_LIBRARY_ATAN2)
#define jd_atan2(y, x, T) atan2(y, x)
#endif
#endif
