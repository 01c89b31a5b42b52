// k_coarse7.h -- EXPERIMENT, not in the library: the 2^14-point transform on 256 threads with 64 points each, so that TWO workgroups could
// share a CU (round 3).  Result: correct by its CPU model, but the compiler's schedule of the 64-point pass needs 314 registers (458 for the
// whole transform) against the 256 a second workgroup leaves: 69-476 spilled registers.  Expected gain if it fitted (from k_coarse6_13): ~7 %
// of the coarse kernel.  Left here with its test kernel (coarse_bench.hip: k_fft_test7); C7_STOP cuts the transform short for register counts.
//
// Same function as k_coarse6 (CoarseFreqEstimate::ProcessBasebandData + FreqOffsetEstimateSlot, JAERO/coarsefreqestimate.cpp:90-137,
// JAERO/oqpskdemodulator.cpp:629-677).  What k_coarse6_13 showed at the MSK rates: two independent workgroups on a CU
// overlap what one workgroup cannot -- one's LDS exchanges and HBM phases with the other's arithmetic.  At 2^14 points that needs a
// workgroup of 256 threads (four wavefronts, one per SIMD; two workgroups = two wavefronts per SIMD, 256 registers each) holding 64 points per
// thread, and 64 KiB of LDS per workgroup: every exchange moves a plane in two halves.
//
//   n = 256 n1 + 16 n2 + n3        k = k1 + 64 k2 + 1024 k3        (n1, k1 < 64;  n2, n3, k2, k3 < 16)
//   pass 1: FFT64 over n1 (radix-4 stage + four FFT16; outputs in split-4 order: slot 16 m + q holds X[4 q + m]), x W_N^(k1 (n mod 256))
//   exchange 1 (halves k1 < 32 / >= 32): L = (k1 & 31) * 256 + 16 n2 + n3
//   pass 2: four FFT16 over n2 (k1 = k1a + 16 g), x W_256^(k2 n3)
//   exchange 2 (halves k2 < 8 / >= 8): L = k1 + 64 (k2 & 7) + 513 n3
//   pass 3: four FFT16 over n3 (k2 = k2lo + 4 h): X[t + 256 h + 1024 k3] -> slot h + 4 k3
// natural order on entry and exit (slot = index >> 8, thread = index & 255).  Index maps and bank behaviour: tests/test_coarse_fft14_e32_model.py
// (model_fft64).
#pragma once
#include "../../jaero_amd/csrc/k_coarse6.h"

#ifndef C7_STOP
#define C7_STOP 0
#endif
#define C7_XCH 8208 // doubles: half a plane (exchange 2: 16 rows at stride 513)

__device__ __forceinline__ constexpr int c7_k(int s) { return 4 * (s & 15) + (s >> 4); }

// in-place forward 64-point DFT, natural order in, split-4 order out
__device__ __forceinline__ void c7_fft64(CV<64> &x)
{
#pragma clang fp contract(fast)
#pragma unroll
    for (int j = 0; j < 16; j++)
    {
        const double ar = x.r[j], ai = x.i[j], br = x.r[j + 16], bi = x.i[j + 16], cr = x.r[j + 32], ci = x.i[j + 32], dr = x.r[j + 48], di = x.i[j + 48];
        const double s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;   // a + c, a - c
        const double s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;   // b + d, b - d
        // m = 0: a + b + c + d;  m = 2: a - b + c - d;  m = 1: (a - c) - i (b - d);  m = 3: (a - c) + i (b - d)
        x.r[j] = s0r + s2r; x.i[j] = s0i + s2i;
        double u1r = s1r + s3i, u1i = s1i - s3r;
        double u2r = s0r - s2r, u2i = s0i - s2i;
        double u3r = s1r - s3i, u3i = s1i + s3r;
        cmul_w64(u1r, u1i, j);
        cmul_w64(u2r, u2i, (2 * j) & 63);
        cmul_w64(u3r, u3i, (3 * j) & 63);
        x.r[j + 16] = u1r; x.i[j + 16] = u1i;
        x.r[j + 32] = u2r; x.i[j + 32] = u2i;
        x.r[j + 48] = u3r; x.i[j + 48] = u3i;
        if ((j & 3) == 3) C6_FENCE;
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
        CV<16> in, out;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = x.r[16 * m + j]; in.i[j] = x.i[16 * m + j]; }
        regfft<16>(in, out);
#pragma unroll
        for (int j = 0; j < 16; j++) { x.r[16 * m + j] = out.r[j]; x.i[16 * m + j] = out.i[j]; }
        C6_FENCE;
    }
}

// x[slot of k] *= p1^k for k = 1 .. 63 (split-4 order): e_q = p1^(4q) by a chain, the three other residues one product more each
__device__ __forceinline__ void c7_twiddle64(CV<64> &x, const double2 p1)
{
#pragma clang fp contract(fast)
    auto app = [&](int slot, const double2 w) __attribute__((always_inline)) {
        const double r = x.r[slot] * w.x - x.i[slot] * w.y, i = x.r[slot] * w.y + x.i[slot] * w.x;
        x.r[slot] = r; x.i[slot] = i;
    };
    const double2 p2 = c6_sq(p1), p3 = cmul2(p2, p1), p4 = c6_sq(p2);
    app(16, p1); app(32, p2); app(48, p3);
    double2 e = p4;
#pragma unroll
    for (int q = 1; q < 16; q++)
    {
        app(q, e);
        app(16 + q, cmul2(e, p1));
        app(32 + q, cmul2(e, p2));
        app(48 + q, cmul2(e, p3));
        if (q < 15) e = cmul2(e, p4);
    }
}

// In-place forward 2^14-point DFT of a 256-thread workgroup's data, natural distribution in and out.  xch: C7_XCH doubles.
__device__ __forceinline__ void wg_fft14_e64(CV<64> &d, double *xch, const double2 *__restrict__ tw, int t)
{
#pragma clang fp contract(fast)
    const double2 st1 = tw[t], st2 = tw[64 * (t & 15)]; // W_N^(n mod 256); W_256^n3
    const int k1a = t >> 4, n3 = t & 15;
    const int e1r = k1a * 256 + n3;   // reader of exchange 1: + 16 (g & 1) * 256 + 16 n2
    const int e2w = k1a + 513 * n3;   // writer of exchange 2: + 16 g + 64 (k2 & 7)
    // ---- pass 1 ----
    c7_fft64(d);
#if C7_STOP == 5
    return;
#endif
    C6_FENCE;
    c7_twiddle64(d, st1);
    C6_FENCE;
#if C7_STOP == 1
    return;
#endif
    // ---- exchange 1: a plane in two halves (k1 < 32, k1 >= 32) ----
    auto exchange1 = [&](double (&v)[64]) __attribute__((always_inline)) {
        double nv[64];
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            c6_bar();
#pragma unroll
            for (int s = 0; s < 64; s++)
                if ((c7_k(s) >> 5) == half) (xch + (c7_k(s) & 31) * 256)[t] = v[s];
            c6_bar();
#pragma unroll
            for (int g = 2 * half; g < 2 * half + 2; g++)
#pragma unroll
                for (int n2 = 0; n2 < 16; n2++) nv[16 * g + n2] = (xch + 16 * (g & 1) * 256 + 16 * n2)[e1r];
        }
#pragma unroll
        for (int s = 0; s < 64; s++) v[s] = nv[s];
    };
    exchange1(d.r);
    exchange1(d.i);
    C6_FENCE;
#if C7_STOP == 2
    return;
#endif
    // ---- pass 2: slots 16 g + n2 -> 16 g + k2 ----
#pragma unroll
    for (int g = 0; g < 4; g++)
    {
        CV<16> in, out;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = d.r[16 * g + j]; in.i[j] = d.i[16 * g + j]; }
        regfft<16>(in, out);
        c4_twiddle16(out, st2);
#pragma unroll
        for (int j = 0; j < 16; j++) { d.r[16 * g + j] = out.r[j]; d.i[16 * g + j] = out.i[j]; }
        C6_FENCE;
    }
#if C7_STOP == 3
    return;
#endif
    // ---- exchange 2: a plane in two halves (k2 < 8, k2 >= 8); reader t3 = k1 + 64 k2lo, slot 16 h + n3 holds k2 = k2lo + 4 h ----
    auto exchange2 = [&](double (&v)[64]) __attribute__((always_inline)) {
        double nv[64];
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            c6_bar();
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int k2 = 8 * half; k2 < 8 * half + 8; k2++) (xch + 16 * g + 64 * (k2 & 7))[e2w] = v[16 * g + k2];
            c6_bar();
#pragma unroll
            for (int h = 2 * half; h < 2 * half + 2; h++)
#pragma unroll
                for (int m = 0; m < 16; m++) nv[16 * h + m] = (xch + 256 * (h & 1) + 513 * m)[t];
        }
#pragma unroll
        for (int s = 0; s < 64; s++) v[s] = nv[s];
    };
    exchange2(d.r);
    exchange2(d.i);
    C6_FENCE;
#if C7_STOP == 4
    return;
#endif
    // ---- pass 3: FFT16 over n3 for h = 0 .. 3; X[t + 256 h + 1024 k3] -> slot h + 4 k3 (natural) ----
    {
        CV<64> o;
#pragma unroll
        for (int h = 0; h < 4; h++)
        {
            CV<16> in, out;
#pragma unroll
            for (int j = 0; j < 16; j++) { in.r[j] = d.r[16 * h + j]; in.i[j] = d.i[16 * h + j]; }
            regfft<16>(in, out);
#pragma unroll
            for (int k3 = 0; k3 < 16; k3++) { o.r[4 * k3 + h] = out.r[k3]; o.i[4 * k3 + h] = out.i[k3]; }
            C6_FENCE;
        }
#pragma unroll
        for (int s = 0; s < 64; s++) { d.r[s] = o.r[s]; d.i[s] = o.i[s]; }
    }
}
