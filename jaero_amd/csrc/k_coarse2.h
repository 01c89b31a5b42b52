// k_coarse2.h -- the register-FFT building blocks: CV<L> (L complex points in registers), regfft<L> (radix-4 / radix-2 DFT of them),
// c2_log10, c4_twiddle16, c4_lds_barrier (wg_fft<LOG2N>, the 512-thread E x 32 x 16 transform k_trident used until round 4, left with it).  The coarse-frequency estimator kernels themselves are in k_coarse6.h (round 3); the first
// register-resident one, k_coarse2<LOG2N>, lived here (rounds 1-2).
#pragma once
#include "jaero_device.h"
#include "fft_consts.h"
#include "k_coarse.h"

#define C2_THREADS 512

template <int L>
struct CV
{
    double r[L], i[L];
};

// y = x * W_64^IDX  (IDX is a compile-time constant after unrolling)
__device__ __forceinline__ void cmul_w64(double &re, double &im, int idx)
{
#pragma clang fp contract(fast)
    idx &= 63;
    if (idx == 0) return;
    if (idx == 16) { const double t = re; re = im; im = -t; return; }   // * (-i)
    if (idx == 32) { re = -re; im = -im; return; }
    if (idx == 48) { const double t = re; re = -im; im = t; return; }   // * (+i)
    const double wr = jd_w64r(idx), wi = jd_w64i(idx); // literals (fft_consts.h), not table loads
    const double nr = re * wr - im * wi;
    const double ni = re * wi + im * wr;
    re = nr; im = ni;
}

// forward DFT of length L held in registers (radix-4 decimation in frequency, radix-2 tail), natural order in & out
template <int L>
__device__ __forceinline__ void regfft(const CV<L> &x, CV<L> &y)
{
#pragma clang fp contract(fast)
    if constexpr (L == 1) { y.r[0] = x.r[0]; y.i[0] = x.i[0]; }
    else if constexpr (L == 2)
    {
        y.r[0] = x.r[0] + x.r[1]; y.i[0] = x.i[0] + x.i[1];
        y.r[1] = x.r[0] - x.r[1]; y.i[1] = x.i[0] - x.i[1];
    }
    else
    {
        constexpr int Q = L / 4;
        CV<Q> u0, u1, u2, u3, z0, z1, z2, z3;
#pragma unroll
        for (int j = 0; j < Q; j++)
        {
            const double ar = x.r[j], ai = x.i[j], br = x.r[j + Q], bi = x.i[j + Q];
            const double cr = x.r[j + 2 * Q], ci = x.i[j + 2 * Q], dr = x.r[j + 3 * Q], di = x.i[j + 3 * Q];
            const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
            const double t2r = br + dr, t2i = bi + di;
            const double t3r = (bi - di), t3i = -(br - dr); // -i (b - d)
            u0.r[j] = t0r + t2r; u0.i[j] = t0i + t2i;
            double v1r = t1r + t3r, v1i = t1i + t3i, v2r = t0r - t2r, v2i = t0i - t2i, v3r = t1r - t3r, v3i = t1i - t3i;
            cmul_w64(v1r, v1i, j * (64 / L));
            cmul_w64(v2r, v2i, 2 * j * (64 / L));
            cmul_w64(v3r, v3i, 3 * j * (64 / L));
            u1.r[j] = v1r; u1.i[j] = v1i; u2.r[j] = v2r; u2.i[j] = v2i; u3.r[j] = v3r; u3.i[j] = v3i;
        }
        regfft<Q>(u0, z0); regfft<Q>(u1, z1); regfft<Q>(u2, z2); regfft<Q>(u3, z3);
#pragma unroll
        for (int q = 0; q < Q; q++)
        {
            y.r[4 * q + 0] = z0.r[q]; y.i[4 * q + 0] = z0.i[q];
            y.r[4 * q + 1] = z1.r[q]; y.i[4 * q + 1] = z1.i[q];
            y.r[4 * q + 2] = z2.r[q]; y.i[4 * q + 2] = z2.i[q];
            y.r[4 * q + 3] = z3.r[q]; y.i[4 * q + 3] = z3.i[q];
        }
    }
}

// 32-point DFT with a bounded live set: one in-place radix-2 stage, then the two 16-point halves one after the other (regfft<32>
// transforms all 32 points at every level and needs ~256 registers while it does)
__device__ __forceinline__ void regfft32_seq(CV<32> &x, CV<32> &y)
{
#pragma clang fp contract(fast)
    CV<16> lo, hi, z;
#pragma unroll
    for (int j = 0; j < 16; j++)
    {
        const double ar = x.r[j], ai = x.i[j], br = x.r[j + 16], bi = x.i[j + 16];
        lo.r[j] = ar + br; lo.i[j] = ai + bi;
        double dr = ar - br, di = ai - bi;
        cmul_w64(dr, di, 2 * j); // W_32^j
        hi.r[j] = dr; hi.i[j] = di;
    }
    __builtin_amdgcn_sched_barrier(0);
    regfft<16>(lo, z);
#pragma unroll
    for (int q = 0; q < 16; q++) { y.r[2 * q] = z.r[q]; y.i[2 * q] = z.i[q]; }
    __builtin_amdgcn_sched_barrier(0);
    regfft<16>(hi, z);
#pragma unroll
    for (int q = 0; q < 16; q++) { y.r[2 * q + 1] = z.r[q]; y.i[2 * q + 1] = z.i[q]; }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ double2 cmul2(const double2 a, const double2 b)
{
#pragma clang fp contract(fast)
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// tws[k] = base * step^k for k < 32, from two table values, by products of depth <= 6 (no per-element table gathers:
// 64 distinct 16-byte addresses per wave instruction were what bounded the first version of this kernel).
template <int E>
__device__ __forceinline__ void twiddle_powers(const double2 base, const double2 step, double2 (&B)[4], double2 (&A)[8])
{
    const double2 s2 = cmul2(step, step), s3 = cmul2(s2, step), s4 = cmul2(s2, s2);
    B[0] = base; B[1] = cmul2(base, step); B[2] = cmul2(base, s2); B[3] = cmul2(base, s3);
    A[0] = make_double2(1.0, 0.0);
    A[1] = s4; A[2] = cmul2(s4, s4); A[3] = cmul2(A[2], s4); A[4] = cmul2(A[2], A[2]);
    A[5] = cmul2(A[4], s4); A[6] = cmul2(A[3], A[3]); A[7] = cmul2(A[4], A[3]);
}

// log10(x) for x >= 1, ~2 ulp, ~30 instructions (ocml's correctly-rounded log10 costs ~110 and the kernel needs N of them per
// estimate).  x = m 2^e with m in [sqrt(1/2), sqrt(2)); log(m) = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716: odd series to s^19.
// The smoothed dB spectrum y[] only feeds an arg-max over bins (the emitted estimate is a bin index), so the last ulps of y never
// reach an output.
__device__ __forceinline__ double c2_log10(double x)
{
#pragma clang fp contract(fast)
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double d = m + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double sx = (m - 1.0) * r;
    const double z = sx * sx;
    double q = 1.0 / 19.0;
    q = fma(q, z, 1.0 / 17.0);
    q = fma(q, z, 1.0 / 15.0);
    q = fma(q, z, 1.0 / 13.0);
    q = fma(q, z, 1.0 / 11.0);
    q = fma(q, z, 1.0 / 9.0);
    q = fma(q, z, 1.0 / 7.0);
    q = fma(q, z, 1.0 / 5.0);
    q = fma(q, z, 1.0 / 3.0);
    const double lnm = 2.0 * fma(sx * z, q, sx);
    return fma((double)e, 0.30102999566398119521, lnm * 0.43429448190325182765);
}

// (The estimator kernel built on wg_fft, k_coarse2<LOG2N>, served the MSK rates until round 3: 6.76 ms per 65 536 estimates of 2^13 points,
// half of its threads idle in the 32-point pass.  k_coarse6.h replaces it: k_coarse6_13, 5.64 ms.  wg_fft stays for k_trident.)

// v[k] *= step^k (k < 16), the powers by products of depth <= 6 (k_pre8400.h's 4096-point transform)
__device__ __forceinline__ void c4_twiddle16(CV<16> &v, const double2 step) // v[k] *= step^k
{
#pragma clang fp contract(fast)
    double2 B[4], A[8];
    twiddle_powers<16>(make_double2(1.0, 0.0), step, B, A);
#pragma unroll
    for (int k = 1; k < 16; k++)
    {
        const double2 w = (k < 4) ? B[k & 3] : cmul2(A[k >> 2], B[k & 3]);
        const double r = v.r[k] * w.x - v.i[k] * w.y, i = v.r[k] * w.y + v.i[k] * w.x;
        v.r[k] = r; v.i[k] = i;
    }
}


// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for every global load and STORE in
// flight: behind the y[] update that is 32 stores per thread on their way to HBM, behind the ring prefetch 32 loads -- the prefetch
// was issued early precisely so that the peak search would run under it.
__device__ __forceinline__ void c4_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
