// k_coarse2.h -- register-resident coarse frequency estimator (replaces the four-step LDS/scratch version).
//
// Same function as k_coarse.h (CoarseFreqEstimate::ProcessBasebandData + FreqOffsetEstimateSlot,
// JAERO/coarsefreqestimate.cpp:90-137, JAERO/oqpskdemodulator.cpp:629-677, JAERO/mskdemodulator.cpp:490-519), but the
// three N-point fp64 FFTs never leave the chip:
//   * one 512-thread workgroup per channel-estimate (2 waves per SIMD, <=256 VGPRs each), each thread holds
//     E = N/512 complex points (E = 32 for N = 2^14, 16 for 2^13) in VGPRs;
//   * N = E x 32 x 16: an E-point FFT per thread in registers, exchange through LDS, 32-point FFTs in registers,
//     exchange, 16-point FFTs -- real and imaginary planes are exchanged one after the other so the exchange buffer is
//     N doubles (padded: 135 KB for N = 2^14), LDS strides chosen bank-conflict free for ds_write_b64 / ds_read_b64;
//   * the output distribution of one transform (thread v holds bins k = v mod 512) is exactly the input distribution of
//     the next, so FFT -> band-limit mask -> IFFT -> square -> FFT -> |.| -> dB smoothing runs register to register;
//     the inverse transform is the forward one with real/imag planes swapped (unnormalised, as FFTWrapper's x N / N).
// HBM traffic per estimate: the packed ring (4 B/sample) + y[] read/write; algorithmic bytes: 524 288 B.
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

// In-place forward N-point DFT of the workgroup's data, 512 threads, N = E x 32 x 16 (E = N/512 = 32 or 16).
// On entry thread t holds x[s*512 + t] in slot s (s < E); on exit thread t holds X[s*512 + t] in slot s.
// xch: LDS exchange buffer of max(E*528, 512*(E+1)) doubles.
// TWS: the twiddle table holds W_(N*TWS)^k (a table made for a TWS times longer transform), read with stride TWS.
template <int LOG2N, int TWS = 1>
__device__ __forceinline__ void wg_fft(CV<(1 << LOG2N) / C2_THREADS> &d, double *xch, const double2 *__restrict__ tw, int t)
{
#pragma clang fp contract(fast)
    constexpr int N = 1 << LOG2N;
    constexpr int E = N / C2_THREADS;          // 32 or 16
    constexpr int LOGE = (E == 32) ? 5 : 4;
    constexpr int G3 = E / 16;                 // 16-point FFTs per thread in pass 3
    constexpr int S1 = 528, S2 = E + 1;
    const int n3 = t & 15, n2 = t >> 4;        // pass-1 identity of this thread: (n2, n3) = n mod 512
    // the three table values of this transform, requested before the first butterfly (each used to be loaded, and waited
    // for, where it is consumed)
    const double2 tw_p1 = tw[((16 * n2) & (N - 1)) * TWS];
    const double2 tw_p2b = tw[((n3 * (t >> 4)) & (N - 1)) * TWS], tw_p2s = tw[((n3 * E) & (N - 1)) * TWS];

    // ---- pass 1: E-point FFT over n1, twiddle W_N^(16*n2*k1) ----
    {
        CV<E> a;
        regfft<E>(d, a);
        __builtin_amdgcn_sched_barrier(0);
        double2 B[4], A[8];
        twiddle_powers<E>(make_double2(1.0, 0.0), tw_p1, B, A);
#pragma unroll
        for (int k1 = 0; k1 < E; k1++)
        {
            const double2 w = (k1 < 4) ? B[k1 & 3] : cmul2(A[k1 >> 2], B[k1 & 3]);
            d.r[k1] = a.r[k1] * w.x - a.i[k1] * w.y;
            d.i[k1] = a.r[k1] * w.y + a.i[k1] * w.x;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- exchange 1: L1[k1][n2][n3] (k1 stride 528); pass-2 thread u: n3 = u&15, k1 = u>>4 (active if k1 < E) ----
    CV<32> b;
    const int k1u = t >> 4;
    const bool act2 = k1u < E;
    {
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < E; k1++) xch[k1 * S1 + t] = d.r[k1];
        __syncthreads();
        if (act2)
        {
#pragma unroll
            for (int m = 0; m < 32; m++) b.r[m] = xch[k1u * S1 + m * 16 + n3];
        }
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < E; k1++) xch[k1 * S1 + t] = d.i[k1];
        __syncthreads();
        if (act2)
        {
#pragma unroll
            for (int m = 0; m < 32; m++) b.i[m] = xch[k1u * S1 + m * 16 + n3];
        }
    }
    // ---- pass 2: 32-point FFT over n2, twiddle W_N^(n3*(k1 + E*k2)) ----
    if (act2)
    {
        CV<32> c;
        __builtin_amdgcn_sched_barrier(0);
        regfft<32>(b, c);
        __builtin_amdgcn_sched_barrier(0);
        double2 B[4], A[8];
        twiddle_powers<E>(tw_p2b, tw_p2s, B, A);
#pragma unroll
        for (int k2 = 0; k2 < 32; k2++)
        {
            const double2 w = (k2 < 4) ? B[k2 & 3] : cmul2(A[k2 >> 2], B[k2 & 3]);
            b.r[k2] = c.r[k2] * w.x - c.i[k2] * w.y;
            b.i[k2] = c.r[k2] * w.y + c.i[k2] * w.x;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- exchange 2: M[(k2*16 + n3)][k1] (row stride E+1); pass-3 thread v: k1 = v & (E-1), k2 = (v >> LOGE) + (512/E) r ----
    {
        const int k1v = t & (E - 1), k2b = t >> LOGE;
        CV<16> c[G3];
        __syncthreads();
        if (act2)
        {
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++) xch[(k2 * 16 + n3) * S2 + k1u] = b.r[k2];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < G3; r++)
#pragma unroll
            for (int m = 0; m < 16; m++) c[r].r[m] = xch[((k2b + (C2_THREADS / E) * r) * 16 + m) * S2 + k1v];
        __syncthreads();
        if (act2)
        {
#pragma unroll
            for (int k2 = 0; k2 < 32; k2++) xch[(k2 * 16 + n3) * S2 + k1u] = b.i[k2];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < G3; r++)
#pragma unroll
            for (int m = 0; m < 16; m++) c[r].i[m] = xch[((k2b + (C2_THREADS / E) * r) * 16 + m) * S2 + k1v];
        // ---- pass 3: 16-point FFT over n3; X[k1 + E*k2 + 32E*k3] -> slot r + G3*k3 ----
#pragma unroll
        for (int r = 0; r < G3; r++)
        {
            CV<16> o;
            regfft<16>(c[r], o);
#pragma unroll
            for (int k3 = 0; k3 < 16; k3++) { d.r[r + G3 * k3] = o.r[k3]; d.i[r + G3 * k3] = o.i[k3]; }
        }
    }
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

template <int LOG2N>
__global__ __launch_bounds__(C2_THREADS) void k_coarse2(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                           int nlist, const double2 *__restrict__ tw)
{
    constexpr int N = 1 << LOG2N;
    constexpr int E = N / C2_THREADS;
    extern __shared__ __attribute__((aligned(16))) double xch[];
    __shared__ double red_val[C2_THREADS];
    __shared__ int red_idx[C2_THREADS];
    __shared__ int sh_bigchange;
    const int t0 = threadIdx.x;
    const int nchp = g.nchp;

    CV<E> d;
    for (int li = blockIdx.x; li < nlist; li += gridDim.x)
    {
        int t = t0; // opaque once per estimate: what is derived from it is recomputed (1-2 instructions), not hoisted and spilled
        asm volatile("" : "+v"(t));
        const int ch = chan_list ? chan_list[li] : li;
        const double2 *__restrict__ ring = p.bbring + (size_t)ch * N;
        const int bb_ptr = p.I[(size_t)I_BB_PTR * nchp + ch];
        const double lockingbw = p.S[(size_t)S_LOCKINGBW * nchp + ch];
        const double hzperbin = g.Fs / ((double)N);
        const int startbin = (int)fmax(round(lockingbw / hzperbin), 1.0);
        const int stopbin = N - startbin;
        const int expectedpeakbin = (int)round(g.fb / (2.0 * hzperbin));
        double *__restrict__ y = p.y + (size_t)ch * N;

        // bbtmpbuff[j] = bbcycbuff[(ptr+j)%N] (time order); for every list entry but the first, these loads were issued
        // while the previous estimate was in its peak search / state machine (d is free there), hiding the HBM latency
        if (li == (int)blockIdx.x)
        {
#pragma unroll
            for (int s = 0; s < E; s++)
            {
                const double2 v = ring[(bb_ptr + s * C2_THREADS + t) & (N - 1)];
                d.r[s] = v.x; d.i[s] = v.y;
            }
        }
        wg_fft<LOG2N>(d, xch, tw, t);
        // band limit (fb != 8400 boxcar, coarsefreqestimate.cpp:99) then inverse transform = forward on swapped planes
#pragma unroll
        for (int s = 0; s < E; s++)
        {
            const int k = s * C2_THREADS + t;
            const bool z = (k >= startbin) && (k <= stopbin);
            const double re = z ? 0.0 : d.r[s], im = z ? 0.0 : d.i[s];
            d.r[s] = im; d.i[s] = re;
        }
        wg_fft<LOG2N>(d, xch, tw, t);
        // swap back (x N / N = 1), square
#pragma unroll
        for (int s = 0; s < E; s++)
        {
            const double re = d.i[s], im = d.r[s];
            d.r[s] = re * re - im * im;
            d.i[s] = re * im + im * re;
        }
        wg_fft<LOG2N>(d, xch, tw, t);
        __syncthreads(); // the exchange buffer is free: it receives a copy of y for the fold below
        // smooth with fftshift: y[i] = y[i]*0.9 + 0.1*10*log10(fmax(abs(out[i]),1)), out[i] = X[i ^ N/2]
        // all old y values are requested before the log10s (their registers: d.i, dead once only |X|^2 is kept): written as
        // one load-compute-store per element, every element waited out a full HBM round trip (vmcnt counts the stores too)
        {
            double yv[E];
#pragma unroll
            for (int s = 0; s < E; s++) d.r[s] = d.r[s] * d.r[s] + d.i[s] * d.i[s];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < E; s++) yv[s] = (y + ((s * C2_THREADS) ^ (N / 2)))[t]; // (s*512 + t) ^ N/2: uniform base + t
            __builtin_amdgcn_sched_barrier(0); // or the scheduler sinks every load to its use again
            // 10*log10(max(|X|,1)) == 5*log10(max(|X|^2,1)): no hypot; differs from the reference expression by <= 1 ulp
#pragma unroll
            for (int s = 0; s < E; s++) d.r[s] = 5.0 * c2_log10(fmax(d.r[s], 1.0));
#pragma unroll
            for (int s = 0; s < E; s++)
            {
                const int ib = (s * C2_THREADS) ^ (N / 2);
                const double yn = yv[s] * 0.9 + d.r[s];
                (y + ib)[t] = yn;
                (xch + ib)[t] = yn;
            }
        }
        __syncthreads();
        {
            const int ln = li + (int)gridDim.x;
            if (ln < nlist)
            {
                const int chn = chan_list ? chan_list[ln] : ln;
                const double2 *__restrict__ ringn = p.bbring + (size_t)chn * N;
                const int bpn = p.I[(size_t)I_BB_PTR * nchp + chn];
#pragma unroll
                for (int s = 0; s < E; s++)
                {
                    const double2 v = ringn[(bpn + s * C2_THREADS + t) & (N - 1)];
                    d.r[s] = v.x; d.i[s] = v.y;
                }
            }
        }

        // fold + peak search (:116-131)
        const int i0 = (int)round((-lockingbw / hzperbin) + ((double)(N / 2)));
        const int i1 = (int)round((lockingbw / hzperbin) + ((double)(N / 2)));
        double best = 0;
        int besti = -1;
        for (int i = i0 + t; i < i1; i += C2_THREADS)
        {
            if ((i < 0) || (i >= N)) continue;
            double val = 0;
            for (int j = -1; j <= 1; j++)
            {
                if (((i - expectedpeakbin - j) < 0) || ((i + expectedpeakbin + j) >= N)) continue;
                val += (xch[i - expectedpeakbin - j] + xch[i + expectedpeakbin + j]);
            }
            if (val > best) { best = val; besti = i; }
        }
        red_val[t] = best;
        red_idx[t] = besti;
        __syncthreads();
        for (int s = C2_THREADS / 2; s > 0; s >>= 1)
        {
            if (t < s)
            {
                const double ov = red_val[t + s];
                const int oi = red_idx[t + s];
                const double mv = red_val[t];
                const int mi = red_idx[t];
                if (oi >= 0 && (mi < 0 || ov > mv || (ov == mv && oi < mi))) { red_val[t] = ov; red_idx[t] = oi; }
            }
            __syncthreads();
        }
        if (t == 0) sh_bigchange = coarse_slot(g, p, ch, (red_idx[0] >= 0) ? red_idx[0] : (N / 2), N, hzperbin, lockingbw);
        __syncthreads();
        if (sh_bigchange)
        {
            double2 *ringw = p.bbring + (size_t)ch * N;
            for (int i = t; i < N; i += C2_THREADS) { y[i] = 20; ringw[i] = make_double2(0.0, 0.0); }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// 2^14-point transform out of 16-point pieces only: 16384 = 16 x 16 x 16 x 4, four passes, natural order in AND out, all 512
// threads busy in every pass, never more than one 16-point FFT's worth of temporaries on top of the 32 points a thread holds
// (wg_fft<14> and the middle pass of wg_fft<13> run 32-point FFTs in registers -- 32 in, 32 out, 256 VGPRs -- and spill).
//   n = n1*1024 + n2*64 + n3*4 + n4,   k = k1 + 16*k2 + 256*k3 + 4096*k4      (n1, n2, n3, k1, k2, k3 < 16;  n4, k4 < 4)
//   pass 1: FFT16 over n1, x W_16384^(k1*(n mod 1024))     pass 2: FFT16 over n2, x W_1024^(k2*(n mod 64))
//   pass 3: FFT16 over n3, x W_64^(k3*n4)                  pass 4: radix-4 over n4
// A thread always holds 32 points = the 16 values of the digit being transformed x one more bit:
//   in     slot 2*n1+b   b = n2>>3          thread t  = (n2&7)*64 + n3*4 + n4                (= natural: n = slot*512 + t)
//   pass 2 slot 2*n2+c   c = k1&1           thread t' = (k1>>1)*64 + n3*4 + n4
//   pass 3 slot 2*n3+b1  b1 = n4>>1         thread t''= (n4&1)*256 + k2*16 + k1
//   pass 4 slot n4*8+h   h = k3>>1          thread    = (k3&1)*256 + k2*16 + k1    -> out slot k4*8+h  (= natural: k = slot*512 + t)
// The three exchanges go through LDS one plane at a time; index maps chosen so that a wavefront's 64 lanes always touch 64
// consecutive doubles, except the writes of exchange 2 (row stride 257 doubles: two lanes per 8-byte bank, the minimum).
// xch: 64*257 doubles.
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

__device__ __forceinline__ void wg_fft14_r16(CV<32> &d, double *xch, const double2 *__restrict__ tw, int t)
{
#pragma clang fp contract(fast)
    constexpr int S2 = 257;
    // ---- pass 1 ----
    const double2 tw1[2] = {tw[t], tw[512 + t]}; // both requested before the first butterfly
#pragma unroll
    for (int b = 0; b < 2; b++)
    {
        CV<16> in, out;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = d.r[2 * j + b]; in.i[j] = d.i[2 * j + b]; }
        regfft<16>(in, out);
        c4_twiddle16(out, tw1[b]);
#pragma unroll
        for (int j = 0; j < 16; j++) { d.r[2 * j + b] = out.r[j]; d.i[2 * j + b] = out.i[j]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    const double2 step2 = tw[16 * (t & 63)]; // pass 2's twiddle base: in flight during exchange 1
    // ---- exchange 1: L[(k1*16 + n2)*64 + r2] ----
    {
        const int rbase = (t >> 6) * 2048 + (t & 63); // reader: ((2*k1hi + c)*16 + n2)*64 + r2
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) xch[((s >> 1) * 16 + (s & 1) * 8) * 64 + t] = d.r[s];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) d.r[s] = xch[rbase + ((s & 1) * 16 + (s >> 1)) * 64];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) xch[((s >> 1) * 16 + (s & 1) * 8) * 64 + t] = d.i[s];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) d.i[s] = xch[rbase + ((s & 1) * 16 + (s >> 1)) * 64];
    }
    // ---- pass 2 ----
    {
        const double2 step = step2;
#pragma unroll
        for (int c = 0; c < 2; c++)
        {
            CV<16> in, out;
#pragma unroll
            for (int j = 0; j < 16; j++) { in.r[j] = d.r[2 * j + c]; in.i[j] = d.i[2 * j + c]; }
            regfft<16>(in, out);
            c4_twiddle16(out, step);
#pragma unroll
            for (int j = 0; j < 16; j++) { d.r[2 * j + c] = out.r[j]; d.i[2 * j + c] = out.i[j]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- exchange 2: L[r2*257 + k2*16 + k1] ----
    {
        const int wbase = (t & 63) * S2 + (t >> 6) * 2; // writer: r2*257 + k2*16 + 2*k1hi + c
        const int rbase = (t >> 8) * S2 + (t & 255);     // reader: (n3*4 + 2*b1 + b0)*257 + u
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) xch[wbase + (s >> 1) * 16 + (s & 1)] = d.r[s];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) d.r[s] = xch[rbase + ((s >> 1) * 4 + (s & 1) * 2) * S2];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) xch[wbase + (s >> 1) * 16 + (s & 1)] = d.i[s];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) d.i[s] = xch[rbase + ((s >> 1) * 4 + (s & 1) * 2) * S2];
    }
    // ---- pass 3: twiddle W_64^(k3*n4), n4 = 2*b1 + b0, b0 = t >> 8 ----
    {
        // wave-uniform (waves 0-3 / 4-7): as a scalar the twiddles below are selects between literals; as a per-lane
        // value they were 64 dependent table loads per transform, each waited for on the spot
        const bool b0 = __builtin_amdgcn_readfirstlane(t >> 8) != 0;
#pragma unroll
        for (int b1 = 0; b1 < 2; b1++)
        {
            CV<16> in, out;
#pragma unroll
            for (int j = 0; j < 16; j++) { in.r[j] = d.r[2 * j + b1]; in.i[j] = d.i[2 * j + b1]; }
            regfft<16>(in, out);
#pragma unroll
            for (int k3 = 0; k3 < 16; k3++)
            {
                const int e0 = (k3 * (2 * b1)) & 63, e1 = (k3 * (2 * b1 + 1)) & 63;
                const double wr = b0 ? jd_w64r(e1) : jd_w64r(e0), wi = b0 ? jd_w64i(e1) : jd_w64i(e0);
                d.r[2 * k3 + b1] = out.r[k3] * wr - out.i[k3] * wi;
                d.i[2 * k3 + b1] = out.r[k3] * wi + out.i[k3] * wr;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- exchange 3: L[(k3*4 + n4)*256 + u] ----
    {
        const int u = t & 255, q = t >> 8; // writer: n4 = 2*b1 + q; reader: k3 = 2*h + q
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) xch[((s >> 1) * 4 + (s & 1) * 2 + q) * 256 + u] = d.r[s];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) d.r[s] = xch[((2 * (s & 7) + q) * 4 + (s >> 3)) * 256 + u];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) xch[((s >> 1) * 4 + (s & 1) * 2 + q) * 256 + u] = d.i[s];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; s++) d.i[s] = xch[((2 * (s & 7) + q) * 4 + (s >> 3)) * 256 + u];
    }
    // ---- pass 4: radix-4 over n4 (slots h, h+8, h+16, h+24) ----
#pragma unroll
    for (int h = 0; h < 8; h++)
    {
        const double ar = d.r[h], ai = d.i[h], br = d.r[h + 8], bi = d.i[h + 8];
        const double cr = d.r[h + 16], ci = d.i[h + 16], er = d.r[h + 24], ei = d.i[h + 24];
        const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
        const double t2r = br + er, t2i = bi + ei;
        const double t3r = (bi - ei), t3i = -(br - er); // -i (b - e)
        d.r[h] = t0r + t2r; d.i[h] = t0i + t2i;
        d.r[h + 8] = t1r + t3r; d.i[h + 8] = t1i + t3i;
        d.r[h + 16] = t0r - t2r; d.i[h + 16] = t0i - t2i;
        d.r[h + 24] = t1r - t3r; d.i[h + 24] = t1i - t3i;
    }
}

// one transform; the thread index is laundered per call so that nothing derived from it inside is shared between the three calls
// of an estimate and kept live (spilled) across everything in between
__device__ __forceinline__ void c4_fft(CV<32> &d, double *xch, const double2 *__restrict__ tw, int t)
{
    int tt = t;
    asm volatile("" : "+v"(tt));
    wg_fft14_r16(d, xch, tw, tt);
}

// k_coarse2<14> with the radix-16 transform above, the per-estimate opaque thread index and the fold from LDS of k_coarse3.
// Measured (MI355X, 65536 estimates per launch): 20.2 ms (k_coarse3 24.7, k_coarse2<14> 28.5); 64 bytes of scratch per thread
// instead of ~500, i.e. the ~48 GB of spill traffic per launch are gone.  Then 14.9 ms with no scratch at all, once no load is
// waited for where it is issued (pass-3 twiddles as literals, table values and y[] requested ahead).
// W8400 (fb == 8400, k_pre8400.h): the band limit is the centre-weighted window of coarsefreqestimate.cpp:61-74,100
// instead of the boxcar of :99.
#define C4_TABN 3584 // W8400: window table entries kept in LDS behind the exchange buffer (28 KiB): lockingbw < 10.49 kHz
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for every global load and STORE in
// flight: behind the y[] update that is 32 stores per thread on their way to HBM, behind the ring prefetch 32 loads -- the prefetch
// was issued early precisely so that the peak search would run under it.
__device__ __forceinline__ void c4_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <bool W8400>
__device__ __forceinline__ void coarse4_body(const JGeom g, const JPtrs p, const int *__restrict__ chan_list, int nlist, const double2 *__restrict__ tw)
{
    constexpr int N = 1 << 14;
    constexpr int E = 32;
    extern __shared__ __attribute__((aligned(16))) double xch[];
    __shared__ double red_val[C2_THREADS / 64]; // one entry per wavefront
    __shared__ int red_idx[C2_THREADS / 64];
    __shared__ int sh_bigchange;
    const int t0 = threadIdx.x;
    const int nchp = g.nchp;
    int tab_startbin = -1; // W8400: the startbin the window table behind the exchange buffer was made for

    CV<E> d;
    for (int li = blockIdx.x; li < nlist; li += gridDim.x)
    {
        int t = t0; // opaque once per estimate: what derives from it is 1-2 instructions, but hoisted out of the persistent loop ~100 live registers
        asm volatile("" : "+v"(t));
        const int ch = chan_list ? chan_list[li] : li;
        const double2 *__restrict__ ring = p.bbring + (size_t)ch * N;
        const int bb_ptr = p.I[(size_t)I_BB_PTR * nchp + ch];
        const double lockingbw = p.S[(size_t)S_LOCKINGBW * nchp + ch];
        const double hzperbin = g.Fs / ((double)N);
        const int startbin = (int)fmax(round(lockingbw / hzperbin), 1.0);
        const int stopbin = N - startbin;
        const int expectedpeakbin = (int)round(g.fb / (2.0 * hzperbin));
        double *__restrict__ y = p.y + (size_t)ch * N;

        // bbtmpbuff[j] = bbcycbuff[(ptr+j)%N] (time order); for every list entry but the first, these loads were issued
        // while the previous estimate was in its peak search / state machine (d is free there), hiding the HBM latency
        if (li == (int)blockIdx.x)
        {
#pragma unroll
            for (int s = 0; s < E; s++)
            {
                const double2 v = ring[(bb_ptr + s * C2_THREADS + t) & (N - 1)];
                d.r[s] = v.x; d.i[s] = v.y;
            }
        }
        c4_fft(d, xch, tw, t);
        // band limit (fb != 8400 boxcar, coarsefreqestimate.cpp:99) then inverse transform = forward on swapped planes
        if constexpr (W8400)
        {
            // window[0] = 1, window[i] = window[N - i] = cos^2(pi/2 * i / startbin) for 1 <= i <= startbin, 0 elsewhere (:61-74).  Its
            // startbin + 1 distinct values come from a table in LDS: entry startbin + 1 = 0 stands for every bin the window zeroes.  The
            // table sits behind the exchange buffer and is rebuilt only when startbin changes (a persistent workgroup serves ~256
            // estimates, normally all with one locking bandwidth); a window wider than that space (lockingbw >= 10.49 kHz) is made per
            // estimate in the idle exchange buffer.  (Round 1 evaluated 32 cosines per thread and estimate and spilled; one table per
            // estimate cost 6 cosines per thread and three barriers: 17.9 ms per 65 536 estimates against 14.8 for the boxcar.)
            const bool persistent = startbin < C4_TABN - 1;
            double *wt = persistent ? xch + 64 * 257 : xch;
            if (!persistent || startbin != tab_startbin)
            {
                c4_lds_barrier();
                for (int i = t; i <= startbin + 1; i += C2_THREADS)
                {
                    const double c = cos(M_PI_2 * ((double)i) / ((double)startbin));
                    wt[i] = (i == 0) ? 1.0 : ((i <= startbin) ? c * c : 0.0);
                }
                c4_lds_barrier();
                if (persistent) tab_startbin = startbin;
            }
            // applied eight at a time as they are read: all 32 weights in registers beside the 32 points spill (and without the fence the
            // scheduler hoists all 32 LDS reads to the top, which is the same thing)
#pragma unroll
            for (int s0 = 0; s0 < E; s0 += 8)
            {
#pragma unroll
                for (int s = s0; s < s0 + 8; s++)
                {
                    const int k = s * C2_THREADS + t;
                    const int i = (k <= N / 2) ? k : N - k;
                    const double w = wt[i <= startbin ? i : startbin + 1];
                    const double re = d.r[s] * w, im = d.i[s] * w;
                    d.r[s] = im; d.i[s] = re;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!persistent) c4_lds_barrier(); // the next transform's exchanges reuse the buffer
        }
        else
        {
#pragma unroll
            for (int s = 0; s < E; s++)
            {
                const int k = s * C2_THREADS + t;
                const bool z = (k >= startbin) && (k <= stopbin);
                const double re = z ? 0.0 : d.r[s], im = z ? 0.0 : d.i[s];
                d.r[s] = im; d.i[s] = re;
            }
        }
        c4_fft(d, xch, tw, t);
        // swap back (x N / N = 1), square
#pragma unroll
        for (int s = 0; s < E; s++)
        {
            const double re = d.i[s], im = d.r[s];
            d.r[s] = re * re - im * im;
            d.i[s] = re * im + im * re;
        }
        c4_fft(d, xch, tw, t);
        c4_lds_barrier(); // the exchange buffer is free: it receives a copy of y for the fold below
        // smooth with fftshift: y[i] = y[i]*0.9 + 0.1*10*log10(fmax(abs(out[i]),1)), out[i] = X[i ^ N/2]
        // all 32 old y values are requested before the log10s (64 registers, free once only |X|^2 is kept of d): written as one
        // load-compute-store per element, every element waited out a full HBM round trip (vmcnt counts the stores too)
        {
            double yv[E];
#pragma unroll
            for (int s = 0; s < E; s++) d.r[s] = d.r[s] * d.r[s] + d.i[s] * d.i[s];
            __builtin_amdgcn_sched_barrier(0); // d.i is dead from here: its registers take the y values
#pragma unroll
            for (int s = 0; s < E; s++) yv[s] = (y + ((s * C2_THREADS) ^ (N / 2)))[t]; // (s*512 + t) ^ N/2: uniform base + t
            __builtin_amdgcn_sched_barrier(0); // or the scheduler sinks every load to its use again
            // 10*log10(max(|X|,1)) == 5*log10(max(|X|^2,1)): no hypot; differs from the reference expression by <= 1 ulp
#pragma unroll
            for (int s = 0; s < E; s++) d.r[s] = 5.0 * c2_log10(fmax(d.r[s], 1.0));
#pragma unroll
            for (int s = 0; s < E; s++)
            {
                const int ib = (s * C2_THREADS) ^ (N / 2);
                const double yn = yv[s] * 0.9 + d.r[s];
                (y + ib)[t] = yn;
                (xch + ib)[t] = yn;
            }
        }
        c4_lds_barrier(); // the fold reads the LDS copy; the stores to y[] drain in the background
        {
            const int ln = li + (int)gridDim.x;
            if (ln < nlist)
            {
                const int chn = chan_list ? chan_list[ln] : ln;
                const double2 *__restrict__ ringn = p.bbring + (size_t)chn * N;
                const int bpn = p.I[(size_t)I_BB_PTR * nchp + chn];
#pragma unroll
                for (int s = 0; s < E; s++)
                {
                    const double2 v = ringn[(bpn + s * C2_THREADS + t) & (N - 1)];
                    d.r[s] = v.x; d.i[s] = v.y;
                }
            }
        }

        // fold + peak search (:116-131)
        const int i0 = (int)round((-lockingbw / hzperbin) + ((double)(N / 2)));
        const int i1 = (int)round((lockingbw / hzperbin) + ((double)(N / 2)));
        double best = 0;
        int besti = -1;
        for (int i = i0 + t; i < i1; i += C2_THREADS)
        {
            if ((i < 0) || (i >= N)) continue;
            double val = 0;
            for (int j = -1; j <= 1; j++)
            {
                if (((i - expectedpeakbin - j) < 0) || ((i + expectedpeakbin + j) >= N)) continue;
                val += (xch[i - expectedpeakbin - j] + xch[i + expectedpeakbin + j]);
            }
            if (val > best) { best = val; besti = i; }
        }
        // first maximum over the workgroup (ties: the lower bin, as the reference's ascending scan keeps the first): wavefront
        // reduction through DPP-free shuffles, then one LDS round for the eight wavefront results -- no barrier drains the ring
        // prefetch that is in flight
        {
            double bv = best;
            int bi = besti;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
            {
                const double ov = __shfl_xor(bv, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            if ((t & 63) == 0) { red_val[t >> 6] = bv; red_idx[t >> 6] = bi; }
            c4_lds_barrier();
            if (t == 0)
            {
                for (int w = 1; w < C2_THREADS / 64; w++)
                {
                    const double ov = red_val[w];
                    const int oi = red_idx[w];
                    if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
                }
                red_idx[0] = bi;
            }
        }
        if (t == 0) sh_bigchange = coarse_slot(g, p, ch, (red_idx[0] >= 0) ? red_idx[0] : (N / 2), N, hzperbin, lockingbw);
        c4_lds_barrier();
        if (sh_bigchange)
        {
            double2 *ringw = p.bbring + (size_t)ch * N;
            for (int i = t; i < N; i += C2_THREADS) { y[i] = 20; ringw[i] = make_double2(0.0, 0.0); }
        }
        c4_lds_barrier(); // LDS reuse only: the next estimate is another channel, and its ring rows are already on their way
    }
}

__global__ __launch_bounds__(C2_THREADS) void k_coarse4(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                           int nlist, const double2 *__restrict__ tw)
{
    coarse4_body<false>(g, p, chan_list, nlist, tw);
}
__global__ __launch_bounds__(C2_THREADS) void k_coarse4_w8400(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                                 int nlist, const double2 *__restrict__ tw)
{
    coarse4_body<true>(g, p, chan_list, nlist, tw);
}


