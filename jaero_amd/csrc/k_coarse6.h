// k_coarse6.h -- the 2^14-point coarse frequency estimator with TWO LDS exchanges per transform: 16384 = 32 x 32 x 16 (round 3).
//
// Same function as k_coarse5 / k_coarse4 (CoarseFreqEstimate::ProcessBasebandData + FreqOffsetEstimateSlot,
// JAERO/coarsefreqestimate.cpp:90-137, JAERO/oqpskdemodulator.cpp:629-677).  What round 3 measured: a transform's
// LDS traffic and its arithmetic do not overlap on this CU however they are arranged, so the only way left to shorten an estimate is to
// move fewer bytes through LDS.  16 x 16 x 16 x 4 needs three exchanges of all 32 points a thread holds; 32 x 32 x 16 needs two:
//
//   n = 512 n1 + 16 n2 + n3        k = k1 + 32 k2 + 1024 k3        (n1, n2, k1, k2 < 32;  n3, k3 < 16)
//   pass 1: FFT32 over n1, x W_N^(k1 (n mod 512))      exchange 1      pass 2: FFT32 over n2, x W_512^(k2 n3)      exchange 2
//   pass 3: two FFT16 over n3
//
// natural order on entry AND exit (slot = index >> 9, thread = index & 511), so ring / y[] accesses are whole 512-byte rows and the three
// transforms chain in registers.  The 32-point FFTs that made the first kernel of this shape (k_coarse2<14>) spill 359 registers are done
// in place: one radix-2 stage, then a 16-point FFT on each half; their outputs stay in SPLIT order (slot j < 16 = X[2j], slot 16 + j =
// X[2j + 1]), which only changes compile-time addresses of the exchange behind them.  Twiddles: even powers by a chain of products with
// step^2, each odd power one product more -- two live values instead of a table of 31.  Index maps, twiddles and bank behaviour:
// tests/test_coarse_fft14_e32_model.py.
#pragma once
#include "k_coarse2.h"

#ifndef C4_TABN
#define C4_TABN 3584 // W8400: window table entries kept in LDS behind the exchange buffer (28 KiB): lockingbw < 10.49 kHz
#endif
__device__ __forceinline__ void c6_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define C6_FENCE __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ double2 c6_sq(const double2 a)
{
#pragma clang fp contract(fast)
    return make_double2(a.x * a.x - a.y * a.y, 2.0 * (a.x * a.y));
}

#ifndef C6_TRACE
#define C6_TRACE(i) // scripts/ubench/coarse_trace.hip defines it to record a clock per phase
#endif

// (value, bin) of the wavefront's first maximum in every lane: value descending, bin ascending, bin < 0 = no candidate.  The order is total,
// so any reduction tree gives the reference's ascending scan's answer.  Six DPP steps (quad permutes, half-row / row mirrors, the two
// row broadcasts of gfx9), then lane 63 holds the result.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void c6_argmax_step(double &bv, int &bi)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(bv), __double2loint(bv), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(bv), __double2hiint(bv), CTRL, ROWMASK, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, ROWMASK, 0xf, false);
    const double ov = __hiloint2double(hi, lo);
    const bool take = (oi >= 0) & ((bi < 0) | (ov > bv) | ((ov == bv) & (oi < bi)));
    bv = take ? ov : bv;
    bi = take ? oi : bi;
}
__device__ __forceinline__ void c6_wave_argmax(double &bv, int &bi)
{
    c6_argmax_step<0xB1, 0xf>(bv, bi);  // quad_perm [1,0,3,2]
    c6_argmax_step<0x4E, 0xf>(bv, bi);  // quad_perm [2,3,0,1]
    c6_argmax_step<0x141, 0xf>(bv, bi); // row_half_mirror
    c6_argmax_step<0x140, 0xf>(bv, bi); // row_mirror: every lane of a row of 16 holds the row's result
    c6_argmax_step<0x142, 0xa>(bv, bi); // row_bcast:15 into rows 1 and 3
    c6_argmax_step<0x143, 0xc>(bv, bi); // row_bcast:31 into rows 2 and 3: lane 63 holds the wavefront's
    bv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bv), 63), __builtin_amdgcn_readlane(__double2loint(bv), 63));
    bi = __builtin_amdgcn_readlane(bi, 63);
}

#define C6_XCH 16448 // doubles: one plane of the whole transform (exchange 2: 16 rows at stride 513, + 8208 for the upper half of k2)

__device__ __forceinline__ constexpr int c6_k(int s) { return s < 16 ? 2 * s : 2 * (s - 16) + 1; }

// in-place forward 32-point DFT, natural order in, split order out
__device__ __forceinline__ void c6_fft32(CV<32> &x)
{
#pragma clang fp contract(fast)
#pragma unroll
    for (int j = 0; j < 16; j++)
    {
        const double ar = x.r[j], ai = x.i[j], br = x.r[j + 16], bi = x.i[j + 16];
        x.r[j] = ar + br; x.i[j] = ai + bi;
        double dr = ar - br, di = ai - bi;
        cmul_w64(dr, di, 2 * j); // W_32^j
        x.r[j + 16] = dr; x.i[j + 16] = di;
    }
    C6_FENCE;
#pragma unroll
    for (int h = 0; h < 2; h++)
    {
        CV<16> in, out;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = x.r[16 * h + j]; in.i[j] = x.i[16 * h + j]; }
        regfft<16>(in, out);
#pragma unroll
        for (int j = 0; j < 16; j++) { x.r[16 * h + j] = out.r[j]; x.i[16 * h + j] = out.i[j]; }
        C6_FENCE;
    }
}

// x[slot of k] *= p1^k for k = 1 .. 31 (split order): even powers e_j = p1^(2j) by a chain, each odd power one product more; three live
// values instead of a table of 31.  (Four interleaved chains of depth 5 instead of one of depth 15 measured no faster: 13.34 vs 13.19 ms.)
__device__ __forceinline__ void c6_twiddle32(CV<32> &x, const double2 p1)
{
#pragma clang fp contract(fast)
    auto app = [&](int slot, const double2 w) __attribute__((always_inline)) {
        const double r = x.r[slot] * w.x - x.i[slot] * w.y, i = x.r[slot] * w.y + x.i[slot] * w.x;
        x.r[slot] = r; x.i[slot] = i;
    };
    const double2 p2 = c6_sq(p1);
    double2 e = p2;
    app(16, p1);
#pragma unroll
    for (int j = 1; j < 16; j++)
    {
        app(j, e);
        app(16 + j, cmul2(e, p1));
        if (j < 15) e = cmul2(e, p2);
    }
}

// In-place forward 2^14-point DFT of the workgroup's data, natural distribution in and out.  xch: C6_XCH doubles.
__device__ __forceinline__ void wg_fft14_e32(CV<32> &d, double *xch, const double2 *__restrict__ tw, int t)
{
#pragma clang fp contract(fast)
    const double2 st1 = tw[t], st2 = tw[32 * (t & 15)]; // W_N^(n mod 512); W_512^n3 -- requested before the first butterfly
    const int k1u = t >> 4, n3 = t & 15, odd = k1u & 1;
    const int e1w0 = t, e1w1 = (t + 16) & 511;                  // exchange 1 writer: even / odd k1 rows (odd rows rotated by one n2 row)
    const int e1r = k1u * 512 + odd * 16 + n3;                  // reader: + 16 m, except the one row that wraps
    const int e1rw = e1r + 496 - 512 * odd;                     //   m = 31
    // exchange 2: row stride 513 -- a 64-bit LDS access is served 16 lanes at a time from 16 eight-byte bank pairs, and the 16 lanes of a
    // writer group differ in n3 only: an ODD stride spreads them over all 16 (514, chosen for a 32-lane rule, measured 403 M conflict cycles
    // per launch: SQ_LDS_BANK_CONFLICT)
    const int e2w = k1u + n3 * 513;                             // writer: + (k2 & 15) * 32 + (k2 >> 4) * 8208
    const int e2r = t;                                          // reader: + n3 * 513 + k2hi * 8208

    // ---- pass 1 ----
    c6_fft32(d);
    c6_twiddle32(d, st1);
    C6_FENCE;
    // ---- exchange 1, a plane at a time ----
    c6_bar(); // the buffer is free (previous transform's last reads / the fold)
#pragma unroll
    for (int s = 0; s < 32; s++) xch[c6_k(s) * 512 + ((c6_k(s) & 1) ? e1w1 : e1w0)] = d.r[s];
    c6_bar();
#pragma unroll
    for (int m = 0; m < 32; m++) d.r[m] = xch[(m < 31 ? e1r + 16 * m : e1rw)];
    c6_bar();
#pragma unroll
    for (int s = 0; s < 32; s++) xch[c6_k(s) * 512 + ((c6_k(s) & 1) ? e1w1 : e1w0)] = d.i[s];
    c6_bar();
#pragma unroll
    for (int m = 0; m < 32; m++) d.i[m] = xch[(m < 31 ? e1r + 16 * m : e1rw)];
    C6_FENCE;
    // ---- pass 2 ----
    c6_fft32(d);
    c6_twiddle32(d, st2);
    C6_FENCE;
    // ---- exchange 2 ----
    c6_bar();
#pragma unroll
    for (int s = 0; s < 32; s++) xch[e2w + (c6_k(s) & 15) * 32 + (c6_k(s) >> 4) * 8208] = d.r[s];
    c6_bar();
#pragma unroll
    for (int m = 0; m < 32; m++) d.r[m] = xch[e2r + (m & 15) * 513 + (m >> 4) * 8208]; // slot m = n3 + 16 k2hi
    c6_bar();
#pragma unroll
    for (int s = 0; s < 32; s++) xch[e2w + (c6_k(s) & 15) * 32 + (c6_k(s) >> 4) * 8208] = d.i[s];
    c6_bar();
#pragma unroll
    for (int m = 0; m < 32; m++) d.i[m] = xch[e2r + (m & 15) * 513 + (m >> 4) * 8208];
    C6_FENCE;
    // ---- pass 3: FFT16 over n3 for k2hi = 0, 1; X[.. + 1024 k3] -> slot 2 k3 + k2hi (= natural: k = slot * 512 + t) ----
    {
        CV<16> in, o0, o1;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = d.r[j]; in.i[j] = d.i[j]; }
        regfft<16>(in, o0);
        C6_FENCE;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = d.r[16 + j]; in.i[j] = d.i[16 + j]; }
        regfft<16>(in, o1);
#pragma unroll
        for (int k3 = 0; k3 < 16; k3++) { d.r[2 * k3] = o0.r[k3]; d.i[2 * k3] = o0.i[k3]; d.r[2 * k3 + 1] = o1.r[k3]; d.i[2 * k3 + 1] = o1.i[k3]; }
    }
}


// In-place forward 2^13-point DFT of a 256-thread workgroup's data, natural distribution in and out (slot = index >> 8, thread = index & 255):
// 8192 = 32 x 16 x 16, the same two exchanges.  Half the threads of k_coarse6 with the same 32 points each, so TWO workgroups share a CU
// (2 x 64 KiB of LDS, one wavefront of each per SIMD): while one is in an exchange or waits for HBM the other computes -- the overlap a
// single workgroup cannot have.  k_coarse2<13> (16 x 32 x 16 on 512 threads) left half of them idle in its 32-point pass.
//   n = 256 n1 + 16 n2 + n3        k = k1 + 32 k2 + 512 k3        (n1, k1 < 32;  n2, n3, k2, k3 < 16)
//   pass 1: FFT32 over n1, x W_N^(k1 (n mod 256))     exchange 1     pass 2: two FFT16 over n2 (k1 = k1a, k1a + 16), x W_256^(k2 n3)
//   exchange 2     pass 3: two FFT16 over n3 (k2 = k2lo, k2lo + 8)
// Exchange 1: L = k1 * 256 + 16 n2 + n3 (both sides touch consecutive doubles per 16 lanes).  Exchange 2: L = k1 + 32 k2 + 513 n3 (the 16
// lanes of a writer group differ in n3 only: the odd stride spreads them over the 16 bank pairs; readers touch 64 consecutive doubles).
#define C6_XCH13 8208
__device__ __forceinline__ void wg_fft13_e32(CV<32> &d, double *xch, const double2 *__restrict__ tw, int t)
{
#pragma clang fp contract(fast)
    const double2 st1 = tw[t], st2 = tw[32 * (t & 15)]; // W_N^(n mod 256); W_256^n3
    const int k1a = t >> 4, n3 = t & 15;
    const int e1r = k1a * 256 + n3;   // reader of exchange 1: + (m >> 4) * 4096 + (m & 15) * 16
    const int e2w = k1a + 513 * n3;   // writer of exchange 2: + 16 g + 32 k2
    // ---- pass 1 ----
    c6_fft32(d);
    c6_twiddle32(d, st1);
    C6_FENCE;
    // ---- exchange 1, a plane at a time ----
    c6_bar();
#pragma unroll
    for (int s = 0; s < 32; s++) (xch + c6_k(s) * 256)[t] = d.r[s];
    c6_bar();
#pragma unroll
    for (int m = 0; m < 32; m++) d.r[m] = (xch + (m >> 4) * 4096 + (m & 15) * 16)[e1r];
    c6_bar();
#pragma unroll
    for (int s = 0; s < 32; s++) (xch + c6_k(s) * 256)[t] = d.i[s];
    c6_bar();
#pragma unroll
    for (int m = 0; m < 32; m++) d.i[m] = (xch + (m >> 4) * 4096 + (m & 15) * 16)[e1r];
    C6_FENCE;
    // ---- pass 2: slots 0..15 = n2 for k1 = k1a, 16..31 for k1 = k1a + 16 ----
#pragma unroll
    for (int g2 = 0; g2 < 2; g2++)
    {
        CV<16> in, out;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = d.r[16 * g2 + j]; in.i[j] = d.i[16 * g2 + j]; }
        regfft<16>(in, out);
        c4_twiddle16(out, st2);
#pragma unroll
        for (int j = 0; j < 16; j++) { d.r[16 * g2 + j] = out.r[j]; d.i[16 * g2 + j] = out.i[j]; }
        C6_FENCE;
    }
    // ---- exchange 2: slot 16 g + k2 -> L = (k1a + 16 g) + 32 k2 + 513 n3; reader t3 = k1 + 32 k2lo, slot 16 h + n3 ----
    c6_bar();
#pragma unroll
    for (int s = 0; s < 32; s++) (xch + (s >> 4) * 16 + (s & 15) * 32)[e2w] = d.r[s];
    c6_bar();
#pragma unroll
    for (int m = 0; m < 32; m++) d.r[m] = (xch + (m >> 4) * 256 + (m & 15) * 513)[t];
    c6_bar();
#pragma unroll
    for (int s = 0; s < 32; s++) (xch + (s >> 4) * 16 + (s & 15) * 32)[e2w] = d.i[s];
    c6_bar();
#pragma unroll
    for (int m = 0; m < 32; m++) d.i[m] = (xch + (m >> 4) * 256 + (m & 15) * 513)[t];
    C6_FENCE;
    // ---- pass 3: FFT16 over n3 for h = 0, 1; X[t + 256 h + 512 k3] -> slot 2 k3 + h (natural) ----
    {
        CV<16> in, o0, o1;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = d.r[j]; in.i[j] = d.i[j]; }
        regfft<16>(in, o0);
        C6_FENCE;
#pragma unroll
        for (int j = 0; j < 16; j++) { in.r[j] = d.r[16 + j]; in.i[j] = d.i[16 + j]; }
        regfft<16>(in, o1);
#pragma unroll
        for (int k3 = 0; k3 < 16; k3++) { d.r[2 * k3] = o0.r[k3]; d.i[2 * k3] = o0.i[k3]; d.r[2 * k3 + 1] = o1.r[k3]; d.i[2 * k3 + 1] = o1.i[k3]; }
    }
}

template <int LOG2N>
__device__ __forceinline__ void c6_fft(CV<32> &d, double *xch, const double2 *__restrict__ tw, int t)
{
    int tt = t; // laundered per call: nothing derived from it inside is shared between the three calls of an estimate and kept live
    asm volatile("" : "+v"(tt));
    if constexpr (LOG2N == 14) wg_fft14_e32(d, xch, tw, tt);
    else wg_fft13_e32(d, xch, tw, tt);
}

// the 8400 bps window's startbin + 2 table entries (coarsefreqestimate.cpp:61-74).  Out of line on purpose: it runs when a channel's locking
// bandwidth changes, and inlined into the persistent estimate loop the cosine's constants were hoisted out of that loop and kept -- spilled --
// across every estimate's three transforms.
__device__ __noinline__ void c6_build_window(double *wt, int startbin, int t, int nthreads)
{
    for (int i = t; i <= startbin + 1; i += nthreads)
    {
        const double c = cos(M_PI_2 * ((double)i) / ((double)startbin));
        wt[i] = (i == 0) ? 1.0 : ((i <= startbin) ? c * c : 0.0);
    }
}

template <bool W8400, int LOG2N = 14>
__device__ __forceinline__ void coarse6_body(const JGeom g, const JPtrs p, const int *__restrict__ chan_list, int nlist, const double2 *__restrict__ tw)
{
    constexpr int N = 1 << LOG2N;
    constexpr int E = 32;
    constexpr int NT = N / E;                                  // 512 threads for 2^14, 256 for 2^13
    constexpr int XCH = LOG2N == 14 ? C6_XCH : C6_XCH13;
    extern __shared__ __attribute__((aligned(16))) double xch[];
    __shared__ double red_val[NT / 64]; // one entry per wavefront
    __shared__ int red_idx[NT / 64];
    const int t0 = threadIdx.x;
    const int nchp = g.nchp;
    int tab_startbin = -1; // W8400: the startbin the window table behind the exchange buffer was made for

    CV<E> d;
    int ch_next = ((int)blockIdx.x < nlist) ? (chan_list ? jd_sload(chan_list + blockIdx.x) : (int)blockIdx.x) : 0;
    int bp_next = jd_sload(p.I + (size_t)I_BB_PTR * nchp + ch_next);
    for (int li = blockIdx.x; li < nlist; li += gridDim.x)
    {
        int t = t0; // opaque once per estimate (what derives from it is 1-2 instructions; hoisted out of the persistent loop, ~100 live registers)
        asm volatile("" : "+v"(t));
        const int ch = ch_next, bb_ptr = bp_next;
        const double2 *__restrict__ ring = p.bbring + (size_t)ch * N;
        // the next estimate's channel and ring position: scalar loads (jaero_device.h), requested a whole estimate before the prefetch that
        // needs them
        const int ln = li + (int)gridDim.x;
        const bool has_next = ln < nlist;
        if (has_next)
        {
            ch_next = chan_list ? jd_sload(chan_list + ln) : ln;
            bp_next = jd_sload(p.I + (size_t)I_BB_PTR * nchp + ch_next);
        }
        const double lockingbw = jd_sload(p.S + (size_t)S_LOCKINGBW * nchp + ch);
        double fs_l = g.Fs; // opaque per estimate, as t: hoisted out of the loop Fs / N would be kept (spilled) across it, and a reload's wait
        asm volatile("" : "+s"(fs_l)); // stands behind every vector load in flight (vmcnt counts in order)
        // wave-uniform values that vector instructions compute (there is no scalar fp64): the integers go to scalar registers at once, the
        // doubles are formed again where the epilogue needs them -- left in vector registers they stay live across the three transforms
        // (k_coarse6_w8400 spilled 17 of them until round 4, and a reload's wait stands behind every load in flight)
        int startbin, expectedpeakbin;
        {
            const double hzperbin = fs_l * (1.0 / ((double)N)); // N a power of two: the same bits as the reference's quotient
            startbin = __builtin_amdgcn_readfirstlane((int)fmax(round(lockingbw / hzperbin), 1.0));
            expectedpeakbin = __builtin_amdgcn_readfirstlane((int)round(g.fb / (2.0 * hzperbin)));
        }
        const int stopbin = N - startbin;
        double *__restrict__ y = p.y + (size_t)ch * N;

        // bbtmpbuff[j] = bbcycbuff[(ptr+j)%N] (time order); for every list entry but the first these loads were issued while the
        // previous estimate was in its peak search / state machine
        if (li == (int)blockIdx.x)
        {
#pragma unroll
            for (int s = 0; s < E; s++)
            {
                const double2 v = ring[(bb_ptr + s * NT + t) & (N - 1)];
                d.r[s] = v.x; d.i[s] = v.y;
            }
        }
        C6_TRACE(0);
        c6_fft<LOG2N>(d, xch, tw, t);
        C6_TRACE(1);
        // band limit (fb != 8400 boxcar, coarsefreqestimate.cpp:99) then inverse transform = forward on swapped planes
        if constexpr (W8400)
        {
            // window[0] = 1, window[i] = window[N - i] = cos^2(pi/2 * i / startbin) for 1 <= i <= startbin, 0 elsewhere (:61-74).  Its
            // startbin + 1 distinct values come from a table in LDS behind the exchange buffer (entry startbin + 1 = 0 stands for every bin
            // the window zeroes), rebuilt only when startbin changes; a window wider than that space (lockingbw >= 10.49 kHz) is made per
            // estimate in the idle exchange buffer.
            const bool persistent = startbin < C4_TABN - 1;
            double *wt = persistent ? xch + XCH : xch;
            if (!persistent || startbin != tab_startbin)
            {
                c6_bar();
                c6_build_window(wt, startbin, t, NT);
                c6_bar();
                if (persistent) tab_startbin = startbin; // (a scalar register: startbin is one)
            }
#pragma unroll
            for (int s0 = 0; s0 < E; s0 += 8)
            {
#pragma unroll
                for (int s = s0; s < s0 + 8; s++)
                {
                    const int k = s * NT + t;
                    const int i = (k <= N / 2) ? k : N - k;
                    const double w = wt[i <= startbin ? i : startbin + 1];
                    const double re = d.r[s] * w, im = d.i[s] * w;
                    d.r[s] = im; d.i[s] = re;
                }
                C6_FENCE;
            }
            if (!persistent) c6_bar(); // the next transform's exchanges reuse the buffer
        }
        else
        {
#pragma unroll
            for (int s = 0; s < E; s++)
            {
                const int k = s * NT + t;
                const bool z = (k >= startbin) && (k <= stopbin);
                const double re = z ? 0.0 : d.r[s], im = z ? 0.0 : d.i[s];
                d.r[s] = im; d.i[s] = re;
            }
        }
        c6_fft<LOG2N>(d, xch, tw, t);
        C6_TRACE(2);
        // swap back (x N / N = 1), square
#pragma unroll
        for (int s = 0; s < E; s++)
        {
            const double re = d.i[s], im = d.r[s];
            d.r[s] = re * re - im * im;
            d.i[s] = re * im + im * re;
        }
        c6_fft<LOG2N>(d, xch, tw, t);
        C6_TRACE(3);
        // ---- epilogue.  Round 4 timed it phase by phase (scripts/ubench/coarse_trace.hip): 21 of an estimate's 52 us, most of it latency
        // chains with the whole workgroup waiting -- the fold's range checks as branches (three LDS round trips per candidate bin: 4.2 us), a
        // wavefront reduction through ds_bpermute (0.9 us), thread 0 alone loading the channel's state and evaluating the slot between two
        // barriers (3 us) -- and the rest the CU's own memory phase: a CU gets ~45 GB/s out of loads that miss L2 (the 45 loads behind the y
        // stores take a wavefront 7-8 us to issue), and no registers are free to request any of it a transform earlier (the transform needs
        // 196 of 256; holding 20 y values across it made it slower than the wait they save, touching the lines with one-dword loads cost more
        // issue time than it saved).  What is here now: 13.2 -> 12.4 ms per 65 536 estimates.
        c6_bar(); // the exchange buffer is free: it receives a copy of y for the fold below
        // smooth with fftshift: y[i] = y[i]*0.9 + 0.1*10*log10(fmax(abs(out[i]),1)), out[i] = X[i ^ N/2]
        // all 32 old y values are requested before the log10s (their registers: the imaginary plane, dead once only |X|^2 is kept); y and the
        // ring are streamed (read once, written once per estimate): non-temporal accesses
        CoarseSlotState cst;
        {
            // One pass per slot: log10, smooth, store, LDS copy -- and, into the four registers that frees, the NEXT estimate's ring entry of
            // that slot.  Requested in one burst behind the barrier, the 32 ring loads (and the 32 y stores in front of them) stalled every
            // wavefront of the workgroup at the same place for 7 us (a CU issues ~45 GB/s of loads that miss L2); spread over the logarithms,
            // one wavefront of a SIMD waits at a load while the other computes (12.6 -> 12.4 ms per 65 536 estimates).
            double yv[E];
#pragma unroll
            for (int s = 0; s < E; s++) d.r[s] = d.r[s] * d.r[s] + d.i[s] * d.i[s];
            C6_FENCE; // d.i is dead from here: its registers take the y values
#pragma unroll
            for (int s = 0; s < E; s++) yv[s] = __builtin_nontemporal_load((y + ((s * NT) ^ (N / 2))) + t); // (s*NT + t) ^ N/2: uniform base + t
            C6_FENCE; // or the scheduler sinks every load to its use again
            // the channel's acquisition state, needed behind the peak search: in front of everything else that is requested below (vmcnt
            // retires in order), as ordinary loads
            cst = coarse_slot_load_v(g, p, ch);
            // laundered: known since the top of the estimate, the 32 ring addresses would otherwise be computed there and kept (spilled) across
            // the three transforms -- and every reload waits for all vector loads in flight
            int bpn = bp_next, chn = ch_next, tp = t;
            asm volatile("" : "+v"(bpn), "+v"(chn), "+v"(tp));
            typedef double c6_v2 __attribute__((ext_vector_type(2)));
            const c6_v2 *__restrict__ ringn = (const c6_v2 *)(p.bbring + (size_t)chn * N);
            const int toffp = bpn + tp;
#pragma unroll
            for (int s = 0; s < E; s++)
            {
                const int ib = (s * NT) ^ (N / 2);
                // 10*log10(max(|X|,1)) == 5*log10(max(|X|^2,1)): no hypot; differs from the reference expression by <= 1 ulp
                const double yn = yv[s] * 0.9 + 5.0 * c2_log10(fmax(d.r[s], 1.0));
                __builtin_nontemporal_store(yn, (y + ib) + t);
                (xch + ib)[t] = yn;
                {
                    // unconditional: without a next estimate ch_next / bp_next still name this one (valid memory, result unused) -- under
                    // `if (has_next)` the old contents of d.r[s] had to stay live beside the new ones (151 spilled registers)
                    const c6_v2 v = __builtin_nontemporal_load(ringn + ((toffp + s * NT) & (N - 1)));
                    d.r[s] = v.x; d.i[s] = v.y;
                }
                if ((s & 3) == 3) C6_FENCE; // four slots at a time: the scheduler may not gather the loads at either end again
            }
            C6_TRACE(4);
            c6_bar(); // the fold reads the LDS copy; the stores to y[] drain in the background
            C6_TRACE(5);
        }
        C6_TRACE(6);
        // fold + peak search (:116-131)
        double fs_e = g.Fs; // opaque again: the value above must not be kept for this
        asm volatile("" : "+s"(fs_e));
        const double hzperbin = fs_e * (1.0 / ((double)N));
        const int i0 = __builtin_amdgcn_readfirstlane((int)round((-lockingbw / hzperbin) + ((double)(N / 2))));
        const int i1 = __builtin_amdgcn_readfirstlane((int)round((lockingbw / hzperbin) + ((double)(N / 2))));
        double best = 0;
        int besti = -1;
        // every index the fold touches lies inside the spectrum (always, unless lockingbw + fb/2 reaches Fs/2): no per-term range checks
        const bool fold_inside = (i0 - expectedpeakbin - 1 >= 0) && (i1 + expectedpeakbin < N) && (i0 >= 0);
        if (fold_inside)
        {
            // four candidate bins at a time, straight-line: their 24 LDS reads are requested together
            const int nblk = (i1 - i0 + 4 * NT - 1) / (4 * NT);
            for (int b = 0; b < nblk; b++)
            {
                double val[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int i = i0 + t + (4 * b + u) * NT;
                    const int ic = i < i1 ? i : i1 - 1;
                    const double *lo = xch + (ic - expectedpeakbin), *hi = xch + (ic + expectedpeakbin);
                    double v = 0;
                    v += (lo[1] + hi[-1]);  // j = -1
                    v += (lo[0] + hi[0]);   // j = 0
                    v += (lo[-1] + hi[1]);  // j = 1
                    val[u] = v;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int i = i0 + t + (4 * b + u) * NT;
                    const bool take = (i < i1) & (val[u] > best);
                    best = take ? val[u] : best;
                    besti = take ? i : besti;
                }
            }
        }
        else
        {
            for (int i = i0 + t; i < i1; i += NT)
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
        }
        C6_TRACE(7);
        // first maximum over the workgroup (ties: the lower bin, as the reference's ascending scan keeps the first): DPP moves inside a
        // wavefront, one LDS round for the wavefronts' results, and then EVERY thread finishes the reduction and evaluates the slot itself
        // (thread 0 writes): no single-thread section with the workgroup waiting at a second barrier behind it, no flag to broadcast, and no
        // barrier that would drain the ring prefetch in flight
        int bigchange;
        {
            double bv = best;
            int bi = besti;
            c6_wave_argmax(bv, bi);
            if ((t & 63) == 0) { red_val[t >> 6] = bv; red_idx[t >> 6] = bi; }
            C6_TRACE(8);
            c6_bar();
            bv = red_val[0]; bi = red_idx[0];
#pragma unroll
            for (int w = 1; w < NT / 64; w++)
            {
                const double ov = red_val[w];
                const int oi = red_idx[w];
                const bool take = (oi >= 0) & ((bi < 0) | (ov > bv) | ((ov == bv) & (oi < bi)));
                bv = take ? ov : bv;
                bi = take ? oi : bi;
            }
            bigchange = coarse_slot_apply(g, p, ch, cst, (bi >= 0) ? bi : (N / 2), N, hzperbin, lockingbw, t == 0);
            C6_TRACE(9);
        }
        if (bigchange)
        {
            __syncthreads(); // rare (AFC recentre): this estimate's y stores must have landed before other threads overwrite the same rows
            double2 *ringw = p.bbring + (size_t)ch * N;
            for (int i = t; i < N; i += NT) { y[i] = 20; ringw[i] = make_double2(0.0, 0.0); }
        }
        C6_TRACE(10);
        // no barrier here: the next use of LDS is behind the first barrier of the next estimate's transform
    }
}

__global__ __launch_bounds__(C2_THREADS) void k_coarse6(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                           int nlist, const double2 *__restrict__ tw)
{
    coarse6_body<false>(g, p, chan_list, nlist, tw);
}
__global__ __launch_bounds__(C2_THREADS) void k_coarse6_w8400(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                                 int nlist, const double2 *__restrict__ tw)
{
    coarse6_body<true>(g, p, chan_list, nlist, tw);
}
// N = 2^13 (the MSK rates): 256 threads, two workgroups per CU (launch 2 x #CUs workgroups, C6_XCH13 doubles of dynamic LDS each)
__global__ __launch_bounds__(256, 2) void k_coarse6_13(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                        int nlist, const double2 *__restrict__ tw)
{
    coarse6_body<false, 13>(g, p, chan_list, nlist, tw);
}
