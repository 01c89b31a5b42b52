// k_coarse4.h -- the 2^14-point coarse-frequency kernel of rounds 1-2 (one stream, four barriers per exchange), kept OUTSIDE the product
// library as the A/B partner of k_coarse5 in scripts/ubench/coarse_bench.hip (k_coarse5.h replaced it in libjaero_hip.so in round 3).
#pragma once
#include "../../jaero_amd/csrc/k_coarse2.h"

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


