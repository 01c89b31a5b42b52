// k_pre8400.h -- prefilter of the 8400 bps C-channel branch of OqpskDemodulator::writeData.
//
// Written from the oracle restatement (oracle/jaero_oracle.c, fb == 8400); on an MI355X it reproduces the two reference goldens
// (every soft byte, estimate frequencies to 5e-12) and, in banks of 5 and 67 channels, the oracle (tests/test_gpu_parity.py).  Not
// measured or tuned yet: the prefilter costs about as much as the rest of the path, and k_coarse4_w8400 spills.
//
// Reference (JAERO/oqpskdemodulator.cpp:343-381): for the whole write, PCM is mixed down with mixer_fir_pre, filtered with
// JFastFir (kernel RRC alpha 0.6, 2049 taps, nfft 4096: out[m] = sum_k h[k] x[m - L - k], L = nfft - K + 1 = 2048, the behaviour
// JAERO/tests/jfastfir_tests.cpp pins), and mixed up again with the conjugate of the same oscillator restarted from the phase it
// had before the write.  At the end of every write the oscillator's frequency becomes the mean of mixer2's over that write
// (:607-608).  Here:
//   k_pre8400_mix  one lane per channel, sequential over the write (the oscillator phase is a running sum): down-mixed samples go
//                  into a per-channel history ring xring[slot][channel], the up-mix table index of every sample into cidx
//   k_pre8400_fft  the filter by overlap-save on that ring, then the up-mix (below); its round-off differs from the reference's FFT
//                  overlap-add (other transform, other order): the soft-symbol tolerance of the north star applies, as for the Hilbert
//                  filter of the burst path.  (The direct form k_pre8400_fir, 127 ms per step, left the library in round 3.)
#pragma once
#include "jaero_device.h"
#include "k_coarse2.h" // CV<>, regfft<16>, c4_twiddle16, c4_lds_barrier

#define PRE_K 2049                 // taps
#define PRE_L 2048                 // JFastFir latency nfft - K + 1

struct JPre
{
    double2 *xring;        // [nchp / 4][ring][4] down-mixed history, slot = absolute sample index & (ring - 1): the four channels of a k_pre8400_fft workgroup are one
                           // contiguous 64-byte row per slot and one contiguous stretch per window (round 6; [ring][nchp] until then: see PRE_XI)
    unsigned short *cidx;  // [nchp / 4][cap][4] table index of the up-mix oscillator for every sample of the current write
    double2 *out;          // [nchp / 4][cap][4] prefiltered samples of the current write (cval_prefiltered); k_oqpsk_fb<.., PRE8400> reads it with JD_G4
    int cap;               // rows of cidx / out = the bank's largest write
    const double *taps;    // [PRE_K]
    int ring;              // power of two >= max_write + 3 * PRE_L
    const double2 *H;      // [4096] DFT of the taps (zero-padded to 4096) / 4096, natural order        (k_pre8400_fft)
    const double2 *tw;     // [4096] exp(-2 pi i k / 4096)                                               (k_pre8400_fft)
    long long *hold;       // [nchp] absolute sample index up to which a channel's outputs are exact zeros: 2048 samples behind a
                           // setSettings of that channel alone (JFastFir::SetKernel queues L zeros in front of the new filter's output)
};

// index of (slot, channel) in JPre::xring.  As [ring][nchp] a transform workgroup's window was 4096 sectors a megabyte apart (65 536 channels x 16 bytes per
// slot): every one of them its own page for the address translation.  Group-major, a window is 256 KB in one piece; k_pre8400_mix's 64 lanes write 16
// sectors per sample, each of them the next one of its group's stream.
#define PRE_XI(slot, ch, ring) JD_G4(slot, ch, ring)

// fields of JPtrs::S used by the prefilter (appended to the state enum in jaero_device.h): S_PRE_PTR, S_PRE_STEP, S_PRE_FSUM

__global__ __launch_bounds__(64) void k_pre8400_mix(const JGeom g, const JPtrs p, const JPre q, const int16_t *__restrict__ pcm, int pcm_stride,
                                                    int n, long long n0, int nprev)
{
    const int lane = threadIdx.x, ch = blockIdx.x * 64 + lane, nchp = g.nchp;
    const bool live = ch < g.nch;
    // blockIdx.y = one of gridDim.y stretches of the write (round 6): with one wavefront per 64 channels and the whole write each, 1024 wavefronts streamed
    // 5.4 GB with eight requests in flight apiece (3.3 ms per 4096-sample step of 65 536 channels).  A stretch starts from the oscillators' state at its
    // first sample, which it gets by taking the steps before it without touching memory (a few thousand additions); every stretch reads the state and
    // none writes it -- the last one leaves the state after the whole write in S_PRE_*_NEXT, k_pre8400_commit (next on the stream) makes it current.
    double ptr = p.S[(size_t)S_PRE_PTR * nchp + ch], step = p.S[(size_t)S_PRE_STEP * nchp + ch];
    if (nprev > 0)
    {
        // mixer_fir_pre.SetFreq(mixer2_freq_sum / i) at the end of the previous write (WaveTable::SetFreq(double), DSP.cpp:151-156)
        double freq = p.S[(size_t)S_PRE_FSUM * nchp + ch] / ((double)nprev);
        if (freq < 0) freq = 0;
        step = (freq) * ((double)JD_WTSIZE) / 48000.0;
    }
    const int ntb = (int)gridDim.y, tb = (int)blockIdx.y;
    const int chunk = (((n + ntb - 1) / ntb) + 7) & ~7;
    const int lo = min(n, tb * chunk), hi = (tb == ntb - 1) ? n : min(n, lo + chunk);
    const double2 *__restrict__ cis = p.cis;
    const int rmask = q.ring - 1;
    // down (:354-364) and up (:371-379: SetPhaseDeg(savedphase), then the same number of frames): two independent oscillators with
    // the same step, advanced in one loop (as two loops each waited for its own table gathers and stores: 3.6 ms per step)
    const double savedphase = (360.0 * ptr / ((double)JD_WTSIZE)); // GetPhaseDeg
    double phase = fmod(savedphase, 360.0);
    while (phase < 0) phase += 360.0;
    double pd = ptr, sd = step;
    double pu = (phase / 360.0) * ((double)JD_WTSIZE), su = step;
    // Eight samples at a time (round 6): the two oscillators advance by running sums that depend on nothing loaded, so the eight table indices of a
    // group are known before any of its table values or PCM samples arrives.  One sample per iteration, each iteration waited for its own gather (an L2
    // round trip): 3.6 ms per 4096-sample step of 65 536 channels, nearly all of it latency.  Same operations per sample, in the same order.
    for (int k = 0; k < lo; k++) { jd_wt_next(pd, sd); jd_wt_next(pu, su); } // the steps of the stretches in front of this one
    int i = lo;
    for (; i + 8 <= hi; i += 8)
    {
        int id[8], iu[8];
        short sv[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            id[u] = jd_cisidx(pd); iu[u] = jd_cisidx(pu);
            jd_wt_next(pd, sd);
            jd_wt_next(pu, su);
            sv[u] = live ? pcm[(size_t)(i + u) * pcm_stride + ch] : (short)0;
        }
        double2 cv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) cv[u] = cis[id[u]];
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            const double dval = ((double)sv[u]) / 32768.0;
            q.cidx[PRE_XI(i + u, ch, q.cap)] = (unsigned short)iu[u];
            q.xring[PRE_XI((int)((n0 + i + u) & rmask), ch, q.ring)] = make_double2(cv[u].x * dval, cv[u].y * dval);
        }
    }
    for (; i < hi; i++)
    {
        const short s = live ? pcm[(size_t)i * pcm_stride + ch] : (short)0;
        const double dval = ((double)s) / 32768.0;
        const double2 c = cis[jd_cisidx(pd)];
        q.cidx[PRE_XI(i, ch, q.cap)] = (unsigned short)jd_cisidx(pu);
        q.xring[PRE_XI((int)((n0 + i) & rmask), ch, q.ring)] = make_double2(c.x * dval, c.y * dval);
        jd_wt_next(pd, sd);
        jd_wt_next(pu, su);
    }
    if (tb == ntb - 1)
    {
        p.S[(size_t)S_PRE_PTR_NEXT * nchp + ch] = pu;
        p.S[(size_t)S_PRE_STEP_NEXT * nchp + ch] = su;
    }
}
__global__ __launch_bounds__(64) void k_pre8400_commit(const JGeom g, const JPtrs p, int nprev)
{
    const int ch = blockIdx.x * 64 + threadIdx.x, nchp = g.nchp;
    p.S[(size_t)S_PRE_PTR * nchp + ch] = p.S[(size_t)S_PRE_PTR_NEXT * nchp + ch];
    p.S[(size_t)S_PRE_STEP * nchp + ch] = p.S[(size_t)S_PRE_STEP_NEXT * nchp + ch];
    if (nprev > 0) p.S[(size_t)S_PRE_FSUM * nchp + ch] = 0.0; // the sum over the write that is about to be demodulated starts here (:607-608)
}

// k_pre8400_fft: the prefilter by overlap-save, i.e. the way the reference's JFastFir computes it (nfft 4096, 2049
// taps, 2048 fresh outputs per transform pair) -- 4 x 4096-point transforms per channel and 4096-sample write instead of 2049 multiply-
// adds per output (k_pre8400_fir: 127 ms per 65 536-channel step, four times the rest of the path).
//
//   grid (nchp / 4, 2048-sample blocks the write touches), 1024 threads = 4 channels x 256 threads, 128 KiB LDS.  Thread (c = tid & 3, T = tid >> 2): the
//   four lanes c of a T touch the four channels' 16-byte entries of one ring slot / output row = one complete 64-byte sector.
//   Block b: window x[m0 - 4096 .. m0 - 1], m0 = 2048 (floor(n0 / 2048) + b); its circular convolution with h is the linear one at window indices
//   2048 .. 4095, which are out[m0 .. m0 + 2047] (latency 2048 = nfft - K + 1 as in JFastFir).
//   4096 = 16 x 16 x 16, 16 points per thread, thread T holds index T + 256 s in slot s on entry AND on exit of a transform:
//     pass 1  n = 256 n1 + n2 (n2 = T): 16-point DFT over n1 -> k1, times W_4096^(n2 k1)
//     exch 1  L = k1 * 256 + 16 m1 + m2          writer T = 16 m1 + m2 slot k1 -> reader T = 16 k1 + m2 slot m1
//     pass 2  16-point DFT over m1 -> q1, times W_256^(m2 q1)
//     exch 2  L = (q1 * 16 + m2) * 16 + (k1 ^ m2) writer T = 16 k1 + m2 slot q1 -> reader T = k1 + 16 q1 slot m2
//     pass 3  16-point DFT over m2 -> q2: X[k1 + 16 q1 + 256 q2] = X[T + 256 s]
//   LDS address = 4 L + c: a wavefront (16 T x 4 c) touches 64 consecutive doubles in every access but the writes of exchange 2, where
//   the swizzle k1 ^ m2 spreads each half wavefront over all banks.  Planes (re, im) go through the buffer one after the other.
//   The inverse transform is the forward one on the conjugate (1 / 4096 is folded into H).
//   Not bit-identical to the reference's FFT (JFFT is not vendored; other butterfly order): differences ~1e-16 of the signal peak, as
//   with the direct form; the soft-symbol tolerance applies.
#define PF_THREADS 1024

__device__ __forceinline__ void pf_exchange1(double (&v)[16], double *xch, int T, int c)
{
    const int wb = T * 4 + c, rb = ((T >> 4) * 256 + (T & 15)) * 4 + c;
#pragma unroll
    for (int s = 0; s < 16; s++) xch[wb + s * 1024] = v[s];
    c4_lds_barrier();
#pragma unroll
    for (int s = 0; s < 16; s++) v[s] = xch[rb + s * 64];
    c4_lds_barrier();
}

__device__ __forceinline__ void pf_exchange2(double (&v)[16], double *xch, int T, int c)
{
    const int hi = T >> 4, lo = T & 15;
    // writer: k1 = hi, m2 = lo, slot q1; reader: q1 = hi, k1 = lo, slot m2
    const int wb = (lo * 16 + (hi ^ lo)) * 4 + c;
#pragma unroll
    for (int s = 0; s < 16; s++) xch[wb + s * 1024] = v[s];
    c4_lds_barrier();
#pragma unroll
    for (int s = 0; s < 16; s++) v[s] = xch[((hi * 16 + s) * 16 + (lo ^ s)) * 4 + c];
    c4_lds_barrier();
}

__device__ __forceinline__ void pf_fft4096(CV<16> &d, double *xch, const double2 *__restrict__ tw, int T, int c)
{
#pragma clang fp contract(fast)
    const double2 st1 = tw[T], st2 = tw[16 * (T & 15)];
    CV<16> o;
    regfft<16>(d, o);
    c4_twiddle16(o, st1);
    pf_exchange1(o.r, xch, T, c);
    pf_exchange1(o.i, xch, T, c);
    regfft<16>(o, d);
    c4_twiddle16(d, st2);
    pf_exchange2(d.r, xch, T, c);
    pf_exchange2(d.i, xch, T, c);
    regfft<16>(d, o);
#pragma unroll
    for (int s = 0; s < 16; s++) { d.r[s] = o.r[s]; d.i[s] = o.i[s]; }
}

__global__ __launch_bounds__(PF_THREADS) void k_pre8400_fft(const JGeom g, const JPtrs p, const JPre q, int n, long long n0)
{
    extern __shared__ double pf_xch[]; // 4 channels x 4096 doubles
    const int tid = threadIdx.x, c = tid & 3, T = tid >> 2;
    const int nchp = g.nchp, ch = blockIdx.x * 4 + c;
    // transform blocks sit at absolute multiples of 2048 samples, as JFastFir's do whatever the write sizes are: an output is an exact
    // zero here exactly where it is one there (both of the 2048-sample input blocks behind it all zero -- the first 2048 outputs of a
    // stream, digital silence); round-off in place of those zeros is what the AGC behind the filter would amplify to full scale
    const long long m0 = ((n0 >> 11) + blockIdx.y) << 11;
    const int i0 = (int)(m0 - n0); // may be negative: the part of the block that belonged to the previous write is not stored
    const int rmask = q.ring - 1;
    const long long hold_until = q.hold[ch];
    CV<16> d;
    {
        const double2 *__restrict__ xr = q.xring;
        const long long b = m0 - 2 * PRE_L + T;
#pragma unroll
        for (int s = 0; s < 16; s++)
        {
            const double2 v = xr[PRE_XI((int)((b + 256 * s) & rmask), ch, q.ring)];
            d.r[s] = v.x; d.i[s] = v.y;
        }
    }
    pf_fft4096(d, pf_xch, q.tw, T, c);
    {
        const double2 *__restrict__ H = q.H + T;
#pragma unroll
        for (int s = 0; s < 16; s++)
        {
            const double2 h = H[256 * s];
            const double yr = d.r[s] * h.x - d.i[s] * h.y, yi = d.r[s] * h.y + d.i[s] * h.x;
            d.r[s] = yr; d.i[s] = -yi;
        }
    }
    pf_fft4096(d, pf_xch, q.tw, T, c);
#pragma unroll
    for (int s = 8; s < 16; s++)
    {
        const int i = i0 + T + 256 * (s - 8);
        if (i >= 0 && i < n)
        {
            const bool held = n0 + i < hold_until;
            const double yr = held ? 0.0 : d.r[s], yi = held ? 0.0 : -d.i[s];
            // cval_prefiltered[i] *= mixer_fir_pre.WTCISValue_conj()
            const double2 cj = p.cis[q.cidx[PRE_XI(i, ch, q.cap)]];
            const double bre = cj.x, bim = -cj.y;
            q.out[PRE_XI(i, ch, q.cap)] = make_double2(yr * bre - yi * bim, yr * bim + yi * bre);
        }
    }
}

// setSettings on ONE channel of an 8400 bps bank (oqpskdemodulator.cpp:278-283: fir_pre.SetKernel): the channel's prefilter starts again --
// empty history, and JFastFir's L = 2048 queued zeros in front of its first output.  The transform blocks stay on the bank's grid (absolute
// multiples of 2048 samples; the reference object's own grid restarts at the call): the filtered values are the same linear convolution
// either way, up to transform round-off; what differs is at which samples round-off takes the place of an exact zero should that channel
// later carry digital silence.
__global__ void k_pre8400_restart(const JGeom g, const JPre q, int ch, long long now)
{
    const int nchp = g.nchp;
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < q.ring; slot += gridDim.x * blockDim.x) q.xring[PRE_XI(slot, ch, q.ring)] = make_double2(0.0, 0.0);
    if (blockIdx.x == 0 && threadIdx.x == 0) q.hold[ch] = now + PRE_L;
}
