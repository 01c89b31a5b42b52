// k_coarse5.h -- the 2^14-point coarse frequency estimator as TWO STREAMS per thread (round 3; replaces k_coarse4).
//
// Same function as k_coarse4 (CoarseFreqEstimate::ProcessBasebandData + FreqOffsetEstimateSlot, JAERO/coarsefreqestimate.cpp:90-137,
// JAERO/oqpskdemodulator.cpp:629-677), same factorisation 16384 = 16 x 16 x 16 x 4 out of 16-point register FFTs, one 512-thread
// workgroup per estimate, 32 points per thread.  What k_coarse4 left on the table: its eight wavefronts move
// through the same phase between the same barriers, so the LDS exchanges (~15 us per estimate) run while the VALU idles and the
// butterflies (~29 us) while the LDS idles.  Here the 32 points of a thread are two independent streams of 16:
//
//   n = n1*1024 + n2*64 + n3*4 + 2h + q         k = k1 + 16 k2 + 256 k3 + 4096 k4
//
// h (bit 1 of the index) is passive in passes 1-3 (FFT16 over n1, n2, n3) and only meets its partner in pass 4 (radix-4 over n4 =
// 2h + q), so stream h = 0 and stream h = 1 are separate 16-point problems with separate exchanges until the last pass.  While one
// stream's values travel through LDS (ds_write .. barrier .. ds_read .. barrier) the other stream's FFT16 and twiddles issue on the
// VALU; the exchange buffer holds both planes of ONE stream (2 x 64.3 KiB), so an exchange is two barrier intervals instead of four.
//
// Distribution D* (the same on entry and on exit, so FFT -> band limit -> FFT -> square -> FFT run register to register):
//   element e: stream e1 (bit 1), slot e >> 10, thread ((e >> 2) & 255) << 1 | (e & 0)... precisely t = ((e >> 2) & 255) * 2 + (e & 1).
// Global accesses in D*: lane pairs touch 2 consecutive elements, the two streams of a slot cover each other's gaps (ring: 2 x 32 B of
// every 64 B; y: 2 x 16 B of every 32 B), issued back to back.  Index maps, twiddles and LDS bank behaviour are pinned by the CPU model
// tests/test_coarse_fft14_model.py.
#pragma once
#include "../../jaero_amd/csrc/k_coarse2.h"

#ifndef C5_MIX
#define C5_MIX 0 // experiment: one LDS instruction after every few VALU instructions (sched_group_barrier) -- measured, no gain
#endif
#ifndef C4_TABN
#define C4_TABN 3584 // W8400: window table entries kept in LDS behind the exchange buffer (28 KiB): lockingbw < 10.49 kHz
#endif
#define C5_PLANE 8224                    // doubles per plane buffer (exchange 2 needs 31*257 + 256 = 8223)
#define C5_XCH (2 * C5_PLANE)            // 16448 doubles = 131 584 B: r plane of a stream in [0, 8224), i plane behind it

__device__ __forceinline__ void c5_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define C5_FENCE __builtin_amdgcn_sched_barrier(0)
// One LDS instruction after every NV VALU instructions, 16 times (a piece is 16 ds_write2st64_b64 / ds_read2..._b64 = 32 64-bit operations):
// scripts/ubench/lds_valu_overlap.hip -- a wavefront's LDS instructions each hold its issue for ~16 ns when they come as one block (the
// block takes as long as the whole CU's LDS traffic) and ~6 ns when other instructions of the same wavefront sit between them.
#if C5_MIX
#define C5_MIX1(M, NV) __builtin_amdgcn_sched_group_barrier(M, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
#define C5_MIX16(M, NV) do { C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) \
                             C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) C5_MIX1(M, NV) } while (0)
#else
#define C5_MIX16(M, NV) do { } while (0)
#endif

// v[k] *= step^k, k = 1 .. 15: powers by squares and one product each (depth <= 4, 14 complex products), then the 15 applications
__device__ __forceinline__ double2 c5_sq(const double2 a)
{
#pragma clang fp contract(fast)
    return make_double2(a.x * a.x - a.y * a.y, 2.0 * (a.x * a.y));
}
__device__ __forceinline__ void c5_twiddle16(CV<16> &v, const double2 p1)
{
#pragma clang fp contract(fast)
    double2 p[16];
    p[1] = p1;
    p[2] = c5_sq(p[1]); p[3] = cmul2(p[2], p[1]); p[4] = c5_sq(p[2]); p[5] = cmul2(p[4], p[1]); p[6] = c5_sq(p[3]); p[7] = cmul2(p[4], p[3]);
    p[8] = c5_sq(p[4]); p[9] = cmul2(p[8], p[1]); p[10] = c5_sq(p[5]); p[11] = cmul2(p[8], p[3]); p[12] = c5_sq(p[6]); p[13] = cmul2(p[8], p[5]);
    p[14] = c5_sq(p[7]); p[15] = cmul2(p[8], p[7]);
#pragma unroll
    for (int k = 1; k < 16; k++)
    {
        const double r = v.r[k] * p[k].x - v.i[k] * p[k].y, i = v.r[k] * p[k].y + v.i[k] * p[k].x;
        v.r[k] = r; v.i[k] = i;
    }
}

__device__ __forceinline__ void c5_fft16(CV<16> &v)
{
    CV<16> o;
    regfft<16>(v, o);
#pragma unroll
    for (int j = 0; j < 16; j++) { v.r[j] = o.r[j]; v.i[j] = o.i[j]; }
}

// pass-3 twiddle W_64^(k3 * (2H + q)), q wave-uniform (scalar): selects between literals
template <int H>
__device__ __forceinline__ void c5_twiddle_p3(CV<16> &v, const bool q)
{
#pragma clang fp contract(fast)
#pragma unroll
    for (int k3 = 1; k3 < 16; k3++)
    {
        const int e0 = (k3 * (2 * H)) & 63, e1 = (k3 * (2 * H + 1)) & 63;
        const double wr = q ? jd_w64r(e1) : jd_w64r(e0), wi = q ? jd_w64i(e1) : jd_w64i(e0);
        const double r = v.r[k3] * wr - v.i[k3] * wi, i = v.r[k3] * wi + v.i[k3] * wr;
        v.r[k3] = r; v.i[k3] = i;
    }
}

// exchange pieces: W = 32 ds_write_b64 (both planes of one stream), R = 32 ds_read_b64.  Offsets are compile-time multiples of the slot.
#define C5_W(v, base, stride) \
    _Pragma("unroll") for (int s_ = 0; s_ < 16; s_++) { xa[(base) + s_ * (stride)] = v.r[s_]; xb[(base) + s_ * (stride)] = v.i[s_]; }
#define C5_R(v, base, stride) \
    _Pragma("unroll") for (int s_ = 0; s_ < 16; s_++) { v.r[s_] = xa[(base) + s_ * (stride)]; v.i[s_] = xb[(base) + s_ * (stride)]; }

// In-place forward 2^14-point DFT of the workgroup's data in distribution D*, streams a (h = 0) and b (h = 1).
// On entry the exchange buffer may still be read by other wavefronts (the caller's previous use): the first write is behind a barrier.
template <int ABL = 0>
__device__ __forceinline__ void wg_fft14_2s(CV<16> &a, CV<16> &b, double *xa, const double2 *__restrict__ tw, int t)
{
#pragma clang fp contract(fast)
    constexpr bool DOV = !(ABL & 4), DOL = !(ABL & 8); // micro-benchmarks: transforms without their VALU pieces / without their LDS pieces
    double *xb = xa + C5_PLANE;
    // table values of this transform: requested before the first butterfly
    const int r1 = ((t >> 1) << 2) | (t & 1);                  // n mod 1024 of stream 0 in pass 1 (stream 1: + 2)
    const double2 s1a = tw[r1], s1b = tw[r1 + 2];
    const int r2 = (((t >> 1) & 15) << 2) | (t & 1);           // n mod 64 of stream 0 in pass 2 (thread t2 = k1<<5 | n3<<1 | q)
    const double2 s2a = tw[16 * r2], s2b = tw[16 * (r2 + 2)];
    const int e1w = t, e1r = (t >> 5) * 512 + (t & 31);        // exchange 1: write slot k1 at e1w + 512 k1, read slot n2 at e1r + 32 n2
    const int k1 = t >> 5;
    const int e2w = ((((k1 >> 1) & 1) << 7) | ((k1 >> 2) << 1) | (k1 & 1)) + 257 * (t & 31);   // + 8 k2
    const int e2r = (t & 255) + 257 * (t >> 8);                // + 514 n3
    const int e3w = t;                                          // + 512 k3
    const int e3r = (t >> 7) * 512 + (t & 127);                // + 2048 k3hi + 256 q + 128 b1
    const bool q3 = __builtin_amdgcn_readfirstlane(t >> 8) != 0; // pass 3: q = top thread bit, wave-uniform

    // A piece of LDS traffic of one stream (L: 16 ds_write2st64 / ds_read2 instructions) shares its barrier interval with a piece of the
    // other stream's arithmetic (V).  Tried and dropped: the two wavefronts of a SIMD taking their barriers at different places (one behind
    // every V piece, the other behind every L piece) so that one computes while the other sits in the LDS queue -- no gain, a wavefront's L
    // piece takes as long with four wavefronts issuing as with eight (scripts/ubench/lds_valu_overlap.hip: "phased skewed").
#define C5_SLOT(L, V, M, NV) do { if constexpr (DOL) { L } if constexpr (DOV) { V; } C5_MIX16(M, NV); C5_FENCE; c5_bar(); C5_FENCE; } while (0)
#define C5_R3(v) _Pragma("unroll") for (int s_ = 0; s_ < 16; s_++) { \
        const int off_ = ((s_ >> 1) & 3) * 2048 + (s_ >> 3) * 256 + (s_ & 1) * 128; v.r[s_] = xa[e3r + off_]; v.i[s_] = xb[e3r + off_]; }
    // V0
    if constexpr (DOV) { c5_fft16(a); c5_twiddle16(a, s1a); }
    C5_FENCE;
    c5_bar();                                   // the buffer is free (previous transform's last reads / the fold)
    C5_FENCE;
    C5_SLOT(C5_W(a, e1w, 512), c5_fft16(b), 0x200, 10);              // L1 V1
    C5_SLOT(C5_R(a, e1r, 32), c5_twiddle16(b, s1b), 0x100, 7);       // L2 V2
    C5_SLOT(C5_W(b, e1w, 512), c5_fft16(a), 0x200, 10);              // L3 V3
    C5_SLOT(C5_R(b, e1r, 32), c5_twiddle16(a, s2a), 0x100, 7);       // L4 V4
    C5_SLOT(C5_W(a, e2w, 8), c5_fft16(b), 0x200, 10);                // L5 V5
    C5_SLOT(C5_R(a, e2r, 514), c5_twiddle16(b, s2b), 0x100, 7);      // L6 V6
    C5_SLOT(C5_W(b, e2w, 8), c5_fft16(a), 0x200, 10);                // L7 V7
    C5_SLOT(C5_R(b, e2r, 514), c5_twiddle_p3<0>(a, q3), 0x100, 4);   // L8 V8
    C5_SLOT(C5_W(a, e3w, 512), c5_fft16(b), 0x200, 10);              // L9 V9
    C5_SLOT(C5_R3(a), c5_twiddle_p3<1>(b, q3), 0x100, 4);            // L10 V10: reader slot index q*8 + k3hi*2 + b1
    C5_SLOT(C5_W(b, e3w, 512), (void)0, 0x200, 0);                   // L11
    if constexpr (DOL) { C5_R3(b) }                                   // L12
    C5_FENCE;
#undef C5_SLOT
#undef C5_R3
    // pass 4: radix-4 over n4 = 2h + q for each (k3hi, b1): x0 = a[q=0], x1 = a[q=1], x2 = b[q=0], x3 = b[q=1]
    // -> stream' b1, slot k4*4 + k3hi
    {
        CV<16> oa, ob;
#pragma unroll
        for (int g = 0; g < 8; g++) // g = k3hi*2 + b1
        {
            const double ar = a.r[g], ai = a.i[g], br = a.r[8 + g], bi = a.i[8 + g];
            const double cr = b.r[g], ci = b.i[g], er = b.r[8 + g], ei = b.i[8 + g];
            const double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
            const double t2r = br + er, t2i = bi + ei;
            const double t3r = (bi - ei), t3i = -(br - er); // -i (x1 - x3)
            const int k3hi = g >> 1;
            CV<16> &o = (g & 1) ? ob : oa;
            o.r[0 * 4 + k3hi] = t0r + t2r; o.i[0 * 4 + k3hi] = t0i + t2i;
            o.r[1 * 4 + k3hi] = t1r + t3r; o.i[1 * 4 + k3hi] = t1i + t3i;
            o.r[2 * 4 + k3hi] = t0r - t2r; o.i[2 * 4 + k3hi] = t0i - t2i;
            o.r[3 * 4 + k3hi] = t1r - t3r; o.i[3 * 4 + k3hi] = t1i - t3i;
        }
#pragma unroll
        for (int s = 0; s < 16; s++) { a.r[s] = oa.r[s]; a.i[s] = oa.i[s]; b.r[s] = ob.r[s]; b.i[s] = ob.i[s]; }
    }
}

template <int ABL = 0>
__device__ __forceinline__ void c5_fft(CV<16> &a, CV<16> &b, double *xch, const double2 *__restrict__ tw, int t)
{
    int tt = t; // laundered per call: nothing derived from it inside is shared between the three calls of an estimate and kept live
    asm volatile("" : "+v"(tt));
    wg_fft14_2s<ABL>(a, b, xch, tw, tt);
}

// element index held in (stream H, slot s) by thread t
#define C5_IDX(H, s, t) (((s) << 10) | (((t) >> 1) << 2) | ((H) << 1) | ((t) & 1))

// ABL (micro-benchmarks only, scripts/ubench/coarse_bench.hip): bit 0 = no ring / y traffic, bit 1 = no transforms, bit 2 = transforms
// without their VALU pieces, bit 3 = transforms without their LDS pieces
template <bool W8400, int ABL = 0>
__device__ __forceinline__ void coarse5_body(const JGeom g, const JPtrs p, const int *__restrict__ chan_list, int nlist, const double2 *__restrict__ tw)
{
    constexpr int N = 1 << 14;
    extern __shared__ __attribute__((aligned(16))) double xch[];
    __shared__ double red_val[C2_THREADS / 64]; // one entry per wavefront
    __shared__ int red_idx[C2_THREADS / 64];
    __shared__ int sh_bigchange;
    const int t0 = threadIdx.x;
    const int nchp = g.nchp;
    int tab_startbin = -1; // W8400: the startbin the window table behind the exchange buffer was made for

    CV<16> a, b;
    int ch_next = ((int)blockIdx.x < nlist) ? (chan_list ? jd_sload(chan_list + blockIdx.x) : (int)blockIdx.x) : 0;
    int bp_next = jd_sload(p.I + (size_t)I_BB_PTR * nchp + ch_next);
    for (int li = blockIdx.x; li < nlist; li += gridDim.x)
    {
        int t = t0; // opaque once per estimate (what derives from it is 1-2 instructions; hoisted out of the persistent loop, ~100 live registers)
        asm volatile("" : "+v"(t));
        const int ch = ch_next, bb_ptr = bp_next;
        const double2 *__restrict__ ring = p.bbring + (size_t)ch * N;
        // the next estimate's channel and ring position: requested a whole estimate before the prefetch that needs them (they used to be two
        // dependent round trips in front of it)
        const int ln = li + (int)gridDim.x;
        const bool has_next = ln < nlist;
        if (has_next)
        {
            ch_next = chan_list ? jd_sload(chan_list + ln) : ln; // scalar loads (k_coarse.h): no vmcnt wait behind the vector traffic
            bp_next = jd_sload(p.I + (size_t)I_BB_PTR * nchp + ch_next);
        }
        const double lockingbw = jd_sload(p.S + (size_t)S_LOCKINGBW * nchp + ch);
        double fs_l = g.Fs; // opaque per estimate, as t: hoisted out of the loop Fs / N would be kept (spilled) across it, and a reload's wait
        asm volatile("" : "+s"(fs_l)); // stands behind every vector load in flight (vmcnt counts in order)
        const double hzperbin = fs_l * (1.0 / ((double)N)); // N a power of two: the same bits as the reference's quotient
        const int startbin = (int)fmax(round(lockingbw / hzperbin), 1.0);
        const int stopbin = N - startbin;
        const int expectedpeakbin = (int)round(g.fb / (2.0 * hzperbin));
        double *__restrict__ y = p.y + (size_t)ch * N;

        // bbtmpbuff[j] = bbcycbuff[(ptr+j)%N] (time order) in D*; for every list entry but the first these loads were issued while the
        // previous estimate was in its peak search / state machine
        if (li == (int)blockIdx.x && !(ABL & 1))
        {
#pragma unroll
            for (int s = 0; s < 16; s++)
            {
                const double2 v0 = ring[(bb_ptr + C5_IDX(0, s, t)) & (N - 1)];
                const double2 v1 = ring[(bb_ptr + C5_IDX(1, s, t)) & (N - 1)];
                a.r[s] = v0.x; a.i[s] = v0.y; b.r[s] = v1.x; b.i[s] = v1.y;
            }
        }
        if constexpr (ABL & 1)
        {
            if (li == (int)blockIdx.x)
            {
#pragma unroll
                for (int s = 0; s < 16; s++) { a.r[s] = (double)(t + s); a.i[s] = (double)(t - s); b.r[s] = (double)(t ^ s); b.i[s] = 1.0; }
            }
        }
        if constexpr (!(ABL & 2)) c5_fft<ABL>(a, b, xch, tw, t);
        // band limit (fb != 8400 boxcar, coarsefreqestimate.cpp:99) then inverse transform = forward on swapped planes
        if constexpr (W8400)
        {
            // window[0] = 1, window[i] = window[N - i] = cos^2(pi/2 * i / startbin) for 1 <= i <= startbin, 0 elsewhere (:61-74).  Its
            // startbin + 1 distinct values come from a table in LDS behind the exchange buffer (entry startbin + 1 = 0 stands for every bin
            // the window zeroes), rebuilt only when startbin changes; a window wider than that space (lockingbw >= 10.49 kHz) is made per
            // estimate in the idle exchange buffer.
            const bool persistent = startbin < C4_TABN - 1;
            double *wt = persistent ? xch + C5_XCH : xch;
            if (!persistent || startbin != tab_startbin)
            {
                c5_bar();
                for (int i = t; i <= startbin + 1; i += C2_THREADS)
                {
                    const double c = cos(M_PI_2 * ((double)i) / ((double)startbin));
                    wt[i] = (i == 0) ? 1.0 : ((i <= startbin) ? c * c : 0.0);
                }
                c5_bar();
                if (persistent) tab_startbin = startbin;
            }
#pragma unroll
            for (int s0 = 0; s0 < 16; s0 += 4)
            {
#pragma unroll
                for (int s = s0; s < s0 + 4; s++)
                {
                    const int k0 = C5_IDX(0, s, t), k1 = C5_IDX(1, s, t);
                    const int i0 = (k0 <= N / 2) ? k0 : N - k0, i1 = (k1 <= N / 2) ? k1 : N - k1;
                    const double w0 = wt[i0 <= startbin ? i0 : startbin + 1], w1 = wt[i1 <= startbin ? i1 : startbin + 1];
                    const double re0 = a.r[s] * w0, im0 = a.i[s] * w0, re1 = b.r[s] * w1, im1 = b.i[s] * w1;
                    a.r[s] = im0; a.i[s] = re0; b.r[s] = im1; b.i[s] = re1;
                }
                C5_FENCE;
            }
            if (!persistent) c5_bar(); // the next transform's exchanges reuse the buffer
        }
        else
        {
#pragma unroll
            for (int s = 0; s < 16; s++)
            {
                const int k0 = C5_IDX(0, s, t), k1 = C5_IDX(1, s, t);
                const bool z0 = (k0 >= startbin) && (k0 <= stopbin), z1 = (k1 >= startbin) && (k1 <= stopbin);
                const double re0 = z0 ? 0.0 : a.r[s], im0 = z0 ? 0.0 : a.i[s], re1 = z1 ? 0.0 : b.r[s], im1 = z1 ? 0.0 : b.i[s];
                a.r[s] = im0; a.i[s] = re0; b.r[s] = im1; b.i[s] = re1;
            }
        }
        if constexpr (!(ABL & 2)) c5_fft<ABL>(a, b, xch, tw, t);
        // swap back (x N / N = 1), square
#pragma unroll
        for (int s = 0; s < 16; s++)
        {
            const double re0 = a.i[s], im0 = a.r[s], re1 = b.i[s], im1 = b.r[s];
            a.r[s] = re0 * re0 - im0 * im0; a.i[s] = re0 * im0 + im0 * re0;
            b.r[s] = re1 * re1 - im1 * im1; b.i[s] = re1 * im1 + im1 * re1;
        }
        if constexpr (!(ABL & 2)) c5_fft<ABL>(a, b, xch, tw, t);
        c5_bar(); // the exchange buffer is free: it receives a copy of y for the fold below
        // smooth with fftshift: y[i] = y[i]*0.9 + 0.1*10*log10(fmax(abs(out[i]),1)), out[i] = X[i ^ N/2]
        // All 32 old y values are requested before the log10s, into the registers of the imaginary planes (dead once only |X|^2 is kept).
        // y[] is this kernel's private state (nothing else reads it; bigchange() sets every entry), so in HBM it is kept in the ORDER THE
        // THREADS HOLD IT: entry (stream H, slot s, thread t) -- the smoothed value of bin C5_IDX(H, s, t) ^ N/2 -- at (H*16 + s)*512 + t:
        // every access a whole 512-byte row.  In bin order the two streams of a slot own alternate 16-byte pieces; written one stream after
        // the other (tried: stream a first, its half of the next ring requested under stream b's log10s, 1.6 % faster) a launch moved 51.7
        // GB instead of 34.6 (partial-sector writes and their fills), written back to back still 37.8.  The ring keeps the reference's
        // order (the sample loop fills it): its 32-byte pieces of the two streams are requested back to back.
        {
            double ya[16], yb[16];
            const int toff = ((t >> 1) << 2) | (t & 1);
#pragma unroll
            for (int s = 0; s < 16; s++) { a.r[s] = a.r[s] * a.r[s] + a.i[s] * a.i[s]; b.r[s] = b.r[s] * b.r[s] + b.i[s] * b.i[s]; }
            C5_FENCE;
            if constexpr (!(ABL & 1))
            {
#pragma unroll
                for (int s = 0; s < 16; s++) { ya[s] = (y + s * 512)[t]; yb[s] = (y + (16 + s) * 512)[t]; } // uniform base + lane: whole 512-byte rows
            }
            else
            {
#pragma unroll
                for (int s = 0; s < 16; s++) { ya[s] = 1.0; yb[s] = 2.0; }
            }
            C5_FENCE; // or the scheduler sinks every load to its use again
            // 10*log10(max(|X|,1)) == 5*log10(max(|X|^2,1)): no hypot; differs from the reference expression by <= 1 ulp
#pragma unroll
            for (int s = 0; s < 16; s++) { a.r[s] = 5.0 * c2_log10(fmax(a.r[s], 1.0)); b.r[s] = 5.0 * c2_log10(fmax(b.r[s], 1.0)); }
#pragma unroll
            for (int s = 0; s < 16; s++)
            {
                const double na = ya[s] * 0.9 + a.r[s], nb = yb[s] * 0.9 + b.r[s];
                if constexpr (!(ABL & 1)) { (y + s * 512)[t] = na; (y + (16 + s) * 512)[t] = nb; }
                (xch + ((s ^ 8) << 10))[toff] = na; (xch + (((s ^ 8) << 10) + 2))[toff] = nb;
            }
        }
        c5_bar(); // the fold reads the LDS copy; the stores to y[] drain in the background
        if constexpr (!(ABL & 1))
        {
            if (has_next)
            {
                // laundered: known since the top of the estimate, the 32 ring addresses would otherwise be computed there and kept (spilled)
                // across the three transforms
                int bpn = bp_next, chn = ch_next, tp = t;
                asm volatile("" : "+v"(bpn), "+v"(chn), "+v"(tp));
                const double2 *__restrict__ ringn = p.bbring + (size_t)chn * N;
                const int toffp = bpn + (((tp >> 1) << 2) | (tp & 1));
#pragma unroll
                for (int s = 0; s < 16; s++)
                {
                    const double2 v0 = ringn[(toffp + (s << 10)) & (N - 1)];
                    const double2 v1 = ringn[(toffp + (s << 10) + 2) & (N - 1)];
                    a.r[s] = v0.x; a.i[s] = v0.y; b.r[s] = v1.x; b.i[s] = v1.y;
                }
            }
        }
        else
        {
#pragma unroll
            for (int s = 0; s < 16; s++) { a.r[s] = (double)(t + s); a.i[s] = (double)(t - s); b.r[s] = (double)(t ^ s); b.i[s] = 1.0; }
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
        // reduction through shuffles, then one LDS round for the eight wavefront results -- no barrier drains the ring prefetch in flight
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
            c5_bar();
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
        c5_bar();
        if (sh_bigchange)
        {
            __syncthreads(); // rare (AFC recentre): this estimate's y stores must have landed before other threads overwrite the same rows
            double2 *ringw = p.bbring + (size_t)ch * N;
            for (int i = t; i < N; i += C2_THREADS) { y[i] = 20; ringw[i] = make_double2(0.0, 0.0); }
        }
        // no barrier here: the next use of LDS is behind the first barrier of the next estimate's transform (wg_fft14_2s slot 1), and
        // red_val / red_idx / sh_bigchange are next written behind several more
    }
}

__global__ __launch_bounds__(C2_THREADS) void k_coarse5(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                           int nlist, const double2 *__restrict__ tw)
{
    coarse5_body<false>(g, p, chan_list, nlist, tw);
}
__global__ __launch_bounds__(C2_THREADS) void k_coarse5_w8400(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                                 int nlist, const double2 *__restrict__ tw)
{
    coarse5_body<true>(g, p, chan_list, nlist, tw);
}
