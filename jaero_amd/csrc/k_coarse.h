// k_coarse.h -- coarse frequency estimator + acquisition state machine.
//
// Re-implements CoarseFreqEstimate::ProcessBasebandData (JAERO/coarsefreqestimate.cpp:90-137) and the
// FreqOffsetEstimateSlot it signals (JAERO/oqpskdemodulator.cpp:629-677, JAERO/mskdemodulator.cpp:490-519)
// for a list of channels, one 256-thread workgroup per channel (workgroups stride over the list).
//
//   ring (time order) --FFT N--> zero bins [startbin,stopbin] --IFFT (x N/N)--> square --FFT N--> fftshift
//   --> y = 0.9 y + 10 log10(max(|X|,1)) --> folded 3-bin peak search --> freq_offset_est --> slot logic.
//
// The N = 2^13 / 2^14 point fp64 complex FFT is a four-step (N1 x N2) transform: N2 column FFTs of length N1 in an
// LDS tile, twiddle, transposed write to an L2-resident scratch slab, then N1 row FFTs of length N2 in LDS.
// Twiddles W_N^k come from a host-generated table (shared by every channel, L2 resident).
#pragma once
#include "jaero_device.h"

#define CO_THREADS 256
#define CO_TILE_ELEMS 2048 // 32 KiB of double2

// radix-2 DIT FFT of `M` independent length-L vectors held in LDS; element (i, m) at tile[i*si + m*sm].
// Input must already be in bit-reversed order along i.  tw = W_N^k table, twstride = N / L.
template <int LOG2L>
__device__ __forceinline__ void lds_fft(double2 *tile, int M, int si, int sm, const double2 *__restrict__ tw, int twstride,
                                        bool inverse, int tid)
{
    constexpr int L = 1 << LOG2L;
    const int nb = (L / 2) * M;
#pragma unroll 1
    for (int s = 0; s < LOG2L; s++)
    {
        const int half = 1 << s;
        for (int b = tid; b < nb; b += CO_THREADS)
        {
            const int m = b % M;      // vector index fastest -> neighbouring threads touch neighbouring vectors
            const int j = b / M;
            const int k = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + k;
            const int i1 = i0 + half;
            double2 w = tw[(size_t)(k << (LOG2L - 1 - s)) * twstride];
            if (inverse) w.y = -w.y;
            double2 a = tile[i0 * si + m * sm];
            double2 c = tile[i1 * si + m * sm];
            double2 t;
            t.x = c.x * w.x - c.y * w.y;
            t.y = c.x * w.y + c.y * w.x;
            tile[i0 * si + m * sm] = make_double2(a.x + t.x, a.y + t.y);
            tile[i1 * si + m * sm] = make_double2(a.x - t.x, a.y - t.y);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int bitrev(int x, int bits) { return (int)(__brev((unsigned)x) >> (32 - bits)); }

// One N-point transform: dst = DFT(load(n)) for n < N.  src values come from `load`; dst may alias the buffer `load`
// reads from (all reads of pass 1 complete before pass 2 writes dst).  T is an N-element scratch.
template <int LOG2N, class Loader>
__device__ void fft_four_step(Loader load, double2 *__restrict__ T, double2 *__restrict__ dst, double2 *tile,
                              const double2 *__restrict__ tw, bool inverse, int tid)
{
    constexpr int N = 1 << LOG2N;
    constexpr int LOG2N1 = LOG2N / 2, LOG2N2 = LOG2N - LOG2N1;
    constexpr int N1 = 1 << LOG2N1, N2 = 1 << LOG2N2;
    constexpr int BC = CO_TILE_ELEMS / N1; // columns per batch
    constexpr int BR = CO_TILE_ELEMS / N2; // rows per batch
    // pass 1: column FFTs (length N1 over n1, stride N2), twiddle W_N^(n2*k1), store T[k1*N2 + n2]
    for (int b = 0; b < N2 / BC; b++)
    {
        for (int e = tid; e < N1 * BC; e += CO_THREADS)
        {
            const int c = e % BC, n1 = e / BC;
            tile[bitrev(n1, LOG2N1) * BC + c] = load(n1 * N2 + b * BC + c);
        }
        __syncthreads();
        lds_fft<LOG2N1>(tile, BC, BC, 1, tw, N / N1, inverse, tid);
        for (int e = tid; e < N1 * BC; e += CO_THREADS)
        {
            const int c = e % BC, k1 = e / BC;
            const int n2 = b * BC + c;
            double2 w = tw[(size_t)((n2 * k1) & (N - 1))];
            if (inverse) w.y = -w.y;
            const double2 v = tile[k1 * BC + c];
            T[k1 * N2 + n2] = make_double2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
        }
        __syncthreads();
    }
    // pass 2: row FFTs (length N2 over n2), store dst[k1 + N1*k2]
    for (int b = 0; b < N1 / BR; b++)
    {
        for (int e = tid; e < N2 * BR; e += CO_THREADS)
        {
            const int n2 = e % N2, r = e / N2;
            tile[r * N2 + bitrev(n2, LOG2N2)] = T[(b * BR + r) * N2 + n2];
        }
        __syncthreads();
        lds_fft<LOG2N2>(tile, BR, 1, N2, tw, N / N2, inverse, tid);
        for (int e = tid; e < N2 * BR; e += CO_THREADS)
        {
            const int r = e % BR, k2 = e / BR; // r fastest: BR consecutive k1 -> contiguous dst
            dst[(b * BR + r) + N1 * k2] = tile[r * N2 + k2];
        }
        __syncthreads();
    }
}

// Thread-0 epilogue of one estimate: emptyingcountdown (coarsefreqestimate.cpp:133-135), FreqOffsetEstimateSlot
// (oqpskdemodulator.cpp:629-677 / mskdemodulator.cpp:490-519), coarseCounter reset, status row.  Returns 1 when the
// AFC recentre fired (caller then performs bigchange(): y[i]=20 and zeroes the ring).
__device__ __forceinline__ int coarse_slot(const JGeom &g, const JPtrs &p, int ch, int zmaxloc, int N, double hzperbin, double lockingbw)
{
    const int nchp = g.nchp;
    int *I = p.I + ch;
    double *S = p.S + ch;
#define CI(f) I[(size_t)(f) * nchp]
#define CS(f) S[(size_t)(f) * nchp]
    double freq_offset_est = -((double)(zmaxloc - N / 2)) * hzperbin * 0.5;
    int emptying = CI(I_EMPTYING);
    if (emptying > 0) { emptying--; freq_offset_est = 0; }
    CI(I_EMPTYING) = emptying;

    const int flags = CI(I_FLAGS);
    const bool afc = flags & JF_AFC, dcd = flags & JF_DCD;
    const double mse = CS(S_MSE), thr = CS(S_THRESH);
    double m2_freq = CS(S_M2_FREQ), m2_step = CS(S_M2_STEP);
    double mc_freq = CS(S_MC_FREQ), mc_step = CS(S_MC_STEP);
    int countdown = CI(I_COUNTDOWN);
    bool big = false;
    if (g.kind == 1) // OQPSK
    {
        int countdown2 = CI(I_COUNTDOWN2);
        if ((mse < thr) && (!dcd))
        {
            if (countdown2 > 0) countdown2--;
            else jd_wt_setfreq(m2_freq, m2_step, mc_freq + freq_offset_est, g.Fs);
        }
        else countdown2 = 5;
        CI(I_COUNTDOWN2) = countdown2;
        if ((mse > thr) && (fabs(m2_freq - (mc_freq + freq_offset_est)) > 3.0))
            jd_wt_setfreq(m2_freq, m2_step, mc_freq + freq_offset_est, g.Fs);
        if ((afc) && (mse < thr) && (fabs(m2_freq - mc_freq) > 3.0))
        {
            if (countdown > 0) countdown--;
            else big = true;
        }
        else countdown = 4;
    }
    else // MSK
    {
        if ((mse > thr) && (fabs(m2_freq - (mc_freq + freq_offset_est)) > 0.0))
            jd_wt_setfreq(m2_freq, m2_step, mc_freq + freq_offset_est, g.Fs);
        if ((afc) && (dcd) && (fabs(m2_freq - mc_freq) > 2.0))
        {
            if (countdown > 0) countdown--;
            else big = true;
        }
        else countdown = 4;
    }
    if (big)
    {
        const double lbw = lockingbw; // demodulator's lockingbw == estimator's (oqpsk passes 2*bw/2)
        jd_wt_setfreq(mc_freq, mc_step, m2_freq, g.Fs);
        if (mc_freq < lbw / 2.0) jd_wt_setfreq(mc_freq, mc_step, lbw / 2.0, g.Fs);
        if (mc_freq > (g.Fs / 2.0 - lbw / 2.0)) jd_wt_setfreq(mc_freq, mc_step, g.Fs / 2.0 - lbw / 2.0, g.Fs);
        CI(I_EMPTYING) = 4; // coarsefreqestimate->bigchange()
    }
    CI(I_COUNTDOWN) = countdown;
    CS(S_M2_FREQ) = m2_freq; CS(S_M2_STEP) = m2_step;
    CS(S_MC_FREQ) = mc_freq; CS(S_MC_STEP) = mc_step;
    CI(I_COARSE_CNT) = 0; // :426 coarseCounter = 0
    const int nest = CI(I_NEST);
    if (g.flags & 2u)
    {
        const int lc = CI(I_LOG_CNT);
        if (lc < g.log_cap)
        {
            double *row = p.slog + ((size_t)ch * g.log_cap + lc) * 6;
            row[0] = (double)nest; row[1] = m2_freq; row[2] = mc_freq; row[3] = mse; row[4] = CS(S_EB_EBNO);
            row[5] = (mse > thr) ? 0.0 : 1.0;
            CI(I_LOG_CNT) = lc + 1;
        }
        else CI(I_OVERFLOW) = CI(I_OVERFLOW) | 4;
    }
    CI(I_NEST) = nest + 1;
#undef CI
#undef CS
    return big ? 1 : 0;
}

template <int LOG2N>
__global__ __launch_bounds__(CO_THREADS) void k_coarse(const JGeom g, const JPtrs p, const int *__restrict__ chan_list,
                                                       int nlist, double2 *__restrict__ scratch,
                                                       const double2 *__restrict__ tw)
{
    constexpr int N = 1 << LOG2N;
    __shared__ double2 tile[CO_TILE_ELEMS];
    __shared__ double red_val[CO_THREADS];
    __shared__ int red_idx[CO_THREADS];
    __shared__ int sh_bigchange;
    const int tid = threadIdx.x;
    const int nchp = g.nchp;
    double2 *B = scratch + (size_t)blockIdx.x * 2 * N;
    double2 *T = B + N;

    for (int li = blockIdx.x; li < nlist; li += gridDim.x)
    {
        const int ch = chan_list ? chan_list[li] : li;
        const double2 *__restrict__ ring = p.bbring + (size_t)ch * N;
        const int bb_ptr = p.I[(size_t)I_BB_PTR * nchp + ch];
        const double lockingbw = p.S[(size_t)S_LOCKINGBW * nchp + ch];
        const double hzperbin = g.Fs / ((double)N);
        const int startbin = (int)fmax(round(lockingbw / hzperbin), 1.0);
        const int stopbin = N - startbin;
        const int expectedpeakbin = (int)round(g.fb / (2.0 * hzperbin));
        double *__restrict__ y = p.y + (size_t)ch * N;

        // FFT 1: bbtmpbuff[j] = bbcycbuff[(ptr+j)%N] = CIS[idx] * dval
        auto load_ring = [&](int j) -> double2 { return ring[(bb_ptr + j) & (N - 1)]; };
        fft_four_step<LOG2N>(load_ring, T, B, tile, tw, false, tid);
        // IFFT of the band-limited spectrum (fb != 8400 boxcar branch, coarsefreqestimate.cpp:99)
        auto load_masked = [&](int i) -> double2 {
            if (i >= startbin && i <= stopbin) return make_double2(0.0, 0.0);
            return B[i];
        };
        fft_four_step<LOG2N>(load_masked, T, B, tile, tw, true, tid);
        // FFT of the squared signal
        auto load_sq = [&](int i) -> double2 {
            const double2 v = B[i];
            return make_double2(v.x * v.x - v.y * v.y, v.x * v.y + v.y * v.x);
        };
        fft_four_step<LOG2N>(load_sq, T, B, tile, tw, false, tid);

        // smooth (with fftshift): y[i] = y[i]*0.9 + 0.1*10*log10(fmax(abs(out[i]),1))
        for (int i = tid; i < N; i += CO_THREADS)
        {
            const double2 v = B[(i + N / 2) & (N - 1)];
            y[i] = y[i] * 0.9 + (0.1 * 10) * log10(fmax(hypot(v.x, v.y), 1.0));
        }
        __syncthreads();

        // fold + peak search (:116-131)
        const int i0 = (int)round((-lockingbw / hzperbin) + ((double)(N / 2)));
        const int i1 = (int)round((lockingbw / hzperbin) + ((double)(N / 2)));
        double best = 0;
        int besti = -1;
        for (int i = i0 + tid; i < i1; i += CO_THREADS)
        {
            if ((i < 0) || (i >= N)) continue;
            double val = 0;
            for (int j = -1; j <= 1; j++)
            {
                if (((i - expectedpeakbin - j) < 0) || ((i + expectedpeakbin + j) >= N)) continue;
                val += (y[i - expectedpeakbin - j] + y[i + expectedpeakbin + j]);
            }
            if (val > best) { best = val; besti = i; }
        }
        red_val[tid] = best;
        red_idx[tid] = besti;
        __syncthreads();
        for (int s = CO_THREADS / 2; s > 0; s >>= 1)
        {
            if (tid < s)
            {
                const double ov = red_val[tid + s];
                const int oi = red_idx[tid + s];
                const double mv = red_val[tid];
                const int mi = red_idx[tid];
                // strict '>' scanning upwards == largest value, lowest index on ties
                if (oi >= 0 && (mi < 0 || ov > mv || (ov == mv && oi < mi))) { red_val[tid] = ov; red_idx[tid] = oi; }
            }
            __syncthreads();
        }

        if (tid == 0) sh_bigchange = coarse_slot(g, p, ch, (red_idx[0] >= 0) ? red_idx[0] : (N / 2), N, hzperbin, lockingbw);
        __syncthreads();
        if (sh_bigchange)
        {
            // bigchange(): y[i]=20 ; bbcycbuff[j]=0
            double2 *ringw = p.bbring + (size_t)ch * N;
            for (int i = tid; i < N; i += CO_THREADS) { y[i] = 20; ringw[i] = make_double2(0.0, 0.0); }
        }
        __syncthreads();
    }
}
