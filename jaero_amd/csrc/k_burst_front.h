// k_burst_front.h -- producer side of the burst demodulators: Hilbert FIR, burst-timing detector, trident check.
// Reference: JAERO/burstoqpskdemodulator.cpp:323-515, JAERO/burstmskdemodulator.cpp:376-569, JAERO/DSP.cpp:754-794 (QJHilbertFilter),
// JAERO/DSP.h:491-576 (PeakDetector), JAERO/fftrwrapper.cpp:19-27.
#pragma once
#include "burst_device.h"
#include "jaero_device.h"
#include "k_coarse2.h"
#include "k_coarse6.h" // wg_fft13_e32
#include "k_pre8400.h" // pf_fft4096

#define BLDF(f) (p.S[(size_t)(f) * nchp + ch])
#define BLDI(f) (p.I[(size_t)(f) * nchp + ch])

// ------------------------------------------------------------------------------------------------ Hilbert
// hfir.update(hfirbuff) (JAERO/burstoqpskdemodulator.cpp:344, JAERO/DSP.cpp:754-794): the 2048-tap kernel is
//   kernel[1024] = -1, kernel[odd k] = j (2/N)/tan(pi (k/N - 1/2)), everything else 0, and JFastFir delays by L = nfft-K+1.
// So   re y[n] = -x[n-L-1024],   im y[n] = sum_{j<512} h[2j+1] (x[n-L-(2j+1)] - x[n-L-(2047-2j)])   (h[N-k] = -h[k]).
// The PCM history ring is kept in cells of four consecutive samples per channel: int16 index ((slot >> 2) * nchp + ch) * 4 + (slot & 3).
// (The time-parallel direct form of this filter, k_hilbert<PH>, 8.9 ms per 2048-sample segment against 1.4 ms, left the library in round 3.)
__device__ __forceinline__ size_t hb_idx(int slot, int nchp, int ch) { return ((size_t)(slot >> 2) * nchp + ch) * 4 + (slot & 3); }
// The same filter by overlap-save (the reference's QJHilbertFilter IS a JFastFir): the kernel's real part is a single tap (-1 at
// k = 1024), so re y is a delayed copy of the input and only the imaginary taps g[k] (odd k, REAL values) need a convolution.  With
// real taps two real channels share one complex transform pair: z = x_a + j x_b,  g (*) z = (g (*) x_a) + j (g (*) x_b).
//   grid (nchp / 8, 2048-sample blocks the segment touches), 1024 threads = 4 channel pairs x 256 threads, 128 KiB LDS; thread
//   (c = tid & 3, T = tid >> 2) serves channels ch0 + 2c, ch0 + 2c + 1: the four c of a T read one whole 64-byte history cell row
//   (8 channels x 4 samples) and write 64 contiguous bytes of him.
//   Block: outputs m0 .. m0 + 2047 (m0 an absolute multiple of 2048) = window indices 2048 .. 4095 of x[m0 - L - 2048 .. m0 - L + 2047]
//   convolved with g (2048 taps): im y[m] = sum_k g[k] x[m - L - k].  4096-point transforms: pf_fft4096 (k_pre8400.h).
// 1.4 ms per 2048-sample segment of 65 536 channels against 8.9 ms for the direct form it replaced.  Not
// bit-identical to the direct form (other summation order: ~1e-15 of full scale), nor to the reference's transform.
__global__ __launch_bounds__(PF_THREADS) void k_hilbert_fft(const BGeom g, const BPtrs p, int ns, long long n0)
{
    extern __shared__ double hf_xch[]; // 4 pairs x 4096 doubles
    const int tid = threadIdx.x, c = tid & 3, T = tid >> 2;
    const int nchp = g.nchp, H = g.hist_len;
    const int cha = blockIdx.x * 8 + 2 * c; // and cha + 1
    const long long m0 = ((n0 >> 11) + blockIdx.y) << 11;
    const int i0 = (int)(m0 - n0); // may be negative: that part of the block belonged to the previous segment
    const int16_t *__restrict__ hist = p.pcmhist;
    CV<16> d;
    {
        // time of window index 0, made non-negative by a multiple of the ring length (before the stream starts the ring holds zeros)
        // (one modulo per thread + a conditional wrap per element was measured SLOWER: 1.66 against 1.43 ms per segment)
        const long long tb = m0 - g.hil_lat - 2048 + T + 4LL * H;
#pragma unroll
        for (int s = 0; s < 16; s++)
        {
            const int slot = (int)((tb + 256 * s) % H);
            const size_t o = hb_idx(slot, nchp, cha);
            d.r[s] = (double)hist[o];
            d.i[s] = (double)hist[o + 4];
        }
    }
    pf_fft4096(d, hf_xch, p.tw12, T, c);
    {
        const double2 *__restrict__ Hh = p.hilH + T;
#pragma unroll
        for (int s = 0; s < 16; s++)
        {
            const double2 h = Hh[256 * s];
            const double yr = d.r[s] * h.x - d.i[s] * h.y, yi = d.r[s] * h.y + d.i[s] * h.x;
            d.r[s] = yr; d.i[s] = -yi;
        }
    }
    pf_fft4096(d, hf_xch, p.tw12, T, c);
    const int grp = cha >> 6, lane = cha & 63;
#pragma unroll
    for (int s = 8; s < 16; s++)
    {
        const int i = i0 + T + 256 * (s - 8);
        if (i >= 0 && i < ns)
        {
            // re y[m] = -x[m - L - 1024] is a delayed copy of the input: k_burst_front reads it from the history ring itself (2 bytes instead of
            // 8 written here and 8 read there).  The taps carry no scaling, so the sums are scaled here, as the reference scales its input.
            const size_t q = ((size_t)grp * g.maxseg + i) * 64 + lane;
            *(double2 *)(p.him + q) = make_double2(d.r[s] / 32768.0, (-d.i[s]) / 32768.0);
        }
    }
}

// PCM frames -> history ring cells (frame-major [n][stride])
__global__ void k_hist_push_frames(const int16_t *__restrict__ src, int stride, int nch, int16_t *__restrict__ hist, int nchp, int H,
                                   int slot0, int n)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (c >= nchp || i >= n) return;
    int slot = slot0 + i; if (slot >= H) slot -= H;
    hist[hb_idx(slot, nchp, c)] = (c < nch) ? src[(size_t)i * stride + c] : (int16_t)0;
}
// channel-major [nch][n] -> ring cells, through a 64x64 LDS tile
__global__ void k_hist_push_chmajor(const int16_t *__restrict__ src, int nch, int n, int16_t *__restrict__ hist, int nchp, int H, int slot0)
{
    __shared__ int16_t tile[64][65];
    const int c0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
    {
        const int c = c0 + r, i = i0 + tx;
        tile[r][tx] = (c < nch && i < n) ? src[(size_t)c * n + i] : (int16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4)
    {
        const int i = i0 + r, c = c0 + tx;
        if (i < n && c < nchp)
        {
            int slot = (slot0 + i) % H;
            hist[hb_idx(slot, nchp, c)] = tile[tx][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ front end
// Per sample (burstoqpskdemodulator.cpp:372-410 == burstmskdemodulator.cpp:413-441): AGC, the delay lines, the burst-timing
// signal, the peak detector and the trident-buffer fill counter.  d1, d2, bt_d1 and tridentbuffer are all read out of ONE ring
// of AGC'd analytic samples (cvre/cvim).  All ring addresses are data independent, so a block of FB samples' worth of loads is
// issued before the block's serial arithmetic (every lag is >= 9 > FB, so nothing a block needs is written inside it).
#define FB 8
__device__ __forceinline__ int bwrap(int s, int len) { return s >= len ? s - len : s; }
__device__ __forceinline__ int bback(int s, int lag, int len) { s -= lag; return s < 0 ? s + len : s; }

// HOLD: the launches that cover the first bt_lag samples behind a setSettings (burst_host.h keeps count): bt_d1 was refilled with zeros there
// (Delay<>::setdelay, DSP.h:349-356) while the ring its two taps are read from still holds the samples d1 needs later.
template <bool HOLD>
__global__ __launch_bounds__(64) void k_burst_front(const BGeom g, const BPtrs p, int n, long long n0)
{
    const int lane = threadIdx.x, grp = blockIdx.x, ch = grp * 64 + lane, nchp = g.nchp;
    double agc_sum = BLDF(BS_AGC_SUM), ma1_re = BLDF(BS_MA1_RE), ma1_im = BLDF(BS_MA1_IM), mav1_sum = BLDF(BS_MAV1_SUM), lastdy = BLDF(BS_LASTDY);
    int cntdown = BLDI(BI_CNTDOWN), maxposcd = BLDI(BI_MAXPOSCD), tri_ptr = BLDI(BI_TRI_PTR);
    const int flags = BLDI(BI_FLAGS);
    int ev_pos = -1, ev_cnt = BLDI(BI_EV_CNT);
    int hold = HOLD ? BLDI(BI_BT_HOLD) : 0;
    const bool trace = (g.flags & 8u) != 0; // JAERO_FLAG_TRACE

    const double *__restrict__ him = p.him + (size_t)grp * g.maxseg * 64 + lane;
    const int16_t *__restrict__ hist = p.pcmhist;
    // re y[m] = -x[m - L - 1024] (k_hilbert_fft): slot of sample n0's, advanced with the samples
    const int s_re0 = (int)((n0 - g.hil_lat - 1024 + 4LL * g.hist_len) % g.hist_len);
    double *agc_ring = p.agc_ring + (size_t)grp * g.agc_len * 64 + lane;
    double *cvre = p.cvre + (size_t)grp * g.cv_len * 64 + lane;
    double *cvim = p.cvim + (size_t)grp * g.cv_len * 64 + lane;
    double *ma1r = p.ma1re + (size_t)grp * g.ma1_len * 64 + lane;
    double *ma1i = p.ma1im + (size_t)grp * g.ma1_len * 64 + lane;
    double *mav1 = p.mav1 + (size_t)grp * g.mav1_len * 64 + lane;
    double *fa = p.fa + (size_t)grp * g.fa_len * 64 + lane;
    double *bt = p.bt + (size_t)grp * g.bt_len * 64 + lane;

    int s_agc = (int)(n0 % g.agc_len), s_cv = (int)(n0 % g.cv_len), s_ma1 = (int)(n0 % g.ma1_len), s_mav1 = (int)(n0 % g.mav1_len);
    int s_fa = (int)(n0 % g.fa_len), s_bt = (int)(n0 % g.bt_len);
    const double agc_len_d = (double)g.agc_len, ma1_len_d = (double)g.ma1_len, mav1_len_d = (double)g.mav1_len;
    const double btw = g.bt_w, btwc = 1.0 - g.bt_w, faw = g.fa_w, fawc = 1.0 - g.fa_w;
    const int twoPL = 2 * g.PL;

    for (int i0 = 0; i0 < n; i0 += FB)
    {
        double x_re[FB], x_im[FB], o_agc[FB], o_m1r[FB], o_m1i[FB], o_mav[FB], fa_old[FB], fa_new[FB], bt_2[FB], bt_1[FB];
        double co_r[FB], co_i[FB], cn_r[FB], cn_i[FB];
#pragma unroll
        for (int k = 0; k < FB; k++)
        {
            const int ii = (i0 + k < n) ? i0 + k : n - 1;
            const int d = ii - i0; // == k except past the end (results unused there)
            // PCM -> double as the reference does (x / 32768.0)
            x_re[k] = -(((double)hist[hb_idx(bwrap(s_re0 + ii, g.hist_len), nchp, ch)]) / 32768.0); x_im[k] = him[(size_t)ii * 64];
            o_agc[k] = agc_ring[(size_t)bwrap(s_agc + d, g.agc_len) * 64];
            o_m1r[k] = ma1r[(size_t)bwrap(s_ma1 + d, g.ma1_len) * 64];
            o_m1i[k] = ma1i[(size_t)bwrap(s_ma1 + d, g.ma1_len) * 64];
            o_mav[k] = mav1[(size_t)bwrap(s_mav1 + d, g.mav1_len) * 64];
            const int cs = bwrap(s_cv + d, g.cv_len);
            const int c_old = bback(cs, g.bt_lag, g.cv_len), c_new = bwrap(c_old + 1, g.cv_len);
            co_r[k] = cvre[(size_t)c_old * 64]; co_i[k] = cvim[(size_t)c_old * 64];
            cn_r[k] = cvre[(size_t)c_new * 64]; cn_i[k] = cvim[(size_t)c_new * 64];
            const int fs = bwrap(s_fa + d, g.fa_len);
            const int f_old = bback(fs, g.fa_lag, g.fa_len), f_new = bwrap(f_old + 1, g.fa_len);
            fa_old[k] = fa[(size_t)f_old * 64]; fa_new[k] = fa[(size_t)f_new * 64];
            const int bs = bwrap(s_bt + d, g.bt_len);
            bt_2[k] = bt[(size_t)bback(bs, twoPL, g.bt_len) * 64];
            bt_1[k] = bt[(size_t)bback(bs, g.PL, g.bt_len) * 64];
        }
#pragma unroll
        for (int k = 0; k < FB; k++)
        {
            const int i = i0 + k;
            if (i >= n) break;
            // agc->Update(std::abs(cval)); cval*=agc->AGCVal  (DSP.cpp:370-379)
            const double a = hypot(x_re[k], x_im[k]);
            agc_sum = agc_sum - o_agc[k];
            agc_sum = agc_sum + fabs(a);
            agc_ring[(size_t)s_agc * 64] = fabs(a);
            double gain = 1.414213562 / fmax(agc_sum / agc_len_d, 0.000001);
            gain = fmax(gain, 0.000001);
            const double c_re = x_re[k] * gain, c_im = x_im[k] * gain;
            cvre[(size_t)s_cv * 64] = c_re; cvim[(size_t)s_cv * 64] = c_im;
            // bt_d1.update(cval) (Delay<cpx>, DSP.h:341-379): weighting*newer + (1-weighting)*older; with an integer delay
            // (burst MSK) the "newer" entry is the one just written
            double nr = cn_r[k], ni = cn_i[k], orr = co_r[k], oi = co_i[k];
            if (g.bt_lag == 1) { nr = c_re; ni = c_im; }
            if (HOLD)
            {
                // the older tap was written bt_lag samples ago, the newer one bt_lag - 1: zeros until setSettings lies that far back
                if (hold > 0) { orr = 0.0; oi = 0.0; }
                if (hold > 1) { nr = 0.0; ni = 0.0; }
                if (hold > 0) hold--;
            }
            const double dl_re = btw * nr + btwc * orr, dl_im = btw * ni + btwc * oi;
            // cval*std::conj(dl)
            const double pr = c_re * dl_re - c_im * (-dl_im), pi = c_re * (-dl_im) + c_im * dl_re;
            // bt_ma1.UpdateSigned (TMovingAverage<complex>, DSP.h:184-192)
            ma1_re = ma1_re - o_m1r[k]; ma1_im = ma1_im - o_m1i[k];
            ma1_re = ma1_re + pr; ma1_im = ma1_im + pi;
            ma1r[(size_t)s_ma1 * 64] = pr; ma1i[(size_t)s_ma1 * 64] = pi;
            double fastarm = hypot(ma1_re / ma1_len_d, ma1_im / ma1_len_d);
            // mav1->UpdateSigned
            mav1_sum = mav1_sum - o_mav[k];
            mav1_sum = mav1_sum + fastarm;
            mav1[(size_t)s_mav1 * 64] = fastarm;
            fastarm = mav1_sum / mav1_len_d;
            // fastarm -= bt_ma_diff.update(fastarm)
            fa[(size_t)s_fa * 64] = fastarm;
            fastarm -= (faw * fa_new[k] + fawc * fa_old[k]);
            if (fastarm < 0) fastarm = 0;
            double bt_sig = fastarm * fastarm;
            if (bt_sig > 500) bt_sig = 500;
            // PeakDetector::update (DSP.h:528-560)
            bt[(size_t)s_bt * 64] = bt_sig;
            const double dy = bt_sig - bt_2[k];
            const double vald = bt_1[k];
            // d3.findmaxpos: first strict maximum scanning oldest -> newest over 2*PL+1 entries of the lane's peak-detector ring.  Done by
            // the whole wavefront for each lane that triggered (ring positions are wave-uniform): 19 strided loads + a shuffle reduction
            // instead of 1171 dependent loads under one active lane -- with ~2.7 triggers per wavefront and 2048-sample segment that
            // serial scan was most of this kernel's time (8.5 ms average against 3.2 ms for a segment without triggers)
            const bool trig = (!cntdown) && (vald > g.pd_thr) && (lastdy >= 0 && dy < 0);
            unsigned long long tm = __ballot(trig);
            if (tm)
            {
                const int pos0 = bback(s_bt, twoPL, g.bt_len);
                // the ring this kernel also WRITES (through bt, this iteration's entry included): a plain pointer, and the stores of all
                // lanes made visible to the wavefront before other lanes' columns are read
                const double *btg = p.bt + (size_t)grp * g.bt_len * 64;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this wavefront's own stores have reached memory order before the cross-lane reads
                while (tm)
                {
                    const int src = __ffsll((long long)tm) - 1;
                    tm &= tm - 1;
                    double bv = 0.0;
                    int bq = -1;
                    for (int q = lane; q <= twoPL; q += 64)
                    {
                        const double v = btg[(size_t)bwrap(pos0 + q, g.bt_len) * 64 + src];
                        if (bq < 0 || v > bv) { bv = v; bq = q; }
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1)
                    {
                        const double ov = __shfl_xor(bv, off, 64);
                        const int oq = __shfl_xor(bq, off, 64);
                        if (oq >= 0 && (bq < 0 || ov > bv || (ov == bv && oq < bq))) { bv = ov; bq = oq; }
                    }
                    if (lane == src) { cntdown = twoPL; maxposcd = bq; }
                }
            }
            if (cntdown > 0) cntdown--;
            lastdy = dy;
            bool fire = false;
            if (!maxposcd) { maxposcd--; fire = true; }
            else if (maxposcd > 0) maxposcd--;
            if (fire)
            {
                tri_ptr = 0;
                if (trace && ev_cnt < g.ev_cap)
                {
                    double *e = p.evlog + ((size_t)ch * g.ev_cap + ev_cnt) * 3;
                    e[0] = (double)(n0 + i); e[1] = BEV_PEAK; e[2] = 0; ev_cnt++;
                }
            }
            if (tri_ptr < g.tri_sz) tri_ptr++;
            else if (tri_ptr == g.tri_sz) { tri_ptr++; ev_pos = i; }
            s_agc = bwrap(s_agc + 1, g.agc_len); s_cv = bwrap(s_cv + 1, g.cv_len); s_ma1 = bwrap(s_ma1 + 1, g.ma1_len);
            s_mav1 = bwrap(s_mav1 + 1, g.mav1_len); s_fa = bwrap(s_fa + 1, g.fa_len); s_bt = bwrap(s_bt + 1, g.bt_len);
        }
    }
    BLDF(BS_AGC_SUM) = agc_sum; BLDF(BS_MA1_RE) = ma1_re; BLDF(BS_MA1_IM) = ma1_im; BLDF(BS_MAV1_SUM) = mav1_sum; BLDF(BS_LASTDY) = lastdy;
    BLDI(BI_CNTDOWN) = cntdown; BLDI(BI_MAXPOSCD) = maxposcd; BLDI(BI_TRI_PTR) = tri_ptr; BLDI(BI_EV_POS) = ev_pos; BLDI(BI_EV_CNT) = ev_cnt;
    (void)flags;
    if (HOLD) BLDI(BI_BT_HOLD) = hold;
    // which channels of this group filled their trident buffer in this segment: k_ev_compact turns the masks into the event list IN CHANNEL ORDER
    // (an atomic append listed them in arrival order, and k_trident's window reads -- a 64-byte sector per 8-byte sample, shared by eight
    // neighbouring channels -- found nothing of their neighbours' in L2)
    const unsigned long long em = __ballot(ev_pos >= 0 && ch < g.nch);
    if (lane == 0) p.ev_mask[grp] = em;
}

// the event masks of a segment -> the list of channels with a trident event, ascending, and their number.  One workgroup: a thread takes a run
// of consecutive groups, an LDS scan gives its place in the list.
__global__ __launch_bounds__(1024) void k_ev_compact(const unsigned long long *__restrict__ mask, int ngroups, int *__restrict__ ev_list, int *__restrict__ ev_count)
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (ngroups + 1023) / 1024;
    const int g0 = t * per, g1 = (g0 + per < ngroups) ? g0 + per : ngroups;
    int cnt = 0;
    for (int gq = g0; gq < g1; gq++) cnt += __popcll(mask[gq]);
    part[t] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1)
    {
        const int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int base = part[t] - cnt;
    for (int gq = g0; gq < g1; gq++)
    {
        unsigned long long m = mask[gq];
        while (m)
        {
            const int ln = __ffsll((long long)m) - 1;
            m &= m - 1;
            ev_list[base++] = gq * 64 + ln;
        }
    }
    if (t == 1023) *ev_count = part[1023];
}

// ------------------------------------------------------------------------------------------------ trident check
// The two fftr->transform calls (burstoqpskdemodulator.cpp:416-426, burstmskdemodulator.cpp:458-473) are 2^15-point FFTs of
// <= 2^14 real samples followed by zeros.  X[2q + r] = sum_n (x[n] W_32768^(n r)) W_16384^(n q): two 2^14-point transforms per
// window, run register-resident with wg_fft<14> (k_coarse2.h).  Only bins below N/2 are ever read by the reference.
#define TRI_N 32768
#define TRI_H 16384

// first maximum over the workgroup: the largest value, among equals the lowest index (what an ascending scan with a strict compare keeps);
// idx < 0 = "no candidate".  Wavefront reduction through shuffles, one LDS round for the eight wavefront results.  All threads get the result.
template <int NTHREADS>
__device__ __forceinline__ void tri_argmax_first(double &v, int &idx, double *red_val, int *red_idx, int t)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        const double ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
    __syncthreads(); // the previous reduction's results have been read
    if ((t & 63) == 0) { red_val[t >> 6] = v; red_idx[t >> 6] = idx; }
    __syncthreads();
    v = red_val[0]; idx = red_idx[0];
#pragma unroll
    for (int w = 1; w < NTHREADS / 64; w++)
    {
        const double ov = red_val[w];
        const int oi = red_idx[w];
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
}

// One workgroup per event (persistent over the device-side event list).  Round 3: the spectra never leave the chip (before, both went through
// a global scratch between barrier-separated loops: ~1 MB written and read per event, 54-94 GB per busy launch).  The eight 2^13-point
// transforms of an event (X[4q + r] for the base and the top window, r = 0 .. 3) leave bin k = 4 (s * 512 + t) + r of BOTH windows in the same
// register of the same thread.  The strongest base bin is tracked on the fly with its complex value (first maximum: ties go to the lower bin).
//   burst OQPSK: the trident sum d[k - b] + d[k + b] - d[k], d = |top| - |base|, has b = round(fb / 4 / hzperbin) = 1792 = 4 x 448 at the one
//     rate this demodulator exists for (10.5 kbps at 48 kHz; burst_create refuses anything else), so it only combines bins of ONE residue
//     class r: passes run base r, top r, the eight |base| values of a thread wait in registers, the class's 4096 differences go through
//     LDS (32 KiB behind the exchange buffer) and are searched at once.
//   burst MSK: d = |top| and the two searches are per-bin conditions around the strongest base bin: four base passes, the reduction, four top
//     passes whose candidates are tracked on the fly.  Nothing is stored at all.
// Round 4: 256 threads x 32 points on wg_fft13_e32 (k_coarse6.h: 32 x 16 x 16, two exchanges, natural order in and out) instead of 512 x 16
// on wg_fft<13> (16 x 32 x 16: half its threads idle in the middle pass, 74 spilled registers); two workgroups per CU (64 KiB of LDS each),
// the residue class's differences share the exchange buffer (it is idle between a transform's last read and the next one's first barrier).
#define TRI_THREADS 256
#define TRI_XCH C6_XCH13
template <bool oq> // burst OQPSK / burst MSK: the two searches keep different things across the transforms
__global__ __launch_bounds__(TRI_THREADS, 2) void k_trident(const BGeom g, const BPtrs p, long long n0)
{
    extern __shared__ __attribute__((aligned(16))) double xch[]; // TRI_XCH doubles
    __shared__ double red_val[TRI_THREADS / 64];
    __shared__ int red_idx[TRI_THREADS / 64];
    __shared__ double sh_bb[2];
    double *dl = xch; // one residue class of trident differences (4096 doubles), between two transforms
    const int t = threadIdx.x, nchp = g.nchp;
    const int nev = *p.ev_count;
    const double hzperbin = g.Fs / ((double)TRI_N);
    const int b = jd_qround((0.25 * g.fb) / hzperbin), b4 = b >> 2;       // OQPSK: 1792, 448
    const int psb = jd_qround((0.5 * g.fb) / hzperbin);                   // MSK
    // The list is in channel order and a window read fetches a sector that eight neighbouring channels share: workgroups that run at the same
    // time on the same XCD (one L2 each; workgroup b runs on XCD b mod 8) take NEIGHBOURING events -- XCD x the x-th eighth of the list.
    const int nx = gridDim.x < 8 ? 1 : 8;
    const int xcd = blockIdx.x % nx, wx = blockIdx.x / nx, nwx = (gridDim.x + nx - 1 - xcd) / nx; // this workgroup's rank among its XCD's, their number
    const int e_lo = (int)(((long long)nev * xcd) / nx), e_hi = (int)(((long long)nev * (xcd + 1)) / nx);
    for (int li = e_lo + wx; li < e_hi; li += nwx)
    {
        const int ch = p.ev_list[li];
        const int grp = ch >> 6, lane = ch & 63;
        const int evp = p.I[(size_t)BI_EV_POS * nchp + ch];
        // window: cval_d[e - tri_sz + k] = cv[e - tri_sz - D1 + k], e = n0 + evp
        const long long w0 = n0 + evp - g.tri_sz - g.D1 + 8LL * g.cv_len;
        const double *__restrict__ cvre = p.cvre + (size_t)grp * g.cv_len * 64 + lane;
        double mag[16];                         // OQPSK: |base[k]| of this thread's bins of the current class
        double bv = -1.0, bre = 0.0, bim = 0.0; // strongest base bin of this thread ...
        int bi = -1;
        double minval = 0.0;                    // ... and of the event (MSK: known after the fourth pass)
        int minvalbin = 0;
        double mv = 0.0; int mi = -1;           // OQPSK: largest trident sum
        double lv = 0.0; int lidx = -1;         // MSK: strongest top bin below / above the strongest base bin
        double hv = 0.0; int hidx = -1;
        // X[4q + r] = sum_{n<8192} (x[n] + x[n+8192] (-j)^r) W_32768^(n r) W_8192^(n q): four 2^13-point transforms per window (the
        // windows are <= 2^14 samples followed by zeros).  2^13 points over 256 threads = 32 per thread: n = s * 256 + t on entry,
        // q = s * 256 + t on exit.  Only bins below N/2 are needed: q < 4096 (slots s < 16).
        for (int pass = 0; pass < 8; pass++)
        {
            const int which = oq ? (pass & 1) : (pass >> 2), r = oq ? (pass >> 1) : (pass & 3); // which: 0 base, 1 top
            const int off = which ? g.nb : 0, len = which ? g.nt : g.nb;
            if (!oq && pass == 4)
            {
                minval = bv; minvalbin = bi;
                tri_argmax_first<TRI_THREADS>(minval, minvalbin, red_val, red_idx, t);
                if (bi == minvalbin) { sh_bb[0] = bre; sh_bb[1] = bim; } // exactly one thread holds that bin
                if (!(minval > 0.0)) minvalbin = 0; // MSK starts from minval = 0 with a strict compare
            }
            CV<32> d;
            const int slot0 = (int)((w0 + off) % g.cv_len); // wave-uniform; off + len <= tri_sz < cv_len: at most one wrap below
#pragma unroll
            for (int s = 0; s < 32; s++)
            {
                const int n = s * TRI_THREADS + t;
                // rows past the window are zeros whatever their twiddle: at 10.5 kbps / 1200 bps the windows (1170 / 5040, 2960 samples) fill
                // 5 / 20, 12 of the 32 rows, and the table loads and products of the others were most of this loop
                if (s * TRI_THREADS >= len) { d.r[s] = 0.0; d.i[s] = 0.0; continue; }
                double x0 = 0.0, x1 = 0.0;
                if (n < len)
                {
                    int sl = slot0 + n; if (sl >= g.cv_len) sl -= g.cv_len;
                    x0 = cvre[(size_t)sl * 64];
                }
                if (len > 8192 && n + 8192 < len) // burst MSK 600: the 12000-sample base window folds once
                {
                    int sl = slot0 + n + 8192; while (sl >= g.cv_len) sl -= g.cv_len;
                    x1 = cvre[(size_t)sl * 64];
                }
                // (x0 + x1 (-j)^r) W_32768^(n r)
                double fr = x0, fi = 0.0;
                if (r == 0) fr += x1; else if (r == 1) fi -= x1; else if (r == 2) fr -= x1; else fi += x1;
                if (r == 0) { d.r[s] = fr; d.i[s] = fi; }
                else
                {
                    const int m = n * r;                          // < 3 * 8192
                    double2 w = p.tw15[m & (TRI_H - 1)];          // W_32768^m for m < 16384; W^(m+16384) = -W^m
                    if (m & TRI_H) { w.x = -w.x; w.y = -w.y; }
                    d.r[s] = fr * w.x - fi * w.y; d.i[s] = fr * w.y + fi * w.x;
                }
            }
            wg_fft13_e32(d, xch, p.tw14, t);
            // thread t slot s holds X[4 (s*256 + t) + r]
            if (which == 0)
            {
#pragma unroll
                for (int s = 0; s < 16; s++)
                {
                    const int k = 4 * (s * TRI_THREADS + t) + r;
                    const double a = hypot(d.r[s], d.i[s]);
                    mag[s] = a;
                    if (a > bv || (a == bv && k < bi)) { bv = a; bi = k; bre = d.r[s]; bim = d.i[s]; }
                }
            }
            else if (oq)
            {
                __syncthreads(); // every thread has read its last exchange values: the buffer takes the class's differences
#pragma unroll
                for (int s = 0; s < 16; s++) dl[s * TRI_THREADS + t] = hypot(d.r[s], d.i[s]) - mag[s]; // d[4 q + r] at q
                __syncthreads();
                // firstbin = b <= k < lstbin = N/2 - b, k = 4 q + r:  b4 <= q < 4096 - b4 for every r (b is a multiple of 4)
                for (int q = b4 + t; q < (TRI_H >> 2) - b4; q += TRI_THREADS)
                {
                    const double tv = dl[q - b4] + dl[q + b4] - dl[q];
                    const int k = 4 * q + r;
                    if (mi < 0 || tv > mv || (tv == mv && k < mi)) { mv = tv; mi = k; }
                }
            }
            else
            {
#pragma unroll
                for (int s = 0; s < 16; s++)
                {
                    const int k = 4 * (s * TRI_THREADS + t) + r;
                    if (k > 50)
                    {
                        const double a = hypot(d.r[s], d.i[s]);
                        if ((k < minvalbin - (psb / 2)) && (a > lv || (a == lv && lidx >= 0 && k < lidx))) { lv = a; lidx = k; }
                        if ((k > minvalbin + (psb / 2)) && (a > hv || (a == hv && hidx >= 0 && k < hidx))) { hv = a; hidx = k; }
                    }
                }
            }
        }
        TriResult res;
        res.pad = 0;
        if (oq)
        {
            minval = bv; minvalbin = bi;
            tri_argmax_first<TRI_THREADS>(minval, minvalbin, red_val, red_idx, t);
            if (bi == minvalbin) { sh_bb[0] = bre; sh_bb[1] = bim; }
            if (!(minval > 0.0)) minvalbin = 0;
            tri_argmax_first<TRI_THREADS>(mv, mi, red_val, red_idx, t); // (its barriers also publish sh_bb)
            const double maxval = mv;
            const int maxvalbin = mi;
            res.ok = (maxval > 500.0) && (fabs((((double)(maxvalbin - minvalbin))) * hzperbin) < 20.0);
            // with an all-zero base window base[0] = 0 stands in, as in the reference
            const double bbx = (minval > 0.0) ? sh_bb[0] : 0.0, bby = (minval > 0.0) ? sh_bb[1] : 0.0;
            const double carrierphase = atan2(bby, bbx) - (M_PI / 4.0);
            res.freq = hzperbin * (double)minvalbin;
            res.phase_deg = (180.0 / M_PI) * carrierphase;
            res.vol_gain = 1.4142 * 500.0 / minval;
            res.metric = maxval;
        }
        else
        {
            tri_argmax_first<TRI_THREADS>(lv, lidx, red_val, red_idx, t);
            const int maxtoppos = (lidx >= 0) ? lidx : 0;
            tri_argmax_first<TRI_THREADS>(hv, hidx, red_val, red_idx, t);
            const int maxtopposhigh = (hidx >= 0) ? hidx : 0;
            const int distfrompeak = abs(maxtoppos - minvalbin);
            res.ok = (minval > 500.0) && (abs(distfrompeak - psb) < abs(psb / 20));
            const double bbx = (minval > 0.0) ? sh_bb[0] : 0.0, bby = (minval > 0.0) ? sh_bb[1] : 0.0;
            const double carrierphase = atan2(bby, bbx) - (M_PI / 4.0);
            res.freq = ((double)((maxtopposhigh + maxtoppos) / 2)) * hzperbin;
            res.phase_deg = (180.0 / M_PI) * carrierphase;
            res.vol_gain = 1.4142 * (500.0 / (minval / 3));
            res.metric = minval;
        }
        if (t == 0) p.tri[ch] = res;
        __syncthreads(); // sh_bb and the reduction words are free for the next event
    }
}
