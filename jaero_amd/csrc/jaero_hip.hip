// jaero_hip.hip -- libjaero_hip.so: C ABI (include/jaero_hip.h) + host-side bank scheduler for the gfx950 kernels.
//
// Host responsibilities (everything numerical happens in the kernels):
//   * own the per-bank device state (layout: jaero_device.h),
//   * split each jaero_write at the samples where the reference runs its coarse-frequency estimate synchronously
//     inside writeData (bbcycbuff_ptr % (bbnfft/4) == 0, JAERO/oqpskdemodulator.cpp:416) -- a host mirror of the two
//     integer counters involved (bbcycbuff_ptr, coarseCounter) predicts those samples exactly, so no device->host
//     round trip is needed in steady state,
//   * launch [sample kernel] [coarse kernel] [sample kernel] ... on the caller's stream.
// No CPU fallback exists: without a HIP device jaero_create fails.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/jaero_hip.h"
#include "jaero_device.h"
#include "k_oqpsk_fb.h"
#include "k_msk.h"
#include "k_msk_fb.h"
#include "k_pre8400.h"
#include "k_coarse.h"
#include "k_coarse2.h"
#include "k_coarse6.h"
#include "k_viterbi.h"
#include "k_viterbi_lanes.h"
#include "burst_device.h"
#include "k_burst_front.h"
#include "k_burst_demod.h"
#include "k_burst_msk_fb.h"
#include "k_burst_settings.h"
#include "k_aerol.h"
#include "k_aerol_burst.h"

static thread_local std::string g_last_error;
static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
#define HIPCHK(x)                                                                                        \
    do {                                                                                                 \
        hipError_t _e = (x);                                                                             \
        if (_e != hipSuccess) return fail(JAERO_EHIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// right behind a kernel launch inside jaero_write: a launch that was refused (bad configuration, missing LDS attribute) is reported by
// name at the launch that failed, not as a stale error at the end of the call
#define LAUNCHCHK(what)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = hipGetLastError();                                                               \
        if (_e != hipSuccess) return fail(JAERO_EHIP, "launch of %s failed: %s (%s:%d)", what, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define OQ_LDSN 39 // matched-filter history slots kept in LDS (rest in VGPRs) + the taps: 39.5 KiB per wavefront -> 4 wavefronts per CU
#define MSK_LDSN_1200 39 // of 80 taps: four wavefronts per CU (rings + the wavefront's copy of the taps: 39.6 KiB)
#define MSK_LDSN_40 24   // of 40 taps (1200 bps at 24 kHz, 600 bps at 12 kHz)
#define MSK_LDSN_20 12   // of 20 taps (1200 bps at 12 kHz)

struct ProfSlot { double ms = 0; int launches = 0; };

// Host mirror of the two integer counters that decide WHEN the reference runs its coarse-frequency estimate
// (bbcycbuff_ptr and coarseCounter, JAERO/oqpskdemodulator.cpp:410-431 == JAERO/mskdemodulator.cpp:350-368).
// A "segment" is a run of samples handed to one sample-kernel launch; it ends at the first sample (over all channels)
// whose ring write makes bbcycbuff_ptr % (cpuReduce ? nfft : nfft/4) == 0.  For that sample only the ring write
// ("A-part") is done; the rest of the sample ("B-part") runs in the next segment, after the coarse kernel.
struct Mirror
{
    int nch = 0, nchp = 0, nfft = 0, Fs_int = 0;
    std::vector<int> flags, bbptr, cnt;
    int pending = 0;        // A-part of the next sample already done
    long long nB_total = 0; // B-parts executed so far (uniform ring slots derive from it)
    std::vector<int> fired; // channels whose estimate fires at the end of the last planned segment

    // number of A-steps (>=1) until channel ch fires, given its coarseCounter before its next A-step
    long long steps_to_trigger(int ch, int cnt_before_first_a) const
    {
        if (!(flags[ch] & JF_CPUREDUCE))
        {
            const int q = nfft / 4;
            return q - (bbptr[ch] % q);
        }
        long long j0 = (long long)Fs_int - cnt_before_first_a + 1;
        if (j0 < 1) j0 = 1;
        const int r = nfft - (bbptr[ch] % nfft);
        return j0 + r - 1;
    }
    // Plans the segment starting at sample `pos` of a write of `nsamples`; returns its length n (samples pos..pos+n-1),
    // sets skip_a/only_a, fills `fired`, updates the counters, and returns the position the next segment starts at.
    int next_segment(int pos, int nsamples, int &n, int &skip_a, int &only_a)
    {
        long long dmin = (long long)1 << 60;
        for (int ch = 0; ch < nch; ch++)
        {
            const long long d = steps_to_trigger(ch, cnt[ch] + (pending ? 1 : 0));
            if (d < dmin) dmin = d;
        }
        const int first_a = pos + (pending ? 1 : 0);
        const long long trig_sample = (long long)first_a + dmin - 1;
        const bool trig = trig_sample < nsamples;
        const int end = trig ? (int)trig_sample + 1 : nsamples;
        n = end - pos;
        skip_a = pending;
        only_a = trig ? 1 : 0;
        const int acount = n - (pending ? 1 : 0);
        const int bcount = n - (trig ? 1 : 0);
        fired.clear();
        for (int ch = 0; ch < nchp; ch++)
        {
            const int cnt0 = cnt[ch] + (pending ? 1 : 0);
            long long fills;
            if (!(flags[ch] & JF_CPUREDUCE)) fills = acount;
            else
            {
                long long j0 = (long long)Fs_int - cnt0 + 1;
                if (j0 < 1) j0 = 1;
                fills = (long long)acount - (j0 - 1);
                if (fills < 0) fills = 0;
            }
            const bool f = trig && ch < nch && steps_to_trigger(ch, cnt0) == dmin;
            bbptr[ch] = (int)((bbptr[ch] + fills) % nfft);
            cnt[ch] += bcount;
            if (f) { cnt[ch] = 0; fired.push_back(ch); }
        }
        nB_total += bcount;
        pending = trig ? 1 : 0;
        return trig ? end - 1 : end;
    }
};

struct jaero_ctx
{
    int device = 0;
    JGeom g{};
    JPtrs p{};
    unsigned flags = 0;
    int max_write = 0;
    int soft_cap_req = 0; // softbit_capacity as given to jaero_create (0 = default for the rate): a re-created bank asks for the same
    std::vector<void *> allocs;
    // device helpers
    int16_t *d_pcm_frames = nullptr; // [max_write][nchp] staging for channel-major / host input
    int16_t *d_pcm_raw = nullptr;    // [nch*max_write] staging for host input
    double2 *d_tw = nullptr;
    int *d_emitted = nullptr; // burst banks: per-channel count of soft bits already emitted (jaero_softbits_view)
    int msk_ldsn = 0; // MSK: matched-filter inputs kept in LDS (the rest of fir_n in registers)
    std::vector<int> dly_t0; // MSK: shared delay-line slot at which each channel's delayedsmpl pointer last restarted (jaero_set_settings)
    int msk_pairs = 0; // MSK with an 80-tap filter: front/back pairs per workgroup of k_msk_fb (0 = k_msk_samples)
    int oq_pairs = 0; // 10.5 kbps OQPSK: front/back pairs per workgroup of k_oqpsk_fb (0 = the single-wavefront kernel k_oqpsk_samples)
    int oq_ldsn = OQ_LDSN;
    JTaps28 oq_taps{}; // the 28 distinct values of the (bitwise symmetric) 55-tap RRC, scalar operands of k_oqpsk_fb's filter
    // fb == 8400 (k_pre8400.h): prefilter buffers, samples written so far, size of the previous write
    bool pre8400 = false, pre_direct = false;
    JPre pre{};
    long long pre_n0 = 0;
    int pre_nprev = 0;
    int *d_chanlist = nullptr;
    int coarse2_grid = 0; // workgroups of the persistent coarse-estimate kernels = CUs (k_coarse6_13: twice that)
    jaero_status *d_status = nullptr;
    int16_t *d_pack = nullptr; size_t pack_elems = 0;
    // host mirrors
    std::vector<jaero_settings> settings;
    Mirror m;
    // burst kinds (JAERO_KIND_BURST_*): their own geometry / state (burst_device.h); g/p above stay unused
    bool burst = false;
    BGeom bg{};
    BPtrs bp{};
    int tri_grid = 0, tri_lds = 0;
    long long nsamples_total = 0; // samples written so far (uniform ring slots and event time stamps derive from it)
    int bt_hold_left = 0;         // burst: samples behind the last jaero_set_settings for which k_burst_front<true> must run (bt_d1 refilled with zeros)
    // generic views of the per-channel output buffers (either kind)
    int o_nchp = 0, o_nch = 0, o_soft_cap = 0, o_sym_cap = 0;
    int16_t *o_soft = nullptr; double *o_sym = nullptr;
    int *o_soft_cnt = nullptr, *o_sym_cnt = nullptr, *o_overflow = nullptr, *o_flags = nullptr;
    int *o_nrx = nullptr; // burst: soft bits pushed but not yet emitted (RxDataBits.size()), kept at the tail of the buffer
    // profiling
    bool prof = false;
    ProfSlot slots[5];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    struct EvUse { int which; int idx; };
    std::vector<EvUse> ev_used;
    size_t ev_next = 0;
    hipStream_t last_stream = nullptr;
    hipEvent_t order_ev = nullptr; // a write on another stream than the previous one waits for what was enqueued on that one (setters included)
    bool poisoned = false;         // a launch inside a write failed: the host's schedule mirror has advanced past the device state
};

// ------------------------------------------------------------------------------------------ small kernels
__global__ void k_transpose_pcm(const int16_t *__restrict__ src /*[nch][n]*/, int16_t *__restrict__ dst /*[n][nchp]*/, int nch,
                                int nchp, int n)
{
    __shared__ int16_t tile[64][65];
    const int c0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 256 threads: 4 rows per pass
    for (int r = ty; r < 64; r += 4)
    {
        const int c = c0 + r, i = i0 + tx;
        tile[r][tx] = (c < nch && i < n) ? src[(size_t)c * n + i] : (int16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4)
    {
        const int i = i0 + r, c = c0 + tx;
        if (i < n && c < nchp) dst[(size_t)i * nchp + c] = tile[tx][r];
    }
}

__global__ void k_center_freq(const JGeom g, const JPtrs p, int ch_first, int nch_apply, double freq_center_in)
{
    // CenterFreqChangedSlot (oqpskdemodulator.cpp:291-310, mskdemodulator.cpp:265-282); one block per channel
    const int ch = ch_first + blockIdx.x;
    if (blockIdx.x >= nch_apply) return;
    const int nchp = g.nchp;
    if (threadIdx.x == 0)
    {
        double fc = freq_center_in;
        const double fb = g.fb, Fs = g.Fs;
        if (g.kind == 1 && g.fb == 8400) {} // no clamp at 8400 bps (oqpskdemodulator.cpp:293: `if(fb!=8400)`)
        else if (g.kind == 1)
        {
            if (fc < (0.5 * fb)) fc = 0.5 * fb;
            if (fc > (Fs / 2.0 - 0.5 * fb)) fc = Fs / 2.0 - 0.5 * fb;
        }
        else
        {
            if (fc < (0.75 * fb)) fc = 0.75 * fb;
            if (fc > (Fs / 2.0 - 0.75 * fb)) fc = Fs / 2.0 - 0.75 * fb;
        }
        double *S = p.S + ch;
        const int flags = p.I[(size_t)I_FLAGS * nchp + ch];
        const double lbw = S[(size_t)S_LOCKINGBW * nchp];
        // mixer_center.SetFreq(freq_center,Fs)  (DSP.cpp:142-149)
        double mc_freq = fc;
        if (mc_freq < 0) mc_freq = 0;
        double mc_step = (mc_freq) * ((double)JD_WTSIZE) / ((float)Fs);
        double mc_ptr = S[(size_t)S_MC_PTR * nchp];
        while (((int)mc_ptr) >= JD_WTSIZE) mc_ptr -= JD_WTSIZE;
        double m2_freq = S[(size_t)S_M2_FREQ * nchp], m2_step = S[(size_t)S_M2_STEP * nchp];
        if (flags & JF_AFC) jd_wt_setfreq(m2_freq, m2_step, mc_freq, Fs);
        if ((m2_freq - mc_freq) > (lbw / 2.0)) jd_wt_setfreq(m2_freq, m2_step, mc_freq + (lbw / 2.0), Fs);
        if ((m2_freq - mc_freq) < (-lbw / 2.0)) jd_wt_setfreq(m2_freq, m2_step, mc_freq - (lbw / 2.0), Fs);
        S[(size_t)S_MC_FREQ * nchp] = mc_freq; S[(size_t)S_MC_STEP * nchp] = mc_step; S[(size_t)S_MC_PTR * nchp] = mc_ptr;
        S[(size_t)S_M2_FREQ * nchp] = m2_freq; S[(size_t)S_M2_STEP * nchp] = m2_step;
    }
    double2 *ring = p.bbring + (size_t)ch * g.nfft;
    for (int i = threadIdx.x; i < g.nfft; i += blockDim.x) ring[i] = make_double2(0.0, 0.0); // bbcycbuff[j]=0
}

__global__ void k_status(const JGeom g, const JPtrs p, int ch_first, int n, jaero_status *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int ch = ch_first + k, nchp = g.nchp;
    jaero_status st;
    st.mse = p.S[(size_t)S_MSE * nchp + ch];
    st.ebno = p.S[(size_t)S_EB_EBNO * nchp + ch];
    st.freq_est = p.S[(size_t)S_M2_FREQ * nchp + ch];
    st.freq_center = p.S[(size_t)S_MC_FREQ * nchp + ch];
    st.signal = (st.mse > p.S[(size_t)S_THRESH * nchp + ch]) ? 0 : 1;
    st.n_estimates = p.I[(size_t)I_NEST * nchp + ch];
    out[k] = st;
}

__global__ void k_pack_soft(const int *__restrict__ soft_cnt, const int16_t *__restrict__ soft, int soft_cap, int16_t *dst, int capc)
{
    const int ch = blockIdx.x;
    const int cnt = min(soft_cnt[ch], capc);
    const int16_t *src = soft + (size_t)ch * soft_cap;
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) dst[(size_t)ch * capc + k] = src[k];
}

__global__ void k_status_burst(const BGeom g, const BPtrs p, int ch_first, int n, jaero_status *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int ch = ch_first + k, nchp = g.nchp;
    jaero_status st;
    st.mse = p.S[(size_t)BS_MSE * nchp + ch];
    st.ebno = p.S[(size_t)BS_EB_EBNO * nchp + ch];
    st.freq_est = p.S[(size_t)BS_M2_FREQ * nchp + ch];
    st.freq_center = (g.kind == JAERO_KIND_BURST_MSK_D) ? p.S[(size_t)BS_MC_FREQ * nchp + ch] : st.freq_est;
    st.signal = p.I[(size_t)BI_STARTSTOP * nchp + ch] > 0 ? 1 : 0; // between SignalStatus(true) and SignalStatus(false)
    st.n_estimates = p.I[(size_t)BI_EV_CNT * nchp + ch];
    out[k] = st;
}

__global__ void k_fill_int(int *p, int n, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------ helpers
template <class T>
static int dalloc(jaero_ctx *c, T **ptr, size_t count, bool zero = true)
{
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(JAERO_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    if (zero) { e = hipMemset(q, 0, bytes); if (e != hipSuccess) return fail(JAERO_EHIP, "hipMemset failed: %s", hipGetErrorString(e)); }
    c->allocs.push_back(q);
    *ptr = (T *)q;
    return 0;
}

// RootRaisedCosine::design (JAERO/DSP.h:316-338)
static std::vector<double> rrc_design(double alpha, int firsize, double samplerate, double symbol_freq)
{
    if ((firsize % 2) == 0) firsize += 1;
    std::vector<double> P(firsize);
    double T = (samplerate) / (symbol_freq);
    for (int i = 0; i < firsize; i++)
    {
        if (i == ((firsize - 1) / 2)) P[i] = (4.0 * alpha + M_PI - M_PI * alpha) / (M_PI * sqrt(T));
        else
        {
            double fi = (((double)i) - ((double)(firsize - 1)) / 2.0);
            if (fabs(1.0 - pow(4.0 * alpha * fi / T, 2)) < 0.0000000001)
                P[i] = (alpha * ((M_PI - 2.0) * cos(M_PI / (4.0 * alpha)) + (M_PI + 2.0) * sin(M_PI / (4.0 * alpha))) / (M_PI * sqrt(2.0 * T)));
            else
                P[i] = (4.0 * alpha / (M_PI * sqrt(T)) * (cos((1.0 + alpha) * M_PI * fi / T) + T / (4.0 * alpha * fi) * sin((1.0 - alpha) * M_PI * fi / T)) / (1.0 - pow(4.0 * alpha * fi / T, 2)));
        }
    }
    return P;
}

// weighting used by Delay<double>::update (JAERO/DSP.h:357-374) for ring position 0
static double delay_weight(double fractdelay)
{
    int size = (int)ceil(fractdelay) + 1;
    double dptr = 0.0 - fractdelay;
    while (floor(dptr) < 0) dptr += (double)size;
    int iptr = (int)floor(dptr);
    return dptr - (double)iptr;
}

static int validate_settings(const jaero_settings &s)
{
    if (s.kind < JAERO_KIND_MSK || s.kind > JAERO_KIND_BURST_OQPSK) return fail(JAERO_ENOTSUP, "kind %d not implemented", s.kind);
    const bool oq = s.kind == JAERO_KIND_OQPSK || s.kind == JAERO_KIND_BURST_OQPSK;
    // the continuous MSK demodulator also runs at 24 and 12 kHz (MskDemodulator::dataReceived re-applies its settings with the sample rate
    // of the incoming audio, JAERO/mskdemodulator.cpp:528-537): matched filters of 2 Fs / fb = 20, 40, 80 or 160 taps
    const bool fs_ok = s.Fs == 48000 || (s.kind == JAERO_KIND_MSK && (s.Fs == 24000 || s.Fs == 12000));
    if (!fs_ok) return fail(JAERO_ENOTSUP, "Fs = %g is not implemented (48000; continuous MSK also 24000 and 12000)", s.Fs);
    // fb = 8400 (C channel): the continuous demodulator with a 2^14-point coarse FFT (k_pre8400.h)
    const bool allow84 = s.kind == JAERO_KIND_OQPSK && s.coarsefreqest_fft_power == 14;
    if (oq && s.fb != 10500 && !(s.fb == 8400 && allow84))
        return fail(JAERO_ENOTSUP, "OQPSK: fb must be 10500, or 8400 for the continuous demodulator with coarsefreqest_fft_power 14");
    if (!oq && s.fb != 600 && s.fb != 1200) return fail(JAERO_ENOTSUP, "MSK: fb must be 600 or 1200");
    const bool burst = s.kind >= JAERO_KIND_BURST_MSK;
    if (!burst && s.coarsefreqest_fft_power != 13 && s.coarsefreqest_fft_power != 14) return fail(JAERO_ENOTSUP, "coarsefreqest_fft_power must be 13 or 14");
    if (!(s.lockingbw > 0) || !(s.freq_center >= 0)) return fail(JAERO_EINVAL, "bad lockingbw/freq_center");
    return 0;
}

// per-channel scalar initial state = constructor + setSettings of the reference
// stride = the column pitch of S / I (the bank's nchp, or 1 for a one-column scratch with ch = 0: jaero_set_settings)
static void init_channel_scalars(const jaero_ctx *c, const jaero_settings &s, std::vector<double> &S, std::vector<int> &I, int ch, bool fresh, int stride = 0)
{
    const int nchp = stride > 0 ? stride : c->g.nchp;
    auto SS = [&](int f) -> double & { return S[(size_t)f * nchp + ch]; };
    auto II = [&](int f) -> int & { return I[(size_t)f * nchp + ch]; };
    double fc = s.freq_center;
    if (fc > ((s.Fs / 2.0) - (s.lockingbw / 2.0))) fc = ((s.Fs / 2.0) - (s.lockingbw / 2.0));
    const double step_fc = (fc < 0 ? 0 : fc) * ((double)JD_WTSIZE) / ((float)(double)(int)s.Fs);
    SS(S_M2_FREQ) = fc < 0 ? 0 : fc; SS(S_M2_STEP) = step_fc;
    SS(S_MC_FREQ) = fc < 0 ? 0 : fc; SS(S_MC_STEP) = step_fc;
    const double stf = (s.kind == JAERO_KIND_OQPSK) ? s.fb : s.fb / 2;
    SS(S_ST_FREQ) = stf; SS(S_ST_STEP) = stf * ((double)JD_WTSIZE) / ((float)(double)(int)s.Fs);
    SS(S_LOCKINGBW) = s.lockingbw; SS(S_THRESH) = s.signalthreshold;
    if (fresh)
    {
        SS(S_PRE_PTR) = 0; SS(S_PRE_FSUM) = 0;
        SS(S_PRE_STEP) = 8000.0 * ((double)JD_WTSIZE) / ((float)(double)(int)s.Fs); // mixer_fir_pre.SetFreq(freq_center, Fs) in the ctor: 8000 Hz
        SS(S_M2_PTR) = 0; SS(S_MC_PTR) = 0; SS(S_ST_PTR) = 0; SS(S_ST_LAST) = 0;
        SS(S_MSE) = (s.kind == JAERO_KIND_OQPSK) ? 100.0 : 10.0;
        SS(S_DIFF_LAST) = -1.0;
        II(I_COUNTDOWN) = 4; II(I_COUNTDOWN2) = 5; II(I_EMPTYING) = 1;
    }
    if (s.kind == JAERO_KIND_MSK) SS(S_MSE) = 10.0; // setSettings resets mse (mskdemodulator.cpp:180)
}

static void fill_geometry(JGeom &g, const jaero_settings &s, int nch, unsigned flags)
{
    memset(&g, 0, sizeof g);
    g.kind = s.kind; g.nch = nch; g.nchp = (nch + 63) / 64 * 64; g.ngroups = g.nchp / 64;
    g.Fs = s.Fs; g.fb = s.fb; g.Fs_int = (int)s.Fs;
    g.nfft_log2 = s.coarsefreqest_fft_power; g.nfft = 1 << g.nfft_log2;
    g.ebno_len = (int)(2 * s.Fs);
    g.flags = flags;
    if (s.kind == JAERO_KIND_OQPSK)
    {
        g.fir_n = 55;
        g.agc_len = (int)round(4 * s.Fs);
        g.marg_len = 800; g.dt_len = 401; g.pm_len = 400; g.msema_len = 400;
        g.ee = 0.4;
        double T = s.Fs / (s.fb / 2);
        g.w4 = delay_weight(T / 4.0); g.w8 = delay_weight(T / 8.0);
        g.res_b0 = 0.00032714218939589035; g.res_b1 = 0; g.res_b2 = 0.00032714218939589035;
        g.res_a1 = -0.39005299948210803; g.res_a2 = 0.99934571562120822;
        if (s.fb == 8400) // the 10 Hz resonator and ee of oqpskdemodulator.cpp:231-251
        {
            g.res_b0 = 0.0012845857864470789; g.res_b1 = 0; g.res_b2 = -0.0012845857864470789;
            g.res_a1 = -0.90681461999279889; g.res_a2 = 0.99743082842710584;
            g.ee = 0.65;
        }
        g.lf_b0 = 0.0010275610653672064; g.lf_b1 = 0.0020551221307344128; g.lf_b2 = 0.0010275610653672064;
        g.lf_a1 = -1.9207386815577139; g.lf_a2 = 0.92509247310306331;
        g.stref_freq = s.fb;
        g.sps = 1; g.sps2 = 1;
    }
    else
    {
        g.sps = (int)(s.Fs / s.fb); g.sps2 = g.sps / 2;
        g.fir_n = 2 * g.sps;
        g.agc_len = (int)round(1 * s.Fs);
        g.marg_len = g.sps; g.dt_len = g.sps / 2 + 1; g.pm_len = 1; g.msema_len = 600;
        // resonator and sampling-point offset: mskdemodulator.cpp:196-233 (one resonator for every rate other than 48 kHz)
        if (s.fb >= 1200)
        {
            g.correctionfactor = 0.6;
            g.res_a1 = -1.993312819378528; g.res_a2 = 0.999476538254407;
            g.res_b0 = 2.617308727964618e-04; g.res_b1 = 0; g.res_b2 = -2.617308727964618e-04;
            g.ee = 0.025;
            if (s.Fs != 48000) g.ee = 0.05;
        }
        else
        {
            g.correctionfactor = 1.0;
            g.res_a1 = -1.998196509168551; g.res_a2 = 0.999738234875681;
            g.res_b0 = 1.308825621597620e-04; g.res_b1 = 0; g.res_b2 = -1.308825621597620e-04;
            g.ee = 0.025;
            if (s.Fs != 48000) g.ee = 0.0125;
        }
        if (s.Fs != 48000)
        {
            g.res_a1 = -1.974342917561558; g.res_a2 = 0.998953350377616;
            g.res_b0 = 5.233248111921052e-04; g.res_b1 = 0; g.res_b2 = -5.233248111921052e-04;
        }
        g.stref_freq = s.fb / 2;
    }
    g.win_len = ((flags & JAERO_FLAG_EBNO) && g.ebno_len > g.agc_len) ? g.ebno_len : g.agc_len;
}

// Tables of the overlap-save filters (k_pre8400_fft, k_hilbert_fft): H = DFT_4096(taps, zero-padded) / 4096 and exp(-2 pi i k / 4096),
// summed in long double on the host.  kernel_fn = the kernel whose dynamic LDS limit is raised to the 128 KiB exchange buffer.
static int fft4096_tables(const std::vector<double> &taps, double2 **d_H, double2 **d_tw, const void *kernel_fn)
{
    const int N = 2 * PRE_L;
    std::vector<long double> cr(N), ci(N);
    for (int k = 0; k < N; k++) { const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)N; cr[k] = cosl(a); ci[k] = sinl(a); }
    std::vector<double2> H(N), tw(N);
    for (int k = 0; k < N; k++)
    {
        long double sr = 0, si = 0;
        for (int j = 0; j < (int)taps.size(); j++) { const int e = (int)(((long long)k * j) & (N - 1)); sr += (long double)taps[j] * cr[e]; si += (long double)taps[j] * ci[e]; }
        H[k].x = (double)(sr / N); H[k].y = (double)(si / N);
        tw[k].x = (double)cr[k]; tw[k].y = (double)ci[k];
    }
    HIPCHK(hipMalloc((void **)d_H, sizeof(double2) * N));
    HIPCHK(hipMalloc((void **)d_tw, sizeof(double2) * N));
    HIPCHK(hipMemcpy(*d_H, H.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(*d_tw, tw.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
    HIPCHK(hipFuncSetAttribute(kernel_fn, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * PRE_L * (int)sizeof(double)));
    return 0;
}

#include "burst_host.h"

// ------------------------------------------------------------------------------------------ create / destroy
extern "C" int jaero_abi_version(void) { return JAERO_ABI_VERSION; }
extern "C" const char *jaero_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *jaero_strerror(int code)
{
    switch (code)
    {
    case JAERO_OK: return "ok";
    case JAERO_EINVAL: return "invalid argument";
    case JAERO_ENODEV: return "no usable HIP device";
    case JAERO_ENOMEM: return "out of memory";
    case JAERO_EHIP: return "HIP runtime error";
    case JAERO_EOVERFLOW: return "output capacity exceeded";
    case JAERO_ENOTSUP: return "not supported";
    default: return "unknown error";
    }
}
extern "C" int jaero_num_channels(const jaero_ctx *ctx) { return ctx ? ctx->o_nch : 0; }

extern "C" void jaero_destroy(jaero_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    // this bank's work only (its last stream and the default stream its control-plane copies use): other banks keep running until hipFree's
    // own implicit synchronisation, which cannot be avoided
    hipStreamSynchronize(c->last_stream);
    hipStreamSynchronize(0);
#ifdef FB_TRACE_BUILD
    if (!c->burst)
    {
        // trace build only: what the traced pair of this bank's sample loop accumulated, one JSON line on stderr per destroyed bank
        unsigned long long tr[20] = {0}, z[20] = {0};
        if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_fb_trace), sizeof(tr)) == hipSuccess && tr[8] + tr[18] > 0)
        {
            fprintf(stderr, "{\"fb_trace\": {\"channels\": %d, \"clock_hz\": 100000000", c->o_nch);
            for (int h = 0; h < 2; h++)
            {
                fprintf(stderr, ", \"%s\": {\"samples\": %llu, \"ticks\": [", h ? "back" : "front", tr[h * 10 + 8]);
                for (int k = 0; k < 8; k++) fprintf(stderr, "%s%llu", k ? ", " : "", tr[h * 10 + k]);
                fprintf(stderr, "]}");
            }
            fprintf(stderr, "}}\n");
        }
        hipMemcpyToSymbol(HIP_SYMBOL(g_fb_trace), z, sizeof(z));
    }
#endif
    for (void *q : c->allocs) hipFree(q);
    for (auto &e : c->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (c->order_ev) hipEventDestroy(c->order_ev);
    delete c;
}

static void launch_pre8400_filter(const JGeom &g, const JPtrs &p, const JPre &q, int n, long long n0, bool direct, hipStream_t st)
{
    (void)direct;
    hipLaunchKernelGGL(k_pre8400_fft, dim3(g.nchp / 4, (int)(((n0 + n - 1) >> 11) - (n0 >> 11) + 1)), dim3(PF_THREADS), 4 * 2 * PRE_L * (int)sizeof(double), st, g, p, q, n, n0);
}

extern "C" int jaero_create(int device, int nchannels, const jaero_settings *settings, int per_channel_stride, unsigned flags,
                            int max_write_samples, int softbit_capacity, jaero_ctx **out)
{
    if (!out || !settings || nchannels <= 0 || max_write_samples <= 0) return fail(JAERO_EINVAL, "jaero_create: bad arguments");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(JAERO_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(JAERO_ENODEV, "device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(JAERO_ENODEV, "device %d is %s; libjaero_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);

    auto sat = [&](int ch) -> const jaero_settings & {
        return per_channel_stride ? *(const jaero_settings *)((const char *)settings + (size_t)ch * per_channel_stride) : settings[0];
    };
    const jaero_settings &s0 = sat(0);
    int rc = validate_settings(s0);
    if (rc) return rc;
    for (int ch = 1; ch < nchannels; ch++)
    {
        const jaero_settings &s = sat(ch);
        if (s.kind != s0.kind || s.fb != s0.fb || s.Fs != s0.Fs || s.coarsefreqest_fft_power != s0.coarsefreqest_fft_power)
            return fail(JAERO_EINVAL, "channel %d: kind/fb/Fs/fft_power must match channel 0 within one bank", ch);
        if ((rc = validate_settings(s))) return rc;
    }

    jaero_ctx *c = new jaero_ctx();
    c->device = device;
    c->flags = flags;
    c->max_write = max_write_samples;
    c->soft_cap_req = softbit_capacity;
    if (s0.kind >= JAERO_KIND_BURST_MSK)
    {
        std::vector<jaero_settings> all(nchannels);
        for (int ch = 0; ch < nchannels; ch++) all[ch] = sat(ch);
        rc = burst_create(c, all, prop, softbit_capacity);
        if (rc) { jaero_destroy(c); return rc; }
        *out = c;
        return 0;
    }
    fill_geometry(c->g, s0, nchannels, flags);
    JGeom &g = c->g;
    if (softbit_capacity <= 0)
        softbit_capacity = (int)ceil(2.0 * max_write_samples * g.fb / g.Fs) + 64;
    g.soft_cap = (softbit_capacity + 7) & ~7; // 16-byte aligned rows: what the Aero-L bank's fast input path wants when it reads this buffer in place
    g.sym_cap = (flags & JAERO_FLAG_CAPTURE_SYMBOLS) ? g.soft_cap / 2 + 8 : 0;
    g.log_cap = (flags & JAERO_FLAG_STATUS_LOG) ? (int)ceil(g.soft_cap * g.Fs / g.fb / (g.nfft / 4)) + 16 : 0;
    const int nchp = g.nchp, ng = g.ngroups;

#define DA(ptr, count)                                              \
    do { if ((rc = dalloc(c, &(ptr), (size_t)(count)))) { jaero_destroy(c); return rc; } } while (0)
    DA(c->p.S, (size_t)S_NFIELDS * nchp);
    DA(c->p.I, (size_t)I_NFIELDS * nchp);
    DA(c->p.win, (size_t)ng * g.win_len * 64);
    DA(c->p.bbring, (size_t)nchp * g.nfft);
    DA(c->p.y, (size_t)nchp * g.nfft);
    DA(c->p.marg, (size_t)nchp * g.marg_len);
    DA(c->p.dt, (size_t)nchp * g.dt_len);
    DA(c->p.pm, (size_t)nchp * g.pm_len);
    DA(c->p.msema, (size_t)nchp * g.msema_len);
    DA(c->p.firsave, (size_t)ng * 2 * g.fir_n * 64);
    if (g.kind == JAERO_KIND_OQPSK && g.fb == 8400)
    {
        c->pre8400 = true;
        int ring = 1;
        // a write's first transform block starts up to 2047 samples before the write and looks 4096 samples further back
        while (ring < max_write_samples + 3 * PRE_L) ring <<= 1;
        c->pre.ring = ring;
        c->pre.cap = max_write_samples;
        DA(c->pre.xring, (size_t)ring * nchp);
        DA(c->pre.cidx, (size_t)max_write_samples * nchp);
        DA(c->pre.out, (size_t)max_write_samples * nchp);
        DA(c->pre.hold, (size_t)nchp); // zeros (DA clears): nothing held
        double *d_pre_taps = nullptr;
        DA(d_pre_taps, PRE_K);
        const std::vector<double> pt = rrc_design(0.6, 2048, g.Fs, g.fb / 2); // rrc_pre_imp (oqpskdemodulator.cpp:281)
        if ((int)pt.size() != PRE_K) { jaero_destroy(c); return fail(JAERO_EHIP, "prefilter design returned %zu taps", pt.size()); }
        HIPCHK(hipMemcpy(d_pre_taps, pt.data(), sizeof(double) * PRE_K, hipMemcpyHostToDevice));
        c->pre.taps = d_pre_taps;
        {
            double2 *dH = nullptr, *dtw = nullptr;
            if ((rc = fft4096_tables(pt, &dH, &dtw, (const void *)k_pre8400_fft))) { jaero_destroy(c); return rc; }
            c->pre.H = dH; c->pre.tw = dtw;
            c->allocs.push_back(dH); c->allocs.push_back(dtw);
            c->pre_direct = false; // (the time-domain form k_pre8400_fir and its A/B switch left the library in round 3)
        }
        HIPCHK(hipFuncSetAttribute((const void *)k_coarse6_w8400, hipFuncAttributeMaxDynamicSharedMemorySize, (C6_XCH + C4_TABN) * (int)sizeof(double)));
    }
    if (g.kind == JAERO_KIND_MSK) { DA(c->p.dly, (size_t)ng * (g.sps + 1) * 64); DA(c->p.dly8, (size_t)ng * (g.sps2 + 1) * 64); }
    DA(c->p.soft, (size_t)nchp * g.soft_cap);
    if (g.sym_cap) DA(c->p.sym, (size_t)nchp * g.sym_cap * 3);
    if (g.log_cap) DA(c->p.slog, (size_t)nchp * g.log_cap * 6);
    DA(c->d_pcm_frames, (size_t)max_write_samples * nchp);
    DA(c->d_pcm_raw, (size_t)max_write_samples * nchannels);
    DA(c->d_chanlist, nchp);
    DA(c->d_status, nchp);
    {
        c->coarse2_grid = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    DA(c->d_tw, g.nfft);
    double2 *d_cis = nullptr;
    double *d_taps = nullptr;
    DA(d_cis, JD_WTSIZE);
    DA(d_taps, 2 * g.fir_n);
#undef DA
    c->p.cis = d_cis; c->p.taps2 = d_taps;

    // TrigLookUp (JAERO/DSP.cpp:11-30): generated on the host so the table bits match the reference's libm
    {
        std::vector<double2> cis(JD_WTSIZE);
        for (int i = 0; i < JD_WTSIZE; i++)
        {
            cis[i].y = (sin(2 * M_PI * ((double)i) / JD_WTSIZE));
            cis[i].x = (sin(M_PI_2 + 2 * M_PI * ((double)i) / JD_WTSIZE));
        }
        HIPCHK(hipMemcpy(d_cis, cis.data(), sizeof(double2) * JD_WTSIZE, hipMemcpyHostToDevice));
        std::vector<double2> tw(g.nfft);
        for (int i = 0; i < g.nfft; i++) { double a = -2.0 * M_PI * ((double)i) / ((double)g.nfft); tw[i].x = cos(a); tw[i].y = sin(a); }
        HIPCHK(hipMemcpy(c->d_tw, tw.data(), sizeof(double2) * g.nfft, hipMemcpyHostToDevice));
        std::vector<double> taps;
        if (g.kind == JAERO_KIND_OQPSK) taps = rrc_design(g.fb == 8400 ? 0.6 : 1.0, 55, g.Fs, g.fb / 2);
        else
        {
            taps.resize(g.fir_n);
            const double SPS = (double)g.sps;
            for (int i = 0; i < 2 * SPS; i++) taps[i] = sin(M_PI * i / (2.0 * SPS)) / (2.0 * SPS);
        }
        std::vector<double> t2(2 * g.fir_n);
        for (int i = 0; i < 2 * g.fir_n; i++) t2[i] = taps[i % g.fir_n];
        HIPCHK(hipMemcpy(d_taps, t2.data(), sizeof(double) * t2.size(), hipMemcpyHostToDevice));
        if (g.kind == JAERO_KIND_OQPSK && g.fir_n == 55)
        {
            bool sym = true;
            for (int i = 0; i < 55; i++) sym = sym && memcmp(&taps[i], &taps[54 - i], sizeof(double)) == 0;
            c->oq_pairs = sym ? -1 : 0; // -1: front/back pairs allowed (decided below); 0: asymmetric taps -> the single-wavefront kernel reads them from LDS
            for (int i = 0; i < 28; i++) c->oq_taps.t[i] = taps[i];
        }
    }
    // scalar state
    {
        std::vector<double> S((size_t)S_NFIELDS * nchp, 0.0);
        std::vector<int> I((size_t)I_NFIELDS * nchp, 0);
        c->settings.resize(nchp);
        for (int ch = 0; ch < nchp; ch++)
        {
            const jaero_settings &s = sat(ch < nchannels ? ch : 0);
            c->settings[ch] = s;
            init_channel_scalars(c, s, S, I, ch, true);
        }
        HIPCHK(hipMemcpy(c->p.S, S.data(), S.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->p.I, I.data(), I.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    c->o_nch = nchannels; c->o_nchp = nchp; c->o_soft_cap = g.soft_cap; c->o_sym_cap = g.sym_cap;
    c->o_soft = c->p.soft; c->o_sym = c->p.sym;
    c->o_soft_cnt = c->p.I + (size_t)I_SOFT_CNT * nchp; c->o_sym_cnt = c->p.I + (size_t)I_SYM_CNT * nchp;
    c->o_overflow = c->p.I + (size_t)I_OVERFLOW * nchp; c->o_flags = c->p.I + (size_t)I_FLAGS * nchp;
    c->m.nch = nchannels; c->m.nchp = nchp; c->m.nfft = g.nfft; c->m.Fs_int = g.Fs_int;
    c->m.flags.assign(nchp, 0); c->m.bbptr.assign(nchp, 0); c->m.cnt.assign(nchp, 0);
    // dynamic LDS for the matched-filter rings
    if (g.kind == JAERO_KIND_MSK)
    {
        if (g.fir_n != 20 && g.fir_n != 40 && g.fir_n != 80 && g.fir_n != 160)
            return fail(JAERO_ENOTSUP, "MSK matched filter of %d taps (fb %g at Fs %g) has no kernel", g.fir_n, g.fb, g.Fs);
#define MSK_ATTR(F, L) \
    { \
        const int lds_bytes = (2 * (L) * 64 + (F)) * (int)sizeof(double); \
        HIPCHK(hipFuncSetAttribute((const void *)k_msk_samples<F, L, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
        HIPCHK(hipFuncSetAttribute((const void *)k_msk_samples<F, L, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
        HIPCHK(hipFuncSetAttribute((const void *)k_msk_samples<F, L, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
        HIPCHK(hipFuncSetAttribute((const void *)k_msk_samples<F, L, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
    }
        c->msk_ldsn = g.fir_n == 40 ? MSK_LDSN_40 : MSK_LDSN_20; // 80 / 160 taps: set with the pair kernel below
        if (g.fir_n == 40) MSK_ATTR(40, MSK_LDSN_40) else if (g.fir_n == 20) MSK_ATTR(20, MSK_LDSN_20) // 80 and 160 taps: k_msk_fb below
#undef MSK_ATTR
        if (g.fir_n == 160)
        {
            // 160 taps: two pairs per workgroup, each wavefront alone on a SIMD (k_msk_fb.h, MFB2_*)
            c->msk_pairs = 2;
            c->msk_ldsn = MFB2_LDSN;
#define MFA(E, C) HIPCHK(hipFuncSetAttribute((const void *)k_msk_fb<160, MFB2_LDSN, E, C, 2, MFB2_TB>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * mfb_pair_doubles<160, MFB2_LDSN, MFB2_TB>() * (int)sizeof(double)))
            MFA(false, false); MFA(false, true); MFA(true, false); MFA(true, true);
#undef MFA
        }
        if (g.fir_n == 80)
        {
            // front / back pairs (k_msk_fb.h).  Banks of at most two channel groups per CU: one pair per workgroup, the halves on different
            // SIMDs, 36 history entries per arm in LDS and 44 in the front half's registers (256 channels: 82 -> 113 Msamples/s).  Larger
            // banks: four pairs per workgroup, each pair on one SIMD, 32 entries in LDS, 26 in the front half's and the 22 oldest in the
            // back half's registers (65 536 channels: 9.2 -> 10.6 Gsamples/s).
            const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            c->msk_pairs = g.ngroups > 2 * ncu ? 4 : 1;
            if (c->msk_pairs == 1)
            {
                c->msk_ldsn = MFB_LDSN;
#define MFA(E, C) HIPCHK(hipFuncSetAttribute((const void *)k_msk_fb<80, MFB_LDSN, E, C, 1, MFB1_TB>, hipFuncAttributeMaxDynamicSharedMemorySize, mfb_pair_doubles<80, MFB_LDSN, MFB1_TB>() * (int)sizeof(double)))
                MFA(false, false); MFA(false, true); MFA(true, false); MFA(true, true);
#undef MFA
            }
            else if (c->msk_pairs == 4)
            {
                c->msk_ldsn = MFB4_LDSN;
#define MFA(E, C) HIPCHK(hipFuncSetAttribute((const void *)k_msk_fb<80, MFB4_LDSN, E, C, 4, MFB4_TB>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * mfb_pair_doubles<80, MFB4_LDSN, MFB4_TB>() * (int)sizeof(double)))
                MFA(false, false); MFA(false, true); MFA(true, false); MFA(true, true);
#undef MFA
            }
        }
    }
    if (g.kind == JAERO_KIND_OQPSK)
    {
        // front / back pairs (k_oqpsk_fb.h; at 8400 bps the two halves take turns, see there).  Four pairs per workgroup put one front and one back wavefront on every SIMD of a
        // CU -- worth it once there are more channel groups than two per CU; smaller banks get one pair per workgroup (two SIMDs per
        // 64 channels).  (The single-wavefront kernel of round 1 and its A/B switch left the library in round 3.)
        const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (!c->oq_pairs) { jaero_destroy(c); return fail(JAERO_ENOTSUP, "matched-filter taps are not bitwise symmetric: k_oqpsk_fb takes the 28 distinct taps as scalar arguments"); }
        c->oq_pairs = g.ngroups > 2 * ncu ? 4 : 1;
        static_assert(JD_SYMREC_LEN == 800, "the combined symbol-record ring of k_oqpsk_fb assumes the reference's window lengths (800 / 400 / 400 / 400)");
        if (c->oq_pairs)
        {
            if ((rc = dalloc(c, &c->p.symrec, (size_t)nchp * JD_SYMREC_LEN * 8))) { jaero_destroy(c); return rc; }
#define FBA(E, C, PP, X) HIPCHK(hipFuncSetAttribute((const void *)k_oqpsk_fb<55, FB_LDSN, E, C, PP, X>, hipFuncAttributeMaxDynamicSharedMemorySize, PP * fb_pair_doubles<FB_LDSN>() * (int)sizeof(double)))
            if (c->pre8400)
            {
                FBA(false, false, 1, true); FBA(false, true, 1, true); FBA(true, false, 1, true); FBA(true, true, 1, true);
                FBA(false, false, 4, true); FBA(false, true, 4, true); FBA(true, false, 4, true); FBA(true, true, 4, true);
            }
            else
            {
                FBA(false, false, 1, false); FBA(false, true, 1, false); FBA(true, false, 1, false); FBA(true, true, 1, false);
                FBA(false, false, 4, false); FBA(false, true, 4, false); FBA(true, false, 4, false); FBA(true, true, 4, false);
            }
#undef FBA
        }
    }
    if (g.nfft_log2 == 14) HIPCHK(hipFuncSetAttribute((const void *)k_coarse6, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH * (int)sizeof(double)));
    else HIPCHK(hipFuncSetAttribute((const void *)k_coarse6_13, hipFuncAttributeMaxDynamicSharedMemorySize, C6_XCH13 * (int)sizeof(double)));
    HIPCHK(hipDeviceSynchronize());
    *out = c;
    return 0;
}

// ------------------------------------------------------------------------------------------ control surface
// flags[ch] = (flags[ch] & ~mask) | bits for ch in [lo, hi): by value, on the stream of the last write (as the write that follows will
// be, see jaero_write) -- no host buffer is in flight, so a second call right behind the first cannot disturb it
__global__ void k_set_flag_bits(int *flags, int lo, int hi, int mask, int bits)
{
    const int ch = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < hi) flags[ch] = (flags[ch] & ~mask) | bits;
}
static int upload_flags(jaero_ctx *c, int lo, int hi, int mask, int bits)
{
    hipLaunchKernelGGL(k_set_flag_bits, dim3((hi - lo + 255) / 256), dim3(256), 0, c->last_stream, c->o_flags, lo, hi, mask, bits);
    HIPCHK(hipGetLastError());
    return 0;
}

// a bank whose write failed part-way (device state and the host's schedule mirror disagree) accepts nothing but jaero_destroy: setters would
// change a state nobody can continue from, readers would hand out half-written outputs without saying so
#define POISONCHK(c, who) do { if ((c) && (c)->poisoned) return fail(JAERO_EHIP, who ": an earlier jaero_write of this bank failed part-way; destroy the bank and create a new one"); } while (0)

extern "C" int jaero_set_flags(jaero_ctx *c, int channel, int afc, int sql, int cpu_reduce)
{
    POISONCHK(c, "jaero_set_flags");
    if (!c || channel < -1 || channel >= c->o_nch) return fail(JAERO_EINVAL, "jaero_set_flags: bad channel");
    HIPCHK(hipSetDevice(c->device));
    const int lo = channel < 0 ? 0 : channel, hi = channel < 0 ? c->o_nchp : channel + 1;
    const int bits = (afc ? JF_AFC : 0) | (sql ? JF_SQL : 0) | (cpu_reduce ? JF_CPUREDUCE : 0);
    for (int ch = lo; ch < hi; ch++) c->m.flags[ch] = (c->m.flags[ch] & JF_DCD) | bits;
    return upload_flags(c, lo, hi, JF_AFC | JF_SQL | JF_CPUREDUCE, bits);
}

extern "C" int jaero_set_dcd(jaero_ctx *c, int channel, int dcd)
{
    POISONCHK(c, "jaero_set_dcd");
    if (!c || channel < -1 || channel >= c->o_nch) return fail(JAERO_EINVAL, "jaero_set_dcd: bad channel");
    HIPCHK(hipSetDevice(c->device));
    const int lo = channel < 0 ? 0 : channel, hi = channel < 0 ? c->o_nchp : channel + 1;
    for (int ch = lo; ch < hi; ch++) c->m.flags[ch] = (c->m.flags[ch] & ~JF_DCD) | (dcd ? JF_DCD : 0);
    return upload_flags(c, lo, hi, JF_DCD, dcd ? JF_DCD : 0);
}

extern "C" int jaero_center_freq_changed(jaero_ctx *c, int channel, double hz)
{
    POISONCHK(c, "jaero_center_freq_changed");
    if (!c || channel < -1 || channel >= c->o_nch) return fail(JAERO_EINVAL, "jaero_center_freq_changed: bad channel");
    HIPCHK(hipSetDevice(c->device));
    if (c->burst)
    {
        // BurstOqpskDemodulator::CenterFreqChangedSlot does nothing (burstoqpskdemodulator.cpp:284-289)
        if (c->bg.kind == JAERO_KIND_BURST_OQPSK) return 0;
        // BurstMskDemodulator::CenterFreqChangedSlot (burstmskdemodulator.cpp:327-342), on the bank's stream like every other call
        const int lo = channel < 0 ? 0 : channel, n = channel < 0 ? c->o_nch : 1;
        hipLaunchKernelGGL(k_burst_msk_center_freq, dim3((n + 63) / 64), dim3(64), 0, c->last_stream, c->bg, c->bp, lo, n, hz, (long long)c->nsamples_total);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int lo = channel < 0 ? 0 : channel, n = channel < 0 ? c->g.nch : 1;
    hipLaunchKernelGGL(k_center_freq, dim3(n), dim3(256), 0, c->last_stream, c->g, c->p, lo, n, hz);
    HIPCHK(hipGetLastError());
    return 0;
}

// setSettings on live channels [ch_lo, ch_hi): the scalar fields it rewrites, the counters it restarts and the rings it recreates
// (= zeroes), for those channels' columns only.  blockIdx.x = channel - ch_lo, blockIdx.y strides over ring slots.
struct JSetVals
{
    int nS, nI;
    int fS[40]; double vS[40];   // S[fS[k]][ch] = vS[k]
    int fI[16]; int vI[16];      // I[fI[k]][ch] = vI[k]
    int zero_eb, zero_msk;       // MSK also recreates the EbNo meter, marg and the quadrature delay line
    int dly_rot;                 // MSK: slots by which the channel's delayedsmpl column is rotated (see below)
};
__global__ void k_apply_settings(const JGeom g, const JPtrs p, int ch_lo, const JSetVals v)
{
    const int ch = ch_lo + blockIdx.x, nchp = g.nchp;
    const int grp = ch >> 6, lane = ch & 63;
    if (blockIdx.y == 0)
    {
        for (int k = threadIdx.x; k < v.nS; k += blockDim.x) p.S[(size_t)v.fS[k] * nchp + ch] = v.vS[k];
        for (int k = threadIdx.x; k < v.nI; k += blockDim.x) p.I[(size_t)v.fI[k] * nchp + ch] = v.vI[k];
    }
    const int tid = blockIdx.y * blockDim.x + threadIdx.x, nth = gridDim.y * blockDim.x;
    // OQPSK: the AGC is re-created but the EbNo meter keeps its buffer, and both live in the one window ring: I_AGC_HOLD instead of zeros
    if (v.zero_eb) for (int s = tid; s < g.win_len; s += nth) p.win[((size_t)grp * g.win_len + s) * 64 + lane] = 0.0;
    for (int s = tid; s < 2 * g.fir_n; s += nth) p.firsave[((size_t)grp * 2 * g.fir_n + s) * 64 + lane] = 0.0;
    if (v.zero_msk)
    {
        for (int s = tid; s < g.marg_len; s += nth) p.marg[(size_t)ch * g.marg_len + s] = 0.0;
        for (int s = tid; s < g.sps2 + 1; s += nth) p.dly8[((size_t)grp * (g.sps2 + 1) + s) * 64 + lane] = 0.0;
        // delayedsmpl.setLength(SPS) keeps the buffer's contents and restarts its pointer at 0 (DSP.h:446-453).  Here the ring slot is
        // shared by the 64 channels of a wavefront (slot = samples so far mod SPS+1), so restarting ONE channel's pointer means
        // rotating that channel's column by the distance between the shared slot now and the slot its pointer last restarted at.
        if (tid == 0 && v.dly_rot > 0)
        {
            const int L = g.sps + 1; // 41 or 81
            double2 tmp[96];
            for (int s = 0; s < L; s++) tmp[s] = p.dly[((size_t)grp * L + s) * 64 + lane];
            for (int s = 0; s < L; s++) p.dly[((size_t)grp * L + (s + v.dly_rot) % L) * 64 + lane] = tmp[s];
        }
    }
}

// setSettings on the channels [lo, hi) of a bank whose kind, rates and FFT size stay what they are: only those channels' columns are touched,
// by one small kernel on the stream of the last jaero_write (no device synchronisation, no copy of the bank's state)
static int apply_live_settings(jaero_ctx *c, int lo, int hi, const jaero_settings *s)
{
    const JGeom &g = c->g;
    HIPCHK(hipSetDevice(c->device));
    const int nchp = g.nchp;
    // what init_channel_scalars(fresh = false) writes, computed once (the same for every addressed channel) on a one-column scratch
    std::vector<double> S((size_t)S_NFIELDS, 0.0);
    std::vector<int> I((size_t)I_NFIELDS, 0);
    init_channel_scalars(c, *s, S, I, 0, false, 1);
    JSetVals v{};
    auto setS = [&](int f, double x) { v.fS[v.nS] = f; v.vS[v.nS] = x; v.nS++; };
    auto setI = [&](int f, int x) { v.fI[v.nI] = f; v.vI[v.nI] = x; v.nI++; };
    for (int f : {S_M2_FREQ, S_M2_STEP, S_MC_FREQ, S_MC_STEP, S_ST_FREQ, S_ST_STEP, S_LOCKINGBW, S_THRESH}) setS(f, S[(size_t)f]);
    if (g.kind == JAERO_KIND_OQPSK)
        for (int f : {S_AGC_SUM, S_D1, S_D41_1, S_D41_2, S_D41_3, S_D42_1, S_D42_2, S_D42_3, S_D8_1, S_D8_2, S_RES_X1, S_RES_X2, S_RES_Y1, S_RES_Y2}) setS(f, 0.0);
    else
    {
        setS(S_MSE, 10.0); // setSettings resets mse (mskdemodulator.cpp:180)
        for (int f : {S_AGC_SUM, S_RES_X1, S_RES_X2, S_RES_Y1, S_RES_Y2, S_EB_ESUM, S_EB_E2SUM, S_EB_EBNO, S_MARG_SUM, S_MFB_A0_RE, S_MFB_A0_IM}) setS(f, 0.0);
        setI(I_MARG_POS, 0); setI(I_DT_POS, 0);
        v.zero_eb = 1; v.zero_msk = 1;
    }
    setI(I_BB_PTR, 0); setI(I_COARSE_CNT, 0);
    if (g.kind == JAERO_KIND_OQPSK) setI(I_AGC_HOLD, g.agc_len); // new AGC(4, Fs): an empty window; the ring's position and contents stay (the meter's)
    else setI(I_AGC_POS, 0);
    for (int ch = lo; ch < hi; ch++)
    {
        c->settings[ch] = *s;
        c->m.bbptr[ch] = 0; c->m.cnt[ch] = 0;
    }
    const int ny = (hi - lo) >= 256 ? 4 : 64; // one channel: 64 blocks share its 192 000-slot AGC column; the whole bank: four per channel
    if (g.kind == JAERO_KIND_MSK)
    {
        // runs of channels whose delayedsmpl pointer last restarted at the same shared slot get one launch each (normally: one run)
        const int L = g.sps + 1;
        const int now = (int)(c->m.nB_total % L);
        if (c->dly_t0.empty()) c->dly_t0.assign(nchp, 0);
        for (int a = lo; a < hi;)
        {
            int b = a + 1;
            while (b < hi && c->dly_t0[b] == c->dly_t0[a]) b++;
            v.dly_rot = ((now - c->dly_t0[a]) % L + L) % L;
            hipLaunchKernelGGL(k_apply_settings, dim3(b - a, ny), dim3(256), 0, c->last_stream, g, c->p, a, v);
            for (int ch = a; ch < b; ch++) c->dly_t0[ch] = now;
            a = b;
        }
    }
    else hipLaunchKernelGGL(k_apply_settings, dim3(hi - lo, ny), dim3(256), 0, c->last_stream, g, c->p, lo, v);
    HIPCHK(hipGetLastError());
    return 0;
}

// k_carry_dly: MSK delayedsmpl.setLength(SamplesPerSymbol) keeps the buffer's first min(old, new) entries IN BUFFER ORDER and restarts the
// pointer at 0 (DSP.h:446-453).  Buffer index j of channel ch lives in the old bank's shared slot (t0[ch] + j) mod Lo (t0 = the shared slot at
// which the channel's pointer last restarted) and in slot j of the new bank, whose shared slot counter starts at 0.
__global__ void k_carry_dly(const double2 *__restrict__ od, int Lo, const int *__restrict__ t0, double2 *__restrict__ nd, int Ln, int nchp)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nchp) return;
    const int grp = ch >> 6, lane = ch & 63, keep = Lo < Ln ? Lo : Ln, s0 = t0[ch];
    for (int j = 0; j < keep; j++) nd[((size_t)grp * Ln + j) * 64 + lane] = od[((size_t)grp * Lo + (s0 + j) % Lo) * 64 + lane];
}

// setSettings that changes what a bank fixes (bit rate, sample rate, FFT size; any setSettings of an 8400 bps bank, whose prefilter restarts):
// the reference rebuilds AGC, matched filters, delays, resonator (and at 8400 bps the prefilter) INSIDE the old object, which keeps its oscillator
// phases, loop-filter and rotator states, moving-average windows, the coarse ring's contents and the smoothed spectrum (oqpskdemodulator.cpp:
// 175-289, mskdemodulator.cpp:135-263).  Here: a sibling bank is created for the new settings, those survivors are copied into it, the same
// in-place setSettings as above runs on it, and it takes the place of the old bank behind the handle.  Whole banks only; control plane (it
// allocates and synchronises).  Outputs not read yet move along; device pointers obtained from the views are stale.
static void prof_collect(jaero_ctx *c);
static int rebank_with_carry_over(jaero_ctx *c, const jaero_settings *s)
{
    const JGeom og = c->g;
    if (c->poisoned) return fail(JAERO_EHIP, "jaero_set_settings: a launch inside an earlier jaero_write failed; this bank's state cannot be carried over");
    if (og.kind == JAERO_KIND_OQPSK && s->Fs != og.Fs) return fail(JAERO_ENOTSUP, "jaero_set_settings: an OQPSK bank keeps its sample rate");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    jaero_ctx *n = nullptr;
    int rc = jaero_create(c->device, og.nch, s, 0, c->flags, c->max_write, c->soft_cap_req, &n);
    if (rc) return rc;
    const JGeom &ng = n->g;
    const int nchp = og.nchp;
    auto fin = [&](int code) { jaero_destroy(n); return code; };
#define CP(dst, src, bytes) do { if (hipMemcpy((dst), (src), (bytes), hipMemcpyDeviceToDevice) != hipSuccess) return fin(fail(JAERO_EHIP, "jaero_set_settings: carry-over copy failed")); } while (0)
#define CP2(dst, dpitch, src, spitch, width, rows) do { if (hipMemcpy2D((dst), (dpitch), (src), (spitch), (width), (rows), hipMemcpyDeviceToDevice) != hipSuccess) return fin(fail(JAERO_EHIP, "jaero_set_settings: carry-over copy failed")); } while (0)
    CP(n->p.S, c->p.S, sizeof(double) * (size_t)S_NFIELDS * nchp);
    CP(n->p.I, c->p.I, sizeof(int) * (size_t)I_NFIELDS * nchp);
    {
        // outputs not read yet (soft bits, captured symbols, status rows) move to the new bank's buffers; its capacities follow the new rate
        std::vector<int> cnt(3 * (size_t)nchp);
        static_assert(I_SYM_CNT == I_SOFT_CNT + 1 && I_LOG_CNT == I_SOFT_CNT + 2, "output counters are consecutive columns");
        if (hipMemcpy(cnt.data(), c->p.I + (size_t)I_SOFT_CNT * nchp, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost) != hipSuccess) return fin(fail(JAERO_EHIP, "jaero_set_settings: reading the output counters failed"));
        int mx[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) for (int ch = 0; ch < og.nch; ch++) mx[k] = cnt[(size_t)k * nchp + ch] > mx[k] ? cnt[(size_t)k * nchp + ch] : mx[k];
        if (mx[0] > ng.soft_cap || mx[1] > ng.sym_cap || mx[2] > ng.log_cap)
            return fin(fail(JAERO_EINVAL, "jaero_set_settings: unread outputs (%d soft bits, %d symbols, %d status rows) exceed the new bank's buffers; read them first", mx[0], mx[1], mx[2]));
        if (mx[0]) CP2(n->p.soft, sizeof(int16_t) * ng.soft_cap, c->p.soft, sizeof(int16_t) * og.soft_cap, sizeof(int16_t) * mx[0], (size_t)nchp);
        if (mx[1]) CP2(n->p.sym, sizeof(double) * 3 * ng.sym_cap, c->p.sym, sizeof(double) * 3 * og.sym_cap, sizeof(double) * 3 * mx[1], (size_t)nchp);
        if (mx[2]) CP2(n->p.slog, sizeof(double) * 6 * ng.log_cap, c->p.slog, sizeof(double) * 6 * og.log_cap, sizeof(double) * 6 * mx[2], (size_t)nchp);
    }
    // windows that survive: msema (both kinds); marg / dt / pm of the OQPSK demodulator (MSK: marg is new, dt keeps a prefix, below)
    CP(n->p.msema, c->p.msema, sizeof(double) * (size_t)nchp * og.msema_len);
    if (og.kind == JAERO_KIND_OQPSK)
    {
        CP(n->p.marg, c->p.marg, sizeof(double) * (size_t)nchp * og.marg_len);
        CP(n->p.dt, c->p.dt, sizeof(double2) * (size_t)nchp * og.dt_len);
        CP(n->p.pm, c->p.pm, sizeof(double) * (size_t)nchp * og.pm_len);
        if (c->p.symrec && n->p.symrec) CP(n->p.symrec, c->p.symrec, sizeof(double) * (size_t)nchp * JD_SYMREC_LEN * 8); // k_oqpsk_fb keeps the four windows in one record ring
        // the OQPSK EbNo meter is only told the new rates (setup_update, DSP.cpp:723-727): its window -- the newest ebno_len entries of the window
        // ring, at the position that came over with I -- stays; the AGC is new (I_AGC_HOLD, set by apply_live_settings below)
        if ((c->flags & JAERO_FLAG_EBNO) && og.win_len == ng.win_len) CP(n->p.win, c->p.win, sizeof(double) * (size_t)og.ngroups * og.win_len * 64);
    }
    else
    {
        const int keep = og.dt_len < ng.dt_len ? og.dt_len : ng.dt_len; // dt.setLength: DelayThing keeps the first entries (DSP.h:446-453)
        CP2(n->p.dt, sizeof(double2) * ng.dt_len, c->p.dt, sizeof(double2) * og.dt_len, sizeof(double2) * keep, (size_t)nchp);
        CP(n->p.pm, c->p.pm, sizeof(double) * (size_t)nchp * (og.pm_len < ng.pm_len ? og.pm_len : ng.pm_len));
        int *d_t0 = nullptr;
        if (c->dly_t0.empty()) c->dly_t0.assign(nchp, 0);
        if (hipMalloc(&d_t0, sizeof(int) * nchp) != hipSuccess) return fin(fail(JAERO_ENOMEM, "jaero_set_settings: out of device memory"));
        hipMemcpy(d_t0, c->dly_t0.data(), sizeof(int) * nchp, hipMemcpyHostToDevice);
        // the old bank's shared slot counter stands at nB_total: a channel whose pointer restarted at slot t0 has its buffer index 0 there
        hipLaunchKernelGGL(k_carry_dly, dim3((nchp + 255) / 256), dim3(256), 0, 0, (const double2 *)c->p.dly, og.sps + 1, (const int *)d_t0, (double2 *)n->p.dly, ng.sps + 1, nchp);
        hipStreamSynchronize(0); // this copy's stream only: other banks on the GPU keep running
        hipFree(d_t0);
    }
    {
        // bbcycbuff.resize / y.resize: same size = untouched, otherwise the first entries stay (new ones are zero)
        const int keep = og.nfft < ng.nfft ? og.nfft : ng.nfft;
        CP2(n->p.bbring, sizeof(double2) * ng.nfft, c->p.bbring, sizeof(double2) * og.nfft, sizeof(double2) * keep, (size_t)nchp);
        CP2(n->p.y, sizeof(double) * ng.nfft, c->p.y, sizeof(double) * og.nfft, sizeof(double) * keep, (size_t)nchp);
    }
    if (og.kind == JAERO_KIND_OQPSK)
    {
        // mixer_fir_pre is not touched by setSettings: its phase stays.  Its frequency is set at the end of every write to mixer2_freq_sum / i
        // (oqpskdemodulator.cpp:607-608; applied here by k_pre8400_mix at the start of the next write from S_PRE_FSUM / nprev), and that sum
        // only grows in the 8400 bps branch (:447): after a write at another rate the reference's prefilter oscillator stands at 0 Hz, which
        // is what its first 8400 bps write then mixes with.  Same here: sum 0 over the length of the last write.
        n->pre_nprev = c->pre_nprev;
        if (!c->pre8400 && hipMemset(n->p.S + (size_t)S_PRE_FSUM * nchp, 0, sizeof(double) * (size_t)nchp) != hipSuccess) return fin(fail(JAERO_EHIP, "memset"));
    }
#undef CP
#undef CP2
    n->m.flags = c->m.flags;
    // kernel timings of the launches since the last jaero_profile_read belong to the handle, not to the bank behind it: drained into the
    // totals here (their events go with the old bank)
    if (c->prof) prof_collect(c);
    n->prof = c->prof;
    for (size_t k = 0; k < sizeof(c->slots) / sizeof(c->slots[0]); k++) n->slots[k] = c->slots[k];
    if ((rc = apply_live_settings(n, 0, ng.nchp, s))) return fin(rc); // the padding lanes too: their window positions came over with I and must fit the new lengths
    if (hipStreamSynchronize(n->last_stream) != hipSuccess || hipStreamSynchronize(0) != hipSuccess) return fin(fail(JAERO_EHIP, "jaero_set_settings: carry-over failed"));
    std::swap(*c, *n);
    jaero_destroy(n); // the old bank
    return 0;
}

extern "C" int jaero_set_settings(jaero_ctx *c, int channel, const jaero_settings *s)
{
    // setSettings on a live object (oqpskdemodulator.cpp:175-289, mskdemodulator.cpp:135-263): retunes the mixers
    // (phase kept), recreates AGC / matched filters / timing delays / resonator, restarts the coarse ring pointer.
    if (!c || !s || channel < -1 || channel >= c->o_nch) return fail(JAERO_EINVAL, "jaero_set_settings: bad arguments");
    POISONCHK(c, "jaero_set_settings");
    int rc = validate_settings(*s);
    if (rc) return rc;
    if (c->burst) return burst_set_settings(c, channel, s);
    const JGeom &g = c->g;
    if (s->kind != g.kind) return fail(JAERO_EINVAL, "jaero_set_settings: the kind of a bank is fixed (another demodulator class in the reference); create a new bank");
    const bool whole = channel < 0 || g.nch == 1;
    if (s->fb != g.fb || s->Fs != g.Fs || s->coarsefreqest_fft_power != g.nfft_log2)
    {
        if (!whole) return fail(JAERO_EINVAL, "jaero_set_settings: fb/Fs/fft_power are shared by the channels of a bank; change them for the whole bank (channel = -1)");
        return rebank_with_carry_over(c, s);
    }
    if (c->pre8400)
    {
        // fb = 8400: setSettings also re-creates the prefilter (JFastFir::SetKernel: empty history, 2048 zeros of latency, transform blocks
        // re-aligned to that moment).  The whole bank: re-created behind the handle, blocks re-aligned.  One channel of several: its column
        // of the prefilter history is emptied and its outputs held at exact zeros for 2048 samples (k_pre8400_restart); the transform blocks
        // stay on the bank's grid (round 5; refused until then)
        if (whole) return rebank_with_carry_over(c, s);
        HIPCHK(hipSetDevice(c->device));
        hipLaunchKernelGGL(k_pre8400_restart, dim3(16), dim3(256), 0, c->last_stream, g, c->pre, channel, c->pre_n0);
        HIPCHK(hipGetLastError());
        return apply_live_settings(c, channel, channel + 1, s);
    }
    HIPCHK(hipSetDevice(c->device));
    return apply_live_settings(c, channel < 0 ? 0 : channel, channel < 0 ? g.nch : channel + 1, s);
}

// ------------------------------------------------------------------------------------------ profiling helpers
static int prof_begin(jaero_ctx *c, int which, hipStream_t st)
{
    if (!c->prof) return -1;
    if (c->ev_next >= c->ev_pool.size())
    {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        c->ev_pool.push_back({a, b});
    }
    const int idx = (int)c->ev_next++;
    hipEventRecord(c->ev_pool[idx].first, st);
    c->ev_used.push_back({which, idx});
    return idx;
}
static void prof_end(jaero_ctx *c, int idx, hipStream_t st)
{
    if (idx >= 0) hipEventRecord(c->ev_pool[idx].second, st);
}
static void prof_collect(jaero_ctx *c)
{
    for (auto &u : c->ev_used)
    {
        float ms = 0;
        hipEventSynchronize(c->ev_pool[u.idx].second);
        if (hipEventElapsedTime(&ms, c->ev_pool[u.idx].first, c->ev_pool[u.idx].second) == hipSuccess)
        {
            c->slots[u.which].ms += ms;
            c->slots[u.which].launches++;
        }
    }
    c->ev_used.clear();
    c->ev_next = 0;
}
extern "C" int jaero_profile_enable(jaero_ctx *c, int on)
{
    if (!c) return fail(JAERO_EINVAL, "null ctx");
    c->prof = on != 0;
    return 0;
}
extern "C" int jaero_profile_read(jaero_ctx *c, int which, double *total_ms, int *launches, int reset)
{
    if (!c || which < 0 || which > 4) return fail(JAERO_EINVAL, "jaero_profile_read: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    prof_collect(c);
    if (total_ms) *total_ms = c->slots[which].ms;
    if (launches) *launches = c->slots[which].launches;
    if (reset) c->slots[which] = ProfSlot();
    return 0;
}

extern "C" int jaero_profile_kernel(jaero_ctx *c, int which, char *buf, int cap)
{
    if (!c || !buf || cap < 2 || which < 0 || which > 4) return fail(JAERO_EINVAL, "jaero_profile_kernel: bad arguments");
    const JGeom &g = c->g;
    const char *nm = "";
    if (c->burst)
    {
        static const char *bn[5] = {"k_burst_oqpsk_demod", "k_trident", "k_hist_push", "k_hilbert_fft", "k_burst_front"};
        nm = bn[which];
        if (which == 0 && c->bg.kind == JAERO_KIND_BURST_MSK) nm = "k_burst_msk_fb";
    }
    else if (which == 0)
    {
        if (g.kind == JAERO_KIND_OQPSK) nm = "k_oqpsk_fb<";
        else nm = c->msk_pairs ? "k_msk_fb<" : "k_msk_samples<";
    }
    else if (which == 1) nm = (g.nfft_log2 == 14) ? (c->pre8400 ? "k_coarse6_w8400" : "k_coarse6") : "k_coarse6_13";
    else if (which == 2) nm = "k_transpose_pcm";
    snprintf(buf, (size_t)cap, "%s", nm);
    return 0;
}

// ------------------------------------------------------------------------------------------ write
static void launch_samples(jaero_ctx *c, const int16_t *frames, int stride, int n, int skipA, int onlyA, hipStream_t st, int pos)
{
    const JGeom &g = c->g;
    const bool eb = (c->flags & JAERO_FLAG_EBNO) != 0, cs = (c->flags & JAERO_FLAG_CAPTURE_SYMBOLS) != 0;
    const dim3 grid(g.ngroups), block(64);
    if (g.kind == JAERO_KIND_OQPSK)
    {
        {
            // front / back wavefront pairs (k_oqpsk_fb.h): PAIRS pairs per workgroup
            const int fsb = (int)(c->m.nB_total % FB_LDSN);

            const int P = c->oq_pairs;
            const dim3 gridp((g.ngroups + P - 1) / P), blockp(P * 128);
            const int ldsp = P * fb_pair_doubles<FB_LDSN>() * (int)sizeof(double);
            const double2 *pf = c->pre8400 ? (const double2 *)(c->pre.out + (size_t)pos * 4) : nullptr; // row `pos` of every channel group (JD_G4)
#define LFB(E, C, PP, X) hipLaunchKernelGGL((k_oqpsk_fb<55, FB_LDSN, E, C, PP, X>), gridp, blockp, ldsp, st, g, c->p, frames, stride, n, skipA, onlyA, fsb, c->oq_taps, pf, c->pre.cap)
#define LFBP(E, C) { if (c->pre8400) { if (P == 4) LFB(E, C, 4, true); else LFB(E, C, 1, true); } else { if (P == 4) LFB(E, C, 4, false); else LFB(E, C, 1, false); } }
            if (eb && cs) LFBP(true, true) else if (eb) LFBP(true, false) else if (cs) LFBP(false, true) else LFBP(false, false)
#undef LFBP
#undef LFB
            return;
        }
        return; // (c->oq_pairs is always set: jaero_create refuses a bank it could not serve with k_oqpsk_fb)
    }
    else
    {
        const int ldsn = c->msk_ldsn;
        const int lds = (2 * ldsn * 64 + g.fir_n) * (int)sizeof(double); // rings + this wavefront's copy of the taps
        const int fs = (int)(c->m.nB_total % ldsn), ds = (int)(c->m.nB_total % (g.sps + 1)), d8 = (int)(c->m.nB_total % (g.sps2 + 1));
        if (c->msk_pairs)
        {
            const int P = c->msk_pairs;
            const dim3 gridp((g.ngroups + P - 1) / P), blockp(P * 128);
            const int ldsp = (P == 4 ? 4 * mfb_pair_doubles<80, MFB4_LDSN, MFB4_TB>() : (P == 2 ? 2 * mfb_pair_doubles<160, MFB2_LDSN, MFB2_TB>() : mfb_pair_doubles<80, MFB_LDSN, MFB1_TB>())) * (int)sizeof(double);
#define LMF2(E, C) hipLaunchKernelGGL((k_msk_fb<160, MFB2_LDSN, E, C, 2, MFB2_TB>), gridp, blockp, ldsp, st, g, c->p, frames, stride, n, skipA, onlyA, fs, ds, d8)
            if (P == 2)
            {
                if (eb && cs) LMF2(true, true); else if (eb) LMF2(true, false); else if (cs) LMF2(false, true); else LMF2(false, false);
                return;
            }
#undef LMF2
#define LMF(E, C, PP, LL, TT) hipLaunchKernelGGL((k_msk_fb<80, LL, E, C, PP, TT>), gridp, blockp, ldsp, st, g, c->p, frames, stride, n, skipA, onlyA, fs, ds, d8)
#define LMFP(E, C) { if (P == 4) LMF(E, C, 4, MFB4_LDSN, MFB4_TB); else LMF(E, C, 1, MFB_LDSN, MFB1_TB); }
            if (eb && cs) LMFP(true, true) else if (eb) LMFP(true, false) else if (cs) LMFP(false, true) else LMFP(false, false)
#undef LMFP
#undef LMF
            return;
        }
#define LM(F, L, E, C) hipLaunchKernelGGL((k_msk_samples<F, L, E, C>), grid, block, lds, st, g, c->p, frames, stride, n, skipA, onlyA, fs, ds, d8)
#define LMS(F, L) { if (eb && cs) LM(F, L, true, true); else if (eb) LM(F, L, true, false); else if (cs) LM(F, L, false, true); else LM(F, L, false, false); }
        if (g.fir_n == 40) LMS(40, MSK_LDSN_40) else LMS(20, MSK_LDSN_20) // 80 and 160 taps never get here (k_msk_fb)
#undef LMS
#undef LM
    }
}

static void launch_coarse(jaero_ctx *c, const int *d_list, int nlist, hipStream_t st)
{
    // register-resident FFTs: one 512-thread workgroup per CU (the whole register file, ~130 KiB of LDS), persistent over the list
    const int grid = nlist < c->coarse2_grid ? nlist : c->coarse2_grid;
    if (c->g.nfft_log2 == 14)
    {
        // 2^14 = 32 x 32 x 16 in registers, two LDS exchanges per transform, one plane at a time in LDS (k_coarse6.h)
        if (c->pre8400)
            hipLaunchKernelGGL(k_coarse6_w8400, dim3(grid), dim3(C2_THREADS), (C6_XCH + C4_TABN) * (int)sizeof(double), st, c->g, c->p, d_list, nlist, c->d_tw);
        else
            hipLaunchKernelGGL(k_coarse6, dim3(grid), dim3(C2_THREADS), C6_XCH * (int)sizeof(double), st, c->g, c->p, d_list, nlist, c->d_tw);
    }
    else
    {
        // 2^13 = 32 x 16 x 16 on 256 threads: two workgroups per CU (k_coarse6.h)
        const int grid2 = nlist < 2 * c->coarse2_grid ? nlist : 2 * c->coarse2_grid;
        hipLaunchKernelGGL(k_coarse6_13, dim3(grid2), dim3(256), C6_XCH13 * (int)sizeof(double), st, c->g, c->p, d_list, nlist, c->d_tw);
    }
}

// Work enqueued on `st` is ordered behind everything the bank enqueued before (on last_stream), and `st` becomes last_stream: the rule for
// every entry point that takes a caller stream and touches device state (jaero_write, jaero_discard_softbits) -- a discard on stream X
// followed by a rate-changing jaero_set_settings (which synchronises last_stream only) must not read the counters before the discard lands.
static int order_behind_last(jaero_ctx *c, hipStream_t st)
{
    if (st != c->last_stream)
    {
        if (!c->order_ev) HIPCHK(hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->order_ev, c->last_stream));
        HIPCHK(hipStreamWaitEvent(st, c->order_ev, 0));
    }
    c->last_stream = st;
    return 0;
}

extern "C" int jaero_write(jaero_ctx *c, const int16_t *pcm, int nsamples, int layout, int is_device_ptr, void *stream)
{
    if (!c || !pcm || nsamples < 0) return fail(JAERO_EINVAL, "jaero_write: bad arguments");
    if (nsamples == 0) return 0;
    if (nsamples > c->max_write) return fail(JAERO_EINVAL, "jaero_write: nsamples %d exceeds max_write_samples %d", nsamples, c->max_write);
    if (layout != JAERO_PCM_CHANNEL_MAJOR && layout != JAERO_PCM_FRAME_MAJOR) return fail(JAERO_EINVAL, "jaero_write: bad layout");
    if (c->poisoned) return fail(JAERO_EHIP, "jaero_write: an earlier write of this bank failed part-way (its schedule mirror is ahead of the device state): destroy the bank and create a new one");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    { const int rc = order_behind_last(c, st); if (rc) return rc; } // setters, discards and earlier writes were enqueued on last_stream
    // poisoned is raised where the first launch that advances device state or the schedule mirror is about to be enqueued (a failed copy of
    // the input, transpose or history push leaves both as they were: the caller may simply write again) and lowered when the whole write is in
    if (c->burst)
    {
        const int rc = burst_write(c, pcm, nsamples, layout, is_device_ptr, st);
        if (rc == 0) c->poisoned = false;
        return rc;
    }
    const JGeom &g = c->g;
    const int nch = g.nch, nchp = g.nchp;

    const int16_t *frames = nullptr;
    int stride = 0;
    const int16_t *dsrc = pcm;
    if (!is_device_ptr)
    {
        HIPCHK(hipMemcpyAsync(c->d_pcm_raw, pcm, sizeof(int16_t) * (size_t)nch * nsamples, hipMemcpyHostToDevice, st));
        dsrc = c->d_pcm_raw;
    }
    if (layout == JAERO_PCM_FRAME_MAJOR) { frames = dsrc; stride = nch; }
    else
    {
        const int pi = prof_begin(c, 2, st);
        hipLaunchKernelGGL(k_transpose_pcm, dim3(nchp / 64, (nsamples + 63) / 64), dim3(256), 0, st, dsrc, c->d_pcm_frames, nch, nchp, nsamples);
        LAUNCHCHK("k_transpose_pcm");
        prof_end(c, pi, st);
        frames = c->d_pcm_frames; stride = nchp;
    }

    c->poisoned = true; // from here on device state and mirror advance together or not at all
    if (c->pre8400)
    {
        // the whole write is prefiltered first (oqpskdemodulator.cpp:343-381); its oscillator takes the mean of mixer2's frequency over
        // the previous write (:607-608)
        // eight stretches per write once there are fewer channel groups than eight per SIMD (k_pre8400.h)
        const int ntb = (g.ngroups >= 8192 || nsamples < 512) ? 1 : 8;
        hipLaunchKernelGGL(k_pre8400_mix, dim3(g.ngroups, ntb), dim3(64), 0, st, g, c->p, c->pre, frames, stride, nsamples, c->pre_n0, c->pre_nprev);
        LAUNCHCHK("k_pre8400_mix");
        hipLaunchKernelGGL(k_pre8400_commit, dim3(g.ngroups), dim3(64), 0, st, g, c->p, c->pre_nprev);
        LAUNCHCHK("k_pre8400_commit");
        launch_pre8400_filter(g, c->p, c->pre, nsamples, c->pre_n0, c->pre_direct, st);
        LAUNCHCHK("the 8400 bps prefilter");
        c->pre_n0 += nsamples;
    }
    if (g.kind == JAERO_KIND_OQPSK) c->pre_nprev = nsamples; // (at every rate: a bank re-created for 8400 bps needs the length of the last write, rebank_with_carry_over)
    int pos = 0;
    while (pos < nsamples)
    {
        int n = 0, skip_a = 0, only_a = 0;
        const long long nb_before = c->m.nB_total;
        const int next = c->m.next_segment(pos, nsamples, n, skip_a, only_a);
        if (n > 0)
        {
            const long long nb_after = c->m.nB_total;
            c->m.nB_total = nb_before; // ring slots are those at the START of the segment
            const int pi = prof_begin(c, 0, st);
            launch_samples(c, frames + (size_t)pos * stride, stride, n, skip_a, only_a, st, pos);
            LAUNCHCHK("the sample loop");
            prof_end(c, pi, st);
            c->m.nB_total = nb_after;
        }
        if (only_a)
        {
            const int nlist = (int)c->m.fired.size();
            const int *dl = nullptr;
            if (nlist != nch)
            {
                HIPCHK(hipMemcpyAsync(c->d_chanlist, c->m.fired.data(), sizeof(int) * nlist, hipMemcpyHostToDevice, st));
                dl = c->d_chanlist;
            }
            const int pi = prof_begin(c, 1, st);
            launch_coarse(c, dl, nlist, st);
            LAUNCHCHK("the coarse-frequency estimate");
            prof_end(c, pi, st);
        }
        pos = next;
    }
    HIPCHK(hipGetLastError());
    c->poisoned = false;
    return 0;
}

// Host-only: the segmentation jaero_write would perform for one channel with the given flags over a sequence of
// writes; returns the global sample indices at which the coarse estimate fires (used by CPU tests; no device needed).
extern "C" int jaero_debug_schedule(int fft_power, int Fs, int cpu_reduce, const int *write_sizes, int nwrites,
                                    long long *trigger_samples, int cap, int *segments_out)
{
    if (!write_sizes || !trigger_samples || fft_power < 4 || fft_power > 20) return fail(JAERO_EINVAL, "jaero_debug_schedule: bad arguments");
    Mirror m;
    m.nch = 1; m.nchp = 64; m.nfft = 1 << fft_power; m.Fs_int = Fs;
    m.flags.assign(64, cpu_reduce ? JF_CPUREDUCE : 0); m.bbptr.assign(64, 0); m.cnt.assign(64, 0);
    long long base = 0;
    int ntrig = 0, nseg = 0;
    for (int w = 0; w < nwrites; w++)
    {
        int pos = 0;
        const int ns = write_sizes[w];
        while (pos < ns)
        {
            int n, sa, oa;
            const int next = m.next_segment(pos, ns, n, sa, oa);
            nseg++;
            if (oa) { if (ntrig < cap) trigger_samples[ntrig] = base + pos + n - 1; ntrig++; }
            pos = next;
        }
        base += ns;
    }
    if (segments_out) *segments_out = nseg;
    return ntrig;
}

// Host-only: the same for a whole group of channels that draw their flags independently and change them between writes -- the part of the
// scheduler that decides which lanes of a wavefront fire at which sample (Mirror::steps_to_trigger / next_segment / fired).
// events[k] = {before_write, channel (-1: all), kind, value}: kind 0 = setAFC/SQL/CPUReduce bits (JF_*), kind 1 = DCD, kind 2 = setSettings
// (the channel's ring position and counter restart); applied before write `before_write`.  Records (sample, channel) per firing.
extern "C" int jaero_debug_schedule_lanes(int fft_power, int Fs, int nch, const int *flags0, const int *write_sizes, int nwrites,
                                          const int *events, int nevents, long long *trig_sample_channel, int cap, int *segments_out)
{
    if (!write_sizes || !trig_sample_channel || !flags0 || (nevents > 0 && !events) || nch < 1 || fft_power < 4 || fft_power > 20)
        return fail(JAERO_EINVAL, "jaero_debug_schedule_lanes: bad arguments");
    Mirror m;
    m.nch = nch; m.nchp = (nch + 63) / 64 * 64; m.nfft = 1 << fft_power; m.Fs_int = Fs;
    m.flags.assign(m.nchp, 0); m.bbptr.assign(m.nchp, 0); m.cnt.assign(m.nchp, 0);
    for (int ch = 0; ch < nch; ch++) m.flags[ch] = flags0[ch] & (JF_AFC | JF_SQL | JF_CPUREDUCE | JF_DCD);
    long long base = 0;
    int ntrig = 0, nseg = 0;
    for (int w = 0; w < nwrites; w++)
    {
        for (int k = 0; k < nevents; k++)
        {
            const int *e = events + 4 * k;
            if (e[0] != w) continue;
            if (e[1] < -1 || e[1] >= nch) return fail(JAERO_EINVAL, "jaero_debug_schedule_lanes: bad channel in event %d", k);
            const int lo = e[1] < 0 ? 0 : e[1], hi = e[1] < 0 ? m.nchp : e[1] + 1;
            for (int ch = lo; ch < hi; ch++)
            {
                if (e[2] == 0) m.flags[ch] = (m.flags[ch] & JF_DCD) | (e[3] & (JF_AFC | JF_SQL | JF_CPUREDUCE)); // jaero_set_flags
                else if (e[2] == 1) m.flags[ch] = (m.flags[ch] & ~JF_DCD) | (e[3] ? JF_DCD : 0);               // jaero_set_dcd
                else { m.bbptr[ch] = 0; m.cnt[ch] = 0; }                                                        // jaero_set_settings
            }
        }
        int pos = 0;
        const int ns = write_sizes[w];
        while (pos < ns)
        {
            int n, sa, oa;
            const int next = m.next_segment(pos, ns, n, sa, oa);
            nseg++;
            if (oa)
                for (int ch : m.fired)
                {
                    if (ntrig < cap) { trig_sample_channel[2 * ntrig] = base + pos + n - 1; trig_sample_channel[2 * ntrig + 1] = ch; }
                    ntrig++;
                }
            pos = next;
        }
        base += ns;
    }
    if (segments_out) *segments_out = nseg;
    return ntrig;
}

// Test hook: the 8400 bps prefilter kernel (k_pre8400_fft) alone (one channel, unity up-mix) on n complex samples, kernel RRC(alpha, 2048 taps + 1, 48 kHz, fsym):
// out[m] = sum_k h[k] x[m - 2048 - k], what JFastFir::update returns for SetKernel(points, 4096) -- the operation the reference's
// own test vectors pin (JAERO/tests/jfastfir_tests.cpp:31-58, tests/test_jfastfir_vectors.py).
extern "C" int jaero_debug_prefilter(int device, const double *in_reim, int n, double alpha, double fsym, double *out_reim)
{
    if (!in_reim || !out_reim || n <= 0) return fail(JAERO_EINVAL, "jaero_debug_prefilter: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(JAERO_ENODEV, "no HIP device available");
    HIPCHK(hipSetDevice(device));
    const std::vector<double> taps = rrc_design(alpha, 2048, 48000.0, fsym);
    if ((int)taps.size() != PRE_K) return fail(JAERO_EINVAL, "prefilter design returned %zu taps", taps.size());
    JGeom g{}; g.nch = 1; g.nchp = 64; g.ngroups = 1;
    JPtrs p{};
    JPre q{};
    int ring = 1;
    while (ring < n + 3 * PRE_L + 64) ring <<= 1;
    q.ring = ring;
    q.cap = n;
    double2 *d_cis = nullptr; double *d_taps = nullptr;
    HIPCHK(hipMalloc((void **)&q.xring, sizeof(double2) * (size_t)ring * 64));
    HIPCHK(hipMalloc((void **)&q.cidx, sizeof(unsigned short) * (size_t)n * 64));
    HIPCHK(hipMalloc((void **)&q.out, sizeof(double2) * (size_t)n * 64));
    HIPCHK(hipMalloc((void **)&d_cis, sizeof(double2) * 4));
    HIPCHK(hipMalloc((void **)&d_taps, sizeof(double) * PRE_K));
    HIPCHK(hipMemset(q.xring, 0, sizeof(double2) * (size_t)ring * 64));
    HIPCHK(hipMemset(q.cidx, 0, sizeof(unsigned short) * (size_t)n * 64));
    HIPCHK(hipMalloc((void **)&q.hold, sizeof(long long) * 64));
    HIPCHK(hipMemset(q.hold, 0, sizeof(long long) * 64));
    const double2 one[4] = {{1.0, 0.0}, {1.0, 0.0}, {1.0, 0.0}, {1.0, 0.0}}; // table entry 0 = cis(0): the up-mix multiplies by its conjugate
    HIPCHK(hipMemcpy(d_cis, one, sizeof one, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_taps, taps.data(), sizeof(double) * PRE_K, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy2D(q.xring, sizeof(double2) * 4, in_reim, sizeof(double2), sizeof(double2), (size_t)n, hipMemcpyHostToDevice)); // channel 0 of every slot (PRE_XI)
    p.cis = d_cis; q.taps = d_taps;
    {
        double2 *dH = nullptr, *dtw = nullptr;
        int rc = fft4096_tables(taps, &dH, &dtw, (const void *)k_pre8400_fft);
        if (rc) return rc;
        q.H = dH; q.tw = dtw;
        launch_pre8400_filter(g, p, q, n, 0LL, false, 0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        hipFree(dH); hipFree(dtw);
    }
    HIPCHK(hipMemcpy2D(out_reim, sizeof(double2), q.out, sizeof(double2) * 4, sizeof(double2), (size_t)n, hipMemcpyDeviceToHost));
    hipFree(q.xring); hipFree(q.cidx); hipFree(q.out); hipFree(q.hold); hipFree(d_cis); hipFree(d_taps);
    return 0;
}

// Test hook: the prefiltered samples of the last write of an 8400 bps bank (cval_prefiltered, oqpskdemodulator.cpp:343-381), channel ch.
extern "C" int jaero_debug_read_prefiltered(jaero_ctx *c, int ch, double *out_reim, int n)
{
    if (!c || !out_reim || !c->pre8400 || ch < 0 || ch >= c->g.nch || n <= 0 || n > c->pre_nprev) return fail(JAERO_EINVAL, "jaero_debug_read_prefiltered: bad arguments");
    HIPCHK(hipStreamSynchronize(c->last_stream));
    HIPCHK(hipMemcpy2D(out_reim, sizeof(double2), c->pre.out + JD_G4(0, ch, c->pre.cap), sizeof(double2) * 4, sizeof(double2), (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------ outputs
static int check_overflow(jaero_ctx *c, int ch, int bit)
{
    int ov = 0;
    HIPCHK(hipMemcpy(&ov, c->o_overflow + ch, sizeof(int), hipMemcpyDeviceToHost));
    if (ov & bit)
    {
        int z = ov & ~bit;
        HIPCHK(hipMemcpy(c->o_overflow + ch, &z, sizeof(int), hipMemcpyHostToDevice));
        return fail(JAERO_EOVERFLOW, "channel %d overflowed its output buffer (flag %d); data was dropped", ch, bit);
    }
    return 0;
}

extern "C" int jaero_read_softbits(jaero_ctx *c, int ch, int16_t *dst, int cap, int *n)
{
    POISONCHK(c, "jaero_read_softbits");
    if (!c || !dst || !n || ch < 0 || ch >= c->o_nch || cap < 0) return fail(JAERO_EINVAL, "jaero_read_softbits: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    int cnt = 0;
    int *dcnt = c->o_soft_cnt + ch;
    HIPCHK(hipMemcpy(&cnt, dcnt, sizeof(int), hipMemcpyDeviceToHost));
    int pend = 0;
    if (c->o_nrx) HIPCHK(hipMemcpy(&pend, c->o_nrx + ch, sizeof(int), hipMemcpyDeviceToHost));
    const int emitted = cnt - pend;
    const int take = emitted < cap ? emitted : cap;
    int16_t *src = c->o_soft + (size_t)ch * c->o_soft_cap;
    if (take) HIPCHK(hipMemcpy(dst, src, sizeof(int16_t) * take, hipMemcpyDeviceToHost));
    if (take < cnt)
    {
        std::vector<int16_t> tmp(cnt - take);
        HIPCHK(hipMemcpy(tmp.data(), src + take, sizeof(int16_t) * (cnt - take), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(src, tmp.data(), sizeof(int16_t) * (cnt - take), hipMemcpyHostToDevice));
    }
    const int rest = cnt - take;
    HIPCHK(hipMemcpy(dcnt, &rest, sizeof(int), hipMemcpyHostToDevice));
    *n = take;
    return check_overflow(c, ch, 1);
}

extern "C" int jaero_read_softbits_all(jaero_ctx *c, int16_t *dst, int capc, int *counts)
{
    POISONCHK(c, "jaero_read_softbits_all");
    if (!c || !dst || !counts || capc <= 0) return fail(JAERO_EINVAL, "jaero_read_softbits_all: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->last_stream;
    const int nch = c->o_nch, nchp = c->o_nchp;
    const size_t need = (size_t)nch * capc;
    if (need > c->pack_elems)
    {
        int16_t *q = nullptr;
        if (hipMalloc((void **)&q, need * sizeof(int16_t)) != hipSuccess) return fail(JAERO_ENOMEM, "pack buffer");
        c->allocs.push_back(q);
        c->d_pack = q; c->pack_elems = need;
    }
    hipLaunchKernelGGL(k_pack_soft, dim3(nch), dim3(256), 0, st, c->o_soft_cnt, c->o_soft, c->o_soft_cap, c->d_pack, capc);
    HIPCHK(hipMemcpyAsync(dst, c->d_pack, need * sizeof(int16_t), hipMemcpyDeviceToHost, st));
    std::vector<int> cnt(nchp), ov(nchp);
    HIPCHK(hipMemcpyAsync(cnt.data(), c->o_soft_cnt, sizeof(int) * nchp, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(ov.data(), c->o_overflow, sizeof(int) * nchp, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int rc = 0;
    for (int ch = 0; ch < nch; ch++)
    {
        counts[ch] = cnt[ch] < capc ? cnt[ch] : capc;
        if (cnt[ch] > capc || (ov[ch] & 1)) rc = JAERO_EOVERFLOW;
    }
    if (c->burst)
        return fail(JAERO_ENOTSUP, "jaero_read_softbits_all: burst banks keep a pending (not yet emitted) tail per channel; use jaero_read_softbits");
    HIPCHK(hipMemsetAsync(c->o_soft_cnt, 0, sizeof(int) * nchp, st));
    if (rc)
    {
        HIPCHK(hipMemsetAsync(c->o_overflow, 0, sizeof(int) * nchp, st));
        return fail(rc, "at least one channel produced more soft bits than fit (cap_per_channel=%d or device capacity)", capc);
    }
    return 0;
}

// burst banks: only what the reference would have emitted (groups completed so far) is handed on; the pending tail stays
__global__ void k_burst_emitted(const int *__restrict__ cnt, const int *__restrict__ nrx, int *__restrict__ emitted, int nch)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < nch) emitted[ch] = cnt[ch] - nrx[ch];
}
// after the emitted part was consumed: move the pending tail to the front of the channel's buffer
__global__ void k_burst_keep_tail(int *__restrict__ cnt, const int *__restrict__ nrx, int16_t *__restrict__ soft, int cap, int nch)
{
    const int ch = blockIdx.x;
    if (ch >= nch) return;
    const int c = cnt[ch], pend = nrx[ch], first = c - pend;
    int16_t *row = soft + (size_t)ch * cap;
    // pend is at most one group (< 64 entries): one wavefront, read everything before writing
    const int t = threadIdx.x;
    int16_t v = 0;
    if (t < pend) v = row[first + t];
    __syncthreads();
    if (t < pend) row[t] = v;
    if (t == 0) cnt[ch] = pend;
}

extern "C" int jaero_softbits_view(jaero_ctx *c, void **dev_softbits, void **dev_counts, int *capacity)
{
    POISONCHK(c, "jaero_softbits_view");
    if (!c) return fail(JAERO_EINVAL, "null ctx");
    if (dev_softbits) *dev_softbits = c->o_soft;
    if (c->burst && dev_counts)
    {
        HIPCHK(hipSetDevice(c->device));
        if (!c->d_emitted)
        {
            void *q = nullptr;
            if (hipMalloc(&q, sizeof(int) * c->o_nchp) != hipSuccess) return fail(JAERO_ENOMEM, "emitted counts");
            c->allocs.push_back(q);
            c->d_emitted = (int *)q;
        }
        hipLaunchKernelGGL(k_burst_emitted, dim3((c->o_nch + 255) / 256), dim3(256), 0, c->last_stream, c->o_soft_cnt, c->o_nrx, c->d_emitted, c->o_nch);
        *dev_counts = c->d_emitted;
        if (capacity) *capacity = c->o_soft_cap;
        return 0;
    }
    if (dev_counts) *dev_counts = c->o_soft_cnt;
    if (capacity) *capacity = c->o_soft_cap;
    return 0;
}

extern "C" int jaero_discard_softbits(jaero_ctx *c, void *stream)
{
    POISONCHK(c, "jaero_discard_softbits");
    if (!c) return fail(JAERO_EINVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    { const int rc = order_behind_last(c, (hipStream_t)stream); if (rc) return rc; }
    if (c->burst)
    {
        // what jaero_softbits_view reported as emitted is gone; the not yet emitted tail of every channel moves to the front
        hipLaunchKernelGGL(k_burst_keep_tail, dim3(c->o_nch), dim3(64), 0, (hipStream_t)stream, c->o_soft_cnt, c->o_nrx, c->o_soft, c->o_soft_cap, c->o_nch);
        HIPCHK(hipGetLastError());
        return 0;
    }
    HIPCHK(hipMemsetAsync(c->o_soft_cnt, 0, sizeof(int) * c->o_nchp, (hipStream_t)stream));
    HIPCHK(hipMemsetAsync(c->o_sym_cnt, 0, sizeof(int) * c->o_nchp, (hipStream_t)stream));
    return 0;
}

extern "C" int jaero_read_status(jaero_ctx *c, int ch, jaero_status *stt)
{
    POISONCHK(c, "jaero_read_status");
    if (!c || !stt || ch < 0 || ch >= c->o_nch) return fail(JAERO_EINVAL, "jaero_read_status: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    if (c->burst) hipLaunchKernelGGL(k_status_burst, dim3(1), dim3(64), 0, c->last_stream, c->bg, c->bp, ch, 1, c->d_status);
    else hipLaunchKernelGGL(k_status, dim3(1), dim3(64), 0, c->last_stream, c->g, c->p, ch, 1, c->d_status);
    HIPCHK(hipMemcpyAsync(stt, c->d_status, sizeof(jaero_status), hipMemcpyDeviceToHost, c->last_stream));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    return 0;
}

static int read_rows(jaero_ctx *c, int ch, double *rows, int caprows, int *nrows, int *cnt_base, double *base, int cap, int w, int ovbit)
{
    if (!c || !rows || !nrows || ch < 0 || ch >= c->o_nch) return fail(JAERO_EINVAL, "bad arguments");
    if (!base || !cnt_base) return fail(JAERO_EINVAL, "this output was not enabled in jaero_create flags (or does not exist for this kind)");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    int cnt = 0;
    int *dcnt = cnt_base + ch;
    HIPCHK(hipMemcpy(&cnt, dcnt, sizeof(int), hipMemcpyDeviceToHost));
    const int take = cnt < caprows ? cnt : caprows;
    double *src = base + (size_t)ch * cap * w;
    if (take) HIPCHK(hipMemcpy(rows, src, sizeof(double) * w * take, hipMemcpyDeviceToHost));
    if (take < cnt)
    {
        std::vector<double> tmp((size_t)w * (cnt - take));
        HIPCHK(hipMemcpy(tmp.data(), src + (size_t)take * w, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(src, tmp.data(), sizeof(double) * tmp.size(), hipMemcpyHostToDevice));
    }
    const int rest = cnt - take;
    HIPCHK(hipMemcpy(dcnt, &rest, sizeof(int), hipMemcpyHostToDevice));
    *nrows = take;
    return check_overflow(c, ch, ovbit);
}
extern "C" int jaero_read_status_log(jaero_ctx *c, int ch, double *rows, int caprows, int *nrows)
{
    POISONCHK(c, "jaero_read_status_log");
    if (c && c->burst) return fail(JAERO_ENOTSUP, "burst banks have an event log (jaero_read_events), not a status log");
    return read_rows(c, ch, rows, caprows, nrows, c ? c->p.I + (size_t)I_LOG_CNT * c->g.nchp : nullptr, c ? c->p.slog : nullptr, c ? c->g.log_cap : 0, 6, 4);
}
extern "C" int jaero_read_symbols(jaero_ctx *c, int ch, double *rows, int caprows, int *nrows)
{
    POISONCHK(c, "jaero_read_symbols");
    return read_rows(c, ch, rows, caprows, nrows, c ? c->o_sym_cnt : nullptr, c ? c->o_sym : nullptr, c ? c->o_sym_cap : 0, 3, 2);
}
extern "C" int jaero_read_events(jaero_ctx *c, int ch, double *rows, int caprows, int *nrows)
{
    POISONCHK(c, "jaero_read_events");
    if (c && !c->burst) return fail(JAERO_ENOTSUP, "continuous banks have a status log (jaero_read_status_log), not an event log");
    return read_rows(c, ch, rows, caprows, nrows, c ? c->bp.I + (size_t)BI_EV_CNT * c->bg.nchp : nullptr, c ? c->bp.evlog : nullptr, c ? c->bg.ev_cap : 0, 3, 4);
}

// ------------------------------------------------------------------------------------------ Viterbi
// Two layouts of the same decoder: one block per wavefront (k_viterbi: ~57 SIMD cycles per step and block, fills the chip from a few
// thousand blocks) and one block per lane (k_viterbi_lanes: ~12 cycles per step and block, but needs >= 64 blocks per SIMD-wave to
// pay off).  The lane layout wins once the per-wavefront layout has more than ~14 waves queued per SIMD.
#define VL_MIN_BLOCKS 16384
static inline size_t viterbi_hist_bytes(int nblocks) { return (size_t)((nblocks + 63) / 64) * VT_CAP * 64 * sizeof(unsigned long long); }
static int g_viterbi_layout = 0; // test hook jaero_debug_viterbi_layout: 0 = by size (the product behaviour), 1 = wave, 2 = lanes
extern "C" int jaero_debug_viterbi_layout(int mode)
{
    if (mode < 0 || mode > 2) return fail(JAERO_EINVAL, "jaero_debug_viterbi_layout: mode must be 0 (by size), 1 (wave) or 2 (lanes)");
    g_viterbi_layout = mode;
    return 0;
}
static inline bool viterbi_use_lanes(int nblocks, int nsoft, int pad)
{
    if ((nsoft + pad) / 2 < 4 * VT_ORDER) return false;
    if (g_viterbi_layout == 1) return false;
    if (g_viterbi_layout == 2) return true;
    return nblocks >= VL_MIN_BLOCKS;
}
static void viterbi_launch(hipStream_t st, const uint8_t *d_soft, int nsoft, const uint8_t *d_ov, int pad, uint8_t *d_out, int out_stride,
                           int out_start, int out_want, int nblocks, const int *valid, unsigned long long *hist, int tiled = 0, int packed = 0, int force_lanes = 0, int pitch = 0, const int *lens = nullptr)
{
    if (pitch <= 0) pitch = nsoft; // distance between the rows of d_soft (row-major input only)
    if (hist && (tiled || force_lanes || viterbi_use_lanes(nblocks, nsoft, pad))) // tiled input / packed output exist only in the lane layout
    {
        const int waves = (nblocks + 63) / 64;
        int ncu = 256;
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
        if (waves <= ncu * 4) // one wavefront per SIMD is enough: use the entry point that cannot be stacked two to a SIMD
            hipLaunchKernelGGL(k_viterbi_lanes, dim3(waves), dim3(64), 0, st, d_soft, nsoft, d_ov, pad, d_out, out_stride, out_start, out_want, nblocks,
                               valid, hist, tiled, packed, pitch, lens);
        else
            hipLaunchKernelGGL(k_viterbi_lanes_x2, dim3(waves), dim3(64), 0, st, d_soft, nsoft, d_ov, pad, d_out, out_stride, out_start, out_want, nblocks,
                               valid, hist, tiled, packed, pitch, lens);
    }
    else
        hipLaunchKernelGGL(k_viterbi, dim3(nblocks), dim3(64), 0, st, d_soft, nsoft, d_ov, pad, d_out, out_stride, out_start, out_want, nblocks, valid, lens, pitch);
}
static int viterbi_run(int device, const uint8_t *soft, int nblocks, int nsoft, int pad, uint8_t *overlap, uint8_t *bits_out,
                       int out_stride, int out_start, int out_want, int is_device_ptr, hipStream_t st)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(JAERO_ENODEV, "no HIP device available");
    HIPCHK(hipSetDevice(device));
    const uint8_t *d_soft = soft; uint8_t *d_out = bits_out; uint8_t *d_ov = overlap;
    void *t_soft = nullptr, *t_out = nullptr, *t_ov = nullptr;
    const size_t out_bytes = (size_t)nblocks * out_stride;
    if (!is_device_ptr)
    {
        HIPCHK(hipMalloc(&t_soft, (size_t)nblocks * nsoft));
        HIPCHK(hipMalloc(&t_out, out_bytes));
        HIPCHK(hipMemcpyAsync(t_soft, soft, (size_t)nblocks * nsoft, hipMemcpyHostToDevice, st));
        d_soft = (const uint8_t *)t_soft; d_out = (uint8_t *)t_out;
        if (overlap)
        {
            HIPCHK(hipMalloc(&t_ov, (size_t)nblocks * 64));
            HIPCHK(hipMemcpyAsync(t_ov, overlap, (size_t)nblocks * 64, hipMemcpyHostToDevice, st));
            d_ov = (uint8_t *)t_ov;
        }
    }
    HIPCHK(hipMemsetAsync(d_out, 0, out_bytes, st));
    void *t_hist = nullptr;
    if (viterbi_use_lanes(nblocks, nsoft, pad)) HIPCHK(hipMallocAsync(&t_hist, viterbi_hist_bytes(nblocks), st));
    viterbi_launch(st, d_soft, nsoft, (const uint8_t *)d_ov, pad, d_out, out_stride, out_start, out_want, nblocks, nullptr, (unsigned long long *)t_hist);
    HIPCHK(hipGetLastError());
    if (t_hist) HIPCHK(hipFreeAsync(t_hist, st));
    if (!is_device_ptr)
    {
        HIPCHK(hipMemcpyAsync(bits_out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        hipFree(t_soft); hipFree(t_out); if (t_ov) hipFree(t_ov);
    }
    return 0;
}

extern "C" int jaero_viterbi_decode_soft(int device, const uint8_t *soft, int nblocks, int nsoft, uint8_t *bits_out, int is_device_ptr, void *stream)
{
    if (!soft || !bits_out || nblocks <= 0 || nsoft < 4 * VT_ORDER || (nsoft & 1)) return fail(JAERO_EINVAL, "jaero_viterbi_decode_soft: bad arguments");
    return viterbi_run(device, soft, nblocks, nsoft, 0, nullptr, bits_out, nsoft / 2, 0, nsoft / 2, is_device_ptr, (hipStream_t)stream);
}

__global__ void k_viterbi_overlap_update(const uint8_t *__restrict__ soft, int nsoft, uint8_t *__restrict__ overlap, int nstreams,
                                         const int *__restrict__ valid = nullptr, int tiled = 0, int pitch = 0)
{
    // soft_bits_overlap_buffer_uchar = soft_bits_in.right(62); resize(62)  (jconvolutionalcodec.cpp:197-198)
    const int b = blockIdx.x;
    if (b >= nstreams) return;
    if (valid && !valid[b]) return;
    const int k = 62;
    const int t = threadIdx.x;
    if (t < k)
    {
        uint8_t v = 0;
        // byte q of row b: row-major, or the tiled layout of k_viterbi_lanes ([wavefront][16-byte group][lane][16])
        auto at = [&](int q) -> uint8_t {
            return tiled ? soft[(size_t)(b >> 6) * 64 * nsoft + ((size_t)(q >> 4) * 64 + (b & 63)) * 16 + (q & 15)] : soft[(size_t)b * (pitch > 0 ? pitch : nsoft) + q];
        };
        if (nsoft >= k) v = at(nsoft - k + t);
        else if (t < nsoft) v = at(t);
        overlap[(size_t)b * 64 + t] = v;
    }
    if (t == 62) overlap[(size_t)b * 64 + 62] = (uint8_t)k;
}

extern "C" int jaero_viterbi_continuous(int device, const uint8_t *soft, int nstreams, int nsoft, int paddinglength, uint8_t *overlap_state,
                                        uint8_t *bits_out, int *nbits_out, int is_device_ptr, void *stream)
{
    if (!soft || !bits_out || !overlap_state || nstreams <= 0 || nsoft < 64 || (nsoft & 1) || paddinglength < 0 || (paddinglength & 1))
        return fail(JAERO_EINVAL, "jaero_viterbi_continuous: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int want = nsoft / 2;
    // host needs each stream's overlap length to report nbits (first call of a stream returns fewer bits)
    std::vector<uint8_t> hov;
    const uint8_t *ovh = overlap_state;
    if (is_device_ptr)
    {
        hov.resize((size_t)nstreams * 64);
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipMemcpyAsync(hov.data(), overlap_state, hov.size(), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        ovh = hov.data();
    }
    if (nbits_out)
        for (int b = 0; b < nstreams; b++)
        {
            const int total = (int)ovh[(size_t)b * 64 + 62] + nsoft + paddinglength;
            int nb = total / 2 - (paddinglength + 1);
            if (nb > want) nb = want;
            if (nb < 0) nb = 0;
            nbits_out[b] = nb;
        }
    int rc = viterbi_run(device, soft, nstreams, nsoft, paddinglength, overlap_state, bits_out, want, paddinglength + 1, want, is_device_ptr, st);
    if (rc) return rc;
    if (is_device_ptr)
    {
        hipLaunchKernelGGL(k_viterbi_overlap_update, dim3(nstreams), dim3(64), 0, st, soft, nsoft, overlap_state, nstreams);
        HIPCHK(hipGetLastError());
    }
    else
    {
        for (int b = 0; b < nstreams; b++)
        {
            uint8_t *ov = overlap_state + (size_t)b * 64;
            memset(ov, 0, 64);
            memcpy(ov, soft + (size_t)b * nsoft + nsoft - 62, 62);
            ov[62] = 62;
        }
    }
    return 0;
}

#include "aerol_host.h"
#include "ingest_host.h"
#include "edge_host.h"
