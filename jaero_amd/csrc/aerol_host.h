// aerol_host.h -- C ABI of the Aero-L bit pipeline bank; included at the end of jaero_hip.hip.
#pragma once

struct jaero_aerol_ctx
{
    int device = 0;
    AGeom g{};
    APtrs p{};
    std::vector<void *> allocs;
    int16_t *d_soft = nullptr; int *d_counts = nullptr; int stage_stride = 0;
    unsigned long long *d_vhist = nullptr; // k_viterbi_lanes history scratch (large banks only)
    hipStream_t last_stream = nullptr;
    // HIP-event timing of the three kernel classes (0 = k_aerol_bits, 1 = k_viterbi + overlap update, 2 = k_aerol_post)
    bool prof = false;
    double prof_ms[3] = {0, 0, 0};
    int prof_n[3] = {0, 0, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_which;
    void *cmode = nullptr; // fb = 8400: C-channel state (aerolc_state, aerolc.h); g / p above stay unused
};

static void aprof_begin(jaero_aerol_ctx *c, int which, hipStream_t st)
{
    if (!c->prof) return;
    hipEvent_t a, b;
    if (c->ev_which.size() < c->ev_pool.size()) { a = c->ev_pool[c->ev_which.size()].first; }
    else
    {
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        c->ev_pool.push_back({a, b});
    }
    c->ev_which.push_back(which);
    hipEventRecord(c->ev_pool[c->ev_which.size() - 1].first, st);
}
static void aprof_end(jaero_aerol_ctx *c, hipStream_t st)
{
    if (!c->prof || c->ev_which.empty()) return;
    hipEventRecord(c->ev_pool[c->ev_which.size() - 1].second, st);
}
static void aprof_collect(jaero_aerol_ctx *c)
{
    for (size_t i = 0; i < c->ev_which.size(); i++)
    {
        float ms = 0;
        hipEventSynchronize(c->ev_pool[i].second);
        if (hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second) == hipSuccess) { c->prof_ms[c->ev_which[i]] += ms; c->prof_n[c->ev_which[i]]++; }
    }
    c->ev_which.clear();
}
extern "C" int jaero_aerol_profile_enable(jaero_aerol_ctx *c, int on)
{
    if (!c) return fail(JAERO_EINVAL, "null ctx");
    c->prof = on != 0;
    return 0;
}
extern "C" int jaero_aerol_profile_read(jaero_aerol_ctx *c, int which, double *total_ms, int *launches, int reset)
{
    if (!c || which < 0 || which > 2) return fail(JAERO_EINVAL, "jaero_aerol_profile_read: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    aprof_collect(c);
    if (total_ms) *total_ms = c->prof_ms[which];
    if (launches) *launches = c->prof_n[which];
    if (reset) { c->prof_ms[which] = 0; c->prof_n[which] = 0; }
    return 0;
}

template <class T>
static int aalloc(jaero_aerol_ctx *c, T **ptr, size_t count)
{
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(JAERO_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    e = hipMemset(q, 0, bytes);
    if (e != hipSuccess) return fail(JAERO_EHIP, "hipMemset failed: %s", hipGetErrorString(e));
    c->allocs.push_back(q);
    *ptr = (T *)q;
    return 0;
}

static void aerolc_free(jaero_aerol_ctx *c);
extern "C" void jaero_aerol_destroy(jaero_aerol_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (void *q : c->allocs) hipFree(q);
    for (auto &e : c->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    aerolc_free(c);
    delete c;
}

#include "aerolc.h"
static void aerolc_free(jaero_aerol_ctx *c) { delete (aerolc_state *)c->cmode; c->cmode = nullptr; }

static int aerol_create(int device, int nchannels, int fb, int max_softbits_per_write, int su_capacity, int burst, jaero_aerol_ctx **out)
{
    if (!out || nchannels <= 0 || max_softbits_per_write <= 0) return fail(JAERO_EINVAL, "jaero_aerol_create: bad arguments");
    *out = nullptr;
    const bool cmode = fb == 8400 && !burst; // C channel (aerolc.h)
    if (fb != 600 && fb != 1200 && fb != 10500 && !cmode) return fail(JAERO_ENOTSUP, "jaero_aerol_create: fb must be 600, 1200, 10500 or (continuous mode) 8400");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(JAERO_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(JAERO_ENODEV, "device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    jaero_aerol_ctx *c = new jaero_aerol_ctx();
    c->device = device;
    if (cmode)
    {
        int rc = aerolc_create(c, nchannels, su_capacity);
        if (!rc) rc = aalloc(c, &c->d_soft, (size_t)nchannels * max_softbits_per_write);
        if (!rc) rc = aalloc(c, &c->d_counts, (size_t)(nchannels + 63) / 64 * 64);
        if (rc) { jaero_aerol_destroy(c); return rc; }
        c->stage_stride = max_softbits_per_write;
        c->g.nch = nchannels; c->g.nchp = (nchannels + 63) / 64 * 64; c->g.fb = fb;
        HIPCHK(hipDeviceSynchronize());
        *out = c;
        return 0;
    }
    AGeom &g = c->g;
    g.nch = nchannels; g.nchp = (nchannels + 63) / 64 * 64; g.fb = fb;
    // AeroL::setSettings (JAERO/aerol.cpp:990-1072), burstmode = false
    switch (fb)
    {
    case 600: g.N = 6; g.dl2_sz = 576 - 6 + 1; g.NumberOfBits = 1152; g.BitsInHeader = 16; g.TotalNumberOfBits = 16 + 1152 + 32; g.oqpsk = 0; break;
    case 1200: g.N = 9; g.dl2_sz = 576 - 6 + 1; g.NumberOfBits = 1152; g.BitsInHeader = 16; g.TotalNumberOfBits = 16 + 1152 + 32; g.oqpsk = 0; break;
    default: g.N = 78; g.dl2_sz = 4992 - 6 + 1; g.NumberOfBits = 4992; g.BitsInHeader = 16 + 178; g.TotalNumberOfBits = 16 + 178 + 4992 + 64; g.oqpsk = 1; break;
    }
    g.blocksz = g.N * 64;
    g.burst = burst ? 1 : 0;
    if (burst)
    {
        // setSettings(fb, true) (aerol.cpp:996-1003,1062-1070): one second of bits as frame countdown; the block is the R/T packet
        // collector's (RTChannelDeleaveFECScram: up to 95 interleaver columns)
        g.TotalNumberOfBits = g.oqpsk ? fb : 3 * fb; // 1 s (10500 bps) / 3 s (600, 1200 bps) of bits
        g.blocksz = RT_BLOCKSZ;
    }
    g.idx_sat = (1000000000 - g.BitsInHeader) % g.blocksz;
    g.info_cap = g.NumberOfBits / 16 + 16;
    if (su_capacity <= 0) su_capacity = burst ? 256 : 32 * (g.NumberOfBits / 2 / 96) + 8; // 32 frames (burst: 256 packet rows) between reads
    g.su_cap = su_capacity; g.ev_cap = 256;
    int rc;
#define AA(ptr, count) do { if ((rc = aalloc(c, &(ptr), (size_t)(count)))) { jaero_aerol_destroy(c); return rc; } } while (0)
    AA(c->p.I, (size_t)AI_NFIELDS * g.nchp);
    AA(c->p.rx, (size_t)g.nchp * g.blocksz);
    AA(c->p.deint, (size_t)g.nchp * g.blocksz);
    AA(c->p.vbits, (size_t)g.nchp * (g.blocksz / 2));
    AA(c->p.overlap, (size_t)g.nchp * 64);
    AA(c->p.dl2, (size_t)g.nchp * g.dl2_sz);
    AA(c->p.info, (size_t)g.nchp * g.info_cap);
    AA(c->p.sus, (size_t)g.nchp * g.su_cap * 16);
    AA(c->p.events, (size_t)g.nchp * g.ev_cap * 3);
    uint8_t *d_scr = nullptr;
    AA(d_scr, 5000);
    c->stage_stride = max_softbits_per_write;
    AA(c->d_soft, (size_t)g.nch * max_softbits_per_write);
    AA(c->d_counts, g.nchp);
    unsigned *d_scrw = nullptr;
    AA(d_scrw, 5000 / 32 + 2);
    if (!burst && viterbi_use_lanes(g.nch, g.blocksz, 24))
    {
        // large bank: one block per lane in the Viterbi, tiled deinterleaver output, decoded bits and delay line packed 32 per word
        AA(c->d_vhist, viterbi_hist_bytes(g.nch) / sizeof(unsigned long long));
        g.packed = 1;
        // deinterleaver output: row-major.  The tiled layout ([wavefront][16-byte group][lane][16], k_viterbi_lanes reads 8 x 1 KiB per chunk)
        // was built when cold rows cost the decoder 0.85 ms; with its chunk prefetch it no longer gains anything and the tiled writes cost
        // 0.14 ms (3.51 vs 3.36 ms per step): not used (its switch left the library in round 3; the kernels keep the code path).
        g.tiled = 0;
        g.dl2_words = (g.dl2_sz + 31) / 32 + 1;
        AA(c->p.dl2w, (size_t)g.nchp * g.dl2_words);
    }
    // R/T packet search in a large bank: the trial decodes (each channel's own length) one block per lane as well, bits out one per byte
    if (burst && viterbi_use_lanes(g.nch, 128, 0)) AA(c->d_vhist, viterbi_hist_bytes(g.nch) / sizeof(unsigned long long));
#undef AA
    c->p.scr = d_scr;
    {
        // AeroLScrambler (JAERO/aerol.h:397-420)
        std::vector<uint8_t> scr(5000);
        int state[15] = {1, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1};
        for (int k = 0; k < 5000; k++)
        {
            const int val0 = state[0] ^ state[14];
            scr[k] = (uint8_t)val0;
            for (int i = 14; i > 0; i--) state[i] = state[i - 1];
            state[0] = val0;
        }
        HIPCHK(hipMemcpy(d_scr, scr.data(), 5000, hipMemcpyHostToDevice));
        std::vector<unsigned> scrw(5000 / 32 + 2, 0u);
        for (int k = 0; k < 5000; k++) scrw[k >> 5] |= (unsigned)scr[k] << (k & 31);
        HIPCHK(hipMemcpy(d_scrw, scrw.data(), scrw.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        c->p.scrw = d_scrw;
        // AeroL constructor state (aerol.cpp:904-977): cntr = 1000000000, blockcnt = -1, DataCarrierDetect(false) emitted
        std::vector<int> I((size_t)AI_NFIELDS * g.nchp, 0);
        std::vector<long long> ev((size_t)g.nchp * g.ev_cap * 3, 0);
        for (int ch = 0; ch < g.nchp; ch++)
        {
            I[(size_t)AI_CNTR * g.nchp + ch] = 1000000000;
            I[(size_t)AI_BLOCKCNT * g.nchp + ch] = -1;
            I[(size_t)AI_EV_CNT * g.nchp + ch] = 1; // row 0 = [0, DCD, 0]
        }
        if (burst)
            for (int ch = 0; ch < g.nchp; ch++)
            {
                I[(size_t)BI_RT_BLOCKPTR * g.nchp + ch] = 0;       // RTChannelDeleaveFECScram(): resetblockptr()
                I[(size_t)BI_RT_LAST * g.nchp + ch] = RT_NOTHING;
            }
        HIPCHK(hipMemcpy(c->p.I, I.data(), I.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->p.events, ev.data(), ev.size() * sizeof(long long), hipMemcpyHostToDevice));
    }
    HIPCHK(hipDeviceSynchronize());
    *out = c;
    return 0;
}

extern "C" int jaero_aerol_create(int device, int nchannels, int fb, int max_softbits_per_write, int su_capacity, jaero_aerol_ctx **out)
{
    return aerol_create(device, nchannels, fb, max_softbits_per_write, su_capacity, 0, out);
}
extern "C" int jaero_aerol_create_burst(int device, int nchannels, int fb, int max_softbits_per_write, int packet_row_capacity, jaero_aerol_ctx **out)
{
    return aerol_create(device, nchannels, fb, max_softbits_per_write, packet_row_capacity, 1, out);
}
extern "C" int jaero_aerol_read_packets(jaero_aerol_ctx *c, int ch, int32_t *rows, int caprows, int *nrows);

// = processDemodulatedSoftBits for every channel: soft[ch * stride + k], k < counts[ch]
extern "C" int jaero_aerol_write(jaero_aerol_ctx *c, const int16_t *soft, const int *counts, int stride, int max_count, int is_device_ptr, void *stream)
{
    if (!c || !soft || !counts || stride <= 0 || max_count < 0 || max_count > stride) return fail(JAERO_EINVAL, "jaero_aerol_write: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    c->last_stream = st;
    const AGeom &g = c->g;
    const int16_t *dsoft = soft;
    const int *dcounts = counts;
    if (!is_device_ptr)
    {
        if (stride > c->stage_stride) return fail(JAERO_EINVAL, "jaero_aerol_write: stride %d exceeds max_softbits_per_write %d", stride, c->stage_stride);
        for (int ch = 0; ch < g.nch; ch++) // a count above max_count would be cut short by the round budget, one above stride read past the row
            if (counts[ch] < 0 || counts[ch] > max_count) return fail(JAERO_EINVAL, "jaero_aerol_write: counts[%d] = %d outside [0, max_count = %d]", ch, counts[ch], max_count);
        HIPCHK(hipMemcpyAsync(c->d_soft, soft, sizeof(int16_t) * (size_t)g.nch * stride, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(c->d_counts, counts, sizeof(int) * g.nch, hipMemcpyHostToDevice, st));
        dsoft = c->d_soft; dcounts = c->d_counts;
    }
    if (max_count == 0) return 0;
    if (c->cmode) return aerolc_write(c, dsoft, dcounts, stride, max_count, st);
    if (g.burst)
    {
        // R/T packet search: a round per trial length a channel can reach in this write (every 192 soft bits, plus 128 and 320)
        const int rounds = max_count / 192 + 4;
        const int *valid = c->p.I + (size_t)AI_HAS_BLOCK * g.nchp;
        const int *lens = c->p.I + (size_t)BI_TRIAL_LEN * g.nchp;
        const dim3 grid(g.nchp / 64), block(64);
        for (int r = 0; r < rounds; r++)
        {
            aprof_begin(c, 0, st);
            if (((((size_t)dsoft) | ((size_t)stride * 2)) & 15) == 0) hipLaunchKernelGGL(k_aerolb_bits<true>, grid, block, 0, st, g, c->p, dsoft, dcounts, stride);
            else hipLaunchKernelGGL(k_aerolb_bits<false>, grid, block, 0, st, g, c->p, dsoft, dcounts, stride);
            hipLaunchKernelGGL(k_aerolb_deint, dim3((g.nch + 3) / 4), dim3(256), 0, st, g, c->p);
            aprof_end(c, st);
            aprof_begin(c, 1, st);
            // trial lengths are 128, 320, 512, .. (k_aerolb_bits): never below the lane layout's minimum of 4 * VT_ORDER steps
            viterbi_launch(st, (const uint8_t *)c->p.deint, RT_BLOCKSZ, (const uint8_t *)nullptr, 0, c->p.vbits, RT_BLOCKSZ / 2, 0, RT_BLOCKSZ / 2, g.nch, valid,
                           c->d_vhist, 0, 0, c->d_vhist != nullptr, 0, lens);
            aprof_end(c, st);
            aprof_begin(c, 2, st);
            hipLaunchKernelGGL(k_aerolb_post, grid, block, 0, st, g, c->p);
            aprof_end(c, st);
        }
        hipLaunchKernelGGL(k_aerol_end_write, grid, block, 0, st, g, c->p, dcounts);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // every round finishes at most one interleaver block per channel (the reference completes a block -- Viterbi, descrambling,
    // CRC and its data-carrier-detect update -- before it looks at the next soft bit)
    const int rounds = max_count / g.blocksz + 2;
    const int *valid = c->p.I + (size_t)AI_HAS_BLOCK * g.nchp;
    const dim3 grid(g.nchp / 64), block(64);
    const bool bulk = g.oqpsk != 0; // 10.5 kbps: locked channels jump over the body of a frame (k_aerol_bits / k_aerol_bulk)
    if (bulk) hipLaunchKernelGGL(k_aerol_scan, dim3((g.nch + 3) / 4), dim3(256), 0, st, g, c->p, dsoft, dcounts, stride);
    for (int r = 0; r < rounds; r++)
    {
        aprof_begin(c, 0, st);
        if (bulk)
        {
            hipLaunchKernelGGL(k_aerol_bits<true>, grid, block, 0, st, g, c->p, dsoft, dcounts, stride);
            hipLaunchKernelGGL(k_aerol_bulk, dim3((g.nch + 3) / 4), dim3(256), 0, st, g, c->p, dsoft, stride);
        }
        else hipLaunchKernelGGL(k_aerol_bits<false>, grid, block, 0, st, g, c->p, dsoft, dcounts, stride);
        hipLaunchKernelGGL(k_aerol_deint, dim3((g.nch + 3) / 4), dim3(256), 0, st, g, c->p);
        aprof_end(c, st);
        aprof_begin(c, 1, st);
        viterbi_launch(st, (const uint8_t *)c->p.deint, g.blocksz, (const uint8_t *)c->p.overlap, 24, c->p.vbits, g.blocksz / 2, 25, g.blocksz / 2,
                       g.nch, valid, c->d_vhist, g.tiled, g.packed /* bits out, 32 per word */, g.packed /* lane layout */);
        hipLaunchKernelGGL(k_viterbi_overlap_update, dim3(g.nch), dim3(64), 0, st, (const uint8_t *)c->p.deint, g.blocksz, c->p.overlap, g.nch, valid, g.tiled);
        aprof_end(c, st);
        aprof_begin(c, 2, st);
        if (g.packed) hipLaunchKernelGGL(k_aerol_post_packed, grid, block, 0, st, g, c->p);
        else hipLaunchKernelGGL(k_aerol_post, grid, block, 0, st, g, c->p);
        aprof_end(c, st);
    }
    hipLaunchKernelGGL(k_aerol_end_write, grid, block, 0, st, g, c->p, dcounts);
    HIPCHK(hipGetLastError());
    return 0;
}

static int aerol_read_rows(jaero_aerol_ctx *c, int ch, void *rows, int caprows, int *nrows, int cnt_field, const void *base, int cap, size_t rowbytes, int ovbit)
{
    if (!c || !rows || !nrows || ch < 0 || ch >= c->g.nch || caprows < 0) return fail(JAERO_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    int cnt = 0;
    int *dcnt = c->p.I + (size_t)cnt_field * c->g.nchp + ch;
    HIPCHK(hipMemcpy(&cnt, dcnt, sizeof(int), hipMemcpyDeviceToHost));
    const int take = cnt < caprows ? cnt : caprows;
    char *src = (char *)base + (size_t)ch * cap * rowbytes;
    if (take) HIPCHK(hipMemcpy(rows, src, rowbytes * take, hipMemcpyDeviceToHost));
    if (take < cnt)
    {
        std::vector<char> tmp(rowbytes * (size_t)(cnt - take));
        HIPCHK(hipMemcpy(tmp.data(), src + rowbytes * take, tmp.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(src, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    }
    const int rest = cnt - take;
    HIPCHK(hipMemcpy(dcnt, &rest, sizeof(int), hipMemcpyHostToDevice));
    *nrows = take;
    int ov = 0;
    int *dov = c->p.I + (size_t)AI_OVERFLOW * c->g.nchp + ch;
    HIPCHK(hipMemcpy(&ov, dov, sizeof(int), hipMemcpyDeviceToHost));
    if (ov & ovbit)
    {
        const int z = ov & ~ovbit;
        HIPCHK(hipMemcpy(dov, &z, sizeof(int), hipMemcpyHostToDevice));
        return fail(JAERO_EOVERFLOW, "Aero-L channel %d overflowed an output buffer (flag %d); rows were dropped", ch, ovbit);
    }
    return 0;
}
extern "C" int jaero_aerol_read_sus(jaero_aerol_ctx *c, int ch, int32_t *rows, int caprows, int *nrows)
{
    if (c && c->cmode) { aerolc_state *cs = (aerolc_state *)c->cmode; return aerolc_read(c, ch, rows, caprows, nrows, CI_SU_CNT, cs->p.sus, cs->g.su_cap, 16 * sizeof(int32_t), 1); }
    if (c && c->g.burst) return fail(JAERO_ENOTSUP, "jaero_aerol_read_sus: burst-mode bank (use jaero_aerol_read_packets)");
    return aerol_read_rows(c, ch, rows, caprows, nrows, AI_SU_CNT, c ? c->p.sus : nullptr, c ? c->g.su_cap : 0, 16 * sizeof(int32_t), 1);
}
extern "C" int jaero_aerol_read_packets(jaero_aerol_ctx *c, int ch, int32_t *rows, int caprows, int *nrows)
{
    if (c && !c->g.burst) return fail(JAERO_ENOTSUP, "jaero_aerol_read_packets: not a burst-mode bank (use jaero_aerol_read_sus)");
    return aerol_read_rows(c, ch, rows, caprows, nrows, AI_SU_CNT, c ? c->p.sus : nullptr, c ? c->g.su_cap : 0, 16 * sizeof(int32_t), 1);
}
extern "C" int jaero_aerol_read_events(jaero_aerol_ctx *c, int ch, long long *rows, int caprows, int *nrows)
{
    if (c && c->cmode) { aerolc_state *cs = (aerolc_state *)c->cmode; return aerolc_read(c, ch, rows, caprows, nrows, CI_EV_CNT, cs->p.events, cs->g.ev_cap, 3 * sizeof(long long), 2); }
    return aerol_read_rows(c, ch, rows, caprows, nrows, AI_EV_CNT, c ? c->p.events : nullptr, c ? c->g.ev_cap : 0, 3 * sizeof(long long), 2);
}
// = AeroL::updateDCD (aerol.cpp:1109-1122), which the reference drives from a 1 s wall-clock QTimer: the caller ticks it once per
// second of signal time.  dcd_out (optional, [nchannels]) receives the datacd flags afterwards.
__global__ void k_aerol_tick_dcd(const AGeom g, const APtrs p, int *dcd_out)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= g.nch) return;
    int dcdcount = ALD(AI_DCDCOUNT), datacd = ALD(AI_DATACD), ev_cnt = ALD(AI_EV_CNT), overflow = ALD(AI_OVERFLOW);
    if (dcdcount > 0) dcdcount -= 3;
    else if (dcdcount < 0) dcdcount = 0;
    if (datacd && !dcdcount)
    {
        datacd = 0;
        const long long bitidx = ((long long)(unsigned)ALD(AI_NBITS_LO)) | ((long long)ALD(AI_NBITS_HI) << 32);
        aerol_event(g, p, ch, ev_cnt, overflow, bitidx, 0, 0);
    }
    ALD(AI_DCDCOUNT) = dcdcount; ALD(AI_DATACD) = datacd; ALD(AI_EV_CNT) = ev_cnt; ALD(AI_OVERFLOW) = overflow;
    if (dcd_out) dcd_out[ch] = datacd;
}
extern "C" int jaero_aerol_tick_dcd(jaero_aerol_ctx *c, int *dcd_out_host)
{
    if (!c) return fail(JAERO_EINVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    if (c->cmode)
    {
        aerolc_state *cs = (aerolc_state *)c->cmode;
        hipLaunchKernelGGL(k_aerolc_tick_dcd, dim3(cs->g.nchp / 64), dim3(64), 0, c->last_stream, cs->g, cs->p, dcd_out_host ? c->d_counts : nullptr);
        if (dcd_out_host)
        {
            HIPCHK(hipMemcpyAsync(dcd_out_host, c->d_counts, sizeof(int) * cs->g.nch, hipMemcpyDeviceToHost, c->last_stream));
            HIPCHK(hipStreamSynchronize(c->last_stream));
        }
        HIPCHK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_aerol_tick_dcd, dim3(c->g.nchp / 64), dim3(64), 0, c->last_stream, c->g, c->p, dcd_out_host ? c->d_counts : nullptr);
    if (dcd_out_host)
    {
        HIPCHK(hipMemcpyAsync(dcd_out_host, c->d_counts, sizeof(int) * c->g.nch, hipMemcpyDeviceToHost, c->last_stream));
        HIPCHK(hipStreamSynchronize(c->last_stream));
    }
    HIPCHK(hipGetLastError());
    return 0;
}
