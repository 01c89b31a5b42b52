// TEST INFRASTRUCTURE ONLY: the device code of the R/T packet search (jaero_amd/csrc/k_aerol_burst.h: k_aerolb_bits, k_aerolb_post;
// k_aerol.h: k_aerol_end_write) compiled as plain host functions and driven thread by thread in the rounds of jaero_aerol_write
// (aerol_host.h), with the deinterleaver written out here (the kernel's goes through LDS in wavefront lockstep) and the oracle's
// Decode_soft (oracle/viterbi_oracle.c) standing in for k_viterbi.  This checks the bank logic -- per-channel state arrays, ragged
// writes, trial rounds, block words finished across launches -- against oracle/aerol_oracle.c on the CPU, where no GPU is needed.
// Built by tests/test_aerolb_emul.py with g++ -Itests/host_emul/stub.
#include <vector>

#define AEROLB_EMUL_COUNT
static long long g_aerolb_fast_groups = 0, g_aerolb_fast_fill_groups = 0; // groups of eight entries k_aerolb_bits<true> took in one go (all / while collecting)
#include "../../jaero_amd/csrc/k_aerol_burst.h"

extern "C" {
#include "../../oracle/viterbi_oracle.h"
}

struct EmulB
{
    AGeom g;
    APtrs p;
    jo_codec *codec;
    std::vector<void *> mem;
};
template <class T> static T *zalloc(EmulB *e, size_t n)
{
    void *q = nullptr;
    if (posix_memalign(&q, 64, (n ? n : 1) * sizeof(T))) return nullptr;
    memset(q, 0, (n ? n : 1) * sizeof(T));
    e->mem.push_back(q);
    return (T *)q;
}

extern "C" EmulB *emulb_create(int nch, int fb, int su_cap)
{
    EmulB *e = new EmulB();
    AGeom &g = e->g;
    memset(&g, 0, sizeof(g));
    g.nch = nch; g.nchp = (nch + 63) / 64 * 64; g.fb = fb;
    // aerol_create (aerol_host.h), burst = 1
    switch (fb)
    {
    case 600: g.N = 6; g.dl2_sz = 576 - 6 + 1; g.NumberOfBits = 1152; g.BitsInHeader = 16; g.oqpsk = 0; break;
    case 1200: g.N = 9; g.dl2_sz = 576 - 6 + 1; g.NumberOfBits = 1152; g.BitsInHeader = 16; g.oqpsk = 0; break;
    default: g.N = 78; g.dl2_sz = 4992 - 6 + 1; g.NumberOfBits = 4992; g.BitsInHeader = 16 + 178; g.oqpsk = 1; break;
    }
    g.burst = 1;
    g.TotalNumberOfBits = g.oqpsk ? fb : 3 * fb;
    g.blocksz = RT_BLOCKSZ;
    g.idx_sat = (1000000000 - g.BitsInHeader) % g.blocksz;
    g.info_cap = g.NumberOfBits / 16 + 16;
    g.su_cap = su_cap > 0 ? su_cap : 256; g.ev_cap = 256;
    APtrs &p = e->p;
    memset(&p, 0, sizeof(p));
    p.I = zalloc<int>(e, (size_t)AI_NFIELDS * g.nchp);
    p.rx = zalloc<uint8_t>(e, (size_t)g.nchp * g.blocksz);
    p.deint = zalloc<uint8_t>(e, (size_t)g.nchp * g.blocksz);
    p.vbits = zalloc<uint8_t>(e, (size_t)g.nchp * (g.blocksz / 2));
    p.sus = zalloc<int32_t>(e, (size_t)g.nchp * g.su_cap * 16);
    p.events = zalloc<long long>(e, (size_t)g.nchp * g.ev_cap * 3);
    uint8_t *scr = zalloc<uint8_t>(e, 5000);
    int state[15] = {1, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1}; // AeroLScrambler, as aerol_create
    for (int k = 0; k < 5000; k++)
    {
        const int val0 = state[0] ^ state[14];
        scr[k] = (uint8_t)val0;
        for (int i = 14; i > 0; i--) state[i] = state[i - 1];
        state[0] = val0;
    }
    p.scr = scr;
    for (int ch = 0; ch < g.nchp; ch++)
    {
        p.I[(size_t)AI_CNTR * g.nchp + ch] = 1000000000;
        p.I[(size_t)AI_BLOCKCNT * g.nchp + ch] = -1;
        p.I[(size_t)AI_EV_CNT * g.nchp + ch] = 1;
        p.I[(size_t)BI_RT_BLOCKPTR * g.nchp + ch] = 0;
        p.I[(size_t)BI_RT_LAST * g.nchp + ch] = RT_NOTHING;
    }
    e->codec = jo_codec_create(0);
    return e;
}
extern "C" void emulb_destroy(EmulB *e)
{
    if (!e) return;
    jo_codec_destroy(e->codec);
    for (void *q : e->mem) free(q);
    delete e;
}

template <class F> static void each_thread(int nthreads, F f)
{
    for (int t = 0; t < nthreads; t++)
    {
        blockIdx.x = t / 64; threadIdx.x = t % 64; blockDim.x = 64;
        f();
    }
}

// one jaero_aerol_write: soft [nch][stride] int16, counts [nch]; wide = -1: by alignment (as the library), 0 / 1: forced
extern "C" int emulb_write(EmulB *e, const int16_t *soft, const int *counts, int stride, int max_count, int wide)
{
    const AGeom &g = e->g;
    const APtrs &p = e->p;
    const bool aligned = ((((size_t)soft) | ((size_t)stride * 2)) & 15) == 0;
    if (wide == 1 && !aligned) return -1;
    const bool w = wide < 0 ? aligned : wide != 0;
    const int rounds = max_count / 192 + 4;
    std::vector<uint8_t> out(RT_BLOCKSZ / 2 + 64);
    for (int r = 0; r < rounds; r++)
    {
        if (w) each_thread(g.nchp, [&] { k_aerolb_bits<true>(g, p, soft, counts, stride); });
        else each_thread(g.nchp, [&] { k_aerolb_bits<false>(g, p, soft, counts, stride); });
        for (int ch = 0; ch < g.nch; ch++)
        {
            if (!ALD(AI_HAS_BLOCK)) continue;
            // deinterleave_ba(block, cols) / deinterleaveMSK_ba as k_aerolb_deint has them
            const int len = ALD(BI_TRIAL_LEN), cols = len / 64;
            const uint8_t *blk = p.rx + (size_t)ch * RT_BLOCKSZ;
            uint8_t *dst = p.deint + (size_t)ch * RT_BLOCKSZ;
            for (int lane = 0; lane < 64; lane++)
            {
                const int perm = (lane * 27) & 63;
                if (g.oqpsk)
                    for (int j = 0; j < cols; j++) dst[j * 64 + lane] = blk[perm * cols + j];
                else
                {
                    for (int j = 0; j < 5 && j < cols; j++) dst[j * 64 + lane] = blk[perm * 5 + j];
                    for (int proc = 5; proc + 3 <= cols; proc += 3)
                        for (int j = 0; j < 3; j++) dst[(proc + j) * 64 + lane] = blk[64 * proc + perm * 3 + j];
                }
            }
            // Decode_soft of this channel's length: decoded bit k -> vbits[k] for k < len / 2 (the last 6 are the post kernel's)
            const int nb = jo_decode_soft(e->codec, dst, len, out.data());
            uint8_t *vb = p.vbits + (size_t)ch * (RT_BLOCKSZ / 2);
            memcpy(vb, out.data(), (size_t)(nb < len / 2 ? nb : len / 2));
        }
        each_thread(g.nchp, [&] { k_aerolb_post(g, p); });
    }
    each_thread(g.nchp, [&] { k_aerol_end_write(g, p, counts); });
    return 0;
}

// which 0: packet rows (16 x int32), 1: event rows (3 x int64); returns the number of rows copied
extern "C" int emulb_read(EmulB *e, int ch, int which, void *rows, int cap)
{
    const AGeom &g = e->g;
    const APtrs &p = e->p;
    if (which == 0)
    {
        const int n = std::min(ALD(AI_SU_CNT), cap);
        memcpy(rows, p.sus + (size_t)ch * g.su_cap * 16, (size_t)n * 16 * sizeof(int32_t));
        return n;
    }
    const int n = std::min(ALD(AI_EV_CNT), cap);
    memcpy(rows, p.events + (size_t)ch * g.ev_cap * 3, (size_t)n * 3 * sizeof(long long));
    return n;
}
extern "C" long long emulb_fast_groups() { return g_aerolb_fast_groups; }
extern "C" long long emulb_fast_fill_groups() { return g_aerolb_fast_fill_groups; }
extern "C" int emulb_overflow(EmulB *e, int ch)
{
    const AGeom &g = e->g;
    const APtrs &p = e->p;
    return ALD(AI_OVERFLOW);
}
