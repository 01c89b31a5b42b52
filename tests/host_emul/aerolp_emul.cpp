// TEST INFRASTRUCTURE ONLY: the device code of the continuous Aero-L bit pipeline (jaero_amd/csrc/k_aerol.h: k_aerol_bits<false> with
// aerol_bit_a / aerol_bit_b, k_aerol_post with aerol_frame_end, k_aerol_end_write) compiled as plain host functions and driven thread by
// thread in the rounds of jaero_aerol_write (aerol_host.h), with the deinterleaver written out here (the kernel's goes through LDS in
// wavefront lockstep) and the oracle's Decode_Continuous (oracle/viterbi_oracle.c) standing in for k_viterbi.  The jumping variant
// k_aerol_bits<true> needs k_aerol_scan / k_aerol_bulk, which are wavefront kernels: not emulated (the GPU tests cover them).
// Built by tests/test_aerolp_emul.py with g++ -Itests/host_emul/stub.
#include <vector>

#include "../../jaero_amd/csrc/k_aerol.h"

extern "C" {
#include "../../oracle/viterbi_oracle.h"
}

struct EmulP
{
    AGeom g;
    APtrs p;
    std::vector<jo_codec *> codec;
    std::vector<void *> mem;
};
template <class T> static T *zalloc(EmulP *e, size_t n)
{
    void *q = nullptr;
    if (posix_memalign(&q, 64, (n ? n : 1) * sizeof(T) + 64)) return nullptr;
    memset(q, 0, (n ? n : 1) * sizeof(T) + 64);
    e->mem.push_back(q);
    return (T *)q;
}

extern "C" EmulP *emulp_create(int nch, int fb, int su_cap)
{
    EmulP *e = new EmulP();
    AGeom &g = e->g;
    memset(&g, 0, sizeof(g));
    g.nch = nch; g.nchp = (nch + 63) / 64 * 64; g.fb = fb;
    // aerol_create (aerol_host.h), burst = 0, small bank (bits one per byte, k_aerol_post)
    switch (fb)
    {
    case 600: g.N = 6; g.dl2_sz = 576 - 6 + 1; g.NumberOfBits = 1152; g.BitsInHeader = 16; g.TotalNumberOfBits = 16 + 1152 + 32; g.oqpsk = 0; break;
    case 1200: g.N = 9; g.dl2_sz = 576 - 6 + 1; g.NumberOfBits = 1152; g.BitsInHeader = 16; g.TotalNumberOfBits = 16 + 1152 + 32; g.oqpsk = 0; break;
    default: g.N = 78; g.dl2_sz = 4992 - 6 + 1; g.NumberOfBits = 4992; g.BitsInHeader = 16 + 178; g.TotalNumberOfBits = 16 + 178 + 4992 + 64; g.oqpsk = 1; break;
    }
    g.blocksz = g.N * 64;
    g.idx_sat = (1000000000 - g.BitsInHeader) % g.blocksz;
    g.info_cap = g.NumberOfBits / 16 + 16;
    g.su_cap = su_cap > 0 ? su_cap : 32 * (g.NumberOfBits / 2 / 96) + 8; g.ev_cap = 256;
    APtrs &p = e->p;
    memset(&p, 0, sizeof(p));
    p.I = zalloc<int>(e, (size_t)AI_NFIELDS * g.nchp);
    p.rx = zalloc<uint8_t>(e, (size_t)g.nchp * g.blocksz);
    p.deint = zalloc<uint8_t>(e, (size_t)g.nchp * g.blocksz);
    p.vbits = zalloc<uint8_t>(e, (size_t)g.nchp * (g.blocksz / 2));
    p.dl2 = zalloc<uint8_t>(e, (size_t)g.nchp * g.dl2_sz);
    p.info = zalloc<uint8_t>(e, (size_t)g.nchp * g.info_cap);
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
    }
    for (int ch = 0; ch < nch; ch++) e->codec.push_back(jo_codec_create(24));
    return e;
}
extern "C" void emulp_destroy(EmulP *e)
{
    if (!e) return;
    for (jo_codec *c : e->codec) jo_codec_destroy(c);
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

extern "C" int emulp_write(EmulP *e, const int16_t *soft, const int *counts, int stride, int max_count)
{
    const AGeom &g = e->g;
    const APtrs &p = e->p;
    const int rounds = max_count / g.blocksz + 2;
    std::vector<uint8_t> out(g.blocksz / 2 + 64);
    for (int r = 0; r < rounds; r++)
    {
        each_thread(g.nchp, [&] { k_aerol_bits<false>(g, p, soft, counts, stride); });
        for (int ch = 0; ch < g.nch; ch++)
        {
            if (!ALD(AI_HAS_BLOCK)) continue;
            // AeroLInterleaver::deinterleave_ba as k_aerol_deint has it: out[j*64 + i] = block[((i*27) % 64) * N + j]
            const uint8_t *blk = p.rx + (size_t)ch * g.blocksz;
            uint8_t *dst = p.deint + (size_t)ch * g.blocksz;
            for (int i = 0; i < 64; i++)
                for (int j = 0; j < g.N; j++) dst[j * 64 + i] = blk[((i * 27) & 63) * g.N + j];
            const int nb = jo_decode_continuous(e->codec[ch], dst, g.blocksz, out.data());
            memcpy(p.vbits + (size_t)ch * (g.blocksz / 2), out.data(), (size_t)(nb < g.blocksz / 2 ? nb : g.blocksz / 2));
        }
        each_thread(g.nchp, [&] { k_aerol_post(g, p); });
    }
    each_thread(g.nchp, [&] { k_aerol_end_write(g, p, counts); });
    return 0;
}

// which 0: signal-unit rows (16 x int32), 1: event rows (3 x int64); returns the number of rows copied
extern "C" int emulp_read(EmulP *e, int ch, int which, void *rows, int cap)
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
extern "C" int emulp_overflow(EmulP *e, int ch)
{
    const AGeom &g = e->g;
    const APtrs &p = e->p;
    return ALD(AI_OVERFLOW);
}
