// TEST INFRASTRUCTURE ONLY: the device code of jaero_amd/csrc/aerolc.h (k_aerolc_bits, k_aerolc_post, k_aerolc_end_write) compiled as
// plain host functions and driven thread by thread, with the oracle's continuous Viterbi (oracle/viterbi_oracle.c) standing in for
// k_viterbi.  This checks the bank logic of those kernels -- per-channel state arrays, [slot][channel] layouts, ragged writes,
// rounds -- against oracle/aerol_oracle.c on the CPU, where no GPU is needed.  Built by tests/test_aerolc_emul.py with g++.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define __global__
#define __device__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __restrict__
struct emul_dim { int x; };
static thread_local emul_dim blockIdx, threadIdx, blockDim;
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
// k_aerolc_bulk orders its two stretches with a wavefront fence + vmcnt(0): here a workgroup's threads run one after the other
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __shared__ static // one workgroup at a time
#define CC_BULK_SYNC() ((void)0)
// AeroLcrc16::calcusingbytes as jaero_amd/csrc/k_aerol.h:aerol_crc16 has it
static inline unsigned aerol_crc16(const uint8_t *bytes, int n)
{
    unsigned crc = 0xFFFFu;
    for (int i = 0; i < n; i++)
    {
        unsigned mb = bytes[i];
        for (int k = 0; k < 8; k++)
        {
            const unsigned message_bit = mb & 1u;
            mb >>= 1;
            const unsigned crc_bit = crc & 1u;
            crc >>= 1;
            if (crc_bit ^ message_bit) crc ^= 0x8408u;
        }
    }
    return (~crc) & 0xFFFFu;
}
#define AEROLC_KERNELS_ONLY
#include "../../jaero_amd/csrc/aerolc.h"

extern "C" {
#include "../../oracle/viterbi_oracle.h"
}

struct Emul
{
    CGeom g;
    CPtrs p;
    std::vector<jo_codec *> codec;
    std::vector<void *> mem;
};
template <class T> static T *zalloc(Emul *e, size_t n) { T *q = (T *)calloc(n ? n : 1, sizeof(T)); e->mem.push_back(q); return q; }

extern "C" Emul *emul_create(int nch, int su_cap)
{
    Emul *e = new Emul();
    CGeom &g = e->g;
    g.nch = nch; g.nchp = (nch + 63) / 64 * 64; g.su_cap = su_cap > 0 ? su_cap : 192; g.v_cap = (g.su_cap + 2) / 3; g.ev_cap = 256;
    e->p.I = zalloc<int>(e, (size_t)CI_NFIELDS * g.nchp);
    e->p.B = zalloc<unsigned long long>(e, (size_t)4 * g.nchp);
    e->p.dep = zalloc<uint8_t>(e, (size_t)g.nchp * CC_PITCH);
    e->p.vbits = zalloc<uint8_t>(e, (size_t)g.nchp * (CC_NSOFT / 2) + 64);
    e->p.overlap = zalloc<uint8_t>(e, (size_t)g.nchp * 64);
    e->p.dl2 = zalloc<uint8_t>(e, (size_t)CC_PREV_PITCH * g.nchp + 64);
    e->p.sus = zalloc<int32_t>(e, (size_t)g.nchp * g.su_cap * 16);
    e->p.voice = zalloc<uint8_t>(e, (size_t)g.nchp * g.v_cap * 304);
    e->p.events = zalloc<long long>(e, (size_t)g.nchp * g.ev_cap * 3);
    uint8_t *scr = zalloc<uint8_t>(e, 5000);
    int state[15] = {1, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1};
    for (int k = 0; k < 5000; k++)
    {
        const int val0 = state[0] ^ state[14];
        scr[k] = (uint8_t)val0;
        for (int i = 14; i > 0; i--) state[i] = state[i - 1];
        state[0] = val0;
    }
    e->p.scr = scr;
    unsigned long long *scrf = zalloc<unsigned long long>(e, 50);
    for (int y = 0; y < 25; y++)
        for (int i = 0; i < 108; i++) scrf[2 * y + i / 64] |= (unsigned long long)(scr[109 * y + 1 + i] & 1) << (i % 64);
    e->p.scrf = scrf;
    for (size_t k = 0; k < (size_t)g.nchp * CC_PITCH; k++) if ((k % CC_PITCH) % 4 == 3) e->p.dep[k] = 128; // as aerolc_create
    for (int ch = 0; ch < g.nchp; ch++)
    {
        e->p.I[(size_t)CI_CNTR * g.nchp + ch] = 1000000000;
        e->p.I[(size_t)CI_EV_CNT * g.nchp + ch] = 1;
    }
    for (int ch = 0; ch < nch; ch++) e->codec.push_back(jo_codec_create(24));
    return e;
}
extern "C" void emul_destroy(Emul *e)
{
    for (void *q : e->mem) free(q);
    for (jo_codec *c : e->codec) jo_codec_destroy(c);
    delete e;
}
template <class F> static void launch(int nblocks, int nthreads, F f)
{
    blockDim.x = nthreads;
    for (int b = 0; b < nblocks; b++)
        for (int t = 0; t < nthreads; t++) { blockIdx.x = b; threadIdx.x = t; f(); }
}
// = aerolc_write: soft [nch][stride] int16, counts [nch]
extern "C" void emul_write(Emul *e, const int16_t *soft, const int *counts, int stride, int max_count)
{
    const CGeom &g = e->g;
    std::vector<int> cnt(g.nchp, 0);
    for (int ch = 0; ch < g.nch; ch++) cnt[ch] = counts[ch];
    const int rounds = max_count / (CC_FRAME - 112) + 2; // as aerolc_write
    for (int r = 0; r < rounds; r++)
    {
        launch(g.nchp / 64, 64, [&] { k_aerolc_bits(g, e->p, soft, cnt.data(), stride); });
        // (the product: one launch for both stretches and both phases, konly = phase = -1; here a launch boundary per stretch and, workgroup by
        // workgroup, every thread's staging before any thread's lookups)
        for (int k = 0; k < 2; k++)
            for (int b = 0; b < g.nch; b++)
                for (int ph = 0; ph < 2; ph++)
                    for (int t = 0; t < 64; t++) { blockDim.x = 64; blockIdx.x = b; threadIdx.x = t; k_aerolc_bulk(g, e->p, soft, stride, k, ph); }
        for (int ch = 0; ch < g.nch; ch++) // k_viterbi + k_viterbi_overlap_update for the channels with a complete frame
            if (e->p.I[(size_t)CI_HAS_BLOCK * g.nchp + ch])
            {
                uint8_t out[CC_NSOFT / 2 + 16];
                const int nb = jo_decode_continuous(e->codec[ch], e->p.dep + (size_t)ch * CC_PITCH, CC_NSOFT, out);
                memcpy(e->p.vbits + (size_t)ch * (CC_NSOFT / 2), out, (size_t)(nb < CC_NSOFT / 2 ? nb : CC_NSOFT / 2)); // unwritten tail keeps its old content, as on the GPU
            }
        launch(g.nchp / 64, 64, [&] { k_aerolc_post(g, e->p); });
    }
    launch(g.nchp / 64, 64, [&] { k_aerolc_end_write(g, e->p, cnt.data()); });
}
extern "C" void emul_tick(Emul *e) { launch(e->g.nchp / 64, 64, [&] { k_aerolc_tick_dcd(e->g, e->p, nullptr); }); }
// drain: returns the count and copies rows
extern "C" int emul_read(Emul *e, int ch, int which, void *dst, int caprows)
{
    const CGeom &g = e->g;
    const int field = which == 0 ? CI_SU_CNT : which == 1 ? CI_V_CNT : CI_EV_CNT;
    const size_t rowbytes = which == 0 ? 64 : which == 1 ? 304 : 24;
    const int cap = which == 0 ? g.su_cap : which == 1 ? g.v_cap : g.ev_cap;
    const char *base = which == 0 ? (const char *)e->p.sus : which == 1 ? (const char *)e->p.voice : (const char *)e->p.events;
    int &cnt = e->p.I[(size_t)field * g.nchp + ch];
    const int take = cnt < caprows ? cnt : caprows;
    memcpy(dst, base + (size_t)ch * cap * rowbytes, rowbytes * take);
    cnt = 0;
    return take;
}
