// k_viterbi_lanes.h -- the same decoder as k_viterbi.h (libcorrect's correct_convolutional_decode_soft as JConvolutionalCodec
// drives it, JAERO/jconvolutionalcodec.cpp:98,169), laid out for BANKS: one code block per LANE, 64 blocks per wavefront.
//
// k_viterbi.h spends one wavefront per block (64 trellis states = 64 lanes): 2 ds_bpermute + a ballot + an LDS write per step and
// ~14 wave instructions per step and block.  A bank decodes tens of thousands of equally long blocks at once (the Aero-L pipeline:
// one 4992-soft-bit block per channel and frame), so here the 64 path metrics of a block live in 32 VGPRs of ITS lane as packed
// uint16 pairs R[m] = (pm[2m], pm[2m+1]) and the add-compare-select is VOP3P:
//     new(2m, 2m+1) = pk_min( bcast(pm[m]) + (d(t), d(t^3)),  bcast(pm[m+32]) + (d(t^3), d(t)) ),   t = table[2m]
// (both generator polynomials have bits 0 and 6 set, so table[2m+1] = table[2m|64] = table[2m]^3).  The broadcasts are the op_sel
// modifiers of v_pk_add_u16 -- no data movement.  uint16 wrap-around is the hardware's, as in libcorrect.  6 VOP3P instructions per
// pair of states and step = 192 per step for 64 blocks = 3 per block and step.
// The 64 decision bits of a step are two VGPRs; the 140-slice history ring is [wave][slice][lane] uint64 in global memory (written
// and read back as coalesced 512-byte rows -- libcorrect's traceback schedule depends only on the step count, so all lanes trace
// back at the same steps).  Blocks whose overlap length differs (the first block of a stream has none) run as separate passes.
#pragma once
#include "k_viterbi.h"
#include <utility>

typedef unsigned short vl_us2 __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr unsigned vl_par(unsigned x) { return (x ^ (x >> 1) ^ (x >> 2) ^ (x >> 3) ^ (x >> 4) ^ (x >> 5) ^ (x >> 6) ^ (x >> 7)) & 1u; }
__host__ __device__ constexpr unsigned vl_tab(unsigned sr) { return vl_par(sr & 109u) | (vl_par(sr & 79u) << 1); }

__device__ __forceinline__ unsigned vl_add_b0(unsigned a, unsigned b) // (a.lo + b.lo, a.lo + b.hi)
{
    unsigned r;
    asm("v_pk_add_u16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned vl_add_b1(unsigned a, unsigned b) // (a.hi + b.lo, a.hi + b.hi)
{
    unsigned r;
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned vl_min(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(vl_us2, a), __builtin_bit_cast(vl_us2, b)));
}
__device__ __forceinline__ unsigned vl_sub(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (vl_us2)(__builtin_bit_cast(vl_us2, a) - __builtin_bit_cast(vl_us2, b)));
}
template <int H> __device__ __forceinline__ unsigned vl_half(unsigned r) { return H ? (r >> 16) : (r & 0xFFFFu); }

// bit position of state s in the 64-bit decision word of a step
__device__ __forceinline__ unsigned vl_bitpos(unsigned s) { return ((s >> 1) & 15u) | ((s & 1u) << 4) | (s & 32u); }

// Two steady-state butterfly pairs as ONE asm block: new states (2M .. 2M+3) from old M, M+1 (the two halves of a = R[M>>1]) and old
// M+32, M+33 (b = R[(M>>1)+16]); decision bits or-ed into acc at bits M&15, (M&15)+1 (even states) and 16 + the same (odd states).
// bmL* = (d(t), d(t^3)), bmH* = (d(t^3), d(t)) with t = table[2M] (index 0) and table[2M+2] (index 1).  The two chains are interleaved
// so that no packed op is consumed by the very next instruction (the compiler keeps one wait state between dependent VOP3P ops on
// gfx950, and pads separate asm statements with s_nop because it cannot see into them).
template <int M>
__device__ __forceinline__ void vl_acs2(unsigned a, unsigned b, unsigned bmL0, unsigned bmH0, unsigned bmL1, unsigned bmH1, unsigned &n0, unsigned &n1,
                                        unsigned &acc)
{
    static_assert((M & 1) == 0, "even M");
    unsigned l0, h0, l1, h1;
    asm("v_pk_add_u16 %2, %7, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_add_u16 %4, %7, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_u16 %3, %8, %10 op_sel_hi:[0,1]\n\t"
        "v_pk_add_u16 %5, %8, %12 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_min_u16 %0, %2, %3\n\t"
        "v_pk_min_u16 %1, %4, %5\n\t"
        "v_pk_sub_u16 %2, %2, %0\n\t"
        "v_pk_sub_u16 %4, %4, %1\n\t"
        "v_pk_min_u16 %2, %2, 1 op_sel_hi:[1,0]\n\t"
        "v_pk_min_u16 %4, %4, 1 op_sel_hi:[1,0]\n\t"
        "v_lshl_or_b32 %6, %2, %13, %6\n\t"
        "v_lshl_or_b32 %6, %4, %14, %6"
        : "=&v"(n0), "=&v"(n1), "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "+v"(acc)
        : "v"(a), "v"(b), "v"(bmL0), "v"(bmH0), "v"(bmL1), "v"(bmH1), "i"(M & 15), "i"((M & 15) + 1));
}
template <int... K>
__device__ __forceinline__ void vl_acs_all(const unsigned (&R)[32], unsigned (&T)[32], const unsigned (&BM)[4], unsigned &accA, unsigned &accB,
                                           std::integer_sequence<int, K...>)
{
    (vl_acs2<2 * K>(R[K], R[K + 16], BM[vl_tab(4u * K)], BM[vl_tab(4u * K) ^ 3u], BM[vl_tab(4u * K + 2u)], BM[vl_tab(4u * K + 2u) ^ 3u], T[2 * K],
                    T[2 * K + 1], (K < 8) ? accA : accB),
     ...);
}

// MODE 0: steady state (ties -> low predecessor).  MODE -1: warm-up (no compare, no history).  MODE k = 1..6: the last six steps,
// only states that are multiples of 2^k are updated (ties -> high predecessor), the others keep their metric and record 0.
template <int MODE>
__device__ __forceinline__ unsigned long long vl_step(const unsigned (&R)[32], unsigned (&T)[32], unsigned s0, unsigned s1)
{
    unsigned BM[4];
    {
        const unsigned n0 = 255u - s0, n1 = 255u - s1;
        const unsigned d0 = s0 + s1, d1 = n0 + s1, d2 = s0 + n1, d3 = n0 + n1;
        BM[0] = d0 | (d3 << 16); BM[3] = d3 | (d0 << 16);
        BM[1] = d1 | (d2 << 16); BM[2] = d2 | (d1 << 16);
    }
    unsigned accA = 0, accB = 0;
    if (MODE == 0)
    {
        vl_acs_all(R, T, BM, accA, accB, std::make_integer_sequence<int, 16>{});
        return (unsigned long long)accA | ((unsigned long long)accB << 32);
    }
#pragma unroll
    for (int m = 0; m < 32; m++)
    {
        const unsigned p = vl_tab(2u * (unsigned)m);
        const unsigned a = R[m >> 1], b = R[(m >> 1) + 16];
        if (MODE == -1)
        {
            T[m] = (m & 1) ? vl_add_b1(a, BM[p]) : vl_add_b0(a, BM[p]);
        }
        else
        {
            const int half = 1 << (MODE > 0 ? MODE - 1 : 0); // state 2m is a multiple of 2^MODE  <=>  m is a multiple of 2^(MODE-1)
            if ((m % half) == 0)
            {
                const unsigned L = (m & 1) ? vl_add_b1(a, BM[p]) : vl_add_b0(a, BM[p]);
                const unsigned Hh = (m & 1) ? vl_add_b1(b, BM[p ^ 3u]) : vl_add_b0(b, BM[p ^ 3u]);
                const unsigned N = vl_min(L, Hh);
                T[m] = (N & 0xFFFFu) | (R[m] & 0xFFFF0000u);
                const unsigned bit = (((Hh - N) & 0xFFFFu) == 0u) ? 1u : 0u; // he <= le
                if (m < 16) accA |= bit << m;
                else accB |= bit << (m - 16);
            }
            else T[m] = R[m];
        }
    }
    return (unsigned long long)accA | ((unsigned long long)accB << 32);
}

// first minimum over the states that are multiples of `skip`; returns the state, `sub` = distances[best] (libcorrect's quirk: a
// state is only taken if strictly below 65535, otherwise best stays 0)
__device__ __forceinline__ unsigned vl_search(const unsigned (&R)[32], unsigned skip, unsigned &sub)
{
    unsigned m;
    if (skip == 1)
    {
        unsigned t = R[0];
#pragma unroll
        for (int k = 1; k < 32; k++) t = vl_min(t, R[k]);
        m = min(t & 0xFFFFu, t >> 16);
    }
    else
    {
        m = 0xFFFFu;
#pragma unroll
        for (int s = 0; s < 64; s += 2) // odd states are never multiples of skip >= 2
            if ((s & (skip - 1)) == 0) m = min(m, R[s >> 1] & 0xFFFFu);
    }
    unsigned best = 0;
#pragma unroll
    for (int s = 63; s >= 0; s--)
    {
        const unsigned v = (s & 1) ? (R[s >> 1] >> 16) : (R[s >> 1] & 0xFFFFu);
        if ((s & (skip - 1)) == 0 && v == m) best = s; // (s & (skip-1)) is wave-uniform
    }
    if (m >= 65535u) best = 0;
    sub = (m >= 65535u) ? (R[0] & 0xFFFFu) : m;
    return best;
}

__device__ __forceinline__ void vl_renorm(unsigned (&R)[32], unsigned skip, unsigned sub)
{
    if (skip == 1)
    {
        const unsigned s2 = sub | (sub << 16);
#pragma unroll
        for (int k = 0; k < 32; k++) R[k] = vl_sub(R[k], s2);
    }
    else
    {
#pragma unroll
        for (int s = 0; s < 64; s += 2)
            if ((s & (skip - 1)) == 0) R[s >> 1] = ((R[s >> 1] - sub) & 0xFFFFu) | (R[s >> 1] & 0xFFFF0000u);
    }
}

struct VlRun // wave-uniform bookkeeping of history_buffer
{
    int index, len, renorm, outpos;
};

// hist: [gridDim.x][VT_CAP][64] uint64 scratch.  Same stream convention as k_viterbi (overlap ++ soft ++ pad x 128).
__global__ __launch_bounds__(64) void k_viterbi_lanes(const uint8_t *__restrict__ soft, int nsoft, const uint8_t *__restrict__ overlap, int pad,
                                                      uint8_t *__restrict__ out, int out_stride, int out_start, int out_want, int nblocks,
                                                      const int *__restrict__ valid, unsigned long long *__restrict__ hist)
{
    const unsigned lane = threadIdx.x;
    const int b0 = blockIdx.x * 64 + (int)lane;
    const bool inrange = b0 < nblocks;
    const int b = inrange ? b0 : 0;
    bool todo = inrange && (!valid || valid[b] != 0);
    if (!__any(todo)) return;
    const uint8_t *in = soft + (size_t)b * nsoft;
    const uint8_t *ov = overlap ? overlap + (size_t)b * 64 : in;
    const int my_ovl = overlap ? (int)ov[62] : 0;
    uint8_t *o = out + (size_t)b * out_stride;
    unsigned long long *hw = hist + (size_t)blockIdx.x * VT_CAP * 64 + lane;

    while (__any(todo))
    {
        // this pass: the lanes whose overlap length equals that of the first pending lane
        const int ovl = __shfl(my_ovl, __ffsll((long long)__ballot(todo)) - 1);
        const bool mine = todo && my_ovl == ovl;
        todo = todo && !mine;
        const int total = ovl + nsoft + pad;
        const int sets = total / 2;

        auto getpair = [&](int i, unsigned &s0, unsigned &s1) { // soft bytes 2i, 2i+1 of the stream (ovl is even: 0 or 62)
            const int k = 2 * i;
            if (k + 1 < ovl) { const unsigned v = *(const unsigned short *)(ov + k); s0 = v & 255u; s1 = v >> 8; }
            else if (k >= ovl && k - ovl + 1 < nsoft) { const unsigned v = *(const unsigned short *)(in + (k - ovl)); s0 = v & 255u; s1 = v >> 8; }
            else
            {
                auto one = [&](int q) -> unsigned { return q < ovl ? ov[q] : (q - ovl < nsoft ? in[q - ovl] : 128u); };
                s0 = one(k); s1 = one(k + 1);
            }
        };

        unsigned R[32], T[32];
#pragma unroll
        for (int k = 0; k < 32; k++) R[k] = 0;
        VlRun h = {0, 0, 0, 0};

        auto traceback = [&](unsigned bestpath, int min_tb) {
            int index = h.index;
            const int f = h.len - min_tb;
            unsigned word = 0; // decoded bits (one per byte) collected newest first = descending output index; stored four at a time
            int cnt = 0;
#pragma unroll 4
            for (int j = 0; j < h.len; j++)
            {
                index = (index == 0) ? VT_CAP - 1 : index - 1;
                const unsigned long long w = hw[(size_t)index * 64];
                const unsigned hb = (unsigned)(w >> vl_bitpos(bestpath)) & 1u;
                bestpath = (bestpath | (hb << 6)) >> 1;
                if (j >= min_tb)
                {
                    const int k = h.outpos + (f - 1 - (j - min_tb)) - out_start; // fetched[] is newest first, written reversed
                    word = (word << 8) | hb;
                    cnt++;
                    if (cnt == 4 || j == h.len - 1)
                    {
                        if (mine)
                        {
                            if (cnt == 4 && k >= 0 && k + 3 < out_want) __builtin_memcpy(o + k, &word, 4);
                            else
                                for (int t = 0; t < cnt; t++)
                                    if (k + t >= 0 && k + t < out_want) o[k + t] = (uint8_t)(word >> (8 * t));
                        }
                        cnt = 0;
                        word = 0;
                    }
                }
            }
            h.outpos += f;
            h.len -= f;
        };
        auto after = [&](unsigned (&P)[32], unsigned long long w, unsigned skip) { // history_buffer_process_skip
            hw[(size_t)h.index * 64] = w;
            h.index++;
            if (h.index == VT_CAP) h.index = 0;
            h.renorm++;
            h.len++;
            if (h.renorm == VT_RENORM)
            {
                h.renorm = 0;
                unsigned sub;
                const unsigned best = vl_search(P, skip, sub);
                vl_renorm(P, skip, sub);
                if (h.len == VT_CAP) traceback(best, VT_MINTB);
            }
            else if (h.len == VT_CAP)
            {
                unsigned sub;
                const unsigned best = vl_search(P, skip, sub);
                traceback(best, VT_MINTB);
            }
        };
        auto copy = [&]() {
#pragma unroll
            for (int k = 0; k < 32; k++) R[k] = T[k];
        };

        int i = 0;
        unsigned s0, s1;
        for (; i < VT_ORDER - 1 && i < sets; i++) // warm-up
        {
            getpair(i, s0, s1);
            vl_step<-1>(R, T, s0, s1);
            copy();
        }
        const int nend = sets - VT_ORDER + 1; // first tail step
        auto single = [&]() {
            getpair(i, s0, s1);
            const unsigned long long w = vl_step<0>(R, T, s0, s1);
            copy();
            after(R, w, 1);
            i++;
        };
        // steady state.  Rows that are 16-byte aligned (the Aero-L bank's are): single steps until step i starts a 16-byte group of
        // the block, then 8 steps per 16-byte load (the next group requested while this one is decoded), two steps per iteration
        // (R -> T -> R).  Everything else, and the ends, one step at a time.
        const bool rows16 = ((((size_t)soft) | (size_t)nsoft) & 15) == 0;
        if (rows16)
        {
            while (i < nend && (2 * i < ovl || ((2 * i - ovl) & 15) != 0)) single();
            int ngroups = 0;
            if (i < nend) ngroups = min((nend - i) / 8, (nsoft - (2 * i - ovl)) / 16);
            if (ngroups > 0)
            {
                const uint4 *cp = (const uint4 *)__builtin_assume_aligned(in + (2 * i - ovl), 16);
                uint4 nx = cp[0];
                for (int c = 0; c < ngroups; c++)
                {
                    uint4 cur = nx;
                    if (c + 1 < ngroups) nx = cp[c + 1];
#pragma nounroll
                    for (int q = 0; q < 4; q++)
                    {
                        const unsigned wd = cur.x;
                        cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
                        unsigned long long w = vl_step<0>(R, T, wd & 255u, (wd >> 8) & 255u);
                        after(T, w, 1);
                        w = vl_step<0>(T, R, (wd >> 16) & 255u, wd >> 24);
                        after(R, w, 1);
                    }
                    i += 8;
                }
            }
        }
        while (i < nend) single();
        // tail: i = sets-6 .. sets-1, skip = 2, 4, .. 64
#define VL_TAIL(K)                                               \
    if (i < sets && sets - i == VT_ORDER - (K))                  \
    {                                                            \
        getpair(i, s0, s1);                                      \
        const unsigned long long w = vl_step<K>(R, T, s0, s1);   \
        copy();                                                  \
        after(R, w, 1u << (K));                                  \
        i++;                                                     \
    }
        VL_TAIL(1) VL_TAIL(2) VL_TAIL(3) VL_TAIL(4) VL_TAIL(5) VL_TAIL(6)
#undef VL_TAIL
        traceback(0u, 0); // history_buffer_flush
    }
}
