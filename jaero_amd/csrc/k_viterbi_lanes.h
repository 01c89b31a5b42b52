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
// Soft bytes reach the steps through LDS (64-step chunks of every lane's row, [group][lane] uint4), so the step loop has no vmcnt
// wait: the history store of a step is never waited for except in the traceback.
// Measured (MI355X, 65536 blocks of 5078 soft bytes, one wavefront per SIMD): 1.7-1.8 ms per launch = 0.70 ns per block and
// step, of which ~1.35 ms is the step loop (a lone wavefront issues one VALU instruction per ~2.2 ns, 225 instructions per step) and
// the rest the tracebacks (2 x one history-load latency each, ~10 instructions per slice).  Every load whose latency is exposed is
// either prefetched (soft chunks, hand-issued into AGPRs) or batched (history slices, 72 at a time): with the compiler's own
// placement the same kernel took 2.5-2.7 ms whenever its rows and history were cold, 1.8 ms warm.  k_viterbi.h on the same input:
// 11.5 ms.
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

#define VL_TB_BATCH 72 // history slices fetched together in the traceback (a multiple of 4; VT_CAP = 2 batches, 144 VGPRs)
static_assert(VL_TB_BATCH % 4 == 0, "the traceback stores four decoded bits at a time");

// hist: [gridDim.x][VT_CAP][64] uint64 scratch.  Same stream convention as k_viterbi (overlap ++ soft ++ pad x 128).
__device__ __forceinline__ void vl_decode(const uint8_t *__restrict__ soft, int nsoft, const uint8_t *__restrict__ overlap, int pad,
                                          uint8_t *__restrict__ out, int out_stride, int out_start, int out_want, int nblocks,
                                          const int *__restrict__ valid, unsigned long long *__restrict__ hist, int tiled, uint4 *lds_soft, int packed, int pitch,
                                          const int *__restrict__ lens)
{
    const unsigned lane = threadIdx.x;
    const int b0 = blockIdx.x * 64 + (int)lane;
    const bool inrange = b0 < nblocks;
    const int b = inrange ? b0 : 0;
    bool todo = inrange && (!valid || valid[b] != 0);
    if (!__any(todo)) return;
    // input rows: row-major [block][pitch] (pitch >= nsoft; a pitch that is a multiple of 16 and >= nsoft rounded up to 16 keeps the
    // fast path below for a row length that is not, e.g. the 5460 soft symbols of an Aero-L C-channel frame in rows of 5472), or
    // (tiled; nsoft a multiple of 16) [wavefront][16-byte group][lane][16]: the 64 rows a
    // wavefront decodes interleaved in 16-byte pieces, so that every chunk load is 8 x 1 KiB contiguous (the Aero-L deinterleaver
    // writes this layout; with one row per lane a load touches 64 pages and the same kernel runs 0.85 ms slower on cold rows)
    const uint8_t *in = tiled ? soft + (size_t)blockIdx.x * 64 * nsoft + lane * 16 : soft + (size_t)b * pitch;
    const int gstride = tiled ? 64 : 1; // distance between consecutive 16-byte groups of a row, in groups
    auto inbyte = [&](int q) -> unsigned { return in[(size_t)(q >> 4) * gstride * 16 + (q & 15)]; };
    const uint8_t *ov = overlap ? overlap + (size_t)b * 64 : in;
    const int my_ovl = overlap ? (int)ov[62] : 0;
    // lens (optional, row-major input): block b's own length <= nsoft (an R/T packet trial, k_aerolb_bits); its decoded bits end at lens[b] / 2
    const int my_len = lens ? lens[b] : nsoft;
    uint8_t *o = out + (size_t)b * out_stride;
    unsigned long long *hw = hist + (size_t)blockIdx.x * VT_CAP * 64 + lane;
    // a partial last 16-byte group of a row is read whole, which the pitch must cover
    const bool rows16 = ((((size_t)soft) | (size_t)pitch) & 15) == 0 && (tiled || pitch >= (nsoft + 15) / 16 * 16); // every row 16-byte aligned (always so when tiled)

    while (__any(todo))
    {
        // this pass: the lanes whose overlap length (and block length) equal those of the first pending lane
        const int first = __ffsll((long long)__ballot(todo)) - 1;
        const int ovl = __builtin_amdgcn_readfirstlane(__shfl(my_ovl, first)); // SGPR: all control flow below is scalar
        const int rowlen = lens ? __builtin_amdgcn_readfirstlane(__shfl(my_len, first)) : nsoft;
        const int want = lens ? min(out_want, rowlen / 2) : out_want;
        const int ngroups = (rowlen + 15) / 16; // 16-byte groups of this pass's rows
        const bool mine = todo && my_ovl == ovl && my_len == rowlen;
        todo = todo && !mine;
        const int total = ovl + rowlen + pad;
        const int sets = total / 2;

        auto getpair = [&](int i, unsigned &s0, unsigned &s1) { // soft bytes 2i, 2i+1 of the stream, any position (slow)
            auto one = [&](int q) -> unsigned { return q < ovl ? ov[q] : (q - ovl < rowlen ? inbyte(q - ovl) : 128u); };
            s0 = one(2 * i); s1 = one(2 * i + 1);
        };
        // Fast source for the steps whose two soft bytes lie inside a 16-byte aligned block: 128-byte chunks (64 steps) of every lane's
        // row are staged in LDS as [group][lane] uint4, fetched one chunk ahead into registers.  The per-step reads are then LDS reads
        // (lgkmcnt): the history store of every step (vmcnt) is never waited for outside the traceback.
        const int fast_lo = (ovl + 1) / 2, fast_hi = (rows16 && (ovl & 1) == 0) ? (ovl + rowlen) / 2 : 0; // steps [fast_lo, fast_hi)
        const uint4 *cp = (const uint4 *)__builtin_assume_aligned(in, 16);
        const unsigned short *lp = (const unsigned short *)lds_soft + lane * 8;
        // The next chunk is requested as soon as the current one is in LDS, into registers, with hand-written loads: the compiler
        // sinks an ordinary prefetch down to its use (and then every chunk costs a full memory latency, ~0.9 ms per launch when the
        // rows are cold).  They land in accumulation VGPRs, which nothing else here uses, and are not touched before the explicit
        // vmcnt(0) below.
        typedef unsigned vl_u4 __attribute__((ext_vector_type(4)));
        vl_u4 pre[8];
        int chunk = -1, pre_chunk = -1;
        auto issue = [&](int c) {
#pragma unroll
            for (int g = 0; g < 8; g++)
            {
                const int gi = c * 8 + g;
                const uint4 *a = cp + (size_t)(gi < ngroups ? gi : ngroups - 1) * gstride;
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=a"(pre[g]) : "v"(a) : "memory"); // straight into accumulation VGPRs
            }
            pre_chunk = c;
        };
        auto need_chunk = [&](int c) {
            if (c == chunk) return;
            if (pre_chunk != c) issue(c);
            // the wait "produces" the eight values, so that no copy of them can be scheduled above it
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+a"(pre[0]), "+a"(pre[1]), "+a"(pre[2]), "+a"(pre[3]), "+a"(pre[4]), "+a"(pre[5]), "+a"(pre[6]), "+a"(pre[7])
                         :
                         : "memory");
#pragma unroll
            for (int g = 0; g < 8; g++) ((vl_u4 *)lds_soft)[g * 64 + lane] = pre[g];
            chunk = c;
            if ((c + 1) * 8 < ngroups) issue(c + 1);
        };
        auto ldspair = [&](int t, unsigned &s0, unsigned &s1) { // step slot t of the staged chunk
            const unsigned v = lp[(t >> 3) * 512 + (t & 7)];
            s0 = v & 255u; s1 = v >> 8;
        };

        unsigned R[32], T[32];
#pragma unroll
        for (int k = 0; k < 32; k++) R[k] = 0;
        VlRun h = {0, 0, 0, 0};

        // history_buffer_traceback.  Slices are fetched VL_TB_BATCH at a time (their addresses do not depend on the path).  Iteration j
        // (newest slice first) yields decoded bit k(j) = K0 - j once j >= min_tb; four iterations make one 4-byte store (one bit per
        // byte, ascending k = descending j); groups that straddle min_tb, len or the output window go byte by byte.
        // packed output (one bit per decoded bit, 32-bit words, bit k of a row in word k>>5 at position k&31): the decoded index is the
        // same for all lanes, so word boundaries are wave-uniform.  A traceback hands down bits from high k to low k; the word it starts
        // in is finished by the NEXT traceback (which continues above it), the word it ends in was started by the previous one.
        unsigned pk_cur = 0, pk_top = 0, pk_pend = 0;
        int pk_cur_idx = -1, pk_top_idx = -1, pk_pend_idx = -1;
        bool pk_is_top = false;
        unsigned *ow = (unsigned *)o;
        auto pk_store = [&](int idx, unsigned v) { if (mine) ow[idx] = v; };
        auto pk_bits = [&](int k, unsigned v) { // k descending within a traceback; v = the bits k, k+1, .. (all in word k>>5)
            const int widx = k >> 5;
            if (pk_cur_idx < 0) { pk_cur = 0; pk_cur_idx = widx; pk_is_top = true; }
            else if (widx != pk_cur_idx)
            {
                if (pk_is_top) { pk_top = pk_cur; pk_top_idx = pk_cur_idx; pk_is_top = false; }
                else pk_store(pk_cur_idx, pk_cur);
                pk_cur = 0; pk_cur_idx = widx;
            }
            pk_cur |= v << (k & 31);
        };
        auto pk_end = [&]() { // after the last bit of a traceback
            if (pk_cur_idx < 0) return;
            if (pk_pend_idx == pk_cur_idx) pk_cur |= pk_pend;
            else if (pk_pend_idx >= 0) pk_store(pk_pend_idx, pk_pend);
            if (pk_is_top) { pk_pend = pk_cur; pk_pend_idx = pk_cur_idx; }
            else { pk_store(pk_cur_idx, pk_cur); pk_pend = pk_top; pk_pend_idx = pk_top_idx; }
            pk_cur_idx = -1; pk_is_top = false;
        };

        auto traceback = [&](unsigned bestpath, int min_tb) {
            const int len = h.len, f = len - min_tb;
            const int K0 = h.outpos + f - 1 + min_tb - out_start;
            int index = h.index;
#pragma nounroll
            for (int j0 = 0; j0 < len; j0 += VL_TB_BATCH)
            {
                unsigned long long wv[VL_TB_BATCH];
#pragma unroll
                for (int u = 0; u < VL_TB_BATCH; u++)
                {
                    index = (index == 0) ? VT_CAP - 1 : index - 1;
                    wv[u] = hw[(size_t)index * 64]; // slices beyond len are read (valid scratch) and their bits discarded
                }
#pragma unroll
                for (int g = 0; g < VL_TB_BATCH / 4; g++)
                {
                    unsigned word = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++)
                    {
                        const unsigned hb = (unsigned)(wv[4 * g + u] >> vl_bitpos(bestpath)) & 1u;
                        bestpath = (bestpath | (hb << 6)) >> 1;
                        word = (word << 8) | hb;
                    }
                    const int j = j0 + 4 * g, klow = K0 - (j + 3);
                    if (packed)
                    {
                        if (j >= min_tb && j + 3 < len && klow >= 0 && klow + 3 < want && (klow >> 5) == ((klow + 3) >> 5))
                            pk_bits(klow, ((word * 0x00204081u) >> 21) & 15u); // bytes 0..3 of `word` -> bits 0..3
                        else
                        {
#pragma unroll
                            for (int u = 0; u < 4; u++)
                            {
                                const int jj = j + u, k = K0 - jj;
                                if (jj >= min_tb && jj < len && k >= 0 && k < want) pk_bits(k, (word >> (8 * (3 - u))) & 1u);
                            }
                        }
                    }
                    else if (j >= min_tb && j + 3 < len && klow >= 0 && klow + 3 < want)
                    {
                        if (mine) __builtin_memcpy(o + klow, &word, 4);
                    }
                    else
                    {
#pragma unroll
                        for (int u = 0; u < 4; u++)
                        {
                            const int jj = j + u, k = K0 - jj;
                            if (jj >= min_tb && jj < len && k >= 0 && k < want && mine) o[k] = (uint8_t)(word >> (8 * (3 - u)));
                        }
                    }
                }
            }
            if (packed) pk_end();
            h.outpos += f;
            h.len -= f;
            // slices fetched beyond len are never read: retire them here, or the compiler guards the first reuse of their registers --
            // inside the step loop -- with a vmcnt(0) that then also waits for every step's history store
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        };
        auto record = [&](unsigned long long w) { // history_buffer_process_skip minus the renormalise / traceback events
            hw[(size_t)h.index * 64] = w;
            h.index++;
            if (h.index == VT_CAP) h.index = 0;
        };
        auto copy = [&]() {
#pragma unroll
            for (int k = 0; k < 32; k++) R[k] = T[k];
        };

        int i = 0;
        unsigned s0, s1, c0, c1;
        for (; i < VT_ORDER - 1 && i < sets; i++) // warm-up
        {
            getpair(i, s0, s1);
            vl_step<-1>(R, T, s0, s1);
            copy();
        }
        const int nend = sets - VT_ORDER + 1; // first tail step
        // Steady state + tail.  libcorrect renormalises every VT_RENORM steps and traces back whenever VT_CAP slices are buffered: both
        // are functions of the step count alone, so the steps run in event-free stretches (two per iteration, R -> T -> R) and the
        // events are handled at one place, on R.
        for (;;)
        {
            unsigned skip = 1;
            if (i < nend)
            {
                int run = min(nend - i, min(VT_RENORM - h.renorm, VT_CAP - h.len));
                if (i >= fast_lo && i < fast_hi)
                {
                    const int pbyte = 2 * i - ovl;
                    need_chunk(pbyte >> 7);
                    int t = (pbyte & 127) >> 1;
                    run = min(run, min(64 - t, fast_hi - i));
                    h.renorm += run;
                    h.len += run;
                    i += run;
                    unsigned ra = 0x8080u, rb = 0x8080u; // raw byte pairs; split at use so the LDS latency hides behind a pair of steps
                    if (run >= 2) { ra = lp[(t >> 3) * 512 + (t & 7)]; rb = lp[((t + 1) >> 3) * 512 + ((t + 1) & 7)]; }
                    for (; run >= 2; run -= 2, t += 2)
                    {
                        unsigned na = 0x8080u, nb = 0x8080u;
                        if (run >= 4) { na = lp[((t + 2) >> 3) * 512 + ((t + 2) & 7)]; nb = lp[((t + 3) >> 3) * 512 + ((t + 3) & 7)]; }
                        unsigned long long w = vl_step<0>(R, T, ra & 255u, ra >> 8);
                        record(w);
                        w = vl_step<0>(T, R, rb & 255u, rb >> 8);
                        record(w);
                        ra = na; rb = nb;
                    }
                    if (run)
                    {
                        ldspair(t, s0, s1);
                        const unsigned long long w = vl_step<0>(R, T, s0, s1);
                        copy();
                        record(w);
                    }
                }
                else
                {
                    getpair(i, s0, s1);
                    const unsigned long long w = vl_step<0>(R, T, s0, s1);
                    copy();
                    record(w);
                    h.renorm++;
                    h.len++;
                    i++;
                }
            }
            else
            {
                // tail: i = sets-6 .. sets-1, only multiples of skip = 2, 4, .. 64 are updated
                getpair(i, s0, s1);
                const int kk = VT_ORDER - (sets - i);
                unsigned long long w = 0;
                switch (kk)
                {
                case 1: w = vl_step<1>(R, T, s0, s1); break;
                case 2: w = vl_step<2>(R, T, s0, s1); break;
                case 3: w = vl_step<3>(R, T, s0, s1); break;
                case 4: w = vl_step<4>(R, T, s0, s1); break;
                case 5: w = vl_step<5>(R, T, s0, s1); break;
                default: w = vl_step<6>(R, T, s0, s1); break;
                }
                copy();
                record(w);
                skip = 1u << kk;
                h.renorm++;
                h.len++;
                i++;
            }
            // events, one code site each (the traceback is large): renormalise every VT_RENORM steps, trace back when VT_CAP slices are
            // buffered, and after the last step flush from state 0 (history_buffer_flush)
            const bool last = i >= sets;
            const bool ren = h.renorm == VT_RENORM, full = h.len == VT_CAP;
            unsigned best = 0;
            if (ren || full)
            {
                unsigned sub;
                best = vl_search(R, skip, sub);
                if (ren) { h.renorm = 0; vl_renorm(R, skip, sub); }
            }
#pragma nounroll
            for (int pass = 0; pass < 2; pass++)
                if (pass == 0 ? full : last) traceback(pass == 0 ? best : 0u, pass == 0 ? VT_MINTB : 0);
            if (last) break;
        }
        if (packed && pk_pend_idx >= 0) pk_store(pk_pend_idx, pk_pend);
    }
}

// Two entry points over the same body.  The decoder needs ~240 registers, so two wavefronts fit on a SIMD -- and for a bank of
// <= 1024 wavefronts (one per SIMD of an MI355X) the dispatcher then stacks pairs on some SIMDs while others idle: 2.5 ms instead
// of 1.7 ms per launch.  amdgpu_waves_per_eu(1, 1) makes the register allocation large enough that only one fits (capping the CU at
// four workgroups through the LDS request does not help, the four still share SIMDs); banks that need more than one wavefront
// per SIMD use the (1, 2) entry.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_viterbi_lanes(
    const uint8_t *__restrict__ soft, int nsoft, const uint8_t *__restrict__ overlap, int pad, uint8_t *__restrict__ out, int out_stride, int out_start,
    int out_want, int nblocks, const int *__restrict__ valid, unsigned long long *__restrict__ hist, int tiled, int packed, int pitch,
    const int *__restrict__ lens)
{
    __shared__ uint4 lds_soft[8 * 64];
    vl_decode(soft, nsoft, overlap, pad, out, out_stride, out_start, out_want, nblocks, valid, hist, tiled, lds_soft, packed, pitch, lens);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_viterbi_lanes_x2(
    const uint8_t *__restrict__ soft, int nsoft, const uint8_t *__restrict__ overlap, int pad, uint8_t *__restrict__ out, int out_stride, int out_start,
    int out_want, int nblocks, const int *__restrict__ valid, unsigned long long *__restrict__ hist, int tiled, int packed, int pitch,
    const int *__restrict__ lens)
{
    __shared__ uint4 lds_soft[8 * 64];
    vl_decode(soft, nsoft, overlap, pad, out, out_stride, out_start, out_want, nblocks, valid, hist, tiled, lds_soft, packed, pitch, lens);
}
