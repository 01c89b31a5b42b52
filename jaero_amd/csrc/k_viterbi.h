// k_viterbi.h -- batched K=7 r=1/2 {109,79} soft-decision Viterbi, one code block per wavefront.
//
// Reproduces correct_convolutional_decode_soft of quiet/libcorrect as JConvolutionalCodec drives it
// (JAERO/jconvolutionalcodec.cpp:98,169): 64 trellis states = 64 lanes, path metrics in one VGPR per lane
// (uint16 arithmetic, wrapping), predecessor metrics fetched with two __shfl per step (add-compare-select butterfly),
// the 64 decision bits of a step are one __ballot word kept in an LDS history ring of 35+105 slices, traceback and
// renormalisation follow libcorrect's schedule (traceback every 105 steps once 140 slices are buffered, renormalise
// every 128 steps, best state = first minimum, final flush from state 0).
// PARITY UNPINNED: libcorrect is not in /root/reference (see oracle/viterbi_oracle.c header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VT_ORDER 7
#define VT_MINTB (5 * VT_ORDER)
#define VT_GROUP (15 * VT_ORDER)
#define VT_CAP (VT_MINTB + VT_GROUP)
#define VT_RENORM (65535 / (2 * 255))

__device__ __forceinline__ unsigned vt_parity(unsigned x) { return __popc(x) & 1u; }
__device__ __forceinline__ unsigned vt_table(unsigned sr) { return vt_parity(sr & 109u) | (vt_parity(sr & 79u) << 1); }

struct VtHist
{
    unsigned long long hist[VT_CAP];
    unsigned char fetched[VT_CAP];
};

// traceback (history_buffer_traceback); all lanes execute redundantly with wave-uniform values
__device__ __forceinline__ int vt_traceback(VtHist &h, unsigned bestpath, int min_tb, int index, int len)
{
    int fetched = 0;
    for (int j = 0; j < min_tb; j++)
    {
        index = (index == 0) ? VT_CAP - 1 : index - 1;
        const unsigned hb = (unsigned)((h.hist[index] >> bestpath) & 1ull);
        bestpath = (bestpath | (hb << 6)) >> 1;
    }
    for (int j = min_tb; j < len; j++)
    {
        index = (index == 0) ? VT_CAP - 1 : index - 1;
        const unsigned hb = (unsigned)((h.hist[index] >> bestpath) & 1ull);
        bestpath = (bestpath | (hb << 6)) >> 1;
        if (threadIdx.x == 0) h.fetched[fetched] = (unsigned char)hb;
        fetched++;
    }
    return fetched;
}

// first-minimum search over lanes whose state is a multiple of `skip`
__device__ __forceinline__ unsigned vt_search(unsigned pm, unsigned lane, unsigned skip, unsigned &minval)
{
    unsigned v = ((lane & (skip - 1)) == 0) ? pm : 0xFFFFFFFFu;
    unsigned m = v;
    for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o));
    minval = m;
    // libcorrect: leasterror starts at USHRT_MAX and uses '<' -> a state is only taken if strictly below 65535
    const unsigned long long eq = __ballot(v == m && m < 65535u);
    return eq ? (unsigned)(__ffsll((long long)eq) - 1) : 0u;
}

// One wavefront per block/stream.  Input bytes of stream b are the concatenation
//   overlap[b][0..ovl)  ++  soft[b*nsoft .. +nsoft)  ++  pad x 128
// Decoded bit k (k < sets-6) is written to out[b*out_stride + (k - out_start)] when 0 <= k-out_start < out_want.
__global__ __launch_bounds__(64) void k_viterbi(const uint8_t *__restrict__ soft, int nsoft, const uint8_t *__restrict__ overlap,
                                                int pad, uint8_t *__restrict__ out, int out_stride, int out_start, int out_want,
                                                int nblocks, const int *__restrict__ valid = nullptr, const int *__restrict__ lens = nullptr, int pitch = 0)
{
    __shared__ VtHist h;
    const unsigned lane = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= nblocks) return;
    if (valid && !valid[b]) return; // Aero-L pipeline: only the channels that completed a block this round
    const uint8_t *in = soft + (size_t)b * (pitch > 0 ? pitch : nsoft); // rows are pitch (default nsoft) apart; lens (optional): this block's own length <= nsoft
    if (lens) { nsoft = lens[b]; if (out_want > nsoft / 2) out_want = nsoft / 2; }
    const uint8_t *ov = overlap ? overlap + (size_t)b * 64 : nullptr;
    const int ovl = ov ? (int)ov[62] : 0;
    const int total = ovl + nsoft + pad;
    const int sets = total / 2;
    uint8_t *o = out + (size_t)b * out_stride;

    auto getsoft = [&](int i) -> unsigned {
        if (i < ovl) return ov[i];
        i -= ovl;
        if (i < nsoft) return in[i];
        return 128u;
    };

    const unsigned out_low = vt_table(lane), out_high = vt_table(lane | 64u);
    const int src_low = (int)(lane >> 1), src_high = (int)((lane >> 1) | 32u);
    unsigned pm = 0;
    int index = 0, len = 0, renorm = 0, outpos = 0;

    auto emit = [&](int fetched) {
        // bit_writer_write_bitlist_reversed: fetched[] holds newest-first
        __syncthreads();
        for (int q = (int)lane; q < fetched; q += 64)
        {
            const int k = outpos + q - out_start;
            if (k >= 0 && k < out_want) o[k] = h.fetched[fetched - 1 - q];
        }
        __syncthreads();
        outpos += fetched;
    };

    unsigned chunk = 0;
    for (int i = 0; i < sets; i++)
    {
        // 64 soft bytes (32 steps) are fetched at once, one per lane, then broadcast with __shfl
        if ((i & 31) == 0) chunk = getsoft(2 * i + (int)lane);
        const unsigned s0 = (unsigned)__shfl((int)chunk, 2 * (i & 31)), s1 = (unsigned)__shfl((int)chunk, 2 * (i & 31) + 1);
        // metric_soft_distance_linear for the four 2-bit outputs
        const unsigned d0a = s0, d0b = 255u - s0; // |s0-0|, |s0-255|
        const unsigned d1a = s1, d1b = 255u - s1;
        auto dist = [&](unsigned outbits) -> unsigned { return ((outbits & 1u) ? d0b : d0a) + ((outbits & 2u) ? d1b : d1a); };
        const unsigned pl = (unsigned)__shfl((int)pm, src_low);
        const unsigned ph = (unsigned)__shfl((int)pm, src_high);
        if (i < VT_ORDER - 1)
        {
            // warm-up: write_errors[j] = dist(table[j]) + read_errors[j>>1], no history
            pm = (dist(out_low) + pl) & 0xFFFFu;
            continue;
        }
        unsigned skip = 1;
        unsigned hbit;
        const unsigned le = (dist(out_low) + pl) & 0xFFFFu;
        const unsigned he = (dist(out_high) + ph) & 0xFFFFu;
        if (i < sets - VT_ORDER + 1)
        {
            hbit = (le <= he) ? 0u : 1u;
            pm = hbit ? he : le;
        }
        else
        {
            skip = 1u << (VT_ORDER - (sets - i));
            const bool active = (lane & (skip - 1)) == 0;
            hbit = (le < he) ? 0u : 1u;
            if (active) pm = hbit ? he : le;
            else hbit = 0;
        }
        const unsigned long long word = __ballot(hbit != 0);
        if (lane == 0) h.hist[index] = word;
        __syncthreads();
        // history_buffer_process_skip
        index++;
        if (index == VT_CAP) index = 0;
        renorm++;
        len++;
        if (renorm == VT_RENORM)
        {
            renorm = 0;
            unsigned minval;
            const unsigned best = vt_search(pm, lane, skip, minval);
            // distances[bestpath] is subtracted from every searched state (bestpath=0 if none was < 65535)
            const unsigned sub = (unsigned)__shfl((int)pm, (int)best);
            if ((lane & (skip - 1)) == 0) pm = (pm - sub) & 0xFFFFu;
            if (len == VT_CAP)
            {
                const int f = vt_traceback(h, best, VT_MINTB, index, len);
                len -= f;
                emit(f);
            }
        }
        else if (len == VT_CAP)
        {
            unsigned minval;
            const unsigned best = vt_search(pm, lane, skip, minval);
            const int f = vt_traceback(h, best, VT_MINTB, index, len);
            len -= f;
            emit(f);
        }
    }
    // history_buffer_flush
    {
        const int f = vt_traceback(h, 0u, 0, index, len);
        emit(f);
    }
}
