/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or called from the product path
 * (jaero_amd/, libjaero_hip.so).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED.  JAERO's Viterbi arithmetic is quiet/libcorrect, a third-party dependency that is absent
 * from /root/reference (cloned at HEAD without tag or commit pin: ci-linux-build.sh:118-131, linked as
 * -lcorrect: JAERO/JAERO.pro:188) and none of the reference's tests touch JConvolutionalCodec.  This file
 * restates libcorrect's published portable (non-SSE) convolutional codec from its algorithm:
 *   - lookup table      : table[sr] bit j = parity(sr & poly[j]), sr = order-bit shift register, newest bit in LSB
 *   - encoder           : per message bit (MSB first) shift in, emit table[sr] LSB first, flush order+1 zero bits
 *   - soft metric       : linear, sum |soft - (bit ? 255 : 0)| into uint16 path metrics (wrap mod 2^16)
 *   - decoder           : warm-up for the first order-1 sets, add-compare-select ("low <= high" keeps low) for the
 *                         middle, zero-input constrained tail for the last order-1 sets ("low < high" keeps low),
 *                         history ring of 5*order + 15*order slices, renormalise every 65535/(rate*255) sets,
 *                         best state = first minimum, traceback emits the bit shifted out of the oldest position,
 *                         final flush traces everything back from state 0; output packed MSB first.
 * It is anchored on the reference's call sites: correct_convolutional_create(2,7,{109,79})
 * (JAERO/jconvolutionalcodec.cpp:12-16,23; JAERO/aerol.cpp:936-940) and correct_convolutional_decode_soft
 * (JAERO/jconvolutionalcodec.cpp:98,169).  What pins it in tests: encode -> channel -> decode round trips and
 * bit-error-free decoding at usable SNR (tests/test_viterbi_*.py).
 *
 * The second half restates JConvolutionalCodec::Decode_Continuous / Decode_soft
 * (JAERO/jconvolutionalcodec.cpp:151-201, 90-119) on top of it.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include "ref/shim/correct.h"
#include "viterbi_oracle.h"

typedef uint16_t distance_t;
typedef unsigned int shift_register_t;

struct correct_convolutional
{
    unsigned int *table;
    size_t rate;
    size_t order;
    unsigned int numstates; /* 1 << order, as libcorrect */
};

static unsigned int parity_u(unsigned int x)
{
    unsigned int p = 0;
    while (x) { p ^= (x & 1u); x >>= 1; }
    return p;
}

correct_convolutional *correct_convolutional_create(size_t rate, size_t order, const correct_convolutional_polynomial_t *poly)
{
    if (order > 8 * sizeof(shift_register_t) || rate < 2) return NULL;
    correct_convolutional *conv = (correct_convolutional *)calloc(1, sizeof(*conv));
    conv->rate = rate;
    conv->order = order;
    conv->numstates = 1u << order;
    conv->table = (unsigned int *)malloc(sizeof(unsigned int) * (1u << order));
    for (shift_register_t i = 0; i < (1u << order); i++)
    {
        unsigned int out = 0, mask = 1;
        for (size_t j = 0; j < rate; j++)
        {
            if (parity_u(i & poly[j])) out |= mask;
            mask <<= 1;
        }
        conv->table[i] = out;
    }
    return conv;
}

void correct_convolutional_destroy(correct_convolutional *conv)
{
    if (!conv) return;
    free(conv->table);
    free(conv);
}

/* ---- MSB-first bit writer ---- */
typedef struct { uint8_t *bytes; size_t len; uint8_t byte; unsigned int byte_len; } bit_writer;

static void bw_init(bit_writer *w, uint8_t *bytes) { w->bytes = bytes; w->len = 0; w->byte = 0; w->byte_len = 0; }
static void bw_write1(bit_writer *w, unsigned int bit)
{
    w->byte = (uint8_t)((w->byte << 1) | (bit & 1u));
    w->byte_len++;
    if (w->byte_len == 8) { w->bytes[w->len++] = w->byte; w->byte = 0; w->byte_len = 0; }
}
static void bw_flush(bit_writer *w)
{
    if (w->byte_len) { w->bytes[w->len++] = (uint8_t)(w->byte << (8 - w->byte_len)); w->byte = 0; w->byte_len = 0; }
}

size_t correct_convolutional_encode_len(correct_convolutional *conv, size_t msg_len)
{
    return conv->rate * (8 * msg_len + conv->order + 1);
}

size_t correct_convolutional_encode(correct_convolutional *conv, const uint8_t *msg, size_t msg_len, uint8_t *encoded)
{
    shift_register_t sr = 0;
    unsigned int shiftmask = (1u << conv->order) - 1;
    bit_writer w;
    bw_init(&w, encoded);
    for (size_t i = 0; i < 8 * msg_len; i++)
    {
        unsigned int bit = (msg[i >> 3] >> (7 - (i & 7))) & 1u;
        sr = ((sr << 1) | bit) & shiftmask;
        unsigned int out = conv->table[sr];
        for (size_t j = 0; j < conv->rate; j++) { bw_write1(&w, out & 1u); out >>= 1; }
    }
    for (size_t i = 0; i < conv->order + 1; i++)
    {
        sr = (sr << 1) & shiftmask;
        unsigned int out = conv->table[sr];
        for (size_t j = 0; j < conv->rate; j++) { bw_write1(&w, out & 1u); out >>= 1; }
    }
    bw_flush(&w);
    return correct_convolutional_encode_len(conv, msg_len);
}

/* ---- decoder ---- */
static distance_t soft_distance_linear(unsigned int hard_x, const uint8_t *soft_y, size_t len)
{
    distance_t dist = 0;
    for (size_t i = 0; i < len; i++)
    {
        unsigned int soft_x = (hard_x & 1u) ? 255u : 0u;
        hard_x >>= 1;
        int d = (int)soft_y[i] - (int)soft_x;
        dist = (distance_t)(dist + ((d < 0) ? -d : d));
    }
    return dist;
}
static distance_t hard_distance(unsigned int x, unsigned int y)
{
    unsigned int v = x ^ y, c = 0;
    while (v) { c += v & 1u; v >>= 1; }
    return (distance_t)c;
}

typedef struct
{
    unsigned int min_traceback_length, traceback_group_length, cap;
    unsigned int num_states;
    shift_register_t highbit;
    uint8_t **history;
    uint8_t *fetched;
    unsigned int index, len;
    unsigned int renormalize_interval, renormalize_counter;
} history_buffer;

static shift_register_t hb_search(const history_buffer *buf, const distance_t *distances, unsigned int search_every)
{
    shift_register_t bestpath = 0;
    distance_t leasterror = 65535;
    for (shift_register_t state = 0; state < buf->num_states; state += search_every)
    {
        if (distances[state] < leasterror) { leasterror = distances[state]; bestpath = state; }
    }
    return bestpath;
}

static void hb_traceback(history_buffer *buf, shift_register_t bestpath, unsigned int min_traceback_length, bit_writer *out)
{
    unsigned int fetched_index = 0;
    shift_register_t highbit = buf->highbit;
    unsigned int index = buf->index;
    unsigned int cap = buf->cap;
    for (unsigned int j = 0; j < min_traceback_length; j++)
    {
        if (index == 0) index = cap - 1; else index--;
        uint8_t h = buf->history[index][bestpath];
        shift_register_t pathbit = h ? highbit : 0;
        bestpath |= pathbit;
        bestpath >>= 1;
    }
    unsigned int len = buf->len;
    for (unsigned int j = min_traceback_length; j < len; j++)
    {
        if (index == 0) index = cap - 1; else index--;
        uint8_t h = buf->history[index][bestpath];
        shift_register_t pathbit = h ? highbit : 0;
        bestpath |= pathbit;
        bestpath >>= 1;
        buf->fetched[fetched_index++] = pathbit ? 1 : 0;
    }
    for (unsigned int k = fetched_index; k > 0; k--) bw_write1(out, buf->fetched[k - 1]); /* reversed */
    buf->len -= fetched_index;
}

static void hb_process_skip(history_buffer *buf, distance_t *distances, bit_writer *out, unsigned int skip)
{
    buf->index++;
    if (buf->index == buf->cap) buf->index = 0;
    buf->renormalize_counter++;
    buf->len++;
    if (buf->renormalize_counter == buf->renormalize_interval)
    {
        buf->renormalize_counter = 0;
        shift_register_t bestpath = hb_search(buf, distances, skip);
        distance_t min_distance = distances[bestpath];
        for (shift_register_t s = 0; s < buf->num_states; s += skip) distances[s] = (distance_t)(distances[s] - min_distance);
        if (buf->len == buf->cap) hb_traceback(buf, bestpath, buf->min_traceback_length, out);
    }
    else if (buf->len == buf->cap)
    {
        shift_register_t bestpath = hb_search(buf, distances, skip);
        hb_traceback(buf, bestpath, buf->min_traceback_length, out);
    }
}

static ssize_t conv_decode(correct_convolutional *conv, size_t num_encoded_bits, uint8_t *msg,
                           const uint8_t *soft, const uint8_t *hard_bytes)
{
    const size_t rate = conv->rate, order = conv->order;
    if (num_encoded_bits % rate) return -1;
    size_t sets = num_encoded_bits / rate;
    const unsigned int nstates = conv->numstates / 2; /* 1 << (order-1) */
    const shift_register_t highbit = 1u << (order - 1);

    history_buffer hb;
    hb.min_traceback_length = 5 * order;
    hb.traceback_group_length = 15 * order;
    hb.cap = hb.min_traceback_length + hb.traceback_group_length;
    hb.num_states = nstates;
    hb.highbit = highbit;
    hb.history = (uint8_t **)malloc(sizeof(uint8_t *) * hb.cap);
    for (unsigned int i = 0; i < hb.cap; i++) hb.history[i] = (uint8_t *)calloc(nstates, 1);
    hb.fetched = (uint8_t *)malloc(hb.cap);
    hb.index = 0; hb.len = 0;
    hb.renormalize_interval = 65535u / (unsigned int)(rate * 255u);
    hb.renormalize_counter = 0;

    distance_t *errA = (distance_t *)calloc(conv->numstates, sizeof(distance_t));
    distance_t *errB = (distance_t *)calloc(conv->numstates, sizeof(distance_t));
    distance_t *read_errors = errA, *write_errors = errB;
    distance_t *distances = (distance_t *)calloc(1u << rate, sizeof(distance_t));

    bit_writer w;
    bw_init(&w, msg);

    /* hard input: MSB-first bit reader, first-read bit becomes bit 0 of the symbol */
    size_t hard_pos = 0;
#define READ_HARD_SYMBOL(var)                                                                         \
    do { unsigned int _o = 0;                                                                         \
         for (size_t _j = 0; _j < rate; _j++) {                                                       \
             unsigned int _b = (hard_bytes[hard_pos >> 3] >> (7 - (hard_pos & 7))) & 1u; hard_pos++;  \
             _o |= _b << _j; }                                                                        \
         var = _o; } while (0)

    /* warm-up: states are still being filled, no decisions recorded */
    for (size_t i = 0; i < order - 1 && i < sets; i++)
    {
        unsigned int out = 0;
        if (!soft) READ_HARD_SYMBOL(out);
        for (size_t j = 0; j < ((size_t)1 << (i + 1)); j++)
        {
            unsigned int last = (unsigned int)(j >> 1);
            distance_t dist = soft ? soft_distance_linear(conv->table[j], soft + i * rate, rate)
                                   : hard_distance(conv->table[j], out);
            write_errors[j] = (distance_t)(dist + read_errors[last]);
        }
        distance_t *t = read_errors; read_errors = write_errors; write_errors = t;
    }

    /* inner: full add-compare-select */
    if (sets + 1 > order)
    for (size_t i = order - 1; i < sets - order + 1; i++)
    {
        if (soft) { for (unsigned int j = 0; j < (1u << rate); j++) distances[j] = soft_distance_linear(j, soft + i * rate, rate); }
        else { unsigned int out; READ_HARD_SYMBOL(out); for (unsigned int j = 0; j < (1u << rate); j++) distances[j] = hard_distance(j, out); }
        uint8_t *history = hb.history[hb.index];
        for (shift_register_t succ = 0; succ < nstates; succ++)
        {
            /* predecessors: succ>>1 (oldest bit 0) and (succ>>1)|highbit/2 (oldest bit 1) */
            shift_register_t plow = succ >> 1, phigh = (succ >> 1) | (highbit >> 1);
            distance_t low_error = (distance_t)(distances[conv->table[succ]] + read_errors[plow]);
            distance_t high_error = (distance_t)(distances[conv->table[succ | highbit]] + read_errors[phigh]);
            if (low_error <= high_error) { write_errors[succ] = low_error; history[succ] = 0; }
            else { write_errors[succ] = high_error; history[succ] = 1; }
        }
        hb_process_skip(&hb, write_errors, &w, 1);
        distance_t *t = read_errors; read_errors = write_errors; write_errors = t;
    }

    /* tail: only zero input bits allowed, successors are multiples of skip */
    if (sets + 1 > order)
    for (size_t i = sets - order + 1; i < sets; i++)
    {
        if (soft) { for (unsigned int j = 0; j < (1u << rate); j++) distances[j] = soft_distance_linear(j, soft + i * rate, rate); }
        else { unsigned int out; READ_HARD_SYMBOL(out); for (unsigned int j = 0; j < (1u << rate); j++) distances[j] = hard_distance(j, out); }
        uint8_t *history = hb.history[hb.index];
        unsigned int skip = 1u << (order - (sets - i));
        for (shift_register_t low = 0; low < highbit; low += skip)
        {
            shift_register_t base = low >> 1;
            distance_t low_error = (distance_t)(distances[conv->table[low]] + read_errors[base]);
            distance_t high_error = (distance_t)(distances[conv->table[low | highbit]] + read_errors[(highbit >> 1) + base]);
            if (low_error < high_error) { write_errors[low] = low_error; history[low] = 0; }
            else { write_errors[low] = high_error; history[low] = 1; }
        }
        hb_process_skip(&hb, write_errors, &w, skip);
        distance_t *t = read_errors; read_errors = write_errors; write_errors = t;
    }
#undef READ_HARD_SYMBOL

    hb_traceback(&hb, 0, 0, &w); /* history_buffer_flush */
    /* libcorrect returns the byte count of whole bytes written; trailing (sets-(order-1))%8 bits stay in the
       writer.  We flush them so callers can see every decoded bit; the return value keeps libcorrect's meaning. */
    size_t whole = w.len;
    bw_flush(&w);

    for (unsigned int i = 0; i < hb.cap; i++) free(hb.history[i]);
    free(hb.history); free(hb.fetched); free(errA); free(errB); free(distances);
    return (ssize_t)whole;
}

ssize_t correct_convolutional_decode_soft(correct_convolutional *conv, const correct_convolutional_soft_t *encoded,
                                          size_t num_encoded_bits, uint8_t *msg)
{
    return conv_decode(conv, num_encoded_bits, msg, encoded, NULL);
}

ssize_t correct_convolutional_decode(correct_convolutional *conv, const uint8_t *encoded, size_t num_encoded_bits, uint8_t *msg)
{
    return conv_decode(conv, num_encoded_bits, msg, NULL, encoded);
}

/* ------------------------------------------------------------------------------------------------------------
 * JConvolutionalCodec restatement (JAERO/jconvolutionalcodec.cpp)
 * ---------------------------------------------------------------------------------------------------------- */
struct jo_codec
{
    correct_convolutional *convol;
    int constraint, nparitybits, paddinglength;
    uint8_t overlap[64];
    int overlap_len;
};

jo_codec *jo_codec_create(int paddinglength)
{
    /* JConvolutionalCodec ctor + SetCode(2,7,{109,79},padding): jconvolutionalcodec.cpp:8-29, aerol.cpp:936-940 */
    jo_codec *c = (jo_codec *)calloc(1, sizeof(*c));
    correct_convolutional_polynomial_t poly[2] = {109, 79};
    c->convol = correct_convolutional_create(2, 7, poly);
    c->constraint = 7;
    c->nparitybits = 2;
    c->paddinglength = paddinglength;
    c->overlap_len = 0;
    return c;
}

void jo_codec_destroy(jo_codec *c)
{
    if (!c) return;
    correct_convolutional_destroy(c->convol);
    free(c);
}

void jo_codec_reset(jo_codec *c) { c->overlap_len = 0; }

/* jconvolutionalcodec.cpp:151-201.  bits_out: one byte per bit, capacity >= n/2.  Returns number of bits.
 * Bits the reference would read from never-written bytes of its scratch buffer are defined as 0 here. */
int jo_decode_continuous(jo_codec *c, const uint8_t *soft_in, int n, uint8_t *bits_out)
{
    const int k = 62; /* :153 */
    int total = c->overlap_len + n + c->paddinglength;
    uint8_t *buf = (uint8_t *)malloc((size_t)total);
    memcpy(buf, c->overlap, (size_t)c->overlap_len);               /* :155 append */
    memcpy(buf + c->overlap_len, soft_in, (size_t)n);
    memset(buf + c->overlap_len + n, 128, (size_t)c->paddinglength); /* :158-160 */
    int decoded_sz = total / c->nparitybits + 1;                   /* :167 */
    uint8_t *decoded = (uint8_t *)calloc((size_t)decoded_sz + 8, 1);
    correct_convolutional_decode_soft(c->convol, buf, (size_t)total, decoded); /* :169 */
    int dbits = total / c->nparitybits;                            /* :172 */
    /* :175-190 unpack MSB first, then :194 mid(paddinglength+1, n/nparitybits) */
    int start = c->paddinglength + 1;
    int want = n / c->nparitybits;
    int nout = 0;
    for (int b = start; b < dbits && nout < want; b++)
    {
        int byte = b >> 3;
        uint8_t v = (byte < decoded_sz) ? decoded[byte] : 0;
        bits_out[nout++] = (v >> (7 - (b & 7))) & 1u;
    }
    /* :197-198 keep the last k soft bytes of THIS input (zero padded on the right if n<k, as resize does) */
    memset(c->overlap, 0, sizeof(c->overlap));
    if (n >= k) memcpy(c->overlap, soft_in + n - k, (size_t)k);
    else memcpy(c->overlap, soft_in, (size_t)n);
    c->overlap_len = k;
    free(buf); free(decoded);
    return nout;
}

/* jconvolutionalcodec.cpp:90-119.  bits_out capacity >= size/2; returns number of bits (size/2). */
int jo_decode_soft(jo_codec *c, const uint8_t *soft_in, int size, uint8_t *bits_out)
{
    c->overlap_len = 0;                                            /* :93 clear */
    int decoded_sz = size / c->nparitybits;                        /* :97 */
    uint8_t *decoded = (uint8_t *)calloc((size_t)decoded_sz + 8, 1);
    correct_convolutional_decode_soft(c->convol, soft_in, (size_t)size, decoded);
    int dbits = size / c->nparitybits;
    for (int b = 0; b < dbits; b++) bits_out[b] = (decoded[b >> 3] >> (7 - (b & 7))) & 1u;
    free(decoded);
    return dbits;
}

/* convenience for tests: encode message bytes with the Aero code, one output byte per coded bit (0/1) */
int jo_encode_bits(const uint8_t *msg, int msg_len, uint8_t *coded_bits_out)
{
    correct_convolutional_polynomial_t poly[2] = {109, 79};
    correct_convolutional *cv = correct_convolutional_create(2, 7, poly);
    size_t nb = correct_convolutional_encode_len(cv, (size_t)msg_len);
    uint8_t *packed = (uint8_t *)calloc(nb / 8 + 2, 1);
    correct_convolutional_encode(cv, msg, (size_t)msg_len, packed);
    for (size_t i = 0; i < nb; i++) coded_bits_out[i] = (packed[i >> 3] >> (7 - (i & 7))) & 1u;
    free(packed);
    correct_convolutional_destroy(cv);
    return (int)nb;
}
