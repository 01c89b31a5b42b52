/* ORACLE -- TEST INFRASTRUCTURE ONLY (see viterbi_oracle.c). */
#ifndef VITERBI_ORACLE_H
#define VITERBI_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct jo_codec jo_codec;
jo_codec *jo_codec_create(int paddinglength);
void jo_codec_destroy(jo_codec *c);
void jo_codec_reset(jo_codec *c);
int jo_decode_continuous(jo_codec *c, const uint8_t *soft_in, int n, uint8_t *bits_out);
int jo_decode_soft(jo_codec *c, const uint8_t *soft_in, int size, uint8_t *bits_out);
int jo_encode_bits(const uint8_t *msg, int msg_len, uint8_t *coded_bits_out);
#ifdef __cplusplus
}
#endif
#endif
