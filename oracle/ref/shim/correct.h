/* TEST INFRASTRUCTURE ONLY (oracle/_ref build).
 * Prototype shim for quiet/libcorrect's public header, which JAERO includes through
 * JAERO/jconvolutionalcodec.h:5-11 but does not vendor (cloned at HEAD, unpinned: ci-linux-build.sh:118-131).
 * Only the entry points JAERO calls are declared (JAERO/jconvolutionalcodec.cpp:12-16,23,68,98,169,235).
 * They are implemented by oracle/viterbi_oracle.c (a restatement of libcorrect's published algorithm). */
#ifndef ORACLE_SHIM_CORRECT_H
#define ORACLE_SHIM_CORRECT_H
#include <stdint.h>
#include <stddef.h>
#include <sys/types.h>
#ifdef __cplusplus
extern "C" {
#endif
struct correct_convolutional;
typedef struct correct_convolutional correct_convolutional;
typedef uint16_t correct_convolutional_polynomial_t;
typedef uint8_t correct_convolutional_soft_t;
correct_convolutional *correct_convolutional_create(size_t inv_rate, size_t order, const correct_convolutional_polynomial_t *poly);
void correct_convolutional_destroy(correct_convolutional *conv);
size_t correct_convolutional_encode_len(correct_convolutional *conv, size_t msg_len);
size_t correct_convolutional_encode(correct_convolutional *conv, const uint8_t *msg, size_t msg_len, uint8_t *encoded);
ssize_t correct_convolutional_decode(correct_convolutional *conv, const uint8_t *encoded, size_t num_encoded_bits, uint8_t *msg);
ssize_t correct_convolutional_decode_soft(correct_convolutional *conv, const correct_convolutional_soft_t *encoded, size_t num_encoded_bits, uint8_t *msg);
#ifdef __cplusplus
}
#endif
#endif
