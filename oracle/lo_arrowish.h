/* Arrow-semantics helpers (TEST ORACLE — see lo_common.h). */
#ifndef LO_ARROWISH_H
#define LO_ARROWISH_H
#include "lo_common.h"
#ifdef __cplusplus
extern "C" {
#endif

LO_EXPORT size_t lo_filter_values(int width, const void* values, size_t n, const uint8_t* sel, void* out);
LO_EXPORT size_t lo_filter_bits(const uint8_t* bits, size_t n, const uint8_t* sel, uint8_t* out);
LO_EXPORT int lo_cmp_scalar(int kind, int width, const void* values, size_t k, int op, const void* literal,
                            uint8_t* out_bits);
LO_EXPORT void lo_prep_null_mask(const uint8_t* values, const uint8_t* validity, size_t nbits, uint8_t* out);
LO_EXPORT size_t lo_and_then(const uint8_t* left, size_t left_bits, const uint8_t* right, size_t right_bits,
                             uint8_t* out);
LO_EXPORT int lo_like_match(const uint8_t* s, size_t sl, const uint8_t* pattern, size_t pl);
LO_EXPORT int lo_contains(const uint8_t* s, size_t sl, const uint8_t* needle, size_t nl);
LO_EXPORT int lo_bytes_cmp(const uint8_t* a, size_t al, const uint8_t* b, size_t bl);

#ifdef __cplusplus
}
#endif
#endif
