/* FSST symbol tables, decode, and a restated trainer/compressor (TEST ORACLE — see lo_common.h). */
#ifndef LO_FSST_H
#define LO_FSST_H
#include "lo_common.h"
#ifdef __cplusplus
extern "C" {
#endif

#define LO_FSST_ESC 255

typedef struct {
    int32_t n;          /* number of symbols, <= 255 */
    uint8_t len[256];   /* symbol lengths 1..8 */
    uint64_t sym[256];  /* symbol bytes, little endian: first byte in the lowest-order byte */
} lo_symtab;

LO_EXPORT int lo_symtab_load(const uint8_t* bytes, size_t len, lo_symtab* st);
LO_EXPORT size_t lo_symtab_save(const lo_symtab* st, uint8_t* out);
LO_EXPORT size_t lo_fsst_decompress(const lo_symtab* st, const uint8_t* in, size_t in_len, uint8_t* out, size_t cap);
LO_EXPORT size_t lo_fsst_decompressed_len(const lo_symtab* st, const uint8_t* in, size_t in_len);
LO_EXPORT size_t lo_fsst_compress(const lo_symtab* st, const uint8_t* in, size_t len, uint8_t* out);
LO_EXPORT void lo_fsst_train(const uint8_t* data, const int32_t* offsets, size_t n, lo_symtab* st);

#ifdef __cplusplus
}
#endif
#endif
