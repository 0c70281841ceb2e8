/* Integer / Decimal / Float(ALP) / date parts (TEST ORACLE — see lo_common.h). */
#ifndef LO_PRIMITIVE_H
#define LO_PRIMITIVE_H
#include "lo_common.h"
#ifdef __cplusplus
extern "C" {
#endif

enum lo_date_field { LO_DATE_YEAR = 0, LO_DATE_MONTH = 1, LO_DATE_DAY = 2, LO_DATE_DOW = 3 };

typedef struct {
    int32_t logical;
    int32_t phys;
    uint32_t len;
    int32_t nullable;
    int32_t all_null;
    int32_t bit_width;
    int32_t value_width; /* bytes per decoded Arrow value (decimal: 16) */
    int32_t lane_bits;   /* FastLanes lane type */
    uint64_t reference;
    uint64_t bitpacked_off;
    int32_t decimal_is256, decimal_precision, decimal_scale;
    int32_t alp_e, alp_f;
    uint64_t patch_len, patch_indices_off, patch_values_off;
} lo_array_info;

LO_EXPORT void lo_ipc_header_write(uint8_t* out, int logical, int phys);
LO_EXPORT int lo_ipc_header_read(const uint8_t* bytes, size_t len, int* logical, int* phys);
LO_EXPORT int lo_array_info_get(const uint8_t* bytes, size_t len, lo_array_info* info);

LO_EXPORT size_t lo_prim_encode_bound(int phys, size_t n);
LO_EXPORT int64_t lo_prim_encode(int phys, const void* values, const uint8_t* validity, size_t n, uint8_t* out,
                                 size_t cap);
LO_EXPORT size_t lo_float_encode_bound(int phys, size_t n);
LO_EXPORT int64_t lo_float_encode(int phys, const void* values, const uint8_t* validity, size_t n, uint8_t* out,
                                  size_t cap);
LO_EXPORT size_t lo_decimal_encode_bound(size_t n);
LO_EXPORT int64_t lo_decimal_encode(int is256, int precision, int scale, const void* values_i128,
                                    const uint8_t* validity, size_t n, uint8_t* out, size_t cap);

LO_EXPORT int lo_fixed_to_arrow(const uint8_t* bytes, size_t len, void* out_values, uint8_t* out_validity);
LO_EXPORT int64_t lo_fixed_filter(const uint8_t* bytes, size_t len, const uint8_t* sel, void* out_values,
                                  uint8_t* out_validity, int* nullable);
LO_EXPORT int64_t lo_fixed_eval_predicate(const uint8_t* bytes, size_t len, int op, int lit_tag, const void* lit,
                                          const uint8_t* sel, uint8_t* out_values, uint8_t* out_validity,
                                          int* nullable);

LO_EXPORT int32_t lo_date_component(int field, int32_t days);
LO_EXPORT int32_t lo_ymd_to_epoch_days(int32_t year, uint32_t month, uint32_t day);
LO_EXPORT int32_t lo_timestamp_to_days(int64_t value, int unit);
LO_EXPORT int32_t lo_date_lossy_days(int field, int32_t component);

#ifdef __cplusplus
}
#endif
#endif
