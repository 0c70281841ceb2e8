// Arrow -> Liquid transcoder entry points (host).  See lc_transcode.cpp.
#pragma once

#include <functional>
#include <string_view>
#include <vector>

#include "lc_fsst.hpp"
#include "lc_host.hpp"

namespace lc {

using StringGetter = std::function<std::string_view(size_t)>;

// Per-ColumnAccessPath FSST state (reference: LiquidCompressorStates, src/core/src/cache/utils.rs:86-129).
struct SymtabProvider {
    virtual ~SymtabProvider() = default;
    virtual const SymbolTable* find(uint64_t path_id) = 0;
    // registers `st` for `path_id` unless one exists already; returns the registered table
    virtual const SymbolTable* insert(uint64_t path_id, const SymbolTable& st) = 0;
};

lc_status transcode_primitive(int phys, const void* values, const uint8_t* validity, size_t n,
                              std::vector<uint8_t>& out);
lc_status transcode_decimal(int width, int precision, int scale, const void* values, const uint8_t* validity,
                            size_t n, std::vector<uint8_t>& out);
lc_status transcode_byte_view(int arrow_type, const StringGetter& get, const uint8_t* validity, size_t n,
                              const SymbolTable& st, bool build_fingerprints, std::vector<uint8_t>& out);
lc_status transcode_arrow(const struct ArrowArray* array, const struct ArrowSchema* schema, int32_t hint,
                          SymtabProvider& symtabs, uint64_t path_id, std::vector<uint8_t>& out);

}  // namespace lc
