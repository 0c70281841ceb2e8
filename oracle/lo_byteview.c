/*
 * LiquidByteViewArray<FsstArray>: dictionary keys + prefix keys + FSST values + fingerprints — TEST ORACLE
 * (see lo_common.h).
 *
 * Follows (all under src/core/src/liquid_array/):
 *   build            byte_view_array/conversions.rs:260-373; utils/mod.rs:147-161 (dictionary builder)
 *   PrefixKey        raw/fsst_buffer.rs:160-212
 *   CompactOffsets   raw/fsst_buffer.rs:261-383, :762-846
 *   RawFsstBuffer    raw/fsst_buffer.rs:47-157
 *   fingerprints     byte_view_array/fingerprint.rs:5-74
 *   bytes            byte_view_array/serialization.rs:14-326
 *   predicates       byte_view_array/comparisons.rs:21-183, :325-501, :598-651; helpers.rs:44-92
 *   filter / decode  byte_view_array/mod.rs:215-290, :357-362, :421-424; helpers.rs:14-64
 */
#include "lo_byteview.h"
#include "lo_arrowish.h"
#include "lo_primitive.h"
#include <math.h>

/* fingerprint.rs:21-28 */
uint32_t lo_fingerprint(const uint8_t* s, size_t l) {
    uint32_t bits = 0;
    for (size_t i = 0; i < l; i++) bits |= 1u << (s[i] & 31);
    return bits;
}

/* fingerprint.rs:59-74: `%inner%` with a non-empty inner holding no `%` / `_` */
int lo_substring_pattern(const uint8_t* p, size_t pl, const uint8_t** inner, size_t* il) {
    if (pl < 2) return 0;
    if (p[0] != '%' || p[pl - 1] != '%') return 0;
    if (pl - 2 == 0) return 0;
    for (size_t i = 1; i + 1 < pl; i++)
        if (p[i] == '%' || p[i] == '_') return 0;
    *inner = p + 1;
    *il = pl - 2;
    return 1;
}

/* fsst_buffer.rs:267-296 (least squares, rounded) */
static void fit_line(const uint32_t* offsets, size_t n, int32_t* slope, int32_t* intercept) {
    if (n <= 1) { *slope = 0; *intercept = n ? (int32_t)offsets[0] : 0; return; }
    double nf = (double)n;
    double sum_x = (double)(n * (n - 1) / 2);
    double sum_y = 0, sum_xy = 0;
    for (size_t i = 0; i < n; i++) { sum_y += (double)offsets[i]; sum_xy += (double)i * (double)offsets[i]; }
    double sum_x_sq = (double)(n * (n - 1) * (2 * n - 1) / 6);
    double s = (nf * sum_xy - sum_x * sum_y) / (nf * sum_x_sq - sum_x * sum_x);
    double ic = (sum_y - s * sum_x) / nf;
    double rs = round(s), ri = round(ic);
    /* Rust `as i32` saturates */
    if (rs > 2147483647.0) rs = 2147483647.0;
    if (rs < -2147483648.0) rs = -2147483648.0;
    if (ri > 2147483647.0) ri = 2147483647.0;
    if (ri < -2147483648.0) ri = -2147483648.0;
    *slope = (rs != rs) ? 0 : (int32_t)rs;
    *intercept = (ri != ri) ? 0 : (int32_t)ri;
}

/* fsst_buffer.rs:298-359 + :762-784; returns bytes written */
static size_t compact_offsets_write(const uint32_t* offsets, size_t n, uint8_t* out) {
    if (n == 0) { /* from_offsets(&[]) :300-308 */
        memset(out, 0, 8);
        out[8] = 1;
        return 9;
    }
    int32_t slope, intercept;
    fit_line(offsets, n, &slope, &intercept);
    int32_t mn = INT32_MAX, mx = INT32_MIN;
    int32_t* res = (int32_t*)malloc(n * sizeof(int32_t));
    for (size_t i = 0; i < n; i++) {
        int32_t predicted = (int32_t)((uint32_t)slope * (uint32_t)i + (uint32_t)intercept);
        int32_t r = (int32_t)((uint32_t)offsets[i] - (uint32_t)predicted);
        res[i] = r;
        if (r < mn) mn = r;
        if (r > mx) mx = r;
    }
    int ob = (mn >= -128 && mx <= 127) ? 1 : (mn >= -32768 && mx <= 32767) ? 2 : 4;
    memcpy(out, &slope, 4);
    memcpy(out + 4, &intercept, 4);
    out[8] = (uint8_t)ob;
    uint8_t* p = out + 9;
    for (size_t i = 0; i < n; i++) {
        if (ob == 1) *p = (uint8_t)(int8_t)res[i];
        else if (ob == 2) { int16_t t = (int16_t)res[i]; memcpy(p, &t, 2); }
        else memcpy(p, &res[i], 4);
        p += ob;
    }
    free(res);
    return 9 + n * (size_t)ob;
}

/* fsst_buffer.rs:786-846 + :365-368 */
static int compact_offsets_read(const uint8_t* b, size_t len, uint32_t** out, uint32_t* count, int* offset_bytes) {
    if (len < 9) return LO_ERR_CORRUPT;
    int32_t slope, intercept;
    memcpy(&slope, b, 4);
    memcpy(&intercept, b + 4, 4);
    int ob = b[8];
    if (ob != 1 && ob != 2 && ob != 4) return LO_ERR_CORRUPT;
    size_t payload = len - 9;
    if (payload % (size_t)ob) return LO_ERR_CORRUPT;
    size_t n = payload / (size_t)ob;
    uint32_t* o = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) {
        int32_t r;
        if (ob == 1) r = (int8_t)b[9 + i];
        else if (ob == 2) { int16_t t; memcpy(&t, b + 9 + 2 * i, 2); r = t; }
        else memcpy(&r, b + 9 + 4 * i, 4);
        int32_t predicted = (int32_t)((uint32_t)slope * (uint32_t)i + (uint32_t)intercept);
        o[i] = (uint32_t)predicted + (uint32_t)r;
    }
    *out = o;
    *count = (uint32_t)n;
    *offset_bytes = ob;
    return LO_OK;
}

size_t lo_bv_encode_bound(size_t n, size_t data_len) {
    return 64 + 12 + 2 * data_len + 8 + 16 + lo_bm_bytes(n) + 8 + ((n + 1023) / 1024) * 2048 + 9 + 4 * (n + 2) + 8 +
           8 * (n + 1) + data_len + 8 + 4 * (n + 1) + 64;
}

/* conversions.rs:260-373 + serialization.rs:122-220 */
int64_t lo_bv_encode_dict(int arrow_type, const uint16_t* keys, const uint8_t* key_validity, size_t n,
                          const int32_t* doff, const uint8_t* ddata, size_t d, const lo_symtab* st,
                          int build_fingerprints, uint8_t* out, size_t cap) {
    size_t dict_bytes = d ? (size_t)(doff[d] - doff[0]) : 0;
    if (cap < lo_bv_encode_bound(n > d ? n : d, dict_bytes)) return LO_ERR_CAPACITY;
    /* shared prefix :269-307 */
    size_t sp_len = 0;
    const uint8_t* sp = NULL;
    if (d > 0) {
        sp = ddata + doff[0];
        sp_len = (size_t)(doff[1] - doff[0]);
        for (size_t i = 1; i < d && sp_len > 0; i++) {
            const uint8_t* v = ddata + doff[i];
            size_t vl = (size_t)(doff[i + 1] - doff[i]);
            size_t c = 0;
            while (c < sp_len && c < vl && sp[c] == v[c]) c++;
            sp_len = c;
        }
    }
    size_t pos = 16 + 20;
    memset(out, 0, 40);
    pos = lo_align8(pos); /* 40 */
    /* A) RawFsstBuffer :138-144 */
    size_t fsst_start = pos;
    uint32_t* offs = (uint32_t*)malloc((d + 1) * sizeof(uint32_t));
    uint8_t* cbuf = out + pos + 12;
    size_t clen = 0;
    offs[0] = 0;
    for (size_t i = 0; i < d; i++) {
        const uint8_t* v = ddata + doff[i];
        size_t vl = (size_t)(doff[i + 1] - doff[i]);
        clen += lo_fsst_compress(st, v, vl, cbuf + clen);
        offs[i + 1] = (uint32_t)clen;
    }
    lo_wr_u64(out + pos, (uint64_t)dict_bytes);
    lo_wr_u32(out + pos + 8, (uint32_t)clen);
    pos += 12 + clen;
    size_t fsst_raw_size = pos - fsst_start;
    while (pos & 7) out[pos++] = 0;
    /* C) keys as BitPackedArray<u16> at bit width 16 (serialization.rs:141-150) */
    size_t keys_start = pos;
    pos += lo_bitpacked_write(16, 16, keys, key_validity, n, out + pos);
    size_t keys_size = pos - keys_start;
    while (pos & 7) out[pos++] = 0;
    /* E) compact offsets */
    size_t off_start = pos;
    pos += compact_offsets_write(offs, d + 1, out + pos);
    size_t co_size = pos - off_start;
    while (pos & 7) out[pos++] = 0;
    /* G) prefix keys: PrefixKey::new(suffix after shared prefix) fsst_buffer.rs:175-187 */
    for (size_t i = 0; i < d; i++) {
        const uint8_t* v = ddata + doff[i];
        size_t vl = (size_t)(doff[i + 1] - doff[i]);
        const uint8_t* rem = sp_len < vl ? v + sp_len : v;
        size_t rl = sp_len < vl ? vl - sp_len : 0;
        memset(out + pos, 0, 8);
        memcpy(out + pos, rem, rl < 7 ? rl : 7);
        out[pos + 7] = rl >= 255 ? 255 : (uint8_t)rl;
        pos += 8;
    }
    while (pos & 7) out[pos++] = 0;
    /* I) shared prefix */
    if (sp_len) memcpy(out + pos, sp, sp_len);
    pos += sp_len;
    while (pos & 7) out[pos++] = 0;
    /* K) fingerprints over the FULL value (conversions.rs:353-355) */
    size_t fp_size = 0;
    if (build_fingerprints) {
        for (size_t i = 0; i < d; i++) {
            lo_wr_u32(out + pos, lo_fingerprint(ddata + doff[i], (size_t)(doff[i + 1] - doff[i])));
            pos += 4;
        }
        fp_size = d * 4;
    }
    lo_ipc_header_write(out, LO_LOGICAL_BYTE_VIEW, arrow_type);
    lo_wr_u32(out + 16, (uint32_t)keys_size);
    lo_wr_u32(out + 20, (uint32_t)co_size);
    lo_wr_u32(out + 24, (uint32_t)sp_len);
    lo_wr_u32(out + 28, (uint32_t)fsst_raw_size);
    lo_wr_u32(out + 32, (uint32_t)fp_size);
    free(offs);
    return (int64_t)pos;
}

/* GenericByteDictionaryBuilder::append_option (utils/mod.rs:147-161): first-occurrence keys, null -> null key */
int64_t lo_bv_encode(int arrow_type, const int32_t* offsets, const uint8_t* data, const uint8_t* validity, size_t n,
                     const lo_symtab* st, int build_fingerprints, uint8_t* out, size_t cap) {
    size_t hcap = 1;
    while (hcap < 2 * n + 16) hcap <<= 1;
    int32_t* table = (int32_t*)malloc(hcap * sizeof(int32_t));
    for (size_t i = 0; i < hcap; i++) table[i] = -1;
    uint16_t* keys = (uint16_t*)calloc(n ? n : 1, sizeof(uint16_t));
    int32_t* first_row = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    size_t d = 0;
    for (size_t i = 0; i < n; i++) {
        if (validity && !lo_get_bit(validity, i)) { keys[i] = 0; continue; }
        const uint8_t* v = data + offsets[i];
        size_t vl = (size_t)(offsets[i + 1] - offsets[i]);
        uint64_t h = 1469598103934665603ull;
        for (size_t b = 0; b < vl; b++) h = (h ^ v[b]) * 1099511628211ull;
        size_t slot = (size_t)(h ^ (h >> 29)) & (hcap - 1);
        for (;;) {
            int32_t e = table[slot];
            if (e < 0) {
                if (d >= 65536) { free(table); free(keys); free(first_row); return LO_ERR_UNSUPPORTED; }
                table[slot] = (int32_t)d;
                first_row[d] = (int32_t)i;
                keys[i] = (uint16_t)d;
                d++;
                break;
            }
            int32_t r = first_row[e];
            size_t rl = (size_t)(offsets[r + 1] - offsets[r]);
            if (rl == vl && memcmp(data + offsets[r], v, vl) == 0) { keys[i] = (uint16_t)e; break; }
            slot = (slot + 1) & (hcap - 1);
        }
    }
    /* dictionary values buffer */
    size_t dl = 0;
    for (size_t k = 0; k < d; k++) dl += (size_t)(offsets[first_row[k] + 1] - offsets[first_row[k]]);
    int32_t* doff = (int32_t*)malloc((d + 1) * sizeof(int32_t));
    uint8_t* dd = (uint8_t*)malloc(dl ? dl : 1);
    doff[0] = 0;
    for (size_t k = 0; k < d; k++) {
        int32_t r = first_row[k];
        size_t vl = (size_t)(offsets[r + 1] - offsets[r]);
        memcpy(dd + doff[k], data + offsets[r], vl);
        doff[k + 1] = doff[k] + (int32_t)vl;
    }
    int64_t rc = lo_bv_encode_dict(arrow_type, keys, validity, n, doff, dd, d, st, build_fingerprints, out, cap);
    free(table); free(keys); free(first_row); free(doff); free(dd);
    return rc;
}

/* serialization.rs:223-325 */
int lo_bv_parse(const uint8_t* bytes, size_t len, lo_bv* bv) {
    memset(bv, 0, sizeof(*bv));
    int logical, phys;
    int rc = lo_ipc_header_read(bytes, len, &logical, &phys);
    if (rc) return rc;
    if (logical != LO_LOGICAL_BYTE_VIEW || len < 40) return LO_ERR_CORRUPT;
    bv->arrow_type = phys;
    uint32_t keys_size = lo_rd_u32(bytes + 16), co_size = lo_rd_u32(bytes + 20), sp_size = lo_rd_u32(bytes + 24),
             fsst_size = lo_rd_u32(bytes + 28), fp_size = lo_rd_u32(bytes + 32);
    size_t cur = lo_align8(36);
    if (cur + fsst_size > len || fsst_size < 12) return LO_ERR_CORRUPT;
    bv->uncompressed_bytes = lo_rd_u64(bytes + cur);
    bv->fsst_len = lo_rd_u32(bytes + cur + 8);
    if (12 + (size_t)bv->fsst_len > fsst_size) return LO_ERR_CORRUPT;
    bv->fsst = bytes + cur + 12;
    cur = lo_align8(cur + fsst_size);
    if (cur + keys_size > len) return LO_ERR_CORRUPT;
    lo_bitpacked_view kv;
    rc = lo_bitpacked_parse(bytes + cur, keys_size, &kv);
    if (rc) return rc;
    bv->n = kv.len;
    bv->nullable = kv.has_nulls || kv.all_null;
    bv->all_null = kv.all_null;
    bv->keys = (uint16_t*)calloc((size_t)kv.len + 1, sizeof(uint16_t));
    if (!kv.all_null) lo_bitunpack(16, kv.bit_width, kv.values, kv.len, bv->keys);
    if (bv->nullable) {
        bv->key_validity = (uint8_t*)calloc(lo_bm_bytes(kv.len) + 1, 1);
        if (!kv.all_null && kv.nulls) memcpy(bv->key_validity, kv.nulls, lo_bm_bytes(kv.len));
    }
    cur = lo_align8(cur + keys_size);
    if (cur + co_size > len) { lo_bv_free(bv); return LO_ERR_CORRUPT; }
    uint32_t count = 0;
    bv->compact_offsets_size = co_size;
    if (co_size > 0) {
        rc = compact_offsets_read(bytes + cur, co_size, &bv->offsets, &count, &bv->offset_bytes);
        if (rc) { lo_bv_free(bv); return rc; }
    } else {
        bv->offsets = (uint32_t*)calloc(1, sizeof(uint32_t));
        bv->offset_bytes = 1;
    }
    bv->d = count ? count - 1 : 0;
    cur = lo_align8(cur + co_size);
    if (cur + (size_t)bv->d * 8 > len) { lo_bv_free(bv); return LO_ERR_CORRUPT; }
    bv->prefix_keys = bytes + cur;
    cur = lo_align8(cur + (size_t)bv->d * 8);
    if (cur + sp_size > len) { lo_bv_free(bv); return LO_ERR_CORRUPT; }
    bv->shared_prefix = bytes + cur;
    bv->shared_prefix_len = sp_size;
    cur = lo_align8(cur + sp_size);
    if (cur + fp_size > len && fp_size) { lo_bv_free(bv); return LO_ERR_CORRUPT; }
    if (fp_size) {
        if (fp_size != bv->d * 4) { lo_bv_free(bv); return LO_ERR_CORRUPT; }
        bv->fingerprints = bytes + cur;
    }
    return LO_OK;
}

void lo_bv_free(lo_bv* bv) {
    free(bv->keys); free(bv->key_validity); free(bv->offsets);
    bv->keys = NULL; bv->key_validity = NULL; bv->offsets = NULL;
}

/* ---------- dictionary-level predicate evaluation (comparisons.rs) ---------- */
static uint8_t* decode_entry(const lo_bv* bv, const lo_symtab* st, uint32_t i, size_t* out_len) {
    const uint8_t* c = bv->fsst + bv->offsets[i];
    size_t cl = bv->offsets[i + 1] - bv->offsets[i];
    size_t dl = lo_fsst_decompressed_len(st, c, cl);
    uint8_t* buf = (uint8_t*)malloc(dl + 8);
    *out_len = lo_fsst_decompress(st, c, cl, buf, dl + 8);
    return buf;
}

/* comparisons.rs:21-82.  The reference compares FSST-compressed bytes against the needle compressed by the same
 * Compressor (:51, :73-77) — equivalent to comparing the decoded value with the plain needle because compression
 * is a deterministic injective function; the oracle decodes so it does not depend on matching greedy choices. */
static void dict_equals(const lo_bv* bv, const lo_symtab* st, const uint8_t* needle, size_t nl, uint8_t* res) {
    memset(res, 0, bv->d ? bv->d : 1);
    size_t spl = bv->shared_prefix_len;
    if (nl < spl || memcmp(needle, bv->shared_prefix, spl) != 0) return;
    const uint8_t* ns = needle + spl;
    size_t nsl = nl - spl;
    for (uint32_t i = 0; i < bv->d; i++) {
        const uint8_t* pk = bv->prefix_keys + 8 * (size_t)i;
        int known = pk[7] != 255;
        size_t l = pk[7];
        if (nsl <= 7) {
            if (known && l == nsl && memcmp(pk, ns, l) == 0) res[i] = 1;
            continue;
        }
        if (known) { if (l != nsl) continue; }
        else if (nsl < 255) continue;
        if (memcmp(pk, ns, 7) != 0) continue;
        size_t vl;
        uint8_t* v = decode_entry(bv, st, i, &vl);
        if (vl == nl && memcmp(v, needle, nl) == 0) res[i] = 1;
        free(v);
    }
}

/* comparisons.rs:469-501: -1 undecided, else 0/1 */
static int shared_prefix_decides(const lo_bv* bv, const uint8_t* needle, size_t nl, int op) {
    size_t spl = bv->shared_prefix_len;
    size_t m = nl < spl ? nl : spl;
    int c = m ? memcmp(bv->shared_prefix, needle, m) : 0;
    if (c < 0) return (op == LO_LT || op == LO_LE) ? 1 : 0;
    if (c > 0) return (op == LO_GT || op == LO_GE) ? 1 : 0;
    if (nl < spl) return (op == LO_GT || op == LO_GE) ? 1 : 0;
    return -1;
}

/* comparisons.rs:114-151 + :351-405 */
static void dict_ordering(const lo_bv* bv, const lo_symtab* st, const uint8_t* needle, size_t nl, int op,
                          uint8_t* res, uint32_t* n_ambiguous) {
    uint32_t amb = 0;
    int dec = shared_prefix_decides(bv, needle, nl, op);
    if (dec >= 0) {
        /* :357-359 — vec![result; dictionary_keys.len()]; only the first d entries are ever read */
        memset(res, dec, bv->d ? bv->d : 1);
        if (n_ambiguous) *n_ambiguous = 0;
        return;
    }
    const uint8_t* ns = needle + bv->shared_prefix_len;
    size_t nsl = nl - bv->shared_prefix_len;
    size_t cmp_len = nsl < 7 ? nsl : 7;
    for (uint32_t i = 0; i < bv->d; i++) {
        const uint8_t* pk = bv->prefix_keys + 8 * (size_t)i;
        if (cmp_len == 0) { /* :367-378 */
            int empty = pk[7] == 0;
            res[i] = (uint8_t)(op == LO_LT ? 0 : op == LO_LE ? empty : op == LO_GT ? !empty : 1);
            continue;
        }
        int c = memcmp(pk, ns, cmp_len);
        if (c < 0) res[i] = (uint8_t)(op == LO_LT || op == LO_LE);
        else if (c > 0) res[i] = (uint8_t)(op == LO_GT || op == LO_GE);
        else { /* ambiguous: decompress and compare the full value (:121-148) */
            amb++;
            size_t vl;
            uint8_t* v = decode_entry(bv, st, i, &vl);
            int o = lo_bytes_cmp(v, vl, needle, nl);
            res[i] = (uint8_t)(op == LO_LT ? o < 0 : op == LO_LE ? o <= 0 : op == LO_GT ? o > 0 : o >= 0);
            free(v);
        }
    }
    if (n_ambiguous) *n_ambiguous = amb;
}

/* comparisons.rs:159-183 + :598-651.  NOTE the reference inverts dict results for NotContains ONLY inside
 * apply_like_match_on_candidates, i.e. only when at least one fingerprint candidate exists (:167-180). */
static void dict_like_substring(const lo_bv* bv, const lo_symtab* st, const uint8_t* inner, size_t il, int negate,
                                uint8_t* res, uint32_t* n_candidates) {
    memset(res, 0, bv->d ? bv->d : 1);
    uint32_t nfp = lo_fingerprint(inner, il); /* fingerprint of the RAW inner bytes (compute_fingerprint_candidates) */
    uint32_t cands = 0;
    int has_escape = 0; /* `inner` always sits inside its pattern: inner[-1] and inner[il] are the two '%' */
    for (size_t i = 0; i < il; i++) has_escape |= inner[i] == '\\';
    for (uint32_t i = 0; i < bv->d; i++) {
        uint32_t fp = lo_rd_u32(bv->fingerprints + 4 * (size_t)i);
        if ((fp & nfp) != nfp) continue;
        cands++;
        size_t vl;
        uint8_t* v = decode_entry(bv, st, i, &vl);
        /* apply_like_match_on_candidates re-forms "%inner%" and runs Arrow LIKE (comparisons.rs:629-634): identical
         * to memmem unless the inner part holds a backslash, which Arrow treats as an escape */
        if (has_escape ? lo_like_match(v, vl, inner - 1, il + 2) : lo_contains(v, vl, inner, il)) res[i] = 1;
        free(v);
    }
    if (cands > 0 && negate)
        for (uint32_t i = 0; i < bv->d; i++) res[i] = !res[i];
    if (n_candidates) *n_candidates = cands;
}

int lo_bv_dict_results(const uint8_t* bytes, size_t len, const lo_symtab* st, int op, const uint8_t* needle,
                       size_t nlen, uint8_t* out_dict, uint32_t* out_candidates) {
    lo_bv bv;
    int rc = lo_bv_parse(bytes, len, &bv);
    if (rc) return rc;
    uint32_t c = 0;
    if (op == LO_EQ || op == LO_NE) {
        dict_equals(&bv, st, needle, nlen, out_dict);
        if (op == LO_NE) for (uint32_t i = 0; i < bv.d; i++) out_dict[i] = !out_dict[i];
    } else if (op <= LO_GE) {
        dict_ordering(&bv, st, needle, nlen, op, out_dict, &c);
    } else {
        const uint8_t* inner; size_t il;
        if (!bv.fingerprints || !lo_substring_pattern(needle, nlen, &inner, &il)) { lo_bv_free(&bv); return LO_ERR_UNSUPPORTED; }
        dict_like_substring(&bv, st, inner, il, op == LO_NOT_LIKE, out_dict, &c);
    }
    if (out_candidates) *out_candidates = c;
    lo_bv_free(&bv);
    return LO_OK;
}

/* arrow filter on the u16 keys (helpers.rs:44-64) */
static size_t filter_keys(const lo_bv* bv, const uint8_t* sel, uint16_t* fk, uint8_t* fvalid) {
    size_t k = 0;
    for (size_t i = 0; i < bv->n; i++) {
        if (sel && !lo_get_bit(sel, i)) continue;
        fk[k] = bv->keys[i];
        if (fvalid) {
            if ((k & 7) == 0) fvalid[k >> 3] = 0;
            if (!bv->nullable || lo_get_bit(bv->key_validity, i)) lo_set_bit(fvalid, k);
        }
        k++;
    }
    return k;
}

int64_t lo_bv_eval_predicate(const uint8_t* bytes, size_t len, const lo_symtab* st, int op, int lit_tag,
                             const uint8_t* lit, size_t lit_len, const uint8_t* sel, uint8_t* out_values,
                             uint8_t* out_validity, int* nullable) {
    lo_bv bv;
    int rc = lo_bv_parse(bytes, len, &bv);
    if (rc) return rc;
    uint16_t* fk = (uint16_t*)malloc(((size_t)bv.n + 1) * sizeof(uint16_t));
    uint8_t* fvalid = (uint8_t*)calloc(lo_bm_bytes(bv.n) + 1, 1);
    size_t k = filter_keys(&bv, sel, fk, fvalid);
    memset(out_values, 0, lo_bm_bytes(k));
    if (out_validity) memcpy(out_validity, fvalid, lo_bm_bytes(k));
    if (nullable) *nullable = bv.nullable;
    int64_t ret = (int64_t)k;
    if (lit_tag == LO_LIT_BOOL) { /* helpers.rs:72-79 UnsupportedExpression::Constant */
        if (lit[0]) {
            memset(out_values, 0xFF, lo_bm_bytes(k));
            if (k & 7) out_values[lo_bm_bytes(k) - 1] &= (uint8_t)((1u << (k & 7)) - 1);
        }
        goto done;
    }
    if (lit_tag != LO_LIT_BYTES) { ret = LO_ERR_ARG; goto done; }
    uint8_t* dres = (uint8_t*)calloc((size_t)bv.d + 1, 1);
    int invert_values = 0; /* compare_not_equals inverts the final VALUES buffer (:85-90) */
    if (op == LO_EQ || op == LO_NE) {
        dict_equals(&bv, st, lit, lit_len, dres);
        invert_values = (op == LO_NE);
    } else if (op <= LO_GE) {
        dict_ordering(&bv, st, lit, lit_len, op, dres, NULL);
    } else if (op == LO_LIKE || op == LO_NOT_LIKE) {
        const uint8_t* inner; size_t il;
        if (bv.fingerprints) {
            if (!lo_substring_pattern(lit, lit_len, &inner, &il)) { free(dres); ret = LO_ERR_UNSUPPORTED; goto done; } /* expect() panics */
            dict_like_substring(&bv, st, inner, il, op == LO_NOT_LIKE, dres, NULL);
        } else {
            /* helpers.rs:86-91 -> None -> eval_predicate_on_array(filtered.to_arrow_array(), expr) (mod.rs:360-361):
             * Arrow `like` over every selected row; nulls stay null, values under nulls unspecified (kept 0). */
            for (uint32_t i = 0; i < bv.d; i++) {
                size_t vl;
                uint8_t* v = decode_entry(&bv, st, i, &vl);
                int m = lo_like_match(v, vl, lit, lit_len);
                dres[i] = (uint8_t)(op == LO_NOT_LIKE ? !m : m);
                free(v);
            }
        }
    } else { free(dres); ret = LO_ERR_UNSUPPORTED; goto done; }
    /* map_dictionary_results_to_array_results (comparisons.rs:325-347): only valid rows, never deref null keys */
    for (size_t i = 0; i < k; i++) {
        if (bv.nullable && !lo_get_bit(fvalid, i)) continue;
        if (fk[i] < bv.d && dres[fk[i]]) lo_set_bit(out_values, i);
    }
    if (invert_values) {
        for (size_t b = 0; b < lo_bm_bytes(k); b++) out_values[b] = (uint8_t)~out_values[b];
        if (k & 7) out_values[lo_bm_bytes(k) - 1] &= (uint8_t)((1u << (k & 7)) - 1);
    }
    free(dres);
done:
    free(fk); free(fvalid);
    lo_bv_free(&bv);
    return ret;
}

/* filter -> to_arrow_array (mod.rs:421-424, :266-290): key filter, decode referenced dictionary entries, cast
 * Dictionary -> value type.  The logical content is emitted as a plain offsets/data/validity triple. */
int64_t lo_bv_filter_to_arrow(const uint8_t* bytes, size_t len, const lo_symtab* st, const uint8_t* sel,
                              int32_t* out_offsets, uint8_t* out_data, size_t data_cap, uint8_t* out_validity,
                              size_t* data_len, int* nullable) {
    lo_bv bv;
    int rc = lo_bv_parse(bytes, len, &bv);
    if (rc) return rc;
    uint16_t* fk = (uint16_t*)malloc(((size_t)bv.n + 1) * sizeof(uint16_t));
    uint8_t* fvalid = (uint8_t*)calloc(lo_bm_bytes(bv.n) + 1, 1);
    size_t k = filter_keys(&bv, sel, fk, fvalid);
    if (nullable) *nullable = bv.nullable;
    if (out_validity) memcpy(out_validity, fvalid, lo_bm_bytes(k));
    /* decode each dictionary entry at most once */
    uint8_t** dec = (uint8_t**)calloc((size_t)bv.d + 1, sizeof(uint8_t*));
    size_t* dlen = (size_t*)calloc((size_t)bv.d + 1, sizeof(size_t));
    size_t total = 0;
    int64_t ret = (int64_t)k;
    if (out_offsets) out_offsets[0] = 0;
    for (size_t i = 0; i < k; i++) {
        size_t vl = 0;
        if (!bv.nullable || lo_get_bit(fvalid, i)) {
            uint16_t key = fk[i];
            if (key >= bv.d) { ret = LO_ERR_CORRUPT; break; }
            if (!dec[key]) dec[key] = decode_entry(&bv, st, key, &dlen[key]);
            vl = dlen[key];
            if (out_data) {
                if (total + vl > data_cap) { ret = LO_ERR_CAPACITY; break; }
                memcpy(out_data + total, dec[key], vl);
            }
        }
        total += vl;
        if (out_offsets) out_offsets[i + 1] = (int32_t)total;
    }
    if (data_len) *data_len = total;
    for (uint32_t i = 0; i < bv.d; i++) free(dec[i]);
    free(dec); free(dlen); free(fk); free(fvalid);
    lo_bv_free(&bv);
    return ret;
}
