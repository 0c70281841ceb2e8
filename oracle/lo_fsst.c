/*
 * FSST (Fast Static Symbol Table) — TEST ORACLE (see lo_common.h).
 *
 * The reference delegates to the third-party crate fsst-rs 0.5.10 (not vendored).  Call sites:
 *   decompress_into   src/core/src/liquid_array/raw/fsst_buffer.rs:103, :566, :653
 *   compress_into     fsst_buffer.rs:73; byte_view_array/comparisons.rs:541-547
 *   Compressor::train fsst_buffer.rs:391-397
 *   symbol table save/load format  fsst_buffer.rs:848-920: [n:u8][len:u8 x n][sym:u64 LE x n]
 * Decode is fully determined by the symbol table: code c < 255 appends symbol c (len[c] bytes, LE in a u64);
 * code 255 is an escape and the next byte is a literal.
 * Training / greedy matching restate the published FSST algorithm (Boncz, Neumann, Leis, VLDB 2020):
 * parity with fsst-rs's choices is UNPINNED and irrelevant to decode / predicate results.
 */
#include "lo_fsst.h"

int lo_symtab_load(const uint8_t* bytes, size_t len, lo_symtab* st) {
    memset(st, 0, sizeof(*st));
    if (len < 1) return LO_ERR_CORRUPT;
    int n = bytes[0];
    if (1 + (size_t)n * 9 > len) return LO_ERR_CORRUPT;
    st->n = n;
    for (int i = 0; i < n; i++) {
        st->len[i] = bytes[1 + i];
        if (st->len[i] == 0 || st->len[i] > 8) return LO_ERR_CORRUPT;
        st->sym[i] = lo_rd_u64(bytes + 1 + n + 8 * (size_t)i);
    }
    return LO_OK;
}

size_t lo_symtab_save(const lo_symtab* st, uint8_t* out) {
    int n = st->n;
    out[0] = (uint8_t)n;
    for (int i = 0; i < n; i++) out[1 + i] = st->len[i];
    for (int i = 0; i < n; i++) lo_wr_u64(out + 1 + n + 8 * (size_t)i, st->sym[i]);
    return 1 + (size_t)n * 9;
}

size_t lo_fsst_decompressed_len(const lo_symtab* st, const uint8_t* in, size_t in_len) {
    size_t o = 0;
    for (size_t i = 0; i < in_len; i++) {
        uint8_t c = in[i];
        if (c == LO_FSST_ESC) { i++; if (i < in_len) o++; }
        else o += st->len[c];
    }
    return o;
}

size_t lo_fsst_decompress(const lo_symtab* st, const uint8_t* in, size_t in_len, uint8_t* out, size_t cap) {
    size_t o = 0;
    for (size_t i = 0; i < in_len; i++) {
        uint8_t c = in[i];
        if (c == LO_FSST_ESC) {
            i++;
            if (i >= in_len) break;
            if (o + 1 > cap) return (size_t)-1;
            out[o++] = in[i];
        } else {
            uint8_t l = st->len[c];
            if (o + l > cap) return (size_t)-1;
            uint64_t s = st->sym[c];
            for (uint8_t b = 0; b < l; b++) out[o + b] = (uint8_t)(s >> (8 * b));
            o += l;
        }
    }
    return o;
}

/* ---- longest-match lookup: tiny open-addressing map (len, bytes) -> code ---- */
#define HT_SIZE 1024
typedef struct {
    uint64_t key[HT_SIZE];
    uint8_t klen[HT_SIZE];
    int16_t code[HT_SIZE];
} sym_map;

static inline uint32_t hsh(uint64_t k, int l) {
    uint64_t x = (k ^ ((uint64_t)l << 56)) * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> 54) & (HT_SIZE - 1);
}
static void map_clear(sym_map* m) { for (int i = 0; i < HT_SIZE; i++) m->code[i] = -1; }
static void map_put(sym_map* m, uint64_t k, int l, int code) {
    uint32_t h = hsh(k, l);
    while (m->code[h] >= 0) {
        if (m->key[h] == k && m->klen[h] == l) return;
        h = (h + 1) & (HT_SIZE - 1);
    }
    m->key[h] = k; m->klen[h] = (uint8_t)l; m->code[h] = (int16_t)code;
}
static int map_get(const sym_map* m, uint64_t k, int l) {
    uint32_t h = hsh(k, l);
    while (m->code[h] >= 0) {
        if (m->key[h] == k && m->klen[h] == l) return m->code[h];
        h = (h + 1) & (HT_SIZE - 1);
    }
    return -1;
}
static inline uint64_t load_le(const uint8_t* p, size_t avail) {
    uint64_t v = 0;
    size_t l = avail < 8 ? avail : 8;
    memcpy(&v, p, l);
    return v;
}
static inline uint64_t lmask(int l) { return l >= 8 ? ~(uint64_t)0 : ((((uint64_t)1) << (8 * l)) - 1); }

static void map_build(sym_map* m, const lo_symtab* st) {
    map_clear(m);
    for (int i = 0; i < st->n; i++) map_put(m, st->sym[i] & lmask(st->len[i]), st->len[i], i);
}

/* returns code (>=0) and its length, or -1 when no symbol matches at p */
static inline int find_longest(const sym_map* m, const uint8_t* p, size_t avail, int* out_len) {
    uint64_t w = load_le(p, avail);
    int maxl = avail < 8 ? (int)avail : 8;
    for (int l = maxl; l >= 1; l--) {
        int c = map_get(m, w & lmask(l), l);
        if (c >= 0) { *out_len = l; return c; }
    }
    return -1;
}

size_t lo_fsst_compress(const lo_symtab* st, const uint8_t* in, size_t len, uint8_t* out) {
    sym_map* m = (sym_map*)malloc(sizeof(sym_map));
    map_build(m, st);
    size_t o = 0, pos = 0;
    while (pos < len) {
        int l = 0;
        int c = find_longest(m, in + pos, len - pos, &l);
        if (c >= 0) { out[o++] = (uint8_t)c; pos += (size_t)l; }
        else { out[o++] = LO_FSST_ESC; out[o++] = in[pos++]; }
    }
    free(m);
    return o;
}

/* ---- training: 5 generations of count + select (FSST paper, Algorithm 3, simplified) ---- */
typedef struct { uint64_t sym; uint8_t len; uint64_t gain; } cand;

static int cand_cmp(const void* a, const void* b) {
    const cand* x = (const cand*)a; const cand* y = (const cand*)b;
    if (x->gain != y->gain) return x->gain < y->gain ? 1 : -1;
    if (x->len != y->len) return x->len < y->len ? 1 : -1;
    if (x->sym != y->sym) return x->sym < y->sym ? -1 : 1;
    return 0;
}

static int cand_key_cmp(const void* x, const void* y) {
    const cand* p = (const cand*)x; const cand* q = (const cand*)y;
    if (p->len != q->len) return p->len < q->len ? -1 : 1;
    if (p->sym != q->sym) return p->sym < q->sym ? -1 : 1;
    return 0;
}

void lo_fsst_train(const uint8_t* data, const int32_t* offsets, size_t n, lo_symtab* st) {
    memset(st, 0, sizeof(*st));
    if (n == 0) return;
    /* deterministic sample of ~64 KiB */
    size_t total = (size_t)(offsets[n] - offsets[0]);
    size_t step = total / 65536 + 1;
    /* codes: 0..255 = raw bytes (escapes), 256.. = symbols of the current table */
    uint32_t* count1 = (uint32_t*)calloc(512, sizeof(uint32_t));
    uint32_t* count2 = (uint32_t*)calloc(512 * 512, sizeof(uint32_t));
    sym_map* m = (sym_map*)malloc(sizeof(sym_map));
    cand* cands = (cand*)malloc(sizeof(cand) * (512 + 65536));
    for (int gen = 0; gen < 5; gen++) {
        memset(count1, 0, 512 * sizeof(uint32_t));
        memset(count2, 0, 512 * 512 * sizeof(uint32_t));
        map_build(m, st);
        for (size_t i = 0; i < n; i += step) {
            const uint8_t* s = data + offsets[i];
            size_t sl = (size_t)(offsets[i + 1] - offsets[i]);
            size_t pos = 0;
            int prev = -1;
            while (pos < sl) {
                int l = 1;
                int c = find_longest(m, s + pos, sl - pos, &l);
                int code = c >= 0 ? 256 + c : s[pos];
                if (c < 0) l = 1;
                count1[code]++;
                if (prev >= 0) count2[prev * 512 + code]++;
                if (l > 1) { /* also count the first byte so single bytes can bootstrap */
                    count1[s[pos]]++;
                    if (prev >= 0) count2[prev * 512 + s[pos]]++;
                }
                prev = code;
                pos += (size_t)l;
            }
        }
        /* candidates */
        size_t nc = 0;
        uint64_t csym[512]; uint8_t clen[512];
        for (int c = 0; c < 512; c++) {
            if (c < 256) { csym[c] = (uint64_t)c; clen[c] = 1; }
            else if (c - 256 < st->n) { csym[c] = st->sym[c - 256] & lmask(st->len[c - 256]); clen[c] = st->len[c - 256]; }
            else { csym[c] = 0; clen[c] = 0; }
        }
        for (int c = 0; c < 512; c++) {
            if (!count1[c] || !clen[c]) continue;
            cands[nc].sym = csym[c]; cands[nc].len = clen[c];
            cands[nc].gain = (uint64_t)count1[c] * clen[c];
            nc++;
        }
        for (int a = 0; a < 512 && nc < 512 + 65000; a++) {
            if (!clen[a]) continue;
            for (int b = 0; b < 512 && nc < 512 + 65000; b++) {
                uint32_t cnt = count2[a * 512 + b];
                if (cnt < 2 || !clen[b]) continue;
                int la = clen[a], lb = clen[b];
                if (la >= 8) continue;
                int lc = la + lb > 8 ? 8 : la + lb;
                uint64_t s = csym[a] | (csym[b] << (8 * la));
                s &= lmask(lc);
                cands[nc].sym = s; cands[nc].len = (uint8_t)lc;
                cands[nc].gain = (uint64_t)cnt * (uint64_t)lc;
                nc++;
            }
        }
        /* merge duplicates: sort by (len,sym) then accumulate gains */
        {
            qsort(cands, nc, sizeof(cand), cand_key_cmp);
            size_t w = 0;
            for (size_t i = 0; i < nc; i++) {
                if (w > 0 && cands[w - 1].len == cands[i].len && cands[w - 1].sym == cands[i].sym) cands[w - 1].gain += cands[i].gain;
                else cands[w++] = cands[i];
            }
            nc = w;
        }
        qsort(cands, nc, sizeof(cand), cand_cmp);
        lo_symtab next;
        memset(&next, 0, sizeof(next));
        for (size_t i = 0; i < nc && next.n < 255; i++) {
            /* a single byte seen once is not worth a code (an escape costs the same 2 bytes) */
            if (cands[i].len == 1 && cands[i].gain < 2) continue;
            next.sym[next.n] = cands[i].sym;
            next.len[next.n] = cands[i].len;
            next.n++;
        }
        *st = next;
    }
    free(count1); free(count2); free(m); free(cands);
}
