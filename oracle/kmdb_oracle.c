/*
 * kmdb_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See kmdb_oracle.h.
 *
 * Every function restates (does not copy) the behaviour of the cited reference code in
 * /root/reference/src.  Single-threaded, scalar, written for clarity.
 */
#define _FILE_OFFSET_BITS 64
#include "kmdb_oracle.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <stdarg.h>
#include <math.h>
#include <sys/types.h>

static char g_err[512];
const char* kmo_last_error(void) { return g_err; }
static void set_err(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}

/* ------------------------------------------------------------------------------------ */
/* Elias gamma.  Code shape (elias_gamma.h:104-128): value v>=1 with bit length L is     */
/* written as (L-1) one bits, a zero bit, then the low (L-1) bits of v; the stream is     */
/* MSB-first inside little-endian uint64 words (word 0 bit 63 is the first bit).          */
/* ------------------------------------------------------------------------------------ */
static inline uint32_t get_bit(const uint64_t* d, uint32_t pos) {
    return (uint32_t)((d[pos >> 6] >> (63u - (pos & 63u))) & 1u);
}

uint32_t kmo_gamma_decode(const uint64_t* data, uint32_t num_bits, uint32_t* out) {
    /* CEliasGamma::Decode(input, size_in_bits, output) (elias_gamma.h:371-378): decode
     * values until the bit cursor reaches num_bits. */
    uint32_t pos = 0, n = 0;
    while (pos < num_bits) {
        uint32_t ones = 0;
        while (get_bit(data, pos)) { ++ones; ++pos; }
        ++pos;                              /* the terminating zero */
        uint32_t v = 1;
        for (uint32_t i = 0; i < ones; ++i) { v = (v << 1) | get_bit(data, pos); ++pos; }
        out[n++] = v;
    }
    return n;
}

static inline void put_bit(uint64_t* d, uint32_t pos, uint32_t b) {
    if (b) d[pos >> 6] |= 1ull << (63u - (pos & 63u));
}

uint32_t kmo_gamma_encode(const uint32_t* values, uint32_t n, uint64_t* out) {
    uint32_t pos = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = values[i], L = 0;
        for (uint32_t t = v; t; t >>= 1) ++L;
        for (uint32_t j = 0; j + 1 < L; ++j) put_bit(out, pos++, 1);
        put_bit(out, pos++, 0);
        for (int j = (int)L - 2; j >= 0; --j) put_bit(out, pos++, (v >> j) & 1u);
    }
    return pos;
}

/* pattern_t::decodeSamples (pattern.cpp:99-109): last id is explicit, the stream holds the
 * l-1 deltas in append order, ids are rebuilt right-to-left by subtraction. */
void kmo_decode_local(const kmo_pattern* p, uint32_t* out) {
    uint32_t l = p->num_local;
    if (!l) return;
    out[l - 1] = p->last_sample_id;
    if (l > 1) {
        kmo_gamma_decode(p->data, p->num_bits, out);
        for (int i = (int)l - 2; i >= 0; --i) out[i] = out[i + 1] - out[i];
    }
}

/* decode_pattern_samples (similarity_calculator.h:61-77): walk node -> parent -> ..., each
 * node writes its local ids right-to-left, giving the full ascending list. */
uint32_t kmo_decode_chain(const kmo_db* db, int64_t pid, uint32_t* out) {
    const kmo_pattern* p = &db->patterns[pid];
    uint32_t n = p->num_samples;
    uint32_t* w = out + n;
    int64_t cur = pid;
    while (cur >= 0) {
        const kmo_pattern* c = &db->patterns[cur];
        w -= c->num_local;
        kmo_decode_local(c, w);
        cur = c->parent_id;
    }
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* .db reader (prefix_kmer_db.cpp:578-748, hashmap_lp.h:546-605, pattern.cpp:50-95)       */
/* ------------------------------------------------------------------------------------ */
static int rd(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

kmo_db* kmo_db_load(const char* path, int mode) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_err("Cannot open k-mer database %s", path); return NULL; }
    kmo_db* db = (kmo_db*)calloc(1, sizeof(kmo_db));
    int ok = 1;
    ok &= rd(f, &db->format_word, 8);
    ok &= rd(f, &db->kmer_length, 4);
    ok &= rd(f, &db->fraction, 8);
    ok &= rd(f, &db->start_fraction, 8);
    ok &= rd(f, &db->alphabet, 4);
    ok &= rd(f, &db->is_initialized, 1);
    ok &= rd(f, &db->kmers_count, 8);
    ok &= rd(f, &db->n_samples, 8);
    if (!ok) { set_err("truncated header in %s", path); goto fail; }
    db->sample_names = (char**)calloc(db->n_samples ? db->n_samples : 1, sizeof(char*));
    db->sample_kmers = (uint64_t*)calloc(db->n_samples ? db->n_samples : 1, 8);
    for (uint64_t i = 0; i < db->n_samples; ++i) {
        uint64_t len;
        if (!rd(f, &db->sample_kmers[i], 8) || !rd(f, &len, 8)) { set_err("truncated sample table"); goto fail; }
        db->sample_names[i] = (char*)malloc(len + 1);
        if (len && !rd(f, db->sample_names[i], len)) { set_err("truncated sample name"); goto fail; }
        db->sample_names[i][len] = 0;
    }
    if (!rd(f, &db->n_buckets, 8)) { set_err("truncated bucket count"); goto fail; }
    db->tables = (kmo_hashtable*)calloc(db->n_buckets ? db->n_buckets : 1, sizeof(kmo_hashtable));
    if (!(db->format_word & 1ull)) { set_err("non-raw hashtable serialisation is not supported (SURVEY quirk 3)"); goto fail; }
    for (uint64_t b = 0; b < db->n_buckets; ++b) {
        kmo_hashtable* t = &db->tables[b];
        ok = rd(f, &t->max_fill, 8) && rd(f, &t->filled, 8) && rd(f, &t->allocated, 8) &&
             rd(f, &t->size_when_restruct, 8) && rd(f, &t->mask, 8) && rd(f, &t->ht_memory, 8) &&
             rd(f, &t->ht_total, 8) && rd(f, &t->ht_match, 8);
        if (!ok) { set_err("truncated hashtable header"); goto fail; }
        uint64_t bv_words = (t->allocated + 63) / 64;
        if (mode == 2) {
            if (fseeko(f, (off_t)(8 * bv_words + 8 * t->filled), SEEK_CUR)) { set_err("seek failed"); goto fail; }
            continue;
        }
        uint64_t* bv = (uint64_t*)malloc(8 * (bv_words ? bv_words : 1));
        uint64_t* items = (uint64_t*)malloc(8 * (t->filled ? t->filled : 1));
        if ((bv_words && !rd(f, bv, 8 * bv_words)) || (t->filled && !rd(f, items, 8 * t->filled))) {
            free(bv); free(items); set_err("truncated hashtable body"); goto fail;
        }
        t->slots = (uint64_t*)malloc(8 * t->allocated);
        uint64_t it = 0;
        for (uint64_t i = 0; i < t->allocated; ++i) {
            if (bv[i >> 6] & (1ull << (i & 63))) t->slots[i] = items[it++];
            else t->slots[i] = ((uint64_t)0x7fffffffu) << 32;     /* key 0, val INT32_MAX = empty (hashmap_lp.h:78) */
        }
        free(bv); free(items);
    }
    if (!rd(f, &db->n_patterns, 8)) { set_err("truncated pattern count"); goto fail; }
    db->patterns = (kmo_pattern*)calloc(db->n_patterns ? db->n_patterns : 1, sizeof(kmo_pattern));
    {
        /* patterns arrive in blocks: u64 blockBytes + packed patterns (prefix_kmer_db.cpp:714-738) */
        size_t cap = 1 << 20, used = 0;
        char* blob = (char*)malloc(cap);
        uint64_t pid = 0;
        /* first pass: slurp all blocks */
        size_t* block_off = NULL; size_t nblocks = 0;
        while (pid < db->n_patterns) {
            uint64_t bs;
            if (!rd(f, &bs, 8)) { free(blob); set_err("truncated pattern block header"); goto fail; }
            if (used + bs > cap) { while (used + bs > cap) cap *= 2; blob = (char*)realloc(blob, cap); }
            if (bs && !rd(f, blob + used, bs)) { free(blob); set_err("truncated pattern block"); goto fail; }
            /* count patterns in this block */
            size_t o = used, end = used + bs;
            while (o < end) {
                uint32_t nb; memcpy(&nb, blob + o + 28, 4);
                o += 40 + (size_t)((nb + 127) / 128) * 16;
                ++pid;
            }
            used = end;
            block_off = (size_t*)realloc(block_off, (++nblocks) * sizeof(size_t));
        }
        free(block_off);
        db->blob_ = blob; db->blob_bytes_ = used;
        size_t o = 0;
        for (uint64_t i = 0; i < db->n_patterns; ++i) {
            kmo_pattern* p = &db->patterns[i];
            memcpy(&p->num_kmers, blob + o, 8);
            memcpy(&p->parent_id, blob + o + 8, 8);
            memcpy(&p->num_samples, blob + o + 16, 4);
            memcpy(&p->num_local, blob + o + 20, 4);
            memcpy(&p->last_sample_id, blob + o + 24, 4);
            memcpy(&p->num_bits, blob + o + 28, 4);
            memcpy(&p->is_parent, blob + o + 32, 4);     /* bytes 36..39 undefined on disk (pattern.cpp:35-37) */
            size_t db_bytes = (size_t)((p->num_bits + 127) / 128) * 16;
            p->data = db_bytes ? (const uint64_t*)(blob + o + 40) : NULL;
            o += 40 + db_bytes;
        }
        db->pattern_section_bytes = used;
    }
    fclose(f);
    return db;
fail:
    fclose(f);
    kmo_db_free(db);
    return NULL;
}

void kmo_db_free(kmo_db* db) {
    if (!db) return;
    if (db->sample_names) { for (uint64_t i = 0; i < db->n_samples; ++i) free(db->sample_names[i]); free(db->sample_names); }
    free(db->sample_kmers);
    if (db->tables) { for (uint64_t b = 0; b < db->n_buckets; ++b) free(db->tables[b].slots); free(db->tables); }
    free(db->patterns);
    free(db->blob_);
    free(db);
}

/* ------------------------------------------------------------------------------------ */
/* all2all, tree form (similarity_calculator.cpp:42-438)                                  */
/* ------------------------------------------------------------------------------------ */
int kmo_all2all_dense(const kmo_db* db, uint32_t* out) {
    uint64_t N = db->n_samples, P = db->n_patterns;
    memset(out, 0, (size_t)(N ? N * (N - 1) / 2 : 0) * 4);
    /* :64-72 — internal nodes receive the k-mer counts of their whole subtree */
    int64_t* W = (int64_t*)malloc(8 * (P ? P : 1));
    for (uint64_t i = 0; i < P; ++i) W[i] = db->patterns[i].num_kmers;
    for (int64_t i = (int64_t)P - 1; i > 0; --i) {
        int64_t par = db->patterns[i].parent_id;
        if (par >= 0) W[par] += W[i];
    }
    uint32_t* buf = (uint32_t*)malloc(4 * (N ? N : 1));
    for (uint64_t pid = 0; pid < P; ++pid) {
        const kmo_pattern* p = &db->patterns[pid];
        uint32_t n = kmo_decode_chain(db, (int64_t)pid, buf);
        uint32_t to_add = (uint32_t)W[pid];                    /* :222 int64 -> uint32 */
        /* :156-160 + :213-233 — one row_add per LOCAL id: row = id, columns = everything before it */
        for (uint32_t i = n - p->num_local; i < n; ++i) {
            uint64_t a = buf[i];
            uint32_t* row = out + a * (a - 1) / 2;             /* array.h:140 */
            for (uint32_t u = 0; u < i; ++u) row[buf[u]] += to_add;   /* simd/row_add_avx2.cpp:30-124 */
        }
    }
    free(buf); free(W);
    return 0;
}

/* all2all_sp semantics (:442-657): no subtree accumulation, every pattern adds its own
 * num_kmers to all C(n,2) pairs; bubbles (bubble_helper.h:127-152, array.h:416-422) add the
 * same amounts later, so the final cell values equal this dense flat-form matrix. */
int kmo_all2all_flat(const kmo_db* db, uint32_t* out) {
    uint64_t N = db->n_samples, P = db->n_patterns;
    memset(out, 0, (size_t)(N ? N * (N - 1) / 2 : 0) * 4);
    uint32_t* buf = (uint32_t*)malloc(4 * (N ? N : 1));
    for (uint64_t pid = 0; pid < P; ++pid) {
        const kmo_pattern* p = &db->patterns[pid];
        uint32_t n = kmo_decode_chain(db, (int64_t)pid, buf);
        uint32_t to_add = (uint32_t)p->num_kmers;
        for (uint32_t j = 1; j < n; ++j) {
            uint64_t a = buf[j];
            uint32_t* row = out + a * (a - 1) / 2;
            for (uint32_t k = 0; k < j; ++k) row[buf[k]] += to_add;
        }
    }
    free(buf);
    return 0;
}

void kmo_update_counts(const kmo_db* db, uint64_t* tree_updates, uint64_t* flat_updates, uint64_t* sum_matrix) {
    uint64_t t = 0, fl = 0, s = 0;
    for (uint64_t pid = 0; pid < db->n_patterns; ++pid) {
        const kmo_pattern* p = &db->patterns[pid];
        uint64_t n = p->num_samples, l = p->num_local;
        t += (n - l) * l + l * (l - 1) / 2;
        fl += n * (n - 1) / 2;
        s += (uint64_t)p->num_kmers * (n * (n - 1) / 2);       /* SURVEY §7 checksum identity */
    }
    if (tree_updates) *tree_updates = t;
    if (flat_updates) *flat_updates = fl;
    if (sum_matrix) *sum_matrix = s;
}

/* ------------------------------------------------------------------------------------ */
/* new2all: hash lookup + per-pattern hit histogram + decode-scatter (:809-925)           */
/* ------------------------------------------------------------------------------------ */
static inline uint32_t fmix32(uint32_t h) {                    /* hashmap_lp.h:53-64 */
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

int32_t kmo_ht_find(const kmo_hashtable* ht, uint32_t key) {   /* hashmap_lp.h:308-333 */
    if (!ht->slots) return -1;
    uint64_t h = (uint64_t)fmix32(key) & ht->mask;
    for (;;) {
        uint64_t it = ht->slots[h];
        int32_t val = (int32_t)(it >> 32);
        if (val == 0x7fffffff) return -1;
        if ((uint32_t)it == key) return val;
        h = (h + 1) & ht->mask;
    }
}

int kmo_one2all(const kmo_db* db, const uint64_t* kmers, size_t n, uint32_t* out) {
    uint64_t N = db->n_samples, P = db->n_patterns;
    memset(out, 0, 4 * (size_t)N);
    int32_t* hits = (int32_t*)calloc(P ? P : 1, 4);
    for (size_t i = 0; i < n; ++i) {
        uint64_t prefix = kmers[i] >> 32;                       /* types.h:25-27 */
        uint32_t suffix = (uint32_t)kmers[i];
        if (prefix >= db->n_buckets) continue;
        int32_t pid = kmo_ht_find(&db->tables[prefix], suffix);
        if (pid < 0) continue;
        if (db->patterns[pid].num_kmers == 0) continue;         /* :847-848 */
        ++hits[pid];
    }
    uint32_t* buf = (uint32_t*)malloc(4 * (N ? N : 1));
    for (uint64_t pid = 0; pid < P; ++pid) {
        if (!hits[pid]) continue;
        uint32_t ns = kmo_decode_chain(db, (int64_t)pid, buf);
        for (uint32_t i = 0; i < ns; ++i) out[buf[i]] += (uint32_t)hits[pid];   /* :896-917 */
    }
    free(buf); free(hits);
    return 0;
}

/* db2db_sp (similarity_calculator.cpp:1225-1540) restated densely: out[r * n_col + c] = number of k-mers shared by
 * sample r of db_row and sample c of db_col.  The reference merge-joins the sorted (suffix, pattern) lists of every
 * prefix (:1252-1285), counts equal (pattern_row, pattern_col) pairs (:1300-1330) and adds the count to every pair of
 * samples of the two patterns' full lists (:1340-1500).  Here: look every k-mer of db_col up in db_row's tables. */
int kmo_db2db_dense(const kmo_db* db_row, const kmo_db* db_col, uint32_t* out) {
    const uint64_t nr = db_row->n_samples, nc = db_col->n_samples;
    memset(out, 0, 4 * (size_t)(nr * nc ? nr * nc : 1));
    if (db_row->kmer_length != db_col->kmer_length) return 1;
    uint32_t* br = (uint32_t*)malloc(4 * (nr ? nr : 1));
    uint32_t* bc = (uint32_t*)malloc(4 * (nc ? nc : 1));
    for (uint64_t b = 0; b < db_col->n_buckets && b < db_row->n_buckets; ++b) {
        const kmo_hashtable* t = &db_col->tables[b];
        if (!t->slots) continue;
        for (uint64_t s = 0; s < t->allocated; ++s) {
            const int32_t pc = (int32_t)(t->slots[s] >> 32);
            if (pc == INT32_MAX) continue;                     /* empty slot (hashmap_lp.h:78) */
            const int32_t pr = kmo_ht_find(&db_row->tables[b], (uint32_t)t->slots[s]);
            if (pr < 0) continue;
            const uint32_t n1 = kmo_decode_chain(db_row, pr, br), n2 = kmo_decode_chain(db_col, pc, bc);
            for (uint32_t i = 0; i < n1; ++i)
                for (uint32_t j = 0; j < n2; ++j) out[(size_t)br[i] * nc + bc[j]] += 1u;
        }
    }
    free(br); free(bc);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* query-side k-mer extraction (kmer_extract.h:13-97), nt alphabet                        */
/* ------------------------------------------------------------------------------------ */
static inline uint64_t mh_fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static uint64_t minhash_hash(uint64_t kmer, uint64_t k_div_4) {  /* filter.h:96-115 */
    uint64_t h = kmer;
    h *= 0x87c37b91114253d5ull; h = rotl64(h, 31); h *= 0x4cf5ad432745937full;
    uint64_t h1 = 42 ^ h; h1 ^= k_div_4;
    uint64_t h2 = 42 ^ k_div_4;
    h1 += h2; h2 += h1;
    h1 = mh_fmix64(h1); h2 = mh_fmix64(h2);
    h1 += h2; h2 += h1;
    return h1 ^ h2;
}

/* KmerHelper::extract (kmer_extract.h:13-97) over any of the reference's alphabets (alphabet.h:22-66: the groups of a comma-separated
 * description are the symbols, upper and lower case alike; bits per symbol = ceil(log2(groups)); alphabet.h:79-86 lists them:
 * nt "A,C,G,TU", aa "K,R,E,D,Q,N,C,G,H,I,L,V,M,F,Y,W,P,S,T,A", aa11_diamond "KREDQN,C,G,H,ILV,M,F,Y,W,P,STA", aa12_mmseqs
 * "AST,C,DN,EQ,FY,G,H,IV,KR,LM,P,W", aa6_dayhoff "STPAG,NDEQ,HRK,MILV,FYW,C"; every protein alphabet preserves the strand). */
size_t kmo_extract_kmers_alphabet(const char* seq, size_t len, uint32_t k, const char* groups, double fraction, double start_fraction,
                                  int preserve_strand, uint64_t* out) {
    int8_t map[256];
    memset(map, -1, sizeof map);
    int size = 1;
    for (const char* g = groups; *g; ++g) {
        if (*g == ',') { ++size; continue; }
        const unsigned char c = (unsigned char)*g;
        map[(unsigned char)tolower(c)] = map[(unsigned char)toupper(c)] = (int8_t)(size - 1);     /* alphabet.h:52-58 */
    }
    int bits = 0;
    while ((1 << bits) < size) ++bits;                          /* alphabet.h:36 */
    if (len < k) return 0;
    uint64_t mask = ((uint32_t)bits * k >= 64) ? ~0ull : ((1ull << ((uint32_t)bits * k)) - 1);
    uint32_t shift_hi = (k - 1) * (uint32_t)bits;
    /* kmer_extract.h:37-45: force at least an 8-bit prefix above the 32-bit suffix */
    int prefix_bits = (int)k * bits - 32;
    uint32_t pshift = 0; uint64_t tail_mask = 0;
    if (prefix_bits < 8) { pshift = (uint32_t)(8 - prefix_bits); tail_mask = (1ull << pshift) - 1; }
    int use_filter = fraction < 1.0;                            /* filter.h:135-142 */
    uint64_t min_thr = (uint64_t)((double)UINT64_MAX * start_fraction);
    uint64_t max_thr = (uint64_t)((double)UINT64_MAX * (start_fraction + fraction));
    uint64_t k_div_4 = (uint64_t)ceil((double)k / 4);

    uint64_t fwd = 0, rev = 0;
    int omit = 0;
    size_t cnt = 0;
    for (size_t i = 0; i < len; ++i) {
        int s = map[(unsigned char)seq[i]];
        if (s < 0) { s = 0; omit = (i < k - 1) ? (int)i + 1 : (int)k; }   /* :52-56, :64-68 */
        fwd = ((fwd << bits) + (uint64_t)s) & mask;
        rev = (rev >> bits) + ((uint64_t)(size - 1 - s) << shift_hi);     /* :59, :73 (3 - s in the prologue: the same for nt, unused otherwise) */
        if (i < k - 1) continue;
        if (omit > 0) { --omit; continue; }
        uint64_t can = preserve_strand ? fwd : (fwd < rev ? fwd : rev);
        can = (can << pshift) | (can & tail_mask);
        if (use_filter) {
            uint64_t h = minhash_hash(can, k_div_4);
            if (!(h >= min_thr && h < max_thr)) continue;
        }
        out[cnt++] = can;
    }
    return cnt;
}
size_t kmo_extract_kmers(const char* seq, size_t len, uint32_t k, double fraction, double start_fraction,
                         int preserve_strand, uint64_t* out) {
    return kmo_extract_kmers_alphabet(seq, len, k, "A,C,G,TU", fraction, start_fraction, preserve_strand, out);   /* alphabet.h:80 */
}

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y);
}
size_t kmo_sort_unique(uint64_t* kmers, size_t n) {
    if (!n) return 0;
    qsort(kmers, n, 8, cmp_u64);
    size_t w = 1;
    for (size_t i = 1; i < n; ++i) if (kmers[i] != kmers[w - 1]) kmers[w++] = kmers[i];
    return w;
}
