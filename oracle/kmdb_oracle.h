/*
 * kmdb_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the kmer-db v2.3.1 all2all / all2all-sp / new2all hot path,
 * written from the behaviour of the reference (file:line citations are relative to
 * /root/reference/src).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this.  The product path (kmer-db_amd/) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against the
 * reference's own golden CSVs (test/virus/k18*.csv, test/synth/{a2a,n2a}*) and, when
 * oracle/_ref is built (the real reference hot path compiled from /root/reference), against
 * the reference's raw matrices on the same .db files.
 */
#ifndef KMDB_ORACLE_H
#define KMDB_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pattern_t header as stored by pattern_t::pack (pattern.cpp:15-46) */
typedef struct {
    int64_t  num_kmers;        /* on-disk value (NOT subtree-accumulated) */
    int64_t  parent_id;        /* -1 for roots */
    uint32_t num_samples;      /* ids in this node and all ancestors */
    uint32_t num_local;        /* ids stored in this node */
    uint32_t last_sample_id;
    uint32_t num_bits;         /* gamma bitstream length */
    uint32_t is_parent;
    const uint64_t* data;      /* ((num_bits+127)/128)*2 words, or NULL */
} kmo_pattern;

/* hash_map_lp<uint32_t,int32_t> restored slot-exact (hashmap_lp.h:546-605) */
typedef struct {
    double   max_fill;
    uint64_t filled, allocated, size_when_restruct, mask, ht_memory, ht_total, ht_match;
    uint64_t* slots;           /* allocated items: low 32 = key, high 32 = val; val==INT32_MAX => empty. NULL if skipped */
} kmo_hashtable;

typedef struct {
    uint64_t format_word;
    uint32_t kmer_length;
    double   fraction;
    double   start_fraction;
    int32_t  alphabet;
    uint8_t  is_initialized;
    uint64_t kmers_count;
    uint64_t n_samples;
    char**   sample_names;
    uint64_t* sample_kmers;    /* stored as u64 on disk; reference keeps u32 */
    uint64_t n_buckets;
    kmo_hashtable* tables;
    uint64_t n_patterns;
    kmo_pattern* patterns;
    uint64_t pattern_section_bytes; /* sum over patterns of 40 + data bytes (SURVEY §8d B_pat) */
    void*    blob_;            /* owns pattern data */
    size_t   blob_bytes_;
} kmo_db;

/* mode: 0 = Everything, 2 = SkipHashtables (kmer_db.h:55-60) */
kmo_db* kmo_db_load(const char* path, int mode);
void    kmo_db_free(kmo_db* db);
const char* kmo_last_error(void);

/* CEliasGamma::Decode (elias_gamma.h:371-378, code shape :104-128): returns #values */
uint32_t kmo_gamma_decode(const uint64_t* data, uint32_t num_bits, uint32_t* out);
/* reverse direction, used only to fabricate test inputs */
uint32_t kmo_gamma_encode(const uint32_t* values, uint32_t n, uint64_t* out_words /* zeroed */);

/* pattern_t::decodeSamples (pattern.cpp:99-109) */
void kmo_decode_local(const kmo_pattern* p, uint32_t* out);
/* decode_pattern_samples (similarity_calculator.h:61-77): full ascending list, returns num_samples */
uint32_t kmo_decode_chain(const kmo_db* db, int64_t pid, uint32_t* out);

/* SimilarityCalculator::all2all (similarity_calculator.cpp:42-438), tree form.
 * out: N(N-1)/2 uint32, row i at i(i-1)/2 (array.h:136-140). Zeroed here. */
int kmo_all2all_dense(const kmo_db* db, uint32_t* out);
/* flat form (all2all_sp semantics, :442-657 incl. bubbles, array.h:391-446) into the same dense layout */
int kmo_all2all_flat(const kmo_db* db, uint32_t* out);
/* cell-update counters for reporting (tree form and flat form) and sum of matrix */
void kmo_update_counts(const kmo_db* db, uint64_t* tree_updates, uint64_t* flat_updates, uint64_t* sum_matrix);

/* hash_map_lp::find (hashmap_lp.h:308-333) with fmix32 (:53-64); returns pid or -1 */
int32_t kmo_ht_find(const kmo_hashtable* ht, uint32_t key);
/* SimilarityCalculator::one2all<false> (:809-925): out[N] zeroed here */
int kmo_one2all(const kmo_db* db, const uint64_t* kmers, size_t n, uint32_t* out);
/* db2db_sp (similarity_calculator.cpp:1225-1540), dense: out[r * db_col->n_samples + c] = shared k-mers of row sample r and column sample c */
int kmo_db2db_dense(const kmo_db* db_row, const kmo_db* db_col, uint32_t* out);

/* KmerHelper::extract (kmer_extract.h:13-97) for the nt alphabet (alphabet.h:80), with
 * MinHashFilter (filter.h:28-115).  Returns number of k-mers written. */
size_t kmo_extract_kmers(const char* seq, size_t len, uint32_t k, double fraction, double start_fraction,
                         int preserve_strand, uint64_t* out);
/* the same over any alphabet of alphabet.h:79-86, given by its comma-separated groups (nt "A,C,G,TU", aa "K,R,E,...", aa11_diamond, ...) */
size_t kmo_extract_kmers_alphabet(const char* seq, size_t len, uint32_t k, const char* groups, double fraction, double start_fraction,
                                  int preserve_strand, uint64_t* out);
/* sort + unique (kmer_extract.h:99-118); returns new count */
size_t kmo_sort_unique(uint64_t* kmers, size_t n);

#ifdef __cplusplus
}
#endif
#endif
