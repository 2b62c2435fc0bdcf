/*
 * kmdb_amd.h — C ABI of the MI355X-native common-k-mer counting engine.
 *
 * Drop-in boundary for the hot path of refresh-bio/kmer-db v2.3.1.  The reference has no
 * FFI layer; its operator interface for this path is the C++ class SimilarityCalculator
 * (reference src/similarity_calculator.h:4-16) called from the mode consoles.  Each entry
 * point below replaces exactly one of those call sites (plain pointers and sizes only;
 * no torch / STL types cross this boundary).  INTEGRATION.md shows the binding a kmer-db
 * maintainer would add inside the reference's consoles.
 *
 * All functions return 0 on success, non-zero on failure; kmdb_last_error() then returns
 * the message (the front-end turns it into the reference's "ERROR: <msg>" / exit(-1),
 * reference src/main.cpp:51-59).  The GPU entry points FAIL (they never fall back to a CPU
 * path) when no gfx950 device / HIP runtime is usable.
 */
#ifndef KMDB_AMD_H
#define KMDB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMDB_ABI_VERSION 7

/* ---------------------------------------------------------------------------------------
 * Host-side view of a loaded database = what the reference hands to SimilarityCalculator:
 *   PrefixKmerDb::getPatterns()      (prefix_kmer_db.h:112)   -> SoA pattern headers + bits
 *   PrefixKmerDb::getHashtables()    (prefix_kmer_db.h:108)   -> bucket table + slots
 *   AbstractKmerDb::getSamplesCount()(kmer_db.h:67)
 * Pattern fields mirror pattern_t (pattern.h:42-53) as stored on disk (pattern.cpp:15-46).
 * The view is read-only: unlike the reference (similarity_calculator.cpp:64-72) the engine
 * never mutates num_kmers.
 * ------------------------------------------------------------------------------------- */
typedef struct kmdb_db_view {
    uint32_t abi_version;          /* KMDB_ABI_VERSION */
    uint32_t kmer_length;
    uint64_t n_samples;
    uint64_t n_patterns;
    const int64_t*  num_kmers;     /* [P] on-disk values (not subtree sums) */
    const int64_t*  parent_id;     /* [P] -1 for roots; parent_id[p] < p */
    const uint32_t* num_samples;   /* [P] ids in node + ancestors */
    const uint32_t* num_local;     /* [P] ids stored in the node */
    const uint32_t* last_sample_id;/* [P] */
    const uint32_t* num_bits;      /* [P] gamma bitstream length */
    const uint64_t* data_offset;   /* [P] index (in uint64 words) of the node's stream in `data` */
    const uint64_t* data;          /* all gamma streams, MSB-first in little-endian uint64 words */
    uint64_t n_data_words;
    /* hashtables: needed by new2all only (all2all loads with SkipHashtables,
     * console_all2all.cpp:26); n_buckets may be 0. item = {u32 key; i32 val} packed in a
     * uint64 (key low), val == INT32_MAX means empty (hashmap_lp.h:71-78). */
    uint64_t n_buckets;
    const uint64_t* bucket_offset; /* [n_buckets+1] first slot of every bucket in `slots` */
    const uint64_t* slots;         /* capacity of bucket b = bucket_offset[b+1]-bucket_offset[b], a power of two */
} kmdb_db_view;

typedef struct kmdb_opts {
    uint32_t abi_version;          /* KMDB_ABI_VERSION */
    int32_t  device;               /* HIP device ordinal */
    /* pattern-stream sharding of ONE resident database: this call adds the pairs of the patterns in
     * slice `shard_index` of `shard_count` equal slices of the (DFS-ordered) pattern stream; partial
     * matrices from all slices sum (uint32, wrap-around) to the full result.  {0,1} = everything.
     * (Prefix-bucket shards, one per GPU, are made at upload: kmdb_db_upload_shard.) */
    uint32_t shard_index;
    uint32_t shard_count;
    uint32_t bubble_size;          /* all2all-sp: -bubble-size (params.h:78), kept for CLI compat; 0 = default 8000 */
    uint32_t flags;                /* KMDB_FLAG_* */
    void*    stream;               /* hipStream_t to run on, NULL = the engine's own stream */
} kmdb_opts;

#define KMDB_FLAG_FORCE_GLOBAL_ATOMICS 1u   /* A/B reference: tree-form walk, id stack in global scratch, HBM atomics (any N, any depth) */
#define KMDB_FLAG_FORCE_DIRECT         2u   /* A/B reference: tree-form walk, id stack in LDS, HBM atomics */
#define KMDB_FLAG_FORCE_TILE           4u   /* A/B reference: tree-form walk, wave-private LDS tile */
#define KMDB_FLAG_NO_FALLBACK          8u   /* fail instead of taking the HBM-atomics kernel when the block-record pipeline cannot
                                              take the database (kmdb_stats.path tells which one ran) */
#define KMDB_FLAG_ONE_SHOT             16u  /* at upload: the handle serves a process that ends after its call (the command-line front-end):
                                              the upload's host staging buffers stay mapped until kmdb_db_free / process exit instead of being
                                              unmapped by a helper thread after the first call (3.5 GB at 100 M patterns: 0.3 s of address-space
                                              work that stalls every other thread's allocations meanwhile) */
#define KMDB_FLAG_ALL                  31u  /* any other bit in kmdb_opts.flags is rejected */

/* kmdb_stats.path */
#define KMDB_PATH_NONE     0u
#define KMDB_PATH_RECORDS  1u   /* block-record pipeline (default) */
#define KMDB_PATH_TILE     2u   /* v1 LDS tile / direct kernels (forced by a flag) */
#define KMDB_PATH_GLOBAL   3u   /* v1 HBM-atomics kernel (forced, or fallback: a note goes to stderr) */

typedef struct kmdb_db kmdb_db;    /* database resident in HBM */

/* Result of the sparse calls: CSR, library-allocated, free with kmdb_sparse_free().
 * Row i lists (col, val) with val > 0, ascending col — the content of
 * SparseMatrix::data_compacted after compact2 with pass-all filters (array.h:391-446). */
typedef struct kmdb_sparse_rows {
    uint64_t  n_rows;
    uint64_t  nnz;
    uint64_t* row_ptr;             /* [n_rows+1] */
    uint32_t* col;                 /* [nnz] 0-based */
    uint32_t* val;                 /* [nnz] */
    double*   measure;             /* [nnz] or NULL: kmdb_all2all_sparse_filtered with a measure */
} kmdb_sparse_rows;

/* Similarity / distance measures of a cell (common k-mers c, k-mer counts a of the row sample and b of the column sample,
 * k-mer length k): the functions of Params::availableMetrics (params.cpp:14-42), in that order. */
#define KMDB_METRIC_JACCARD     0
#define KMDB_METRIC_MIN         1
#define KMDB_METRIC_MAX         2
#define KMDB_METRIC_COSINE      3
#define KMDB_METRIC_MASH        4
#define KMDB_METRIC_ANI         5
#define KMDB_METRIC_ANI_SHORTER 6
#define KMDB_METRIC_MASH_QUERY  7
#define KMDB_METRIC_NUM_KMERS   8
#define KMDB_METRIC_COUNT       9
/* One bound pair of a CombinedFilter (sparse_filters.h:12-61): the cell passes when lo <= metric(cell) <= hi. */
typedef struct kmdb_cell_filter {
    int32_t metric;                /* KMDB_METRIC_* */
    int32_t reserved;
    double  lo, hi;
} kmdb_cell_filter;

typedef struct kmdb_stats {        /* measurements of the LAST call on this db handle */
    double   kernel_ms;            /* HIP-event time of the whole device pipeline of the call */
    uint64_t algorithmic_bytes;    /* B_pat + 4*N(N-1)/2 (dense) */
    uint64_t tree_updates;         /* cell updates performed (tree form) */
    uint64_t sum_pairs;            /* sum_p w_p*C(n_p,2) = sum of the matrix = k-mer pair comparisons */
    uint64_t device_bytes;         /* HBM footprint of the resident db */
    uint64_t n_segments;
    uint64_t tile_flushes;
    double   k1_ms;                /* block-record pipeline: emit kernel */
    double   k2_ms;                /* block-record pipeline: apply kernel */
    uint64_t n_records;            /* block records per pass (0 when the v1 kernels ran) */
    double   k0_ms;                /* block-record pipeline: gamma decode kernels */
    double   k1n_ms;               /* ... emit, nodes with at most two blocks (DFS stream) */
    double   k1g_ms;               /* ... wide list + emit, nodes with more blocks */
    double   upload_ms;            /* wall-clock of kmdb_db_upload for this handle (host conversion + H2D + device layout) */
    uint64_t n_wide;               /* nodes with more than two blocks */
    uint64_t n_chunks;             /* record chunks = work items of the apply kernel */
    uint32_t path;                 /* KMDB_PATH_* of the last all2all call */
    uint32_t width;                /* sample ids per block */
    uint32_t sized_call;           /* 1: the last call measured its own grid sizes (first call on a handle, two host syncs more) */
    uint32_t n_joined;             /* nodes with many blocks whose records were never written: joined per tile by the second level (ABI 6; 0 when it is off) */
    uint64_t n_patterns;           /* patterns resident in HBM: all of the view's, or for kmdb_db_upload_shard only the nodes whose subtree
                                      holds a k-mer of the shard */
    uint64_t h2d_bytes;            /* bytes kmdb_db_upload[_shard] copied to the device (ABI 6) */
    uint64_t n_direct;             /* block records that were never written: first-block records (X, X) the narrow kernel applied where it emitted them,
                                      tile in registers, one write-back per slice of the pattern stream (ABI 7; n_records counts the written ones) */
} kmdb_stats;

const char* kmdb_last_error(void);
int  kmdb_abi_version(void);
/* number of usable gfx950 devices, 0 if none (never fails) */
int  kmdb_device_count(void);
/* First use of a device (runtime start, the device's context, a first allocation and stream): 0.15 - 0.3 s that depend on nothing the
 * database holds.  The front-end calls this on a helper thread while kmdbh_db_load reads the file (the reference has no counterpart:
 * console_all2all.cpp:25-29 goes straight from deserialize to the computation).  0, or 1 with kmdb_last_error() set in the calling thread. */
int  kmdb_device_prepare(int32_t device);

/* Lay the database out in HBM (replaces PrefixKmerDb living in host RAM after
 * deserialize, prefix_kmer_db.cpp:578-748).  with_hashtables != 0 also uploads the
 * bucket tables (DeserializationMode::Everything vs SkipHashtables, kmer_db.h:55-60). */
int  kmdb_db_upload(const kmdb_db_view* view, const kmdb_opts* opts, int with_hashtables, kmdb_db** out);
/* One prefix-bucket shard of the database (bucket = kmer >> 32, reference src/types.h:25-27, items
 * src/hashmap_lp.h:71-78): the pattern tree stays whole and every pattern keeps only the k-mers of the buckets b with
 * b % shard_count == shard_index, w_s[p] = #{items of those buckets with val == p}.  Partial matrices of the shards sum
 * (uint32, wrap-around) to the matrix of the whole database: one shard per GPU + one RCCL reduce.  The view must carry the
 * hashtables (load mode Everything). */
int  kmdb_db_upload_shard(const kmdb_db_view* view, const kmdb_opts* opts, int with_hashtables, uint32_t shard_index,
                          uint32_t shard_count, kmdb_db** out);
void kmdb_db_free(kmdb_db* db);
/* Waits for the handle's background housekeeping (the upload's host staging buffers are given back by a helper thread after the first
 * call).  A front-end that ends the process right after its call waits here first: the helper's threads free the pages several times
 * faster than the end of the process would. */
void kmdb_db_settle(kmdb_db* db);
int  kmdb_db_stats(const kmdb_db* db, kmdb_stats* out);
/* Why the last all2all call on this handle could not take the block-record pipeline ("" when it did, or before any call): the
 * note the engine prints once on stderr when it falls back to the HBM-atomics kernel (kmdb_stats.path == KMDB_PATH_GLOBAL), e.g.
 * a root path of more than 4096 nodes.  The pointer stays valid until the next call on the handle. */
const char* kmdb_db_fallback_reason(const kmdb_db* db);

/* Replaces SimilarityCalculator::all2all(db, LowerTriangularMatrix&)
 * (similarity_calculator.cpp:42-438; call site console_all2all.cpp:34).
 * out_lower_tri: N(N-1)/2 uint32 in HOST memory, row i at i(i-1)/2 (array.h:136-140). */
int  kmdb_all2all_dense(kmdb_db* db, uint32_t* out_lower_tri, const kmdb_opts* opts);
/* Same, result left in DEVICE memory (caller-owned, N(N-1)/2 uint32, need not be zeroed)
 * so that per-GPU partial matrices can be reduced with RCCL without a host round trip. */
int  kmdb_all2all_dense_device(kmdb_db* db, void* out_lower_tri_dev, const kmdb_opts* opts);

/* Replaces SimilarityCalculator::all2all_sp(db, SparseMatrix&, CBubbleHelper&) followed by
 * SparseMatrix::compact2 with pass-all filters (similarity_calculator.cpp:442-657,
 * array.h:391-446; call sites console_all2all_sparse.cpp:44,79). */
int  kmdb_all2all_sparse(kmdb_db* db, kmdb_sparse_rows* out, const kmdb_opts* opts);
void kmdb_sparse_free(kmdb_sparse_rows* rows);
/* the same with the -min / -max filters applied before the result leaves HBM, and optionally one measure per
 * kept cell.  Replaces all2all_sp + SparseMatrix::compact2 with a CombinedFilter (array.h:391-446, sparse_filters.h:38-61;
 * call site console_all2all_sparse.cpp:44-79) and, with measure >= 0, the `distance -sparse` pass over the written table
 * (console_distance.cpp:96-171).  sample_kmers[i] = k-mer count of sample i (the table's total-kmers row; num_kmers_t =
 * uint32 in the reference).  The device drops the cells that miss a bound by more than a safety margin; the remaining
 * cells are decided, and the measures computed, by kmdbh_metric on the host — bit-identical with the reference's double
 * arithmetic (incl. its libm log).  measure: KMDB_METRIC_* or -1. */
int  kmdb_all2all_sparse_filtered(kmdb_db* db, const kmdb_cell_filter* filters, size_t n_filters, const uint32_t* sample_kmers,
                                  int measure, kmdb_sparse_rows* out, const kmdb_opts* opts);
/* Multi-GPU all2all-sp (BASELINE configs[3]; SURVEY 8e "reduce compacted tiles"): the compaction stage of the two calls above
 * on cells the CALLER accumulated — typically the sum over prefix-bucket shards, after an RCCL reduce / reduce-scatter of the
 * partial matrices of kmdb_all2all_dense_device.  cells_dev points at cell `cell_lo` of the lower triangle (row i at
 * i (i - 1) / 2, array.h:136-140) and holds the flat range [cell_lo, cell_hi) — a rank's reduce-scatter chunk can be passed as
 * it is; the whole triangle is {0, N (N - 1) / 2}.  out has all N rows; rows outside the range are empty and a row cut by a
 * range end lists only its cells inside, so the ranks' outputs concatenate row by row (rank order = ascending columns) to
 * exactly what SparseMatrix::compact2 + saveRowSparse produce (array.h:391-446, 625-637; console_all2all_sparse.cpp:44-96).
 * Filters / measure as in kmdb_all2all_sparse_filtered (the cells are complete sums here, so prefix shards may use them). */
int  kmdb_sparse_from_dense_device(kmdb_db* db, const void* cells_dev, uint64_t cell_lo, uint64_t cell_hi,
                                   const kmdb_cell_filter* filters, size_t n_filters, const uint32_t* sample_kmers, int measure,
                                   kmdb_sparse_rows* out, const kmdb_opts* opts);
/* NOTE on streams: the call reads cells_dev on opts->stream (the handle's own stream when NULL) and does NOT order itself after
 * the caller's producer: the cells must be complete on that stream (or the device idle) before the call — after an RCCL collective
 * on another stream, wait for it first (hipStreamWaitEvent / hipStreamSynchronize). */
/* the measure itself (params.cpp:14-42): uint32 wrap-around integer parts, double arithmetic */
double kmdbh_metric(int metric, uint32_t common, uint32_t cnt_row, uint32_t cnt_col, int kmer_length);
/* KMDB_METRIC_* of a criterion name ("jaccard", "min", ..., "num-kmers"), -1 if unknown */
int    kmdbh_metric_id(const char* name);

/* Replaces T concurrent calls of SimilarityCalculator::one2all<false>
 * (similarity_calculator.cpp:809-925; call site console_new2all.cpp:82): nq queries, each a
 * sorted, duplicate-free k-mer array (KmerHelper::unique, console_new2all.cpp:73).
 * out_dense: nq x N uint32 in host memory, row q = similarities of query q. */
int  kmdb_new2all_batch(kmdb_db* db, const uint64_t* const* kmers, const size_t* counts, size_t nq,
                        uint32_t* out_dense, const kmdb_opts* opts);
/* Replaces one2all_sp (similarity_calculator.cpp:929-1051; call site console_new2all.cpp:78):
 * row q = ascending (sample_id, count) pairs with count > 0. */
int  kmdb_new2all_batch_sparse(kmdb_db* db, const uint64_t* const* kmers, const size_t* counts, size_t nq,
                               kmdb_sparse_rows* out, const kmdb_opts* opts);
/* The same with the query-side loader on the device: replaces, per query, the k-mer extraction of
 * GenomeInputFile::load / KmerHelper (kmer_extract.h:13-118), the minhash filter (filter.h:28-115) and
 * KmerHelper::unique (console_new2all.cpp:73) followed by one2all<false>.  seqs[q] = the query's sequence text
 * (ASCII, records of one sample joined by any non-ACGTU symbol, e.g. '\n'); fraction / start_fraction /
 * preserve_strand are the database's filter settings (kmdbh_db_fraction, kmdbh_db_start_fraction,
 * kmdbh_db_alphabet == 1).  out_kmer_counts[q] = number of unique k-mers of query q (the CSV's total-kmers). */
int  kmdb_new2all_batch_seq(kmdb_db* db, const char* const* seqs, const size_t* seq_lens, size_t nq, double fraction,
                            double start_fraction, int preserve_strand, uint32_t* out_dense, uint64_t* out_kmer_counts,
                            const kmdb_opts* opts);
/* The same for a database over any of the reference's alphabets (ABI 7; alphabet.h:79-126, kmer_extract.h:13-97): `alphabet` =
 * kmdbh_db_alphabet (KMDB_ALPHABET_*), e.g. protein queries against the databases of test/protein.  Records of one sample are
 * joined by any byte that is not a symbol of the alphabet ('\n'). */
int  kmdb_new2all_batch_seq_alphabet(kmdb_db* db, const char* const* seqs, const size_t* seq_lens, size_t nq, double fraction,
                                     double start_fraction, int32_t alphabet, uint32_t* out_dense, uint64_t* out_kmer_counts,
                                     const kmdb_opts* opts);

/* Replaces SimilarityCalculator::db2db_sp(db_row, db_col, SparseMatrix&, bubbles) (similarity_calculator.cpp:1225-1540),
 * the off-diagonal cell of the all2all-parts grid (call sites console_all2all_parts.cpp:180,226): both databases
 * resident with hashtables on the same device.  out: n_samples(db_row) x n_samples(db_col) uint32, row-major, host
 * memory: out[r][c] = number of k-mers shared by sample r of db_row and sample c of db_col.
 * The first call on a handle of up to 4096 samples leaves the full sample lists of its patterns with the handle (n_samples / 8
 * bytes per pattern, at most 8 GB; counted in kmdb_stats.device_bytes, freed by kmdb_db_free): the other cells of the grid that
 * use the part read them instead of rebuilding them. */
int  kmdb_db2db_dense(kmdb_db* db_row, kmdb_db* db_col, uint32_t* out, const kmdb_opts* opts);

/* ---------------------------------------------------------------------------------------
 * One database over the GPUs of a node (north_star: prefix buckets sharded across the GPUs, one RCCL reduce of the
 * partial matrices over xGMI).  Makes the reference's single call sites multi-GPU: SimilarityCalculator::all2all at
 * console_all2all.cpp:31-36 and all2all_sp + compact2 at console_all2all_sparse.cpp:44,79.
 * kmdb_node_upload: shard s of n_shards (kmdb_db_upload_shard: the k-mers of the prefix buckets b with b % n_shards == s) goes to
 * devices[s % D], D = min(n_shards, n_devices); the view must carry the hashtables when n_shards > 1.  One host thread per
 * device inside every call; several shards of one device run one after the other and are summed on the device (a one-GPU box
 * takes any n_shards that way).  With D > 1 the partial matrices meet in ONE ncclReduceScatter (uint32 sum) over flat chunks of the
 * lower triangle — xGMI is point to point, every peer pair sums its chunk over its own link — and every device brings its own
 * chunk to the host (dense) or compacts it where it is (sparse); librccl.so is loaded with dlopen only then.
 * ------------------------------------------------------------------------------------- */
typedef struct kmdb_node kmdb_node;
typedef struct kmdb_node_stats {   /* the LAST call on the node handle; maxima over the devices */
    uint32_t n_shards, n_devices;
    int32_t  rccl_version;         /* ncclGetVersion, 0 when RCCL was not needed */
    uint32_t reserved;
    double   upload_s;             /* kmdb_node_upload, wall clock */
    double   plan_s;               /* of it: the host's plan of all shards (one pass over the hashtable items, one sweep over the tree) */
    double   call_ms;              /* HIP events around the slowest device's kmdb_all2all_dense_device calls (all its shards) */
    double   collective_ms;        /* HIP events around ncclReduceScatter on the slowest device */
    double   d2h_ms;               /* dense: copy of the device's chunk to the host; sparse: compaction + copy of its CSR */
} kmdb_node_stats;
/* One device slot of the node (slot < kmdb_node_stats.n_devices): what THAT device did — an imbalanced shard shows here, not in the maxima */
typedef struct kmdb_node_device_stats {
    int32_t  device;               /* HIP device of the slot */
    uint32_t n_shards;             /* shards that live on it */
    double   upload_s;             /* its thread's share of kmdb_node_upload */
    double   call_ms;              /* HIP events around its own shards' all2all calls (last call) */
    double   collective_ms;        /* ... around its ncclReduceScatter (0 without RCCL) */
    double   d2h_ms;               /* ... around the copy of its chunk / the compaction of its chunk */
    uint64_t h2d_bytes;            /* bytes its uploads sent over PCIe (sum over its shards) */
    uint64_t n_patterns;           /* nodes resident on it (sum over its shards: a prefix shard keeps only the nodes it needs) */
    uint64_t n_records;            /* block records of its shards in the last call */
} kmdb_node_device_stats;
int  kmdb_node_device_stats_get(const kmdb_node* node, uint32_t slot, kmdb_node_device_stats* out);
int  kmdb_node_upload(const kmdb_db_view* view, uint32_t n_shards, const int32_t* devices, uint32_t n_devices, kmdb_node** out);
void kmdb_node_free(kmdb_node* node);
int  kmdb_node_stats_get(const kmdb_node* node, kmdb_node_stats* out);
/* = kmdb_all2all_dense over all shards: out_lower_tri N(N-1)/2 uint32 in host memory */
int  kmdb_node_all2all_dense(kmdb_node* node, uint32_t* out_lower_tri, const kmdb_opts* opts);
/* = kmdb_all2all_sparse_filtered over all shards (filters / measure on the complete sums; n_filters 0 and measure -1: plain
 * kmdb_all2all_sparse); rows concatenate over the devices' chunks in ascending column order */
int  kmdb_node_all2all_sparse(kmdb_node* node, const kmdb_cell_filter* filters, size_t n_filters, const uint32_t* sample_kmers, int measure,
                              kmdb_sparse_rows* out, const kmdb_opts* opts);

/* ---------------------------------------------------------------------------------------
 * Host-side helpers of the front-end (no GPU needed).  They mirror the reference's loader
 * and writer so that the CLI stays byte-compatible; exported so tests can drive them.
 * ------------------------------------------------------------------------------------- */
typedef struct kmdbh_db kmdbh_db;  /* a .db file parsed into flat host arrays */

/* PrefixKmerDb::deserialize (prefix_kmer_db.cpp:578-748). mode: 0 Everything, 2 SkipHashtables */
int  kmdbh_db_load(const char* path, int mode, kmdbh_db** out);
void kmdbh_db_free(kmdbh_db* db);
const kmdb_db_view* kmdbh_db_view(const kmdbh_db* db);
/* Once the database is on the device (kmdb_db_upload / kmdb_node_upload returned) a front-end that needs only the names and k-mer counts
 * from here on gives the pages of the pattern arrays and hashtables back (9 GB at 100 M patterns: otherwise the end of the process pays
 * for them, 0.07 s per GB).  The view's arrays must not be read afterwards; kmdbh_db_free is still due. */
void kmdbh_db_release_patterns(kmdbh_db* db);
uint32_t    kmdbh_db_kmer_length(const kmdbh_db* db);
double      kmdbh_db_fraction(const kmdbh_db* db);
double      kmdbh_db_start_fraction(const kmdbh_db* db);
int32_t     kmdbh_db_alphabet(const kmdbh_db* db);
uint64_t    kmdbh_db_n_samples(const kmdbh_db* db);
const char* kmdbh_db_sample_name(const kmdbh_db* db, uint64_t i);
uint64_t    kmdbh_db_sample_kmers(const kmdbh_db* db, uint64_t i);
uint64_t    kmdbh_db_pattern_section_bytes(const kmdbh_db* db);

/* The host's plan of the prefix shards kmdb_node_upload / kmdb_db_upload_shard work from (no GPU): for every shard s of n_shards the nodes
 * it keeps (those whose subtree holds a k-mer of a bucket b with b % n_shards == s; bucket = kmer >> 32, types.h:25-27) and the
 * k-mers it owns (items of those buckets, hashmap_lp.h:71-78).  The view must carry the hashtables. */
int  kmdbh_shard_plan_counts(const kmdb_db_view* view, uint32_t n_shards, uint64_t* kept_nodes, uint64_t* kmers);

/* KmerHelper::extract + MinHashFilter (kmer_extract.h:13-97, filter.h:28-115), nt alphabets.
 * Writes at most len k-mers to out; returns the count. */
size_t kmdbh_extract_kmers(const char* seq, size_t len, uint32_t k, double fraction, double start_fraction,
                           int preserve_strand, uint64_t* out);
/* The same over any alphabet of the reference (alphabet.h:79-86; ABI 7): `alphabet` = the database's AlphabetType as stored in the
 * file (kmdbh_db_alphabet; alphabet.h:10-18) — n-bit symbols (alphabet.h:36), k <= 64 / bits - 1 (:37), the strand preserved for
 * nt-preserve and every protein alphabet.  Returns 0 for an unknown alphabet or a k the alphabet cannot hold. */
#define KMDB_ALPHABET_NT            0
#define KMDB_ALPHABET_NT_PRESERVE   1
#define KMDB_ALPHABET_AA            2
#define KMDB_ALPHABET_AA11_DIAMOND  3
#define KMDB_ALPHABET_AA12_MMSEQS   4
#define KMDB_ALPHABET_AA6_DAYHOFF   5
#define KMDB_ALPHABET_COUNT         6
size_t kmdbh_extract_kmers_alphabet(const char* seq, size_t len, uint32_t k, int32_t alphabet, double fraction, double start_fraction,
                                    uint64_t* out);
/* Alphabet::mapping (alphabet.h:41-58): symbol code of every byte (-1: not a symbol), number of symbols, bits per symbol, strand flag.
 * 0 on success, 1 for an unknown alphabet. */
int    kmdbh_alphabet_table(int32_t alphabet, int8_t* map256, uint32_t* n_symbols, uint32_t* bits_per_symbol, int* preserve_strand);
/* KmerHelper::unique (kmer_extract.h:112-118): sort + dedupe in place, returns new count */
size_t kmdbh_sort_unique(uint64_t* kmers, size_t n);

/* CSV text (console_all2all.cpp:40-78, console_new2all.cpp:99-160, conversion.h:246-298).
 * Each returns bytes written to `out` (caller sizes it: 10000 + 100*N like the reference). */
size_t kmdbh_format_header(const kmdbh_db* db, char* out, size_t cap);
size_t kmdbh_format_dense_row(const char* name, uint64_t kmers, const uint32_t* row, size_t n, char* out);
size_t kmdbh_format_sparse_row(const char* name, uint64_t kmers, const uint32_t* cols, const uint32_t* vals, size_t n, char* out);

#ifdef __cplusplus
}
#endif
#endif /* KMDB_AMD_H */
