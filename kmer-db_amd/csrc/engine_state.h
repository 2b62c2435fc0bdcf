// engine_state.h — the HBM-resident database shared by the translation units of libkmdb_amd.so
// (engine.hip: layout / entry points, a2a_v1.hip: scatter kernels, a2a_blocks.hip: block-record pipeline).
#pragma once
#include "kmdb_amd.h"
#include "kmdb_internal.h"

#include <thread>
#include "engine_internal.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct Segment { uint32_t first, end; };

// counters of the block-record pipeline, zeroed at the start of every call and read back at its end
enum : uint32_t {
    KCTR_CHUNKS = 0,        // record chunks opened
    KCTR_POOL_OVERFLOW,     // != 0: the chunk pool was too small (results invalid; the call is repeated with a larger pool)
    KCTR_PAIR_OVERFLOW,     // != 0: the pool of extra (block, mask) pairs was too small
    KCTR_NWIDE,             // nodes whose full list touches more than two blocks
    KCTR_LIST_OVERFLOW,     // != 0: a wide node's path did not fit the per-wave entry pool
    KCTR_RAW = 5,           // chunks of the wide pool in use (upper bound: busiest sub-pool x sub-pools)
    KCTR_RECORDS = 6,       // low word of the number of block records written (64-bit atomic: even index)
    KCTR_RECORDS_HI = 7,
    KCTR_SLOW = 8,          // wide nodes that took the climbing path
    KCTR_WIDE_OVERFLOW = 9, // != 0: the wide record pool was too small
    KCTR_WIDE_RECORDS = 10, // records written to the wide pool
    KCTR_ROWJOBS = 11,      // row mode: workgroups of the sort inside the block rows
    KCTR_L2_OVERFLOW = 12,  // != 0: the second level's node indices or entries ran out (results invalid; repeated with larger arrays)
    KCTR_L2_NODES = 13,     // nodes that went to the second level
    KCTR_POOL_USED = 15,    // chunks of the chunk pool in use (upper bound: busiest sub-pool x sub-pools)
    KCTR_K2JOBS = 14,       // row mode: jobs of the apply kernel over the sorted records (one stream, or one part of a long stream, each)
    KCTR_DIRECT = 16,       // first-block records the narrow kernel applied where it emitted them (never written: k1n_kernel<true>)
    KCTR_COUNT = 17
};
constexpr uint32_t KMDB_PAIR_REGIONS = 4096;
constexpr uint32_t KMDB_SUBPOOLS = 256;      // sub-pools of the record chunk pool (one allocation cursor each)   // sub-pools of the extra-pair pool (one allocation cursor each)

struct kmdb_db {
    int device = 0;
    uint64_t N = 0, P = 0;
    uint32_t kmer_length = 0;
    // ---- structural layout (upload): a pure format conversion of the on-disk pattern section, DFS pre-order.
    // Nothing here depends on a decoded sample id.
    uint2* k0in = nullptr;          // [P] local-list head of the node: l, last id, stream bits (packed: kmdb_k0_pack)
    uint32_t* bitrel = nullptr;     // [P] stream position relative to blkbase[i / 256]
    uint64_t* blkbase = nullptr;    // [P / 256]
    uint64_t* bits = nullptr;       // gamma streams bit-packed back to back
    uint64_t n_bit_words = 0;
    uint32_t short_max_ids = 48;                    // (= KMDB_SHORT_MAX_IDS) local lists of more ids (or of more than KMDB_SHORT_MAX_BITS stream bits) are decoded by the long launch
    uint32_t* nl = nullptr;         // [P] n = ids of the node's full list
    int32_t* parent = nullptr;      // [P] DFS index of the parent, -1 for roots
    uint32_t* w = nullptr;          // [P+1] on-disk num_kmers truncated to u32 (last = 0); a prefix shard keeps only its own k-mers
    uint16_t* dflag = nullptr;      // [P] root path length (root = 1) | has-child << 15
    uint32_t* sub_end = nullptr;    // [P] DFS index one past the node's subtree
    uint32_t* long_nodes = nullptr; // nodes whose stream does not fit the short decoder, most work first
    uint32_t n_long = 0;
    uint32_t nseg_nodes = 2048;     // nodes per slice of the DFS stream (one wave each)
    uint32_t n_nsegs = 0;
    uint32_t* nseg_anc = nullptr;   // [n_nsegs][chain_cap] root path of every slice's first node
    uint32_t* nseg_anc_n = nullptr;
    uint32_t chain_cap = 8;         // chain slots per wave = longest root path, rounded up
    uint32_t max_depth = 0, max_n = 0;
    // db2db's list store (db2db.hip): full sample list of every pattern as a bit set, list_sets_nb words per pattern
    unsigned long long* list_sets = nullptr;
    uint32_t list_sets_nb = 0;
    bool list_sets_tried = false;
    // new2all's run index (new2all.hip): the local list of every node as runs of consecutive ids (start | length << rs), built on the
    // first new2all call of the handle
    uint32_t* rl_ofs = nullptr;     // [P + 1]
    uint32_t* rl_runs = nullptr;
    uint4* rl_node = nullptr;       // [P] the node as the walk reads it, one 16-byte load: subtree end, parent, first run (or the id of a one-id list), l | runs << 16
    bool rl_tried = false;
    bool chain_ok = false;          // root paths fit the chain table of the emit kernel
    // ---- per-call working set of the block-record pipeline (contents rebuilt by every call)
    uint32_t width = 64;            // sample ids per block, picked at upload from a sampled estimate
    uint32_t NB = 0, n_states = 0;  // blocks, (bucket, weight class) streams
    unsigned long long* p0_mask = nullptr;   // [P] first (block, mask) pair of every node's local list
    uint32_t* p0_info = nullptr;    // [P] block | npairs << 16
    uint32_t* pair_ofs = nullptr;   // [P] first extra pair (valid when npairs > 1)
    uint16_t* pair_blk = nullptr;   // extra pairs, KMDB_PAIR_REGIONS sub-pools
    unsigned long long* pair_mask = nullptr;
    uint64_t pair_cap = 0;          // entries in total
    uint32_t* pair_cursor = nullptr;        // [KMDB_PAIR_REGIONS * 16] one cursor per region, a cache line apart
    ulonglong2* fn_mask = nullptr;  // [P] for a node with more than two blocks whose parent has at most two: the parent's (F0, F1)
    uint32_t* fn_blk = nullptr;     // [P] and its blocks: first | second << 16 (0xFFFF: none)
    unsigned long long* widebits = nullptr;  // [ceil(P / 64)] nodes with more than two blocks
    uint32_t* wide_cnt = nullptr;   // [words + 1] popcounts / their exclusive scan
    uint32_t* wide_base = nullptr;
    uint32_t* widx = nullptr;       // [wide_cap] the wide nodes, DFS order
    uint32_t* wrun_anc = nullptr;   // [runs][chain_cap] root path of the first node of every run of the wide-node kernel (built by every call)
    uint32_t* wrun_anc_n = nullptr;
    uint64_t wide_cap = 0;
    uint32_t* chunk_key = nullptr;  // [pool_cap] stream of every chunk (n_states: never opened)
    uint32_t* chunk_fill = nullptr; // [pool_cap] records in the chunk
    uint32_t* sorted_key = nullptr; // the chunk table sorted by stream
    uint32_t* sorted_id = nullptr;
    uint32_t* chunk_iota = nullptr; // 0, 1, 2, ...
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int key_bits = 1;
    unsigned char* rec = nullptr;   // [pool_cap * 64] 16-byte record slots ({rows, cols}; diagonal streams pack 8-byte rows)
    uint32_t* recw = nullptr;       // [pool_cap * 64] weights of the classes with w > 1
    uint64_t pool_cap = 0;          // chunks of 64 records, KMDB_SUBPOOLS interleaved sub-pools
    uint32_t* sub_cursor = nullptr; // [KMDB_SUBPOOLS * 16]
    // wide pool: records in arrival order + a device-wide sort by stream (the wide kernel always; the narrow kernel if its
    // per-stream chunks do not work out)
    bool dense_wide = false, dense_narrow = false;
    // the narrow kernel's first-block records (X, X): 0 stream chunks + k2_apply_kernel (round 5), 1 applied inside the narrow kernel (tile in its
    // registers, nothing written), 2 compacted per slice in DFS order + k2d_kernel on the side stream (default).  1 and 2: what they cannot take
    // (weights of 128 and more) goes the dense_narrow way, nothing uses stream chunks
    int k1n_mode = 2;
    unsigned long long* dmask = nullptr;    // mode 2: [n_nsegs * nseg_nodes] F0 of the slice's records
    uint32_t* dwx = nullptr;                //         w | X << 8
    uint32_t* slice_cnt = nullptr;          //         [n_nsegs] records of every slice
    uint32_t* direct_ctr = nullptr; // [KMDB_SUBPOOLS * 16] records applied directly, counted per wave on spread addresses
    uint64_t last_n_direct = 0;
    uint64_t blocks_bytes_counted = 0;   // the block-record pipeline's share of stats.device_bytes as last counted (its arrays grow inside calls)
    uint32_t *wkey = nullptr, *swkey = nullptr;     // wide pool: stream of every record slot (0xFFFFFFFF = never written), and sorted
    void *wrec = nullptr, *swrec = nullptr;         // wide pool: 16-byte records {rows, cols} (weight digit in the key word), and sorted by stream
    uint64_t wide_pool_cap = 0;     // chunks of 64 records
    uint32_t* wsub_cursor = nullptr;
    uint32_t* run_ctr = nullptr;    // wide-node kernel: next run of every class
    // many streams ("row mode"): the wide records go to per-block-row chunks of the chunk pool; the chunk table grouped by key, then a
    // counting sort inside every row
    bool row_mode = false;
    uint32_t rec_pshift = 0;        // many streams, block width <= 54: the weight digit of a block record lives in the spare bits of its column mask (from
                                    // this bit on) and the record travels as 16 bytes; 0: 16 bytes of masks + the 4-byte key word
    uint32_t n_ckeys = 0;                              // keys of the chunk table: streams (+ block rows in row mode); n_ckeys = never opened
    uint32_t *ct_hist = nullptr, *ct_offs = nullptr, *ct_cursor = nullptr;   // [n_ckeys + 2]
    void* ct_tmp = nullptr;
    size_t ct_tmp_bytes = 0;
    uint32_t *rg_hist = nullptr, *rg_offs = nullptr;   // row chunks grouped by row: [NB][rg_blocks] counts / offsets (+ total)
    uint32_t* row_ids = nullptr;                       // the row chunks' ids, grouped by row
    uint32_t rg_blocks = 0;
    void* rg_tmp = nullptr;
    size_t rg_tmp_bytes = 0;
    uint32_t* rs_rows = nullptr;                       // [2][NB + 1] first job / first table entry of every row
    uint32_t *rs_hist = nullptr, *rs_offs = nullptr;   // [rs_entries] per row [stream][job] counts / offsets
    uint64_t rs_entries = 0;
    void* rs_tmp = nullptr;
    size_t rs_tmp_bytes = 0;
    uint64_t sorted_cap = 0;                           // records the sorted arrays (swkey / swrec) hold
    uint32_t last_n_rowjobs = 0, last_n_sorted = 0, last_n_k2jobs = 0, k1w_waves = 0, k1w_slots = 0;   // (k1w_waves: most the pools are sized for; k1w_slots: waves the chip holds at once)
    uint32_t* cs_rows = nullptr;                       // two-pass sort: row starts / first workgroup / first table entry, [3][NB + 1]
    uint32_t *cs_hist = nullptr, *cs_offs = nullptr;   // counting sort of the wide pool: [stream][block] counts / offsets (+ total)
    void* cs_tmp = nullptr;
    size_t cs_tmp_bytes = 0;
    void* sort2_tmp = nullptr;
    size_t sort2_tmp_bytes = 0;
    uint32_t* counters = nullptr;   // [KCTR_COUNT]
    uint32_t* h_counters = nullptr; // pinned host copy
    // ---- second level above the block records (a2a_blocks.hip, L2View): the nodes with many blocks write (node, block, mask) entries instead of
    // their c (c + 1) / 2 records; a tile job joins two blocks' lists and applies the matches.  Row mode with few enough blocks only.
    bool l2_on = false;
    uint32_t l2_min_blocks = 24, l2_node_cap = 0, l2_ent_cap = 0, last_l2_nodes = 0;
    uint32_t* l2_cursors = nullptr;                    // [32 * 16]
    unsigned long long* l2_bitmap = nullptr;           // [NB][l2_node_cap / 64]
    uint32_t *l2_rank = nullptr, *l2_len = nullptr, *l2_loff = nullptr;   // [NB][W] rank directory, [NB] list lengths, [NB + 1] list offsets
    uint32_t *l2_ent_g = nullptr, *l2_node_w = nullptr, *l2_list_w = nullptr;
    uint16_t* l2_ent_blk = nullptr;
    unsigned long long *l2_ent_mask = nullptr, *l2_list_mask = nullptr;
    uint64_t est_records = 0;       // sampled estimate for the chosen width
    // what the previous call found (the pipeline is deterministic per database: grid sizes of the next call)
    bool have_counts = false;
    bool last_call_sized = false;   // the last call measured launch sizes (first call on a handle / a new emit range: extra host syncs)
    uint32_t last_n_wide = 0, last_n_chunks = 0, last_n_raw = 0, last_n_slow = 0;
    uint64_t last_records = 0;
    uint32_t last_emit_lo = 0, last_emit_hi = 0;
    uint32_t n_slices = 1;          // passes over the pattern stream per call (a database whose records do not fit one)
    struct SliceCounts { bool valid = false; uint32_t lo = 0, hi = 0, n_wide = 0, n_chunks = 0, n_raw = 0, n_rowjobs = 0, n_sorted = 0, n_k2jobs = 0; };
    std::vector<SliceCounts> slice_counts;   // what the previous call found, per slice of the pattern stream
    void* scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    // ---- v1 kernels (A/B reference, fallback) and new2all: built lazily on the device from the arrays above
    // host staging buffers of the upload, given back by a helper thread after the first call (or when the handle is freed):
    // unmapping them costs 0.3 s (the HIP runtime had them registered for the copies) and blocks every hipMalloc meanwhile
    std::vector<std::pair<void*, size_t>> staging;
    std::vector<std::pair<void*, size_t>> staging_kept;   // one-shot handle: regions whose pages were dropped but that stay mapped until kmdb_db_free
    std::thread staging_thread;    // gives the staging buffers back (kmdb_release_staging); joined by kmdb_db_settle / kmdb_db_free
    bool one_shot = false;         // KMDB_FLAG_ONE_SHOT at upload: the staging buffers stay until the handle is freed
    bool v1_ready = false;          // the arrays below exist (all of them)
    uint4* meta = nullptr;          // {n, l, last_id, nbits} per node, DFS order
    uint64_t* bitpos = nullptr;     // absolute bit offset of the node's gamma stream
    // new2all: index into the gamma streams of the nodes with more than KMDB_CK_IDS local ids — every KMDB_CK_IDS-th id and
    // the bit position of the code after it, so that the lanes of a workgroup decode one long list in pieces
    uint32_t* ck_ofs = nullptr;     // [P + 1] first checkpoint of the node (a node with a short list has none)
    uint64_t* ck_bit = nullptr;
    uint32_t* ck_id = nullptr;
    uint32_t* wprefix = nullptr;    // P+1, exclusive scan of w (recomputed by every call)
    Segment* segs = nullptr;
    uint32_t n_segs = 0;
    void* v1_scan_tmp = nullptr;
    size_t v1_scan_tmp_bytes = 0;
    uint32_t* stack_scratch = nullptr;  // global kernel: per-wave id stacks
    size_t stack_scratch_words = 0;
    unsigned long long* v1_counters = nullptr;   // [0] tile flushes
    // hashtables (new2all)
    uint64_t n_buckets = 0;
    uint64_t* bucket_offset = nullptr;
    uint64_t* slots = nullptr;
    uint32_t* pid2dfs = nullptr;    // original pattern id -> DFS index
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // side stream: the stream chunks are sorted and applied next to the wide kernel
    hipEvent_t ev_side[2] = {nullptr, nullptr};
    unsigned char* tile_touched = nullptr;   // [n_states] != 0: the last call's apply kernels added something to the tile of that block pair (the sparse
                                             // entry point scans only those tiles)
    uint32_t* k2j_start = nullptr;  // [n_states + 1] many streams: where every stream starts in the sorted arrays
    uint2* k2j_jobs = nullptr;      // [k2j_cap] jobs of the apply kernel: {stream, part of K2J_REC records}
    uint64_t k2j_cap = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_k[4] = {nullptr, nullptr, nullptr, nullptr};   // after decode / narrow / wide / apply
    kmdb_stats stats{};
    bool blocks_prepared = false;   // width estimate + working set of the block-record pipeline exist (made at upload for all2all
                                    // uploads, on the first all2all call for uploads that carry hashtables: new2all / db2db use)
    std::string fallback_reason;    // why the block-record pipeline cannot take this database ("" = it can)
};

// K0 decodes a node in the coalesced DFS-order launch when its stream is short enough to sit in three
// registers; the others (a few percent) go to a second launch, longest list first.
// Head of a node's local list as K0 reads it, 8 bytes: l and the last id in 20 bits each, the length of the gamma stream in 24 (l ids below
// 2^20 need fewer than 1.6 * 2^20 stream bits: a delta of 2 costs 3 bits, the dearest per unit of id range).  Sample ids are therefore
// below KMDB_MAX_SAMPLES; the upload checks all three fields.
constexpr uint32_t KMDB_MAX_STREAM_BITS = 1u << 24;            // (KMDB_ID_BITS, KMDB_MAX_SAMPLES: engine_internal.h)
__host__ __device__ inline uint2 kmdb_k0_pack(uint32_t l, uint32_t last, uint32_t nbits) {
    uint2 r;
    r.x = l | ((nbits >> 12) << KMDB_ID_BITS); r.y = last | ((nbits & 0xFFFu) << KMDB_ID_BITS);
    return r;
}
__host__ __device__ inline uint32_t kmdb_k0_l(uint2 km) { return km.x & (KMDB_MAX_SAMPLES - 1u); }
__host__ __device__ inline uint32_t kmdb_k0_last(uint2 km) { return km.y & (KMDB_MAX_SAMPLES - 1u); }
__host__ __device__ inline uint32_t kmdb_k0_bits(uint2 km) { return ((km.x >> KMDB_ID_BITS) << 12) | (km.y >> KMDB_ID_BITS); }
constexpr uint32_t KMDB_SHORT_MAX_IDS = 48, KMDB_SHORT_MAX_BITS = 128;     // (ids: the default of kmdb_db.short_max_ids, KMDB_SHORT_IDS at upload.  Measured, profiles/r05_j4: 32 -> 48 ids takes 0.14 ms off the decode at c2 and 0.5 ms at 10 000 samples — the lists of a clade of 50 fit —, 56 and 64 lose: the short launch walks a list that spans 64 ids or more twice)
__host__ __device__ inline bool kmdb_long_node(uint32_t l, uint32_t num_bits, uint32_t max_ids) { return l > max_ids || num_bits > KMDB_SHORT_MAX_BITS; }
constexpr int KMDB_CHAIN_MAX = 4096;  // longest root path (in nodes) the chain table of the narrow kernel holds (20 B of LDS per node and wave:
                                      // the deeper the tree, the fewer waves share a workgroup)

// ---- layout.hip: host conversion + device layout of the view (fills the structural arrays and stats)
int kmdb_layout_upload(kmdb_db* db, const kmdb_db_view* v, int with_hashtables, uint32_t shard_index, uint32_t shard_count, kmdb_shard_plan* plan);
// kmdb_db_upload_shard with a plan the caller made for several shards at once (node.hip); plan == nullptr: the shard is planned by itself
int kmdb_db_upload_planned(const kmdb_db_view* v, const kmdb_opts* opts, int with_hashtables, uint32_t shard_index, uint32_t shard_count, kmdb_shard_plan* plan,
                           kmdb_db** out);

// ---- a2a_v1.hip: tree-form scatter kernels (LDS tile / HBM atomics); M is zeroed, wprefix is scanned
int kmdb_v1_run(kmdb_db* db, uint32_t* M, uint32_t seg_begin, uint32_t seg_end, uint32_t flags, hipStream_t st);

// ---- a2a_blocks.hip: block-record pipeline
// upload-time: block width from a sampled estimate, working-set allocation
int kmdb_blocks_prepare(kmdb_db* db);
// per call: decode, narrow / wide emit, apply — everything that depends on a sample id happens here
int kmdb_blocks_run(kmdb_db* db, uint32_t* M, uint32_t emit_lo, uint32_t emit_hi, hipStream_t st);
void kmdb_blocks_release(kmdb_db* db);
uint64_t kmdb_blocks_device_bytes(const kmdb_db* db);
// v1 / new2all node arrays, derived on the device from the compact layout
int kmdb_ensure_v1_arrays(kmdb_db* db);

// layout.hip: unmap db->staging piece by piece on a detached thread
void kmdb_release_staging(kmdb_db* db);
