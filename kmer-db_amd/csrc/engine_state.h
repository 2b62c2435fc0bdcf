// engine_state.h — the HBM-resident database shared by the translation units of libkmdb_amd.so
// (engine.hip: layout / entry points, a2a_v1.hip: scatter kernels, a2a_records.hip: block-record pipeline).
#pragma once
#include "kmdb_amd.h"
#include "kmdb_internal.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct Segment { uint32_t first, end; };

struct kmdb_db {
    int device = 0;
    uint64_t N = 0, P = 0;
    uint4* meta = nullptr;          // {n, l, last_id, nbits} per node, DFS order
    uint64_t* bitpos = nullptr;     // absolute bit offset of the node's gamma stream
    int32_t* parent = nullptr;      // DFS index of the parent, -1 for roots
    uint32_t* w = nullptr;          // on-disk num_kmers truncated to u32, P+1 entries (last = 0)
    uint32_t* sub_end = nullptr;    // DFS index one past the node's subtree
    uint32_t* wprefix = nullptr;    // P+1, exclusive scan of w (recomputed by every call)
    uint64_t* bits = nullptr;
    uint64_t n_bit_words = 0;
    Segment* segs = nullptr;            // equal-COST slices (tree-form updates) for the v1 scatter kernels
    uint32_t n_segs = 0;
    Segment* rsegs = nullptr;           // equal-NODE-COUNT slices for the block-record emit kernels
    uint32_t n_rsegs = 0;
    void* scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    uint32_t* stack_scratch = nullptr;  // global kernel: per-wave id stacks
    size_t stack_scratch_words = 0;
    unsigned long long* counters = nullptr;   // [0] tile flushes
    // hashtables (new2all)
    uint64_t n_buckets = 0;
    uint64_t* bucket_offset = nullptr;
    uint64_t* slots = nullptr;
    uint32_t* pid2dfs = nullptr;    // original pattern id -> DFS index
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    kmdb_stats stats{};
    uint32_t kmer_length = 0;
    // v2 (block record) pipeline state, built at upload when the database qualifies
    bool b2_ready = false;
    uint32_t b2_maxn_pad = 0, b2_dec_cap = 0, b2_nctr = 0, b2_n_items = 0;
    uint32_t b2_width = 64;             // sample ids per block (<= 64), chosen at upload
    uint32_t* b2_table = nullptr;       // [n_segs][nctr] record bases
    unsigned long long* b2_rec_rows = nullptr;   // [total]
    ulonglong2* b2_rec_rc = nullptr;    // [total]
    uint32_t* b2_rec_w = nullptr;       // [total]
    void* b2_items = nullptr;           // B2Item[n_items], dearest weight class first
    uint64_t b2_total = 0;
    hipEvent_t ev_k2[2] = {nullptr, nullptr};
    double k1_ms = 0, k2_ms = 0;
    // v3 front half (K0 decode + batch-parallel K1)
    bool b3_ready = false;
    uint32_t b3_nbw = 0;
    uint32_t* b3_perm = nullptr;        // the nodes with long local lists, longest first
    uint32_t b3_n_long = 0, b3_short_max = 32, b3_chain_cap = 64;
    uint32_t* b3_nl = nullptr;          // n | l << 16 per node
    uint32_t* b3_pair_ofs = nullptr;    // [P+1] CSR of the pairs beyond a node's first
    unsigned long long* b3_p0_mask = nullptr;   // [P] first pair, inline
    uint16_t* b3_p0_info = nullptr;     // [P] block | npairs << 8
    uint8_t* b3_pair_blk = nullptr;
    unsigned long long* b3_pair_mask = nullptr;
    uint32_t* b3_seg_anc = nullptr;     // [n_segs][B3_CHAIN]
    uint32_t* b3_seg_anc_n = nullptr;
    uint64_t b3_total_pairs = 0;
    // K1 split: narrow nodes (full list inside one block) stay in the DFS stream, wide nodes get their own list
    bool b3_split = false;
    uint8_t* b3_depth = nullptr;        // [P] root path length (root = 1)
    uint2* b3_k0in = nullptr;           // [P] decode kernel input: {l | last id << 16, stream bits}
    uint32_t* b3_bitrel = nullptr;      // [P] stream position relative to b3_blkbase[i / 256]
    uint64_t* b3_blkbase = nullptr;
    uint32_t* b3_widx = nullptr;        // [n_wide] DFS index of the wide nodes, DFS order
    int32_t* b3_wparent = nullptr;      // [n_wide] >= 0: position of the (wide) parent in the wide list; -1: none; <= -2: narrow parent -(v + 2)
    ulonglong2* b3_fnarrow = nullptr;   // [P] (F0, F1) of the <= 2-block nodes that have a wider child (written by the narrow kernel)
    uint16_t* b3_wd01 = nullptr;        // [P] blocks of a <= 2-block node: first | second << 8 (0xFF: none)
    Segment* b3_wsegs = nullptr;        // slices of the wide list
    uint32_t* b3_wseg_anc = nullptr;    // [n_wsegs][chain_cap] wide ancestors (DFS index) of the slice's first node, root first
    uint32_t* b3_wseg_anc_n = nullptr;
    int32_t* b3_wseg_np = nullptr;      // [n_wsegs] narrow parent (DFS index) of the topmost wide ancestor, -1: none
    uint32_t b3_n_wide = 0, b3_n_wsegs = 0, b3_wchain_cap = 8;
    Segment* b3_nsegs = nullptr;        // equal-node-count slices of the DFS stream for the narrow kernel
    uint32_t* b3_nseg_anc = nullptr;    // [n_nsegs][chain_cap] root path of the slice's first node
    uint32_t* b3_nseg_anc_n = nullptr;
    uint32_t* b3_ntable = nullptr;      // [n_nsegs][blocks * classes] record bases of the narrow kernel (diagonal buckets)
    uint32_t b3_n_nsegs = 0;
    hipEvent_t ev_k0 = nullptr;
    double k0_ms = 0;
};


// K0 decodes a node in the coalesced DFS-order launch when its stream is short enough to sit in three
// registers; the others (a few percent) go to a second launch, longest list first.
constexpr uint32_t KMDB_SHORT_MAX_IDS = 32, KMDB_SHORT_MAX_BITS = 128;
__host__ __device__ inline bool kmdb_long_node(uint32_t l, uint32_t num_bits) { return l > KMDB_SHORT_MAX_IDS || num_bits > KMDB_SHORT_MAX_BITS; }
constexpr int KMDB_CHAIN_MAX = 192;   // longest root path (in nodes) the batch-parallel emit kernel can hold (slot ids are bytes)

// ---- a2a_v1.hip: tree-form scatter kernels (LDS tile / HBM atomics); M is zeroed, wprefix is scanned
int kmdb_v1_run(kmdb_db* db, uint32_t* M, uint32_t seg_begin, uint32_t seg_end, uint32_t flags, hipStream_t st);

// ---- a2a_records.hip: block-record pipeline
// upload-time: qualify the database, pick the block width, tabulate record counts (count modes of the kernels)
struct kmdb_host_layout {               // host copies of upload-time arrays the preparation needs
    uint32_t max_n;
    bool chain_ok;
    const std::vector<uint32_t>* long_nodes;
    const std::vector<uint32_t>* nl;
    const std::vector<uint32_t>* seg_anc;
    const std::vector<uint32_t>* seg_anc_n;
    const std::vector<int32_t>* parent;
    const std::vector<uint16_t>* depth;
    const std::vector<uint4>* meta;
    const std::vector<uint64_t>* bitpos;
    const std::vector<Segment>* nsegs;            // narrow kernel slices + root paths
    const std::vector<uint32_t>* nseg_anc;
    const std::vector<uint32_t>* nseg_anc_n;
    const std::vector<uint32_t>* w;               // on-disk weights, DFS order
};
int kmdb_records_prepare(kmdb_db* db, const kmdb_host_layout& h);
// per call: decode + emit (+ sequential emit fallback) + apply; records events ev_k0 / ev_k2
int kmdb_records_run(kmdb_db* db, uint32_t* M, uint32_t flags, hipStream_t st);
void kmdb_records_release(kmdb_db* db);
uint64_t kmdb_records_device_bytes(const kmdb_db* db);
