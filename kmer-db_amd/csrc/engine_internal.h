// engine_internal.h — lets the other translation units of libkmdb_amd.so (new2all.hip) reach
// the HBM-resident database that engine.hip owns.  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct kmdb_db;

constexpr uint32_t KMDB_ID_BITS = 20, KMDB_MAX_SAMPLES = 1u << KMDB_ID_BITS;   // sample ids in the device layout (engine_state.h: kmdb_k0_pack)
constexpr uint32_t KMDB_CK_IDS = 32;   // list index of new2all: one checkpoint per 32 local ids of a long list

struct kmdb_engine_view {
    int device;
    uint64_t N, P;
    uint32_t kmer_length;
    const uint4* meta;
    const uint64_t* bitpos;
    const int32_t* parent;
    const uint32_t* w;
    const uint32_t* sub_end;
    const uint64_t* bits;
    const uint32_t* ck_ofs;        // list index of the long local lists (engine_state.h)
    const uint64_t* ck_bit;
    const uint32_t* ck_id;
    uint64_t n_buckets;
    const uint64_t* bucket_offset;
    const uint64_t* slots;
    const uint32_t* pid2dfs;
    uint32_t max_depth;            // nodes on the longest root path
    // list store of db2db.hip, kept with the handle: the full sample list of every pattern as a bit set of list_sets_nb words
    // (built on the first db2db call that can afford it, freed by kmdb_db_free)
    unsigned long long** list_sets;
    uint32_t* list_sets_nb;
    bool* list_sets_tried;
    // run index of new2all.hip, kept with the handle: node i's local ids as runs rl_runs[rl_ofs[i] .. rl_ofs[i + 1]) (start | length << rs, n2a_run_shift)
    uint32_t** rl_ofs;
    uint32_t** rl_runs;
    uint4** rl_node;               // and the walk's 16-byte node records {subtree end, parent, first run or the only id, min(l, 65535) | min(runs, 65535) << 16}
    bool* rl_tried;
    uint64_t* device_bytes;
    void* stream;
    void* ev[4];
};

// also builds the v1 / new2all node arrays on first use; non-zero on failure
int kmdb_engine_get(kmdb_db* db, kmdb_engine_view* out);
void kmdb_engine_set_times(kmdb_db* db, double kernel_ms, double dominant_ms);

// sort + matrix-core accumulation of block records into a dense n_rows x n_cols matrix (a2a_blocks.hip; used by db2db.hip).
// Record = 16 bytes {row mask, column mask} + key word {stream = row block * nbc + column block | (weight digit | digit index << d) << key_bits},
// d = 32 - key_bits - 2; slots never written carry the key 0xFFFFFFFF.
int kmdb_rect_sort_apply(hipStream_t st, uint32_t* wkey, void* wrec, uint32_t nslots, uint32_t nbr, uint32_t nbc, int key_bits, uint32_t* M, uint32_t n_rows,
                         uint32_t n_cols);
