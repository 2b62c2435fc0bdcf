// kmdb_internal.h — shared by the translation units of libkmdb_amd.so (not installed).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

// records the message for kmdb_last_error() and returns a non-zero status
int kmdb_set_error(const std::string& msg);

// Gives the pages inside the regions back to the kernel, on up to `threads` threads (host_db.cpp).  madvise(MADV_DONTNEED) takes the
// address-space lock SHARED: the threads, and the page faults and allocations of every other thread, go on side by side, where a munmap
// of gigabytes holds the lock exclusively for as long as it frees pages.  The regions stay mapped (they read as zeros afterwards):
// unmapping them later, or the end of the process, finds nothing left to free.
void kmdb_drop_pages(const std::vector<std::pair<void*, size_t>>& regions, unsigned threads);

// ---- host_shards.cpp: the prefix shards of one database, planned on the host in ONE pass over its hashtables and ONE sweep over its
// tree, for all shards at once (bucket = kmer >> 32, reference src/types.h:25-27; items src/hashmap_lp.h:71-78).
// Shard s owns the k-mers of the buckets b with b % n_shards == s.  w[s][p] = k-mers of pattern p in shard s; a node is kept by shard s
// when its subtree holds a k-mer of s (bit s & 7 of mask[s >> 3][p]).  Every device then receives only the nodes and streams its
// shards keep — not the whole tree and the hashtables once per device.
struct kmdb_db_view;
// prefix shards per database at most (the plan keeps a counter array per shard; a count beyond this is a caller's mistake, not a configuration)
constexpr uint32_t KMDB_MAX_SHARDS = 4096;
struct kmdb_shard_plan {
    uint64_t P = 0;
    uint32_t n_shards = 0;
    std::vector<uint32_t*> w;                  // [n_shards] -> [P]; nullptr for a shard that was not asked for (or was released)
    std::vector<unsigned char*> mask;          // [(n_shards + 7) / 8] -> [P]
    std::vector<uint64_t> kept;                // [n_shards] nodes the shard keeps
    kmdb_shard_plan() = default;
    kmdb_shard_plan(const kmdb_shard_plan&) = delete;
    kmdb_shard_plan& operator=(const kmdb_shard_plan&) = delete;
    ~kmdb_shard_plan();
    void release_weights(uint32_t shard);      // the shard has been uploaded: its P counters go back
    bool keeps(uint32_t shard, uint64_t p) const { return (mask[shard >> 3][p] >> (shard & 7u)) & 1u; }
};
// plans the listed shards (all of them: kmdb_node_upload; one: kmdb_db_upload_shard); 0, or 1 with the error set
int kmdb_shard_plan_build(const kmdb_db_view* v, uint32_t n_shards, const std::vector<uint32_t>& shards, kmdb_shard_plan* plan);
