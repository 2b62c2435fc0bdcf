// db2db.hip — two-database counting behind kmdb_db2db_dense (include/kmdb_amd.h).
//
// Replaces SimilarityCalculator::db2db_sp (reference src/similarity_calculator.cpp:1225-1540), the cell (row part,
// column part) of the all2all-parts grid (src/console_all2all_parts.cpp:143-331).  The reference merge-joins the
// sorted (suffix, pattern) lists of every k-mer prefix of the two databases, counts the equal (pattern_row,
// pattern_col) pairs and adds the count to every pair of samples of the two patterns.  Here:
//   probe    one thread per hashtable slot of the COLUMN database: its k-mer is looked up in the row database's
//            table of the same prefix (murmur3 fmix32 probe, src/hashmap_lp.h:53-64,308-333);
//            key = DFS index of the row pattern << cbits | DFS index of the column pattern (cbits: as many as the column
//            database's patterns need), or ~0
//   count    radix sort (of the bits in use only) + run-length encode of the keys: (pattern pair, number of shared k-mers)
//   lists    once per handle (kept with it: every cell of an all2all-parts grid uses a part many times): the FULL sample list of
//            every pattern as a bit set over blocks of 64 ids, filled top-down — list(node) = list(parent) | local ids — one
//            launch per tree level.  Parts with more than 4096 samples or a store beyond 8 GB do without it: their waves
//            climb the root path of each pattern of a pair instead (lane d decodes the d-th node of the path)
//   emit     one wave per pattern pair: both lists from the store into LDS, one BLOCK RECORD (row block, column block, row
//            mask, column mask, count) per pair of non-empty blocks into a pool, in arrival order
//   apply    kmdb_rect_sort_apply (a2a_blocks.hip): the all2all pipeline's counting sort by block pair and its matrix-core
//            accumulation kernel, writing the dense rows x columns matrix.
#include "kmdb_amd.h"
#include "kmdb_internal.h"
#include "engine_internal.h"

#include <hip/hip_runtime.h>
#include "prim.h"
#include "hash_probe.h"

#include <algorithm>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace {

constexpr unsigned long long D2_INVALID = ~0ull;

__device__ __forceinline__ uint32_t d2_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

struct D2Db {                                   // device pointers of one resident database
    uint64_t n_buckets;
    const uint64_t* bucket_offset;
    const uint64_t* slots;
    const uint32_t* pid2dfs;
    const uint4* meta;
    const uint64_t* bitpos;
    const int32_t* parent;
    const uint64_t* bits;
};

__global__ void d2_probe_kernel(D2Db row, D2Db col, uint64_t n_col_slots, uint32_t cbits, unsigned long long* __restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_col_slots) return;
    unsigned long long key = D2_INVALID;
    const uint64_t it = col.slots[i];
    const int32_t pc = (int32_t)(it >> 32);
    if (pc != 0x7fffffff) {
        // bucket of this slot: last bucket whose offset is <= i
        uint64_t lo = 0, hi = col.n_buckets;
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (col.bucket_offset[mid] <= i) lo = mid; else hi = mid; }
        const uint64_t b = lo;
        if (b < row.n_buckets) {
            const uint64_t off = row.bucket_offset[b], cap = row.bucket_offset[b + 1] - off;
            if (cap) {
                const int32_t pr = kmdb_probe(row.slots, off, cap, (uint32_t)it);
                if (pr != 0x7fffffff) key = ((unsigned long long)row.pid2dfs[pr] << cbits) | col.pid2dfs[pc];
            }
        }
    }
    keys[i] = key;
}

struct D2Cursor {                               // gamma stream reader (src/elias_gamma.h:104-128)
    const uint64_t* __restrict__ bits;
    uint64_t wi, c0, c1;
    uint32_t s;
    __device__ __forceinline__ D2Cursor(const uint64_t* __restrict__ b, uint64_t pos) : bits(b) {
        wi = pos >> 6; s = (uint32_t)pos & 63u; c0 = bits[wi]; c1 = bits[wi + 1];
    }
    __device__ __forceinline__ uint32_t next() {
        const uint64_t win = s ? ((c0 << s) | (c1 >> (64u - s))) : c0;
        uint32_t ones = (uint32_t)__clzll((long long)~win);
        ones = ones > 31u ? 31u : ones;
        const uint32_t low = (uint32_t)((win << ones) >> (63u - ones));
        s += 2u * ones + 1u;
        if (s >= 64u) { s -= 64u; ++wi; c0 = c1; c1 = bits[wi + 1]; }
        return low | (1u << ones);
    }
};

// full list of the pattern with DFS index `node` as a bit set over the sample ids (64 per word): lane d decodes the d-th
// node of the root path and ORs its local ids in (pattern_t::decodeSamples, src/pattern.cpp:99-109: first id = last - sum of the deltas)
__device__ __forceinline__ void d2_list_bits(const D2Db& db, uint32_t node, unsigned long long* set, uint32_t lane) {
    int64_t r = node;
    uint32_t d = 0;
    while (r >= 0) {                             // every lane walks the path; lane (d mod 64) decodes node d
        if ((d & 63u) == lane) {
            const uint4 m = db.meta[r];
            const uint32_t l = m.y;
            if (l) {
                uint32_t id = m.z;
                if (l > 1) {
                    D2Cursor c1(db.bits, db.bitpos[r]);
                    uint32_t sum = 0;
                    for (uint32_t t = 0; t + 1 < l; ++t) sum += c1.next();
                    id = m.z - sum;
                    D2Cursor c2(db.bits, db.bitpos[r]);
                    for (uint32_t t = 0; t + 1 < l; ++t) { atomicOr(&set[id >> 6], 1ull << (id & 63u)); id += c2.next(); }
                }
                atomicOr(&set[id >> 6], 1ull << (id & 63u));
            }
        }
        r = db.parent[r];
        ++d;
    }
}

// ---- the list store
// One launch per level of the tree: a node is filled in the first round that finds its parent filled by an EARLIER round
// (done[] holds round + 1; a parent filled in this very round reads as either 0 or round + 1, both mean "not yet").
constexpr uint32_t D2_SETS_MAX_NB = 64;                       // parts of up to 4096 samples
constexpr uint64_t D2_SETS_MAX_BYTES = 8ull << 30;
// (done is read and written by different threads of one launch: no __restrict__, relaxed atomic loads)
__global__ __launch_bounds__(256) void d2_sets_round_kernel(D2Db db, uint32_t P, uint32_t nb, uint32_t round, uint32_t* done,
                                                            unsigned long long* sets, uint32_t* __restrict__ n_done) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool fill = i < P && __atomic_load_n(&done[i], __ATOMIC_RELAXED) == 0;
    int32_t par = -1;
    if (fill) {
        par = db.parent[i];
        if (par >= 0) { const uint32_t dr = __atomic_load_n(&done[par], __ATOMIC_RELAXED); fill = dr != 0 && dr != round + 1u; }
    }
    if (fill) {
        unsigned long long* s = sets + (size_t)i * nb;
        if (par >= 0) { const unsigned long long* ps = sets + (size_t)par * nb; for (uint32_t k = 0; k < nb; ++k) s[k] = ps[k]; }
        else for (uint32_t k = 0; k < nb; ++k) s[k] = 0;
        const uint4 m = db.meta[i];
        const uint32_t l = m.y;
        if (l) {
            uint32_t id = m.z;
            if (l > 1) {
                D2Cursor c1(db.bits, db.bitpos[i]);
                uint32_t sum = 0;
                for (uint32_t t = 0; t + 1 < l; ++t) sum += c1.next();
                id = m.z - sum;
                D2Cursor c2(db.bits, db.bitpos[i]);
                uint32_t cw = id >> 6;
                unsigned long long acc = 0;
                for (uint32_t t = 0; t + 1 < l; ++t) {                    // ids ascend: one read-modify-write per word
                    const uint32_t wd = id >> 6;
                    if (wd != cw) { s[cw] |= acc; acc = 0; cw = wd; }
                    acc |= 1ull << (id & 63u);
                    id += c2.next();
                }
                if ((id >> 6) != cw) { s[cw] |= acc; acc = 0; cw = id >> 6; }
                acc |= 1ull << (id & 63u);
                s[cw] |= acc;
            } else s[id >> 6] |= 1ull << (id & 63u);
        }
        done[i] = round + 1u;
    }
    const unsigned long long bal = __ballot(fill);
    if (bal && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)bal) - 1u) atomicAdd(n_done, (uint32_t)__popcll(bal));
}

// Pattern pairs -> block records.  A pair (row pattern, column pattern, c shared k-mers) adds c to every cell (sample of the
// row pattern, sample of the column pattern): with both lists as bit sets over blocks of 64 ids that is one record
// (row block, column block, row mask, column mask, c) per pair of non-empty blocks — the record form of the all2all pipeline
// (a2a_blocks.hip), sorted by block pair and accumulated on the matrix cores by the same kernels.  (Round 1 added the cells
// one by one with HBM atomics: 100 ms for two 1000-sample databases.)
// One wave per pair, persistent waves.  COUNT: only the number of records (sizes the pool); else: records into the pool in
// arrival order, slots taken GRAB at a time from one of 256 cursors (a shared cursor would serialise: same-address atomics).
constexpr uint32_t D2_GRAB = 512, D2_CURSORS = 256;
// 64-bit words of LDS one wave of the pair kernel needs: both bit sets, both compacted mask lists, both 16-bit index lists
__host__ __device__ inline size_t d2_wave_words(uint32_t nbr, uint32_t nbc) { return (size_t)2 * (nbr + nbc) + ((size_t)(nbr + nbc) * 2 + 7) / 8; }
struct D2Pool { uint32_t* wkey; ulonglong2* wrec; uint32_t* cursor; uint32_t region; uint32_t kbits, dbits; uint32_t* overflow; };
template <bool COUNT, bool STORE>
__global__ __launch_bounds__(256) void d2_emit_kernel(D2Db row, D2Db col, const unsigned long long* __restrict__ rsets, const unsigned long long* __restrict__ csets,
                                                      const unsigned long long* __restrict__ pairs, const uint32_t* __restrict__ counts,
                                                      uint32_t npairs, uint32_t nbr, uint32_t nbc, uint32_t cbits, D2Pool pool,
                                                      unsigned long long* __restrict__ n_records) {
    extern __shared__ unsigned long long d2_lds[];          // per wave: row set [nbr], column set [nbc], the non-empty blocks of each, their block indices
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;       // 1 .. 4 waves per workgroup, by the LDS a wave needs
    const size_t per_wave = d2_wave_words(nbr, nbc);
    unsigned long long* rset = d2_lds + (size_t)wave * per_wave;
    unsigned long long* cset = rset + nbr;
    unsigned long long* rmask = cset + nbc;                   // compacted: mask of the a-th non-empty row block
    unsigned long long* cmask = rmask + nbr;
    uint16_t* ridx = (uint16_t*)(cmask + nbc);                // ... and its block index (a block index fits 16 bits up to 4 M samples)
    uint16_t* cidx = ridx + nbr;
    const uint32_t wid = blockIdx.x * wpb + wave, nwaves = gridDim.x * wpb;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long mine = 0;
    uint32_t next = 0, left = 0, sub = wid % D2_CURSORS;      // the wave's current grab of pool slots
    // the pairs are sorted by row pattern: a wave takes a contiguous share and keeps the row pattern's list while it repeats
    const uint32_t i_lo = (uint32_t)((uint64_t)npairs * wid / nwaves), i_hi = (uint32_t)((uint64_t)npairs * (wid + 1u) / nwaves);
    uint32_t cached_pr = 0xFFFFFFFFu, na = 0;
    for (uint32_t i = i_lo; i < i_hi; ++i) {
        const unsigned long long key = pairs[i];
        if (key == D2_INVALID) continue;
        const uint32_t pr = (uint32_t)(key >> cbits), pc = (uint32_t)(key & ((1ull << cbits) - 1ull));
        uint32_t c = counts[i];
        const bool new_row = pr != cached_pr;
        if (STORE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (the previous pair's reads)
            if (new_row) for (uint32_t k = lane; k < nbr; k += 64u) rset[k] = rsets[(size_t)pr * nbr + k];
            for (uint32_t k = lane; k < nbc; k += 64u) cset[k] = csets[(size_t)pc * nbc + k];
        } else {
            for (uint32_t k = lane + (new_row ? 0u : nbr); k < nbr + nbc; k += 64u) rset[k] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (new_row) d2_list_bits(row, pr, rset, lane);
            d2_list_bits(col, pc, cset, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the non-empty blocks of both lists, compacted in order
        uint32_t nb = 0;
        if (new_row) {
            na = 0; cached_pr = pr;
            for (uint32_t k0 = 0; k0 < nbr; k0 += 64u) {
                const uint32_t k = k0 + lane;
                const unsigned long long m = k < nbr ? rset[k] : 0ull;
                const unsigned long long bal = __ballot(m != 0);
                if (m) { const uint32_t o = na + (uint32_t)__popcll(bal & lt_mask); rmask[o] = m; ridx[o] = (uint16_t)k; }
                na += (uint32_t)__popcll(bal);
            }
        }
        for (uint32_t k0 = 0; k0 < nbc; k0 += 64u) {
            const uint32_t k = k0 + lane;
            const unsigned long long m = k < nbc ? cset[k] : 0ull;
            const unsigned long long bal = __ballot(m != 0);
            if (m) { const uint32_t o = nb + (uint32_t)__popcll(bal & lt_mask); cmask[o] = m; cidx[o] = (uint16_t)k; }
            nb += (uint32_t)__popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // one record per block pair and per base-2^dbits digit of c (exact: shifted digits add up in uint32 wrap-around arithmetic)
        uint32_t nd = 0;
        for (uint32_t w = c; w; w >>= pool.dbits) ++nd;
        const uint32_t T = na * nb * nd;
        if (COUNT) { if (lane == 0) mine += T; continue; }
        const uint32_t dmask = (1u << pool.dbits) - 1u;
        for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
            const uint32_t cnt = T - t0 < 64u ? T - t0 : 64u;
            if (left < cnt) {                                  // a fresh grab from the next cursor (the tail of the old one stays unwritten)
                sub = (sub + 61u) % D2_CURSORS;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&pool.cursor[sub * 16u], D2_GRAB);
                base = (uint32_t)__shfl((int)base, 0, 64);
                if (base + D2_GRAB > pool.region) { if (lane == 0) atomicOr(pool.overflow, 1u); base = pool.region - D2_GRAB; }
                next = sub * pool.region + base; left = D2_GRAB;
            }
            const uint32_t t = t0 + lane;
            if (t < T) {
                const uint32_t j = t % nd, ab = t / nd, a = ab / nb, b = ab - a * nb;
                const uint32_t digit = (c >> (j * pool.dbits)) & dmask;
                const uint32_t stream = (uint32_t)ridx[a] * nbc + cidx[b];
                pool.wrec[next + lane] = make_ulonglong2(rmask[a], cmask[b]);
                pool.wkey[next + lane] = stream | ((digit | (j << pool.dbits)) << pool.kbits);
            }
            next += cnt; left -= cnt;
        }
    }
    if (COUNT && lane == 0 && mine) atomicAdd(&n_records[(wid % 64u) * 8u], mine);
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <class T> T* as() { return (T*)p; }
};

D2Db view_of(const kmdb_engine_view& e) {
    return D2Db{e.n_buckets, e.bucket_offset, e.slots, e.pid2dfs, e.meta, e.bitpos, e.parent, e.bits};
}

// the handle's list store (nullptr: this part does without one).  reserve: bytes the call still has to allocate after it (record pool,
// sort): a store that would leave less than that — or take more than a third of what is free — is not made (the part's pairs then climb)
const unsigned long long* d2_list_store(const kmdb_engine_view& e, hipStream_t st, uint64_t reserve) {
    if (*e.list_sets) return *e.list_sets;
    if (*e.list_sets_tried || !e.P) return nullptr;
    *e.list_sets_tried = true;
    if (getenv("KMDB_D2_NO_STORE")) return nullptr;
    const uint32_t nb = (uint32_t)((e.N + 63) / 64);
    const uint64_t bytes = e.P * nb * 8;
    if (nb > D2_SETS_MAX_NB || bytes > D2_SETS_MAX_BYTES || e.P >= (1ull << 31)) return nullptr;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (bytes + (e.P + 1) * 4 + reserve > free_b || bytes > free_b / 3) {
            if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] db2db: no list store (%.2f GB wanted, %.2f GB free, %.2f GB kept for the call)\n", bytes / 1e9, free_b / 1e9, reserve / 1e9);
            return nullptr;
        }
    }
    unsigned long long* sets = nullptr;
    uint32_t* done = nullptr;
    if (hipMalloc((void**)&sets, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMalloc((void**)&done, (e.P + 1) * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(sets); return nullptr; }
    bool ok = hipMemsetAsync(done, 0, (e.P + 1) * 4, st) == hipSuccess;
    const D2Db v = view_of(e);
    uint32_t filled = 0, round = 0;
    while (ok && filled < e.P && round < e.P + 1) {
        // max_depth rounds fill everything; the count is read back after them (and every 32 rounds after, if that was not so)
        const uint32_t upto = round ? round + 32u : std::max<uint32_t>(e.max_depth, 1u);
        for (; round < upto; ++round)
            hipLaunchKernelGGL(d2_sets_round_kernel, dim3((unsigned)((e.P + 255) / 256)), dim3(256), 0, st, v, (uint32_t)e.P, nb, round, done, sets, done + e.P);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&filled, done + e.P, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    }
    (void)hipFree(done);
    if (!ok || filled != e.P) { (void)hipGetLastError(); (void)hipFree(sets); return nullptr; }
    if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] db2db: list store of %llu patterns x %u words (%.2f GB) filled in %u rounds\n", (unsigned long long)e.P, nb, bytes / 1e9, round);
    *e.list_sets = sets; *e.list_sets_nb = nb; *e.device_bytes += bytes;
    return sets;
}

}  // namespace

#define D2_TRY(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

static int db2db_impl(kmdb_db* db_row, kmdb_db* db_col, uint32_t* out, const kmdb_opts* opts, bool allow_store);

extern "C" int kmdb_db2db_dense(kmdb_db* db_row, kmdb_db* db_col, uint32_t* out, const kmdb_opts* opts) {
    if (!db_row || !db_col || !out) return kmdb_set_error("kmdb_db2db_dense: null argument");
    int rc = db2db_impl(db_row, db_col, out, opts, true);
    if (rc && std::strstr(kmdb_last_error(), hipGetErrorString(hipErrorOutOfMemory))) {
        // out of HBM inside the call: the handles' list stores (up to 8 GB each) are the part that can go — the pairs' lists are then built
        // by climbing, as on handles that never had a store — and the call is tried once more
        kmdb_engine_view er, ec;
        if (kmdb_engine_get(db_row, &er) || kmdb_engine_get(db_col, &ec)) return 1;
        bool freed = false;
        for (kmdb_engine_view* e : {&er, &ec})
            if (*e->list_sets) {
                (void)hipSetDevice(e->device);
                (void)hipFree(*e->list_sets);
                *e->device_bytes -= (uint64_t)e->P * *e->list_sets_nb * 8;
                *e->list_sets = nullptr; *e->list_sets_nb = 0; *e->list_sets_tried = true;
                freed = true;
            }
        (void)hipGetLastError();
        if (freed) {
            if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] db2db: out of device memory; list stores dropped, the call is repeated on the climbing path\n");
            rc = db2db_impl(db_row, db_col, out, opts, false);
        }
    }
    return rc;
}

static int db2db_impl(kmdb_db* db_row, kmdb_db* db_col, uint32_t* out, const kmdb_opts* opts, bool allow_store) {
    kmdb_engine_view er, ec;
    if (kmdb_engine_get(db_row, &er)) return 1;
    if (kmdb_engine_get(db_col, &ec)) return 1;
    if (!er.n_buckets || !er.slots || !ec.n_buckets || !ec.slots)
        return kmdb_set_error("kmdb_db2db_dense: both databases must be uploaded with hashtables");
    if (er.kmer_length != ec.kmer_length) return kmdb_set_error("kmdb_db2db_dense: the databases have different k-mer lengths");
    if (er.device != ec.device) return kmdb_set_error("kmdb_db2db_dense: the databases live on different devices");
    // (sample ids take 20 bits here as everywhere: round 4's limit of 65 535 samples per part — a 16-bit block index array of fixed size in the
    // pair kernel — is gone; what bounds a part now is the pair kernel's LDS and the 2^22 block pairs of the stream keys, checked below)
    D2_TRY(hipSetDevice(er.device));
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : (hipStream_t)er.stream;
    const uint64_t nr = er.N, nc = ec.N;
    uint64_t n_slots = 0;
    D2_TRY(hipMemcpy(&n_slots, ec.bucket_offset + ec.n_buckets, 8, hipMemcpyDeviceToHost));
    if (n_slots >= (1ull << 31)) return kmdb_set_error("kmdb_db2db_dense: column database has more than 2^31 hashtable slots");
    DevBuf d_keys, d_keys2, d_uniq, d_cnt, d_nruns, d_out, d_tmp, d_flag;
    D2_TRY(d_keys.alloc(n_slots * 8)); D2_TRY(d_keys2.alloc(n_slots * 8)); D2_TRY(d_uniq.alloc((n_slots + 1) * 8));
    D2_TRY(d_cnt.alloc((n_slots + 1) * 4)); D2_TRY(d_nruns.alloc(16)); D2_TRY(d_out.alloc(nr * nc * 4)); D2_TRY(d_flag.alloc(16));
    D2_TRY(hipMemsetAsync(d_out.p, 0, std::max<uint64_t>(nr * nc * 4, 4), st));
    D2_TRY(hipMemsetAsync(d_flag.p, 0, 16, st));
    hipEvent_t ev0 = (hipEvent_t)er.ev[0], ev3 = (hipEvent_t)er.ev[3];
    D2_TRY(hipEventRecord(ev0, st));
    const bool verbose = getenv("KMDB_VERBOSE") != nullptr;
    auto t_mark = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {                          // KMDB_VERBOSE: host-side wall time of the call's steps (each one waited for)
        if (!verbose) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kmdb] db2db: %-34s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_mark).count());
        t_mark = now;
    };
    // bits of a key: 2^cbits > patterns of the column database, 2^rbits > patterns of the row database (so no key in use is all ones
    // in the sorted bits: the unused slots' ~0 sorts last)
    uint32_t cbits = 1, rbits = 1;
    while ((1ull << cbits) <= ec.P) ++cbits;
    while ((1ull << rbits) <= er.P) ++rbits;
    const unsigned key_end = std::min<unsigned>(64u, cbits + rbits);
    if (n_slots && nr && nc) {
        size_t tb_sort = 0, tb_rle = 0;
        D2_TRY(prim::sort_keys(nullptr, tb_sort, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                                 (int)n_slots, 0, key_end, st));
        D2_TRY(prim::run_length_encode(nullptr, tb_rle, d_keys2.as<unsigned long long>(), d_uniq.as<unsigned long long>(),
                                                     d_cnt.as<uint32_t>(), d_nruns.as<uint32_t>(), (int)n_slots, st));
        D2_TRY(d_tmp.alloc(std::max(tb_sort, tb_rle)));
        const D2Db vr = view_of(er), vc = view_of(ec);
        hipLaunchKernelGGL(d2_probe_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, vr, vc, n_slots, cbits,
                           d_keys.as<unsigned long long>());
        D2_TRY(hipGetLastError());
        D2_TRY(prim::sort_keys(d_tmp.p, tb_sort, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                                 (int)n_slots, 0, key_end, st));
        D2_TRY(prim::run_length_encode(d_tmp.p, tb_rle, d_keys2.as<unsigned long long>(), d_uniq.as<unsigned long long>(),
                                                     d_cnt.as<uint32_t>(), d_nruns.as<uint32_t>(), (int)n_slots, st));
        uint32_t nruns = 0;
        D2_TRY(hipMemcpyAsync(&nruns, d_nruns.p, 4, hipMemcpyDeviceToHost, st));
        D2_TRY(hipStreamSynchronize(st));
        phase("probe + sort + run lengths");
        if (nruns) {
            // pairs -> block records -> sorted by block pair -> accumulated on the matrix cores (a2a_blocks.hip)
            const uint32_t nbr = (uint32_t)((nr + 63) / 64), nbc = (uint32_t)((nc + 63) / 64);
            int key_bits = 1;
            while ((1ull << key_bits) <= (uint64_t)nbr * nbc + 1) ++key_bits;
            const uint32_t dbits = (uint32_t)(32 - key_bits - 2);
            if ((uint64_t)nbr * nbc + 1 >= (1ull << 22)) return kmdb_set_error("kmdb_db2db_dense: more than 2^22 block pairs (parts of " + std::to_string(nr) + " x " + std::to_string(nc) + " samples)");
            const size_t wave_lds = d2_wave_words(nbr, nbc) * 8;
            if (wave_lds > (size_t)(150u << 10)) return kmdb_set_error("kmdb_db2db_dense: the sample lists of a pattern pair do not fit the LDS (parts of " + std::to_string(nr) + " + " + std::to_string(nc) + " samples; about 540 000 in all is the limit)");
            const uint32_t wpb = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (size_t)(150u << 10) / wave_lds));
            const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)nruns + wpb - 1) / wpb, 256 * 8);
            const size_t lds = wave_lds * wpb;
            // both parts' list stores, or neither (a wave that climbs one side's root paths gains nothing from the other's store)
            // (what the call allocates after the stores: six records of 20 bytes per pair, twice — the pool and the sort's copy — and slack)
            const uint64_t reserve = (uint64_t)nruns * 6 * 20 * 2 + (1ull << 30);
            const unsigned long long* rsets = allow_store ? d2_list_store(er, st, reserve) : nullptr;
            const unsigned long long* csets = rsets ? (db_col == db_row ? rsets : d2_list_store(ec, st, reserve)) : nullptr;
            const bool store = rsets && csets;
            phase("list stores");
            DevBuf d_nrec, d_cursor;
            D2_TRY(d_nrec.alloc(64 * 8 * 8)); D2_TRY(d_cursor.alloc(D2_CURSORS * 16 * 4));
            D2_TRY(hipMemsetAsync(d_nrec.p, 0, 64 * 8 * 8, st));
            D2_TRY(hipMemsetAsync(d_cursor.p, 0, D2_CURSORS * 16 * 4, st));
            D2Pool pool{nullptr, nullptr, d_cursor.as<uint32_t>(), 0u, (uint32_t)key_bits, dbits, d_flag.as<uint32_t>() + 1};
            D2_TRY(hipFuncSetAttribute((const void*)d2_emit_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            D2_TRY(hipFuncSetAttribute((const void*)d2_emit_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            D2_TRY(hipFuncSetAttribute((const void*)d2_emit_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            D2_TRY(hipFuncSetAttribute((const void*)d2_emit_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            // the pool: sized for six records per pair first; if that overflows, the
            // records are counted and the pool is made to measure
            DevBuf d_wkey, d_wrec;
            uint64_t slots = 0;
            for (int attempt = 0; attempt < 2; ++attempt) {
                unsigned long long total = 6ull * nruns;
                if (attempt) {
                    D2_TRY(hipMemsetAsync(d_nrec.p, 0, 64 * 8 * 8, st));
                    if (store)
                        hipLaunchKernelGGL((d2_emit_kernel<true, true>), dim3(grid), dim3(64 * wpb), lds, st, vr, vc, rsets, csets, d_uniq.as<unsigned long long>(), d_cnt.as<uint32_t>(),
                                           nruns, nbr, nbc, cbits, pool, d_nrec.as<unsigned long long>());
                    else
                        hipLaunchKernelGGL((d2_emit_kernel<true, false>), dim3(grid), dim3(64 * wpb), lds, st, vr, vc, rsets, csets, d_uniq.as<unsigned long long>(), d_cnt.as<uint32_t>(),
                                           nruns, nbr, nbc, cbits, pool, d_nrec.as<unsigned long long>());
                    D2_TRY(hipGetLastError());
                    unsigned long long h_nrec[64 * 8];
                    D2_TRY(hipMemcpyAsync(h_nrec, d_nrec.p, sizeof h_nrec, hipMemcpyDeviceToHost, st));
                    D2_TRY(hipStreamSynchronize(st));
                    total = 0;
                    for (int q = 0; q < 64; ++q) total += h_nrec[q * 8];
                }
                // per cursor: its share of the records and of the grab tails (up to 63 slots per grab), one unfinished grab per wave
                // that ends there, and room for the uneven ends of the waves' round-robin over the cursors
                const uint64_t grabs = (total + total / 8) / D2_GRAB / D2_CURSORS + (uint64_t)grid * wpb / D2_CURSORS + 130;
                const uint64_t region = grabs * D2_GRAB;
                slots = region * D2_CURSORS;
                if (slots >= (1ull << 31)) return kmdb_set_error("kmdb_db2db_dense: more than 2^31 block records");
                if (d_wkey.p) { (void)hipFree(d_wkey.p); d_wkey.p = nullptr; (void)hipFree(d_wrec.p); d_wrec.p = nullptr; }
                D2_TRY(d_wkey.alloc(slots * 4)); D2_TRY(d_wrec.alloc(slots * 16));
                D2_TRY(hipMemsetAsync(d_wkey.p, 0xFF, slots * 4, st));
                D2_TRY(hipMemsetAsync(d_cursor.p, 0, D2_CURSORS * 16 * 4, st));
                D2_TRY(hipMemsetAsync(d_flag.as<uint32_t>() + 1, 0, 4, st));
                pool.wkey = d_wkey.as<uint32_t>(); pool.wrec = d_wrec.as<ulonglong2>(); pool.region = (uint32_t)region;
                if (store)
                    hipLaunchKernelGGL((d2_emit_kernel<false, true>), dim3(grid), dim3(64 * wpb), lds, st, vr, vc, rsets, csets, d_uniq.as<unsigned long long>(), d_cnt.as<uint32_t>(),
                                       nruns, nbr, nbc, cbits, pool, d_nrec.as<unsigned long long>());
                else
                    hipLaunchKernelGGL((d2_emit_kernel<false, false>), dim3(grid), dim3(64 * wpb), lds, st, vr, vc, rsets, csets, d_uniq.as<unsigned long long>(), d_cnt.as<uint32_t>(),
                                       nruns, nbr, nbc, cbits, pool, d_nrec.as<unsigned long long>());
                D2_TRY(hipGetLastError());
                uint32_t ovf = 0;
                D2_TRY(hipMemcpyAsync(&ovf, d_flag.as<uint32_t>() + 1, 4, hipMemcpyDeviceToHost, st));
                D2_TRY(hipStreamSynchronize(st));
                if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] db2db: %u pattern pairs, pool for %llu records (%s)%s\n", nruns, total, attempt ? "counted" : "estimate", ovf ? ": too small" : "");
                if (!ovf) break;
                if (attempt) return kmdb_set_error("kmdb_db2db_dense: internal error (record pool overflow)");
            }
            phase("pool + emit");
            if (kmdb_rect_sort_apply(st, d_wkey.as<uint32_t>(), d_wrec.p, (uint32_t)slots, nbr, nbc, key_bits, d_out.as<uint32_t>(), (uint32_t)nr, (uint32_t)nc)) return 1;
            phase("sort + apply");
        }
    }
    D2_TRY(hipEventRecord(ev3, st));
    D2_TRY(hipEventSynchronize(ev3));
    float ms = 0;
    D2_TRY(hipEventElapsedTime(&ms, ev0, ev3));
    kmdb_engine_set_times(db_row, ms, ms);
    uint32_t too_long = 0;
    D2_TRY(hipMemcpy(&too_long, d_flag.p, 4, hipMemcpyDeviceToHost));
    if (too_long) return kmdb_set_error("kmdb_db2db_dense: internal error");
    if (nr * nc) D2_TRY(hipMemcpy(out, d_out.p, nr * nc * 4, hipMemcpyDeviceToHost));
    return 0;
}
