// db2db.hip — two-database counting behind kmdb_db2db_dense (include/kmdb_amd.h).
//
// Replaces SimilarityCalculator::db2db_sp (reference src/similarity_calculator.cpp:1225-1540), the cell (row part,
// column part) of the all2all-parts grid (src/console_all2all_parts.cpp:143-331).  The reference merge-joins the
// sorted (suffix, pattern) lists of every k-mer prefix of the two databases, counts the equal (pattern_row,
// pattern_col) pairs and adds the count to every pair of samples of the two patterns.  Here:
//   probe    one thread per hashtable slot of the COLUMN database: its k-mer is looked up in the row database's
//            table of the same prefix (murmur3 fmix32 probe, src/hashmap_lp.h:53-64,308-333);
//            key = DFS index of the row pattern << 32 | DFS index of the column pattern, or ~0
//   count    radix sort + run-length encode of the keys: (pattern pair, number of shared k-mers)
//   scatter  one wave per pattern pair: the full sample lists of both patterns are decoded into LDS (every node of
//            a root path writes its local ids at positions [n - l, n) of the list — no ordering between the lanes
//            that decode different nodes), then out[row sample][column sample] += count over the cross product.
#include "kmdb_amd.h"
#include "kmdb_internal.h"
#include "engine_internal.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>

namespace {

constexpr unsigned long long D2_INVALID = ~0ull;
constexpr uint32_t D2_MAXN = 2048;            // longest full list the scatter kernel stages in LDS

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

__global__ void d2_probe_kernel(D2Db row, D2Db col, uint64_t n_col_slots, unsigned long long* __restrict__ keys) {
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
                const uint64_t mask = cap - 1;
                const uint32_t kk = (uint32_t)it;
                uint64_t h = (uint64_t)d2_fmix32(kk) & mask;
                for (uint64_t step = 0; step < cap; ++step) {        // a full table (corrupt file) ends the probe too
                    const uint64_t r = row.slots[off + h];
                    const int32_t pr = (int32_t)(r >> 32);
                    if (pr == 0x7fffffff) break;
                    if ((uint32_t)r == kk) { key = ((unsigned long long)row.pid2dfs[pr] << 32) | col.pid2dfs[pc]; break; }
                    h = (h + 1) & mask;
                }
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

// full list of the pattern with DFS index `node` into out[0..n): lane d decodes the d-th node of the root path
__device__ __forceinline__ uint32_t d2_decode_list(const D2Db& db, uint32_t node, uint16_t* out, uint32_t lane) {
    const uint32_t n = db.meta[node].x;
    int64_t r = node;
    uint32_t d = 0;
    while (r >= 0) {                             // every lane walks the path; lane (d mod 64) decodes node d
        if ((d & 63u) == lane) {
            const uint4 m = db.meta[r];
            const uint32_t l = m.y;
            if (l) {
                uint32_t id = m.z;
                if (l > 1) {                     // pattern_t::decodeSamples (src/pattern.cpp:99-109)
                    D2Cursor c1(db.bits, db.bitpos[r]);
                    uint32_t sum = 0;
                    for (uint32_t t = 0; t + 1 < l; ++t) sum += c1.next();
                    id = m.z - sum;
                    D2Cursor c2(db.bits, db.bitpos[r]);
                    uint32_t pos = m.x - l;
                    for (uint32_t t = 0; t + 1 < l; ++t) { out[pos++] = (uint16_t)id; id += c2.next(); }
                }
                out[m.x - 1] = (uint16_t)id;
            }
        }
        r = db.parent[r];
        ++d;
    }
    return n;
}

__global__ __launch_bounds__(256) void d2_scatter_kernel(D2Db row, D2Db col, const unsigned long long* __restrict__ pairs,
                                                         const uint32_t* __restrict__ counts, uint32_t npairs, uint32_t n_col,
                                                         uint32_t* __restrict__ out, uint32_t* __restrict__ too_long) {
    __shared__ uint16_t lists[4][2][D2_MAXN];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 4 + wave;
    if (i >= npairs) return;
    const unsigned long long key = pairs[i];
    if (key == D2_INVALID) return;
    const uint32_t pr = (uint32_t)(key >> 32), pc = (uint32_t)key, c = counts[i];
    if (row.meta[pr].x > D2_MAXN || col.meta[pc].x > D2_MAXN) { if (lane == 0) atomicAdd(too_long, 1u); return; }
    uint16_t* A = lists[wave][0];
    uint16_t* B = lists[wave][1];
    const uint32_t n1 = d2_decode_list(row, pr, A, lane);
    const uint32_t n2 = d2_decode_list(col, pc, B, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t total = n1 * n2;
    for (uint32_t t = lane; t < total; t += 64) {
        const uint32_t a = t / n2, b = t - a * n2;
        atomicAdd(&out[(size_t)A[a] * n_col + B[b]], c);
    }
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

}  // namespace

#define D2_TRY(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

extern "C" int kmdb_db2db_dense(kmdb_db* db_row, kmdb_db* db_col, uint32_t* out, const kmdb_opts* opts) {
    if (!db_row || !db_col || !out) return kmdb_set_error("kmdb_db2db_dense: null argument");
    kmdb_engine_view er, ec;
    if (kmdb_engine_get(db_row, &er)) return 1;
    if (kmdb_engine_get(db_col, &ec)) return 1;
    if (!er.n_buckets || !er.slots || !ec.n_buckets || !ec.slots)
        return kmdb_set_error("kmdb_db2db_dense: both databases must be uploaded with hashtables");
    if (er.kmer_length != ec.kmer_length) return kmdb_set_error("kmdb_db2db_dense: the databases have different k-mer lengths");
    if (er.device != ec.device) return kmdb_set_error("kmdb_db2db_dense: the databases live on different devices");
    if (er.N > 65535 || ec.N > 65535) return kmdb_set_error("kmdb_db2db_dense: more than 65535 samples is not supported yet");
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
    if (n_slots && nr && nc) {
        size_t tb_sort = 0, tb_rle = 0;
        D2_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tb_sort, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                                 (int)n_slots, 0, 64, st));
        D2_TRY(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb_rle, d_keys2.as<unsigned long long>(), d_uniq.as<unsigned long long>(),
                                                     d_cnt.as<uint32_t>(), d_nruns.as<uint32_t>(), (int)n_slots, st));
        D2_TRY(d_tmp.alloc(std::max(tb_sort, tb_rle)));
        const D2Db vr = view_of(er), vc = view_of(ec);
        hipLaunchKernelGGL(d2_probe_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, vr, vc, n_slots,
                           d_keys.as<unsigned long long>());
        D2_TRY(hipGetLastError());
        D2_TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tb_sort, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                                 (int)n_slots, 0, 64, st));
        D2_TRY(hipcub::DeviceRunLengthEncode::Encode(d_tmp.p, tb_rle, d_keys2.as<unsigned long long>(), d_uniq.as<unsigned long long>(),
                                                     d_cnt.as<uint32_t>(), d_nruns.as<uint32_t>(), (int)n_slots, st));
        uint32_t nruns = 0;
        D2_TRY(hipMemcpyAsync(&nruns, d_nruns.p, 4, hipMemcpyDeviceToHost, st));
        D2_TRY(hipStreamSynchronize(st));
        if (nruns) {
            hipLaunchKernelGGL(d2_scatter_kernel, dim3((nruns + 3) / 4), dim3(256), 0, st, vr, vc, d_uniq.as<unsigned long long>(),
                               d_cnt.as<uint32_t>(), nruns, (uint32_t)nc, d_out.as<uint32_t>(), d_flag.as<uint32_t>());
            D2_TRY(hipGetLastError());
        }
    }
    D2_TRY(hipEventRecord(ev3, st));
    D2_TRY(hipEventSynchronize(ev3));
    float ms = 0;
    D2_TRY(hipEventElapsedTime(&ms, ev0, ev3));
    kmdb_engine_set_times(db_row, ms, ms);
    uint32_t too_long = 0;
    D2_TRY(hipMemcpy(&too_long, d_flag.p, 4, hipMemcpyDeviceToHost));
    if (too_long) return kmdb_set_error("kmdb_db2db_dense: a pattern with more than 2048 samples is not supported yet");
    if (nr * nc) D2_TRY(hipMemcpy(out, d_out.p, nr * nc * 4, hipMemcpyDeviceToHost));
    return 0;
}
