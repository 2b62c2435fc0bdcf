// new2all.hip — batched query path behind kmdb_new2all_batch (include/kmdb_amd.h).
//
// Replaces SimilarityCalculator::one2all<false> / one2all_sp (reference
// src/similarity_calculator.cpp:809-925, 929-1051).  The reference does, per query,
//   (1) one hash_map_lp::find per k-mer (src/hashmap_lp.h:308-333) and a per-pattern hit count
//       (an unordered_map), then
//   (2) for every hit pattern: decode the whole parent chain, similarities[id] += hits.
// Here a whole BATCH of queries runs through four device steps, all sized by the number of
// k-mers / hits, never by the size of the database:
//   probe   one thread per k-mer: bucket = kmer >> 32 (src/types.h:25-27), murmur3 fmix32 probe
//           (src/hashmap_lp.h:53-64), key (query << 32 | DFS index of the hit pattern) or ~0
//   sort    radix sort of the keys: hits grouped by query, ascending DFS index inside a query
//   count   run-length encode (pattern, hits) + exclusive scan of the hit counts
//   walk    step (2) restated on the DFS layout of engine.hip.  With H(r) = hits in subtree(r) =
//           the hits whose DFS index lies in [r, sub_end[r]),
//                 similarities[s] = sum over nodes r that hold s as a LOCAL id of H(r).
//           Every node with H(r) > 0 is an ancestor-or-self of its first hit h_i and is larger than
//           the previous hit h_{i-1}; so thread i climbs from h_i through the parent links while the
//           node index stays above h_{i-1}: every such node is visited exactly once per query,
//           decodes only its own gamma stream, gets H(r) from two reads of the scanned counts (one
//           binary search over the query's sorted hits), and adds it to its local ids in an LDS
//           histogram of the workgroup, flushed with one global atomic per touched sample.
#include "kmdb_amd.h"
#include "kmdb_internal.h"
#include "engine_internal.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr unsigned long long N2_INVALID = ~0ull;

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {        // murmur3 finaliser (src/hashmap_lp.h:53-64)
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// (1) probe: one thread per k-mer of the batch
__global__ void n2a_probe_kernel(const uint64_t* __restrict__ kmers, const uint64_t* __restrict__ qoff, uint32_t nq, size_t total,
                                 uint64_t n_buckets, const uint64_t* __restrict__ bucket_offset, const uint64_t* __restrict__ slots,
                                 const uint32_t* __restrict__ pid2dfs, const uint32_t* __restrict__ w,
                                 unsigned long long* __restrict__ keys) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        unsigned long long key = N2_INVALID;
        const uint64_t k = kmers[i];
        const uint64_t b = k >> 32;
        if (b < n_buckets) {
            const uint64_t off = bucket_offset[b];
            const uint64_t cap = bucket_offset[b + 1] - off;
            if (cap) {
                const uint64_t mask = cap - 1;
                const uint32_t kk = (uint32_t)k;
                uint64_t h = (uint64_t)fmix32(kk) & mask;
                for (;;) {
                    const uint64_t it = slots[off + h];
                    const int32_t val = (int32_t)(it >> 32);
                    if (val == 0x7fffffff) break;                // empty slot ends the probe (src/hashmap_lp.h:78)
                    if ((uint32_t)it == kk) {
                        const uint32_t d = pid2dfs[val];
                        if (w[d] != 0) {                          // :847-848 skips patterns without k-mers
                            // query of this k-mer: binary search in the batch's offsets
                            uint32_t lo = 0, hi = nq;
                            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (qoff[mid] <= i) lo = mid; else hi = mid; }
                            key = ((unsigned long long)lo << 32) | d;
                        }
                        break;
                    }
                    h = (h + 1) & mask;
                }
            }
        }
        keys[i] = key;
    }
}

// first run of every query in the sorted, run-length-encoded hit list
__global__ void n2a_query_ranges_kernel(const unsigned long long* __restrict__ uniq, uint32_t nruns, uint32_t nq, uint32_t* __restrict__ qstart) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nq) return;
    const unsigned long long target = (unsigned long long)q << 32;
    uint32_t lo = 0, hi = nruns;                                  // first run with key >= target
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (uniq[mid] < target) lo = mid + 1; else hi = mid; }
    qstart[q] = lo;
}

struct N2Cursor {                                                 // gamma stream reader (see BitCursor in engine.hip)
    const uint64_t* __restrict__ bits;
    uint64_t wi, c0, c1;
    uint32_t s;
    __device__ __forceinline__ N2Cursor(const uint64_t* __restrict__ b, uint64_t pos) : bits(b) {
        wi = pos >> 6; s = (uint32_t)pos & 63u; c0 = bits[wi]; c1 = bits[wi + 1];
    }
    __device__ __forceinline__ uint32_t next() {                  // (L-1) ones, a zero, (L-1) low bits (src/elias_gamma.h:104-128)
        const uint64_t win = s ? ((c0 << s) | (c1 >> (64u - s))) : c0;
        uint32_t ones = (uint32_t)__clzll((long long)~win);
        ones = ones > 31u ? 31u : ones;
        const uint32_t low = (uint32_t)((win << ones) >> (63u - ones));
        s += 2u * ones + 1u;
        if (s >= 64u) { s -= 64u; ++wi; c0 = c1; c1 = bits[wi + 1]; }
        return low | (1u << ones);
    }
};

// (4) walk: one thread per distinct (query, hit pattern)
template <bool LDS_HIST>
__global__ __launch_bounds__(256) void n2a_walk_kernel(const unsigned long long* __restrict__ uniq, const uint32_t* __restrict__ csum,
                                                       const uint32_t* __restrict__ qstart, uint32_t nruns, uint32_t nq,
                                                       const uint4* __restrict__ meta, const uint64_t* __restrict__ bitpos,
                                                       const int32_t* __restrict__ parent, const uint32_t* __restrict__ sub_end,
                                                       const uint64_t* __restrict__ bits, uint32_t N, uint32_t* __restrict__ sim) {
    extern __shared__ uint32_t hist[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < nruns && uniq[i] != N2_INVALID;
    const uint32_t myq = live ? (uint32_t)(uniq[i] >> 32) : 0xFFFFFFFFu;
    // the block's runs are sorted by query: loop over the (few) queries it spans
    __shared__ uint32_t q_lo, q_hi;
    if (threadIdx.x == 0) { q_lo = 0xFFFFFFFFu; q_hi = 0; }
    __syncthreads();
    if (live) { atomicMin(&q_lo, myq); atomicMax(&q_hi, myq); }
    __syncthreads();
    if (q_lo == 0xFFFFFFFFu) return;
    for (uint32_t q = q_lo; q <= q_hi; ++q) {
        if (LDS_HIST) {
            for (uint32_t s = threadIdx.x; s < N; s += blockDim.x) hist[s] = 0;
            __syncthreads();
        }
        uint32_t* acc = LDS_HIST ? hist : (sim + (size_t)q * N);
        if (live && myq == q) {
            const uint32_t qs = qstart[q], qe = qstart[q + 1];
            const uint32_t h = (uint32_t)uniq[i];
            const int64_t prev = i > qs ? (int64_t)(uint32_t)uniq[i - 1] : -1;
            const uint32_t cbase = csum[i];
            int64_t r = h;
            while (r > prev) {
                // hits below r: indices [i, ub), ub = first run of this query whose pattern is >= sub_end[r]
                const uint32_t se = sub_end[r];
                uint32_t lo = i + 1, hi = qe;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)uniq[mid] < se) lo = mid + 1; else hi = mid; }
                const uint32_t H = csum[lo] - cbase;
                const uint4 m = meta[r];
                const uint32_t l = m.y;
                if (l) {
                    uint32_t id = m.z;
                    if (l > 1) {
                        // pattern_t::decodeSamples (src/pattern.cpp:99-109): first id = last - sum of the deltas
                        N2Cursor c1(bits, bitpos[r]);
                        uint32_t sum = 0;
                        for (uint32_t t = 0; t + 1 < l; ++t) sum += c1.next();
                        id = m.z - sum;
                        N2Cursor c2(bits, bitpos[r]);
                        for (uint32_t t = 0; t + 1 < l; ++t) { atomicAdd(&acc[id], H); id += c2.next(); }
                    }
                    atomicAdd(&acc[id], H);
                }
                r = parent[r];
            }
        }
        if (LDS_HIST) {
            __syncthreads();
            uint32_t* out = sim + (size_t)q * N;
            for (uint32_t s = threadIdx.x; s < N; s += blockDim.x)
                if (hist[s]) atomicAdd(&out[s], hist[s]);
            __syncthreads();
        }
    }
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <class T> T* as() { return (T*)p; }
};

}  // namespace

#define N2_TRY(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

extern "C" int kmdb_new2all_batch(kmdb_db* dbh, const uint64_t* const* kmers, const size_t* counts, size_t nq,
                                  uint32_t* out_dense, const kmdb_opts* opts) {
    if (!dbh || (nq && (!kmers || !counts || !out_dense))) return kmdb_set_error("kmdb_new2all_batch: null argument");
    kmdb_engine_view e;
    kmdb_engine_get(dbh, &e);
    if (!e.n_buckets || !e.slots) return kmdb_set_error("kmdb_new2all_batch: database was uploaded without hashtables");
    if (nq >= (1ull << 31)) return kmdb_set_error("kmdb_new2all_batch: too many queries in one batch");
    N2_TRY(hipSetDevice(e.device));
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : (hipStream_t)e.stream;
    const uint64_t N = e.N;
    std::vector<uint64_t> qoff(nq + 1, 0);
    for (size_t q = 0; q < nq; ++q) qoff[q + 1] = qoff[q] + counts[q];
    const size_t total = qoff[nq];
    if (total >= (1ull << 32) - 2) return kmdb_set_error("kmdb_new2all_batch: more than 2^32 k-mers in one batch; split it");
    if (!nq) return 0;

    DevBuf d_k, d_qoff, d_keys, d_keys2, d_uniq, d_cnt, d_csum, d_nruns, d_qstart, d_sim, d_tmp;
    N2_TRY(d_k.alloc(total * 8));
    N2_TRY(d_qoff.alloc((nq + 1) * 8));
    N2_TRY(d_keys.alloc(total * 8));
    N2_TRY(d_keys2.alloc(total * 8));
    N2_TRY(d_uniq.alloc((total + 1) * 8));
    N2_TRY(d_cnt.alloc((total + 2) * 4));
    N2_TRY(d_csum.alloc((total + 2) * 4));
    N2_TRY(d_nruns.alloc(16));
    N2_TRY(d_qstart.alloc((nq + 2) * 4));
    N2_TRY(d_sim.alloc(nq * N * 4));
    size_t tb_sort = 0, tb_rle = 0, tb_scan = 0;
    N2_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tb_sort, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                             (int)total, 0, 64, st));
    N2_TRY(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb_rle, d_keys2.as<unsigned long long>(), d_uniq.as<unsigned long long>(),
                                                 d_cnt.as<uint32_t>(), d_nruns.as<uint32_t>(), (int)total, st));
    N2_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb_scan, d_cnt.as<uint32_t>(), d_csum.as<uint32_t>(), (int)(total + 1), st));
    N2_TRY(d_tmp.alloc(std::max(tb_sort, std::max(tb_rle, tb_scan))));

    for (size_t q = 0; q < nq; ++q)
        if (counts[q]) N2_TRY(hipMemcpyAsync(d_k.as<uint64_t>() + qoff[q], kmers[q], counts[q] * 8, hipMemcpyHostToDevice, st));
    N2_TRY(hipMemcpyAsync(d_qoff.p, qoff.data(), (nq + 1) * 8, hipMemcpyHostToDevice, st));
    N2_TRY(hipMemsetAsync(d_sim.p, 0, std::max<uint64_t>(nq * N * 4, 4), st));

    hipEvent_t ev0 = (hipEvent_t)e.ev[0], ev3 = (hipEvent_t)e.ev[3];
    N2_TRY(hipEventRecord(ev0, st));
    uint32_t nruns = 0;
    if (total) {
        const unsigned blocks = (unsigned)std::min<size_t>(65535, (total + 255) / 256);
        hipLaunchKernelGGL(n2a_probe_kernel, dim3(blocks), dim3(256), 0, st, d_k.as<uint64_t>(), d_qoff.as<uint64_t>(), (uint32_t)nq,
                           total, e.n_buckets, e.bucket_offset, e.slots, e.pid2dfs, e.w, d_keys.as<unsigned long long>());
        N2_TRY(hipGetLastError());
        // queries need 32 - clz(nq) high bits; sorting all 64 is simplest and the key count is small
        N2_TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tb_sort, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                                 (int)total, 0, 64, st));
        N2_TRY(hipMemsetAsync(d_cnt.p, 0, (total + 2) * 4, st));
        N2_TRY(hipcub::DeviceRunLengthEncode::Encode(d_tmp.p, tb_rle, d_keys2.as<unsigned long long>(), d_uniq.as<unsigned long long>(),
                                                     d_cnt.as<uint32_t>(), d_nruns.as<uint32_t>(), (int)total, st));
        N2_TRY(hipMemcpyAsync(&nruns, d_nruns.p, 4, hipMemcpyDeviceToHost, st));
        N2_TRY(hipStreamSynchronize(st));
        N2_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb_scan, d_cnt.as<uint32_t>(), d_csum.as<uint32_t>(), (int)(nruns + 1), st));
        hipLaunchKernelGGL(n2a_query_ranges_kernel, dim3((unsigned)((nq + 1 + 255) / 256)), dim3(256), 0, st,
                           d_uniq.as<unsigned long long>(), nruns, (uint32_t)nq, d_qstart.as<uint32_t>());
        if (nruns) {
            const unsigned wblocks = (nruns + 255) / 256;
            if (N * 4 <= 64 * 1024)
                hipLaunchKernelGGL(n2a_walk_kernel<true>, dim3(wblocks), dim3(256), N * 4, st, d_uniq.as<unsigned long long>(),
                                   d_csum.as<uint32_t>(), d_qstart.as<uint32_t>(), nruns, (uint32_t)nq, e.meta, e.bitpos, e.parent,
                                   e.sub_end, e.bits, (uint32_t)N, d_sim.as<uint32_t>());
            else
                hipLaunchKernelGGL(n2a_walk_kernel<false>, dim3(wblocks), dim3(256), 0, st, d_uniq.as<unsigned long long>(),
                                   d_csum.as<uint32_t>(), d_qstart.as<uint32_t>(), nruns, (uint32_t)nq, e.meta, e.bitpos, e.parent,
                                   e.sub_end, e.bits, (uint32_t)N, d_sim.as<uint32_t>());
        }
        N2_TRY(hipGetLastError());
    }
    N2_TRY(hipEventRecord(ev3, st));
    N2_TRY(hipEventSynchronize(ev3));
    float ms = 0;
    N2_TRY(hipEventElapsedTime(&ms, ev0, ev3));
    kmdb_engine_set_times(dbh, ms, ms);
    if (nq * N) N2_TRY(hipMemcpy(out_dense, d_sim.p, nq * N * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int kmdb_new2all_batch_sparse(kmdb_db* dbh, const uint64_t* const* kmers, const size_t* counts, size_t nq,
                                         kmdb_sparse_rows* out, const kmdb_opts* opts) {
    if (!out) return kmdb_set_error("kmdb_new2all_batch_sparse: null argument");
    std::memset(out, 0, sizeof *out);
    kmdb_engine_view e;
    if (!dbh) return kmdb_set_error("kmdb_new2all_batch_sparse: null argument");
    kmdb_engine_get(dbh, &e);
    const uint64_t N = e.N;
    std::vector<uint32_t> dense(std::max<uint64_t>(nq * N, 1));
    if (kmdb_new2all_batch(dbh, kmers, counts, nq, dense.data(), opts)) return 1;
    // one2all_sp returns the (sample, count) pairs with count > 0 ordered by sample id (:1040-1047)
    out->n_rows = nq;
    out->row_ptr = (uint64_t*)std::malloc((nq + 1) * 8);
    uint64_t nnz = 0;
    for (size_t q = 0; q < nq; ++q) {
        out->row_ptr[q] = nnz;
        for (uint64_t s = 0; s < N; ++s) nnz += dense[q * N + s] != 0;
    }
    out->row_ptr[nq] = nnz;
    out->nnz = nnz;
    out->col = (uint32_t*)std::malloc(std::max<uint64_t>(nnz, 1) * 4);
    out->val = (uint32_t*)std::malloc(std::max<uint64_t>(nnz, 1) * 4);
    uint64_t o = 0;
    for (size_t q = 0; q < nq; ++q)
        for (uint64_t s = 0; s < N; ++s)
            if (dense[q * N + s]) { out->col[o] = (uint32_t)s; out->val[o] = dense[q * N + s]; ++o; }
    return 0;
}
