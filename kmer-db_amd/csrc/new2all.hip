// new2all.hip — batched query path behind kmdb_new2all_batch (include/kmdb_amd.h).
//
// Replaces SimilarityCalculator::one2all<false> / one2all_sp (reference
// src/similarity_calculator.cpp:809-925, 929-1051).  The reference does, per query,
//   (1) one hash_map_lp::find per k-mer (src/hashmap_lp.h:308-333) and a per-pattern hit count,
//   (2) for every hit pattern: decode the whole parent chain, similarities[id] += hits.
// Here step (2) is restated on the DFS layout (engine.hip): with c[p] = hits on node p, the
// number of hits in the subtree of r is C[sub_end[r]] - C[r] (C = exclusive scan of c), and
//   similarities[s] = sum over nodes r that hold s as a LOCAL id of hits_in_subtree(r),
// so every node is decoded at most once per query and no parent chain is walked.
// Round-1 implementation: one dense pass over the node table per query (count, scan,
// accumulate).  DESIGN.md §new2all lists the planned sparse (hit-driven) formulation.
#include "kmdb_amd.h"
#include "kmdb_internal.h"
#include "engine_internal.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

namespace {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {        // murmur3 finaliser (src/hashmap_lp.h:53-64)
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// (1) probe + per-node hit histogram.  prefix = kmer >> 32 selects the bucket, suffix is the key
// (reference src/types.h:25-27).
__global__ void n2a_lookup_kernel(const uint64_t* __restrict__ kmers, size_t n, uint64_t n_buckets,
                                  const uint64_t* __restrict__ bucket_offset, const uint64_t* __restrict__ slots,
                                  const uint32_t* __restrict__ pid2dfs, const uint32_t* __restrict__ w,
                                  uint32_t* __restrict__ cnt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint64_t k = kmers[i];
        const uint64_t b = k >> 32;
        if (b >= n_buckets) continue;
        const uint64_t off = bucket_offset[b];
        const uint64_t cap = bucket_offset[b + 1] - off;
        if (!cap) continue;
        const uint64_t mask = cap - 1;
        const uint32_t key = (uint32_t)k;
        uint64_t h = (uint64_t)fmix32(key) & mask;
        for (;;) {
            const uint64_t it = slots[off + h];
            const int32_t val = (int32_t)(it >> 32);
            if (val == 0x7fffffff) break;                        // empty slot ends the probe (src/hashmap_lp.h:78)
            if ((uint32_t)it == key) {
                const uint32_t d = pid2dfs[val];
                if (w[d] != 0) atomicAdd(&cnt[d], 1u);          // :847-848 skips patterns without k-mers
                break;
            }
            h = (h + 1) & mask;
        }
    }
}

__device__ __forceinline__ uint64_t n2a_window(const uint64_t* __restrict__ bits, uint64_t pos) {
    const uint64_t wi = pos >> 6;
    const uint32_t s = (uint32_t)pos & 63u;
    const uint64_t w0 = bits[wi], w1 = bits[wi + 1];
    return s ? ((w0 << s) | (w1 >> (64u - s))) : w0;
}
__device__ __forceinline__ uint32_t n2a_gamma(const uint64_t* __restrict__ bits, uint64_t& pos) {
    const uint64_t win = n2a_window(bits, pos);
    uint32_t ones = (uint32_t)__clzll((long long)~win);
    ones = ones > 31u ? 31u : ones;
    const uint32_t low = (uint32_t)((win << ones) >> (63u - ones));
    pos += 2u * ones + 1u;
    return low | (1u << ones);
}

// (2) every node with hits below it adds that count to its local sample ids
template <bool LDS_HIST>
__global__ void n2a_accumulate_kernel(const uint4* __restrict__ meta, const uint64_t* __restrict__ bitpos,
                                      const uint32_t* __restrict__ sub_end, const uint32_t* __restrict__ cpre,
                                      const uint64_t* __restrict__ bits, uint32_t P, uint32_t N, uint32_t* __restrict__ sim) {
    extern __shared__ uint32_t hist[];
    if (LDS_HIST) {
        for (uint32_t s = threadIdx.x; s < N; s += blockDim.x) hist[s] = 0;
        __syncthreads();
    }
    uint32_t* acc = LDS_HIST ? hist : sim;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < P; r += gridDim.x * blockDim.x) {
        const uint32_t H = cpre[sub_end[r]] - cpre[r];
        if (!H) continue;
        const uint4 m = meta[r];
        const uint32_t l = m.y;
        if (!l) continue;
        uint32_t id = m.z;
        if (l > 1) {
            uint64_t pos = bitpos[r];
            uint32_t sum = 0;
            for (uint32_t i = 0; i + 1 < l; ++i) sum += n2a_gamma(bits, pos);
            id = m.z - sum;
            pos = bitpos[r];
            for (uint32_t i = 0; i + 1 < l; ++i) {
                atomicAdd(&acc[id], H);
                id += n2a_gamma(bits, pos);
            }
        }
        atomicAdd(&acc[id], H);
    }
    if (LDS_HIST) {
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < N; s += blockDim.x)
            if (hist[s]) atomicAdd(&sim[s], hist[s]);
    }
}

}  // namespace

extern "C" int kmdb_new2all_batch(kmdb_db* dbh, const uint64_t* const* kmers, const size_t* counts, size_t nq,
                                  uint32_t* out_dense, const kmdb_opts* opts) {
    if (!dbh || (nq && (!kmers || !counts || !out_dense))) return kmdb_set_error("kmdb_new2all_batch: null argument");
    kmdb_engine_view e;
    kmdb_engine_get(dbh, &e);
    if (!e.n_buckets || !e.slots) return kmdb_set_error("kmdb_new2all_batch: database was uploaded without hashtables");
    HIP_TRY(hipSetDevice(e.device));
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : (hipStream_t)e.stream;
    const uint64_t N = e.N, P = e.P;
    size_t max_q = 0;
    for (size_t q = 0; q < nq; ++q) max_q = std::max(max_q, counts[q]);

    uint64_t* d_k = nullptr;
    uint32_t *d_cnt = nullptr, *d_cpre = nullptr, *d_sim = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    auto cleanup = [&]() {
        for (void* p : {(void*)d_k, (void*)d_cnt, (void*)d_cpre, (void*)d_sim, d_tmp}) if (p) (void)hipFree(p);
    };
#define N2_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    N2_TRY(hipMalloc((void**)&d_k, std::max<size_t>(max_q, 1) * 8));
    N2_TRY(hipMalloc((void**)&d_cnt, (P + 1) * 4));
    N2_TRY(hipMalloc((void**)&d_cpre, (P + 1) * 4));
    N2_TRY(hipMalloc((void**)&d_sim, std::max<uint64_t>(nq * N, 1) * 4));
    hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_cpre, (int)(P + 1), st);
    N2_TRY(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
    N2_TRY(hipMemsetAsync(d_sim, 0, std::max<uint64_t>(nq * N, 1) * 4, st));

    hipEvent_t ev0 = (hipEvent_t)e.ev[0], ev3 = (hipEvent_t)e.ev[3];
    N2_TRY(hipEventRecord(ev0, st));
    const bool lds_hist = N * 4 <= 64 * 1024;
    const unsigned acc_blocks = (unsigned)std::min<uint64_t>(2048, (P + 255) / 256 ? (P + 255) / 256 : 1);
    for (size_t q = 0; q < nq; ++q) {
        const size_t n = counts[q];
        N2_TRY(hipMemsetAsync(d_cnt, 0, (P + 1) * 4, st));
        if (n) {
            N2_TRY(hipMemcpyAsync(d_k, kmers[q], n * 8, hipMemcpyHostToDevice, st));
            const unsigned blocks = (unsigned)std::min<size_t>(4096, (n + 255) / 256);
            hipLaunchKernelGGL(n2a_lookup_kernel, dim3(blocks), dim3(256), 0, st, d_k, n, e.n_buckets, e.bucket_offset,
                               e.slots, e.pid2dfs, e.w, d_cnt);
        }
        N2_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_cnt, d_cpre, (int)(P + 1), st));
        if (P) {
            if (lds_hist)
                hipLaunchKernelGGL(n2a_accumulate_kernel<true>, dim3(acc_blocks), dim3(256), N * 4, st, e.meta, e.bitpos,
                                   e.sub_end, d_cpre, e.bits, (uint32_t)P, (uint32_t)N, d_sim + q * N);
            else
                hipLaunchKernelGGL(n2a_accumulate_kernel<false>, dim3(acc_blocks), dim3(256), 0, st, e.meta, e.bitpos,
                                   e.sub_end, d_cpre, e.bits, (uint32_t)P, (uint32_t)N, d_sim + q * N);
        }
        N2_TRY(hipGetLastError());
    }
    N2_TRY(hipEventRecord(ev3, st));
    N2_TRY(hipEventSynchronize(ev3));
    float ms = 0;
    N2_TRY(hipEventElapsedTime(&ms, ev0, ev3));
    kmdb_engine_set_times(dbh, ms, ms);
    if (nq * N) N2_TRY(hipMemcpy(out_dense, d_sim, nq * N * 4, hipMemcpyDeviceToHost));
#undef N2_TRY
    cleanup();
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
