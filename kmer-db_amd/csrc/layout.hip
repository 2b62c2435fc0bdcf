// layout.hip — kmdb_db_upload: the on-disk pattern section (pid order, parent links) becomes the HBM layout of the
// engine (DFS pre-order, bit-packed streams).  A pure format conversion: nothing here decodes a sample id.
//
//   host   (threads)  narrow the header fields, validate them, pack the 128-bit padded gamma streams back to back
//                     in pid order (reference layout: src/pattern.cpp:15-46, streams src/elias_gamma.h:113-125)
//   device            subtree sizes (leaf climb with child counters), children grouped by parent (stable radix sort:
//                     pid order inside a family), pre-order offsets among siblings (scan), pre-order index / depth of
//                     every node by pointer doubling along the parent links, then one gather into DFS order and a
//                     bit-exact re-pack of the streams
// parent_id[p] < p (patterns are appended, reference src/prefix_kmer_db.cpp:219,357); children keep pid order.
#include "device_common.h"
#include "engine_internal.h"

#include "prim.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>

namespace {

struct U32toU64 { __host__ __device__ uint64_t operator()(uint32_t v) const { return v; } };

// children of every node = one run of the parent-sorted child list: first position and length per parent key
// (key = parent + 1, key 0 = the roots); no atomics (a hot parent has 10^5 children, and same-address atomics crawl)
__global__ void lay_child_runs_kernel(const uint32_t* __restrict__ skeys, uint32_t P, uint32_t* __restrict__ start, uint32_t* __restrict__ cnt) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P) return;
    const uint32_t k = skeys[j];
    if (j == 0 || skeys[j - 1] != k) start[k] = j;
    if (j + 1 == P || skeys[j + 1] != k) cnt[k] = j + 1;          // run end for now; turned into a length below
}
__global__ void lay_child_len_kernel(const uint32_t* __restrict__ start, uint32_t n_keys, uint32_t* __restrict__ cnt) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_keys && cnt[k]) cnt[k] -= start[k];
}
// subtree sizes, one tree level per launch, deepest first: size = 1 + sizes of the children (already final)
__global__ void lay_sizes_level_kernel(const uint32_t* __restrict__ dep, uint32_t level, const uint32_t* __restrict__ start, const uint32_t* __restrict__ cnt,
                                       const uint32_t* __restrict__ schild, uint32_t P, uint32_t* __restrict__ size) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P || dep[i] != level) return;
    const uint32_t c = cnt[i + 1], s0 = start[i + 1];
    uint32_t sz = 1;
    for (uint32_t j = 0; j < c; ++j) sz += size[schild[s0 + j]];
    size[i] = sz;
}

__global__ void lay_keys_kernel(const int32_t* __restrict__ parent, uint32_t P, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) { keys[i] = (uint32_t)(parent[i] + 1); vals[i] = i; }
}
__global__ void lay_fill_kernel(uint32_t* __restrict__ dst, uint32_t n, uint32_t val) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = val;
}
__global__ void lay_pid2dfs_kernel(const uint32_t* __restrict__ acc, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = acc[i] - 1u;
}
__global__ void lay_gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
// offset of a child inside its parent's subtree: 1 + sizes of the earlier siblings
__global__ void lay_rel_kernel(const uint32_t* __restrict__ sorted_keys, const uint32_t* __restrict__ sorted_child, const uint32_t* __restrict__ S,
                               const uint32_t* __restrict__ child_begin, uint32_t P, uint32_t* __restrict__ acc) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P) return;
    acc[sorted_child[j]] = 1u + S[j] - S[child_begin[sorted_keys[j]]];
}
// one round of pointer doubling: the sum of `val` along the path to the current ancestor pointer
__global__ void lay_jump_kernel(const uint32_t* __restrict__ val_in, const int32_t* __restrict__ anc_in, uint32_t P,
                                uint32_t* __restrict__ val_out, int32_t* __restrict__ anc_out, uint32_t* __restrict__ active) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int32_t a = anc_in[i];
    uint32_t v = val_in[i];
    int32_t na = -1;
    if (a >= 0) {
        v += val_in[a]; na = anc_in[a];
        if (na >= 0) *active = 1u;
    }
    val_out[i] = v; anc_out[i] = na;
}
__global__ void lay_order_kernel(const uint32_t* __restrict__ acc, uint32_t P, uint32_t* __restrict__ order, uint32_t* __restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t pre = acc[i] - 1u;                       // the virtual root sits at -1
    if (pre >= P) { *bad = 1u; return; }
    order[pre] = i;
}
struct LayStats { unsigned long long alg, upd, pairs; uint32_t max_n, max_depth, bad, n_long; };
__global__ void lay_gather_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ acc, const uint32_t* __restrict__ dep,
                                  const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ size, const int32_t* __restrict__ parent,
                                  const uint32_t* __restrict__ h_l, const uint32_t* __restrict__ h_last, const uint32_t* __restrict__ h_n, const uint32_t* __restrict__ h_nbits,
                                  const uint32_t* __restrict__ h_w, const unsigned long long* __restrict__ h_wfull, uint32_t P, uint32_t short_ids,
                                  uint2* __restrict__ k0in, uint32_t* __restrict__ nl, int32_t* __restrict__ dparent, uint32_t* __restrict__ w,
                                  uint16_t* __restrict__ dflag, uint32_t* __restrict__ sub_end, LayStats* __restrict__ st) {
    unsigned long long alg = 0, upd = 0, pairs = 0;
    uint32_t mn = 0, md = 0, nlong = 0;
    // grid-stride: the statistics end in one atomic per block and counter, so the grid stays small
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= P; i += gridDim.x * blockDim.x) {
        if (i == P) { w[P] = 0; break; }
        const uint32_t pid = order[i];
        const uint32_t l = h_l[pid], n = h_n[pid], nb = h_nbits[pid];
        k0in[i] = kmdb_k0_pack(l, h_last[pid], nb);
        nl[i] = n;
        const int32_t par = parent[pid];
        dparent[i] = par < 0 ? -1 : (int32_t)(acc[par] - 1u);
        w[i] = h_w[pid];
        const uint32_t d = dep[pid];
        dflag[i] = (uint16_t)((d > 0x7FFFu ? 0x7FFFu : d) | (cnt[pid + 1] ? 0x8000u : 0u));
        sub_end[i] = i + size[pid];
        alg += 40ull + (unsigned long long)((nb + 127u) / 128u) * 16ull;
        upd += (unsigned long long)(n - l) * l + (unsigned long long)l * (l ? l - 1 : 0) / 2;
        pairs += (h_wfull ? h_wfull[pid] : (unsigned long long)h_w[pid]) * ((unsigned long long)n * (n ? n - 1 : 0) / 2);
        mn = max(mn, n); md = max(md, d); nlong += kmdb_long_node(l, nb, short_ids) ? 1u : 0u;
    }
    // block reduction of the statistics
    __shared__ unsigned long long s_alg[256], s_upd[256], s_pairs[256];
    __shared__ uint32_t s_mn[256], s_md[256], s_nl[256];
    s_alg[threadIdx.x] = alg; s_upd[threadIdx.x] = upd; s_pairs[threadIdx.x] = pairs; s_mn[threadIdx.x] = mn; s_md[threadIdx.x] = md; s_nl[threadIdx.x] = nlong;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_alg[threadIdx.x] += s_alg[threadIdx.x + s]; s_upd[threadIdx.x] += s_upd[threadIdx.x + s]; s_pairs[threadIdx.x] += s_pairs[threadIdx.x + s];
            s_mn[threadIdx.x] = max(s_mn[threadIdx.x], s_mn[threadIdx.x + s]); s_md[threadIdx.x] = max(s_md[threadIdx.x], s_md[threadIdx.x + s]);
            s_nl[threadIdx.x] += s_nl[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&st->alg, s_alg[0]); atomicAdd(&st->upd, s_upd[0]); atomicAdd(&st->pairs, s_pairs[0]);
        atomicMax(&st->max_n, s_mn[0]); atomicMax(&st->max_depth, s_md[0]); atomicAdd(&st->n_long, s_nl[0]);
    }
}
__global__ void lay_nbits_dfs_kernel(const uint2* __restrict__ k0in, uint32_t P, uint32_t* __restrict__ nb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) nb[i] = kmdb_k0_bits(k0in[i]);
}
__global__ void lay_blkbase_kernel(const uint64_t* __restrict__ dstpos, uint32_t P, uint64_t* __restrict__ blkbase, uint32_t* __restrict__ bitrel) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint64_t b = dstpos[i & ~255u];
    if ((i & 255u) == 0) blkbase[i >> 8] = b;
    bitrel[i] = (uint32_t)(dstpos[i] - b);
}
// re-pack: the stream of DFS node i moves from its pid-order position to its DFS-order position (both bit-granular,
// MSB-first inside little-endian uint64 words)
__global__ void lay_copy_bits_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ srcpos, const uint64_t* __restrict__ dstpos,
                                     const uint2* __restrict__ k0in, const uint64_t* __restrict__ src, uint32_t P, unsigned long long* __restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t nb = kmdb_k0_bits(k0in[i]);
    if (!nb) return;
    uint64_t sp = srcpos[order[i]], dp = dstpos[i];
    for (uint32_t done = 0; done < nb; done += 64) {
        const uint32_t take = nb - done < 64u ? nb - done : 64u;
        const uint32_t ss = (uint32_t)(sp & 63u);
        uint64_t chunk = src[sp >> 6] << ss;
        if (ss && take > 64u - ss) chunk |= src[(sp >> 6) + 1] >> (64u - ss);
        if (take < 64u) chunk &= ~0ull << (64u - take);
        const uint32_t ds = (uint32_t)(dp & 63u);
        atomicOr(&dst[dp >> 6], (unsigned long long)(chunk >> ds));
        if (ds && take > 64u - ds) atomicOr(&dst[(dp >> 6) + 1], (unsigned long long)(chunk << (64u - ds)));
        sp += take; dp += take;
    }
}
// long nodes: selected in DFS order (stable partition), then ordered by work
struct LongNodePred {
    const uint2* k0in;
    uint32_t short_ids;
    __host__ __device__ bool operator()(uint32_t i) const { const uint2 km = k0in[i]; return kmdb_long_node(kmdb_k0_l(km), kmdb_k0_bits(km), short_ids); }
};
__global__ void lay_long_keys_kernel(const uint2* __restrict__ k0in, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ keys) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint2 km = k0in[idx[t]];
    const uint32_t l = kmdb_k0_l(km);
    keys[t] = kmdb_k0_bits(km) - (l ? l - 1u : 0u);                     // work of a node ~ stream bits beyond one per delta
}
// root path of the first node of every slice, root first
__global__ void lay_seg_anc_kernel(const int32_t* __restrict__ parent, const uint16_t* __restrict__ dflag, uint32_t P, uint32_t nseg_nodes, uint32_t n_segs,
                                   uint32_t chain_cap, uint32_t* __restrict__ anc, uint32_t* __restrict__ anc_n) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    int32_t cur = parent[(size_t)s * nseg_nodes];
    uint32_t d = cur < 0 ? 0u : (uint32_t)(dflag[cur] & 0x7FFFu);
    if (d > chain_cap) d = 0;                                  // deeper than the chain table: the pipeline is not used
    anc_n[s] = d;
    while (cur >= 0 && d) { anc[(size_t)s * chain_cap + (--d)] = (uint32_t)cur; cur = parent[cur]; }
}

// Host staging buffers of the upload: anonymous mappings, handed to the database handle and given back piece by piece by a helper
// thread after the first call (kmdb_release_staging).  Unmapping them right after the copies took 0.31 s of a 0.63 s upload
// (the HIP runtime has the ranges registered for its DMA), and a thread that does it meanwhile blocks every hipMalloc of the
// main thread for as long (both need the address-space lock).  (Transparent huge pages cut both the faults and
// the munmap 3 x, but one upload in ten on a fresh box then stalled 5 s in the kernel's page compaction: not used.)
struct HostRegion { void* p; size_t bytes; };
template <class T>
struct HostBuf {
    T* p = nullptr;
    size_t bytes = 0;
    explicit HostBuf(size_t n, bool zero = false) {
        (void)zero;                                             // anonymous mappings are zero-filled
        bytes = (std::max<size_t>(n, 1) * sizeof(T) + 4095) & ~(size_t)4095;
        void* q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        p = q == MAP_FAILED ? nullptr : (T*)q;
    }
    ~HostBuf() { reset(); }
    void reset() { if (p) (void)munmap(p, bytes); p = nullptr; }
    HostRegion release() { HostRegion r{p, bytes}; p = nullptr; return r; }
    T& operator[](size_t i) { return p[i]; }
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
};

template <class T>
struct DevTmp {
    T* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { HIP_TRY(hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T))); return 0; }
    void reset() { if (p) (void)hipFree(p); p = nullptr; }
};

}  // namespace

void kmdb_release_staging(kmdb_db* db) {
    if (db->staging.empty()) return;
    std::vector<std::pair<void*, size_t>> regions;
    regions.swap(db->staging);
    const bool unmap = !db->one_shot;
    if (db->staging_thread.joinable()) db->staging_thread.join();
    if (!unmap) db->staging_kept.insert(db->staging_kept.end(), regions.begin(), regions.end());      // unmapped by kmdb_db_free (they used to leak)
    db->staging_thread = std::thread([regions, unmap]() {
        // the pages first, on several threads under the shared address-space lock (kmdb_drop_pages); what munmap then holds the lock
        // exclusively for is the bookkeeping of empty ranges.  A one-shot handle (the front-end's) leaves even that to the end of the process.
        kmdb_drop_pages(regions, 8);
        if (unmap)
            for (const auto& r : regions) (void)munmap(r.first, r.second);
    });                                                       // joined by kmdb_db_settle / kmdb_db_free
}

int kmdb_layout_upload(kmdb_db* db, const kmdb_db_view* v, int with_hashtables, uint32_t shard_index, uint32_t shard_count, kmdb_shard_plan* plan) {
    const uint64_t P_view = v->n_patterns;
    // A prefix shard planned on the host (host_shards.cpp) lays out only the nodes it keeps: the fields of those nodes are narrowed, their
    // streams packed, and nothing else crosses PCIe — no hashtable slots, no node of another shard's part of the tree.  (A shard that owns
    // no k-mer at all keeps the whole tree with zero weights: every array stays non-empty.)
    // (An upload that carries the hashtables keeps the whole tree: new2all's pattern ids must all resolve.)
    const bool pruned = plan && shard_count > 1 && !with_hashtables && !getenv("KMDB_SHARD_WHOLE_TREE") && plan->kept[shard_index] > 0 && plan->kept[shard_index] < P_view;
    uint64_t P = pruned ? plan->kept[shard_index] : P_view;    // nodes laid out
    uint64_t h2d_bytes = 0;
    const uint64_t N = v->n_samples;
    const bool verbose = getenv("KMDB_VERBOSE") != nullptr;
    auto tphase0 = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!verbose) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kmdb] upload: %-34s %.3f s\n", what, std::chrono::duration<double>(now - tphase0).count());
        tphase0 = now;
    };
    hipStream_t st = db->stream;
    if (const char* e = getenv("KMDB_SHORT_IDS")) if (*e) db->short_max_ids = (uint32_t)std::max(1, std::min(64, atoi(e)));     // (experiment, round 5)
    const unsigned B = 256;
    unsigned G = (unsigned)((P + B - 1) / B), G1 = (unsigned)((P + 1 + B - 1) / B);

    // ---- host: narrow + validate the header fields, pack the streams in pid order
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::min<uint64_t>(std::min(64u, hw), std::max<uint64_t>(1, P_view / 65536));
    HostBuf<int32_t> h_parent(P);                             // not zero-filled: the pages are first touched by the worker threads
    HostBuf<uint32_t> h_ll(P), h_last(P), h_n(P), h_nbits(P), h_w(P);
    if (!h_parent.p || !h_ll.p || !h_last.p || !h_n.p || !h_nbits.p || !h_w.p) return kmdb_set_error("kmdb_db_upload: out of host memory");
    std::vector<uint64_t> part_bits(T + 1, 0);
    std::atomic<int> bad{0};
    // (the parts are ranges of the VIEW's pattern ids; a pruned shard writes the kept nodes of a part to their places in the shorter arrays)
    auto run_parts = [&](auto&& fn) {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back([&, t] { fn(t, P_view * t / T, P_view * (t + 1) / T); });
        fn(0u, (uint64_t)0, P_view / T);
        for (auto& th : pool) th.join();
    };
    const uint32_t* plan_w = plan && shard_count > 1 ? plan->w[shard_index] : nullptr;     // the shard's own k-mers per pattern
    if (plan && shard_count > 1 && !plan_w) return kmdb_set_error("kmdb_db_upload_shard: the shard is not part of the plan");
    HostBuf<uint32_t> h_newidx(pruned ? P_view : 1);           // pruned: a kept node's index among the kept ones
    std::vector<uint64_t> part_kept(T + 1, 0);
    if (pruned) {
        if (!h_newidx.p) return kmdb_set_error("kmdb_db_upload: out of host memory");
        run_parts([&](unsigned t, uint64_t lo, uint64_t hi) {
            uint64_t c = 0;
            for (uint64_t p = lo; p < hi; ++p) c += plan->keeps(shard_index, p) ? 1u : 0u;
            part_kept[t + 1] = c;
        });
        for (unsigned t = 0; t < T; ++t) part_kept[t + 1] += part_kept[t];
        if (part_kept[T] != P) return kmdb_set_error("kmdb_db_upload_shard: internal: the plan's count of kept nodes is off");
        run_parts([&](unsigned t, uint64_t lo, uint64_t hi) {
            uint64_t o = part_kept[t];
            for (uint64_t p = lo; p < hi; ++p) if (plan->keeps(shard_index, p)) h_newidx[p] = (uint32_t)o++;
        });
    }
    run_parts([&](unsigned t, uint64_t lo, uint64_t hi) {
        uint64_t nbsum = 0, o = pruned ? part_kept[t] : lo;
        for (uint64_t p = lo; p < hi; ++p) {
            if (pruned && !plan->keeps(shard_index, p)) continue;
            const int64_t par = v->parent_id[p];
            const uint32_t n = v->num_samples[p], l = v->num_local[p], nb = v->num_bits[p], last = v->last_sample_id[p];
            if (par >= (int64_t)p) bad = 1;
            if (l > n || n > N || (l && last >= N) || nb >= KMDB_MAX_STREAM_BITS) bad = 2;      // (N < KMDB_MAX_SAMPLES: checked by the caller)
            h_parent[o] = par < 0 ? -1 : (pruned ? (int32_t)h_newidx[par] : (int32_t)par);         // (a kept node's parent is kept)
            h_ll[o] = l;
            h_last[o] = l ? last : 0u;
            h_n[o] = n;
            h_nbits[o] = nb;
            h_w[o] = plan_w ? plan_w[p] : (uint32_t)v->num_kmers[p];     // truncated exactly like the reference's to_add (similarity_calculator.cpp:222)
            nbsum += nb;
            ++o;
        }
        part_bits[t + 1] = nbsum;
    });
    if (bad == 1) return kmdb_set_error("kmdb_db_upload: parent_id >= pattern id");
    if (bad) return kmdb_set_error("kmdb_db_upload: inconsistent pattern header");
    for (unsigned t = 0; t < T; ++t) part_bits[t + 1] += part_bits[t];
    const uint64_t total_bits = part_bits[T];
    const uint64_t n_bit_words = (total_bits + 63) / 64 + 16;       // zero padding words for the cursors' look-ahead
    HostBuf<uint64_t> h_bits(n_bit_words, /*zero=*/true);
    if (!h_bits.p) return kmdb_set_error("kmdb_db_upload: out of host memory");
    run_parts([&](unsigned t, uint64_t lo, uint64_t hi) {
        // streams of different threads can share a word at the range boundaries: OR the words in atomically
        uint64_t pos = part_bits[t];
        for (uint64_t p = lo; p < hi; ++p) {
            if (pruned && !plan->keeps(shard_index, p)) continue;
            const uint32_t nb = v->num_bits[p];
            if (!nb) continue;
            const uint64_t* src = v->data + v->data_offset[p];
            for (uint32_t done = 0; done < nb; done += 64) {
                const uint32_t take = std::min<uint32_t>(64, nb - done);
                uint64_t chunk = src[done >> 6];
                if (take < 64) chunk &= ~0ull << (64 - take);
                const uint32_t sh = (uint32_t)(pos & 63);
                __atomic_fetch_or(&h_bits[pos >> 6], chunk >> sh, __ATOMIC_RELAXED);
                if (sh && take > 64 - sh) __atomic_fetch_or(&h_bits[(pos >> 6) + 1], chunk << (64 - sh), __ATOMIC_RELAXED);
                pos += take;
            }
        }
    });
    phase("host: narrow fields + pack streams");

    // ---- H2D
    DevTmp<int32_t> d_parent;
    DevTmp<uint32_t> d_ll, d_last, d_n, d_nbits, d_w;
    DevTmp<uint64_t> d_src;
    if (d_parent.alloc(P) || d_ll.alloc(P) || d_last.alloc(P) || d_n.alloc(P) || d_nbits.alloc(P) || d_w.alloc(P) || d_src.alloc(n_bit_words)) return 1;
    HIP_TRY(hipMemcpyAsync(d_parent.p, h_parent.p, P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_ll.p, h_ll.p, P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_last.p, h_last.p, P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_n.p, h_n.p, P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_nbits.p, h_nbits.p, P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_w.p, h_w.p, P * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_src.p, h_bits.p, n_bit_words * 8, hipMemcpyHostToDevice, st));
    h2d_bytes += P * 24 + n_bit_words * 8;
    DevTmp<unsigned long long> d_wfull;                       // sum_pairs uses the untruncated counts
    uint64_t dev_ht_bytes = 0;
    if (with_hashtables && v->n_buckets) {
        const uint64_t n_slots = v->bucket_offset[v->n_buckets];
        db->n_buckets = v->n_buckets;
        HIP_TRY(hipMalloc((void**)&db->bucket_offset, (v->n_buckets + 1) * 8));
        HIP_TRY(hipMalloc((void**)&db->slots, std::max<uint64_t>(n_slots, 1) * 8));
        HIP_TRY(hipMemcpyAsync(db->bucket_offset, v->bucket_offset, (v->n_buckets + 1) * 8, hipMemcpyHostToDevice, st));
        if (n_slots) HIP_TRY(hipMemcpyAsync(db->slots, v->slots, n_slots * 8, hipMemcpyHostToDevice, st));
        h2d_bytes += (v->n_buckets + 1) * 8 + n_slots * 8;
        dev_ht_bytes = (v->n_buckets + 1) * 8 + n_slots * 8 + P * 4;
    }
    if (shard_count > 1 && !plan) return kmdb_set_error("kmdb_db_upload_shard: internal: a prefix shard without a plan");
    // (prefix-bucket shard: the weights came from the plan — the shard's own k-mer counts, partial matrices of all shards sum to the full
    // one — and, unless the upload carries the hashtables (new2all's pattern ids must all resolve: whole tree), only the kept nodes)
    HIP_TRY(hipStreamSynchronize(st));
    phase("H2D");
    if (pruned && verbose) fprintf(stderr, "[kmdb] upload: prefix shard %u / %u keeps %llu of %llu patterns (%.1f MB over PCIe)\n", shard_index, shard_count,
                                   (unsigned long long)P, (unsigned long long)P_view, h2d_bytes / 1e6);
    if (pruned) db->P = P;
    if (plan && shard_count > 1) plan->release_weights(shard_index);
    h_newidx.reset();
    for (HostRegion r : {h_parent.release(), h_ll.release(), h_last.release(), h_n.release(), h_nbits.release(), h_w.release(), h_bits.release()})
        if (r.p) db->staging.emplace_back(r.p, r.bytes);

    // ---- device: DFS pre-order
    DevTmp<uint32_t> cnt, cstart, size, keys, vals, skeys, schild, ssz, S, acc[2], dep[2], order, flags;
    DevTmp<int32_t> anc[2];
    if (cnt.alloc(P + 2) || cstart.alloc(P + 2) || size.alloc(P) || flags.alloc(4)) return 1;
    phase("  first device temporaries");
    HIP_TRY(hipMemsetAsync(cnt.p, 0, (P + 2) * 4, st));
    HIP_TRY(hipMemsetAsync(cstart.p, 0, (P + 2) * 4, st));
    HIP_TRY(hipMemsetAsync(flags.p, 0, 16, st));
    // children grouped by parent, pid order inside a family
    if (keys.alloc(P) || vals.alloc(P) || skeys.alloc(P) || schild.alloc(P)) return 1;
    HIP_TRY(hipStreamSynchronize(st));
    phase("  temporaries (hipMalloc)");
    hipLaunchKernelGGL(lay_keys_kernel, dim3(G), dim3(B), 0, st, d_parent.p, (uint32_t)P, keys.p, vals.p);
    {
        int end_bit = 1;
        while ((1ull << end_bit) <= P) ++end_bit;
        size_t tb = 0;
        HIP_TRY(prim::sort_pairs(nullptr, tb, keys.p, skeys.p, vals.p, schild.p, (int)P, 0, end_bit, st));
        DevTmp<unsigned char> tmp;
        if (tmp.alloc(tb)) return 1;
        HIP_TRY(hipStreamSynchronize(st));
        phase("  keys + sort temporaries");
        HIP_TRY(prim::sort_pairs(tmp.p, tb, keys.p, skeys.p, vals.p, schild.p, (int)P, 0, end_bit, st));
        HIP_TRY(hipStreamSynchronize(st));
        phase("  radix sort");
    }
    keys.reset(); vals.reset();
    hipLaunchKernelGGL(lay_child_runs_kernel, dim3(G), dim3(B), 0, st, skeys.p, (uint32_t)P, cstart.p, cnt.p);
    hipLaunchKernelGGL(lay_child_len_kernel, dim3(G1), dim3(B), 0, st, cstart.p, (uint32_t)(P + 1), cnt.p);
    phase("  sort by parent");
    // depth of every node: pointer doubling along the parent links (ping-pong buffers)
    for (int k = 0; k < 2; ++k) if (acc[k].alloc(P) || dep[k].alloc(P) || anc[k].alloc(P)) return 1;
    auto jump_all = [&](DevTmp<uint32_t>* val, int* cur_out) -> int {
        int cur = 0;
        HIP_TRY(hipMemcpyAsync(anc[0].p, d_parent.p, P * 4, hipMemcpyDeviceToDevice, st));
        for (int round = 0; round < 40; ++round) {
            HIP_TRY(hipMemsetAsync(flags.p, 0, 4, st));
            hipLaunchKernelGGL(lay_jump_kernel, dim3(G), dim3(B), 0, st, val[cur].p, anc[cur].p, (uint32_t)P, val[cur ^ 1].p, anc[cur ^ 1].p, flags.p);
            uint32_t active = 0;
            HIP_TRY(hipMemcpyAsync(&active, flags.p, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            cur ^= 1;
            if (!active) break;
        }
        *cur_out = cur;
        return 0;
    };
    int dcur = 0;
    hipLaunchKernelGGL(lay_fill_kernel, dim3(G), dim3(B), 0, st, dep[0].p, (uint32_t)P, 1u);
    if (jump_all(dep, &dcur)) return 1;
    dep[dcur ^ 1].reset();
    uint32_t max_depth = 0;
    {
        DevTmp<uint32_t> dmax;
        if (dmax.alloc(1)) return 1;
        size_t tb = 0;
        HIP_TRY(prim::max(nullptr, tb, dep[dcur].p, dmax.p, (int)P, st));
        DevTmp<unsigned char> tmp;
        if (tmp.alloc(tb)) return 1;
        HIP_TRY(prim::max(tmp.p, tb, dep[dcur].p, dmax.p, (int)P, st));
        HIP_TRY(hipMemcpyAsync(&max_depth, dmax.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    phase("  depth");
    for (uint32_t level = max_depth; level >= 1; --level)
        hipLaunchKernelGGL(lay_sizes_level_kernel, dim3(G), dim3(B), 0, st, dep[dcur].p, level, cstart.p, cnt.p, schild.p, (uint32_t)P, size.p);
    HIP_TRY(hipGetLastError());
    phase("  subtree sizes");
    // pre-order offsets among siblings, then their sums along the root paths
    if (ssz.alloc(P) || S.alloc(P + 1)) return 1;
    hipLaunchKernelGGL(lay_gather_u32_kernel, dim3(G), dim3(B), 0, st, size.p, schild.p, (uint32_t)P, ssz.p);
    {
        size_t tb1 = 0;
        HIP_TRY(prim::exclusive_sum(nullptr, tb1, ssz.p, S.p, (int)P, st));
        DevTmp<unsigned char> tmp;
        if (tmp.alloc(tb1)) return 1;
        HIP_TRY(prim::exclusive_sum(tmp.p, tb1, ssz.p, S.p, (int)P, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    ssz.reset();
    hipLaunchKernelGGL(lay_rel_kernel, dim3(G), dim3(B), 0, st, skeys.p, schild.p, S.p, cstart.p, (uint32_t)P, acc[0].p);
    HIP_TRY(hipGetLastError());
    int cur = 0;
    if (jump_all(acc, &cur)) return 1;
    skeys.reset(); schild.reset(); S.reset(); cstart.reset();
    anc[0].reset(); anc[1].reset(); acc[cur ^ 1].reset();
    if (order.alloc(P)) return 1;
    HIP_TRY(hipMemsetAsync(order.p, 0xFF, P * 4, st));
    hipLaunchKernelGGL(lay_order_kernel, dim3(G), dim3(B), 0, st, acc[cur].p, (uint32_t)P, order.p, flags.p + 1);
    phase("device: DFS pre-order");

    // ---- DFS-ordered node arrays
    HIP_TRY(hipMalloc((void**)&db->k0in, std::max<uint64_t>(P, 1) * 8));
    HIP_TRY(hipMalloc((void**)&db->nl, std::max<uint64_t>(P, 1) * 4));
    HIP_TRY(hipMalloc((void**)&db->parent, std::max<uint64_t>(P, 1) * 4));
    HIP_TRY(hipMalloc((void**)&db->w, (P + 1) * 4));
    HIP_TRY(hipMalloc((void**)&db->dflag, std::max<uint64_t>(P, 1) * 2));
    HIP_TRY(hipMalloc((void**)&db->sub_end, std::max<uint64_t>(P, 1) * 4));
    DevTmp<LayStats> d_stats;
    if (d_stats.alloc(1)) return 1;
    HIP_TRY(hipMemsetAsync(d_stats.p, 0, sizeof(LayStats), st));
    if (shard_count <= 1) {
        // the checksum sum_p w_p C(n_p, 2) is defined on the full 64-bit counts
        if (d_wfull.alloc(P)) return 1;
        HIP_TRY(hipMemcpyAsync(d_wfull.p, v->num_kmers, P * 8, hipMemcpyHostToDevice, st));
        h2d_bytes += P * 8;
    }
    hipLaunchKernelGGL(lay_gather_kernel, dim3(std::min<unsigned>(G1, 2048u)), dim3(B), 0, st, order.p, acc[cur].p, dep[dcur].p, cnt.p, size.p, d_parent.p, d_ll.p, d_last.p, d_n.p, d_nbits.p, d_w.p,
                       d_wfull.p, (uint32_t)P, db->short_max_ids, db->k0in, db->nl, db->parent, db->w, db->dflag, db->sub_end, d_stats.p);
    HIP_TRY(hipGetLastError());
    LayStats hs{};
    uint32_t hflags[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(&hs, d_stats.p, sizeof hs, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hflags, flags.p, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (hflags[1]) return kmdb_set_error("kmdb_db_upload: pattern tree is not a forest");
    if (with_hashtables && v->n_buckets) {
        // pid -> DFS index for the hash lookups of new2all
        HIP_TRY(hipMalloc((void**)&db->pid2dfs, std::max<uint64_t>(P, 1) * 4));
        hipLaunchKernelGGL(lay_pid2dfs_kernel, dim3(G), dim3(B), 0, st, acc[cur].p, (uint32_t)P, db->pid2dfs);
        HIP_TRY(hipStreamSynchronize(st));
    }
    d_wfull.reset(); d_ll.reset(); d_last.reset(); d_n.reset(); d_w.reset(); size.reset(); cnt.reset(); dep[dcur].reset(); d_parent.reset();
    phase("device: node arrays");

    // ---- streams re-packed in DFS order
    {
        DevTmp<uint32_t> nb_dfs;
        DevTmp<uint64_t> srcpos, dstpos;
        if (nb_dfs.alloc(P) || srcpos.alloc(P + 1) || dstpos.alloc(P + 1)) return 1;
        hipLaunchKernelGGL(lay_nbits_dfs_kernel, dim3(G), dim3(B), 0, st, db->k0in, (uint32_t)P, nb_dfs.p);
        rocprim::transform_iterator<uint32_t*, U32toU64, uint64_t> it_src(d_nbits.p, U32toU64()), it_dst(nb_dfs.p, U32toU64());
        size_t tb = 0;
        HIP_TRY(prim::exclusive_sum(nullptr, tb, it_src, srcpos.p, (int)P, st));
        DevTmp<unsigned char> tmp;
        if (tmp.alloc(tb)) return 1;
        HIP_TRY(prim::exclusive_sum(tmp.p, tb, it_src, srcpos.p, (int)P, st));
        HIP_TRY(prim::exclusive_sum(tmp.p, tb, it_dst, dstpos.p, (int)P, st));
        HIP_TRY(hipMalloc((void**)&db->bits, n_bit_words * 8));
        HIP_TRY(hipMemsetAsync(db->bits, 0, n_bit_words * 8, st));
        db->n_bit_words = n_bit_words;
        HIP_TRY(hipMalloc((void**)&db->blkbase, ((P + 255) / 256 + 1) * 8));
        HIP_TRY(hipMalloc((void**)&db->bitrel, std::max<uint64_t>(P, 1) * 4));
        hipLaunchKernelGGL(lay_blkbase_kernel, dim3(G), dim3(B), 0, st, dstpos.p, (uint32_t)P, db->blkbase, db->bitrel);
        hipLaunchKernelGGL(lay_copy_bits_kernel, dim3(G), dim3(B), 0, st, order.p, srcpos.p, dstpos.p, db->k0in, d_src.p, (uint32_t)P,
                           (unsigned long long*)db->bits);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
    }
    d_src.reset(); d_nbits.reset(); order.reset();
    phase("device: stream re-pack");

    // ---- per-slice root paths, long nodes
    db->max_n = hs.max_n; db->max_depth = hs.max_depth;
    db->chain_ok = hs.max_depth <= (uint32_t)KMDB_CHAIN_MAX;
    db->chain_cap = std::min<uint32_t>(KMDB_CHAIN_MAX, std::max<uint32_t>(8, (hs.max_depth + 7) / 8 * 8));
    // one wave per slice of 2048 nodes.  (Round 1 used about 8192 slices in all, 12 544 nodes each at the benchmark database,
    // because every wave left its open record chunks half empty; since only the first block's diagonal records go through
    // stream chunks that costs little — measured: narrow kernel 1.83 -> 1.37 ms, chunks 186 k -> 208 k, whole call -0.3 ms;
    // 1024 nodes: no faster, more chunks.)
    db->nseg_nodes = 2048;
    if (const char* e = getenv("KMDB_NSEG")) if (*e) db->nseg_nodes = (uint32_t)std::max<uint64_t>(64, strtoull(e, nullptr, 10) / 64 * 64);
    db->n_nsegs = (uint32_t)((P + db->nseg_nodes - 1) / db->nseg_nodes);
    HIP_TRY(hipMalloc((void**)&db->nseg_anc, std::max<size_t>((size_t)db->n_nsegs * db->chain_cap, 1) * 4));
    HIP_TRY(hipMalloc((void**)&db->nseg_anc_n, std::max<size_t>(db->n_nsegs, 1) * 4));
    if (db->n_nsegs)
        hipLaunchKernelGGL(lay_seg_anc_kernel, dim3((db->n_nsegs + 63) / 64), dim3(64), 0, st, db->parent, db->dflag, (uint32_t)P, db->nseg_nodes, db->n_nsegs,
                           db->chain_cap, db->nseg_anc, db->nseg_anc_n);
    db->n_long = hs.n_long;
    if (hs.n_long) {
        DevTmp<uint32_t> sel, lk, lk2, nsel;
        if (sel.alloc(hs.n_long + 1) || lk.alloc(hs.n_long) || lk2.alloc(hs.n_long) || nsel.alloc(1)) return 1;
        HIP_TRY(hipMalloc((void**)&db->long_nodes, (size_t)hs.n_long * 4));
        rocprim::counting_iterator<uint32_t> first(0u);
        size_t tb = 0;
        HIP_TRY(prim::select_if(nullptr, tb, first, sel.p, nsel.p, (int)P, LongNodePred{db->k0in, db->short_max_ids}, st));
        DevTmp<unsigned char> tmp;
        if (tmp.alloc(tb)) return 1;
        HIP_TRY(prim::select_if(tmp.p, tb, first, sel.p, nsel.p, (int)P, LongNodePred{db->k0in, db->short_max_ids}, st));
        hipLaunchKernelGGL(lay_long_keys_kernel, dim3((hs.n_long + 255) / 256), dim3(256), 0, st, db->k0in, sel.p, hs.n_long, lk.p);
        // most work first; the sort is stable, so equal work keeps ascending DFS order
        size_t tb2 = 0;
        HIP_TRY(prim::sort_pairs_desc(nullptr, tb2, lk.p, lk2.p, sel.p, db->long_nodes, (int)hs.n_long, 0, 32, st));
        DevTmp<unsigned char> tmp2;
        if (tmp2.alloc(tb2)) return 1;
        HIP_TRY(prim::sort_pairs_desc(tmp2.p, tb2, lk.p, lk2.p, sel.p, db->long_nodes, (int)hs.n_long, 0, 32, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    phase("device: slices + long nodes");

    db->stats.algorithmic_bytes = hs.alg + 4ull * (N ? N * (N - 1) / 2 : 0);
    db->stats.tree_updates = hs.upd;
    db->stats.sum_pairs = hs.pairs;
    db->stats.n_segments = db->n_nsegs;
    db->stats.n_patterns = P;
    db->stats.h2d_bytes = h2d_bytes;
    db->stats.device_bytes = P * (8 + 4 + 4 + 4 + 4 + 2 + 4) + n_bit_words * 8 + (uint64_t)db->n_nsegs * db->chain_cap * 4 + dev_ht_bytes;
    return 0;
}
