// engine.hip — MI355X (gfx950) engine behind include/kmdb_amd.h.
//
// Replaces the reference's SimilarityCalculator::all2all / all2all_sp
// (reference src/similarity_calculator.cpp:42-438, 442-657) with a design built for CDNA4:
//
//  * HBM layout (kmdb_db_upload).  The pattern tree is re-laid in DFS PRE-ORDER.  Then
//      - subtree(p) is the contiguous index range [p, sub_end[p]), so the subtree weights the
//        reference accumulates bottom-up (:64-72) are two reads of an exclusive prefix sum;
//      - the full sample list of node p is "the list of the previous node, truncated to
//        n_p - l_p entries, plus p's own l_p local ids": a wave that walks a contiguous DFS
//        range keeps ONE stack of decoded ids and decodes every gamma stream exactly once
//        (the reference re-decodes the whole parent chain per pattern, :126-152);
//      - node headers are 16-byte records {n, l, last_id, nbits} read fully coalesced, and
//        all gamma streams are bit-packed back to back (no 128-bit padding as on disk).
//  * Kernel a2a_tile_kernel.  One wave = one equal-cost segment of the DFS stream.
//      lanes decode 64 nodes' gamma streams in parallel (lane-per-node) into LDS, then the
//      wave replays the nodes in order: truncate/extend the id stack, and for every local id
//      (= matrix row) add the subtree weight to the cells of all earlier ids (= columns):
//      the GPU form of row_add (reference src/simd/row_add_avx2.cpp:30-124).  Updates go to a
//      wave-private lower-triangular TILE in LDS indexed by compact sample indices (the set
//      of samples a DFS neighbourhood touches is small); the tile is written back to the
//      N x N matrix in HBM with one global atomic per non-zero cell when the compact index
//      space overflows or the segment ends.  Lists longer than the tile side go straight to
//      HBM atomics.
//  * Kernel a2a_global_kernel: same walk with the stack in global scratch and plain HBM
//      atomics — any N, used for N > 4096 and as the debugging fallback.
//  No MFMA: this is integer scatter/histogram work (BASELINE.json north_star).
//  uint32 adds wrap and commute, so any schedule is bit-exact with the reference.
#include "device_common.h"
#include "engine_internal.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
int kmdb_set_error(const std::string& msg) { g_last_error = msg; return 1; }
extern "C" const char* kmdb_last_error(void) { return g_last_error.c_str(); }
extern "C" int kmdb_abi_version(void) { return KMDB_ABI_VERSION; }

extern "C" int kmdb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

namespace {
// ------------------------------------------------------------------------------------------
// dense -> CSR compaction for the sparse entry point
// ------------------------------------------------------------------------------------------
__global__ void row_nnz_kernel(const uint32_t* __restrict__ M, uint64_t N, unsigned long long* __restrict__ row_nnz) {
    const uint64_t row = blockIdx.x;
    const uint32_t* r = M + tri64(row);
    uint32_t c = 0;
    for (uint64_t j = threadIdx.x; j < row; j += blockDim.x) c += r[j] != 0;
    __shared__ uint32_t red[256];
    red[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) row_nnz[row] = red[0];
}

// one block per row, ordered compaction with a block-wide running offset
__global__ void row_compact_kernel(const uint32_t* __restrict__ M, uint64_t N, const unsigned long long* __restrict__ row_ptr,
                                   uint32_t* __restrict__ col, uint32_t* __restrict__ val) {
    const uint64_t row = blockIdx.x;
    const uint32_t* r = M + tri64(row);
    __shared__ uint32_t wave_cnt[4];
    __shared__ unsigned long long running;
    if (threadIdx.x == 0) running = row_ptr[row];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t j0 = 0; j0 < row; j0 += blockDim.x) {
        const uint64_t j = j0 + threadIdx.x;
        const uint32_t v = j < row ? r[j] : 0u;
        const unsigned long long bal = __ballot(v != 0);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t k = 0; k < wave; ++k) before += wave_cnt[k];
        const unsigned long long base = running;
        if (v) {
            const unsigned long long o = base + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            col[o] = (uint32_t)j;
            val[o] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) running = base + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// upload: pid order -> DFS pre-order layout
// ------------------------------------------------------------------------------------------
namespace {

template <class T>
int dev_upload(T** dst, const T* src, size_t n) {
    size_t bytes = std::max<size_t>(1, n) * sizeof(T);
    HIP_TRY(hipMalloc((void**)dst, bytes));
    if (n) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace


extern "C" int kmdb_db_upload(const kmdb_db_view* v, const kmdb_opts* opts, int with_hashtables, kmdb_db** out) {
    *out = nullptr;
    if (!v || v->abi_version != KMDB_ABI_VERSION) return kmdb_set_error("kmdb_db_upload: bad view / ABI version");
    const uint64_t P = v->n_patterns, N = v->n_samples;
    if (P >= (1ull << 31)) return kmdb_set_error("kmdb_db_upload: more than 2^31 patterns");
    if (N > 65535) return kmdb_set_error("kmdb_db_upload: more than 65535 samples is not supported yet");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return kmdb_set_error("kmdb_db_upload: no HIP device available (the engine has no CPU fallback)");
    }
    const int device = opts ? opts->device : 0;
    HIP_TRY(hipSetDevice(device));

    const bool verbose = getenv("KMDB_VERBOSE") != nullptr;
    auto tphase0 = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kmdb] upload: %-28s %.2f s\n", what, std::chrono::duration<double>(now - tphase0).count());
        tphase0 = now;
    };
    // ---- children lists (parent_id[p] < p, SURVEY §7 invariants) ------------------------------
    std::vector<uint32_t> child_count(P + 1, 0), order(P), dfs_of(P);
    std::vector<uint32_t> roots;
    for (uint64_t p = 0; p < P; ++p) {
        int64_t par = v->parent_id[p];
        if (par >= (int64_t)p) return kmdb_set_error("kmdb_db_upload: parent_id >= pattern id");
        if (par < 0) roots.push_back((uint32_t)p); else ++child_count[par];
    }
    std::vector<uint64_t> child_begin(P + 1, 0);
    for (uint64_t p = 0; p < P; ++p) child_begin[p + 1] = child_begin[p] + child_count[p];
    std::vector<uint32_t> children(child_begin[P]);
    {
        std::vector<uint64_t> fill(child_begin.begin(), child_begin.end() - 1);
        for (uint64_t p = 0; p < P; ++p) {
            int64_t par = v->parent_id[p];
            if (par >= 0) children[fill[par]++] = (uint32_t)p;
        }
    }
    std::vector<uint32_t> sub_end(P);
    {
        // iterative pre-order; children in increasing pattern id
        std::vector<std::pair<uint32_t, uint64_t>> st;   // (pid, next child cursor)
        uint32_t idx = 0;
        for (uint32_t r : roots) {
            st.emplace_back(r, child_begin[r]);
            order[idx] = r; dfs_of[r] = idx++;
            while (!st.empty()) {
                auto& top = st.back();
                if (top.second < child_begin[top.first + 1]) {
                    uint32_t c = children[top.second++];
                    order[idx] = c; dfs_of[c] = idx++;
                    st.emplace_back(c, child_begin[c]);
                } else {
                    sub_end[dfs_of[top.first]] = idx;
                    st.pop_back();
                }
            }
        }
        if (idx != P) return kmdb_set_error("kmdb_db_upload: pattern tree is not a forest");
    }
    std::vector<uint32_t>().swap(children);

    phase("DFS order");
    // ---- DFS-ordered arrays, bit-packed streams, cost model ----------------------------------
    // (host threads over contiguous DFS ranges: the gathers through `order` are random reads of the view)
    std::vector<uint4> meta(P);
    std::vector<uint64_t> bitpos(P + 1, 0);
    std::vector<int32_t> parent(P);
    std::vector<uint32_t> w(P + 1, 0);
    std::vector<uint64_t> cost_prefix(P + 1, 0);
    uint64_t alg_bytes = 0, tree_updates = 0, sum_pairs = 0;
    uint32_t max_n = 0;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::min<uint64_t>(std::min(32u, hw), std::max<uint64_t>(1, P / 65536));
    struct Part { uint64_t alg = 0, upd = 0, pairs = 0; uint32_t max_n = 0; bool bad = false; };
    std::vector<Part> parts(T);
    auto run_parts = [&](auto&& fn) {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back([&, t] { fn(t, P * t / T, P * (t + 1) / T); });
        fn(0u, (uint64_t)0, P / T);
        for (auto& th : pool) th.join();
    };
    run_parts([&](unsigned t, uint64_t lo, uint64_t hi) {
        Part& pt = parts[t];
        for (uint64_t i = lo; i < hi; ++i) {
            const uint32_t pid = order[i];
            const uint32_t n = v->num_samples[pid], l = v->num_local[pid], nb = v->num_bits[pid];
            if (l > n || n > N) { pt.bad = true; continue; }
            meta[i] = make_uint4(n, l, v->last_sample_id[pid], nb);
            pt.max_n = std::max(pt.max_n, n);
            bitpos[i + 1] = nb;                               // scanned below
            const int64_t par = v->parent_id[pid];
            parent[i] = par < 0 ? -1 : (int32_t)dfs_of[par];
            w[i] = (uint32_t)v->num_kmers[pid];
            pt.upd += (uint64_t)(n - l) * l + (uint64_t)l * (l ? l - 1 : 0) / 2;
            pt.pairs += (uint64_t)v->num_kmers[pid] * ((uint64_t)n * (n ? n - 1 : 0) / 2);
            pt.alg += 40 + (uint64_t)((nb + 127) / 128) * 16;
            // per-node cost in "wave instructions": decode share + one scatter instruction per 64 columns per row
            uint64_t rows_cost = 0;
            if (l) {
                // sum over t in [n-l, n) of (t/64 + 1)
                for (uint32_t blk = (n - l) / 64; blk <= (n - 1) / 64; ++blk) {
                    uint32_t lo2 = std::max<uint32_t>(n - l, blk * 64), hi2 = std::min<uint32_t>(n, blk * 64 + 64);
                    rows_cost += (uint64_t)(hi2 - lo2) * (blk + 1);
                }
            }
            cost_prefix[i + 1] = 4 + l / 2 + rows_cost * 2;   // scanned below
        }
    });
    for (const Part& pt : parts) {
        if (pt.bad) return kmdb_set_error("kmdb_db_upload: inconsistent pattern header");
        alg_bytes += pt.alg; tree_updates += pt.upd; sum_pairs += pt.pairs; max_n = std::max(max_n, pt.max_n);
    }
    for (uint64_t i = 0; i < P; ++i) { bitpos[i + 1] += bitpos[i]; cost_prefix[i + 1] += cost_prefix[i]; }
    const uint64_t total_bits = bitpos[P];
    std::vector<uint64_t> bits((total_bits + 63) / 64 + 16, 0);    // zero padding words for the cursors' look-ahead
    run_parts([&](unsigned, uint64_t lo, uint64_t hi) {
        // streams of different threads can share a word at the range boundaries: OR the words in atomically
        for (uint64_t i = lo; i < hi; ++i) {
            const uint32_t nb = meta[i].w;
            if (!nb) continue;
            const uint64_t* src = v->data + v->data_offset[order[i]];
            uint64_t pos = bitpos[i];
            for (uint32_t done = 0; done < nb; done += 64) {
                const uint32_t take = std::min<uint32_t>(64, nb - done);
                uint64_t chunk = src[done >> 6];
                if (take < 64) chunk &= ~0ull << (64 - take);
                const uint32_t sh = (uint32_t)(pos & 63);
                __atomic_fetch_or(&bits[pos >> 6], chunk >> sh, __ATOMIC_RELAXED);
                if (sh && take > 64 - sh) __atomic_fetch_or(&bits[(pos >> 6) + 1], chunk << (64 - sh), __ATOMIC_RELAXED);
                pos += take;
            }
        }
    });
    bitpos.pop_back();
    alg_bytes += 4ull * (N ? N * (N - 1) / 2 : 0);

    phase("node arrays + bit packing");
    // ---- equal-cost segments -------------------------------------------------------------------
    uint32_t want = (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(1, P / 48));
    want = (want + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK * WAVES_PER_BLOCK;
    std::vector<Segment> segs;
    {
        const uint64_t total = cost_prefix[P];
        uint64_t start = 0;
        for (uint32_t s = 0; s < want && start < P; ++s) {
            uint64_t target = total / want * (s + 1);
            if (s + 1 == want) target = total;
            uint64_t e = std::upper_bound(cost_prefix.begin() + start + 1, cost_prefix.end(), target) - cost_prefix.begin() - 1;
            e = std::max<uint64_t>(e, start + 1);
            e = std::min<uint64_t>(e, P);
            if (s + 1 == want) e = P;
            segs.push_back({(uint32_t)start, (uint32_t)e});
            start = e;
        }
        if (segs.empty()) segs.push_back({0u, (uint32_t)P});
        else segs.back().end = (uint32_t)P;
    }

    auto* db = new kmdb_db();
    db->device = device; db->N = N; db->P = P; db->kmer_length = v->kmer_length;
    db->n_bit_words = bits.size();
    db->n_segs = (uint32_t)segs.size();
    int rc = 0;
    phase("segments");
    rc |= dev_upload(&db->meta, meta.data(), P);
    rc |= dev_upload(&db->bitpos, bitpos.data(), P);
    rc |= dev_upload(&db->parent, parent.data(), P);
    rc |= dev_upload(&db->w, w.data(), P + 1);
    rc |= dev_upload(&db->sub_end, sub_end.data(), P);
    rc |= dev_upload(&db->bits, bits.data(), bits.size());
    rc |= dev_upload(&db->segs, segs.data(), segs.size());
    if (rc) { kmdb_db_free(db); return 1; }
    if (hipMalloc((void**)&db->wprefix, (P + 1) * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&db->counters, 8 * sizeof(unsigned long long)) != hipSuccess) {
        kmdb_db_free(db);
        return kmdb_set_error("kmdb_db_upload: out of device memory");
    }
    hipcub::DeviceScan::ExclusiveSum(nullptr, db->scan_tmp_bytes, db->w, db->wprefix, (int)(P + 1));
    if (hipMalloc(&db->scan_tmp, std::max<size_t>(db->scan_tmp_bytes, 16)) != hipSuccess) {
        kmdb_db_free(db);
        return kmdb_set_error("kmdb_db_upload: out of device memory");
    }
    uint64_t dev_bytes = P * (16 + 8 + 4 + 4 + 4 + 4) + bits.size() * 8 + segs.size() * 8;
    if (with_hashtables && v->n_buckets) {
        db->n_buckets = v->n_buckets;
        rc |= dev_upload(&db->bucket_offset, v->bucket_offset, v->n_buckets + 1);
        rc |= dev_upload(&db->slots, v->slots, v->bucket_offset[v->n_buckets]);
        rc |= dev_upload(&db->pid2dfs, dfs_of.data(), P);
        if (rc) { kmdb_db_free(db); return 1; }
        dev_bytes += (v->n_buckets + 1) * 8 + v->bucket_offset[v->n_buckets] * 8 + P * 4;
    }
    if (hipStreamCreate(&db->stream) != hipSuccess) { kmdb_db_free(db); return kmdb_set_error("hipStreamCreate failed"); }
    for (auto& e : db->ev)
        if (hipEventCreate(&e) != hipSuccess) { kmdb_db_free(db); return kmdb_set_error("hipEventCreate failed"); }
    db->stats.algorithmic_bytes = alg_bytes;
    db->stats.tree_updates = tree_updates;
    db->stats.sum_pairs = sum_pairs;
    db->stats.device_bytes = dev_bytes;
    db->stats.n_segments = segs.size();
    for (auto& e : db->ev_k2)
        if (hipEventCreate(&e) != hipSuccess) { kmdb_db_free(db); return kmdb_set_error("hipEventCreate failed"); }
    if (hipEventCreate(&db->ev_k0) != hipSuccess) { kmdb_db_free(db); return kmdb_set_error("hipEventCreate failed"); }
    {
        // layout metadata for the batch-parallel front half: root-path length of every node, the nodes in
        // order of decreasing local-list length, and the root path of every segment's first node
        std::vector<uint16_t> depth(P, 0);
        uint32_t max_depth = 0;
        for (uint64_t i = 0; i < P; ++i) {
            const uint32_t d = parent[i] < 0 ? 1u : (uint32_t)std::min<uint32_t>(65535u, depth[parent[i]] + 1u);
            depth[i] = (uint16_t)d;
            max_depth = std::max(max_depth, d);
        }
        // K0: nodes with long lists or streams (a few percent) are decoded by a second launch, most work first.
        // Work of a node ~ number of codes that are not "0" (runs of consecutive ids cost one step) ~ stream bits
        // beyond one per delta.
        std::vector<uint32_t> perm, nl(P);
        {
            for (uint64_t i = 0; i < P; ++i) {
                nl[i] = meta[i].x | (meta[i].y << 16);
                if (kmdb_long_node(meta[i].y, meta[i].w)) perm.push_back((uint32_t)i);
            }
            auto work = [&](uint32_t i) -> uint32_t { return meta[i].w - (meta[i].y ? meta[i].y - 1u : 0u); };
            if (!getenv("KMDB_LONG_ORDER_BY_LENGTH"))
                std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return work(a) > work(b); });
            else
                std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return meta[a].y > meta[b].y; });
        }
        // segments of the emit kernels.  Measured: the equal-cost slices of the scatter model (long multi-clade
        // lists weigh more) also balance the emit kernel better than equal node counts do, so they are reused.
        std::vector<Segment> rsegs(segs.begin(), segs.end());
        if (dev_upload(&db->rsegs, rsegs.data(), rsegs.size())) { kmdb_db_free(db); return 1; }
        db->n_rsegs = (uint32_t)rsegs.size();
        std::vector<uint32_t> seg_anc, seg_anc_n(rsegs.size(), 0);
        const bool chain_ok = max_depth <= (uint32_t)KMDB_CHAIN_MAX && !(opts && (opts->flags & KMDB_FLAG_FORCE_SEQ_EMIT));
        db->b3_chain_cap = std::min<uint32_t>(KMDB_CHAIN_MAX, std::max<uint32_t>(8, (max_depth + 7) / 8 * 8));
        const size_t anc_stride = db->b3_chain_cap;
        if (chain_ok) {
            seg_anc.assign(rsegs.size() * anc_stride, 0);
            for (size_t sidx = 0; sidx < rsegs.size(); ++sidx) {
                if (rsegs[sidx].first >= rsegs[sidx].end) continue;
                int32_t cur = parent[rsegs[sidx].first];
                uint32_t d = cur < 0 ? 0u : depth[cur];
                seg_anc_n[sidx] = d;
                while (cur >= 0) { seg_anc[sidx * anc_stride + (--d)] = (uint32_t)cur; cur = parent[cur]; }
            }
        }
        // the narrow kernel is latency-bound per wave, so it gets many small slices of equal node count
        std::vector<Segment> nsegs;
        std::vector<uint32_t> nseg_anc, nseg_anc_n;
        if (chain_ok) {
            uint64_t NSEG = 2048;
            if (const char* e = getenv("KMDB_NSEG")) NSEG = std::max<uint64_t>(64, strtoull(e, nullptr, 10));
            for (uint64_t f = 0; f < P; f += NSEG) nsegs.push_back(Segment{(uint32_t)f, (uint32_t)std::min<uint64_t>(P, f + NSEG)});
            nseg_anc.assign(nsegs.size() * anc_stride, 0);
            nseg_anc_n.assign(nsegs.size(), 0);
            for (size_t sidx = 0; sidx < nsegs.size(); ++sidx) {
                int32_t cur = parent[nsegs[sidx].first];
                uint32_t d = cur < 0 ? 0u : depth[cur];
                nseg_anc_n[sidx] = d;
                while (cur >= 0) { nseg_anc[sidx * anc_stride + (--d)] = (uint32_t)cur; cur = parent[cur]; }
            }
        }
        const kmdb_host_layout hl{max_n, chain_ok, &perm, &nl, &seg_anc, &seg_anc_n, &parent, &depth, &meta, &bitpos, &nsegs, &nseg_anc, &nseg_anc_n, &w};
        phase("copies + emit metadata");
        if (kmdb_records_prepare(db, hl)) { kmdb_db_free(db); return 1; }
        phase("block-record preparation");
    }
    db->stats.device_bytes += kmdb_records_device_bytes(db);
    *out = db;
    return 0;
}

extern "C" void kmdb_db_free(kmdb_db* db) {
    if (!db) return;
    (void)hipSetDevice(db->device);
    void* ptrs[] = {db->meta, db->bitpos, db->parent, db->w, db->sub_end, db->wprefix, db->bits, db->segs, db->rsegs, db->scan_tmp,
                    db->stack_scratch, db->counters, db->bucket_offset, db->slots, db->pid2dfs, nullptr};
    kmdb_records_release(db);
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& e : db->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : db->ev_k2) if (e) (void)hipEventDestroy(e);
    if (db->ev_k0) (void)hipEventDestroy(db->ev_k0);
    if (db->stream) (void)hipStreamDestroy(db->stream);
    delete db;
}

void kmdb_engine_get(kmdb_db* db, kmdb_engine_view* o) {
    o->device = db->device; o->N = db->N; o->P = db->P; o->kmer_length = db->kmer_length;
    o->meta = db->meta; o->bitpos = db->bitpos; o->parent = db->parent; o->w = db->w; o->sub_end = db->sub_end;
    o->bits = db->bits; o->n_buckets = db->n_buckets; o->bucket_offset = db->bucket_offset; o->slots = db->slots;
    o->pid2dfs = db->pid2dfs; o->stream = db->stream;
    for (int i = 0; i < 4; ++i) o->ev[i] = db->ev[i];
}

void kmdb_engine_set_times(kmdb_db* db, double kernel_ms, double dominant_ms) {
    db->stats.kernel_ms = kernel_ms;
    db->stats.dominant_kernel_ms = dominant_ms;
}

extern "C" int kmdb_db_stats(const kmdb_db* db, kmdb_stats* out) {
    if (!db || !out) return kmdb_set_error("kmdb_db_stats: null argument");
    *out = db->stats;
    return 0;
}

// ------------------------------------------------------------------------------------------
// dense all2all
// ------------------------------------------------------------------------------------------
namespace {

// enqueue the whole dense pipeline on `st`; M is device memory of N(N-1)/2 uint32
int run_dense(kmdb_db* db, uint32_t* M, const kmdb_opts* opts, hipStream_t st) {
    const uint64_t N = db->N, P = db->P;
    const uint64_t cells = N ? N * (N - 1) / 2 : 0;
    const uint32_t shard_index = opts ? opts->shard_index : 0, shard_count = opts && opts->shard_count ? opts->shard_count : 1;
    if (shard_index >= shard_count) return kmdb_set_error("kmdb_all2all: shard_index >= shard_count");
    const uint32_t seg_begin = (uint32_t)((uint64_t)db->n_segs * shard_index / shard_count);
    const uint32_t seg_end = (uint32_t)((uint64_t)db->n_segs * (shard_index + 1) / shard_count);
    const uint32_t flags = opts ? opts->flags : 0;

    HIP_TRY(hipEventRecord(db->ev[0], st));
    if (cells) HIP_TRY(hipMemsetAsync(M, 0, cells * 4, st));
    HIP_TRY(hipMemsetAsync(db->counters, 0, 8 * sizeof(unsigned long long), st));
    db->k1_ms = db->k2_ms = -1;
    const bool v1 = !db->b2_ready || shard_count != 1 || !cells ||
                    (flags & (KMDB_FLAG_FORCE_GLOBAL_ATOMICS | KMDB_FLAG_FORCE_DIRECT | KMDB_FLAG_FORCE_TILE));
    if (!v1) {
        // block-record pipeline (flat form: on-disk weights, no subtree sums needed)
        HIP_TRY(hipEventRecord(db->ev[1], st));
        if (kmdb_records_run(db, M, flags, st)) return 1;
    } else {
        // subtree weights (reference similarity_calculator.cpp:64-72): exclusive scan of w in DFS order
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(db->scan_tmp, db->scan_tmp_bytes, db->w, db->wprefix, (int)(P + 1), st));
        HIP_TRY(hipEventRecord(db->ev[1], st));
        if (kmdb_v1_run(db, M, seg_begin, seg_end, flags, st)) return 1;
    }
    HIP_TRY(hipEventRecord(db->ev[2], st));
    return 0;
}

int finish_stats(kmdb_db* db, hipStream_t st) {
    HIP_TRY(hipEventRecord(db->ev[3], st));
    HIP_TRY(hipEventSynchronize(db->ev[3]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, db->ev[0], db->ev[3]));
    HIP_TRY(hipEventElapsedTime(&b, db->ev[1], db->ev[2]));
    db->stats.kernel_ms = a;
    db->stats.dominant_kernel_ms = b;
    if (db->k1_ms >= 0) {
        float k0 = 0, k1 = 0, k2 = 0;
        if (db->k0_ms >= 0) {
            HIP_TRY(hipEventElapsedTime(&k0, db->ev[1], db->ev_k0));
            HIP_TRY(hipEventElapsedTime(&k1, db->ev_k0, db->ev_k2[0]));
        } else
        HIP_TRY(hipEventElapsedTime(&k1, db->ev[1], db->ev_k2[0]));
        HIP_TRY(hipEventElapsedTime(&k2, db->ev_k2[0], db->ev_k2[1]));
        db->k1_ms = k1; db->k2_ms = k2;
        db->stats.dominant_kernel_ms = std::max(k0, std::max(k1, k2));
        db->stats.k0_ms = k0; db->stats.k1_ms = k1; db->stats.k2_ms = k2; db->stats.n_records = db->b2_total;
    } else {
        db->stats.k0_ms = db->stats.k1_ms = db->stats.k2_ms = 0; db->stats.n_records = 0;
    }
    unsigned long long c[8];
    HIP_TRY(hipMemcpy(c, db->counters, sizeof c, hipMemcpyDeviceToHost));
    db->stats.tile_flushes = c[0];
    if (c[1] | c[2] | c[3] | c[4])
        fprintf(stderr, "[kmdb prof] emit kernel, memtime ticks summed over waves: all %llu = inherit %llu + doubling %llu + records %llu + chain %llu (+ fetch); "
                        "compact batches %llu, full-width %llu\n", c[1], c[2], c[5], c[6], c[7], c[3], c[4]);
    return 0;
}

}  // namespace

extern "C" int kmdb_all2all_dense_device(kmdb_db* db, void* out_dev, const kmdb_opts* opts) {
    if (!db || !out_dev) return kmdb_set_error("kmdb_all2all_dense_device: null argument");
    HIP_TRY(hipSetDevice(db->device));
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : db->stream;
    if (run_dense(db, (uint32_t*)out_dev, opts, st)) return 1;
    return finish_stats(db, st);
}

extern "C" int kmdb_all2all_dense(kmdb_db* db, uint32_t* out, const kmdb_opts* opts) {
    if (!db || (!out && db->N > 1)) return kmdb_set_error("kmdb_all2all_dense: null argument");
    HIP_TRY(hipSetDevice(db->device));
    const uint64_t cells = db->N ? db->N * (db->N - 1) / 2 : 0;
    uint32_t* M = nullptr;
    HIP_TRY(hipMalloc((void**)&M, std::max<uint64_t>(cells, 1) * 4));
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : db->stream;
    int rc = run_dense(db, M, opts, st);
    if (!rc) rc = finish_stats(db, st);
    if (!rc && cells && hipMemcpy(out, M, cells * 4, hipMemcpyDeviceToHost) != hipSuccess)
        rc = kmdb_set_error("kmdb_all2all_dense: copy back failed");
    (void)hipFree(M);
    return rc;
}

// ------------------------------------------------------------------------------------------
// sparse all2all: same accumulation in HBM (tree form and flat form give the same cells,
// SURVEY §7), then on-device compaction of the non-zeros into CSR.  Bubbles
// (reference src/bubble_helper.h) only exist to spare the CPU hash maps; their contributions are part
// of the same sums, so bubble_size does not change the result.
// ------------------------------------------------------------------------------------------
extern "C" int kmdb_all2all_sparse(kmdb_db* db, kmdb_sparse_rows* out, const kmdb_opts* opts) {
    if (!db || !out) return kmdb_set_error("kmdb_all2all_sparse: null argument");
    std::memset(out, 0, sizeof *out);
    HIP_TRY(hipSetDevice(db->device));
    const uint64_t N = db->N;
    const uint64_t cells = N ? N * (N - 1) / 2 : 0;
    uint32_t* M = nullptr;
    unsigned long long *row_nnz = nullptr, *row_ptr = nullptr;
    uint32_t *col = nullptr, *val = nullptr;
    void* tmp = nullptr;
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : db->stream;
    int rc = 0;
    auto cleanup = [&]() {
        for (void* p : {(void*)M, (void*)row_nnz, (void*)row_ptr, (void*)col, (void*)val, tmp}) if (p) (void)hipFree(p);
    };
#define SP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    SP_TRY(hipMalloc((void**)&M, std::max<uint64_t>(cells, 1) * 4));
    SP_TRY(hipMalloc((void**)&row_nnz, (N + 1) * 8));
    SP_TRY(hipMalloc((void**)&row_ptr, (N + 1) * 8));
    rc = run_dense(db, M, opts, st);
    if (rc) { cleanup(); return rc; }
    SP_TRY(hipMemsetAsync(row_nnz, 0, (N + 1) * 8, st));
    if (N) hipLaunchKernelGGL(row_nnz_kernel, dim3((unsigned)N), dim3(256), 0, st, M, N, row_nnz);
    size_t tmp_bytes = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, row_nnz, row_ptr, (int)(N + 1), st);
    SP_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
    SP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, row_nnz, row_ptr, (int)(N + 1), st));
    std::vector<unsigned long long> h_ptr(N + 1, 0);
    SP_TRY(hipMemcpyAsync(h_ptr.data(), row_ptr, (N + 1) * 8, hipMemcpyDeviceToHost, st));
    SP_TRY(hipStreamSynchronize(st));
    const uint64_t nnz = h_ptr[N];
    SP_TRY(hipMalloc((void**)&col, std::max<uint64_t>(nnz, 1) * 4));
    SP_TRY(hipMalloc((void**)&val, std::max<uint64_t>(nnz, 1) * 4));
    if (N) hipLaunchKernelGGL(row_compact_kernel, dim3((unsigned)N), dim3(256), 0, st, M, N, row_ptr, col, val);
    rc = finish_stats(db, st);
    if (rc) { cleanup(); return rc; }
    out->n_rows = N;
    out->nnz = nnz;
    out->row_ptr = (uint64_t*)std::malloc((N + 1) * 8);
    out->col = (uint32_t*)std::malloc(std::max<uint64_t>(nnz, 1) * 4);
    out->val = (uint32_t*)std::malloc(std::max<uint64_t>(nnz, 1) * 4);
    for (uint64_t i = 0; i <= N; ++i) out->row_ptr[i] = h_ptr[i];
    if (nnz) {
        SP_TRY(hipMemcpy(out->col, col, nnz * 4, hipMemcpyDeviceToHost));
        SP_TRY(hipMemcpy(out->val, val, nnz * 4, hipMemcpyDeviceToHost));
    }
#undef SP_TRY
    cleanup();
    return 0;
}

extern "C" void kmdb_sparse_free(kmdb_sparse_rows* rows) {
    if (!rows) return;
    std::free(rows->row_ptr); std::free(rows->col); std::free(rows->val);
    std::memset(rows, 0, sizeof *rows);
}

