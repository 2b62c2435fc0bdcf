// engine.hip — C-ABI entry points of the MI355X (gfx950) engine behind include/kmdb_amd.h.
//
// Replaces the reference's SimilarityCalculator::all2all / all2all_sp call sites
// (reference src/similarity_calculator.cpp:42-438, 442-657; console_all2all.cpp:34, console_all2all_sparse.cpp:44).
//   layout.hip      kmdb_db_upload: on-disk pattern section -> DFS pre-order layout in HBM (format conversion only)
//   a2a_blocks.hip  the block-record pipeline: every decode / count / placement step inside the call
//   a2a_v1.hip      tree-form scatter kernels: A/B reference behind flags, and the (announced) fallback
// uint32 adds wrap and commute, so any schedule is bit-exact with the reference.
#include "device_common.h"
#include "engine_internal.h"

#include "prim.h"

#include <sys/mman.h>
#include <algorithm>
#include <cmath>
#include <limits>
#include <thread>
#include <vector>
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

extern "C" int kmdb_device_prepare(int32_t device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return kmdb_set_error("kmdb_device_prepare: no such HIP device"); }
    HIP_TRY(hipSetDevice(device));
    void* p = nullptr;
    hipStream_t s = nullptr;
    HIP_TRY(hipMalloc(&p, 1 << 20));
    HIP_TRY(hipStreamCreate(&s));
    HIP_TRY(hipMemsetAsync(p, 0, 1 << 20, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipStreamDestroy(s));
    HIP_TRY(hipFree(p));
    return 0;
}

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
// Device side of the -min / -max filters: every bound is brought to one of six plain ratios of the cell
// (log-based measures are monotone in theirs) and widened by a safety margin on the host; a cell that misses a widened bound is
// dropped here, the rest is decided on the host with the reference's own arithmetic.
enum { RATIO_JACCARD = 0, RATIO_MIN, RATIO_MAX, RATIO_COSINE, RATIO_QUERY, RATIO_NUM };
constexpr int DEV_FILTER_MAX = 12;      // one bound per criterion of Params::availableMetrics (9) and a few repeats
struct DevFilter {
    int n;                          // bounds in use (0: keep every non-zero cell)
    int kind[DEV_FILTER_MAX];
    double lo[DEV_FILTER_MAX], hi[DEV_FILTER_MAX];
    const uint32_t* counts;         // [N] k-mer counts of the samples
};
__device__ __forceinline__ bool dev_keep(const DevFilter& f, uint32_t c, uint32_t row, uint32_t col) {
    if (c == 0) return false;
    if (f.n == 0) return true;
    const uint32_t a = f.counts[row], b = f.counts[col];
    for (int i = 0; i < f.n; ++i) {
        double x;
        switch (f.kind[i]) {
        case RATIO_JACCARD: x = (double)c / (double)(uint32_t)(a + b - c); break;
        case RATIO_MIN:     x = (double)c / (double)(a < b ? a : b); break;
        case RATIO_MAX:     x = (double)c / (double)(a > b ? a : b); break;
        case RATIO_COSINE:  x = (double)c / sqrt((double)(uint32_t)(a * b)); break;
        case RATIO_QUERY:   x = (double)c / (double)a; break;
        default:            x = (double)c; break;
        }
        if (!(x >= f.lo[i] && x <= f.hi[i])) return false;      // NaN fails, as on the host
    }
    return true;
}
// The columns of row `row` inside the flat cell range [cell_lo, cell_hi) of the lower triangle (row i at i (i - 1) / 2,
// reference src/array.h:136-140): [j0, j1).  M points at cell `cell_lo`.
__device__ __forceinline__ void row_span(uint64_t row, uint64_t cell_lo, uint64_t cell_hi, uint64_t& j0, uint64_t& j1) {
    const uint64_t b = tri64(row), e = b + row;
    j0 = cell_lo > b ? cell_lo - b : 0;
    j1 = cell_hi < e ? (cell_hi > b ? cell_hi - b : 0) : row;
    if (j0 > j1) j0 = j1;
}
__global__ void row_nnz_kernel(const uint32_t* __restrict__ M, uint64_t row_lo, uint64_t cell_lo, uint64_t cell_hi, unsigned long long* __restrict__ row_nnz,
                               const DevFilter f) {
    const uint64_t row = row_lo + blockIdx.x;
    const uint32_t* r = M + ((int64_t)tri64(row) - (int64_t)cell_lo);      // only cells of the range are touched
    uint64_t j0, j1;
    row_span(row, cell_lo, cell_hi, j0, j1);
    uint32_t c = 0;
    for (uint64_t j = j0 + threadIdx.x; j < j1; j += blockDim.x) c += dev_keep(f, r[j], (uint32_t)row, (uint32_t)j) ? 1u : 0u;
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
__global__ void row_compact_kernel(const uint32_t* __restrict__ M, uint64_t row_lo, uint64_t cell_lo, uint64_t cell_hi,
                                   const unsigned long long* __restrict__ row_ptr, uint32_t* __restrict__ col, uint32_t* __restrict__ val, const DevFilter f) {
    const uint64_t row = row_lo + blockIdx.x;
    const uint32_t* r = M + ((int64_t)tri64(row) - (int64_t)cell_lo);
    uint64_t jb, je;
    row_span(row, cell_lo, cell_hi, jb, je);
    __shared__ uint32_t wave_cnt[4];
    __shared__ unsigned long long running;
    if (threadIdx.x == 0) running = row_ptr[row];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t j0 = jb; j0 < je; j0 += blockDim.x) {
        const uint64_t j = j0 + threadIdx.x;
        uint32_t v = j < je ? r[j] : 0u;
        if (!dev_keep(f, v, (uint32_t)row, (uint32_t)j)) v = 0u;
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


// The same two passes over the TOUCHED tiles only (reference: all2all_sp is O(nnz), src/similarity_calculator.cpp:596-638, compact2
// src/array.h:391-446): the apply kernels of the block-record pipeline flag every block pair (X, Y) they add to (kmdb_db.tile_touched);
// a row of block row X looks at the flags of its X + 1 column blocks, 64 at a time, and reads only the cells of flagged tiles.  One
// wave per row (a tile holds at most 64 columns: a lane each); the row's entries come out in ascending column order.
struct TileSkip { const unsigned char* touched; uint32_t width; };
template <bool COMPACT>
__global__ __launch_bounds__(64) void row_tiles_kernel(const uint32_t* __restrict__ M, uint64_t n_rows, const TileSkip ts, unsigned long long* __restrict__ row_nnz,
                                                       const unsigned long long* __restrict__ row_ptr, uint32_t* __restrict__ col, uint32_t* __restrict__ val,
                                                       const DevFilter f) {
    const uint64_t row = blockIdx.x;
    if (row >= n_rows) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t X = (uint32_t)(row / ts.width);
    const uint32_t* r = M + tri64(row);
    const unsigned char* flags = ts.touched + (size_t)X * (X + 1) / 2;
    unsigned long long out = COMPACT ? row_ptr[row] : 0ull;
    uint32_t count = 0;
    for (uint32_t Y0 = 0; Y0 <= X; Y0 += 64) {
        const uint32_t Yl = Y0 + lane;
        unsigned long long m = __ballot(Yl <= X && flags[Yl] != 0);
        while (m) {
            const uint32_t Y = Y0 + (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            const uint64_t j = (uint64_t)Y * ts.width + lane;
            uint32_t v = (lane < ts.width && j < row) ? r[j] : 0u;
            if (!dev_keep(f, v, (uint32_t)row, (uint32_t)j)) v = 0u;
            const unsigned long long bal = __ballot(v != 0);
            if (COMPACT) {
                if (v) { const unsigned long long o = out + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)); col[o] = (uint32_t)j; val[o] = v; }
                out += (uint32_t)__popcll(bal);
            } else count += (uint32_t)__popcll(bal);
        }
    }
    if (!COMPACT && lane == 0) row_nnz[row] = count;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// upload (layout.hip does the conversion)
// ------------------------------------------------------------------------------------------
namespace {

int upload_impl(const kmdb_db_view* v, const kmdb_opts* opts, int with_hashtables, uint32_t shard_index, uint32_t shard_count, kmdb_shard_plan* plan, kmdb_db** out) {
    *out = nullptr;
    if (!v || v->abi_version != KMDB_ABI_VERSION) return kmdb_set_error("kmdb_db_upload: bad view / ABI version");
    if (opts && (opts->flags & ~KMDB_FLAG_ALL)) return kmdb_set_error("kmdb_db_upload: unknown bits in kmdb_opts.flags");
    const uint64_t P = v->n_patterns, N = v->n_samples;
    if (P >= (1ull << 31)) return kmdb_set_error("kmdb_db_upload: more than 2^31 patterns");
    if (N >= KMDB_MAX_SAMPLES) return kmdb_set_error("kmdb_db_upload: " + std::to_string(KMDB_MAX_SAMPLES) + " samples or more are not supported (sample ids take " + std::to_string(KMDB_ID_BITS) + " bits in the device layout)");
    if (shard_count == 0 || shard_index >= shard_count) return kmdb_set_error("kmdb_db_upload_shard: shard_index >= shard_count");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return kmdb_set_error("kmdb_db_upload: no HIP device available (the engine has no CPU fallback)");
    }
    const int device = opts ? opts->device : 0;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(device));
    // a prefix shard uploaded by itself plans itself: its k-mer counts from its own buckets, its nodes by one sweep (host_shards.cpp)
    kmdb_shard_plan own_plan;
    if (shard_count > 1 && !plan) {
        if (kmdb_shard_plan_build(v, shard_count, std::vector<uint32_t>{shard_index}, &own_plan)) return 1;
        plan = &own_plan;
    }
    auto* db = new kmdb_db();
    db->device = device; db->N = N; db->P = P; db->kmer_length = v->kmer_length;
    db->one_shot = opts && (opts->flags & KMDB_FLAG_ONE_SHOT);
    auto fail = [&]() { kmdb_db_free(db); return 1; };
    if (hipStreamCreate(&db->stream) != hipSuccess) { kmdb_set_error("hipStreamCreate failed"); return fail(); }
    for (auto& e : db->ev) if (hipEventCreate(&e) != hipSuccess) { kmdb_set_error("hipEventCreate failed"); return fail(); }
    for (auto& e : db->ev_k) if (hipEventCreate(&e) != hipSuccess) { kmdb_set_error("hipEventCreate failed"); return fail(); }
    if (hipStreamCreate(&db->stream2) != hipSuccess) { kmdb_set_error("hipStreamCreate failed"); return fail(); }
    for (auto& e : db->ev_side) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { kmdb_set_error("hipEventCreate failed"); return fail(); }
    // KMDB_VERBOSE: the four parts of the upload as wholes (the layout and the preparation print their own phases; what those
    // leave out — the release of their temporaries, the first use of the device by the process — shows up here)
    const bool verbose = getenv("KMDB_VERBOSE") != nullptr;
    auto t_mark = t0;
    auto part = [&](const char* what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kmdb] upload part: %-40s %.3f s\n", what, std::chrono::duration<double>(now - t_mark).count());
        t_mark = now;
    };
    part("streams + events (first use of the device)");
    if (kmdb_layout_upload(db, v, with_hashtables, shard_index, shard_count, plan)) return fail();
    part("layout incl. release of its temporaries");
    // the working set of all2all: now for an all2all upload, on the first all2all call for a new2all / db2db upload
    if (!with_hashtables && kmdb_blocks_prepare(db)) return fail();
    part("preparation of the all2all working set");
    if (hipStreamSynchronize(db->stream) != hipSuccess) { kmdb_set_error("kmdb_db_upload: device error"); return fail(); }
    part("final wait");
    db->blocks_bytes_counted = kmdb_blocks_device_bytes(db);
    db->stats.device_bytes += db->blocks_bytes_counted;
    db->stats.width = db->width;
    db->stats.upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] upload: total %.3f s (%llu patterns, %llu samples)\n", db->stats.upload_ms * 1e-3,
                                        (unsigned long long)P, (unsigned long long)N);
    *out = db;
    return 0;
}

}  // namespace

extern "C" int kmdb_db_upload(const kmdb_db_view* v, const kmdb_opts* opts, int with_hashtables, kmdb_db** out) {
    return upload_impl(v, opts, with_hashtables, 0, 1, nullptr, out);
}
extern "C" int kmdb_db_upload_shard(const kmdb_db_view* v, const kmdb_opts* opts, int with_hashtables, uint32_t shard_index, uint32_t shard_count,
                                    kmdb_db** out) {
    return upload_impl(v, opts, with_hashtables, shard_index, shard_count, nullptr, out);
}
int kmdb_db_upload_planned(const kmdb_db_view* v, const kmdb_opts* opts, int with_hashtables, uint32_t shard_index, uint32_t shard_count, kmdb_shard_plan* plan,
                           kmdb_db** out) {
    if (!out) return kmdb_set_error("kmdb_db_upload_shard: null argument");
    return upload_impl(v, opts, with_hashtables, shard_index, shard_count, plan, out);
}

extern "C" void kmdb_db_settle(kmdb_db* db) {
    if (db && db->staging_thread.joinable()) db->staging_thread.join();
}

extern "C" void kmdb_db_free(kmdb_db* db) {
    if (!db) return;
    (void)hipSetDevice(db->device);
    db->one_shot = false;
    kmdb_release_staging(db);
    if (db->staging_thread.joinable()) db->staging_thread.join();
    for (const auto& r : db->staging_kept) (void)munmap(r.first, r.second);      // a one-shot handle's regions: pages long dropped, the mappings go now
    db->staging_kept.clear();
    kmdb_blocks_release(db);
    void* ptrs[] = {db->k0in, db->bitrel, db->blkbase, db->bits, db->nl, db->parent, db->w, db->dflag, db->sub_end, db->long_nodes, db->nseg_anc,
                    db->nseg_anc_n, db->meta, db->bitpos, db->ck_ofs, db->ck_bit, db->ck_id, db->wprefix, db->segs, db->v1_scan_tmp, db->stack_scratch, db->v1_counters,
                    db->bucket_offset, db->slots, db->pid2dfs, db->list_sets, db->rl_ofs, db->rl_runs, db->rl_node};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& e : db->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : db->ev_k) if (e) (void)hipEventDestroy(e);
    for (auto& e : db->ev_side) if (e) (void)hipEventDestroy(e);
    if (db->stream2) (void)hipStreamDestroy(db->stream2);
    if (db->stream) (void)hipStreamDestroy(db->stream);
    delete db;
}

int kmdb_engine_get(kmdb_db* db, kmdb_engine_view* o) {
    if (hipSetDevice(db->device) != hipSuccess) return kmdb_set_error("hipSetDevice failed");
    if (kmdb_ensure_v1_arrays(db)) return 1;
    o->device = db->device; o->N = db->N; o->P = db->P; o->kmer_length = db->kmer_length;
    o->meta = db->meta; o->bitpos = db->bitpos; o->parent = db->parent; o->w = db->w; o->sub_end = db->sub_end;
    o->ck_ofs = db->ck_ofs; o->ck_bit = db->ck_bit; o->ck_id = db->ck_id;
    o->bits = db->bits; o->n_buckets = db->n_buckets; o->bucket_offset = db->bucket_offset; o->slots = db->slots;
    o->pid2dfs = db->pid2dfs; o->stream = db->stream;
    o->max_depth = db->max_depth; o->list_sets = &db->list_sets; o->list_sets_nb = &db->list_sets_nb; o->list_sets_tried = &db->list_sets_tried;
    o->rl_ofs = &db->rl_ofs; o->rl_runs = &db->rl_runs; o->rl_node = &db->rl_node; o->rl_tried = &db->rl_tried;
    o->device_bytes = &db->stats.device_bytes;
    for (int i = 0; i < 4; ++i) o->ev[i] = db->ev[i];
    return 0;
}

void kmdb_engine_set_times(kmdb_db* db, double kernel_ms, double dominant_ms) {
    kmdb_release_staging(db);                                   // (end of a new2all / db2db call)
    db->stats.kernel_ms = kernel_ms;
    (void)dominant_ms;
}

extern "C" const char* kmdb_db_fallback_reason(const kmdb_db* db) { return db ? db->fallback_reason.c_str() : ""; }

extern "C" int kmdb_db_stats(const kmdb_db* db, kmdb_stats* out) {
    if (!db || !out) return kmdb_set_error("kmdb_db_stats: null argument");
    *out = db->stats;
    return 0;
}

// ------------------------------------------------------------------------------------------
// dense all2all
// ------------------------------------------------------------------------------------------
namespace {

// the whole dense pipeline on `st`; M is device memory of N(N-1)/2 uint32
int run_dense(kmdb_db* db, uint32_t* M, const kmdb_opts* opts, hipStream_t st) {
    const uint64_t N = db->N, P = db->P;
    const uint64_t cells = N ? N * (N - 1) / 2 : 0;
    const uint32_t shard_index = opts ? opts->shard_index : 0, shard_count = opts && opts->shard_count ? opts->shard_count : 1;
    if (shard_index >= shard_count) return kmdb_set_error("kmdb_all2all: shard_index >= shard_count");
    const uint32_t flags = opts ? opts->flags : 0;
    if (flags & ~KMDB_FLAG_ALL) return kmdb_set_error("kmdb_all2all: unknown bits in kmdb_opts.flags");
    // slice of the DFS-ordered pattern stream whose pairs this call adds
    const uint32_t lo = (uint32_t)(P * shard_index / shard_count), hi = (uint32_t)(P * (shard_index + 1) / shard_count);

    HIP_TRY(hipEventRecord(db->ev[0], st));
    if (cells) HIP_TRY(hipMemsetAsync(M, 0, cells * 4, st));
    db->stats.path = KMDB_PATH_NONE;
    db->stats.k0_ms = db->stats.k1_ms = db->stats.k1n_ms = db->stats.k1g_ms = db->stats.k2_ms = 0;
    db->stats.n_records = 0; db->stats.n_direct = 0; db->stats.n_wide = 0; db->stats.n_chunks = 0; db->stats.tile_flushes = 0; db->stats.sized_call = 0;
    if (!cells || !P) { HIP_TRY(hipEventRecord(db->ev[1], st)); HIP_TRY(hipEventRecord(db->ev[2], st)); return 0; }
    const bool forced_v1 = (flags & (KMDB_FLAG_FORCE_GLOBAL_ATOMICS | KMDB_FLAG_FORCE_DIRECT | KMDB_FLAG_FORCE_TILE)) != 0;
    if (!forced_v1 && !db->blocks_prepared) {
        if (kmdb_blocks_prepare(db)) return 1;
        db->blocks_bytes_counted = kmdb_blocks_device_bytes(db);
        db->stats.device_bytes += db->blocks_bytes_counted;
        db->stats.width = db->width;
    }
    if (!forced_v1 && db->fallback_reason.empty()) {
        // block-record pipeline (flat form: on-disk weights, no subtree sums needed)
        HIP_TRY(hipEventRecord(db->ev[1], st));
        if (kmdb_blocks_run(db, M, lo, hi, st)) return 1;
        const bool sized = db->last_call_sized;
        if (db->fallback_reason.empty()) {
            HIP_TRY(hipEventRecord(db->ev[2], st));
            db->stats.path = KMDB_PATH_RECORDS;
            db->stats.sized_call = sized ? 1u : 0u;
            return 0;
        }
        if (cells) HIP_TRY(hipMemsetAsync(M, 0, cells * 4, st));          // a partial result of the abandoned attempt
    }
    if (!forced_v1) {
        // Not the fast path: say so, once per handle and reason.
        static thread_local const kmdb_db* noted = nullptr;
        if (noted != db) {
            noted = db;
            fprintf(stderr, "[kmdb] note: the block-record pipeline cannot take this database (%s); running the HBM-atomics kernel, "
                            "expect it to be far slower\n", db->fallback_reason.c_str());
        }
        if (flags & KMDB_FLAG_NO_FALLBACK) return kmdb_set_error("kmdb_all2all: block-record pipeline unavailable: " + db->fallback_reason);
        // a slice of the pattern stream means something else to the two paths (pattern ranges here, 2048-node segments there): partial
        // matrices of slices that took different paths would not add up to the whole
        if (shard_count > 1) return kmdb_set_error("kmdb_all2all: slices of the pattern stream (kmdb_opts.shard_count > 1) need the block-record pipeline: " + db->fallback_reason);
    }
    // v1 kernels: tree form, subtree weights (reference similarity_calculator.cpp:64-72) = exclusive scan of w in DFS order
    // (beyond 4096 samples every v1 call runs a2a_global_kernel, whose stack and decoder hold 32-bit ids: no sample limit of its own — round 5
    // refused 65 536 samples and more here, so a tree deeper than the chain table on such a collection had no path at all; the LDS kernels
    // with their 16-bit ids are only ever chosen up to 4096 samples, kmdb_v1_run)
    if (kmdb_ensure_v1_arrays(db)) return 1;
    HIP_TRY(hipMemsetAsync(db->v1_counters, 0, 8 * sizeof(unsigned long long), st));
    HIP_TRY(prim::exclusive_sum(db->v1_scan_tmp, db->v1_scan_tmp_bytes, db->w, db->wprefix, (int)(P + 1), st));
    HIP_TRY(hipEventRecord(db->ev[1], st));
    const uint32_t seg_begin = (uint32_t)((uint64_t)db->n_segs * shard_index / shard_count) / WAVES_PER_BLOCK * WAVES_PER_BLOCK;
    const uint32_t seg_end = shard_index + 1 == shard_count ? db->n_segs : (uint32_t)((uint64_t)db->n_segs * (shard_index + 1) / shard_count) / WAVES_PER_BLOCK * WAVES_PER_BLOCK;
    if (kmdb_v1_run(db, M, seg_begin, seg_end, forced_v1 ? flags : KMDB_FLAG_FORCE_GLOBAL_ATOMICS, st)) return 1;
    HIP_TRY(hipEventRecord(db->ev[2], st));
    const bool global = !forced_v1 || (flags & KMDB_FLAG_FORCE_GLOBAL_ATOMICS) || N > 4096;
    db->stats.path = global ? KMDB_PATH_GLOBAL : KMDB_PATH_TILE;
    return 0;
}

int finish_stats(kmdb_db* db, hipStream_t st) {
    HIP_TRY(hipEventRecord(db->ev[3], st));
    HIP_TRY(hipEventSynchronize(db->ev[3]));
    float a = 0;
    HIP_TRY(hipEventElapsedTime(&a, db->ev[0], db->ev[3]));
    db->stats.kernel_ms = a;
    if (db->stats.path == KMDB_PATH_RECORDS) {
        float k0 = 0, k1n = 0, k1g = 0, k2 = 0;
        HIP_TRY(hipEventElapsedTime(&k0, db->ev[1], db->ev_k[0]));
        HIP_TRY(hipEventElapsedTime(&k1n, db->ev_k[0], db->ev_k[1]));
        HIP_TRY(hipEventElapsedTime(&k1g, db->ev_k[1], db->ev_k[2]));
        HIP_TRY(hipEventElapsedTime(&k2, db->ev_k[2], db->ev_k[3]));
        db->stats.k0_ms = k0; db->stats.k1n_ms = k1n; db->stats.k1g_ms = k1g; db->stats.k1_ms = k1n + k1g; db->stats.k2_ms = k2;
        db->stats.n_records = db->last_records; db->stats.n_wide = db->last_n_wide; db->stats.n_chunks = db->last_n_chunks;
        db->stats.n_joined = db->l2_on ? db->last_l2_nodes : 0u;
        db->stats.n_direct = db->last_n_direct;
        // the pipeline's arrays grow inside calls (second-level arrays on the first call, pools enlarged by a quarter or doubled): what is
        // resident NOW, not what the preparation allocated
        const uint64_t now = kmdb_blocks_device_bytes(db);
        db->stats.device_bytes = db->stats.device_bytes - db->blocks_bytes_counted + now;
        db->blocks_bytes_counted = now;
    } else if (db->v1_counters && db->stats.path != KMDB_PATH_NONE) {
        unsigned long long c[8];
        HIP_TRY(hipMemcpy(c, db->v1_counters, sizeof c, hipMemcpyDeviceToHost));
        db->stats.tile_flushes = c[0];
    }
    return 0;
}

}  // namespace

extern "C" int kmdb_all2all_dense_device(kmdb_db* db, void* out_dev, const kmdb_opts* opts) {
    if (!db || !out_dev) return kmdb_set_error("kmdb_all2all_dense_device: null argument");
    HIP_TRY(hipSetDevice(db->device));
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : db->stream;
    if (run_dense(db, (uint32_t*)out_dev, opts, st)) return 1;
    const int rc = finish_stats(db, st);
    kmdb_release_staging(db);                                   // the upload's host buffers, once the first call is through
    return rc;
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
    kmdb_release_staging(db);
    return rc;
}

// ------------------------------------------------------------------------------------------
// sparse all2all: same accumulation in HBM (tree form and flat form give the same cells,
// SURVEY §7), then on-device compaction of the non-zeros into CSR.  Bubbles
// (reference src/bubble_helper.h) only exist to spare the CPU hash maps; their contributions are part
// of the same sums, so bubble_size does not change the result.
// ------------------------------------------------------------------------------------------
namespace {
// one host bound -> the device's widened bound on a plain ratio.  mash = -(1/k) log(2 j / (1 + j)) falls with j:
// mash <= m  <=>  j >= E / (2 - E), E = exp(-k m); ani = 1 - mash.
struct RatioBound { int kind; double lo, hi; };
RatioBound ratio_bound(const kmdb_cell_filter& f, int k) {
    const double inf = std::numeric_limits<double>::infinity();
    auto j_of_mash = [k](double m) {                       // the j whose mash distance is m (0 .. inf)
        const double E = std::exp(-(double)k * m);
        if (std::isnan(E)) return std::numeric_limits<double>::quiet_NaN();
        if (E >= 2.0) return std::numeric_limits<double>::infinity();
        return E / (2.0 - E);
    };
    RatioBound r{RATIO_NUM, f.lo, f.hi};
    double mlo = f.lo, mhi = f.hi;                          // bounds on the mash distance, for the log-based measures
    bool logm = false;
    switch (f.metric) {
    case KMDB_METRIC_JACCARD: r.kind = RATIO_JACCARD; break;
    case KMDB_METRIC_MIN: r.kind = RATIO_MIN; break;
    case KMDB_METRIC_MAX: r.kind = RATIO_MAX; break;
    case KMDB_METRIC_COSINE: r.kind = RATIO_COSINE; break;
    case KMDB_METRIC_NUM_KMERS: r.kind = RATIO_NUM; break;
    case KMDB_METRIC_MASH: r.kind = RATIO_JACCARD; logm = true; break;
    case KMDB_METRIC_MASH_QUERY: r.kind = RATIO_QUERY; logm = true; break;
    case KMDB_METRIC_ANI: r.kind = RATIO_JACCARD; logm = true; mlo = 1.0 - f.hi; mhi = 1.0 - f.lo; break;
    case KMDB_METRIC_ANI_SHORTER: r.kind = RATIO_MIN; logm = true; mlo = 1.0 - f.hi; mhi = 1.0 - f.lo; break;
    default: break;
    }
    if (logm) { r.lo = j_of_mash(mhi); r.hi = j_of_mash(mlo); }
    // the margin: 1e-6 relative and a little absolute — far above any rounding difference between the two sides, far below
    // anything that costs transfer volume; a NaN bound keeps everything (the host decides)
    if (std::isnan(r.lo)) r.lo = -inf; else r.lo = r.lo - std::fabs(r.lo) * 1e-6 - 1e-300;
    if (std::isnan(r.hi)) r.hi = inf; else r.hi = r.hi + std::fabs(r.hi) * 1e-6 + 1e-300;
    return r;
}
}  // namespace

static int sparse_impl(kmdb_db* db, bool from_cells, const void* dense_dev, uint64_t cell_lo, uint64_t cell_hi, const kmdb_cell_filter* filters, size_t n_filters,
                       const uint32_t* sample_kmers, int measure, kmdb_sparse_rows* out, const kmdb_opts* opts);

static int check_filters(const char* who, const kmdb_cell_filter* filters, size_t n_filters, const uint32_t* sample_kmers, int measure) {
    if ((n_filters && !filters) || ((n_filters || measure >= 0) && !sample_kmers)) return kmdb_set_error(std::string(who) + ": null argument");
    if (measure >= KMDB_METRIC_COUNT) return kmdb_set_error(std::string(who) + ": unknown measure");
    if (n_filters > (size_t)DEV_FILTER_MAX) return kmdb_set_error(std::string(who) + ": more than " + std::to_string(DEV_FILTER_MAX) + " bounds");
    for (size_t i = 0; i < n_filters; ++i)
        if (filters[i].metric < 0 || filters[i].metric >= KMDB_METRIC_COUNT) return kmdb_set_error(std::string(who) + ": unknown metric in a filter");
    return 0;
}

extern "C" int kmdb_all2all_sparse(kmdb_db* db, kmdb_sparse_rows* out, const kmdb_opts* opts) {
    return sparse_impl(db, false, nullptr, 0, ~0ull, nullptr, 0, nullptr, -1, out, opts);
}

extern "C" int kmdb_all2all_sparse_filtered(kmdb_db* db, const kmdb_cell_filter* filters, size_t n_filters, const uint32_t* sample_kmers, int measure,
                                            kmdb_sparse_rows* out, const kmdb_opts* opts) {
    if (check_filters("kmdb_all2all_sparse_filtered", filters, n_filters, sample_kmers, measure)) return 1;
    // bounds are conditions on the whole cell: a slice of the pattern stream (opts->shard_*) holds partial sums only
    if ((n_filters || measure >= 0) && opts && opts->shard_count > 1) return kmdb_set_error("kmdb_all2all_sparse_filtered: filters need the whole database (shard_count must be 1)");
    return sparse_impl(db, false, nullptr, 0, ~0ull, filters, n_filters, sample_kmers, measure, out, opts);
}

extern "C" int kmdb_sparse_from_dense_device(kmdb_db* db, const void* cells_dev, uint64_t cell_lo, uint64_t cell_hi, const kmdb_cell_filter* filters,
                                             size_t n_filters, const uint32_t* sample_kmers, int measure, kmdb_sparse_rows* out, const kmdb_opts* opts) {
    if (!db || !out) return kmdb_set_error("kmdb_sparse_from_dense_device: null argument");
    if (check_filters("kmdb_sparse_from_dense_device", filters, n_filters, sample_kmers, measure)) return 1;
    const uint64_t cells = db->N ? db->N * (db->N - 1) / 2 : 0;
    if (cell_hi > cells) cell_hi = cells;
    if (cell_lo > cell_hi) return kmdb_set_error("kmdb_sparse_from_dense_device: cell_lo > cell_hi");
    if (!cells_dev && cell_hi > cell_lo) return kmdb_set_error("kmdb_sparse_from_dense_device: null matrix");
    return sparse_impl(db, true, cells_dev, cell_lo, cell_hi, filters, n_filters, sample_kmers, measure, out, opts);
}

// from_cells: compact the caller's cells [cell_lo, cell_hi) (dense_dev points at cell_lo); else accumulate the whole triangle first
static int sparse_impl(kmdb_db* db, bool from_cells, const void* dense_dev, uint64_t cell_lo, uint64_t cell_hi, const kmdb_cell_filter* filters, size_t n_filters,
                       const uint32_t* sample_kmers, int measure, kmdb_sparse_rows* out, const kmdb_opts* opts) {
    if (!db || !out) return kmdb_set_error("kmdb_all2all_sparse: null argument");
    std::memset(out, 0, sizeof *out);
    HIP_TRY(hipSetDevice(db->device));
    const uint64_t N = db->N;
    const uint64_t cells = N ? N * (N - 1) / 2 : 0;
    uint32_t* M = nullptr;
    unsigned long long *row_nnz = nullptr, *row_ptr = nullptr;
    uint32_t *col = nullptr, *val = nullptr, *d_counts = nullptr;
    void* tmp = nullptr;
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : db->stream;
    int rc = 0;
    auto cleanup = [&]() {
        for (void* p : {(void*)M, (void*)row_nnz, (void*)row_ptr, (void*)col, (void*)val, (void*)d_counts, tmp}) if (p) (void)hipFree(p);
    };
    DevFilter df{};
#define SP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); kmdb_sparse_free(out); return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    if (!from_cells) SP_TRY(hipMalloc((void**)&M, std::max<uint64_t>(cells, 1) * 4));
    SP_TRY(hipMalloc((void**)&row_nnz, (N + 1) * 8));
    SP_TRY(hipMalloc((void**)&row_ptr, (N + 1) * 8));
    if (n_filters) {
        SP_TRY(hipMalloc((void**)&d_counts, std::max<uint64_t>(N, 1) * 4));
        SP_TRY(hipMemcpyAsync(d_counts, sample_kmers, N * 4, hipMemcpyHostToDevice, st));
        df.n = (int)n_filters; df.counts = d_counts;
        for (size_t i = 0; i < n_filters; ++i) {
            const RatioBound rb = ratio_bound(filters[i], (int)db->kmer_length);
            df.kind[i] = rb.kind; df.lo[i] = rb.lo; df.hi[i] = rb.hi;
        }
    }
    const uint32_t* cellsp = (const uint32_t*)dense_dev;
    if (!from_cells) {
        rc = run_dense(db, M, opts, st);
        if (rc) { cleanup(); return rc; }
        cellsp = M; cell_lo = 0; cell_hi = cells;
    } else {
        SP_TRY(hipEventRecord(db->ev[0], st)); SP_TRY(hipEventRecord(db->ev[1], st)); SP_TRY(hipEventRecord(db->ev[2], st));
        db->stats.path = KMDB_PATH_NONE;
    }
    // rows that meet the range: row r holds the cells [r (r - 1) / 2, r (r + 1) / 2)
    uint64_t row_lo = 0, row_hi = 0;
    if (cell_hi > cell_lo) {
        auto row_of = [](uint64_t c) { uint64_t r = (uint64_t)((1.0 + std::sqrt(1.0 + 8.0 * (double)c)) / 2.0); while (r * (r - 1) / 2 > c) --r; while ((r + 1) * r / 2 <= c) ++r; return r; };
        row_lo = row_of(cell_lo); row_hi = row_of(cell_hi - 1) + 1;
    }
    SP_TRY(hipMemsetAsync(row_nnz, 0, (N + 1) * 8, st));
    // the block-record pipeline just told which tiles it added to: only those are scanned (KMDB_SP_ALL_TILES=1: every cell, as before — A/B)
    const bool skip = !from_cells && db->stats.path == KMDB_PATH_RECORDS && db->tile_touched && N > 1 && !getenv("KMDB_SP_ALL_TILES");
    const TileSkip ts{db->tile_touched, db->width};
    if (skip) hipLaunchKernelGGL((row_tiles_kernel<false>), dim3((unsigned)N), dim3(64), 0, st, cellsp, N, ts, row_nnz, (const unsigned long long*)nullptr, (uint32_t*)nullptr,
                                 (uint32_t*)nullptr, df);
    else
    if (row_hi > row_lo) hipLaunchKernelGGL(row_nnz_kernel, dim3((unsigned)(row_hi - row_lo)), dim3(256), 0, st, cellsp, row_lo, cell_lo, cell_hi, row_nnz, df);
    size_t tmp_bytes = 0;
    SP_TRY(prim::exclusive_sum(nullptr, tmp_bytes, row_nnz, row_ptr, (int)(N + 1), st));
    SP_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
    SP_TRY(prim::exclusive_sum(tmp, tmp_bytes, row_nnz, row_ptr, (int)(N + 1), st));
    std::vector<unsigned long long> h_ptr(N + 1, 0);
    SP_TRY(hipMemcpyAsync(h_ptr.data(), row_ptr, (N + 1) * 8, hipMemcpyDeviceToHost, st));
    SP_TRY(hipStreamSynchronize(st));
    const uint64_t nnz = h_ptr[N];
    SP_TRY(hipMalloc((void**)&col, std::max<uint64_t>(nnz, 1) * 4));
    SP_TRY(hipMalloc((void**)&val, std::max<uint64_t>(nnz, 1) * 4));
    if (skip) hipLaunchKernelGGL((row_tiles_kernel<true>), dim3((unsigned)N), dim3(64), 0, st, cellsp, N, ts, (unsigned long long*)nullptr, row_ptr, col, val, df);
    else
    if (row_hi > row_lo) hipLaunchKernelGGL(row_compact_kernel, dim3((unsigned)(row_hi - row_lo)), dim3(256), 0, st, cellsp, row_lo, cell_lo, cell_hi, row_ptr, col, val, df);
    rc = finish_stats(db, st);
    if (rc) { cleanup(); return rc; }
    out->n_rows = N;
    out->nnz = nnz;
    out->row_ptr = (uint64_t*)std::malloc((N + 1) * 8);
    out->col = (uint32_t*)std::malloc(std::max<uint64_t>(nnz, 1) * 4);
    out->val = (uint32_t*)std::malloc(std::max<uint64_t>(nnz, 1) * 4);
    if (!out->row_ptr || !out->col || !out->val) { cleanup(); kmdb_sparse_free(out); return kmdb_set_error("kmdb_all2all_sparse: out of host memory for the result"); }
    for (uint64_t i = 0; i <= N; ++i) out->row_ptr[i] = h_ptr[i];
    if (nnz) {
        SP_TRY(hipMemcpy(out->col, col, nnz * 4, hipMemcpyDeviceToHost));
        SP_TRY(hipMemcpy(out->val, val, nnz * 4, hipMemcpyDeviceToHost));
    }
#undef SP_TRY
    cleanup();
    kmdb_release_staging(db);
    if (n_filters || measure >= 0) {
        // the host side: every surviving cell decided by the reference's own arithmetic, rows compacted in place, measures computed
        const int k = (int)db->kmer_length;
        const unsigned nthr = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        std::vector<uint64_t> kept(N + 1, 0);
        auto rows_of = [&](unsigned t, auto&& fn) { for (uint64_t i = (uint64_t)N * t / nthr; i < (uint64_t)N * (t + 1) / nthr; ++i) fn(i); };
        auto parallel = [&](auto&& fn) {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthr; ++t) th.emplace_back([&, t]() { rows_of(t, fn); });
            for (auto& x : th) x.join();
        };
        parallel([&](uint64_t i) {                             // pass 1: filter every row in place (write position <= read position)
            uint64_t w = out->row_ptr[i];
            for (uint64_t e = out->row_ptr[i]; e < out->row_ptr[i + 1]; ++e) {
                const uint32_t c = out->val[e], cj = out->col[e];
                bool ok = true;
                for (size_t q = 0; q < n_filters && ok; ++q) {
                    const double x = kmdbh_metric(filters[q].metric, c, sample_kmers[i], sample_kmers[cj], k);
                    ok = x >= filters[q].lo && x <= filters[q].hi;
                }
                if (ok) { out->col[w] = cj; out->val[w] = c; ++w; }
            }
            kept[i + 1] = w - out->row_ptr[i];
        });
        std::vector<uint64_t> new_ptr(N + 1, 0);
        for (uint64_t i = 0; i < N; ++i) new_ptr[i + 1] = new_ptr[i] + kept[i + 1];
        // pass 2: close the gaps (ascending rows; a row never moves right)
        for (uint64_t i = 0; i < N; ++i)
            if (new_ptr[i] != out->row_ptr[i] && kept[i + 1]) {
                std::memmove(out->col + new_ptr[i], out->col + out->row_ptr[i], kept[i + 1] * 4);
                std::memmove(out->val + new_ptr[i], out->val + out->row_ptr[i], kept[i + 1] * 4);
            }
        for (uint64_t i = 0; i <= N; ++i) out->row_ptr[i] = new_ptr[i];
        out->nnz = new_ptr[N];
        if (measure >= 0) {
            out->measure = (double*)std::malloc(std::max<uint64_t>(out->nnz, 1) * 8);
            if (!out->measure) { kmdb_sparse_free(out); return kmdb_set_error("kmdb_all2all_sparse: out of host memory for the measures"); }
            parallel([&](uint64_t i) {
                for (uint64_t e = out->row_ptr[i]; e < out->row_ptr[i + 1]; ++e)
                    out->measure[e] = kmdbh_metric(measure, out->val[e], sample_kmers[i], sample_kmers[out->col[e]], k);
            });
        }
    }
    return 0;
}

extern "C" void kmdb_sparse_free(kmdb_sparse_rows* rows) {
    if (!rows) return;
    std::free(rows->row_ptr); std::free(rows->col); std::free(rows->val); std::free(rows->measure);
    std::memset(rows, 0, sizeof *rows);
}

