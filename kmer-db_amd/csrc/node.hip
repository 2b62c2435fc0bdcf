// node.hip — one database over the GPUs of a node (north_star: "prefix buckets shard naturally across the 8 GPUs of one
// node with a final RCCL reduce of the per-GPU partial similarity matrices over xGMI").
//
// The reference call sites this makes multi-GPU are SimilarityCalculator::all2all at src/console_all2all.cpp:31-36 and all2all_sp at
// src/console_all2all_sparse.cpp:44 (followed by compact2, :79).  One host thread per device; shard s of S (the k-mers of the prefix
// buckets b with b % S == s: kmdb_db_upload_shard) lives on device s % D.  A call: every device runs kmdb_all2all_dense_device for its
// shards (several shards of one device one after the other, summed on the device), then — with more than one device — ONE RCCL
// reduce-scatter over flat chunks of the lower triangle (xGMI is point to point: every peer pair sums its chunk over its own link,
// SURVEY 8e), and every device brings ITS chunk to the host (dense) or compacts it where it is (sparse:
// kmdb_sparse_from_dense_device).  RCCL is loaded with dlopen when a second device is used, so a one-GPU box needs no librccl
// (KMDB_NODE_FORCE_RCCL=1 runs the same RCCL calls on a one-rank communicator: the part of the path a one-GPU box can exercise).
#include "engine_state.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;            // (optional: a failed collective's peers are released with it)
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
// 0 on success; the library stays loaded for the life of the process
int rccl_load(Rccl& r) {
    static std::mutex mu;
    static Rccl loaded;
    std::lock_guard<std::mutex> g(mu);
    if (!loaded.lib) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) if ((loaded.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!loaded.lib) return kmdb_set_error(std::string("kmdb_node: more than one device needs RCCL and librccl.so could not be loaded: ") + dlerror());
        loaded.GetVersion = (decltype(loaded.GetVersion))dlsym(loaded.lib, "ncclGetVersion");
        loaded.CommInitAll = (decltype(loaded.CommInitAll))dlsym(loaded.lib, "ncclCommInitAll");
        loaded.CommDestroy = (decltype(loaded.CommDestroy))dlsym(loaded.lib, "ncclCommDestroy");
        loaded.CommAbort = (decltype(loaded.CommAbort))dlsym(loaded.lib, "ncclCommAbort");
        loaded.ReduceScatter = (decltype(loaded.ReduceScatter))dlsym(loaded.lib, "ncclReduceScatter");
        loaded.GetErrorString = (decltype(loaded.GetErrorString))dlsym(loaded.lib, "ncclGetErrorString");
        if (!loaded.GetVersion || !loaded.CommInitAll || !loaded.CommDestroy || !loaded.ReduceScatter || !loaded.GetErrorString) {
            loaded.lib = nullptr;
            return kmdb_set_error("kmdb_node: librccl.so lacks one of ncclGetVersion / ncclCommInitAll / ncclCommDestroy / ncclReduceScatter / ncclGetErrorString");
        }
    }
    r = loaded;
    return 0;
}

__global__ void add_u32_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];                             // uint32 wrap-around, like every add of the matrix
}

struct DevSlot {
    int device = 0;
    std::vector<kmdb_db*> shards;                            // the shards s with s % D == this slot's index
    uint32_t* acc = nullptr;                                 // [per * D] the device's partial matrix (the triangle padded to equal chunks)
    uint32_t* tmp = nullptr;                                 // [cells] a further shard's matrix before it is added (only with several shards per device)
    uint32_t* chunk = nullptr;                               // [per] this device's chunk of the reduced matrix (D > 1)
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};      // start of the call, end of the device's own shards, end of the collective, end of the copy
    ncclComm_t comm = nullptr;
    double call_ms = 0, collective_ms = 0, d2h_ms = 0, upload_s = 0;
    uint64_t h2d_bytes = 0, n_patterns = 0, n_records = 0;
    std::string error;                                       // a device thread's failure (kmdb_last_error is per thread)
};

// The device threads of a call meet here before the collective: a thread that failed on its own shards (out of memory, a launch
// error) must not leave the others waiting inside ncclReduceScatter for a rank that never comes.  arrive(ok) returns
// true to everyone only when every thread arrived with ok.
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    size_t n = 1, waiting = 0;
    uint64_t gen = 0;
    bool all_ok = true, result = true;
    bool arrive(bool ok) {
        std::unique_lock<std::mutex> g(mu);
        all_ok = all_ok && ok;
        if (++waiting == n) { result = all_ok; waiting = 0; all_ok = true; ++gen; cv.notify_all(); return result; }
        const uint64_t my = gen;
        cv.wait(g, [&] { return gen != my; });
        return result;
    }
};

}  // namespace

struct kmdb_node {
    uint64_t N = 0, cells = 0, per = 0;
    uint32_t n_shards = 0;
    std::vector<DevSlot> dev;
    Rccl rccl;
    int rccl_version = 0;
    bool use_rccl = false;                                   // more than one device — or KMDB_NODE_FORCE_RCCL=1 on one device: the same calls on a
                                                             // one-rank communicator (what a one-GPU box can exercise of the RCCL path)
    kmdb_node_stats stats{};
    Rendezvous meet;
    std::atomic<bool> aborted{false};                        // a collective failed after the rendezvous: the communicators are gone
};

namespace {

// fn(slot index) on one host thread per device; the first failure's message becomes the caller's kmdb_last_error
template <class F>
int on_devices(kmdb_node* nd, F&& fn) {
    std::vector<std::thread> th;
    for (size_t d = 0; d < nd->dev.size(); ++d)
        th.emplace_back([&, d]() {
            nd->dev[d].error.clear();
            const bool dev_ok = hipSetDevice(nd->dev[d].device) == hipSuccess;
            if (!dev_ok) (void)kmdb_set_error("hipSetDevice failed");
            if (fn(d, dev_ok)) nd->dev[d].error = kmdb_last_error();      // (fn still runs: a call with a collective has a rendezvous every thread must reach)
        });
    for (auto& t : th) t.join();
    for (auto& s : nd->dev)
        if (!s.error.empty()) return kmdb_set_error("device " + std::to_string(s.device) + ": " + s.error);
    return 0;
}

#define NODE_TRY(expr)                                                                                              \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

// the device's own shards into acc; everything stays on the device's stream, nothing waits on the host
int node_own_shards(kmdb_node* nd, size_t d, const kmdb_opts* opts) {
    DevSlot& s = nd->dev[d];
    const size_t D = nd->dev.size();
    kmdb_opts o{};
    if (opts) o = *opts;
    o.abi_version = KMDB_ABI_VERSION; o.device = s.device; o.stream = s.stream;
    if (o.shard_count == 0) { o.shard_index = 0; o.shard_count = 1; }
    NODE_TRY(hipEventRecord(s.ev[0], s.stream));
    if (nd->per * D > nd->cells) NODE_TRY(hipMemsetAsync(s.acc + nd->cells, 0, (nd->per * D - nd->cells) * 4, s.stream));     // the padding of the last chunk
    s.n_records = 0;
    for (size_t k = 0; k < s.shards.size(); ++k) {
        uint32_t* dst = k == 0 ? s.acc : s.tmp;
        if (kmdb_all2all_dense_device(s.shards[k], dst, &o)) return 1;
        kmdb_stats st{};
        if (!kmdb_db_stats(s.shards[k], &st)) s.n_records += st.n_records + st.n_direct;      // (block records, written or applied where they were emitted)
        if (k && nd->cells) {
            hipLaunchKernelGGL(add_u32_kernel, dim3((unsigned)((nd->cells + 255) / 256)), dim3(256), 0, s.stream, s.acc, s.tmp, nd->cells);
            NODE_TRY(hipGetLastError());
        }
    }
    if (s.shards.empty() && nd->cells) NODE_TRY(hipMemsetAsync(s.acc, 0, nd->cells * 4, s.stream));
    NODE_TRY(hipEventRecord(s.ev[1], s.stream));
    return 0;
}
// the partial matrix of device slot d in acc (all its shards), then the node's reduce-scatter behind it ON THE SAME STREAM (no host
// wait in between: the collective starts when the device's last kernel ends); ev[2] marks its end.  The caller queues its copy /
// compaction behind that and waits once.
int node_accumulate(kmdb_node* nd, size_t d, bool dev_ok, const kmdb_opts* opts) {
    DevSlot& s = nd->dev[d];
    s.call_ms = s.collective_ms = s.d2h_ms = 0;
    const int own = dev_ok ? node_own_shards(nd, d, opts) : 1;
    const std::string own_msg = own ? kmdb_last_error() : "";
    if (nd->use_rccl && nd->per) {
        // every device thread arrives here, failed or not; the collective is entered by all or by none
        if (!nd->meet.arrive(own == 0)) {
            // nobody enters the collective.  A thread that succeeded still has its kernels queued: they finish before the call returns, so that
            // the caller may free or reuse what they write
            if (dev_ok) (void)hipStreamSynchronize(s.stream);
            return kmdb_set_error(own ? own_msg : std::string("another device failed before the reduce-scatter: the collective was not entered"));
        }
        const ncclResult_t r = nd->rccl.ReduceScatter(s.acc, s.chunk, nd->per, ncclUint32, ncclSum, s.comm, s.stream);
        if (r != ncclSuccess) {
            // a rank that fails AFTER the rendezvous leaves its peers inside the collective: every communicator of the node is aborted (they are
            // all this process's), which releases them with an error; the handle's collectives are unusable from here on
            const std::string msg = std::string("ncclReduceScatter: ") + nd->rccl.GetErrorString(r);
            if (nd->rccl.CommAbort && !nd->aborted.exchange(true))
                for (auto& o : nd->dev) if (o.comm) { (void)nd->rccl.CommAbort(o.comm); o.comm = nullptr; }
            return kmdb_set_error(msg);
        }
    } else if (own) return kmdb_set_error(own_msg);
    NODE_TRY(hipEventRecord(s.ev[2], s.stream));
    return 0;
}
// after the stream has drained: the device's times from its events
int node_times(kmdb_node* nd, size_t d) {
    DevSlot& s = nd->dev[d];
    float a = 0, b = 0, c = 0;
    NODE_TRY(hipEventElapsedTime(&a, s.ev[0], s.ev[1]));
    NODE_TRY(hipEventElapsedTime(&b, s.ev[1], s.ev[2]));
    NODE_TRY(hipEventElapsedTime(&c, s.ev[2], s.ev[3]));
    s.call_ms = a; s.collective_ms = nd->use_rccl ? b : 0.0; s.d2h_ms = c;
    return 0;
}

// where device slot d's cells of the reduced matrix are: the pointer and the flat range of the triangle
void node_chunk(const kmdb_node* nd, size_t d, const uint32_t*& p, uint64_t& lo, uint64_t& hi) {
    const DevSlot& s = nd->dev[d];
    if (!nd->use_rccl) { p = s.acc; lo = 0; hi = nd->cells; return; }
    p = s.chunk;
    lo = std::min<uint64_t>(nd->cells, nd->per * d);
    hi = std::min<uint64_t>(nd->cells, nd->per * (d + 1));
}

void node_fill_stats(kmdb_node* nd) {
    kmdb_node_stats& st = nd->stats;
    st.n_shards = nd->n_shards; st.n_devices = (uint32_t)nd->dev.size(); st.rccl_version = nd->rccl_version;
    st.call_ms = st.collective_ms = st.d2h_ms = 0;
    for (auto& s : nd->dev) {
        st.call_ms = std::max(st.call_ms, s.call_ms);
        st.collective_ms = std::max(st.collective_ms, s.collective_ms);
        st.d2h_ms = std::max(st.d2h_ms, s.d2h_ms);
    }
}

}  // namespace

extern "C" int kmdb_node_upload(const kmdb_db_view* view, uint32_t n_shards, const int32_t* devices, uint32_t n_devices, kmdb_node** out) {
    if (!out) return kmdb_set_error("kmdb_node_upload: null argument");
    *out = nullptr;
    if (!view || view->abi_version != KMDB_ABI_VERSION) return kmdb_set_error("kmdb_node_upload: bad view / ABI version");
    if (n_shards == 0 || n_devices == 0 || !devices) return kmdb_set_error("kmdb_node_upload: no shards / no devices");
    if (n_shards > KMDB_MAX_SHARDS) return kmdb_set_error("kmdb_node_upload: more than " + std::to_string(KMDB_MAX_SHARDS) + " shards");
    if (n_shards > 1 && view->n_buckets == 0) return kmdb_set_error("kmdb_node_upload: prefix shards need the hashtables (load the database with mode Everything)");
    const uint32_t D = std::min(n_shards, n_devices);           // a device without a shard would only add zeros to the reduce
    for (uint32_t a = 0; a < D; ++a)
        for (uint32_t b = a + 1; b < D; ++b)
            if (devices[a] == devices[b]) return kmdb_set_error("kmdb_node_upload: device " + std::to_string(devices[a]) + " listed twice");
    kmdb_node* nd = nullptr;
    try {                                                       // (nothing may leave through the C boundary: allocation and thread failures end in kmdb_set_error)
    nd = new kmdb_node();
    nd->N = view->n_samples; nd->cells = nd->N ? nd->N * (nd->N - 1) / 2 : 0; nd->n_shards = n_shards;
    nd->per = D > 1 ? (nd->cells + D - 1) / D : nd->cells;
    nd->dev.resize(D);
    for (uint32_t d = 0; d < D; ++d) nd->dev[d].device = devices[d];
    nd->meet.n = D;
    const char* force = getenv("KMDB_NODE_FORCE_RCCL");
    nd->use_rccl = D > 1 || (force && force[0] == '1');
    const auto t0 = std::chrono::steady_clock::now();
    // The shards are planned on the host (host_shards.cpp: one pass over the hashtable items, one sweep over the tree) — all at once while their
    // weight counters (4 bytes per pattern and shard until a shard's upload releases its own) fit a budget, else in rounds of whole multiples of
    // the devices (64 shards of a 10^8-pattern database would have held 25 GB of counters at once); every device thread then
    // narrows, packs and copies only what its own shards keep.
    int rc = on_devices(nd, [&](size_t d, bool dev_ok) -> int {
        if (!dev_ok) return 1;
        DevSlot& s = nd->dev[d];
        NODE_TRY(hipStreamCreate(&s.stream));
        for (auto& e : s.ev) NODE_TRY(hipEventCreate(&e));
        return 0;
    });
    uint64_t budget = 16ull << 30;
    if (const char* e = getenv("KMDB_PLAN_BUDGET_MB")) if (*e) budget = std::max<uint64_t>(1, strtoull(e, nullptr, 10)) << 20;      // (tests: several rounds on a small database)
    const uint64_t per_shard = std::max<uint64_t>(view->n_patterns * 4, 1);
    const uint32_t group = n_shards == 1 ? 1u : (uint32_t)std::min<uint64_t>(n_shards, std::max<uint64_t>(D, budget / per_shard / D * D));
    for (uint32_t g0 = 0; g0 < n_shards && !rc; g0 += group) {
        const uint32_t g1 = std::min(n_shards, g0 + group);
        kmdb_shard_plan plan;
        if (n_shards > 1) {
            const auto p0 = std::chrono::steady_clock::now();
            std::vector<uint32_t> subset;
            for (uint32_t sh = g0; sh < g1; ++sh) subset.push_back(sh);
            if (kmdb_shard_plan_build(view, n_shards, subset, &plan)) { rc = 1; break; }
            nd->stats.plan_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count();
        }
        rc = on_devices(nd, [&](size_t d, bool dev_ok) -> int {
            if (!dev_ok) return 1;
            DevSlot& s = nd->dev[d];
            const auto u0 = std::chrono::steady_clock::now();
            kmdb_opts o{}; o.abi_version = KMDB_ABI_VERSION; o.device = s.device; o.shard_count = 1;
            for (uint32_t sh = g0; sh < g1; ++sh) {
                if (sh % D != d) continue;
                kmdb_db* db = nullptr;
                if (n_shards == 1 ? kmdb_db_upload(view, &o, 0, &db) : kmdb_db_upload_planned(view, &o, 0, sh, n_shards, &plan, &db)) return 1;
                s.shards.push_back(db);
                kmdb_stats st{};
                if (!kmdb_db_stats(db, &st)) { s.h2d_bytes += st.h2d_bytes; s.n_patterns += st.n_patterns; }
            }
            s.upload_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - u0).count();
            return 0;
        });
    }
    if (!rc) rc = on_devices(nd, [&](size_t d, bool dev_ok) -> int {
        if (!dev_ok) return 1;
        DevSlot& s = nd->dev[d];
        const auto u0 = std::chrono::steady_clock::now();
        NODE_TRY(hipMalloc((void**)&s.acc, std::max<uint64_t>(nd->per * D, 1) * 4));
        if (s.shards.size() > 1) NODE_TRY(hipMalloc((void**)&s.tmp, std::max<uint64_t>(nd->cells, 1) * 4));
        if (nd->use_rccl) NODE_TRY(hipMalloc((void**)&s.chunk, std::max<uint64_t>(nd->per, 1) * 4));
        s.upload_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - u0).count();
        return 0;
    });
    if (!rc && nd->use_rccl) {
        // one communicator per device, made by this thread for all of them (ncclCommInitAll); every device thread then uses its own
        rc = rccl_load(nd->rccl);
        if (!rc) {
            std::vector<ncclComm_t> comms(D);
            std::vector<int> devs(D);
            for (uint32_t d = 0; d < D; ++d) devs[d] = nd->dev[d].device;
            const ncclResult_t r = nd->rccl.CommInitAll(comms.data(), (int)D, devs.data());
            if (r != ncclSuccess) rc = kmdb_set_error(std::string("ncclCommInitAll: ") + nd->rccl.GetErrorString(r));
            else for (uint32_t d = 0; d < D; ++d) nd->dev[d].comm = comms[d];
            (void)nd->rccl.GetVersion(&nd->rccl_version);
        }
    }
    if (rc) { const std::string msg = kmdb_last_error(); kmdb_node_free(nd); return kmdb_set_error(msg); }
    nd->stats.upload_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    node_fill_stats(nd);
    *out = nd;
    return 0;
    } catch (const std::exception& e) {
        if (nd) kmdb_node_free(nd);
        return kmdb_set_error(std::string("kmdb_node_upload: ") + e.what());
    }
}

extern "C" void kmdb_node_free(kmdb_node* nd) {
    if (!nd) return;
    for (auto& s : nd->dev) {
        (void)hipSetDevice(s.device);
        if (s.comm && nd->rccl.CommDestroy) (void)nd->rccl.CommDestroy(s.comm);
        for (kmdb_db* db : s.shards) kmdb_db_free(db);
        for (void* p : {(void*)s.acc, (void*)s.tmp, (void*)s.chunk}) if (p) (void)hipFree(p);
        for (auto& e : s.ev) if (e) (void)hipEventDestroy(e);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete nd;
}

extern "C" int kmdb_node_stats_get(const kmdb_node* nd, kmdb_node_stats* out) {
    if (!nd || !out) return kmdb_set_error("kmdb_node_stats_get: null argument");
    *out = nd->stats;
    return 0;
}

extern "C" int kmdb_node_device_stats_get(const kmdb_node* nd, uint32_t slot, kmdb_node_device_stats* out) {
    if (!nd || !out) return kmdb_set_error("kmdb_node_device_stats_get: null argument");
    if (slot >= nd->dev.size()) return kmdb_set_error("kmdb_node_device_stats_get: the node has " + std::to_string(nd->dev.size()) + " device slots");
    const DevSlot& s = nd->dev[slot];
    out->device = s.device; out->n_shards = (uint32_t)s.shards.size();
    out->upload_s = s.upload_s; out->call_ms = s.call_ms; out->collective_ms = s.collective_ms; out->d2h_ms = s.d2h_ms;
    out->h2d_bytes = s.h2d_bytes; out->n_patterns = s.n_patterns; out->n_records = s.n_records;
    return 0;
}

extern "C" int kmdb_node_all2all_dense(kmdb_node* nd, uint32_t* out_lower_tri, const kmdb_opts* opts) {
    if (!nd || (!out_lower_tri && nd->cells)) return kmdb_set_error("kmdb_node_all2all_dense: null argument");
    if (nd->aborted) return kmdb_set_error("kmdb_node_all2all_dense: an earlier collective failed and the node's communicators were aborted (upload again)");
    const int rc = on_devices(nd, [&](size_t d, bool dev_ok) -> int {
        if (node_accumulate(nd, d, dev_ok, opts)) return 1;
        DevSlot& s = nd->dev[d];
        const uint32_t* p; uint64_t lo, hi;
        node_chunk(nd, d, p, lo, hi);
        // every device its own chunk, side by side over PCIe, queued behind its collective: the chunk leaves as soon as it has landed
        if (hi > lo) NODE_TRY(hipMemcpyAsync(out_lower_tri + lo, p, (hi - lo) * 4, hipMemcpyDeviceToHost, s.stream));
        NODE_TRY(hipEventRecord(s.ev[3], s.stream));
        NODE_TRY(hipStreamSynchronize(s.stream));
        return node_times(nd, d);
    });
    node_fill_stats(nd);
    return rc;
}

extern "C" int kmdb_node_all2all_sparse(kmdb_node* nd, const kmdb_cell_filter* filters, size_t n_filters, const uint32_t* sample_kmers, int measure,
                                        kmdb_sparse_rows* out, const kmdb_opts* opts) {
    if (!nd || !out) return kmdb_set_error("kmdb_node_all2all_sparse: null argument");
    if (nd->aborted) return kmdb_set_error("kmdb_node_all2all_sparse: an earlier collective failed and the node's communicators were aborted (upload again)");
    std::memset(out, 0, sizeof *out);
    const size_t D = nd->dev.size();
    std::vector<kmdb_sparse_rows> part(D);
    for (auto& p : part) std::memset(&p, 0, sizeof p);
    int rc = on_devices(nd, [&](size_t d, bool dev_ok) -> int {
        if (node_accumulate(nd, d, dev_ok, opts)) return 1;
        DevSlot& s = nd->dev[d];
        if (s.shards.empty()) return kmdb_set_error("kmdb_node_all2all_sparse: a device without a shard");
        const uint32_t* p; uint64_t lo, hi;
        node_chunk(nd, d, p, lo, hi);
        kmdb_opts o{}; o.abi_version = KMDB_ABI_VERSION; o.device = s.device; o.shard_count = 1; o.stream = s.stream;
        // the cells are complete sums here (all shards, all devices): bounds and measures apply (kmdb_sparse_from_dense_device, on the
        // same stream behind the collective)
        if (kmdb_sparse_from_dense_device(s.shards[0], p, lo, hi, filters, n_filters, sample_kmers, measure, &part[d], &o)) return 1;
        NODE_TRY(hipEventRecord(s.ev[3], s.stream));
        NODE_TRY(hipStreamSynchronize(s.stream));
        return node_times(nd, d);
    });
    node_fill_stats(nd);
    if (!rc) {
        // rows of the devices' chunks side by side: device order = ascending columns inside a row cut by a chunk boundary
        const uint64_t N = nd->N;
        uint64_t nnz = 0;
        for (auto& p : part) nnz += p.nnz;
        out->n_rows = N; out->nnz = nnz;
        out->row_ptr = (uint64_t*)std::malloc((N + 1) * 8);
        out->col = (uint32_t*)std::malloc(std::max<uint64_t>(nnz, 1) * 4);
        out->val = (uint32_t*)std::malloc(std::max<uint64_t>(nnz, 1) * 4);
        if (measure >= 0) out->measure = (double*)std::malloc(std::max<uint64_t>(nnz, 1) * 8);
        if (!out->row_ptr || !out->col || !out->val || (measure >= 0 && !out->measure)) rc = kmdb_set_error("kmdb_node_all2all_sparse: out of host memory for the result");
        else {
            uint64_t w = 0;
            for (uint64_t i = 0; i < N; ++i) {
                out->row_ptr[i] = w;
                for (auto& p : part) {
                    if (!p.row_ptr) continue;
                    const uint64_t a = p.row_ptr[i], b = p.row_ptr[i + 1];
                    if (b == a) continue;
                    std::memcpy(out->col + w, p.col + a, (b - a) * 4);
                    std::memcpy(out->val + w, p.val + a, (b - a) * 4);
                    if (measure >= 0 && p.measure) std::memcpy(out->measure + w, p.measure + a, (b - a) * 8);
                    w += b - a;
                }
            }
            out->row_ptr[N] = w;
        }
    }
    for (auto& p : part) kmdb_sparse_free(&p);
    if (rc) { const std::string msg = kmdb_last_error(); kmdb_sparse_free(out); return kmdb_set_error(msg); }
    return 0;
}
