// host_shards.cpp — which nodes every prefix shard of a database keeps and how many of a pattern's k-mers it owns (kmdb_internal.h:
// kmdb_shard_plan), on the host, before anything goes to a device.
//
// north_star / SURVEY 8e: "prefix buckets shard naturally across the 8 GPUs of one node".  Round 4 made every device thread narrow the
// WHOLE tree on the host, copy the whole tree AND all hashtable slots to its device and prune there: eight times the host work and eight
// times the PCIe bytes.  Here: the hashtable items are read once (every shard's thread reads only its own
// buckets), the keep flags of all shards climb the parent links in one sweep (a parent's id is below its children's: the pattern ids of
// a kmer-db tree ascend along every root path, reference src/prefix_kmer_db.cpp — new patterns are appended — and kmdb_db_upload
// refuses a view where they do not), and a device gets the kept nodes of its shards only.
#include "kmdb_amd.h"
#include "kmdb_internal.h"

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>

namespace {

void* map_zero(size_t bytes) {
    void* q = mmap(nullptr, std::max<size_t>(bytes, 4096), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    return q == MAP_FAILED ? nullptr : q;
}

}  // namespace

kmdb_shard_plan::~kmdb_shard_plan() {
    for (uint32_t s = 0; s < w.size(); ++s) release_weights(s);
    for (auto* m : mask) if (m) (void)munmap(m, std::max<size_t>(P, 4096));
}
void kmdb_shard_plan::release_weights(uint32_t shard) {
    if (shard < w.size() && w[shard]) { (void)munmap(w[shard], std::max<size_t>(P * 4, 4096)); w[shard] = nullptr; }
}

extern "C" int kmdbh_shard_plan_counts(const kmdb_db_view* view, uint32_t n_shards, uint64_t* kept_nodes, uint64_t* kmers) {
    if (!view || !kept_nodes || !kmers || n_shards == 0) return kmdb_set_error("kmdbh_shard_plan_counts: null argument / no shards");
    if (n_shards > KMDB_MAX_SHARDS) return kmdb_set_error("kmdbh_shard_plan_counts: more than " + std::to_string(KMDB_MAX_SHARDS) + " shards");
    kmdb_shard_plan plan;
    std::vector<uint32_t> all;
    try { all.resize(n_shards); } catch (const std::exception&) { return kmdb_set_error("kmdbh_shard_plan_counts: out of host memory"); }
    for (uint32_t s = 0; s < n_shards; ++s) all[s] = s;
    if (kmdb_shard_plan_build(view, n_shards, all, &plan)) return 1;
    for (uint32_t s = 0; s < n_shards; ++s) {
        kept_nodes[s] = plan.kept[s];
        uint64_t sum = 0;
        for (uint64_t p = 0; p < plan.P; ++p) sum += plan.w[s][p];
        kmers[s] = sum;
    }
    return 0;
}

namespace {
// work(t) for t < T on T threads; a thread that cannot be started (std::system_error) has its share done by the caller, and every
// started thread is joined before anything propagates (a joinable std::thread that is destroyed calls std::terminate)
template <class F>
void run_shares(unsigned T, F&& work) {
    std::vector<std::thread> pool;
    std::vector<unsigned> inline_shares{0u};
    for (unsigned t = 1; t < T; ++t) {
        try { pool.emplace_back(work, t); }
        catch (...) { inline_shares.push_back(t); }
    }
    for (unsigned t : inline_shares) work(t);
    for (auto& th : pool) th.join();
}
}  // namespace

static int shard_plan_build_impl(const kmdb_db_view* v, uint32_t n_shards, const std::vector<uint32_t>& shards, kmdb_shard_plan* plan);
// (nothing may leave through the extern "C" entry points above this: allocation and thread failures end in kmdb_set_error — ADVICE round 5)
int kmdb_shard_plan_build(const kmdb_db_view* v, uint32_t n_shards, const std::vector<uint32_t>& shards, kmdb_shard_plan* plan) {
    // a shard per prefix bucket is the finest partition there is, and the plan keeps a counter array per shard: implausible counts are refused
    if (n_shards == 0 || n_shards > KMDB_MAX_SHARDS) return kmdb_set_error("kmdb_db_upload_shard: shard_count must be between 1 and " + std::to_string(KMDB_MAX_SHARDS));
    try {
        return shard_plan_build_impl(v, n_shards, shards, plan);
    } catch (const std::bad_alloc&) {
        return kmdb_set_error("kmdb_db_upload_shard: out of host memory for the shard plan");
    } catch (const std::exception& e) {
        return kmdb_set_error(std::string("kmdb_db_upload_shard: ") + e.what());
    }
}
static int shard_plan_build_impl(const kmdb_db_view* v, uint32_t n_shards, const std::vector<uint32_t>& shards, kmdb_shard_plan* plan) {
    const uint64_t P = v->n_patterns;
    if (!v->n_buckets) return kmdb_set_error("kmdb_db_upload_shard: the view carries no hashtables (load the database with mode Everything)");
    plan->P = P; plan->n_shards = n_shards;
    plan->w.assign(n_shards, nullptr);
    plan->mask.assign((n_shards + 7u) / 8u, nullptr);
    plan->kept.assign(n_shards, 0);
    for (auto& m : plan->mask) if (!(m = (unsigned char*)map_zero(P))) return kmdb_set_error("kmdb_db_upload_shard: out of host memory for the shard plan");
    for (uint32_t s : shards) {
        if (s >= n_shards) return kmdb_set_error("kmdb_db_upload_shard: shard_index >= shard_count");
        if (!plan->w[s] && !(plan->w[s] = (uint32_t*)map_zero(P * 4))) return kmdb_set_error("kmdb_db_upload_shard: out of host memory for the shard plan");
    }
    // ---- one pass over the hashtable items: every planned shard reads its own buckets (threads split the bucket range; a bucket
    // belongs to one shard, so the shards' counters never meet; the threads of one shard add with relaxed atomics)
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t n_slots = v->bucket_offset[v->n_buckets];
    const unsigned T = (unsigned)std::min<uint64_t>(std::min(64u, hw), std::max<uint64_t>(1, n_slots / (1u << 20)));
    {
        auto work = [&](unsigned t) {
            const uint64_t blo = v->n_buckets * t / T, bhi = v->n_buckets * (t + 1) / T;
            for (uint64_t b = blo; b < bhi; ++b) {
                const uint32_t s = (uint32_t)(b % n_shards);
                uint32_t* ws = plan->w[s];
                if (!ws) continue;
                unsigned char* mk = plan->mask[s >> 3];
                const unsigned char bit = (unsigned char)(1u << (s & 7u));
                for (uint64_t j = v->bucket_offset[b]; j < v->bucket_offset[b + 1]; ++j) {
                    const int32_t val = (int32_t)(v->slots[j] >> 32);
                    if (val == INT32_MAX || val < 0 || (uint64_t)val >= P) continue;      // empty slot (hashmap_lp.h: value INT32_MAX), or out of range (ignored like the device pass did)
                    if (__atomic_fetch_add(&ws[val], 1u, __ATOMIC_RELAXED) == 0u) __atomic_fetch_or(&mk[val], bit, __ATOMIC_RELAXED);
                }
            }
        };
        run_shares(T, work);
    }
    // ---- one sweep, children before parents: a kept node keeps its parent (all shards at once: the masks are ORed upwards)
    const size_t G = plan->mask.size();
    for (uint64_t p = P; p-- > 0;) {
        const int64_t par = v->parent_id[p];
        if (par < 0) continue;
        if ((uint64_t)par >= p) return kmdb_set_error("kmdb_db_upload: parent_id >= pattern id");
        for (size_t g = 0; g < G; ++g) { const unsigned char m = plan->mask[g][p]; if (m) plan->mask[g][par] |= m; }
    }
    // nodes kept per shard
    {
        std::vector<std::vector<uint64_t>> part(T, std::vector<uint64_t>(n_shards, 0));
        auto work = [&](unsigned t) {
            for (uint64_t p = P * t / T; p < P * (t + 1) / T; ++p)
                for (size_t g = 0; g < G; ++g) {
                    unsigned m = plan->mask[g][p];
                    while (m) { const unsigned k = (unsigned)__builtin_ctz(m); m &= m - 1u; ++part[t][g * 8 + k]; }
                }
        };
        run_shares(T, work);
        for (unsigned t = 0; t < T; ++t) for (uint32_t s = 0; s < n_shards; ++s) plan->kept[s] += part[t][s];
    }
    return 0;
}
