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
//           (src/hashmap_lp.h:53-64), key (query << pbits | DFS index of the hit pattern; 2^pbits > patterns: only the bits in use are sorted) or ~0
//   sort    radix sort of the keys: hits grouped by query, ascending DFS index inside a query
//   count   run-length encode (pattern, hits) + exclusive scan of the hit counts
//   walk    step (2) restated on the DFS layout of engine.hip.  With H(r) = hits in subtree(r) =
//           the hits whose DFS index lies in [r, sub_end[r]),
//                 similarities[s] = sum over nodes r that hold s as a LOCAL id of H(r).
//           Every node with H(r) > 0 is an ancestor-or-self of its first hit h_i and is larger than
//           the previous hit h_{i-1}; so thread i climbs from h_i through the parent links while the
//           node index stays above h_{i-1}: every such node is visited exactly once per query,
//           gets H(r) from two reads of the scanned counts (a doubling search from the previous
//           subtree end over the query's sorted hits), and adds it to its local ids in an LDS
//           histogram of the workgroup, flushed with one global atomic per touched sample.  A local
//           list of up to 32 ids is decoded by the visiting thread, a longer one by the whole workgroup
//           in pieces of 32 ids from the list index (checkpoints built with the node arrays).
#include "kmdb_amd.h"
#include "kmdb_internal.h"
#include "engine_internal.h"

#include <hip/hip_runtime.h>
#include "prim.h"
#include "hash_probe.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
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
                                 const uint32_t* __restrict__ pid2dfs, const uint32_t* __restrict__ w, uint32_t pbits,
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
                const int32_t val = kmdb_probe(slots, off, cap, (uint32_t)k);
                if (val != 0x7fffffff) {
                    const uint32_t d = pid2dfs[val];
                    if (w[d] != 0) {                              // :847-848 skips patterns without k-mers
                        // query of this k-mer: binary search in the batch's offsets
                        uint32_t lo = 0, hi = nq;
                        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (qoff[mid] <= i) lo = mid; else hi = mid; }
                        key = ((unsigned long long)lo << pbits) | d;
                    }
                }
            }
        }
        keys[i] = key;
    }
}

// first run of every query in the sorted, run-length-encoded hit list
__global__ void n2a_query_ranges_kernel(const unsigned long long* __restrict__ uniq, uint32_t nruns, uint32_t nq, uint32_t pbits, uint32_t* __restrict__ qstart) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nq) return;
    const unsigned long long target = (unsigned long long)q << pbits;
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
    // the "0" codes (deltas of 1) at the cursor, at most `limit` of them: taken in one step
    __device__ __forceinline__ uint32_t zeros(uint32_t limit) {
        const uint64_t win = s ? ((c0 << s) | (c1 >> (64u - s))) : c0;
        uint32_t z = win ? (uint32_t)__clzll((long long)win) : 64u;
        z = z < limit ? z : limit;
        s += z;
        if (s >= 64u) { s -= 64u; ++wi; c0 = c1; c1 = bits[wi + 1]; }
        return z;
    }
};

// (4) walk: one thread per distinct (query, hit pattern).  A thread climbs from its hit towards the root until it meets the
// previous hit's path, so every node on the union of the root paths is visited once per query; the node's local ids get H =
// the number of hit k-mers below it.  Local lists of up to KMDB_CK_IDS ids are decoded by the visiting thread; longer ones
// (a few nodes near the root carry thousands of ids) go to a queue in LDS and are decoded afterwards by ALL threads of the
// workgroup in pieces of KMDB_CK_IDS ids, each piece starting from a checkpoint of the list index (engine_state.h) —
// measured before that: 8 % of the lanes active, the others waiting for a neighbour's long list.
// The threads of a workgroup share one per-query histogram in LDS (4 B per sample).  512 threads: with 10 000 samples two
// workgroups fit a CU (4 waves per SIMD).  1024 threads would fill the SIMDs, and were measured slower (71.4 against 59.0 ms per
// 1000 queries: twice the threads behind every workgroup barrier and on the same histogram); KMDB_N2A_THREADS=1024 runs them.
// ---- the run index: node i's local ids as runs of consecutive ids.  FILL = false counts the runs, true writes them (start | length << rs,
// rs = n2a_run_shift(N): 16 bits of start up to 65 536 samples, else KMDB_ID_BITS, and a run too long for the bits left is stored as
// several).  One thread per node; pattern_t::decodeSamples (src/pattern.cpp:99-109): first id = last - sum of the deltas.
__host__ __device__ inline uint32_t n2a_run_shift(uint32_t N) { return N <= 65536u ? 16u : KMDB_ID_BITS; }
template <bool FILL>
__global__ void n2a_runs_kernel(const uint4* __restrict__ meta, const uint64_t* __restrict__ bitpos, const uint64_t* __restrict__ bits, uint32_t P, uint32_t N,
                                uint32_t* __restrict__ cnt, const uint32_t* __restrict__ ofs, uint32_t* __restrict__ runs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > P) return;
    uint32_t n = 0;
    const uint32_t rs = n2a_run_shift(N), max_len = (1u << (32u - rs)) - 1u;
    if (i < P) {
        const uint4 m = meta[i];
        const uint32_t l = m.y;
        if (l) {
            uint32_t id = m.z;
            if (l > 1) {
                N2Cursor c1(bits, bitpos[i]);
                uint32_t sum = 0;
                for (uint32_t rem = l - 1u; rem;) {
                    const uint32_t z = c1.zeros(rem);
                    sum += z; rem -= z;
                    if (rem) { sum += c1.next(); --rem; }
                }
                id = m.z - sum;
            }
            uint32_t o = FILL ? ofs[i] : 0u, start = id, len = 1;
            auto put = [&]() {                                       // the run [start, start + len)
                uint32_t s0 = start, left = len;
                while (left) {
                    const uint32_t take = left < max_len ? left : max_len;
                    if (FILL) runs[o] = s0 | (take << rs);
                    ++o; ++n; s0 += take; left -= take;
                }
            };
            if (l > 1) {
                N2Cursor c2(bits, bitpos[i]);
                for (uint32_t rem = l - 1u; rem;) {
                    const uint32_t z = c2.zeros(rem);
                    len += z; rem -= z;
                    if (rem) {
                        const uint32_t d = c2.next();
                        --rem;
                        if (d == 1u) ++len;
                        else { put(); start += len - 1u + d; len = 1; }
                    }
                }
            }
            put();
        }
    }
    if (!FILL) cnt[i] = n;
}

// the node as the walk's climb reads it: ONE 16-byte record instead of five loads from four arrays (subtree end, list header, parent, two run
// offsets).  The climb is random access, a node per step and thread: what it moves are memory sectors, not bytes.
__global__ void n2a_nodes_kernel(const uint4* __restrict__ meta, const int32_t* __restrict__ parent, const uint32_t* __restrict__ sub_end,
                                 const uint32_t* __restrict__ ofs, uint32_t P, uint4* __restrict__ node) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint4 m = meta[i];
    const uint32_t l = m.y, ro = ofs[i], nr = ofs[i + 1] - ro;
    node[i] = make_uint4(sub_end[i], (uint32_t)parent[i], l == 1u ? m.z : ro, (l < 0xFFFFu ? l : 0xFFFFu) | ((nr < 0xFFFFu ? nr : 0xFFFFu) << 16));
}

// RUNIDX: the local lists come from the handle's run index (runs of consecutive ids, decoded once per handle) instead of the gamma
// streams: every query used to decode the same lists again — the clade-level nodes are on the root paths of all queries of the clade —
// and the queued long lists were half of the walk's time.
constexpr uint32_t N2_RUNS_PER_PIECE = 8;  // runs of a queued long list per thread and step
template <bool LDS_HIST, uint32_t N2_THREADS, uint32_t N2_QCAP, bool RUNIDX>
__global__ __launch_bounds__(N2_THREADS) void n2a_walk_kernel(const unsigned long long* __restrict__ uniq, const uint32_t* __restrict__ csum,
                                                       const uint32_t* __restrict__ qstart, uint32_t nruns, uint32_t nq,
                                                       const uint4* __restrict__ meta, const uint64_t* __restrict__ bitpos,
                                                       const int32_t* __restrict__ parent, const uint32_t* __restrict__ sub_end,
                                                       const uint64_t* __restrict__ bits, const uint32_t* __restrict__ ck_ofs,
                                                       const uint64_t* __restrict__ ck_bit, const uint32_t* __restrict__ ck_id,
                                                       const uint32_t* __restrict__ rl_ofs, const uint32_t* __restrict__ rl_runs, const uint4* __restrict__ rl_node, uint32_t N, uint32_t pbits,
                                                       uint32_t* __restrict__ sim) {
    const unsigned long long pmask = (1ull << pbits) - 1ull;             // key = query << pbits | pattern; an unused slot's ~0 reads as the largest pattern
    const uint32_t rs = n2a_run_shift(N), rmask = (1u << rs) - 1u;       // run of the run index = start | length << rs
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t q_node[N2_QCAP], q_h[N2_QCAP], q_l[N2_QCAP], q_pre[N2_QCAP + 1], part[N2_THREADS / 64];      // part: wave totals of the scans
    __shared__ uint32_t q_n;
    // The workgroup's own stretch of the sorted hit list (patterns, scanned counts; one entry beyond) stays in LDS: near the leaves a
    // subtree ends a few hits further on, so the searches of the climb mostly stay inside it instead of going to memory step by step.
    __shared__ uint32_t w_pat[N2_THREADS + 1], w_cs[N2_THREADS + 1];
    const uint32_t blk0 = blockIdx.x * blockDim.x;
    const uint32_t i = blk0 + threadIdx.x;
    const unsigned long long my_key = i < nruns ? uniq[i] : N2_INVALID;
    const bool live = my_key != N2_INVALID;
    const uint32_t myq = live ? (uint32_t)(my_key >> pbits) : 0xFFFFFFFFu;
    w_pat[threadIdx.x] = (uint32_t)(my_key & pmask);
    w_cs[threadIdx.x] = i <= nruns ? csum[i] : 0u;
    if (threadIdx.x == 0) {
        const uint32_t j = blk0 + N2_THREADS;
        w_pat[N2_THREADS] = j < nruns ? (uint32_t)(uniq[j] & pmask) : 0xFFFFFFFFu;
        w_cs[N2_THREADS] = j <= nruns ? csum[j] : 0u;
    }
    auto pat_at = [&](uint32_t x) -> uint32_t { const uint32_t o = x - blk0; return o <= N2_THREADS ? w_pat[o] : (uint32_t)(uniq[x] & pmask); };      // x >= blk0
    auto cs_at = [&](uint32_t x) -> uint32_t { const uint32_t o = x - blk0; return o <= N2_THREADS ? w_cs[o] : csum[x]; };
    // the block's runs are sorted by query: loop over the (few) queries it spans
    __shared__ uint32_t q_lo, q_hi;
    if (threadIdx.x == 0) { q_lo = 0xFFFFFFFFu; q_hi = 0; }
    __syncthreads();
    if (live) { atomicMin(&q_lo, myq); atomicMax(&q_hi, myq); }
    __syncthreads();
    if (q_lo == 0xFFFFFFFFu) return;
    for (uint32_t q = q_lo; q <= q_hi; ++q) {
        if (LDS_HIST)
            for (uint32_t s = threadIdx.x; s < N; s += blockDim.x) hist[s] = 0;
        if (threadIdx.x == 0) q_n = 0;
        __syncthreads();
        uint32_t* acc = LDS_HIST ? hist : (sim + (size_t)q * N);
        if (live && myq == q) {
            const uint32_t qs = qstart[q], qe = qstart[q + 1];
            const uint32_t h = (uint32_t)(my_key & pmask);
            const int64_t prev = i > qs ? (int64_t)(threadIdx.x ? w_pat[threadIdx.x - 1u] : (uint32_t)(uniq[i - 1] & pmask)) : -1;
            const uint32_t cbase = w_cs[threadIdx.x];
            int64_t r = h;
            uint32_t ub = i + 1;                               // grows while climbing: an ancestor's subtree contains the node's
            // A step's loads depend on each other (subtree end -> search over the hits -> count; list header -> stream -> ids; parent):
            // the records of the NEXT node of the path are fetched at the top of a step, before the current node's search and decode,
            // so the dependent round trips of consecutive steps overlap.
            // (RUNIDX: the whole node is one 16-byte record — m.y = l (clipped), m.z = the only id of a one-id list; else the arrays of the engine)
            uint32_t se, ro = 0, re = 0;
            uint4 m;
            int32_t par;
            uint64_t bp = 0;
            auto node_of = [&](int64_t x, uint32_t& o_se, uint4& o_m, int32_t& o_par, uint64_t& o_bp, uint32_t& o_ro, uint32_t& o_re) {
                if (RUNIDX && rl_node) {
                    const uint4 v = rl_node[x];
                    const uint32_t nr = v.w >> 16;
                    o_se = v.x; o_par = (int32_t)v.y; o_m = make_uint4(0u, v.w & 0xFFFFu, v.z, 0u); o_ro = v.z;
                    o_re = nr == 0xFFFFu ? rl_ofs[x + 1] : v.z + nr;
                } else {
                    o_se = sub_end[x]; o_m = meta[x]; o_par = parent[x];
                    if (RUNIDX) { o_ro = rl_ofs[x]; o_re = rl_ofs[x + 1]; } else o_bp = bitpos[x];
                }
            };
            node_of(r, se, m, par, bp, ro, re);
            while (r > prev) {
                const int64_t rn = par;
                uint32_t se_n = 0;
                uint4 m_n = make_uint4(0u, 0u, 0u, 0u);
                int32_t par_n = -1;
                uint64_t bp_n = 0;
                uint32_t ro_n = 0, re_n = 0;
                if (rn > prev) node_of(rn, se_n, m_n, par_n, bp_n, ro_n, re_n);
                // hits below r: indices [i, ub), ub = first run of this query whose pattern is >= sub_end[r].  Searched from the
                // previous ub in doubling steps: near the leaves a subtree holds a handful of hits, one or two probes find its end
                uint32_t lo = ub, hi = ub, step = 1;
                while (hi < qe && pat_at(hi) < se) { lo = hi + 1; hi += step; step <<= 1; }
                if (hi > qe) hi = qe;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (pat_at(mid) < se) lo = mid + 1; else hi = mid; }
                ub = lo;
                const uint32_t H = cs_at(lo) - cbase;
                const uint32_t l = m.y;
                bool inl = l != 0;
                if (l > KMDB_CK_IDS) {
                    const uint32_t slot = atomicAdd(&q_n, 1u);
                    if (slot < N2_QCAP) { q_node[slot] = RUNIDX ? ro : ck_ofs[r]; q_h[slot] = H; q_l[slot] = RUNIDX ? re - ro : l; inl = false; }      // (the node's first run / checkpoint; a full queue: added here)
                }
                if (inl && RUNIDX && l > 1) {
                    for (uint32_t k = ro; k < re; ++k) {
                        const uint32_t run = rl_runs[k];
                        for (uint32_t id = run & rmask, t = run >> rs; t; --t, ++id) atomicAdd(&acc[id], H);
                    }
                } else if (inl) {
                    uint32_t id = m.z;
                    if (l > 1) {
                        // pattern_t::decodeSamples (src/pattern.cpp:99-109): first id = last - sum of the deltas
                        N2Cursor c1(bits, bp);
                        uint32_t sum = 0;
                        for (uint32_t rem = l - 1u; rem;) {                 // runs of deltas of 1 in one step each
                            const uint32_t z = c1.zeros(rem);
                            sum += z; rem -= z;
                            if (rem) { sum += c1.next(); --rem; }
                        }
                        id = m.z - sum;
                        N2Cursor c2(bits, bp);
                        for (uint32_t rem = l - 1u; rem;) {
                            const uint32_t z = c2.zeros(rem);
                            for (uint32_t t = 0; t < z; ++t) { atomicAdd(&acc[id], H); ++id; }
                            rem -= z;
                            if (rem) { atomicAdd(&acc[id], H); id += c2.next(); --rem; }
                        }
                    }
                    atomicAdd(&acc[id], H);
                }
                r = rn; se = se_n; m = m_n; par = par_n; bp = bp_n; ro = ro_n; re = re_n;
            }
        }
        __syncthreads();
        // ---- the queued long lists, KMDB_CK_IDS ids per thread and step
        const uint32_t nt = q_n < N2_QCAP ? q_n : N2_QCAP;
        if (nt) {
            constexpr uint32_t PER = N2_QCAP / N2_THREADS ? N2_QCAP / N2_THREADS : 1u;
            uint32_t sum = 0;
            constexpr uint32_t UNIT = RUNIDX ? N2_RUNS_PER_PIECE : KMDB_CK_IDS;      // runs / ids per piece
            for (uint32_t t = threadIdx.x * PER; t < nt && t < (threadIdx.x + 1u) * PER; ++t) sum += (q_l[t] + UNIT - 1u) / UNIT;
            // exclusive scan of the threads' piece counts: inside the waves by shuffles, the wave totals through LDS
            const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
            uint32_t incl = sum;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const uint32_t o2 = (uint32_t)__shfl_up((int)incl, dd, 64);
                if (lane >= (uint32_t)dd) incl += o2;
            }
            if (lane == 63u) part[wv] = incl;
            __syncthreads();
            uint32_t run = incl - sum, all = 0;
            for (uint32_t w2 = 0; w2 < N2_THREADS / 64u; ++w2) { const uint32_t pw = part[w2]; if (w2 < wv) run += pw; all += pw; }
            for (uint32_t t = threadIdx.x * PER; t < nt && t < (threadIdx.x + 1u) * PER; ++t) { q_pre[t] = run; run += (q_l[t] + UNIT - 1u) / UNIT; }
            if (threadIdx.x == 0) q_pre[nt] = all;
            __syncthreads();
            const uint32_t S = q_pre[nt];
            // The long lists are mostly RUNS of consecutive ids (a clade's samples), many of them the same runs, and adding H to
            // every id of every run is what this kernel waited for (same-address LDS atomics).  With the histogram in LDS the
            // queued lists go in as differences instead: +H at the first id of a run, -H behind its last — two atomics per run —
            // and one prefix sum over the histogram turns them back into counts.  The short lists of the climb were added
            // directly: the histogram is brought to difference form first (d[s] = h[s] - h[s-1]), in place, every thread its own
            // contiguous range.
            const uint32_t per = ((N + N2_THREADS - 1u) / N2_THREADS) | 1u;          // odd: the threads' ranges start in different LDS banks
            const uint32_t ra = threadIdx.x * per < N ? threadIdx.x * per : N, rb = ra + per < N ? ra + per : N;
            if (LDS_HIST) {
                uint32_t prev = (ra && ra < N) ? hist[ra - 1u] : 0u;
                __syncthreads();
                for (uint32_t s2 = ra; s2 < rb; ++s2) { const uint32_t v = hist[s2]; hist[s2] = v - prev; prev = v; }
                __syncthreads();
            }
            for (uint32_t g = threadIdx.x; g < S; g += N2_THREADS) {
                uint32_t lo = 0, hi = nt;                       // the task whose pieces contain g: last t with q_pre[t] <= g
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (q_pre[mid] <= g) lo = mid; else hi = mid; }
                const uint32_t piece = g - q_pre[lo], H = q_h[lo], l = q_l[lo];
                if (RUNIDX) {
                    const uint32_t k0 = q_node[lo] + piece * UNIT, left_r = l - piece * UNIT, nk = left_r < UNIT ? left_r : UNIT;
                    for (uint32_t k = 0; k < nk; ++k) {
                        const uint32_t run = rl_runs[k0 + k], start = run & rmask, len = run >> rs;
                        if (LDS_HIST) {
                            atomicAdd(&hist[start], H);
                            if (start + len < N) atomicAdd(&hist[start + len], 0u - H);
                        } else for (uint32_t t = 0; t < len; ++t) atomicAdd(&acc[start + t], H);
                    }
                    continue;
                }
                const uint32_t o = q_node[lo] + piece;
                uint32_t id = ck_id[o];
                const uint32_t left = l - piece * KMDB_CK_IDS, cnt = left < KMDB_CK_IDS ? left : KMDB_CK_IDS;
                N2Cursor c(bits, ck_bit[o]);
                if (LDS_HIST) {
                    uint32_t start = id, len = 1, rem = cnt - 1u;          // the run [start, start + len) under construction
                    while (rem) {
                        const uint32_t z = c.zeros(rem);
                        len += z; rem -= z;
                        if (rem) {
                            const uint32_t d = c.next();
                            --rem;
                            if (d == 1u) ++len;
                            else {
                                atomicAdd(&hist[start], H);
                                if (start + len < N) atomicAdd(&hist[start + len], 0u - H);
                                start += len - 1u + d; len = 1;
                            }
                        }
                    }
                    atomicAdd(&hist[start], H);
                    if (start + len < N) atomicAdd(&hist[start + len], 0u - H);
                } else {
                    atomicAdd(&acc[id], H);
                    for (uint32_t t = 1; t < cnt; ++t) { id += c.next(); atomicAdd(&acc[id], H); }
                }
            }
            if (LDS_HIST) {
                // back to counts: prefix sum over the histogram (range sums, scan of the 512 range sums by waves, ranges rewritten)
                __syncthreads();
                uint32_t sum2 = 0;
                for (uint32_t s2 = ra; s2 < rb; ++s2) sum2 += hist[s2];
                uint32_t incl2 = sum2;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const uint32_t o2 = (uint32_t)__shfl_up((int)incl2, dd, 64);
                    if (lane >= (uint32_t)dd) incl2 += o2;
                }
                if (lane == 63u) part[wv] = incl2;
                __syncthreads();
                uint32_t run2 = incl2 - sum2;
                for (uint32_t w2 = 0; w2 < wv; ++w2) run2 += part[w2];
                for (uint32_t s2 = ra; s2 < rb; ++s2) { run2 += hist[s2]; hist[s2] = run2; }
            }
        }
        if (LDS_HIST) {
            __syncthreads();
            uint32_t* out = sim + (size_t)q * N;
            for (uint32_t s = threadIdx.x; s < N; s += blockDim.x)
                if (hist[s]) atomicAdd(&out[s], hist[s]);
        }
        __syncthreads();
    }
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void reset() { if (p) (void)hipFree(p); p = nullptr; }
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

struct N2aU32toU64 { __host__ __device__ unsigned long long operator()(uint32_t v) const { return v; } };
// the handle's run index (false: this handle does without one — no memory for it — and the walk decodes the gamma streams as before)
static bool n2a_run_index(const kmdb_engine_view& e, hipStream_t st) {
    if (*e.rl_ofs && *e.rl_runs && *e.rl_node) return true;
    if (*e.rl_tried || !e.P || getenv("KMDB_N2A_NO_RUNS")) return false;
    *e.rl_tried = true;
    const uint32_t P = (uint32_t)e.P;
    uint32_t *cnt = nullptr, *ofs = nullptr, *runs = nullptr;
    void* tmp = nullptr;
    auto fail = [&]() { (void)hipGetLastError(); for (void* p : {(void*)cnt, (void*)ofs, (void*)runs, tmp}) if (p) (void)hipFree(p); return false; };
    if (hipMalloc((void**)&cnt, ((size_t)P + 1) * 4) != hipSuccess || hipMalloc((void**)&ofs, ((size_t)P + 1) * 4) != hipSuccess) return fail();
    hipLaunchKernelGGL((n2a_runs_kernel<false>), dim3((P + 1 + 255) / 256), dim3(256), 0, st, e.meta, e.bitpos, e.bits, P, (uint32_t)e.N, cnt, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    size_t tb = 0;
    if (prim::exclusive_sum(nullptr, tb, cnt, ofs, (int)(P + 1), st) != hipSuccess || hipMalloc(&tmp, std::max<size_t>(tb, 16)) != hipSuccess) return fail();
    if (prim::exclusive_sum(tmp, tb, cnt, ofs, (int)(P + 1), st) != hipSuccess) return fail();
    // the offsets are 32 bits wide, and the number of runs is bounded by the sum of the local lists' lengths over up to 2^31 patterns, not
    // by the sample count: the counts are summed in 64 bits first, and a database with 2^32 runs or more does without the index
    // (a wrapped total would have under-allocated `runs`)
    {
        unsigned long long* d_sum = nullptr;
        void* tmp2 = nullptr;
        size_t tb2 = 0;
        rocprim::transform_iterator<uint32_t*, N2aU32toU64, unsigned long long> it(cnt, N2aU32toU64());
        unsigned long long h_sum = 0;
        bool ok = hipMalloc((void**)&d_sum, 8) == hipSuccess && prim::sum(nullptr, tb2, it, d_sum, (int)(P + 1), st) == hipSuccess &&
                  hipMalloc(&tmp2, std::max<size_t>(tb2, 16)) == hipSuccess && prim::sum(tmp2, tb2, it, d_sum, (int)(P + 1), st) == hipSuccess &&
                  hipMemcpyAsync(&h_sum, d_sum, 8, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        if (d_sum) (void)hipFree(d_sum);
        if (tmp2) (void)hipFree(tmp2);
        if (!ok || h_sum >= (1ull << 32)) return fail();
    }
    uint32_t total = 0;
    if (hipMemcpyAsync(&total, ofs + P, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail();
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (size_t)total * 4 + (size_t)P * sizeof(uint4) + (1ull << 30) > free_b / 2) return fail();
    if (hipMalloc((void**)&runs, std::max<size_t>((size_t)total, 1) * 4) != hipSuccess) return fail();
    hipLaunchKernelGGL((n2a_runs_kernel<true>), dim3((P + 1 + 255) / 256), dim3(256), 0, st, e.meta, e.bitpos, e.bits, P, (uint32_t)e.N, (uint32_t*)nullptr, (const uint32_t*)ofs, runs);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail();
    uint4* nodes = nullptr;
    if (hipMalloc((void**)&nodes, (size_t)P * sizeof(uint4)) != hipSuccess) return fail();
    hipLaunchKernelGGL(n2a_nodes_kernel, dim3((P + 255) / 256), dim3(256), 0, st, e.meta, e.parent, e.sub_end, (const uint32_t*)ofs, P, nodes);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipFree(nodes); return fail(); }
    (void)hipFree(cnt); (void)hipFree(tmp);
    *e.rl_ofs = ofs; *e.rl_runs = runs; *e.rl_node = nodes;
    *e.device_bytes += ((uint64_t)P + 1) * 4 + (uint64_t)total * 4 + (uint64_t)P * sizeof(uint4);
    if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] new2all: run index of %u nodes, %u runs (%.2f GB)\n", P, total, (((double)P + 1) * 4 + (double)total * 4) / 1e9);
    return true;
}

// probe / sort / count / walk over a batch whose k-mers (sorted and unique per query, query by query) and
// query offsets are already on the device
static int n2a_run(kmdb_db* dbh, const kmdb_engine_view& e, hipStream_t st, const uint64_t* d_kmers, const uint64_t* d_qoff_p,
                   size_t total, size_t nq, uint32_t* out_dense) {
    const uint64_t N = e.N;
    DevBuf d_keys, d_keys2, d_uniq, d_cnt, d_csum, d_nruns, d_qstart, d_sim, d_tmp;
    N2_TRY(d_keys.alloc(total * 8));
    N2_TRY(d_keys2.alloc(total * 8));
    N2_TRY(d_uniq.alloc((total + 1) * 8));
    N2_TRY(d_cnt.alloc((total + 2) * 4));
    N2_TRY(d_csum.alloc((total + 2) * 4));
    N2_TRY(d_nruns.alloc(16));
    N2_TRY(d_qstart.alloc((nq + 2) * 4));
    N2_TRY(d_sim.alloc(nq * N * 4));
    size_t tb_sort = 0, tb_rle = 0, tb_scan = 0;
    N2_TRY(prim::sort_keys(nullptr, tb_sort, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                             (int)total, 0, 64, st));
    N2_TRY(prim::run_length_encode(nullptr, tb_rle, d_keys2.as<unsigned long long>(), d_uniq.as<unsigned long long>(),
                                                 d_cnt.as<uint32_t>(), d_nruns.as<uint32_t>(), (int)total, st));
    N2_TRY(prim::exclusive_sum(nullptr, tb_scan, d_cnt.as<uint32_t>(), d_csum.as<uint32_t>(), (int)(total + 1), st));
    N2_TRY(d_tmp.alloc(std::max(tb_sort, std::max(tb_rle, tb_scan))));
    N2_TRY(hipMemsetAsync(d_sim.p, 0, std::max<uint64_t>(nq * N * 4, 4), st));

    // the run index of the handle: made on its first new2all call (outside the call's device time, inside its wall time)
    const bool runidx = n2a_run_index(e, st);
    hipEvent_t ev0 = (hipEvent_t)e.ev[0], ev3 = (hipEvent_t)e.ev[3];
    N2_TRY(hipEventRecord(ev0, st));
    uint32_t nruns = 0;
    if (total) {
        const unsigned blocks = (unsigned)std::min<size_t>(65535, (total + 255) / 256);
        // key = query << pbits | pattern with 2^pbits > patterns, 2^qbits >= queries: only the bits in use are sorted (the unused slots' ~0
        // is all ones in those too, and no key in use is — its pattern bits are below 2^pbits - 1 or its query bits below 2^qbits - 1...
        // the pattern field of a key in use is a DFS index < P <= 2^pbits - 1)
        uint32_t pbits = 1, qbits = 1;
        while ((1ull << pbits) <= e.P) ++pbits;
        while ((1ull << qbits) < nq) ++qbits;
        hipLaunchKernelGGL(n2a_probe_kernel, dim3(blocks), dim3(256), 0, st, d_kmers, d_qoff_p, (uint32_t)nq,
                           total, e.n_buckets, e.bucket_offset, e.slots, e.pid2dfs, e.w, pbits, d_keys.as<unsigned long long>());
        N2_TRY(hipGetLastError());
        N2_TRY(prim::sort_keys(d_tmp.p, tb_sort, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                                 (int)total, 0, std::min(64u, pbits + qbits), st));
        N2_TRY(hipMemsetAsync(d_cnt.p, 0, (total + 2) * 4, st));
        N2_TRY(prim::run_length_encode(d_tmp.p, tb_rle, d_keys2.as<unsigned long long>(), d_uniq.as<unsigned long long>(),
                                                     d_cnt.as<uint32_t>(), d_nruns.as<uint32_t>(), (int)total, st));
        N2_TRY(hipMemcpyAsync(&nruns, d_nruns.p, 4, hipMemcpyDeviceToHost, st));
        N2_TRY(hipStreamSynchronize(st));
        N2_TRY(prim::exclusive_sum(d_tmp.p, tb_scan, d_cnt.as<uint32_t>(), d_csum.as<uint32_t>(), (int)(nruns + 1), st));
        hipLaunchKernelGGL(n2a_query_ranges_kernel, dim3((unsigned)((nq + 1 + 255) / 256)), dim3(256), 0, st,
                           d_uniq.as<unsigned long long>(), nruns, (uint32_t)nq, pbits, d_qstart.as<uint32_t>());
        if (nruns) {
            // LDS of a workgroup: the per-query histogram (4 B per sample) + the queue of long lists (16 B per entry) + 12 B per thread
            // (scan scratch, the workgroup's stretch of the hit list)
            // (a queue of 256 entries would let a third workgroup onto a CU and was measured 1.5 x slower: every second hit visits
            // a node with more than 32 local ids, a full queue leaves those to single threads)
            uint32_t threads = 512;
            const uint32_t qcap = 1024;          // (512 entries: a third workgroup per CU, and 47.1 against 45.9 ms — the queue fills up)
            if (const char* ev = getenv("KMDB_N2A_THREADS")) if (atoi(ev) == 1024) threads = 1024;
            const bool lds_hist = N * 4 + qcap * 16 + threads * 8 + 1024 <= 64 * 1024;
            const bool packed_nodes = getenv("KMDB_N2A_NO_NODES") == nullptr;        // (A/B: the climb reads the engine's arrays, five loads per node)
            const unsigned wblocks = (nruns + threads - 1) / threads;
#define N2A_WALK(H, T, Q, R)                                                                                                             \
    hipLaunchKernelGGL((n2a_walk_kernel<H, T, Q, R>), dim3(wblocks), dim3(T), (H) ? N * 4 : 0, st, d_uniq.as<unsigned long long>(),      \
                       d_csum.as<uint32_t>(), d_qstart.as<uint32_t>(), nruns, (uint32_t)nq, e.meta, e.bitpos, e.parent, e.sub_end,       \
                       e.bits, e.ck_ofs, e.ck_bit, e.ck_id, (const uint32_t*)*e.rl_ofs, (const uint32_t*)*e.rl_runs, packed_nodes ? (const uint4*)*e.rl_node : (const uint4*)nullptr, (uint32_t)N, pbits, d_sim.as<uint32_t>())
            if (runidx) {
                if (threads == 1024) { if (lds_hist) N2A_WALK(true, 1024, 1024, true); else N2A_WALK(false, 1024, 1024, true); }
                else { if (lds_hist) N2A_WALK(true, 512, 1024, true); else N2A_WALK(false, 512, 1024, true); }
            } else {
                if (threads == 1024) { if (lds_hist) N2A_WALK(true, 1024, 1024, false); else N2A_WALK(false, 1024, 1024, false); }
                else { if (lds_hist) N2A_WALK(true, 512, 1024, false); else N2A_WALK(false, 512, 1024, false); }
            }
#undef N2A_WALK
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

static int new2all_once(kmdb_db* dbh, const uint64_t* const* kmers, const size_t* counts, size_t nq, uint32_t* out_dense, const kmdb_opts* opts);

extern "C" int kmdb_new2all_batch(kmdb_db* dbh, const uint64_t* const* kmers, const size_t* counts, size_t nq,
                                  uint32_t* out_dense, const kmdb_opts* opts) {
    if (!dbh || (nq && (!kmers || !counts || !out_dense))) return kmdb_set_error("kmdb_new2all_batch: null argument");
    kmdb_engine_view e;
    if (kmdb_engine_get(dbh, &e)) return 1;
    for (size_t q0 = 0; q0 < nq;) {
        size_t q1 = q0;
        uint64_t total = 0;
        do { total += counts[q1]; ++q1; } while (q1 < nq && total + counts[q1] <= (1ull << 30));      // k-mers per piece
        if (new2all_once(dbh, kmers + q0, counts + q0, q1 - q0, out_dense + q0 * e.N, opts)) return 1;
        q0 = q1;
    }
    return 0;
}

static int new2all_once(kmdb_db* dbh, const uint64_t* const* kmers, const size_t* counts, size_t nq, uint32_t* out_dense, const kmdb_opts* opts) {
    kmdb_engine_view e;
    if (kmdb_engine_get(dbh, &e)) return 1;
    if (!e.n_buckets || !e.slots) return kmdb_set_error("kmdb_new2all_batch: database was uploaded without hashtables");
    if (nq >= (1ull << 31)) return kmdb_set_error("kmdb_new2all_batch: too many queries in one batch");
    N2_TRY(hipSetDevice(e.device));
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : (hipStream_t)e.stream;
    std::vector<uint64_t> qoff(nq + 1, 0);
    for (size_t q = 0; q < nq; ++q) qoff[q + 1] = qoff[q] + counts[q];
    const size_t total = qoff[nq];
    if (total >= (1ull << 32) - 2) return kmdb_set_error("kmdb_new2all_batch: a single query of 2^32 k-mers or more is not supported");
    if (!nq) return 0;
    DevBuf d_k, d_qoff;
    N2_TRY(d_k.alloc(total * 8));
    N2_TRY(d_qoff.alloc((nq + 1) * 8));
    for (size_t q = 0; q < nq; ++q)
        if (counts[q]) N2_TRY(hipMemcpyAsync(d_k.as<uint64_t>() + qoff[q], kmers[q], counts[q] * 8, hipMemcpyHostToDevice, st));
    N2_TRY(hipMemcpyAsync(d_qoff.p, qoff.data(), (nq + 1) * 8, hipMemcpyHostToDevice, st));
    return n2a_run(dbh, e, st, d_k.as<uint64_t>(), d_qoff.as<uint64_t>(), total, nq, out_dense);
}

// ------------------------------------------------------------------------------------------
// Query-side k-mer extraction on the device: what the reference's loader + KmerHelper::unique do on
// the host for every query (src/kmer_extract.h:13-118, src/filter.h:28-115, src/console_new2all.cpp:73).
//   extract  one thread per sequence position: 2 bits per base (A,C,G,T/U = 0..3, either case), canonical k-mer =
//            min(forward, reverse complement) unless the strand is preserved, a window with any other symbol or
//            one that crosses a query boundary is dropped, word widened when 2k-32 < 8 (k=18: 40-bit words),
//            optional minhash subsampling (MurmurHash3-derived 64-bit hash in [lo, hi))
//   sort     radix sort by k-mer, then (stable) by query: dropped positions carry query ~0 and end up last
//   unique   head flags + scan: the flat list of sorted unique k-mers query by query, and the per-query counts
// ------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ uint64_t n2_mix64(uint64_t v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull;
    v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull;
    v ^= v >> 33;
    return v;
}

struct ExtractParams {
    const char* seq; const uint64_t* soff; uint32_t nq; uint64_t total_len;
    uint32_t k, widen; int preserve, subsample; uint64_t lo, hi, quarter_len;
    unsigned long long* kmer; uint32_t* qid;
    uint32_t bits, size;           // bits per symbol and symbols of the alphabet (alphabet.h:35-36); nt: 2, 4
    int8_t map[256];               // Alphabet::mapping (alphabet.h:41-58) — travels with the kernel arguments
};

__global__ __launch_bounds__(256) void n2a_extract_kernel(ExtractParams p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.total_len) return;
    // query of this position
    uint32_t lo = 0, hi = p.nq;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (p.soff[mid] <= i) lo = mid; else hi = mid; }
    unsigned long long word = 0;
    uint32_t q = 0xFFFFFFFFu;
    if (i + p.k <= p.soff[lo + 1]) {
        uint64_t fwd = 0, rc = 0;
        bool ok = true;
        const unsigned top = p.bits * (p.k - 1);
        for (uint32_t t = 0; t < p.k; ++t) {
            const int c = p.map[(unsigned char)p.seq[i + t]];
            if (c < 0) { ok = false; break; }
            fwd = (fwd << p.bits) | (uint64_t)c;
            rc = (rc >> p.bits) | ((uint64_t)(p.size - 1u - (uint32_t)c) << top);     // (kmer_extract.h:73; compared only where the strand is not preserved: nt)
        }
        if (ok) {
            uint64_t w = (p.preserve || fwd < rc) ? fwd : rc;
            w = (w << p.widen) | (w & ((1ull << p.widen) - 1ull));
            bool keep = true;
            if (p.subsample) {
                uint64_t a = w * 0x87c37b91114253d5ull;
                a = (a << 31) | (a >> 33);
                a *= 0x4cf5ad432745937full;
                uint64_t h1 = (42ull ^ a) ^ p.quarter_len, h2 = 42ull ^ p.quarter_len;
                h1 += h2; h2 += h1;
                h1 = n2_mix64(h1); h2 = n2_mix64(h2);
                h1 += h2; h2 += h1;
                const uint64_t h = h1 ^ h2;
                keep = h >= p.lo && h < p.hi;
            }
            if (keep) { word = w; q = lo; }
        }
    }
    p.kmer[i] = word;
    p.qid[i] = q;
}

// head of a run of equal (query, k-mer) among the sorted positions
__global__ void n2a_heads_kernel(const unsigned long long* __restrict__ kmer, const uint32_t* __restrict__ qid, uint64_t n,
                                 uint32_t* __restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = qid[i];
    head[i] = (q != 0xFFFFFFFFu && (i == 0 || qid[i - 1] != q || kmer[i - 1] != kmer[i])) ? 1u : 0u;
}

__global__ void n2a_compact_kernel(const unsigned long long* __restrict__ kmer, const uint32_t* __restrict__ head,
                                   const uint32_t* __restrict__ hscan, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && head[i]) out[hscan[i]] = kmer[i];
}

// qoff[q] = number of unique k-mers of the queries before q = scanned heads at the first sorted position of query >= q
__global__ void n2a_query_offsets_kernel(const uint32_t* __restrict__ qid, const uint32_t* __restrict__ hscan, uint64_t n, uint32_t nq,
                                         uint64_t* __restrict__ qoff) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nq) return;
    uint64_t lo = 0, hi = n;                       // first position with qid >= q (dropped positions carry ~0: they sort last)
    while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (qid[mid] < q) lo = mid + 1; else hi = mid; }
    qoff[q] = hscan[lo];                           // hscan has n + 1 entries
}

}  // namespace

// device scratch of the sequence path is ~40 bytes per base and positions are 32-bit: larger batches are run in pieces
static uint64_t n2_seq_budget() {                        // bases per piece (one longer query still goes alone)
    if (const char* e = getenv("KMDB_N2A_BASES_PER_PIECE")) return std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    return 256ull << 20;
}

static int new2all_seq_once(kmdb_db* dbh, const char* const* seqs, const size_t* seq_lens, size_t nq, double fraction,
                            double start_fraction, int32_t alphabet, uint32_t* out_dense, uint64_t* out_kmer_counts,
                            const kmdb_opts* opts);

extern "C" int kmdb_new2all_batch_seq(kmdb_db* dbh, const char* const* seqs, const size_t* seq_lens, size_t nq, double fraction,
                                      double start_fraction, int preserve_strand, uint32_t* out_dense, uint64_t* out_kmer_counts,
                                      const kmdb_opts* opts) {
    return kmdb_new2all_batch_seq_alphabet(dbh, seqs, seq_lens, nq, fraction, start_fraction, preserve_strand ? KMDB_ALPHABET_NT_PRESERVE : KMDB_ALPHABET_NT,
                                           out_dense, out_kmer_counts, opts);
}

extern "C" int kmdb_new2all_batch_seq_alphabet(kmdb_db* dbh, const char* const* seqs, const size_t* seq_lens, size_t nq, double fraction,
                                               double start_fraction, int32_t alphabet, uint32_t* out_dense, uint64_t* out_kmer_counts,
                                               const kmdb_opts* opts) {
    if (!dbh || (nq && (!seqs || !seq_lens || !out_dense || !out_kmer_counts))) return kmdb_set_error("kmdb_new2all_batch_seq: null argument");
    if (alphabet < 0 || alphabet >= KMDB_ALPHABET_COUNT) return kmdb_set_error("kmdb_new2all_batch_seq: unknown alphabet " + std::to_string(alphabet));
    kmdb_engine_view e;
    if (kmdb_engine_get(dbh, &e)) return 1;
    // rows of different queries are independent: cut the batch where the accumulated bases pass the budget
    const uint64_t budget = n2_seq_budget();
    for (size_t q0 = 0; q0 < nq;) {
        size_t q1 = q0;
        uint64_t bases = 0;
        do { bases += seq_lens[q1]; ++q1; } while (q1 < nq && bases + seq_lens[q1] <= budget);
        if (new2all_seq_once(dbh, seqs + q0, seq_lens + q0, q1 - q0, fraction, start_fraction, alphabet, out_dense + q0 * e.N,
                             out_kmer_counts + q0, opts)) return 1;
        q0 = q1;
    }
    return 0;
}

static int new2all_seq_once(kmdb_db* dbh, const char* const* seqs, const size_t* seq_lens, size_t nq, double fraction,
                            double start_fraction, int32_t alphabet, uint32_t* out_dense, uint64_t* out_kmer_counts,
                            const kmdb_opts* opts) {
    kmdb_engine_view e;
    if (kmdb_engine_get(dbh, &e)) return 1;
    if (!e.n_buckets || !e.slots) return kmdb_set_error("kmdb_new2all_batch_seq: database was uploaded without hashtables");
    const uint32_t k = e.kmer_length;
    ExtractParams p{};
    int preserve_strand = 0;
    if (kmdbh_alphabet_table(alphabet, p.map, &p.size, &p.bits, &preserve_strand)) return kmdb_set_error("kmdb_new2all_batch_seq: unknown alphabet");
    if (k == 0 || k > 64u / p.bits - 1u) return kmdb_set_error("kmdb_new2all_batch_seq: k-mer length must be 1.." + std::to_string(64u / p.bits - 1u) + " for this alphabet (alphabet.h:37)");
    if (!nq) return 0;
    N2_TRY(hipSetDevice(e.device));
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : (hipStream_t)e.stream;
    std::vector<uint64_t> soff(nq + 1, 0);
    for (size_t q = 0; q < nq; ++q) soff[q + 1] = soff[q] + seq_lens[q];
    const uint64_t L = soff[nq];
    if (L >= (1ull << 31) - 2) return kmdb_set_error("kmdb_new2all_batch_seq: a single query of 2^31 bases or more; extract its k-mers on the host (kmdbh_extract_kmers) and use kmdb_new2all_batch");
    if (nq >= (1ull << 31)) return kmdb_set_error("kmdb_new2all_batch_seq: too many queries in one batch");
    DevBuf d_seq, d_soff, d_kmer, d_kmer2, d_qid, d_qid2, d_head, d_hscan, d_k, d_qoff, d_tmp;
    N2_TRY(d_seq.alloc(L + 32));
    N2_TRY(d_soff.alloc((nq + 1) * 8));
    N2_TRY(d_kmer.alloc(L * 8)); N2_TRY(d_kmer2.alloc(L * 8));
    N2_TRY(d_qid.alloc(L * 4)); N2_TRY(d_qid2.alloc(L * 4));
    N2_TRY(d_head.alloc((L + 1) * 4)); N2_TRY(d_hscan.alloc((L + 1) * 4));
    N2_TRY(d_qoff.alloc((nq + 1) * 8));
    for (size_t q = 0; q < nq; ++q)
        if (seq_lens[q]) N2_TRY(hipMemcpyAsync(d_seq.as<char>() + soff[q], seqs[q], seq_lens[q], hipMemcpyHostToDevice, st));
    N2_TRY(hipMemcpyAsync(d_soff.p, soff.data(), (nq + 1) * 8, hipMemcpyHostToDevice, st));
    std::vector<uint64_t> qoff(nq + 1, 0);
    size_t total = 0;
    if (L) {
        const int prefix_bits = (int)(p.bits * k) - 32;
        p.seq = d_seq.as<char>(); p.soff = d_soff.as<uint64_t>(); p.nq = (uint32_t)nq; p.total_len = L; p.k = k;
        p.widen = prefix_bits < 8 ? (uint32_t)(8 - prefix_bits) : 0u;
        p.preserve = preserve_strand; p.subsample = fraction < 1.0;
        const double u64max = (double)std::numeric_limits<uint64_t>::max();          // src/filter.h:38-51
        p.lo = (uint64_t)(u64max * start_fraction);
        p.hi = (uint64_t)(u64max * (start_fraction + fraction));
        p.quarter_len = (uint64_t)std::ceil((double)k / 4.0);
        p.kmer = d_kmer.as<unsigned long long>(); p.qid = d_qid.as<uint32_t>();
        const unsigned blocks = (unsigned)((L + 255) / 256);
        hipLaunchKernelGGL(n2a_extract_kernel, dim3(blocks), dim3(256), 0, st, p);
        N2_TRY(hipGetLastError());
        const int kbits = (int)std::min<uint32_t>(64u, p.bits * k + p.widen);
        int qbits = 1;
        while ((1ull << qbits) < nq) ++qbits;
        size_t tb1 = 0, tb2 = 0, tb3 = 0;
        N2_TRY(prim::sort_pairs(nullptr, tb1, d_kmer.as<unsigned long long>(), d_kmer2.as<unsigned long long>(),
                                                  d_qid.as<uint32_t>(), d_qid2.as<uint32_t>(), (int)L, 0, kbits, st));
        N2_TRY(prim::sort_pairs(nullptr, tb2, d_qid2.as<uint32_t>(), d_qid.as<uint32_t>(),
                                                  d_kmer2.as<unsigned long long>(), d_kmer.as<unsigned long long>(), (int)L, 0, 32, st));
        N2_TRY(prim::exclusive_sum(nullptr, tb3, d_head.as<uint32_t>(), d_hscan.as<uint32_t>(), (int)(L + 1), st));
        N2_TRY(d_tmp.alloc(std::max(tb1, std::max(tb2, tb3))));
        N2_TRY(prim::sort_pairs(d_tmp.p, tb1, d_kmer.as<unsigned long long>(), d_kmer2.as<unsigned long long>(),
                                                  d_qid.as<uint32_t>(), d_qid2.as<uint32_t>(), (int)L, 0, kbits, st));
        // dropped positions carry query ~0: all 32 bits take part so that they sort behind every query
        (void)qbits;
        N2_TRY(prim::sort_pairs(d_tmp.p, tb2, d_qid2.as<uint32_t>(), d_qid.as<uint32_t>(),
                                                  d_kmer2.as<unsigned long long>(), d_kmer.as<unsigned long long>(), (int)L, 0, 32, st));
        N2_TRY(hipMemsetAsync(d_head.p, 0, (L + 1) * 4, st));
        hipLaunchKernelGGL(n2a_heads_kernel, dim3(blocks), dim3(256), 0, st, d_kmer.as<unsigned long long>(), d_qid.as<uint32_t>(), L,
                           d_head.as<uint32_t>());
        N2_TRY(prim::exclusive_sum(d_tmp.p, tb3, d_head.as<uint32_t>(), d_hscan.as<uint32_t>(), (int)(L + 1), st));
        uint32_t n_unique = 0;
        N2_TRY(hipMemcpyAsync(&n_unique, d_hscan.as<uint32_t>() + L, 4, hipMemcpyDeviceToHost, st));
        hipLaunchKernelGGL(n2a_query_offsets_kernel, dim3((unsigned)((nq + 1 + 255) / 256)), dim3(256), 0, st, d_qid.as<uint32_t>(),
                           d_hscan.as<uint32_t>(), L, (uint32_t)nq, d_qoff.as<uint64_t>());
        N2_TRY(hipGetLastError());
        N2_TRY(hipStreamSynchronize(st));
        total = n_unique;
        N2_TRY(d_k.alloc(total * 8));
        hipLaunchKernelGGL(n2a_compact_kernel, dim3(blocks), dim3(256), 0, st, d_kmer.as<unsigned long long>(), d_head.as<uint32_t>(),
                           d_hscan.as<uint32_t>(), L, d_k.as<uint64_t>());
        N2_TRY(hipGetLastError());
        N2_TRY(hipMemcpyAsync(qoff.data(), d_qoff.p, (nq + 1) * 8, hipMemcpyDeviceToHost, st));
        N2_TRY(hipStreamSynchronize(st));
    } else {
        N2_TRY(d_k.alloc(16));
        N2_TRY(hipMemsetAsync(d_qoff.p, 0, (nq + 1) * 8, st));
    }
    for (size_t q = 0; q < nq; ++q) out_kmer_counts[q] = qoff[q + 1] - qoff[q];
    // the sort buffers are not needed any more: release them before the probe pipeline allocates its own
    d_kmer.reset(); d_kmer2.reset(); d_qid2.reset(); d_head.reset(); d_hscan.reset(); d_qid.reset();
    return n2a_run(dbh, e, st, d_k.as<uint64_t>(), d_qoff.as<uint64_t>(), total, nq, out_dense);
}

extern "C" int kmdb_new2all_batch_sparse(kmdb_db* dbh, const uint64_t* const* kmers, const size_t* counts, size_t nq,
                                         kmdb_sparse_rows* out, const kmdb_opts* opts) {
    if (!out) return kmdb_set_error("kmdb_new2all_batch_sparse: null argument");
    std::memset(out, 0, sizeof *out);
    kmdb_engine_view e;
    if (!dbh) return kmdb_set_error("kmdb_new2all_batch_sparse: null argument");
    if (kmdb_engine_get(dbh, &e)) return 1;
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
