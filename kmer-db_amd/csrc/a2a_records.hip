// a2a_records.hip — the block-record pipeline of the dense all2all path (default for N <= 2048).
#include "device_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace {
// ------------------------------------------------------------------------------------------
// Block records.  The N x N matrix is cut into blocks (X, Y), X >= Y, of `width` <= 64 consecutive
// sample ids (width chosen per database at upload).  A pattern's ascending id list meets a block in
// a contiguous run, so all pair updates of a pattern with weight w factor into BLOCK RECORDS
//        (X, Y, rowmask, colmask, w):   M[width*X + r][width*Y + c] += w   for r in rowmask, c in colmask
//                                                                            (c < r when X == Y).
// Flat form (all2all_sp semantics, reference similarity_calculator.cpp:596-638): every pattern with
// w > 0 adds its on-disk w to all pairs of its FULL list, i.e. one record per pair of non-empty
// 64-bit words X >= Y of the full list F = F(parent) | local ids.
//
// Per call, four kinds of kernels on one stream:
//   K0  b3_decode_kernel   gamma streams -> local (block, mask) pairs of every node            (thread per node)
//   K1n b3_narrow_kernel   nodes whose full list lies in ONE block (4 in 5): F is one register; DFS stream,
//                          64 nodes per wave step, records (X, X, F) straight into the diagonal buckets
//   K1w b3_emit_kernel     the other nodes, from their own DFS-ordered list; F in up to NBW registers per lane
//                          (compact: only the words the batch touches), record-parallel emission
//   K2  b2_apply_kernel    records -> matrix: 64 records per wave step as bit matrices, int8 MFMA accumulate
//                          (popcount passes per bit plane for the rare weights >= 128)
// Records go straight to their final, bucket-grouped position: the per-(slice, bucket, weight class) record
// counts are a pure function of the database and are tabulated once at upload with the count modes of the
// same kernels (like CSR row pointers), so there is no sort and no global atomic at run time.
// b2_emit_kernel is the sequential (stack replay) front half used when root paths exceed KMDB_CHAIN_MAX.
// ------------------------------------------------------------------------------------------
// block records, struct-of-arrays, one slot per record: rows always; cols only for off-diagonal
// buckets (on the diagonal cols == rows); w only for the classes with w > 1
struct B2Recs { unsigned long long* rows; ulonglong2* rc; uint32_t* w; };   // rows: diagonal buckets; rc = {rows, cols}: others
struct B2Item { uint32_t X, Y, cls, begin, end; };
// Records are grouped by weight class: class 0: w == 1 (no weight stored; popcounts, or unit bytes on the matrix
// cores), class 1: 2 <= w < 128 (the weight is one signed byte of an int8 MFMA operand), class 2: w >= 128 (rare:
// one popcount pass per bit plane).
constexpr uint32_t B2_NCLS = 3;
__host__ __device__ __forceinline__ uint32_t b2_weight_class(uint32_t w) { return w == 1u ? 0u : w < 128u ? 1u : 2u; }

constexpr int B2_WAVES = 4;

struct B2Params {
    A2AParams a;
    uint32_t maxn_pad;            // stack capacity (multiple of 64)
    uint32_t dec_cap;             // decoded ids per batch (>= maxn_pad)
    uint32_t nctr;                // B2_NCLS * number of buckets
    BlockMap bm;
    uint32_t* table;              // [n_segs][nctr]: count mode writes counts, emit mode reads record bases
    B2Recs rec;
    const uint32_t* w;            // on-disk weights, DFS order
};

__device__ __forceinline__ unsigned long long bcast64(unsigned long long v, uint32_t src) {
    return ((unsigned long long)bcast((uint32_t)(v >> 32), src) << 32) | bcast((uint32_t)v, src);
}
__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v, int d) {
    uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, WAVE);
    uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, WAVE);
    return ((unsigned long long)hi << 32) | lo;
}

struct B2Wave {
    unsigned long long* cum;     // [maxn_pad] running mask of same-block ids at positions <= p
    unsigned long long* dcum;    // [dec_cap]  per decoded id: running mask inside its own node
    uint32_t* ctr;               // [nctr]
    uint16_t* dec;               // [dec_cap]
    uint16_t* ent_start;         // [64] first stack position of the k-th distinct block
    uint8_t* ent_blk;            // [64]
    uint8_t* pblk;               // [maxn_pad] block id of position p
    uint32_t nb;                 // distinct blocks on the stack
};

__host__ __device__ inline size_t b2_wave_bytes(uint32_t maxn_pad, uint32_t dec_cap, uint32_t nctr) {
    size_t per_wave = (size_t)maxn_pad * 8 + (size_t)dec_cap * 8 + (size_t)nctr * 4 + (size_t)dec_cap * 2 + 64 * 2 + 64 + maxn_pad;
    return (per_wave + 15) & ~(size_t)15;
}

// extend the stack from `top` to `n` with the ids dec[off ..) / masks dcum[off ..); returns through refs
// what record emission needs.  All arguments wave-uniform.
__device__ __forceinline__ void b2_push(B2Wave& S, const BlockMap bm, uint32_t top, uint32_t n, uint32_t off, uint32_t lane,
                                        uint32_t& nbk, bool& first_is_head, unsigned long long& seed) {
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t prev_blk = 0xFFu;
    seed = 0;
    if (top > 0) { prev_blk = S.pblk[top - 1]; seed = S.cum[top - 1]; }
    prev_blk = __builtin_amdgcn_readfirstlane(prev_blk);
    const uint32_t my_start = lane < S.nb ? S.ent_start[lane] : 0xFFFFu;
    nbk = (uint32_t)__popcll(__ballot(my_start < top));
    if (n - top == 1) {
        // the common case (every internal trie node): one new id
        const uint32_t id = S.dec[off];
        const uint32_t blk = __builtin_amdgcn_readfirstlane(bm.blk(id));
        first_is_head = blk != prev_blk;
        const unsigned long long v = (1ull << bm.bit(id, blk)) | (first_is_head ? 0ull : seed);
        if (lane == 0) {
            S.cum[top] = v;
            S.pblk[top] = (uint8_t)blk;
            if (first_is_head) { S.ent_start[nbk] = (uint16_t)top; S.ent_blk[nbk] = (uint8_t)blk; }
        }
        S.nb = nbk + (first_is_head ? 1u : 0u);
        lds_sync();
        return;
    }
    uint32_t carry_blk = prev_blk;
    uint32_t newheads = 0;
    first_is_head = true;
    for (uint32_t p0 = top; p0 < n; p0 += WAVE) {
        const uint32_t p = p0 + lane;
        const bool act = p < n;
        const uint32_t id = act ? S.dec[off + (p - top)] : 0u;
        unsigned long long v = act ? S.dcum[off + (p - top)] : 0ull;
        const uint32_t blk = bm.blk(id);
        uint32_t pb = (uint32_t)__shfl_up((int)blk, 1, WAVE);
        if (lane == 0) pb = carry_blk;
        const bool head = act && (blk != pb);
        if (blk == prev_blk) v |= seed;            // ids ascend: only the first run can continue the parent's last block
        if (act) { S.cum[p] = v; S.pblk[p] = (uint8_t)blk; }
        const unsigned long long hb = __ballot(head);
        if (head) {
            const uint32_t e = nbk + newheads + (uint32_t)__popcll(hb & lt_mask);
            S.ent_start[e] = (uint16_t)p;
            S.ent_blk[e] = (uint8_t)blk;
        }
        if (p0 == top) first_is_head = (hb & 1ull) != 0;
        newheads += (uint32_t)__popcll(hb);
        const uint32_t lastl = (n - 1 - p0) < 63u ? (n - 1 - p0) : 63u;
        carry_blk = bcast(blk, lastl);
    }
    S.nb = nbk + newheads;
    lds_sync();
}

template <bool EMIT>
__global__ __launch_bounds__(WAVE * B2_WAVES) void b2_emit_kernel(B2Params q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const A2AParams& p = q.a;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg = p.seg_begin + blockIdx.x * B2_WAVES + wave;
    if (seg >= p.seg_end) return;
    // carve this wave's LDS
    unsigned char* base = lds_raw + b2_wave_bytes(q.maxn_pad, q.dec_cap, q.nctr) * wave;
    B2Wave S;
    S.cum = (unsigned long long*)base;
    S.dcum = S.cum + q.maxn_pad;
    S.ctr = (uint32_t*)(S.dcum + q.dec_cap);
    S.dec = (uint16_t*)(S.ctr + q.nctr);
    S.ent_start = S.dec + q.dec_cap;
    S.ent_blk = (uint8_t*)(S.ent_start + 64);
    S.pblk = S.ent_blk + 64;
    S.nb = 0;
    uint32_t* my_table = q.table + (size_t)seg * q.nctr;
    for (uint32_t k = lane; k < q.nctr; k += WAVE) S.ctr[k] = EMIT ? my_table[k] : 0u;

    const Segment sg = p.segs[seg];
    const uint32_t first = __builtin_amdgcn_readfirstlane(sg.first);
    const uint32_t end = __builtin_amdgcn_readfirstlane(sg.end);
    if (first >= end) return;
    uint32_t nbk;
    bool first_is_head;
    unsigned long long seed;
    {
        // ancestors of the first node: decode into dec[] at their stack positions, then push them
        int32_t cur = p.parent[first];
        const uint32_t depth = cur >= 0 ? p.meta[cur].x : 0u;
        while (cur >= 0) {
            const uint4 m = p.meta[cur];
            if (lane == 0) decode_node<uint16_t>(p.bits, p.bitpos[cur], m.y, m.z, S.dec + (m.x - m.y));
            cur = p.parent[cur];
        }
        lds_sync();
        if (lane == 0) {
            uint32_t curblk = 0xFFFFFFFFu;
            unsigned long long acc = 0;
            for (uint32_t i = 0; i < depth; ++i) {
                const uint32_t id = S.dec[i], blk = q.bm.blk(id);
                if (blk != curblk) { curblk = blk; acc = 0; }
                acc |= 1ull << q.bm.bit(id, blk);
                S.dcum[i] = acc;
            }
        }
        lds_sync();
        if (depth) b2_push(S, q.bm, 0, depth, 0, lane, nbk, first_is_head, seed);
        lds_sync();
    }

    const bool prof = (p.dbg & 32u) != 0;
    unsigned long long t_load = 0, t_dec = 0, t_push = 0, t_emit = 0, t0 = 0, t1 = 0;
    for (uint32_t base_i = first; base_i < end;) {
        if (prof) t0 = __builtin_amdgcn_s_memtime();
        const uint32_t i = base_i + lane;
        const bool valid = i < end;
        const uint4 m = valid ? p.meta[i] : make_uint4(0, 0, 0, 0);
        const uint64_t bp = valid ? p.bitpos[i] : 0;
        const uint32_t W = valid ? q.w[i] : 0u;              // on-disk num_kmers (flat form: no subtree sums)
        const uint32_t l = m.y;
        const uint32_t incl = wave_incl_scan(l, lane);
        const unsigned long long fit = __ballot(valid && incl <= q.dec_cap);
        uint32_t cnt = fit == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fit);
        cnt = __builtin_amdgcn_readfirstlane(cnt);      // >= 1: dec_cap >= the longest list
        const uint32_t off = incl - l;
        if (prof) { t1 = __builtin_amdgcn_s_memtime(); t_load += t1 - t0; t0 = t1; }
        if (lane < cnt) decode_node_cum(p.bits, bp, l, m.z, S.dec + off, S.dcum + off, q.bm);
        lds_sync();
        if (prof) { t1 = __builtin_amdgcn_s_memtime(); t_dec += t1 - t0; }
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t nj = bcast(m.x, j), lj = bcast(m.y, j), oj = bcast(off, j), Wj = bcast(W, j);
            const uint32_t top = nj - lj;
            if (lj == 0) continue;
            if (prof) t0 = __builtin_amdgcn_s_memtime();
            b2_push(S, q.bm, top, nj, oj, lane, nbk, first_is_head, seed);
            if (prof) { t1 = __builtin_amdgcn_s_memtime(); t_push += t1 - t0; t0 = t1; }
            if (Wj == 0 || nj < 2) continue;
            // ---- FLAT form (all2all_sp semantics, reference similarity_calculator.cpp:596-638): a pattern with
            // its own on-disk weight w adds w to every pair of its full list.  One record per pair of
            // distinct blocks (kX >= kY) of the list; all pairs of a node go to distinct buckets, so a lane
            // per pair needs no coordination.  Patterns with w == 0 (inner trie nodes) emit nothing.
            const uint32_t nb = S.nb;
            const uint32_t total = nb * (nb + 1) / 2;
            const uint32_t cls = b2_weight_class(Wj);
            for (uint32_t t = lane; t < total; t += WAVE) {
                uint32_t kX = (uint32_t)((__fsqrt_rn(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while (kX * (kX + 1) / 2 > t) --kX;
                while ((kX + 1) * (kX + 2) / 2 <= t) ++kX;
                const uint32_t kY = t - kX * (kX + 1) / 2;
                const uint32_t endX = (kX + 1 < nb) ? S.ent_start[kX + 1] : nj;
                const uint32_t endY = (kY + 1 < nb) ? S.ent_start[kY + 1] : nj;
                const uint32_t bX = S.ent_blk[kX], bY = S.ent_blk[kY];
                const unsigned long long rows = S.cum[endX - 1];
                const unsigned long long cols = S.cum[endY - 1];
                const uint32_t b = (bX * (bX + 1) / 2 + bY) * B2_NCLS + cls;
                const uint32_t slot = atomicAdd(&S.ctr[b], 1u);
                if (EMIT) {
                    if (bX != bY) q.rec.rc[slot] = make_ulonglong2(rows, cols); else q.rec.rows[slot] = rows;
                    if (cls) q.rec.w[slot] = Wj;
                }
            }
            lds_sync();
            if (prof) { t1 = __builtin_amdgcn_s_memtime(); t_emit += t1 - t0; }
        }
        lds_sync();
        base_i += cnt;
    }
    if (prof && lane == 0) {
        atomicAdd(&p.counters[1], t_load); atomicAdd(&p.counters[2], t_dec);
        atomicAdd(&p.counters[3], t_push); atomicAdd(&p.counters[4], t_emit);
    }
    if (!EMIT) {
        lds_sync();
        for (uint32_t k = lane; k < q.nctr; k += WAVE) my_table[k] = S.ctr[k];
    }
}

// ------------------------------------------------------------------------------------------
// v3 front half of the block-record pipeline: no per-node sequential work.
//   K0 b3_decode_kernel  — one THREAD per node, nodes visited in order of decreasing local-list
//        length so that the lanes of a wave decode streams of similar length.  The gamma stream is
//        turned into the node's LOCAL block masks: pairs (block, 64-bit mask), ids ascending so the
//        pairs come out in block order.  This is the only place gamma codes are read.
//   K1 b3_emit_kernel — one WAVE per DFS segment, 64 consecutive nodes per step, one lane each.
//        A node's full list is F_j = F_parent(j) | L_j over NBW 64-bit words held in registers.
//        Parents inside the batch are resolved with pointer doubling across lanes (log2 rounds of
//        cross-lane reads); parents before the batch come from the "chain table" in LDS: the full
//        masks of every node on the root path of the last node of the previous batch (DFS order
//        guarantees every earlier parent is on that path).  Records (flat form, one per pair of
//        non-empty words X >= Y) are written with ballot-ranked, coalesced stores.
// ------------------------------------------------------------------------------------------
constexpr int B3_WAVES = 1;
constexpr int B3_K = 8;            // word registers of the compact path
constexpr uint32_t B3_QCAP = 512;  // record descriptors queued per round
constexpr int B3_CHAIN = KMDB_CHAIN_MAX;

template <bool COUNT, bool LONG>
__global__ __launch_bounds__(256) void b3_decode_kernel(const uint2* __restrict__ k0in, const uint32_t* __restrict__ bitrel,
                                                        const uint64_t* __restrict__ blkbase,
                                                        const uint64_t* __restrict__ bits, const uint32_t* __restrict__ perm,
                                                        uint32_t P, uint32_t short_max, BlockMap bm, unsigned long long* __restrict__ p0_mask, uint16_t* __restrict__ p0_info,
                                                        uint32_t* __restrict__ pair_ofs, uint8_t* __restrict__ pair_blk,
                                                        unsigned long long* __restrict__ pair_mask, uint32_t kdbg) {
    // output per node: the first (block, mask) pair inline — p0_info = block | npairs << 8 — and any
    // further pairs in a CSR side array (pair_ofs counts only the extra pairs)
    // two launches cover the nodes: perm == nullptr walks ALL nodes in DFS order (coalesced) and skips the
    // ones with more than short_max local ids; those few are listed in perm, longest first, and decoded
    // by the second launch so that no wave waits on one long stream
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i;
    if (perm) {
        if (t >= P) return;
        i = perm[t];
    } else {
        // DFS-order launch: the 256 nodes of the block are re-dealt to its threads by decreasing amount of work
        // (counting sort in LDS), so every wave runs decode loops of similar length while all global accesses
        // of the block stay inside its own 256-node window
        __shared__ uint32_t bins[64];
        __shared__ uint16_t order[256];
        if (threadIdx.x < 64) bins[threadIdx.x] = 0;
        __syncthreads();
        const uint2 kt = t < P ? k0in[t] : make_uint2(0u, 0u);          // {l | last id << 16, stream bits}
        const uint2 lt2 = make_uint2(kt.x & 0xFFFFu, kt.y);
        // work of a node ~ number of codes that are not "0" ~ stream bits beyond one per delta
        uint32_t key = 0;                                                // 0: nothing to decode here
        if (t < P && lt2.x > 1 && !kmdb_long_node(lt2.x, lt2.y)) {
            key = 1u + (lt2.y - (lt2.x - 1u));
            key = key > 63u ? 63u : key;
        }
        atomicAdd(&bins[63u - key], 1u);                                 // bin 0 = most work
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t c = bins[threadIdx.x];
            bins[threadIdx.x] = wave_incl_scan(c, threadIdx.x) - c;
        }
        __syncthreads();
        order[atomicAdd(&bins[63u - key], 1u)] = (uint16_t)threadIdx.x;
        __syncthreads();
        i = blockIdx.x * blockDim.x + ((kdbg & 1u) ? threadIdx.x : order[threadIdx.x]);
        if (i >= P) return;
    }
    const uint2 km = k0in[i];
    const uint4 m = make_uint4(0u, km.x & 0xFFFFu, km.x >> 16, km.y);   // {-, l, last id, stream bits}
    const uint32_t l = m.y;
    if (!perm && kmdb_long_node(l, m.w)) return;
    uint32_t npairs = 0, blk0 = 0;
    unsigned long long mask0 = 0;
    if (l == 1 || (kdbg & 4u)) {
        blk0 = bm.blk(m.z); mask0 = 1ull << bm.bit(m.z, blk0); npairs = 1;
    } else if (l) {
        // Pass 1 walks the stream run by run and builds the list RELATIVE to its (still unknown) first id:
        // bit k of R <=> id_0 + k is in the list.  pattern_t::decodeSamples (reference src/pattern.cpp:99-109)
        // gets id_0 the same way: last id minus the sum of the deltas.
        using Cursor = RunCursor<LONG ? 8 : 3, LONG>;
        const uint64_t pos = blkbase[i >> 8] + bitrel[i];
        unsigned long long R = 1ull;
        uint32_t span = 0;
        {
            Cursor c(bits, pos);
            uint32_t rem = l - 1;
            while (rem) {
                uint32_t z, v;
                c.step(rem, z, v);                                     // a run of z consecutive ids, then a gap of v
                if (z) {
                    if (span + z < 64u) R |= ((2ull << (z - 1)) - 1ull) << (span + 1);
                    span += z; rem -= z;
                }
                if (v) {
                    span += v; --rem;
                    if (span < 64u) R |= 1ull << span;
                }
            }
        }
        const uint32_t id0 = m.z - span;
        blk0 = bm.blk(id0);
        const uint32_t bit0 = bm.bit(id0, blk0);
        const unsigned long long wm = bm.width == 64 ? ~0ull : (1ull << bm.width) - 1ull;
        if (span < 64u || (kdbg & 16u)) {
            // the whole list fits the relative mask: cut it at the block boundaries (at most 3 blocks: width >= 32)
            const unsigned long long lo = R << bit0, hi = bit0 ? R >> (64u - bit0) : 0ull;
            auto ext = [&](uint32_t sh) -> unsigned long long {
                return sh == 0 ? lo : sh < 64u ? ((lo >> sh) | (hi << (64u - sh))) : sh == 64u ? hi : sh < 128u ? (hi >> (sh - 64u)) : 0ull;
            };
            mask0 = lo & wm;
            const unsigned long long m1 = ext(bm.width) & wm, m2 = ext(2 * bm.width) & wm;
            npairs = 1u + (m1 != 0) + (m2 != 0);
            if (!COUNT && npairs > 1) {
                uint32_t out = pair_ofs[i];
                if (m1) { pair_blk[out] = (uint8_t)(blk0 + 1); pair_mask[out] = m1; ++out; }
                if (m2) { pair_blk[out] = (uint8_t)(blk0 + 2); pair_mask[out] = m2; }
            }
        } else {
            // wide list: second pass with absolute ids, blocks come out in ascending order
            Cursor c(bits, pos);
            uint32_t out = 0;
            if (!COUNT) out = pair_ofs[i];
            uint32_t curblk = blk0, bit = bit0, rem = l - 1;
            unsigned long long acc = 1ull << bit0;
            auto flush = [&]() {
                if (npairs == 0) mask0 = acc;
                else { if (!COUNT) { pair_blk[out] = (uint8_t)curblk; pair_mask[out] = acc; } ++out; }
                ++npairs;
            };
            while (rem) {
                uint32_t z, v;
                c.step(rem, z, v);
                if (z) {
                    rem -= z;
                    while (z) {
                        const uint32_t room = bm.width - 1u - bit;
                        const uint32_t t = z < room ? z : room;
                        if (t) { acc |= ((2ull << (t - 1)) - 1ull) << (bit + 1); bit += t; z -= t; }
                        if (z) { flush(); ++curblk; acc = 1ull; bit = 0; --z; }
                    }
                }
                if (v) {
                    const uint32_t id = curblk * bm.width + bit + v;
                    --rem;
                    const uint32_t blk = bm.blk(id);
                    if (blk != curblk) { flush(); curblk = blk; acc = 0; }
                    bit = bm.bit(id, blk);
                    acc |= 1ull << bit;
                }
            }
            flush();
        }
    }
    if (COUNT) pair_ofs[i] = npairs ? npairs - 1 : 0;
    else if (!(kdbg & 2u) || mask0 == 0x123456789ull) { p0_mask[i] = mask0; p0_info[i] = (uint16_t)(blk0 | (npairs << 8)); }
}

struct B3Params {
    const uint32_t* nl;            // n | l << 16 per node
    const int32_t* parent;
    const uint32_t* w;
    const Segment* segs;
    const uint32_t* seg_anc;       // [n_segs][B3_CHAIN] root-first ancestors of the segment's first node
    const uint32_t* seg_anc_n;     // [n_segs]
    const unsigned long long* p0_mask;
    const uint16_t* p0_info;
    const uint32_t* pair_ofs;
    const uint8_t* pair_blk;
    const unsigned long long* pair_mask;
    uint32_t n_segs;
    uint32_t maxn_pad;
    uint32_t nctr;
    uint32_t chain_cap;            // slots of the chain table (longest root path of the database, rounded up)
    uint32_t* table;
    B2Recs rec;
    uint32_t dbg;
    unsigned long long* counters;
    // wide-list launch (INDIRECT): lane k of a batch handles node widx[base + k]; parents are positions in the list
    const uint32_t* widx;
    const int32_t* wparent;
    const ulonglong2* fnarrow;     // (F0, F1) of the <= 2-block nodes that have a wider child
    const uint16_t* wd01;          // their blocks: first | second << 8 (0xFF: none)
    const int32_t* seg_np;
    uint32_t seg_row0;             // first row of this launch in the count / base table
    uint32_t* fwords;              // count mode of the all-nodes launch: which words (blocks) every node's full list touches
};

__host__ __device__ inline size_t b3_wave_bytes(uint32_t nbw, uint32_t maxn_pad, uint32_t nctr, uint32_t chain_cap) {
    size_t b = (size_t)chain_cap * nbw * 8 + (size_t)nctr * 4 + (size_t)chain_cap * 4 * 2 + (maxn_pad + 64);
    b = (b + 7) & ~(size_t)7;
    b += (size_t)B3_K * 64 * 8 + 3 * 64 * 4 + 64 * 4;       // record staging: masks [B3_K][64], per-lane start / slots / weight, slot -> word
    b += (size_t)B3_K * 64 + B3_QCAP * 2;                   // word of every staged mask, record queue
    return (b + 15) & ~(size_t)15;
}

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, WAVE);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, WAVE);
    return ((unsigned long long)hi << 32) | lo;
}

// per-wave context of the emit kernel (LDS pointers + lane constants)
struct B3Ctx {
    unsigned long long* chain;     // [B3_CHAIN][NBW] full masks of the nodes on the current root path
    uint32_t* chain_nz;            // [B3_CHAIN] which words of the slot are meaningful (others may be stale)
    uint32_t* chain_n;             // [B3_CHAIN] list length of the node in the slot
    uint8_t* slot_of_n;            // list length -> slot
    uint32_t* ctr;                 // record cursors per (bucket, class)
    unsigned long long* fmat;      // record staging: [slot][lane] full masks of the batch
    uint32_t* st_excl;             // [64] first record of the lane within the batch
    uint32_t* st_nzs;              // [64] non-empty slots of the lane
    uint32_t* st_w;                // [64] weight
    uint32_t* st_word;             // [64] slot -> word
    uint8_t* wsp;                  // [B3_K][64] word of the staged masks (sparse staging)
    uint16_t* queue;               // [B3_QCAP] record descriptors: owner lane | a << 6 | b << 9
    uint32_t lane;
    unsigned long long lt_mask;
};

struct B3Lane {                    // one node per lane
    bool valid;
    uint32_t n, l, w, info, idx;
    int32_t par;
    unsigned long long m0;
    uint32_t npw, npw1;            // wide-list launch: blocks and full masks of a parent with <= 2 blocks (mask 0: none)
    unsigned long long npm, npm1;
    uint32_t po, e1blk;            // first entry of the node's extra pairs and the first of them (fetched ahead)
    unsigned long long e1mask;
};

// One batch of 64 consecutive DFS nodes.  Every lane keeps the full-list masks of its node in W 64-bit
// registers.  W == NBW with IDENT: register s is word s.  Otherwise the batch only touches ku <= W distinct
// words (the usual case: one or two clades) and register s holds word wl[s] — the work then does not grow
// with the number of blocks of the matrix.
template <int W, bool IDENT, int NBW, bool EMIT>
__device__ __forceinline__ void b3_batch(const B3Params& q, const B3Ctx& C, const B3Lane& L, uint32_t base, uint32_t end,
                                         uint32_t U, const uint32_t (&wl)[W], uint32_t ku, uint32_t rootslot, uint32_t inh,
                                         unsigned long long (&tph)[4]) {
    const uint32_t lane = C.lane;
    const bool prof = (q.dbg & 32u) != 0;
    unsigned long long tq = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    auto mark = [&](int k) { if (prof) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tph[k] += t - tq; tq = t; } };
    auto slot_of_word = [&](uint32_t wd) -> uint32_t { return IDENT ? wd : (uint32_t)__popc(U & ((1u << wd) - 1u)); };
    unsigned long long F[W];
    const uint32_t b0 = L.info & 0xFFu, np = L.info >> 8;
    {
        const uint32_t s0 = slot_of_word(b0);
#pragma unroll
        for (int s = 0; s < W; ++s) F[s] = (np != 0 && s0 == (uint32_t)s) ? L.m0 : 0ull;
        if (__ballot(np > 1)) {
            // the first extra pair came with the node record; further ones (rare) are fetched here
            {
                const uint32_t sb = slot_of_word(L.e1blk);
#pragma unroll
                for (int s = 0; s < W; ++s) F[s] |= (np > 1 && sb == (uint32_t)s) ? L.e1mask : 0ull;
            }
            if (__ballot(np > 2)) {
                const uint32_t po = L.po;
                uint32_t mx = np > 1 ? np - 1 : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, d, WAVE); mx = o > mx ? o : mx; }
                mx = __builtin_amdgcn_readfirstlane(mx);
                for (uint32_t t = 1; t < mx; ++t) {
                    if (t + 1 < np) {
                        const uint32_t sb = slot_of_word(q.pair_blk[po + t]);
                        const unsigned long long mk = q.pair_mask[po + t];
#pragma unroll
                        for (int s = 0; s < W; ++s) F[s] |= (sb == (uint32_t)s) ? mk : 0ull;
                    }
                }
            }
        }
    }
    if (__ballot(L.npm != 0)) {
        const uint32_t sp = slot_of_word(L.npw), sp1 = slot_of_word(L.npw1 & 31u);
#pragma unroll
        for (int s = 0; s < W; ++s)
            F[s] |= ((L.npm != 0 && sp == (uint32_t)s) ? L.npm : 0ull) | ((L.npm1 != 0 && sp1 == (uint32_t)s) ? L.npm1 : 0ull);
    }
    const bool inb = L.valid && L.par >= (int32_t)base;
    int pl = inb ? (int)(L.par - (int32_t)base) : -1;
    const int plo = pl;
    if (rootslot != 0xFFFFFFFFu) {
        // parent precedes the batch: it is on the root path of the previous batch's last node
#pragma unroll
        for (int s = 0; s < W; ++s) {
            if (!IDENT && (uint32_t)s >= ku) break;
            const uint32_t wd = IDENT ? (uint32_t)s : wl[s];
            if ((inh >> wd) & 1u) F[s] |= C.chain[(size_t)rootslot * NBW + wd];
        }
    }
    mark(0);
    // ---- pointer doubling over in-batch parents
    while (__ballot(pl >= 0)) {
        const int src = pl >= 0 ? pl : (int)lane;
#pragma unroll
        for (int s = 0; s < W; ++s) {
            if (!IDENT && (uint32_t)s >= ku) break;
            const unsigned long long o = shfl64(F[s], src);
            if (pl >= 0) F[s] |= o;
        }
        const int npl = __shfl(pl, src, WAVE);
        pl = pl >= 0 ? npl : -1;
    }
    mark(1);
    if (!EMIT && q.fwords != nullptr && L.valid) {
        uint32_t fw = 0;
#pragma unroll
        for (int s = 0; s < W; ++s) fw |= F[s] != 0 ? 1u << (IDENT ? (uint32_t)s : wl[s]) : 0u;
        q.fwords[L.idx] = fw;
    }
    // ---- records: flat form, one per pair of non-empty words X >= Y of patterns with w > 0
    const bool act = L.valid && L.w != 0 && L.n >= 2 && !(q.dbg & 512u);
    // Record-parallel emission: a node with p non-empty words owns p(p+1)/2 records (word pairs a >= b).  The
    // batch's records are numbered by a prefix sum over the lanes.  Every owner stages its non-empty masks in LDS
    // ([k-th non-empty word][lane]) and pushes one 16-bit descriptor (lane, a, b) per record into a queue; then the
    // wave takes 64 descriptors at a time, one per lane: read the two masks and their words, reserve the slot in
    // the (X, Y, class) stream with one LDS atomic, store.  No search, three LDS round trips per 64 records.
    {
        uint32_t nzs = 0;
#pragma unroll
        for (int s = 0; s < W; ++s) nzs |= (F[s] != 0 ? 1u : 0u) << s;
        const uint32_t pw = (uint32_t)__popc(nzs);
        const uint32_t myrec = act ? pw * (pw + 1u) / 2u : 0u;
        const uint32_t incl = wave_incl_scan(myrec, lane);
        const uint32_t T = bcast(incl, WAVE - 1);
        if (T == 0) goto chain_update;
        bool fast = true;
        if (IDENT) fast = !__ballot(act && pw > (uint32_t)B3_K);         // more words than staging rows: the general path below
        if (fast) {
            uint32_t r = 0;
#pragma unroll
            for (int s = 0; s < W; ++s) {
                if (!IDENT && (uint32_t)s >= ku) break;
                if (act && F[s] != 0) { C.fmat[r * 64 + lane] = F[s]; C.wsp[r * 64 + lane] = (uint8_t)(IDENT ? (uint32_t)s : wl[s]); ++r; }
            }
            C.st_w[lane] = L.w;
            const uint32_t excl = incl - myrec;
            for (uint32_t q0 = 0; q0 < T; q0 += B3_QCAP) {
                if (myrec) {
                    uint32_t a = 0, b = 0;
                    for (uint32_t i = excl; i < incl; ++i) {
                        if (i - q0 < B3_QCAP) C.queue[i - q0] = (uint16_t)(lane | a << 6 | b << 9);
                        if (++b > a) { ++a; b = 0; }
                    }
                }
                lds_sync();
                const uint32_t tend = T < q0 + B3_QCAP ? T : q0 + B3_QCAP;
                for (uint32_t t0 = q0; t0 < tend; t0 += WAVE) {
                    const uint32_t t = t0 + lane;
                    const bool on = t < tend;
                    const uint32_t d = on ? C.queue[t - q0] : 0u;
                    const uint32_t own = d & 63u, a = (d >> 6) & 7u, b = d >> 9;
                    const unsigned long long FX = C.fmat[a * 64 + own], FY = C.fmat[b * 64 + own];
                    const uint32_t X = C.wsp[a * 64 + own], Y = C.wsp[b * 64 + own];
                    const uint32_t wv = C.st_w[own];
                    const uint32_t cls = b2_weight_class(wv);
                    if (on) {
                        const uint32_t slot = atomicAdd(&C.ctr[(X * (X + 1u) / 2u + Y) * B2_NCLS + cls], 1u);
                        if (EMIT && !(q.dbg & 256u)) {
                            if (X != Y) q.rec.rc[slot] = make_ulonglong2(FX, FY); else q.rec.rows[slot] = FX;
                            if (cls) q.rec.w[slot] = wv;
                        }
                    }
                }
                lds_sync();
            }
            goto chain_update;
        }
        // general path (a lane with more than B3_K words): owner by binary search in the prefix sums, pair from the
        // index within the owner, masks fetched from the owner's registers
        if (!IDENT) {
#pragma unroll
            for (int s = 0; s < W; ++s) {
                if ((uint32_t)s >= ku) break;
                C.fmat[s * 64 + lane] = F[s];
            }
        }
        C.st_excl[lane] = incl - myrec; C.st_nzs[lane] = nzs; C.st_w[lane] = L.w;
        if (!IDENT) {
            uint32_t mw = 0;
#pragma unroll
            for (int s = 0; s < W; ++s) mw = lane == (uint32_t)s ? wl[s] : mw;
            C.st_word[lane] = mw;
        }
        lds_sync();
        for (uint32_t t0 = 0; t0 < T; t0 += WAVE) {
            const uint32_t t = t0 + lane;
            const bool on = t < T;
            uint32_t own = 0;
#pragma unroll
            for (uint32_t step = 32; step >= 1; step >>= 1) {
                const uint32_t cand = own + step;
                if (C.st_excl[cand & 63u] <= t) own = cand;           // cand <= 63 always: own < 64 - step
            }
            const uint32_t qi = on ? t - C.st_excl[own] : 0u;
            uint32_t a = (uint32_t)((__fsqrt_rn((float)(8u * qi + 1u)) - 1.0f) * 0.5f);
            a = a * (a + 1u) / 2u > qi ? a - 1u : a;
            a = (a + 1u) * (a + 2u) / 2u <= qi ? a + 1u : a;
            const uint32_t b = qi - a * (a + 1u) / 2u;
            uint32_t ma = C.st_nzs[own], mb = ma;
            for (uint32_t k = 0; k < a; ++k) ma &= ma - 1u;
            for (uint32_t k = 0; k < b; ++k) mb &= mb - 1u;
            const uint32_t sa = on ? (uint32_t)__builtin_ctz(ma | 0x80000000u) : 0u, sb = on ? (uint32_t)__builtin_ctz(mb | 0x80000000u) : 0u;
            unsigned long long FX = 0, FY = 0;
            if (IDENT) {
                // full-width batches (rare) keep the masks in registers: fetch every word from the owner, keep two
#pragma unroll
                for (int s = 0; s < W; ++s) {
                    const unsigned long long v = shfl64(F[s], (int)own);
                    FX = sa == (uint32_t)s ? v : FX;
                    FY = sb == (uint32_t)s ? v : FY;
                }
            } else { FX = C.fmat[sa * 64 + own]; FY = C.fmat[sb * 64 + own]; }
            const uint32_t X = IDENT ? sa : C.st_word[sa], Y = IDENT ? sb : C.st_word[sb];
            const uint32_t wv = C.st_w[own];
            const uint32_t cls = b2_weight_class(wv);
            if (on) {
                const uint32_t slot = atomicAdd(&C.ctr[(X * (X + 1u) / 2u + Y) * B2_NCLS + cls], 1u);
                if (EMIT && !(q.dbg & 256u)) {
                    if (X != Y) q.rec.rc[slot] = make_ulonglong2(FX, FY); else q.rec.rows[slot] = FX;
                    if (cls) q.rec.w[slot] = wv;
                }
            }
        }
        lds_sync();
    }
chain_update:
    mark(2);
    // ---- chain table for the next batch: root path of this batch's last node
    if (base + WAVE < end) {
        const uint32_t nvalid = (end - base) < (uint32_t)WAVE ? (end - base) : (uint32_t)WAVE;
        unsigned long long anc = 0;
        int cur = (int)nvalid - 1;
        while (cur >= 0) { anc |= 1ull << cur; cur = __builtin_amdgcn_readlane(plo, cur); }
        const uint32_t r = (uint32_t)__builtin_ctzll(anc);                 // in-batch root of that path
        const int32_t rpar = __builtin_amdgcn_readlane(L.par, (int)r);
        const uint32_t rtop = bcast(L.n - L.l, r);
        const uint32_t kept = rpar >= 0 ? (uint32_t)C.slot_of_n[rtop] + 1u : 0u;
        lds_sync();
        if ((anc >> lane) & 1ull) {
            const uint32_t slot = kept + (uint32_t)__popcll(anc & C.lt_mask);
            uint32_t nzw = 0;
#pragma unroll
            for (int s = 0; s < W; ++s) {
                if (!IDENT && (uint32_t)s >= ku) break;
                const uint32_t wd = IDENT ? (uint32_t)s : wl[s];
                if (F[s] != 0) { C.chain[(size_t)slot * NBW + wd] = F[s]; nzw |= 1u << wd; }
            }
            C.chain_nz[slot] = nzw;
            C.chain_n[slot] = L.n;
            C.slot_of_n[L.n] = (uint8_t)slot;
        }
        lds_sync();
    }    mark(3);
}


template <int NBW, bool EMIT, bool INDIRECT>
__global__ __launch_bounds__(WAVE * B3_WAVES) void b3_emit_kernel(B3Params q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg = blockIdx.x * B3_WAVES + wave;
    if (seg >= q.n_segs) return;
    unsigned char* basep = lds_raw + b3_wave_bytes(NBW, q.maxn_pad, q.nctr, q.chain_cap) * wave;
    B3Ctx C;
    C.chain = (unsigned long long*)basep;                                  // [B3_CHAIN][NBW]
    C.ctr = (uint32_t*)(C.chain + (size_t)q.chain_cap * NBW);              // [nctr]
    C.chain_n = C.ctr + q.nctr;                                            // [chain_cap]
    C.chain_nz = C.chain_n + q.chain_cap;                                  // [chain_cap]
    C.slot_of_n = (uint8_t*)(C.chain_nz + q.chain_cap);                    // [maxn_pad + 64]
    C.fmat = (unsigned long long*)(((uintptr_t)(C.slot_of_n + q.maxn_pad + 64) + 7) & ~(uintptr_t)7);   // [B3_K][64]
    C.st_excl = (uint32_t*)(C.fmat + (size_t)B3_K * 64);
    C.st_nzs = C.st_excl + 64; C.st_w = C.st_nzs + 64; C.st_word = C.st_w + 64;
    C.wsp = (uint8_t*)(C.st_word + 64); C.queue = (uint16_t*)(C.wsp + B3_K * 64);
    C.lane = lane;
    C.lt_mask = (1ull << lane) - 1ull;
    uint32_t* my_table = q.table + (size_t)(q.seg_row0 + seg) * q.nctr;
    for (uint32_t k = lane; k < q.nctr; k += WAVE) C.ctr[k] = EMIT ? my_table[k] : 0u;

    const Segment sg = q.segs[seg];
    const uint32_t first = __builtin_amdgcn_readfirstlane(sg.first);
    const uint32_t end = __builtin_amdgcn_readfirstlane(sg.end);
    if (first >= end) return;

    // ---- chain table for the first node: the ancestors' full masks = inclusive OR-scan along the root path,
    // 64 ancestors per round with the last lane's masks carried into the next round
    {
        const uint32_t d = q.seg_anc_n[seg];
        unsigned long long carry[NBW];
#pragma unroll
        for (int w = 0; w < NBW; ++w) carry[w] = 0;
        for (uint32_t c0 = 0; c0 < d; c0 += WAVE) {
            unsigned long long F[NBW];
            const uint32_t k = c0 + lane;
            const bool on = k < d;
            const uint32_t node = on ? q.seg_anc[(size_t)seg * q.chain_cap + k] : 0u;
            const uint32_t info = on ? q.p0_info[node] : 0u;
            const unsigned long long m0 = on ? q.p0_mask[node] : 0ull;
            const uint32_t b0 = info & 0xFFu, np = info >> 8;
#pragma unroll
            for (int w = 0; w < NBW; ++w) F[w] = (np != 0 && b0 == (uint32_t)w) ? m0 : 0ull;
            if (__ballot(np > 1)) {
                const uint32_t po = np > 1 ? q.pair_ofs[node] : 0u;
                uint32_t mx = np > 1 ? np - 1 : 0u;
#pragma unroll
                for (int dd = 32; dd >= 1; dd >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, dd, WAVE); mx = o > mx ? o : mx; }
                mx = __builtin_amdgcn_readfirstlane(mx);
                for (uint32_t t = 0; t < mx; ++t) {
                    if (t + 1 < np) {
                        const uint32_t b = q.pair_blk[po + t];
                        const unsigned long long mk = q.pair_mask[po + t];
#pragma unroll
                        for (int w = 0; w < NBW; ++w) F[w] |= (b == (uint32_t)w) ? mk : 0ull;
                    }
                }
            }
            if (INDIRECT && c0 == 0 && lane == 0) {
                const int32_t np0 = q.seg_np[seg];
                if (np0 >= 0) {
                    const uint32_t wd = q.wd01[np0];
                    const ulonglong2 mk = q.fnarrow[np0];
#pragma unroll
                    for (int w = 0; w < NBW; ++w)
                        F[w] |= (((wd & 0xFFu) == (uint32_t)w) ? mk.x : 0ull) | (((wd >> 8) == (uint32_t)w) ? mk.y : 0ull);
                }
            }
#pragma unroll
            for (int w = 0; w < NBW; ++w) if (lane == 0) F[w] |= carry[w];
#pragma unroll
            for (int s = 1; s < WAVE; s <<= 1) {
#pragma unroll
                for (int w = 0; w < NBW; ++w) {
                    const unsigned long long o = shfl_up64(F[w], s);
                    if (lane >= (uint32_t)s) F[w] |= o;
                }
            }
            if (on) {
                const uint32_t nn = q.nl[node] & 0xFFFFu;
                uint32_t nzw = 0;
#pragma unroll
                for (int w = 0; w < NBW; ++w) { C.chain[(size_t)k * NBW + w] = F[w]; nzw |= (F[w] != 0 ? 1u : 0u) << w; }
                C.chain_nz[k] = nzw;
                C.chain_n[k] = nn;
                C.slot_of_n[nn] = (uint8_t)k;
            }
#pragma unroll
            for (int w = 0; w < NBW; ++w) carry[w] = shfl64(F[w], WAVE - 1);
        }
        lds_sync();
    }

    // node records of the NEXT batch are fetched while the current one is processed
    uint32_t nx_nl = 0, nx_w = 0, nx_info = 0, nx_idx = 0, nx_npw = 0, nx_npw1 = 0, nx_po = 0, nx_e1blk = 0;
    int32_t nx_par = -1;
    unsigned long long nx_m0 = 0, nx_npm = 0, nx_npm1 = 0, nx_e1mask = 0;
    auto fetch = [&](uint32_t b0) {
        const uint32_t k = b0 + lane;
        const bool v = k < end;
        const uint32_t ii = INDIRECT ? (v ? q.widx[k] : 0u) : k;
        nx_idx = ii;
        nx_nl = v ? q.nl[ii] : 0u;
        nx_w = v ? q.w[ii] : 0u;
        nx_par = v ? (INDIRECT ? q.wparent[k] : q.parent[ii]) : -1;
        nx_info = v ? q.p0_info[ii] : 0u;
        nx_m0 = v ? q.p0_mask[ii] : 0ull;
        nx_po = v ? q.pair_ofs[ii] : 0u;
        nx_e1blk = 0; nx_e1mask = 0;
        if ((nx_info >> 8) > 1u) { nx_e1blk = q.pair_blk[nx_po]; nx_e1mask = q.pair_mask[nx_po]; }
        nx_npw = 0; nx_npm = 0; nx_npw1 = 0; nx_npm1 = 0;
        if (INDIRECT && nx_par <= -2) {
            const uint32_t np = (uint32_t)(-(nx_par + 2));
            const uint32_t wd = q.wd01[np];
            const ulonglong2 pm = q.fnarrow[np];
            nx_npw = wd & 0xFFu; nx_npm = pm.x;
            nx_npw1 = wd >> 8; nx_npm1 = (wd >> 8) != 0xFFu ? pm.y : 0ull;
        }
    };
    fetch(first);
    const bool prof = (q.dbg & 32u) != 0;
    unsigned long long t_all = 0, t0 = 0;
    unsigned long long n_compact = 0, n_mid = 0, n_full = 0;
    unsigned long long tph[4] = {0, 0, 0, 0};
    for (uint32_t base = first; base < end; base += WAVE) {
        if (prof) t0 = __builtin_amdgcn_s_memtime();
        B3Lane L;
        L.idx = nx_idx;
        L.valid = base + lane < end;
        L.n = nx_nl & 0xFFFFu; L.l = (nx_nl >> 16) & 0x3FFFu;
        L.w = nx_w; L.par = nx_par; L.info = nx_info; L.m0 = nx_m0; L.npw = nx_npw; L.npm = nx_npm; L.npw1 = nx_npw1; L.npm1 = nx_npm1;
        L.po = nx_po; L.e1blk = nx_e1blk; L.e1mask = nx_e1mask;
        if ((q.dbg & 8192u) && (L.info >> 8) > 1u) L.info = (L.info & 0xFFu) | 0x100u;   // timing experiment: ignore extra pairs
        if (base + WAVE < end) fetch(base + WAVE);
        // words this batch touches: own local words + the words inherited from a parent before the batch
        const uint32_t np = L.info >> 8;
        uint32_t lw = np ? (1u << (L.info & 0xFFu)) : 0u;
        if (np > 1) lw |= 1u << L.e1blk;
        if (__ballot(np > 2)) {
            for (uint32_t t = 1; t + 1 < np; ++t) lw |= 1u << q.pair_blk[L.po + t];
        }
        uint32_t rootslot = 0xFFFFFFFFu, inh = 0;
        if (L.valid && L.par >= 0 && L.par < (int32_t)base) {
            rootslot = C.slot_of_n[L.n - L.l];
            inh = C.chain_nz[rootslot];
        }
        uint32_t U = L.valid ? (lw | inh | (L.npm != 0 ? 1u << L.npw : 0u) | (L.npm1 != 0 ? 1u << L.npw1 : 0u)) : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) U |= (uint32_t)__shfl_xor((int)U, d, WAVE);
        U = __builtin_amdgcn_readfirstlane(U);
        const uint32_t ku = (uint32_t)__popc(U);
        if (false) {
        } else if (ku <= (uint32_t)B3_K) {
            uint32_t wl[B3_K];
            uint32_t rest = U;
#pragma unroll
            for (int s = 0; s < B3_K; ++s) { wl[s] = rest ? (uint32_t)__builtin_ctz(rest) : 0u; rest &= rest - 1; }
            b3_batch<B3_K, false, NBW, EMIT>(q, C, L, base, end, U, wl, ku, rootslot, inh, tph);
            ++n_mid;
        } else {
            uint32_t wl[NBW];
#pragma unroll
            for (int s = 0; s < NBW; ++s) wl[s] = (uint32_t)s;
            b3_batch<NBW, true, NBW, EMIT>(q, C, L, base, end, U, wl, (uint32_t)NBW, rootslot, inh, tph);
            ++n_full;
        }
        if (prof) t_all += __builtin_amdgcn_s_memtime() - t0;
    }
    if (prof && lane == 0) {
        atomicAdd(&q.counters[1], t_all); atomicAdd(&q.counters[2], tph[0]);       // all, load + inherit
        atomicAdd(&q.counters[3], n_mid); atomicAdd(&q.counters[4], n_full);
        atomicAdd(&q.counters[5], tph[1]); atomicAdd(&q.counters[6], tph[2]); atomicAdd(&q.counters[7], tph[3]);   // doubling, records, chain
    }
    if (!EMIT) {
        lds_sync();
        for (uint32_t k = lane; k < q.nctr; k += WAVE) my_table[k] = C.ctr[k];
    }
}

// ------------------------------------------------------------------------------------------
// K1 for the nodes whose full list touches at most TWO blocks (9 nodes in 10).  Lists are ascending and a child only
// appends, so a child's first block is its parent's first block and a second block, once there, stays: the ancestors of
// such a node are of the same kind and a full list is two registers (F0 for block w0, F1 for block w1 > w0 or none).
// The kernel walks the whole DFS stream, 64 nodes per step, one per lane (lanes of wider nodes idle):
//   F = local masks | F(parent), register by register: parents inside the batch by pointer doubling, parents before
//   the batch from chain[depth - 2] in LDS (one slot per depth = the latest node of that depth on the current root
//   path).  Records (w0, w0, F0) for every pattern with k-mers, plus (w1, w0, F1, F0) and (w1, w1, F1) when there
//   is a second block, straight into their streams.
// Nodes that have a wider child also leave (F0, F1) in HBM (fnarrow) for the wide-list launch.
// ------------------------------------------------------------------------------------------
struct B3NParams {
    const uint32_t* nl;            // n | l << 16 | has-wider-child << 30 | wide (more than two blocks) << 31
    const int32_t* parent;
    const uint32_t* w;
    const uint8_t* depth;
    const uint16_t* wd01;          // first block | second block << 8 (0xFF: none)
    const Segment* segs;
    const uint32_t* seg_anc;       // [n_segs][chain_cap] root-first ancestors of the segment's first node
    const uint32_t* seg_anc_n;
    const unsigned long long* p0_mask;
    const uint16_t* p0_info;
    const uint32_t* pair_ofs;
    const unsigned long long* pair_mask;
    ulonglong2* fnarrow;
    uint32_t n_segs, chain_cap, nctr;
    uint32_t* table;               // [n_segs][nctr]
    B2Recs rec;
    uint32_t dbg;
};
constexpr int B3N_WAVES = 4;

__host__ __device__ inline size_t b3n_wave_bytes(uint32_t chain_cap, uint32_t nctr) {
    return ((size_t)chain_cap * 16 + (size_t)nctr * 4 + 15) & ~(size_t)15;
}

template <bool EMIT>
__global__ __launch_bounds__(WAVE * B3N_WAVES) void b3_narrow_kernel(B3NParams q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg = blockIdx.x * B3N_WAVES + wave;
    if (seg >= q.n_segs) return;
    ulonglong2* chain = (ulonglong2*)(lds_raw + b3n_wave_bytes(q.chain_cap, q.nctr) * wave);                 // [chain_cap]
    uint32_t* ctr = (uint32_t*)(chain + q.chain_cap);                                                        // [nctr]
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t* my_table = q.table + (size_t)seg * q.nctr;
    for (uint32_t k = lane; k < q.nctr; k += WAVE) ctr[k] = EMIT ? my_table[k] : 0u;
    const Segment sg = q.segs[seg];
    const uint32_t first = __builtin_amdgcn_readfirstlane(sg.first);
    const uint32_t end = __builtin_amdgcn_readfirstlane(sg.end);
    if (first >= end) return;

    // local masks of a node, sorted into the two registers
    auto locals = [&](uint32_t node, uint32_t info, unsigned long long m0, uint32_t wd, unsigned long long& f0, unsigned long long& f1) {
        f0 = 0; f1 = 0;
        const uint32_t np = info >> 8;
        if (np == 0) return;
        if ((info & 0xFFu) == (wd & 0xFFu)) f0 = m0; else f1 = m0;
        if (np > 1) f1 |= q.pair_mask[q.pair_ofs[node]];             // the second pair can only lie in the second block
    };

    // chain slots of the first node's ancestors: inclusive OR along the root path (the nodes of this kind are a prefix of it)
    {
        const uint32_t d = q.seg_anc_n[seg];
        unsigned long long c0 = 0, c1 = 0;
        for (uint32_t cb = 0; cb < d; cb += WAVE) {
            const uint32_t k = cb + lane;
            const bool on = k < d;
            const uint32_t node = on ? q.seg_anc[(size_t)seg * q.chain_cap + k] : 0u;
            unsigned long long F0 = 0, F1 = 0;
            if (on && !(q.nl[node] >> 31)) locals(node, q.p0_info[node], q.p0_mask[node], q.wd01[node], F0, F1);
            if (lane == 0) { F0 |= c0; F1 |= c1; }
#pragma unroll
            for (int s = 1; s < WAVE; s <<= 1) {
                const unsigned long long o0 = shfl_up64(F0, s), o1 = shfl_up64(F1, s);
                if (lane >= (uint32_t)s) { F0 |= o0; F1 |= o1; }
            }
            if (on) chain[k] = make_ulonglong2(F0, F1);
            c0 = shfl64(F0, WAVE - 1); c1 = shfl64(F1, WAVE - 1);
        }
        lds_sync();
    }

    uint32_t nx_nl = 0, nx_w = 0, nx_info = 0, nx_dep = 0, nx_wd = 0;
    int32_t nx_par = -1;
    unsigned long long nx_m0 = 0, nx_e1 = 0;
    auto fetch = [&](uint32_t b0) {
        const uint32_t ii = b0 + lane;
        const bool v = ii < end;
        nx_nl = v ? q.nl[ii] : 0x80000000u;
        nx_w = v ? q.w[ii] : 0u;
        nx_par = v ? q.parent[ii] : -1;
        nx_dep = v ? q.depth[ii] : 0xFFFFu;
        nx_info = v ? q.p0_info[ii] : 0u;
        nx_m0 = v ? q.p0_mask[ii] : 0ull;
        nx_wd = v ? q.wd01[ii] : 0xFFFFu;
        nx_e1 = 0;
        if (v && !(nx_nl >> 31) && (nx_info >> 8) > 1u) nx_e1 = q.pair_mask[q.pair_ofs[ii]];
    };
    fetch(first);
    for (uint32_t base = first; base < end; base += WAVE) {
        const uint32_t idx = base + lane;
        const uint32_t nl = nx_nl, w = nx_w, info = nx_info, dep = nx_dep, wd = nx_wd;
        const int32_t par = nx_par;
        const unsigned long long m0 = nx_m0, e1 = nx_e1;
        if (base + WAVE < end) fetch(base + WAVE);
        const bool nar = !(nl >> 31);                       // lanes past the end carry the wide flag
        const uint32_t w0 = wd & 0xFFu, w1 = wd >> 8;
        unsigned long long F0 = 0, F1 = 0;
        if (nar && (info >> 8) != 0) {
            if ((info & 0xFFu) == w0) F0 = m0; else F1 = m0;
            F1 |= e1;
        }
        if (nar && par >= 0 && par < (int32_t)base) { const ulonglong2 c = chain[dep - 2u]; F0 |= c.x; F1 |= c.y; }
        int pl = (nar && par >= (int32_t)base) ? (int)(par - (int32_t)base) : -1;
        while (__ballot(pl >= 0)) {
            const int src = pl >= 0 ? pl : (int)lane;
            const unsigned long long o0 = shfl64(F0, src), o1 = shfl64(F1, src);
            if (pl >= 0) { F0 |= o0; F1 |= o1; }
            const int npl = __shfl(pl, src, WAVE);
            pl = pl >= 0 ? npl : -1;
        }
        if (nar && ((nl >> 30) & 1u)) q.fnarrow[idx] = make_ulonglong2(F0, F1);
        // ---- records
        const bool act = nar && w != 0 && (nl & 0xFFFFu) >= 2u && !(q.dbg & 512u);
        const uint32_t cls = b2_weight_class(w);
        // (w0, w0): the lanes of a batch mostly share the block, so one reservation per block and weight class
        unsigned long long pend = __ballot(act && F0 != 0);
        while (pend) {
            const uint32_t X0 = bcast(w0, (uint32_t)__builtin_ctzll(pend));
            const bool mine = act && F0 != 0 && w0 == X0;
            const unsigned long long b0 = __ballot(mine && cls == 0u), b1 = __ballot(mine && cls == 1u), b2 = __ballot(mine && cls == 2u);
            uint32_t mybase = 0;
            if (lane < B2_NCLS) {
                const uint32_t cnt = (uint32_t)__popcll(lane == 0 ? b0 : lane == 1 ? b1 : b2);
                if (cnt) mybase = atomicAdd(&ctr[(X0 * (X0 + 1u) / 2u + X0) * B2_NCLS + lane], cnt);
            }
            const uint32_t base0 = bcast(mybase, 0), base1 = bcast(mybase, 1), base2 = bcast(mybase, 2);
            if (EMIT && mine && !(q.dbg & 256u)) {
                const uint32_t slot = cls == 0u ? base0 + (uint32_t)__popcll(b0 & lt_mask)
                                    : cls == 1u ? base1 + (uint32_t)__popcll(b1 & lt_mask) : base2 + (uint32_t)__popcll(b2 & lt_mask);
                q.rec.rows[slot] = F0;
                if (cls) q.rec.w[slot] = w;
            }
            pend &= ~(b0 | b1 | b2);
        }
        // (w1, w0) and (w1, w1): the second blocks differ from lane to lane, one LDS atomic per record
        if (act && F1 != 0) {
            const uint32_t t1 = w1 * (w1 + 1u) / 2u;
            if (F0 != 0) {
                const uint32_t s1 = atomicAdd(&ctr[(t1 + w0) * B2_NCLS + cls], 1u);
                if (EMIT && !(q.dbg & 256u)) { q.rec.rc[s1] = make_ulonglong2(F1, F0); if (cls) q.rec.w[s1] = w; }
            }
            const uint32_t s2 = atomicAdd(&ctr[(t1 + w1) * B2_NCLS + cls], 1u);
            if (EMIT && !(q.dbg & 256u)) { q.rec.rows[s2] = F1; if (cls) q.rec.w[s2] = w; }
        }
        // ---- chain slots for the next batch: the nodes on the root path of this batch's last node, i.e. the
        // lanes whose depth is smaller than the depth of every later lane
        if (base + WAVE < end) {
            uint32_t m = dep;
#pragma unroll
            for (int s = 1; s < WAVE; s <<= 1) {
                const uint32_t o = (uint32_t)__shfl_down((int)m, s, WAVE);
                if (lane + (uint32_t)s < (uint32_t)WAVE) m = o < m ? o : m;
            }
            uint32_t later = (uint32_t)__shfl_down((int)m, 1, WAVE);
            if (lane == (uint32_t)WAVE - 1u) later = 0xFFFFFFFFu;
            if (nar && dep < later) chain[dep - 1u] = make_ulonglong2(F0, F1);
            lds_sync();
        }
    }
    if (!EMIT) {
        lds_sync();
        for (uint32_t k = lane; k < q.nctr; k += WAVE) my_table[k] = ctr[k];
    }
}

// 64 x 64 bit-matrix transpose across the lanes of a wave: lane i holds row i on entry, column i on exit
__device__ __forceinline__ unsigned long long transpose64(unsigned long long x, uint32_t lane) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    // stage 32: the upper half of the wave swaps its low words with the high words of the lower half — one
    // v_permlane32_swap on gfx950
    {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    }
    // stage 16: 16-bit halves of every word between the lanes that differ in bit 4: v_permlane16_swap brings the
    // partner's word, v_perm_b32 splices the halves
    {
        const bool up = (lane & 16u) != 0;
        const uint32_t sel = up ? 0x03020706u : 0x05040100u;
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const uint32_t plo = up ? a[0] : a[1];
        lo = __builtin_amdgcn_perm(plo, lo, sel);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const uint32_t phi = up ? b[0] : b[1];
        hi = __builtin_amdgcn_perm(phi, hi, sel);
    }
    x = ((unsigned long long)hi << 32) | lo;
    const unsigned long long masks[4] = {0x00FF00FF00FF00FFull, 0x0F0F0F0F0F0F0F0Full, 0x3333333333333333ull, 0x5555555555555555ull};
    int s = 8;
#pragma unroll
    for (int k = 0; k < 4; ++k, s >>= 1) {
        const unsigned long long m = masks[k];
        uint32_t plo, phi;
        if (s == 2) {               // quad_perm [2,3,0,1]
            plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, 0x4E, 0xF, 0xF, false);
            phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), 0x4E, 0xF, 0xF, false);
        } else if (s == 1) {        // quad_perm [1,0,3,2]
            plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, 0xB1, 0xF, 0xF, false);
            phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), 0xB1, 0xF, 0xF, false);
        } else {
            plo = (uint32_t)__shfl_xor((int)(uint32_t)x, s, WAVE);
            phi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), s, WAVE);
        }
        const unsigned long long pv = ((unsigned long long)phi << 32) | plo;
        x = (lane & (uint32_t)s) ? ((x & ~m) | ((pv >> s) & m)) : ((x & m) | ((pv & m) << s));
    }
    return x;
}

// K2, popcount form (weights >= 128 only; everything else goes through b2_apply_item_mfma below): one workgroup per
// (bucket, chunk of records).  A wave takes 64 records per step, one per lane, and turns them into bit matrices over
// the records: lane c holds Ct = "which of the 64 records contain column c", and Rt_r = "which records contain row r"
// is read from LDS.  cell(r, c) += popcount(Ct & Rt_r & plane_b) << b for every bit plane b of the weights that occurs
// in the step.  The cells live in registers (lane c keeps column c of the 64 x 64 block, one register per row) and are
// merged through LDS once per work item.
// DIAG (X == Y, rows == cols): the block is symmetric and only c < r is wanted.  Lane c then takes the rows
// (c + d) mod width for d = 1 .. width/2 instead of all rows: every unordered pair of samples exactly once (for an even
// width the distance width/2 is kept by the lower half of the lanes) — half the row loop; the row mask is a per-lane
// LDS read instead of a broadcast.
template <int CLS, bool DIAG>
__device__ __forceinline__ void b2_apply_item(const B2Recs& rec, const B2Item& it, uint32_t* __restrict__ M, uint32_t dbg, uint32_t bwidth,
                                              uint32_t* acc, unsigned long long (*rtbuf)[64]) {
    if ((dbg & 64u) && CLS == 0) return;
    if ((dbg & 128u) && CLS != 0) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool diag = DIAG;
    const uint32_t njs = DIAG ? bwidth / 2u : bwidth;                    // accumulators in use
    const uint32_t wrapd = bwidth - lane;                                 // DIAG: row of accumulator j = lane + j + 1 (mod width)
    uint32_t a[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) a[r] = 0;
    const unsigned long long* rt = rtbuf[wave];
#define ROWMASK(j) (DIAG ? rt[((uint32_t)(j) + 1u < wrapd ? lane + (uint32_t)(j) + 1u : lane + (uint32_t)(j) + 1u - bwidth) & 63u] : rt[(j)])
    // the next group's records are fetched while the current group is reduced
    unsigned long long nR = 0, nC = 0;
    uint32_t nW = 0;
    auto fetch = [&](uint32_t g) {
        const uint32_t j = g + lane;
        nR = 0; nC = 0; nW = 0;
        if (j < it.end) {
            if (diag) { nR = rec.rows[j]; nC = nR; }
            else { const ulonglong2 rc = rec.rc[j]; nR = rc.x; nC = rc.y; }
            nW = CLS ? rec.w[j] : 1u;
        }
    };
    fetch(it.begin + wave * 64);
    for (uint32_t g0 = it.begin + wave * 64; g0 < it.end; g0 += 256) {
        const unsigned long long R = nR, C = nC;
        const uint32_t W = nW;
        if (g0 + 256 < it.end) fetch(g0 + 256);
        const unsigned long long Ct = transpose64(C, lane);
        if (dbg & 2048u) { if ((uint32_t)Ct + (uint32_t)R + W == 0x12345u) acc[lane] = 1; continue; }
        rtbuf[wave][lane] = diag ? Ct : transpose64(R, lane);       // on the diagonal rows == cols
        lds_sync();
        if (CLS == 0) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if ((uint32_t)(g * 4) < njs) {                        // rows >= the block width never occur
                    asm volatile("" ::: "memory");    // keep a group's LDS reads together (register pressure)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int r = g * 4 + k; a[r] += (uint32_t)__popcll(Ct & ROWMASK(r)); }
                }
            }
        } else {
            uint32_t wor = W;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) wor |= (uint32_t)__shfl_xor((int)wor, d, WAVE);
            wor = __builtin_amdgcn_readfirstlane(wor);
            for (uint32_t wb = wor; wb; wb &= wb - 1) {
                const uint32_t b = (uint32_t)__builtin_ctz(wb);
                const unsigned long long Cb = Ct & __ballot(((W >> b) & 1u) != 0);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    if ((uint32_t)(g * 4) < njs) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const int r = g * 4 + k; a[r] += (uint32_t)__popcll(Cb & ROWMASK(r)) << b; }
                    }
                }
            }
        }
        lds_sync();
    }
    // merge the four waves' registers, then one HBM atomic per non-zero cell (on the diagonal only c < r)
#undef ROWMASK
    if (DIAG) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint32_t d = (uint32_t)j + 1u;
            if (d > njs || lane >= bwidth) continue;
            if (2u * d == bwidth && lane >= d) continue;                 // even width: the opposite sample, once
            const uint32_t r2 = d < wrapd ? lane + d : lane + d - bwidth;
            const uint32_t row = r2 > lane ? r2 : lane, col = r2 > lane ? lane : r2;
            if (a[j]) atomicAdd(&acc[row * 64 + col], a[j]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 64; ++r)
            if (a[r]) atomicAdd(&acc[r * 64 + lane], a[r]);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) {
        const uint32_t v = acc[k];
        if (!v) continue;
        const uint64_t row = (uint64_t)it.X * bwidth + (k >> 6), col = (uint64_t)it.Y * bwidth + (k & 63u);
        if (!(dbg & 1024u)) atomicAdd(&M[tri64(row) + col], v);
    }
}

// The same accumulation on the matrix cores: over the 64 records of a step,
//     cell(r, c) += sum_k  w_k [r in rows_k] * [c in cols_k]    =  (A B)(r, c),   A = 64 x 64 int8 (rows x records, weighted),
//                                                                                B = 64 x 64 int8 (records x cols, 0/1)
// as eight v_mfma_i32_32x32x32_i8 (operand layout probed in profiles/r01_mfma_i8_layout_probe.hip: lane l holds
// A[l & 31][16 (l >> 5) + j], B[16 (l >> 5) + j][l & 31], j < 16; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)).
// The bit matrices R^T / C^T of the step are parked in LDS; a lane turns 16 of their bits into 16 operand bytes with two
// reads of a 256-entry byte-spreading table and ANDs the weights in.  The work does not depend on the weights, so
// every weight below 128 costs the same as 1 — this path takes class 1 and the off-diagonal part of class 0 (on the
// diagonal of class 0 the folded popcount loop is cheaper).
typedef int b2_v4i __attribute__((ext_vector_type(4)));
typedef int b2_v16i __attribute__((ext_vector_type(16)));

template <bool WEIGHTED, bool DIAG>
__device__ __forceinline__ void b2_apply_item_mfma(const B2Recs& rec, const B2Item& it, uint32_t* __restrict__ M, uint32_t dbg, uint32_t bwidth,
                                                   uint32_t* acc, unsigned long long (*rtbuf)[64], unsigned long long (*ctbuf)[64],
                                                   unsigned char (*wbuf)[64], const unsigned long long* lut_ff, const unsigned long long* lut_01) {
    if ((dbg & 64u) && !WEIGHTED) return;
    if ((dbg & 128u) && WEIGHTED) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t half = lane >> 5, l31 = lane & 31u;
    b2_v16i c00 = {}, c01 = {}, c10 = {}, c11 = {};
    const unsigned long long* lut_a = WEIGHTED ? lut_ff : lut_01;
    auto spread = [&](unsigned long long word, uint32_t shift, const unsigned long long* lut) -> b2_v4i {
        const uint32_t f = (uint32_t)(word >> shift) & 0xFFFFu;
        const unsigned long long lo = lut[f & 0xFFu], hi = lut[f >> 8];
        b2_v4i r;
        r[0] = (int)(uint32_t)lo; r[1] = (int)(uint32_t)(lo >> 32); r[2] = (int)(uint32_t)hi; r[3] = (int)(uint32_t)(hi >> 32);
        return r;
    };
    unsigned long long nR = 0, nC = 0;
    uint32_t nW = 0;
    auto fetch = [&](uint32_t g) {
        const uint32_t j = g + lane;
        nR = 0; nC = 0; nW = 0;
        if (j < it.end) {
            if (DIAG) { nR = rec.rows[j]; nC = nR; }
            else { const ulonglong2 rc = rec.rc[j]; nR = rc.x; nC = rc.y; }
            nW = WEIGHTED ? rec.w[j] : 1u;
        }
    };
    fetch(it.begin + wave * 64);
    for (uint32_t g0 = it.begin + wave * 64; g0 < it.end; g0 += 256) {
        const unsigned long long R = nR, C = nC;
        const uint32_t W = nW;
        if (g0 + 256 < it.end) fetch(g0 + 256);
        const unsigned long long Ct = transpose64(C, lane);
        rtbuf[wave][lane] = DIAG ? Ct : transpose64(R, lane);
        if (!DIAG) ctbuf[wave][lane] = Ct;
        if (WEIGHTED) wbuf[wave][lane] = (unsigned char)W;
        lds_sync();
        const unsigned long long* rtp = rtbuf[wave];
        const unsigned long long* ctp = DIAG ? rtbuf[wave] : ctbuf[wave];
        const unsigned long long ra0 = rtp[l31], ra1 = rtp[32u + l31], cb0 = ctp[l31], cb1 = ctp[32u + l31];
#pragma unroll
        for (uint32_t kh = 0; kh < 2; ++kh) {
            const uint32_t shift = 32u * kh + 16u * half;            // records 32 kh + 16 half .. + 15 of the step
            b2_v4i a0 = spread(ra0, shift, lut_a), a1 = spread(ra1, shift, lut_a);
            if (WEIGHTED) {
                const b2_v4i wv = *(const b2_v4i*)(wbuf[wave] + shift);
                a0 &= wv; a1 &= wv;
            }
            const b2_v4i b0 = spread(cb0, shift, lut_01), b1 = spread(cb1, shift, lut_01);
            c00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c11, 0, 0, 0);
        }
        lds_sync();
    }
    // merge the four waves' tiles through the LDS block (on the diagonal only c < r), then one HBM atomic per non-zero cell
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t row0 = (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * half;
        const uint32_t v00 = (uint32_t)c00[r], v01 = (uint32_t)c01[r], v10 = (uint32_t)c10[r], v11 = (uint32_t)c11[r];
        if (v00 && (!DIAG || l31 < row0)) atomicAdd(&acc[row0 * 64 + l31], v00);
        if (v01 && (!DIAG || 32u + l31 < row0)) atomicAdd(&acc[row0 * 64 + 32u + l31], v01);
        if (v10 && (!DIAG || l31 < 32u + row0)) atomicAdd(&acc[(32u + row0) * 64 + l31], v10);
        if (v11 && (!DIAG || l31 < row0)) atomicAdd(&acc[(32u + row0) * 64 + 32u + l31], v11);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) {
        const uint32_t v = acc[k];
        if (!v) continue;
        const uint64_t row = (uint64_t)it.X * bwidth + (k >> 6), col = (uint64_t)it.Y * bwidth + (k & 63u);
        if (!(dbg & 1024u)) atomicAdd(&M[tri64(row) + col], v);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void b2_apply_kernel(const B2Recs rec, const B2Item* __restrict__ items,
                                                       uint32_t* __restrict__ M, uint32_t N, uint32_t dbg, uint32_t bwidth) {
    __shared__ uint32_t acc[64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned long long rtbuf[4][64];
    __shared__ __attribute__((aligned(16))) unsigned long long ctbuf[4][64];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[4][64];
    __shared__ unsigned long long lut_ff[256], lut_01[256];       // byte b -> its 8 bits spread over 8 bytes (0xFF / 0x01 where set)
    const B2Item it = items[blockIdx.x];
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) acc[k] = 0;
    {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((threadIdx.x >> i) & 1u) ? 0xFFull << (8 * i) : 0ull;
        lut_ff[threadIdx.x] = v;
        lut_01[threadIdx.x] = v & 0x0101010101010101ull;
    }
    __syncthreads();
    if (it.X == it.Y) {
        if (it.cls == 0) b2_apply_item_mfma<false, true>(rec, it, M, dbg, bwidth, acc, rtbuf, ctbuf, wbuf, lut_ff, lut_01);
        else if (it.cls == 1) b2_apply_item_mfma<true, true>(rec, it, M, dbg, bwidth, acc, rtbuf, ctbuf, wbuf, lut_ff, lut_01);
        else b2_apply_item<2, true>(rec, it, M, dbg, bwidth, acc, rtbuf);
    } else {
        if (it.cls == 0) b2_apply_item_mfma<false, false>(rec, it, M, dbg, bwidth, acc, rtbuf, ctbuf, wbuf, lut_ff, lut_01);
        else if (it.cls == 1) b2_apply_item_mfma<true, false>(rec, it, M, dbg, bwidth, acc, rtbuf, ctbuf, wbuf, lut_ff, lut_01);
        else b2_apply_item<2, false>(rec, it, M, dbg, bwidth, acc, rtbuf);
    }
}


// ------------------------------------------------------------------------------------------
// host side: launches, upload-time preparation
// ------------------------------------------------------------------------------------------

size_t b2_lds_per_wave(uint32_t maxn_pad, uint32_t dec_cap, uint32_t nctr) { return b2_wave_bytes(maxn_pad, dec_cap, nctr); }

template <bool EMIT>
int b2_launch_emit(kmdb_db* db, uint32_t seg_begin, uint32_t seg_end, uint32_t dbg, hipStream_t st) {
    B2Params q{};
    q.a.meta = db->meta; q.a.bitpos = db->bitpos; q.a.parent = db->parent; q.a.sub_end = db->sub_end;
    q.a.wprefix = db->wprefix; q.a.bits = db->bits; q.a.segs = db->rsegs;
    q.a.seg_begin = seg_begin; q.a.seg_end = seg_end; q.a.dbg = dbg; q.a.counters = db->counters;
    q.maxn_pad = db->b2_maxn_pad; q.dec_cap = db->b2_dec_cap; q.nctr = db->b2_nctr;
    q.bm = BlockMap{db->b2_width, (uint32_t)((1ull << 32) / db->b2_width) + 1u};
    q.table = db->b2_table; q.rec = B2Recs{db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w}; q.w = db->w;
    const size_t lds = b2_lds_per_wave(q.maxn_pad, q.dec_cap, q.nctr) * B2_WAVES;
    HIP_TRY(hipFuncSetAttribute((const void*)b2_emit_kernel<EMIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t nseg = seg_end - seg_begin;
    const uint32_t blocks = (nseg + B2_WAVES - 1) / B2_WAVES;
    if (blocks) hipLaunchKernelGGL((b2_emit_kernel<EMIT>), dim3(blocks), dim3(WAVE * B2_WAVES), lds, st, q);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NBW, bool EMIT, bool INDIRECT>
int b3_launch_emit_t(kmdb_db* db, hipStream_t st, uint32_t dbg, uint32_t* fwords) {
    B3Params q{};
    q.nl = db->b3_nl; q.parent = db->parent; q.w = db->w;
    q.p0_mask = db->b3_p0_mask; q.p0_info = db->b3_p0_info;
    q.pair_ofs = db->b3_pair_ofs; q.pair_blk = db->b3_pair_blk; q.pair_mask = db->b3_pair_mask;
    q.maxn_pad = db->b2_maxn_pad; q.nctr = db->b2_nctr; q.chain_cap = db->b3_chain_cap;
    q.table = db->b2_table; q.rec = B2Recs{db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w}; q.dbg = dbg; q.counters = db->counters;
    q.fwords = fwords;
    if (INDIRECT) {
        q.segs = db->b3_wsegs; q.n_segs = db->b3_n_wsegs; q.seg_anc = db->b3_wseg_anc; q.seg_anc_n = db->b3_wseg_anc_n;
        q.widx = db->b3_widx; q.wparent = db->b3_wparent; q.fnarrow = db->b3_fnarrow; q.wd01 = db->b3_wd01; q.seg_np = db->b3_wseg_np;
        q.seg_row0 = 0; q.chain_cap = db->b3_wchain_cap;
    } else {
        q.segs = db->rsegs; q.n_segs = db->n_rsegs; q.seg_anc = db->b3_seg_anc; q.seg_anc_n = db->b3_seg_anc_n;
    }
    const size_t lds = b3_wave_bytes(NBW, q.maxn_pad, q.nctr, q.chain_cap) * B3_WAVES;
    HIP_TRY(hipFuncSetAttribute((const void*)b3_emit_kernel<NBW, EMIT, INDIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (EMIT && getenv("KMDB_VERBOSE")) {
        static bool once = false;
        if (!once) {
            once = true;
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)b3_emit_kernel<NBW, EMIT, INDIRECT>, WAVE * B3_WAVES, lds);
            fprintf(stderr, "[kmdb] emit kernel%s: NBW %d, width %u, chain_cap %u, LDS/block %zu B, resident blocks/CU %d, segments %u\n",
                    INDIRECT ? " (wide list)" : "", NBW, db->b2_width, q.chain_cap, lds, nb, q.n_segs);
        }
    }
    const uint32_t blocks = (q.n_segs + B3_WAVES - 1) / B3_WAVES;
    if (blocks) hipLaunchKernelGGL((b3_emit_kernel<NBW, EMIT, INDIRECT>), dim3(blocks), dim3(WAVE * B3_WAVES), lds, st, q);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool EMIT, bool INDIRECT>
int b3_launch_emit(kmdb_db* db, hipStream_t st, uint32_t dbg = 0, uint32_t* fwords = nullptr) {
    if (db->b3_nbw <= 8) return b3_launch_emit_t<8, EMIT, INDIRECT>(db, st, dbg, fwords);
    if (db->b3_nbw <= 16) return b3_launch_emit_t<16, EMIT, INDIRECT>(db, st, dbg, fwords);
    if (db->b3_nbw <= 20) return b3_launch_emit_t<20, EMIT, INDIRECT>(db, st, dbg, fwords);
    if (db->b3_nbw <= 24) return b3_launch_emit_t<24, EMIT, INDIRECT>(db, st, dbg, fwords);
    return b3_launch_emit_t<32, EMIT, INDIRECT>(db, st, dbg, fwords);
}

template <bool EMIT>
int b3_launch_narrow(kmdb_db* db, hipStream_t st, uint32_t dbg = 0) {
    B3NParams q{};
    q.nl = db->b3_nl; q.parent = db->parent; q.w = db->w; q.depth = db->b3_depth; q.segs = db->b3_nsegs;
    q.seg_anc = db->b3_nseg_anc; q.seg_anc_n = db->b3_nseg_anc_n; q.p0_mask = db->b3_p0_mask; q.p0_info = db->b3_p0_info;
    q.fnarrow = db->b3_fnarrow; q.n_segs = db->b3_n_nsegs; q.chain_cap = db->b3_chain_cap; q.nctr = db->b2_nctr;
    q.wd01 = db->b3_wd01; q.pair_ofs = db->b3_pair_ofs; q.pair_mask = db->b3_pair_mask;
    q.table = db->b3_ntable; q.rec = B2Recs{db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w}; q.dbg = dbg;
    const size_t lds = b3n_wave_bytes(q.chain_cap, q.nctr) * B3N_WAVES;
    HIP_TRY(hipFuncSetAttribute((const void*)b3_narrow_kernel<EMIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t blocks = (q.n_segs + B3N_WAVES - 1) / B3N_WAVES;
    if (blocks) hipLaunchKernelGGL((b3_narrow_kernel<EMIT>), dim3(blocks), dim3(WAVE * B3N_WAVES), lds, st, q);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool COUNT>
int b3_launch_decode(kmdb_db* db, hipStream_t st) {
    const uint32_t P = (uint32_t)db->P;
    const BlockMap bm{db->b2_width, (uint32_t)((1ull << 32) / db->b2_width) + 1u};
    const uint32_t kdbg = getenv("KMDB_K0_DBG") ? (uint32_t)atoi(getenv("KMDB_K0_DBG")) : 0u;
    if (P && !getenv("KMDB_SKIP_K0A"))
        hipLaunchKernelGGL((b3_decode_kernel<COUNT, false>), dim3((P + 255) / 256), dim3(256), 0, st, db->b3_k0in, db->b3_bitrel, db->b3_blkbase, db->bits,
                           (const uint32_t*)nullptr, P, db->b3_short_max, bm, db->b3_p0_mask, db->b3_p0_info, db->b3_pair_ofs,
                           db->b3_pair_blk, db->b3_pair_mask, kdbg);
    if (db->b3_n_long && !getenv("KMDB_SKIP_K0B"))
        hipLaunchKernelGGL((b3_decode_kernel<COUNT, true>), dim3((db->b3_n_long + 255) / 256), dim3(256), 0, st, db->b3_k0in, db->b3_bitrel,
                           db->b3_blkbase, db->bits, (const uint32_t*)db->b3_perm, db->b3_n_long, db->b3_short_max, bm, db->b3_p0_mask, db->b3_p0_info,
                           db->b3_pair_ofs, db->b3_pair_blk, db->b3_pair_mask, kdbg);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Decide whether the database qualifies for the block-record pipeline and, if so, tabulate the
// per-(segment, bucket) record counts with the count mode of the emit kernel (layout metadata:
// a pure function of the database, like CSR row pointers), turn them into record bases and cut
// the buckets into work items for the apply kernel.
void b2_release_width(kmdb_db* db) {
    void* ptrs[] = {db->b2_table, db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w, db->b2_items, db->b3_pair_ofs,
                    db->b3_pair_blk, db->b3_pair_mask, db->b3_p0_mask, db->b3_p0_info, db->b3_widx, db->b3_wparent,
                    db->b3_fnarrow, db->b3_wsegs, db->b3_wseg_anc, db->b3_wseg_anc_n, db->b3_wseg_np, db->b3_ntable, db->b3_wd01};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    db->b3_widx = nullptr; db->b3_wparent = nullptr; db->b3_fnarrow = nullptr; db->b3_wsegs = nullptr;
    db->b3_wseg_anc = nullptr; db->b3_wseg_anc_n = nullptr; db->b3_wseg_np = nullptr; db->b3_ntable = nullptr; db->b3_wd01 = nullptr;
    db->b3_split = false; db->b3_n_wide = 0; db->b3_n_wsegs = 0;
    db->b2_table = nullptr; db->b2_rec_rows = nullptr; db->b2_rec_rc = nullptr; db->b2_rec_w = nullptr; db->b2_items = nullptr;
    db->b3_pair_ofs = nullptr; db->b3_pair_blk = nullptr; db->b3_pair_mask = nullptr; db->b3_p0_mask = nullptr;
    db->b3_p0_info = nullptr;
    db->b2_ready = db->b3_ready = false;
}

// Everything of the block-record pipeline that depends on the block width: run the count modes of the
// kernels (layout metadata: a pure function of the database, like CSR row pointers), turn the
// per-(segment, bucket) record counts into record bases and cut the buckets into work items for the
// apply kernel.  *fits is false when the width cannot be used.
int b2_prepare_width(kmdb_db* db, uint32_t width, const kmdb_host_layout& h, bool* fits, bool estimate_only = false) {
    const uint32_t max_n = h.max_n;
    const bool chain_ok = h.chain_ok;
    *fits = false;
    const uint64_t N = db->N, P = db->P;
    const uint32_t NB = (uint32_t)((N + width - 1) / width);
    if (NB > 32) return 0;
    db->b2_width = width;
    db->b2_maxn_pad = std::max<uint32_t>(64, (max_n + 63) / 64 * 64);
    db->b2_dec_cap = std::max<uint32_t>(512, db->b2_maxn_pad);
    db->b2_nctr = NB * (NB + 1) / 2 * B2_NCLS;
    if (b2_lds_per_wave(db->b2_maxn_pad, db->b2_dec_cap, db->b2_nctr) * B2_WAVES > 160 * 1024) return 0;
    size_t table_rows = db->n_rsegs;
    size_t tbl = table_rows * db->b2_nctr;
    size_t ntbl = 0;                                       // narrow kernel's own table (split front half only)
    HIP_TRY(hipMalloc((void**)&db->b2_table, tbl * 4));
    HIP_TRY(hipMemset(db->b2_table, 0, tbl * 4));
    const uint32_t nbw = NB <= 8 ? 8 : NB <= 16 ? 16 : NB <= 20 ? 20 : NB <= 24 ? 24 : 32;
    const bool use_b3 = chain_ok && b3_wave_bytes(nbw, db->b2_maxn_pad, db->b2_nctr, db->b3_chain_cap) * B3_WAVES <= 160 * 1024;
    if (use_b3) {
        // K0 count pass -> pair offsets -> K0 emit (the pairs are needed by the record count pass below)
        db->b3_nbw = nbw;
        HIP_TRY(hipMalloc((void**)&db->b3_pair_ofs, (P + 1) * 4));
        HIP_TRY(hipMemset(db->b3_pair_ofs, 0, (P + 1) * 4));
        HIP_TRY(hipMalloc((void**)&db->b3_p0_mask, P * 8));
        HIP_TRY(hipMalloc((void**)&db->b3_p0_info, P * 2));
        if (b3_launch_decode<true>(db, db->stream)) return 1;
        uint32_t* tmp_counts = nullptr;
        HIP_TRY(hipMalloc((void**)&tmp_counts, (P + 1) * 4));
        HIP_TRY(hipMemcpyAsync(tmp_counts, db->b3_pair_ofs, (P + 1) * 4, hipMemcpyDeviceToDevice, db->stream));
        size_t tb = 0;
        void* tmp = nullptr;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, tmp_counts, db->b3_pair_ofs, (int)(P + 1), db->stream));
        HIP_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tb, tmp_counts, db->b3_pair_ofs, (int)(P + 1), db->stream));
        uint32_t total_pairs = 0;
        HIP_TRY(hipMemcpyAsync(&total_pairs, db->b3_pair_ofs + P, 4, hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        (void)hipFree(tmp); (void)hipFree(tmp_counts);
        db->b3_total_pairs = total_pairs;
        HIP_TRY(hipMalloc((void**)&db->b3_pair_blk, std::max<uint32_t>(total_pairs, 1)));
        HIP_TRY(hipMalloc((void**)&db->b3_pair_mask, (size_t)std::max<uint32_t>(total_pairs, 1) * 8));
        if (b3_launch_decode<false>(db, db->stream)) return 1;
        if (estimate_only) {
            // ranking the candidate widths only needs the number of records: the count mode of the all-nodes emit kernel
            HIP_TRY(hipMemcpy(db->b3_nl, h.nl->data(), P * 4, hipMemcpyHostToDevice));
            if (b3_launch_emit<false, false>(db, db->stream)) return 1;
            HIP_TRY(hipStreamSynchronize(db->stream));
            std::vector<uint32_t> counts(tbl);
            HIP_TRY(hipMemcpy(counts.data(), db->b2_table, tbl * 4, hipMemcpyDeviceToHost));
            uint64_t total = 0;
            for (uint32_t v : counts) total += v;
            if (total >= (1ull << 32)) return 0;
            db->b2_total = total;
            *fits = true;
            return 0;
        }
        if (getenv("KMDB_K1_SINGLE")) {
            HIP_TRY(hipMemcpy(db->b3_nl, h.nl->data(), P * 4, hipMemcpyHostToDevice));
            if (b3_launch_emit<false, false>(db, db->stream)) return 1;
        } else {
            // Classify the nodes with the count mode of the all-nodes emit kernel: how many blocks does the full list
            // touch?  (layout metadata, like the counts.)  Narrow nodes (one block) are handled in the DFS stream by
            // the narrow kernel; the wide ones get their own list: DFS index, parent position, slices, root paths.
            uint32_t* d_fw = nullptr;
            HIP_TRY(hipMalloc((void**)&d_fw, P * 4));
            HIP_TRY(hipMemcpy(db->b3_nl, h.nl->data(), P * 4, hipMemcpyHostToDevice));
            if (b3_launch_emit<false, false>(db, db->stream, 0, d_fw)) return 1;
            HIP_TRY(hipStreamSynchronize(db->stream));
            std::vector<uint32_t> fw(P);
            HIP_TRY(hipMemcpy(fw.data(), d_fw, P * 4, hipMemcpyDeviceToHost));
            (void)hipFree(d_fw);
            const std::vector<int32_t>& parent = *h.parent;
            std::vector<uint32_t> nlf(*h.nl), widx;
            std::vector<int32_t> rank(P, -1), wparent;
            std::vector<uint16_t> wd01(P, 0xFFFFu);
            for (uint64_t i = 0; i < P; ++i) {
                const uint32_t f = fw[i], f2 = f & (f - 1u);                   // f2: without the lowest block
                if (f2 & (f2 - 1u)) { rank[i] = (int32_t)widx.size(); widx.push_back((uint32_t)i); nlf[i] |= 1u << 31; }   // > 2 blocks
                else if (f) wd01[i] = (uint16_t)((uint32_t)__builtin_ctz(f) | (f2 ? (uint32_t)__builtin_ctz(f2) << 8 : 0xFF00u));
            }
            const size_t nW = widx.size();
            wparent.resize(nW);
            for (size_t k = 0; k < nW; ++k) {
                const int32_t pp = parent[widx[k]];
                if (pp < 0) wparent[k] = -1;
                else if (rank[pp] >= 0) wparent[k] = rank[pp];
                else { wparent[k] = -(pp + 2); nlf[pp] |= 1u << 30; }
            }
            size_t WSEG = 256;                                         // wide nodes per slice (4 batches)
            if (const char* e = getenv("KMDB_WSEG")) WSEG = std::max<size_t>(64, strtoull(e, nullptr, 10));
            const size_t n_wsegs = (nW + WSEG - 1) / WSEG;
            std::vector<Segment> wsegs(n_wsegs);
            // root paths inside the wide forest are short: the chain table of the wide launch is sized for them
            uint32_t wmaxd = 0;
            {
                std::vector<uint16_t> wd(nW, 0);
                for (size_t k = 0; k < nW; ++k) {
                    wd[k] = (uint16_t)(wparent[k] >= 0 ? wd[wparent[k]] + 1 : 1);
                    wmaxd = std::max<uint32_t>(wmaxd, wd[k]);
                }
            }
            db->b3_wchain_cap = std::min<uint32_t>(db->b3_chain_cap, std::max<uint32_t>(8, (wmaxd + 7) / 8 * 8));
            const size_t stride = db->b3_wchain_cap;
            std::vector<uint32_t> wanc(std::max<size_t>(n_wsegs, 1) * stride, 0), wanc_n(std::max<size_t>(n_wsegs, 1), 0);
            std::vector<int32_t> wnp(std::max<size_t>(n_wsegs, 1), -1);
            std::vector<uint32_t> path;
            for (size_t sg = 0; sg < n_wsegs; ++sg) {
                wsegs[sg] = Segment{(uint32_t)(sg * WSEG), (uint32_t)std::min(nW, (sg + 1) * WSEG)};
                path.clear();
                int32_t cur = parent[widx[sg * WSEG]];
                while (cur >= 0 && rank[cur] >= 0) { path.push_back((uint32_t)cur); cur = parent[cur]; }
                wnp[sg] = cur;                                         // narrow parent of the topmost wide ancestor (or none)
                wanc_n[sg] = (uint32_t)path.size();
                for (size_t t = 0; t < path.size(); ++t) wanc[sg * stride + t] = path[path.size() - 1 - t];
            }
            HIP_TRY(hipMemcpy(db->b3_nl, nlf.data(), P * 4, hipMemcpyHostToDevice));
            auto up = [&](auto** dst, const auto& v) -> int {
                using T = typename std::remove_reference<decltype(v[0])>::type;
                HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(v.size(), 1) * sizeof(T)));
                if (!v.empty()) HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
                return 0;
            };
            if (up(&db->b3_wd01, wd01) || up(&db->b3_widx, widx) || up(&db->b3_wparent, wparent) || up(&db->b3_wsegs, wsegs) || up(&db->b3_wseg_anc, wanc) ||
                up(&db->b3_wseg_anc_n, wanc_n) || up(&db->b3_wseg_np, wnp)) return 1;
            HIP_TRY(hipMalloc((void**)&db->b3_fnarrow, P * 16));
            db->b3_n_wide = (uint32_t)nW; db->b3_n_wsegs = (uint32_t)n_wsegs; db->b3_split = true;
            // count modes of the two run-time kernels, one table row per slice
            (void)hipFree(db->b2_table); db->b2_table = nullptr;
            table_rows = n_wsegs;
            tbl = std::max<size_t>(table_rows, 1) * db->b2_nctr;
            HIP_TRY(hipMalloc((void**)&db->b2_table, tbl * 4));
            HIP_TRY(hipMemset(db->b2_table, 0, tbl * 4));
            ntbl = (size_t)db->b3_n_nsegs * db->b2_nctr;
            HIP_TRY(hipMalloc((void**)&db->b3_ntable, std::max<size_t>(ntbl, 1) * 4));
            HIP_TRY(hipMemset(db->b3_ntable, 0, std::max<size_t>(ntbl, 1) * 4));
            if (b3_launch_narrow<false>(db, db->stream)) return 1;
            if (n_wsegs && b3_launch_emit<false, true>(db, db->stream)) return 1;
            if (getenv("KMDB_VERBOSE")) {
                uint64_t expect = 0, expect_wide = 0;                  // records the block masks imply: p (p + 1) / 2 per pattern with k-mers
                for (uint64_t i = 0; i < P; ++i) {
                    const uint32_t pw = (uint32_t)__builtin_popcount(fw[i]);
                    if (((*h.nl)[i] & 0xFFFFu) >= 2u && h.w && (*h.w)[i]) { expect += (uint64_t)pw * (pw + 1) / 2; if (pw > 2) expect_wide += (uint64_t)pw * (pw + 1) / 2; }
                }
                fprintf(stderr, "[kmdb] width %u: of which %llu from nodes with more than two blocks\n", width, (unsigned long long)expect_wide);
                fprintf(stderr, "[kmdb] width %u: %zu of %llu nodes are wide (%zu slices), %llu records expected\n", width, nW,
                        (unsigned long long)P, n_wsegs, (unsigned long long)expect);
            }
        }
    } else {
        if (b2_launch_emit<false>(db, 0, db->n_rsegs, 0, db->stream)) return 1;
    }
    HIP_TRY(hipStreamSynchronize(db->stream));
    std::vector<uint32_t> counts(tbl);
    HIP_TRY(hipMemcpy(counts.data(), db->b2_table, tbl * 4, hipMemcpyDeviceToHost));
    // bucket-major record order: all records of (bucket, class) c are contiguous, segment by segment
    std::vector<uint32_t> ncounts(ntbl), nbases(ntbl);
    if (ntbl) HIP_TRY(hipMemcpy(ncounts.data(), db->b3_ntable, ntbl * 4, hipMemcpyDeviceToHost));
    if (getenv("KMDB_VERBOSE") && ntbl) {
        uint64_t sn = 0, sw = 0;
        for (uint32_t v : ncounts) sn += v;
        for (uint32_t v : counts) sw += v;
        fprintf(stderr, "[kmdb] width %u: count mode: %llu records from the slim kernel, %llu from the wide list\n", width,
                (unsigned long long)sn, (unsigned long long)sw);
    }
    std::vector<uint32_t> bases(tbl);
    std::vector<uint64_t> cstart(db->b2_nctr + 1, 0);
    uint64_t run = 0;
    for (uint32_t c = 0; c < db->b2_nctr; ++c) {
        cstart[c] = run;
        for (size_t sgi = 0; ntbl && sgi < db->b3_n_nsegs; ++sgi) {        // the slices of the <= 2-block kernel first
            nbases[sgi * db->b2_nctr + c] = (uint32_t)run;
            run += ncounts[sgi * db->b2_nctr + c];
        }
        for (size_t sgi = 0; sgi < table_rows; ++sgi) {
            bases[(size_t)sgi * db->b2_nctr + c] = (uint32_t)run;
            run += counts[(size_t)sgi * db->b2_nctr + c];
        }
    }
    cstart[db->b2_nctr] = run;
    if (getenv("KMDB_VERBOSE")) {
        uint64_t per_cls[B2_NCLS] = {0}, diag_cls[B2_NCLS] = {0};
        for (uint32_t X = 0, c = 0; X < NB; ++X)
            for (uint32_t Y = 0; Y <= X; ++Y)
                for (uint32_t cls = 0; cls < B2_NCLS; ++cls, ++c) {
                    per_cls[cls] += cstart[c + 1] - cstart[c];
                    if (X == Y) diag_cls[cls] += cstart[c + 1] - cstart[c];
                }
        fprintf(stderr, "[kmdb] width %u: records per weight class %llu %llu %llu (on the diagonal %llu %llu %llu)\n", width,
                (unsigned long long)per_cls[0], (unsigned long long)per_cls[1], (unsigned long long)per_cls[2],
                (unsigned long long)diag_cls[0], (unsigned long long)diag_cls[1], (unsigned long long)diag_cls[2]);
    }
    if (run >= (1ull << 32)) return 0;                    // record index must fit 32 bits
    db->b2_total = run;
    std::vector<B2Item> items;
    uint32_t CH = 8192;
    if (const char* e = getenv("KMDB_K2_CHUNK")) CH = std::max<uint32_t>(256, (uint32_t)strtoul(e, nullptr, 10));
    // work items class by class (one launch of the apply kernel per weight class)
    // the dearest class first, in smaller chunks (a class-2 record costs several class-0 records)
    for (uint32_t cls = B2_NCLS; cls-- > 0;) {
        const uint32_t ch = cls == 0 ? CH : cls == 1 ? CH / 2 : CH / 8;
        for (uint32_t X = 0; X < NB; ++X)
            for (uint32_t Y = 0; Y <= X; ++Y) {
                const uint32_t c = (X * (X + 1) / 2 + Y) * B2_NCLS + cls;
                for (uint64_t b = cstart[c]; b < cstart[c + 1]; b += ch)
                    items.push_back({X, Y, cls, (uint32_t)b, (uint32_t)std::min<uint64_t>(b + ch, cstart[c + 1])});
            }
    }
    HIP_TRY(hipMemcpy(db->b2_table, bases.data(), tbl * 4, hipMemcpyHostToDevice));
    if (ntbl) HIP_TRY(hipMemcpy(db->b3_ntable, nbases.data(), ntbl * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&db->b2_rec_rows, std::max<uint64_t>(run, 1) * 8));
    HIP_TRY(hipMalloc((void**)&db->b2_rec_rc, std::max<uint64_t>(run, 1) * 16));
    HIP_TRY(hipMalloc((void**)&db->b2_rec_w, std::max<uint64_t>(run, 1) * 4));
    HIP_TRY(hipMalloc(&db->b2_items, std::max<size_t>(items.size(), 1) * sizeof(B2Item)));
    if (!items.empty()) HIP_TRY(hipMemcpy(db->b2_items, items.data(), items.size() * sizeof(B2Item), hipMemcpyHostToDevice));
    db->b2_n_items = (uint32_t)items.size();
    db->b2_ready = true;
    db->b3_ready = use_b3;
    *fits = true;
    return 0;
}

// Decide whether the database qualifies for the block-record pipeline and pick the block width:
// fewer sample ids per block than 64 pay off when the samples cluster (clades, species) in id ranges
// that a 64-id grid would cut in two.  The candidate with the fewest block records wins.
}  // namespace

int kmdb_records_prepare(kmdb_db* db, const kmdb_host_layout& h) {
    const uint64_t N = db->N, P = db->P;
    const uint32_t max_n = h.max_n;
    const bool chain_ok = h.chain_ok;
    if (N < 2 || P == 0 || N > 2048 || max_n > 1024) return 0;
    if (chain_ok) {
        const std::vector<uint32_t>&perm = *h.long_nodes, &nl = *h.nl, &seg_anc = *h.seg_anc, &seg_anc_n = *h.seg_anc_n;
        db->b3_n_long = (uint32_t)perm.size();
        HIP_TRY(hipMalloc((void**)&db->b3_perm, std::max<size_t>(perm.size(), 1) * 4));
        if (!perm.empty()) HIP_TRY(hipMemcpy(db->b3_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&db->b3_nl, P * 4));
        HIP_TRY(hipMemcpy(db->b3_nl, nl.data(), P * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&db->b3_seg_anc, std::max<size_t>(seg_anc.size(), 1) * 4));
        HIP_TRY(hipMalloc((void**)&db->b3_seg_anc_n, std::max<size_t>(seg_anc_n.size(), 1) * 4));
        if (!seg_anc.empty()) HIP_TRY(hipMemcpy(db->b3_seg_anc, seg_anc.data(), seg_anc.size() * 4, hipMemcpyHostToDevice));
        if (!seg_anc_n.empty()) HIP_TRY(hipMemcpy(db->b3_seg_anc_n, seg_anc_n.data(), seg_anc_n.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint8_t> depth8(P);
        for (uint64_t i = 0; i < P; ++i) depth8[i] = (uint8_t)std::min<uint32_t>(255u, (*h.depth)[i]);
        HIP_TRY(hipMalloc((void**)&db->b3_depth, P));
        HIP_TRY(hipMemcpy(db->b3_depth, depth8.data(), P, hipMemcpyHostToDevice));
        db->b3_n_nsegs = (uint32_t)h.nsegs->size();
        HIP_TRY(hipMalloc((void**)&db->b3_nsegs, std::max<size_t>(h.nsegs->size(), 1) * sizeof(Segment)));
        HIP_TRY(hipMalloc((void**)&db->b3_nseg_anc, std::max<size_t>(h.nseg_anc->size(), 1) * 4));
        HIP_TRY(hipMalloc((void**)&db->b3_nseg_anc_n, std::max<size_t>(h.nseg_anc_n->size(), 1) * 4));
        if (!h.nsegs->empty()) {
            HIP_TRY(hipMemcpy(db->b3_nsegs, h.nsegs->data(), h.nsegs->size() * sizeof(Segment), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(db->b3_nseg_anc, h.nseg_anc->data(), h.nseg_anc->size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(db->b3_nseg_anc_n, h.nseg_anc_n->data(), h.nseg_anc_n->size() * 4, hipMemcpyHostToDevice));
        }
        // K0 input, 12 bytes per node: {l | last id << 16, stream bits} and the stream position relative to the
        // first stream of the node's 256-node block
        std::vector<uint2> k0in(P);
        std::vector<uint32_t> bitrel(P);
        std::vector<uint64_t> blkbase((P + 255) / 256);
        for (uint64_t i = 0; i < P; ++i) {
            const uint4 m = (*h.meta)[i];
            if ((i & 255u) == 0) blkbase[i >> 8] = (*h.bitpos)[i];
            k0in[i] = make_uint2(m.y | (m.z << 16), m.w);
            bitrel[i] = (uint32_t)((*h.bitpos)[i] - blkbase[i >> 8]);
        }
        HIP_TRY(hipMalloc((void**)&db->b3_k0in, P * 8));
        HIP_TRY(hipMemcpy(db->b3_k0in, k0in.data(), P * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&db->b3_bitrel, P * 4));
        HIP_TRY(hipMemcpy(db->b3_bitrel, bitrel.data(), P * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&db->b3_blkbase, blkbase.size() * 8));
        HIP_TRY(hipMemcpy(db->b3_blkbase, blkbase.data(), blkbase.size() * 8, hipMemcpyHostToDevice));
    }
    uint32_t forced = 0;
    if (const char* e = getenv("KMDB_BLOCK_WIDTH")) forced = (uint32_t)strtoul(e, nullptr, 10);
    std::vector<uint32_t> cands = {64, 60, 56, 52, 50, 48, 44, 40, 36, 32};
    if (forced >= 8 && forced <= 64) cands = {forced};
    uint32_t best_w = 0;
    uint64_t best_cost = ~0ull;
    for (uint32_t wd : cands) {
        bool fits = false;
        if (cands.size() == 1) { best_w = wd; break; }
        if (b2_prepare_width(db, wd, h, &fits, /*estimate_only=*/true)) { b2_release_width(db); return 1; }
        if (fits) {
            // records dominate K1/K2; wider per-lane register sets (more blocks) make K1 a little dearer
            const uint64_t cost = db->b2_total + P * (db->b3_nbw > 16 ? (db->b3_nbw - 16) : 0) / 64;
            if (cost < best_cost) { best_cost = cost; best_w = wd; }
        }
        b2_release_width(db);
    }
    if (!best_w) return 0;
    bool fits = false;
    if (b2_prepare_width(db, best_w, h, &fits)) { b2_release_width(db); return 1; }
    if (!fits) b2_release_width(db);
    return 0;
}


void kmdb_records_release(kmdb_db* db) {
    b2_release_width(db);
    void* ptrs[] = {db->b3_perm, db->b3_seg_anc, db->b3_seg_anc_n, db->b3_nl, db->b3_depth, db->b3_k0in, db->b3_bitrel, db->b3_blkbase,
                    db->b3_nsegs, db->b3_nseg_anc, db->b3_nseg_anc_n};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    db->b3_perm = nullptr; db->b3_seg_anc = nullptr; db->b3_seg_anc_n = nullptr; db->b3_nl = nullptr; db->b3_depth = nullptr;
    db->b3_k0in = nullptr; db->b3_bitrel = nullptr; db->b3_blkbase = nullptr;
    db->b3_nsegs = nullptr; db->b3_nseg_anc = nullptr; db->b3_nseg_anc_n = nullptr;
}

uint64_t kmdb_records_device_bytes(const kmdb_db* db) {
    return db->b2_ready ? db->b2_total * 28 + (uint64_t)(db->b3_split ? db->b3_n_wsegs : db->n_rsegs) * db->b2_nctr * 4 + db->P * (db->b3_split ? 35 : 26) +
                              (uint64_t)db->b3_n_wide * 8 : 0;
}

int kmdb_records_run(kmdb_db* db, uint32_t* M, uint32_t flags, hipStream_t st) {
    const uint32_t dbg = flags >> 8;
    // emit block records (K0 + K1, or the sequential emit kernel), then ballot/popcount accumulate per block (K2)
    const bool seq_emit = !db->b3_ready;
    if (seq_emit) {
        if (b2_launch_emit<true>(db, 0, db->n_rsegs, dbg, st)) return 1;
        HIP_TRY(hipEventRecord(db->ev_k0, st));
        db->k0_ms = -1;
    } else {
        if (b3_launch_decode<false>(db, st)) return 1;
        HIP_TRY(hipEventRecord(db->ev_k0, st));
        if (db->b3_split) {
            if (!getenv("KMDB_SKIP_K1N") && b3_launch_narrow<true>(db, st, dbg)) return 1;
            if (db->b3_n_wsegs && !getenv("KMDB_SKIP_K1W") && b3_launch_emit<true, true>(db, st, dbg)) return 1;
        } else if (b3_launch_emit<true, false>(db, st, dbg)) return 1;
        db->k0_ms = 0;
    }
    HIP_TRY(hipEventRecord(db->ev_k2[0], st));
    if (db->b2_n_items && !(dbg & 2))
        hipLaunchKernelGGL(b2_apply_kernel, dim3(db->b2_n_items), dim3(256), 0, st,
                           B2Recs{db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w}, (const B2Item*)db->b2_items, M,
                           (uint32_t)db->N, dbg, db->b2_width);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(db->ev_k2[1], st));
    db->k1_ms = 0;
    return 0;
}
