// a2a_blocks.hip — the block-record pipeline of the dense all2all path (up to about 185 000 samples: 2^22 block pairs).
//
// Replaces SimilarityCalculator::all2all (reference src/similarity_calculator.cpp:42-438, inner loop row_add
// src/simd/row_add_avx2.cpp:30-124) with a design for gfx950.  The N x N matrix is cut into blocks (X, Y), X >= Y,
// of `width` <= 64 consecutive sample ids.  A pattern's ascending id list meets a block in a contiguous run, so
// all pair updates of a pattern with weight w factor into BLOCK RECORDS
//        (X, Y, rowmask, colmask, w):   M[width*X + r][width*Y + c] += w   for r in rowmask, c in colmask
//                                                                            (c < r when X == Y).
// Flat form (all2all_sp semantics, reference similarity_calculator.cpp:596-638): every pattern with w > 0 adds its
// on-disk w to all pairs of its FULL list, i.e. one record per pair of blocks X >= Y the full list touches.
//
// Everything that depends on a decoded sample id happens inside the call (kmdb_blocks_run):
//   K0  k0_decode_kernel x2  gamma streams -> the LOCAL ids of every node as (block, 64-bit mask) pairs (thread per node; the
//                            nodes with long streams in a second launch, most work first)
//   K1n k1n_kernel           walks the DFS stream in slices of 2048 nodes, 64 nodes per wave step, one lane each.  A full list is
//                            the union of the local lists on the root path and ids ascend along it, so the list of a node with
//                            at most TWO blocks is a summary of five registers; in-batch parents by pointer doubling across
//                            lanes, earlier ones from a chain table in LDS (one slot per depth).  Emits the records of those
//                            nodes (first-block diagonal records into per-block stream chunks, the others into the wide pool),
//                            flags the nodes with more blocks and leaves them their parent's summary.
//       wide list            the flagged nodes, compacted (popcount + scan + expand)
//   K1g k1g_kernel           one lane per wide node: list = parent's list (row in LDS, chain table, or a climb) + own pairs;
//                            the records of a batch are numbered by a prefix sum and emitted one per lane from a descriptor
//                            queue into the wide pool, in arrival order; heavy nodes by the whole wave.
//   sort cs_hist / cs_scatter  the wide pool (16-byte records + a key word: stream | weight digit) grouped by stream: counting
//                            sort with per-workgroup LDS histograms and staged tiles, one pass (<= 2048 streams) or two
//                            (by block row, then inside the rows); rocprim radix sort beyond 512 block rows.
//   K2  k2_apply_kernel      stream chunks (side stream, next to K1g), k2_sorted_kernel: the sorted wide pool.  64 records per
//                            wave step as bit matrices, int8 MFMA into a 64 x 64 LDS tile (weights >= 128 as base-128 digit
//                            passes), one HBM atomic per non-zero cell.
// Records are placed WITHOUT a counting pass and WITHOUT hot global atomics (same-address device atomics run at a few million
// per second on this part: measured, profiles/README.md): waves take chunks from 256 sub-pool cursors a few at a time, keep
// their open stream chunks in an LDS table indexed by block, and reserve slots for a group of lanes with a ballot.  The pools are
// sized from a sampled estimate; a pool that turns out too small is enlarged and the call repeated.
#include "device_common.h"
#include "engine_internal.h"

#include "prim.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

__host__ __device__ __forceinline__ uint32_t tri32(uint32_t x) { return x * (x + 1u) / 2u; }

// Workgroups are dealt to the eight XCDs round-robin (workgroup b runs on XCD b % 8), and every XCD has its own L2.  Where neighbouring
// work items write neighbouring memory (the sort kernels: consecutive jobs fill adjacent stretches of every stream, the cache lines at
// the seams are shared), item = xcd_contiguous(b, n) keeps every XCD on ONE contiguous range of items, so that a line is completed in
// one L2 instead of leaving two of them half written.  n must be a multiple of 8 (the launches round up).
constexpr uint32_t RS_XCD_MAP = 1;          // 0: items in workgroup order (A/B)
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t n) {
    if (!RS_XCD_MAP) return b;
    return (b & 7u) * (n >> 3) + (b >> 3);
}

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, WAVE);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, WAVE);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v, int d) {
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, WAVE);
    const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, WAVE);
    return ((unsigned long long)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------
// record pool
// ------------------------------------------------------------------------------------------
constexpr uint32_t CH_SHIFT = 8, CH_REC = 1u << CH_SHIFT;      // records per stream chunk: four wave steps of K2, 4 KB contiguous
constexpr uint32_t WCH_SHIFT = 6, WCH_REC = 1u << WCH_SHIFT;   // records per chunk of the wide pool (grouped by a sort, not by chunk)
// Record of the wide pool: 16 bytes of masks, moved by the sort next to its 4-byte key word.  The key word holds the stream in
// its low `kbits` bits, above it one base-2^dbits digit of the weight and the digit's index (2 bits): a weight of more than
// dbits bits becomes one record per non-zero digit (exact: the matrix is uint32 wrap-around arithmetic, shifted digits add up).
struct __attribute__((aligned(16))) WideRec { unsigned long long rows, cols; };
struct PoolView {
    uint32_t* counters;            // KCTR_*
    uint32_t* chunk_key;           // [pool_cap] stream of the chunk (n_states: never opened)
    uint32_t* chunk_fill;          // [pool_cap] records in the chunk
    unsigned char* rec;            // record slots, 16 bytes each
    uint32_t* recw;
    uint32_t* sub_cursor;          // [KMDB_SUBPOOLS * 16] chunks taken from every sub-pool
    uint32_t sub_cap;              // chunks per sub-pool
    uint32_t pool_cap;             // chunks
    // wide pool (dense mode): records in arrival order with their stream keys; a device-wide sort groups them afterwards
    uint32_t* wkey;                // [wide slots] stream of the record, 0xFFFFFFFF = never written
    WideRec* wrec;                 // [wide slots]
    uint32_t* wsub_cursor;         // [KMDB_SUBPOOLS * 16]
    uint32_t wsub_cap;             // chunks per sub-pool of the wide pool
    uint32_t dense;
    uint32_t kbits, dbits;         // key word of the wide pool: stream bits, weight digit bits                // 1: this kernel writes into the wide pool
    uint32_t row_mode;             // 1: many streams — the wide records go to per-block-row chunks of the chunk pool instead of the arrival-order
    uint32_t n_states;             // pool, so that only the sort inside the rows remains
    uint32_t pshift;               // row mode, != 0: packed records — the weight digit (and its index) sit in the column mask word from bit `pshift` (= the
                                   // block width) on, 64 - pshift - 2 bits of digit; the key word beside the record holds the stream only and stays behind in the sort
};
struct Resv { uint32_t base1, n1, base2; };   // slots [base1, base1 + n1) and [base2, ...) for the rest
__device__ __forceinline__ uint32_t resv_slot(const Resv& r, uint32_t rank) { return rank < r.n1 ? r.base1 + rank : r.base2 + (rank - r.n1); }

// A wave's private allocator.  Chunk ids come from one of KMDB_SUBPOOLS sub-pools (sub-pool p owns the ids p, p + SUBPOOLS,
// p + 2 SUBPOOLS, ...), ARENA_GRAB at a time: one device atomic per 1024 records and wave, spread over 256 cursors (a single
// cursor would serialise: same-address atomics run at a few million per second).  The open chunks live in an LDS table,
// direct-mapped by stream (entry = stream mod table size; with at most that many streams an open chunk is only ever
// closed full, otherwise a colliding stream evicts it).  An entry = {stream, chunk id << 6 | records used}; a chunk that
// fills up is closed at once, so `used` of an open entry is below 64.
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
constexpr uint32_t ST_MAX_BITS = 10;            // at most 1024 open chunks per wave
struct WaveArena {
    uint32_t* t_key;               // LDS [1 << tbits]; not there (direct) when every stream has its own entry
    uint32_t* t_slot;              // LDS [1 << tbits], KEY_NONE = no open chunk
    uint32_t tmask;
    uint32_t direct;
    uint32_t sub;                  // this wave's sub-pool
    uint32_t stock, next;          // wave-uniform: stream chunks left of the last grab, the next of them (index inside the sub-pool)
    uint32_t wstock, wnext, wsub;  // the same for the wide pool (its sub-pool changes with every grab)
    uint32_t dslot;                // wide pool: next slot of the wave's one open chunk (dopen == 0: none)
    uint32_t dopen;
};
constexpr uint32_t ARENA_GRAB = 4;         // stream chunks (256 records) per grab (16 per grab — a quarter of the returning device atomics, which wait for
                                           // the wave's record stores to drain — measured no faster at 10 000 samples: profiles/r05_j8)
constexpr uint32_t WIDE_GRAB = 4;          // wide-pool chunks (64 records) per grab: every wave leaves an unfinished grab behind, and the
                                           // sort reads all slots up to the busiest sub-pool's cursor (16 -> 4: -0.35 ms at the benchmark database)
__host__ __device__ inline uint32_t arena_table_bits(uint32_t n_states) {
    uint32_t b = 4;
    while (b < ST_MAX_BITS && (1u << b) < n_states) ++b;
    return b;
}
// the table is indexed by block: only the records (X, X) of a list's first block go through stream chunks
__host__ __device__ inline size_t arena_table_bytes(uint32_t tbits, uint32_t n_keys) {
    if (tbits == 0) return 16;
    return ((size_t)4 << tbits) * (n_keys <= (1u << tbits) ? 1 : 2);
}
__device__ __forceinline__ void arena_init(WaveArena& A, uint32_t* lds, uint32_t tbits, uint32_t n_states, uint32_t wave_id, uint32_t lane) {
    A.direct = n_states <= (1u << tbits) ? 1u : 0u;
    A.t_slot = lds; A.t_key = lds + (1u << tbits); A.tmask = (1u << tbits) - 1u;
    A.sub = wave_id % KMDB_SUBPOOLS; A.wsub = A.sub; A.stock = 0; A.next = 0; A.wstock = 0; A.wnext = 0; A.dslot = 0; A.dopen = 0;
    if (tbits == 0) return;                                   // dense mode: no table
    for (uint32_t e = lane; e <= A.tmask; e += WAVE) { A.t_slot[e] = KEY_NONE; if (!A.direct) A.t_key[e] = KEY_NONE; }
    lds_sync();
}
__device__ __forceinline__ uint32_t arena_take(WaveArena& A, const PoolView& pv, uint32_t s, uint32_t lane) {
    if (A.stock == 0) {
        uint32_t base = 0;
        A.sub = (A.sub + 61u) % KMDB_SUBPOOLS;                        // every grab from another sub-pool: a wave with much output does not drain one
        if (lane == 0) base = atomicAdd(&pv.sub_cursor[A.sub * 16u], ARENA_GRAB);
        base = bcast(base, 0);
        if (base + ARENA_GRAB > pv.sub_cap) {                        // stays in range; the call is repeated with a larger pool
            if (lane == 0) atomicOr(&pv.counters[KCTR_POOL_OVERFLOW], 1u);
            base = pv.sub_cap - ARENA_GRAB;
        }
        A.next = base; A.stock = ARENA_GRAB;
    }
    const uint32_t id = A.next * KMDB_SUBPOOLS + A.sub;
    ++A.next; --A.stock;
    if (lane == 0) pv.chunk_key[id] = s;
    return id;
}
// next chunk of the wide pool
__device__ __forceinline__ uint32_t arena_take_wide(WaveArena& A, const PoolView& pv, uint32_t lane) {
    if (A.wstock == 0) {
        uint32_t base = 0;
        A.wsub = (A.wsub + 61u) % KMDB_SUBPOOLS;                     // every grab from another sub-pool: waves with much output do not drain one
        if (lane == 0) base = atomicAdd(&pv.wsub_cursor[A.wsub * 16u], WIDE_GRAB);
        base = bcast(base, 0);
        if (base + WIDE_GRAB > pv.wsub_cap) {                     // stays in range; the call is repeated with a larger pool
            if (lane == 0) atomicOr(&pv.counters[KCTR_WIDE_OVERFLOW], 1u);
            base = pv.wsub_cap - WIDE_GRAB;
        }
        A.wnext = base; A.wstock = WIDE_GRAB;
    }
    const uint32_t id = A.wnext * KMDB_SUBPOOLS + A.wsub;
    ++A.wnext; --A.wstock;
    return id;
}
// cnt (1..64) slots of stream s; s and cnt wave-uniform, every lane of the wave calls
// cnt (1..64) slots of the wide pool: records of all streams share the wave's one open chunk, a device-wide sort groups them
__device__ __forceinline__ Resv arena_reserve_wide(WaveArena& A, const PoolView& pv, uint32_t cnt, uint32_t lane) {
    if (!A.dopen) { A.dslot = arena_take_wide(A, pv, lane) << WCH_SHIFT; A.dopen = 1; }
    const uint32_t v = A.dslot, rem = WCH_REC - (v & (WCH_REC - 1u));
    if (cnt < rem) { A.dslot = v + cnt; return Resv{v, cnt, 0u}; }
    if (cnt == rem) { A.dopen = 0; return Resv{v, cnt, 0u}; }
    const uint32_t nv = arena_take_wide(A, pv, lane) << WCH_SHIFT;
    A.dslot = nv + (cnt - rem);
    return Resv{v, rem, nv};
}
// cnt (1..64) slots of stream s, whose open chunk lives in table entry `ent`; all arguments wave-uniform, every lane calls
__device__ __forceinline__ Resv arena_reserve(WaveArena& A, const PoolView& pv, uint32_t ent, uint32_t s, uint32_t cnt, uint32_t lane) {
    const uint32_t e = ent & A.tmask;
    uint32_t v = A.t_slot[e];
    if (A.direct) {
        if (v == KEY_NONE) v = arena_take(A, pv, s, lane) << CH_SHIFT;
    } else {
        const uint32_t key = A.t_key[e];
        if (key != s) {
            if (key != KEY_NONE && lane == 0) pv.chunk_fill[v >> CH_SHIFT] = v & (CH_REC - 1u);        // evicted partly filled
            v = arena_take(A, pv, s, lane) << CH_SHIFT;
        }
    }
    const uint32_t rem = CH_REC - (v & (CH_REC - 1u));
    Resv r{v, cnt, 0u};
    uint32_t nkey = s, nv = v + cnt;
    if (cnt >= rem) {
        if (lane == 0) pv.chunk_fill[v >> CH_SHIFT] = CH_REC;                                      // full
        if (cnt == rem) { nkey = KEY_NONE; nv = KEY_NONE; }
        else {
            nv = arena_take(A, pv, s, lane) << CH_SHIFT;
            r = Resv{v, rem, nv};
            nv += cnt - rem;
        }
    }
    if (lane == 0) { A.t_slot[e] = nv; if (!A.direct) A.t_key[e] = nkey; }
    lds_sync();
    return r;
}
__device__ __forceinline__ void arena_finish(const WaveArena& A, const PoolView& pv, uint32_t lane) {
    if (A.tmask == 0) return;                                   // no stream table (never-written slots of the wide pool keep their 0xFFFFFFFF key)
    for (uint32_t e = lane; e <= A.tmask; e += WAVE) {
        const uint32_t v = A.t_slot[e];
        if (v != KEY_NONE && (A.direct || A.t_key[e] != KEY_NONE)) pv.chunk_fill[v >> CH_SHIFT] = v & (CH_REC - 1u);
    }
}

// diagonal streams (X == Y, cols == rows) pack 8-byte rows into the first half of their chunks
// the records `on` of a wave step into the wide pool: one reservation for all lanes per weight digit (nearly always one digit).
// Every lane of the wave calls.
__device__ __forceinline__ void wide_emit(WaveArena& A, const PoolView& pv, bool on, unsigned long long rows, unsigned long long cols, uint32_t w, uint32_t stream,
                                          uint32_t lane, unsigned long long lt_mask) {
    // (packed records, pv.pshift != 0: the weight digit and its index in the spare bits of the column word, the key word = the stream only)
    const uint32_t dbits = pv.pshift ? (64u - pv.pshift - 2u < 16u ? 64u - pv.pshift - 2u : 16u) : pv.dbits;
    const uint32_t dmask = (1u << dbits) - 1u;
    uint32_t j = 0;
    for (;;) {
        const unsigned long long grp = __ballot(on);
        if (!grp) break;
        const Resv r = arena_reserve_wide(A, pv, (uint32_t)__popcll(grp), lane);
        if (on) {
            const uint32_t slot = resv_slot(r, (uint32_t)__popcll(grp & lt_mask));
            if (pv.pshift) {
                pv.wrec[slot] = WideRec{rows, cols | ((unsigned long long)((w & dmask) | (j << dbits)) << pv.pshift)};
                pv.wkey[slot] = stream;
            } else {
                pv.wrec[slot] = WideRec{rows, cols};
                pv.wkey[slot] = stream | (((w & dmask) | (j << dbits)) << pv.kbits);
            }
        }
        w >>= dbits; ++j;
        on = on && w != 0;
    }
}
// weight of a sorted wide record from its key word
__device__ __forceinline__ uint32_t wide_weight(uint32_t key, uint32_t kbits, uint32_t dbits) {
    const uint32_t f = key >> kbits, dig = f & ((1u << dbits) - 1u), j = f >> dbits;
    const uint32_t sh = j * dbits;
    return sh < 32u ? dig << sh : 0u;
}
// ---- many streams (row mode): the wide records of block row X go to chunks of their own (CH_REC records, key n_states + X), so they
// arrive grouped by row and only the sort inside the rows is left.  A wave keeps, per row, the next free slot of its open chunk and
// the chunk's end in LDS; every lane reserves its slot with ONE LDS atomic (lanes that follow one another with the same row share
// one): no loop over the rows of a step.  A reservation that runs past the chunk's end is the rare case (once per CH_REC records
// and row): the row gets fresh chunks and the lanes beyond the end are renumbered.
struct RowTab { uint32_t* pos; uint32_t* end; uint32_t n; };
__host__ __device__ inline size_t rowtab_bytes(uint32_t n_rows) { return ((size_t)8 * n_rows + 15) & ~(size_t)15; }
__device__ __forceinline__ void rowtab_init(RowTab& R, uint32_t* lds, uint32_t n, uint32_t lane) {
    R.pos = lds; R.end = lds + n; R.n = n;
    for (uint32_t e = lane; e < 2u * n; e += WAVE) lds[e] = 0u;         // end == 0: no open chunk
    lds_sync();
}
__device__ __forceinline__ void row_emit(WaveArena& A, const PoolView& pv, const RowTab& R, bool on, uint32_t X, unsigned long long rows, unsigned long long cols,
                                         uint32_t w, uint32_t stream, uint32_t lane, unsigned long long lt_mask) {
    const uint32_t dbits = pv.pshift ? (64u - pv.pshift - 2u < 16u ? 64u - pv.pshift - 2u : 16u) : pv.dbits;
    const uint32_t dmask = (1u << dbits) - 1u;
    uint32_t j = 0;
    for (;;) {
        const unsigned long long onm = __ballot(on);
        if (!onm) break;
        // runs of consecutive lanes with the same row: the first lane of a run reserves for all of it (one LDS atomic per LANE instead —
        // the lanes of a row then collide in the LDS unit, the ~50 VALU instructions below are gone — measured no faster: profiles/r05_j5)
        const uint32_t pX = (uint32_t)__shfl_up((int)X, 1, WAVE);
        const bool lead = on && (lane == 0u || !((onm >> (lane - 1u)) & 1ull) || pX != X);
        const unsigned long long leaders = __ballot(lead);
        uint32_t ll = lane, cnt = 0;
        if (on) {
            ll = 63u - (uint32_t)__builtin_clzll(leaders & (lt_mask | (1ull << lane)));
            const unsigned long long above = ll == 63u ? 0ull : ~((2ull << ll) - 1ull);
            const unsigned long long after = (leaders | ~onm) & above;
            cnt = (after ? (uint32_t)__builtin_ctzll(after) : 64u) - ll;
        }
        uint32_t old = 0, end = 0;
        if (lead) { end = R.end[X]; old = atomicAdd(&R.pos[X], cnt); }
        uint32_t p = (uint32_t)__shfl((int)old, (int)ll, WAVE) + (lane - ll);
        const uint32_t e = (uint32_t)__shfl((int)end, (int)ll, WAVE);
        bool ovf = on && p >= e;
        unsigned long long pend = __ballot(ovf);
        while (pend) {
            const uint32_t l0 = (uint32_t)__builtin_ctzll(pend);
            const uint32_t X0 = bcast(X, l0), e0 = bcast(e, l0);
            lds_sync();
            const uint32_t tot = R.pos[X0];                        // after every lane's reservation
            const uint32_t over = tot - e0;                        // slots needed beyond the open chunk (no open chunk: e0 == 0, positions count from 0)
            const uint32_t nnew = (over + CH_REC - 1u) >> CH_SHIFT;
            if (lane == 0 && e0) pv.chunk_fill[(e0 - 1u) >> CH_SHIFT] = CH_REC;
            const bool mine = ovf && X == X0;
            const uint32_t qv = p - e0;
            uint32_t cl = 0;
            for (uint32_t i = 0; i < nnew; ++i) {
                cl = arena_take(A, pv, pv.n_states + X0, lane);        // key of a row chunk: n_states + row
                if (lane == 0 && i + 1u < nnew) pv.chunk_fill[cl] = CH_REC;
                if (mine && (qv >> CH_SHIFT) == i) p = (cl << CH_SHIFT) | (qv & (CH_REC - 1u));
            }
            if (lane == 0) { R.pos[X0] = (cl << CH_SHIFT) + (over - ((nnew - 1u) << CH_SHIFT)); R.end[X0] = (cl << CH_SHIFT) + CH_REC; }
            lds_sync();
            ovf = ovf && !mine;
            pend = __ballot(ovf);
        }
        if (on) {
            if (pv.pshift) {
                ((WideRec*)pv.rec)[p] = WideRec{rows, cols | ((unsigned long long)((w & dmask) | (j << dbits)) << pv.pshift)};
                pv.recw[p] = stream;
            } else {
                ((WideRec*)pv.rec)[p] = WideRec{rows, cols};
                pv.recw[p] = stream | (((w & dmask) | (j << dbits)) << pv.kbits);
            }
        }
        w >>= dbits; ++j;
        on = on && w != 0;
    }
}
__device__ __forceinline__ void rowtab_finish(const RowTab& R, const PoolView& pv, uint32_t lane) {
    lds_sync();
    for (uint32_t X = lane; X < R.n; X += WAVE) {
        const uint32_t e = R.end[X];
        if (e) pv.chunk_fill[(e - 1u) >> CH_SHIFT] = CH_REC - (e - R.pos[X]);
    }
}
// stream chunks hold the records (X, X, rows) of one block X: 8-byte rows in the first half of the chunk, weights beside
__device__ __forceinline__ void rec_store_diag(const PoolView& pv, uint32_t slot, unsigned long long rows, uint32_t w) {
    const uint32_t ch = slot >> CH_SHIFT, r = slot & (CH_REC - 1u);
    ((unsigned long long*)(pv.rec + ((size_t)ch << (CH_SHIFT + 4))))[r] = rows;
    pv.recw[slot] = w;
}

__global__ void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void iota_u32_kernel(uint32_t* __restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// ------------------------------------------------------------------------------------------
// K0: gamma streams -> local (block, mask) pairs
// ------------------------------------------------------------------------------------------
struct K0Params {
    const uint2* k0in;
    const uint32_t* bitrel;
    const uint64_t* blkbase;
    const uint64_t* bits;
    const uint32_t* perm;          // nullptr: all nodes in DFS order, long ones skipped; else: the long nodes
    uint32_t P;                    // nodes of this launch
    uint32_t short_ids;            // lists of more ids belong to the long launch (kmdb_db.short_max_ids)
    BlockMap bm;
    unsigned long long* p0_mask;   // first pair inline
    uint32_t* p0_info;             // block | npairs << 16
    uint32_t* pair_ofs;            // further pairs: first entry in the pair pool
    uint16_t* pair_blk;
    unsigned long long* pair_mask;
    uint32_t* pair_cursor;         // [KMDB_PAIR_REGIONS * 16]
    uint32_t n_regions, region_cap, spill_cap;   // sub-pools of region_cap entries each, then a shared spill area
    uint32_t* counters;
};

constexpr int K0_RELW = 8;                 // non-empty 64-id words of a long list kept in registers (more: the list is walked twice)
constexpr int K0_SINGLE_PASS = 1;          // 0: every list spanning 64 ids or more is walked twice, as in round 3 (A/B)
template <bool LONG>
__global__ __launch_bounds__(256) void k0_decode_kernel(const K0Params q) {
    // two launches cover the nodes: perm == nullptr walks ALL nodes in DFS order (coalesced) and skips the ones whose
    // stream does not fit three registers; those few are listed in perm, most work first, and decoded by the second
    // launch so that no wave waits on one long stream
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t i = 0;
    bool live;
    if (q.perm) {
        live = t < q.P;
        if (live) i = q.perm[t];
    } else {
        // DFS-order launch: the 256 nodes of the block are re-dealt to its threads by decreasing amount of work
        // (counting sort in LDS), so every wave runs decode loops of similar length while all global accesses
        // of the block stay inside its own 256-node window
        __shared__ uint32_t bins[64];
        __shared__ uint16_t order[256];
        if (threadIdx.x < 64) bins[threadIdx.x] = 0;
        __syncthreads();
        const uint2 kt = t < q.P ? q.k0in[t] : make_uint2(0u, 0u);          // l, last id, stream bits (kmdb_k0_pack)
        const uint32_t tl = kmdb_k0_l(kt), tbits = kmdb_k0_bits(kt);
        // work of a node ~ number of codes that are not "0" ~ stream bits beyond one per delta
        uint32_t key = 0;                                                    // 0: nothing to decode here
        if (t < q.P && tl > 1 && !kmdb_long_node(tl, tbits, q.short_ids)) {
            key = 1u + (tbits - (tl - 1u));
            key = key > 63u ? 63u : key;
        }
        atomicAdd(&bins[63u - key], 1u);                                     // bin 0 = most work
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t c = bins[threadIdx.x];
            bins[threadIdx.x] = wave_incl_scan(c, threadIdx.x) - c;
        }
        __syncthreads();
        order[atomicAdd(&bins[63u - key], 1u)] = (uint16_t)threadIdx.x;
        __syncthreads();
        i = blockIdx.x * blockDim.x + order[threadIdx.x];
        live = i < q.P;
    }
    const BlockMap bm = q.bm;
    const uint2 km = live ? q.k0in[i] : make_uint2(0u, 0u);
    const uint32_t l = kmdb_k0_l(km), last = kmdb_k0_last(km), nbits = kmdb_k0_bits(km);
    if (!q.perm && kmdb_long_node(l, nbits, q.short_ids)) live = false;
    using Cursor = RunCursor32<LONG ? 16 : 6, LONG>;
    uint32_t npairs = 0, blk0 = 0, need = 0, span = 0, bit0 = 0;
    unsigned long long mask0 = 0, m1 = 0, m2 = 0;
    bool second_pass = false, from_words = false;
    uint64_t pos = 0;
    // (long launch) the non-empty 64-id words of the relative list: K0_RELW of them in registers, filled in order
    unsigned long long rel_w[K0_RELW];
    uint32_t rel_i[K0_RELW], rel_n = 0;
#pragma unroll
    for (int e = 0; e < K0_RELW; ++e) { rel_w[e] = 0; rel_i[e] = 0; }
    auto rel_push = [&](uint32_t idx, unsigned long long w) {
#pragma unroll
        for (int e = 0; e < K0_RELW; ++e) if (rel_n == (uint32_t)e) { rel_w[e] = w; rel_i[e] = idx; }     // (static register indices: no scratch)
        ++rel_n;                                                  // beyond K0_RELW: the list takes the second walk after all
    };
    // the words cut at the block boundaries: (block, mask) pairs in ascending order.  WRITE = false counts them, true stores them
    // (the first one inline, the others in the pair pool from `o` on)
    auto rel_blocks = [&](bool write, uint32_t id0, uint32_t o) -> uint32_t {
        const unsigned long long wm = bm.width == 64 ? ~0ull : (1ull << bm.width) - 1ull;
        uint32_t curblk = 0xFFFFFFFFu, cnt = 0;
        unsigned long long cacc = 0;
        auto flushb = [&]() {
            if (curblk == 0xFFFFFFFFu) return;
            if (write) { if (cnt == 0) mask0 = cacc; else { q.pair_blk[o] = (uint16_t)curblk; q.pair_mask[o] = cacc; ++o; } }
            ++cnt;
        };
#pragma unroll
        for (int e = 0; e < K0_RELW; ++e) {
            if ((uint32_t)e < rel_n) {
                const uint32_t base = id0 + 64u * rel_i[e];
                const unsigned long long word = rel_w[e];
                const uint32_t b = bm.blk(base);
                uint32_t bit = bm.bit(base, b), sh = 0;
#pragma unroll
                for (uint32_t part = 0; part < 3; ++part) {       // 64 ids meet at most three blocks (width >= 32)
                    if (sh < 64u) {
                        const unsigned long long m = ((word >> sh) << bit) & wm;
                        if (m) {
                            if (b + part != curblk) { flushb(); curblk = b + part; cacc = 0; }
                            cacc |= m;
                        }
                        sh += bm.width - bit; bit = 0;
                    }
                }
            }
        }
        flushb();
        return cnt;
    };
    if (live && l == 1) {
        blk0 = bm.blk(last); mask0 = 1ull << bm.bit(last, blk0); npairs = 1;
    } else if (live && l) {
        // Pass 1 walks the stream run by run and builds the list RELATIVE to its (still unknown) first id:
        // bit k of R <=> id_0 + k is in the list.  pattern_t::decodeSamples (reference src/pattern.cpp:99-109)
        // gets id_0 the same way: last id minus the sum of the deltas.
        pos = q.blkbase[i >> 8] + q.bitrel[i];
        unsigned long long R = 1ull;
        {
            Cursor c(q.bits, pos);
            uint32_t rem = l - 1;
            // the long launch keeps the WHOLE relative list, not only its first 64 ids: the non-empty 64-id words (word index, mask)
            // in registers, in ascending order (ids ascend, so a word is complete when the next one starts).  A list that spans 64 ids or
            // more — four fifths of the long nodes of a 10 000-sample collection (profiles/r04_record_stats.py) — then needs no second
            // walk over its stream: its words are cut at the block boundaries once the first id is known.
            uint32_t wk = 0;
            unsigned long long acc = 1ull;
            auto add = [&](uint32_t s0, uint32_t cnt) {               // relative ids [s0, s0 + cnt), cnt >= 1
                while (cnt) {
                    const uint32_t k = s0 >> 6, o = s0 & 63u;
                    const uint32_t take = cnt < 64u - o ? cnt : 64u - o;
                    if (k != wk) { rel_push(wk, acc); wk = k; acc = 0; }
                    acc |= (take == 64u ? ~0ull : ((1ull << take) - 1ull)) << o;
                    s0 += take; cnt -= take;
                }
            };
            while (rem) {
                uint32_t z, v;
                c.step(rem, z, v);                                     // a run of z consecutive ids, then a gap of v
                if (z) {
                    if (LONG) add(span + 1u, z);
                    else if (span + z < 64u) R |= ((2ull << (z - 1)) - 1ull) << (span + 1);
                    span += z; rem -= z;
                }
                if (v) {
                    span += v; --rem;
                    if (LONG) add(span, 1u);
                    else if (span < 64u) R |= 1ull << span;
                }
            }
            if (LONG) { if (span < 64u) R = acc; else rel_push(wk, acc); }
        }
        const uint32_t id0 = last - span;
        blk0 = bm.blk(id0);
        bit0 = bm.bit(id0, blk0);
        const unsigned long long wm = bm.width == 64 ? ~0ull : (1ull << bm.width) - 1ull;
        if (span < 64u) {
            // the whole list fits the relative mask: cut it at the block boundaries (at most 3 blocks: width >= 32)
            const unsigned long long lo = R << bit0, hi = bit0 ? R >> (64u - bit0) : 0ull;
            auto ext = [&](uint32_t sh) -> unsigned long long {
                return sh == 0 ? lo : sh < 64u ? ((lo >> sh) | (hi << (64u - sh))) : sh == 64u ? hi : sh < 128u ? (hi >> (sh - 64u)) : 0ull;
            };
            mask0 = lo & wm;
            m1 = ext(bm.width) & wm; m2 = ext(2 * bm.width) & wm;
            npairs = 1u + (m1 != 0) + (m2 != 0);
            need = npairs - 1u;
        } else if (LONG && K0_SINGLE_PASS && rel_n <= (uint32_t)K0_RELW) {
            // wide list, all its words at hand: the pairs are counted now (exactly) and stored after the reservation
            from_words = true;
            npairs = rel_blocks(false, id0, 0u);
            need = npairs - 1u;
        } else {
            // wide list: second pass with absolute ids; the blocks it can touch bound the reservation
            second_pass = true;
            need = bm.blk(last) - blk0;
            need = need < l - 1u ? need : l - 1u;                 // l ids touch at most l blocks
        }
    }
    // extra pairs: one reservation per wave in the wave's region of the pair pool
    uint32_t out = 0;
    if (__ballot(need != 0u)) {                                  // (most waves of the short launch hold no list with a second block: no scan for them)
        const uint32_t incl = wave_incl_scan(need, lane);
        const uint32_t total = bcast(incl, WAVE - 1);
        if (total) {
            const uint32_t region = (blockIdx.x * 4u + (threadIdx.x >> 6)) % q.n_regions;
            uint32_t base = 0;
            if (lane == WAVE - 1) base = atomicAdd(&q.pair_cursor[region * 16u], total);
            base = bcast(base, WAVE - 1);
            if (base + total <= q.region_cap) out = region * q.region_cap + base + (incl - need);
            else {
                // the wave's sub-pool is full: the shared spill area behind the sub-pools (one cursor, rarely used)
                uint32_t sb = 0;
                if (lane == WAVE - 1) sb = atomicAdd(&q.pair_cursor[q.n_regions * 16u], total);
                sb = bcast(sb, WAVE - 1);
                if (sb + total <= q.spill_cap) out = q.n_regions * q.region_cap + sb + (incl - need);
                else {
                    if (lane == 0) atomicOr(&q.counters[KCTR_PAIR_OVERFLOW], 1u);
                    need = 0; second_pass = false; from_words = false; npairs = npairs ? 1u : 0u; m1 = m2 = 0;    // results invalid; the call is repeated with a larger pool
                }
            }
        }
    }
    if (!live) return;
    if (from_words) {
        (void)rel_blocks(true, last - span, out);
    } else if (!second_pass) {
        if (m1) { q.pair_blk[out] = (uint16_t)(blk0 + 1); q.pair_mask[out] = m1; }
        if (m2) { const uint32_t o2 = out + (m1 != 0); q.pair_blk[o2] = (uint16_t)(blk0 + 2); q.pair_mask[o2] = m2; }
    } else {
        Cursor c(q.bits, pos);
        uint32_t o = out;
        uint32_t curblk = blk0, bit = bit0, rem = l - 1;
        unsigned long long acc = 1ull << bit0;
        auto flush = [&]() {
            if (npairs == 0) mask0 = acc;
            else { q.pair_blk[o] = (uint16_t)curblk; q.pair_mask[o] = acc; ++o; }
            ++npairs;
        };
        while (rem) {
            uint32_t z, v;
            c.step(rem, z, v);
            if (z) {
                rem -= z;
                while (z) {
                    const uint32_t room = bm.width - 1u - bit;
                    const uint32_t tk = z < room ? z : room;
                    if (tk) { acc |= ((2ull << (tk - 1)) - 1ull) << (bit + 1); bit += tk; z -= tk; }
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
    q.p0_mask[i] = mask0;
    q.p0_info[i] = blk0 | (npairs << 16);
    if (npairs > 1) q.pair_ofs[i] = out;
}

// ------------------------------------------------------------------------------------------
// bit transposes and operand types of the matrix-core step (K1n's direct mode, K2, the second level)
// ------------------------------------------------------------------------------------------
// 64 x 64 bit-matrix transpose across the lanes of a wave: lane i holds row i on entry, column i on exit.  Six butterfly stages; stage
// 32 is one v_permlane32_swap over the two words.  Every stage below works on the 32-bit words separately: fetch the partner lane's
// word (lane ^ s), rotate it by s towards this lane's side (v_alignbit_b32: right for the upper lane of a pair, left for the lower one)
// and splice it in under a per-lane mask (v_bfi_b32) — three instructions per word and stage, with the rotate amounts and masks of
// the five stages in ten registers that are computed once per run (TrConst).  The fetches are v_permlane16_swap (stage 16) and DPP row /
// quad permutes: 35 VALU instructions per transpose, no LDS (fetching the stages 16 / 8 / 4 over the LDS crossbar with ds_swizzle_b32
// was measured slower — profiles/r05_experiment_switches.diff; round 3's version, v_perm + ds_bpermute shuffles and 64-bit mask
// arithmetic, took 72 VALU).  profiles/r04_transpose_probe.hip checks the variants against the definition and times them.
struct TrConst { uint32_t amt[5], msk[5]; };
__device__ __forceinline__ TrConst tr_const(uint32_t lane) {
    TrConst c;
    const uint32_t m[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t s = 16u >> k;
        const bool up = (lane & s) != 0;
        c.amt[k] = up ? s : 32u - s;          // the partner's word rotated right by s (upper lane of the pair) or left by s (lower lane)
        c.msk[k] = up ? m[k] : ~m[k];         // the bits taken from the partner
    }
    return c;
}
__device__ __forceinline__ uint32_t tr_fetch(uint32_t v, int k, bool up16) {
    // the word of lane ^ (16 >> k)
    if (k == 0) {
        const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);      // [0]: odd rows <- the even rows below them, [1]: even rows <- the odd rows above
        return up16 ? a[0] : a[1];
    }
    if (k == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false);   // row_ror:8
    if (k == 2) {
        const int a = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);      // row_shl:4 into the banks 0 and 2: lane i <- i + 4
        return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)v, 0x114, 0xF, 0xA, false);        // row_shr:4 into the banks 1 and 3: lane i <- i - 4
    }
    if (k == 3) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);               // quad_perm [1,0,3,2]
}
__device__ __forceinline__ unsigned long long transpose64(unsigned long long x, const TrConst& c) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    }
    const bool up16 = c.msk[0] == 0x0000FFFFu;                         // (lane & 16) != 0
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t pl = tr_fetch(lo, k, up16), ph = tr_fetch(hi, k, up16);
        const uint32_t rl = __builtin_amdgcn_alignbit(pl, pl, c.amt[k]), rh = __builtin_amdgcn_alignbit(ph, ph, c.amt[k]);
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(lo) : "v"(c.msk[k]), "v"(rl));      // (mask & partner) | (~mask & own)
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(hi) : "v"(c.msk[k]), "v"(rh));
    }
    return ((unsigned long long)hi << 32) | lo;
}

typedef int k2_v4i __attribute__((ext_vector_type(4)));
typedef int k2_v16i __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// K1n: the DFS stream, nodes with at most two blocks
// ------------------------------------------------------------------------------------------
// Summary of a stretch of a root path: the first block and its mask, the second block and its mask, and whether
// there is a third one.  Ids ascend along a root path, so "the ancestors' stretch" always holds the smaller blocks.
constexpr uint32_t BNONE = 0x7FFFu;
struct NSum { unsigned long long m0, m1; uint32_t bw; };        // bw = b0 | b1 << 15 | wide << 30

__device__ __forceinline__ NSum nsum_make(uint32_t b0, unsigned long long m0, uint32_t b1, unsigned long long m1, bool wide) {
    return NSum{m0, m1, b0 | (b1 << 15) | ((wide ? 1u : 0u) << 30)};
}
// T = the stretch closer to the root, S = the stretch below it
__device__ __forceinline__ NSum nsum_merge(const NSum& T, const NSum& S) {
    const uint32_t Tb0 = T.bw & 0x7FFFu, Tb1 = (T.bw >> 15) & 0x7FFFu, Sb0 = S.bw & 0x7FFFu, Sb1 = (S.bw >> 15) & 0x7FFFu;
    bool wide = (((T.bw | S.bw) >> 30) & 1u) != 0;
    if (Tb0 == BNONE) return S;
    if (Sb0 == BNONE) return T;
    if (Sb0 == Tb0) return nsum_make(Sb0, S.m0 | T.m0, Sb1, S.m1, wide || Tb1 != BNONE);
    if (Tb1 == BNONE) return nsum_make(Tb0, T.m0, Sb0, S.m0, wide || Sb1 != BNONE);
    if (Tb1 == Sb0) return nsum_make(Tb0, T.m0, Sb0, T.m1 | S.m0, wide || Sb1 != BNONE);
    return nsum_make(Tb0, T.m0, Tb1, T.m1, true);
}
__device__ __forceinline__ NSum nsum_shfl(const NSum& s, int src) {
    return NSum{shfl64(s.m0, src), shfl64(s.m1, src), (uint32_t)__shfl((int)s.bw, src, WAVE)};
}
__device__ __forceinline__ NSum nsum_shfl_up(const NSum& s, int d) {
    return NSum{shfl_up64(s.m0, d), shfl_up64(s.m1, d), (uint32_t)__shfl_up((int)s.bw, d, WAVE)};
}

struct NParams {
    const uint32_t* nl;            // n
    const int32_t* parent;
    const uint32_t* w;
    const uint16_t* dflag;         // depth | has-child << 15
    const uint32_t* seg_anc;       // [n_segs][chain_cap] root-first ancestors of the slice's first node
    const uint32_t* seg_anc_n;
    const unsigned long long* p0_mask;
    const uint32_t* p0_info;
    const uint32_t* pair_ofs;
    const uint16_t* pair_blk;
    const unsigned long long* pair_mask;
    ulonglong2* fn_mask;           // written for the nodes with more than two blocks whose parent has at most two:
    uint32_t* fn_blk;              // the parent's (blocks, masks)
    unsigned long long* widebits;
    uint32_t P, nseg_nodes, n_segs, chain_cap;
    uint32_t emit_lo, emit_hi;
    uint32_t tbits, n_keys;        // log2 of the open-chunk table, blocks
    uint32_t all_wide;             // 1: every record through the wide pool (the stream chunks did not work out)
    PoolView pool;
    // direct mode (k1n_kernel<true>): the first-block records (X, X, F0) never leave the wave — see k1n_kernel
    uint32_t* M;
    uint32_t N, bwidth;
    unsigned char* touched;
    uint32_t* direct_ctr;          // [KMDB_SUBPOOLS * 16]
    // flat mode (k1n_kernel<2>): the first-block records of slice s, compacted, in [s * nseg_nodes, + slice_cnt[s]) — applied by k2d_kernel
    unsigned long long* dmask;     // [n_segs * nseg_nodes] F0
    uint32_t* dwx;                 // [n_segs * nseg_nodes] w | X << 8
    uint32_t* slice_cnt;           // [n_segs]
    uint32_t dbg;                  // experiments (KMDB_K1N_DBG; results are WRONG with any bit set): 1 no write-back, 2 no matrix-core step
};
constexpr int K1N_WAVES = 4;
__host__ __device__ inline size_t k1n_wave_bytes(uint32_t chain_cap, uint32_t tbits, uint32_t n_states, uint32_t n_rows) {
    return (((size_t)chain_cap * 20 + 15) & ~(size_t)15) + arena_table_bytes(tbits, n_states) + rowtab_bytes(n_rows);          // n_rows: 0 unless row mode
}

// The node records are fetched TWO batches ahead (pair_ofs read unconditionally with them), the second pairs one batch ahead: no load of
// the loop waits on another load of the same iteration.  (Round 4 fetched one batch ahead, the second pair of a node behind its p0_info and
// pair_ofs — three dependent round trips inside the fetch: 0.07 - 0.2 ms slower, profiles/r05_j3 / r05_j4.)
//
// DIRECT (round 6; north_star: "accumulated ... before a single HBM write-back per tile"; the reference adds a decoded pattern straight
// into the matrix rows, similarity_calculator.cpp:206-241): all nodes of a slice of the DFS stream but a handful share their first
// block X — the subtree of a root child holds the patterns whose smallest sample id is that child's (profiles/r06_tile_locality.txt:
// 99.5 % of a slice's first-block records lie in its most frequent block) — so the wave that emits the records (X, X, F0, w) applies
// them where they are: the 64 masks of a batch go through the bit transpose and the byte spreading of the apply kernels (k2_apply_mfma)
// straight from the registers that hold them, six v_mfma_i32_32x32x32_i8 add them to the lower triangle of tile (X, X), which stays
// in 48 accumulator registers for the whole slice and is written back ONCE, when the slice ends or its first block changes.  No record
// of these is written, grouped or read again (C2: 47 M of 125 M records, the whole stream-chunk path and the apply kernel beside the
// wide kernel).  The rare others — a lane whose first block is not the wave's current one, a weight of 128 or more — take the
// records' way through the wide pool / the row chunks as in all_wide mode.
template <int MODE, int MINW>
__global__ __launch_bounds__(WAVE * K1N_WAVES) __attribute__((amdgpu_waves_per_eu(MINW, 8))) void k1n_kernel(const NParams q) {
    constexpr bool DIRECT = MODE == 1, FLAT = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ unsigned long long lut_ff[DIRECT ? 256 : 1], lut_01[DIRECT ? 256 : 1];       // byte b -> its 8 bits spread over 8 bytes (0xFF / 0x01 where set)
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[DIRECT ? K1N_WAVES : 1][64];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg = blockIdx.x * (blockDim.x >> 6) + wave;            // 1 .. K1N_WAVES waves per workgroup, by the LDS a wave needs
    if (DIRECT) {
        for (uint32_t b = threadIdx.x; b < 256u; b += blockDim.x) {
            unsigned long long v = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) v |= ((b >> i) & 1u) ? 0xFFull << (8 * i) : 0ull;
            lut_ff[b] = v;
            lut_01[b] = v & 0x0101010101010101ull;
        }
        __syncthreads();
    }
    if (seg >= q.n_segs) return;
    const uint32_t n_rows = q.pool.row_mode ? q.n_keys : 0u;
    unsigned char* wbase = lds_raw + k1n_wave_bytes(q.chain_cap, q.tbits, q.n_keys, n_rows) * wave;
    uint32_t* table = (uint32_t*)wbase;                                                      // open chunks
    ulonglong2* chain_m = (ulonglong2*)(wbase + arena_table_bytes(q.tbits, q.n_keys));      // [chain_cap] one slot per depth:
    uint32_t* chain_b = (uint32_t*)(chain_m + q.chain_cap);                                  // the latest node of that depth on the current root path
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    RowTab RT{nullptr, nullptr, 0u};
    if (n_rows) rowtab_init(RT, (uint32_t*)(wbase + (((size_t)q.chain_cap * 20 + 15) & ~(size_t)15) + arena_table_bytes(q.tbits, q.n_keys)), n_rows, lane);
    const uint32_t first = seg * q.nseg_nodes;
    const uint32_t end = (q.P - first) < q.nseg_nodes ? q.P : first + q.nseg_nodes;

    auto locals = [&](uint32_t info, unsigned long long m0, uint32_t e1blk, unsigned long long e1mask) -> NSum {
        const uint32_t np = info >> 16;
        if (np == 0) return nsum_make(BNONE, 0ull, BNONE, 0ull, false);
        if (np == 1) return nsum_make(info & 0xFFFFu, m0, BNONE, 0ull, false);
        return nsum_make(info & 0xFFFFu, m0, e1blk, e1mask, np > 2);
    };

    // chain slots of the first node's ancestors: inclusive merge along the root path
    {
        const uint32_t d = q.seg_anc_n[seg];
        NSum carry = nsum_make(BNONE, 0ull, BNONE, 0ull, false);
        for (uint32_t cb = 0; cb < d; cb += WAVE) {
            const uint32_t k = cb + lane;
            const bool on = k < d;
            NSum S = nsum_make(BNONE, 0ull, BNONE, 0ull, false);
            if (on) {
                const uint32_t node = q.seg_anc[(size_t)seg * q.chain_cap + k];
                const uint32_t info = q.p0_info[node];
                uint32_t e1b = BNONE; unsigned long long e1m = 0;
                if ((info >> 16) > 1u) { const uint32_t po = q.pair_ofs[node]; e1b = q.pair_blk[po]; e1m = q.pair_mask[po]; }
                S = locals(info, q.p0_mask[node], e1b, e1m);
            }
            if (lane == 0) S = nsum_merge(carry, S);
#pragma unroll
            for (int s = 1; s < WAVE; s <<= 1) {
                const NSum o = nsum_shfl_up(S, s);
                if (lane >= (uint32_t)s) S = nsum_merge(o, S);
            }
            if (on) { chain_m[k] = make_ulonglong2(S.m0, S.m1); chain_b[k] = S.bw; }
            carry = nsum_shfl(S, WAVE - 1);
        }
        lds_sync();
    }

    // node records of the NEXT batch(es) are fetched while the current one is processed
    uint32_t nx_nl = 0, nx_w = 0, nx_info = 0, nx_df = 0, nx_e1b = BNONE, nx_po = 0;
    int32_t nx_par = -1;
    unsigned long long nx_m0 = 0, nx_e1m = 0;
    uint32_t n2_nl = 0, n2_w = 0, n2_info = 0, n2_df = 0x7FFFu, n2_po = 0;          // two batches ahead
    int32_t n2_par = -1;
    unsigned long long n2_m0 = 0;
    auto fetch2 = [&](uint32_t b0) {                                  // (pair_ofs of a node without further pairs is never written: whatever is read is not used)
        const uint32_t ii = b0 + lane;
        const bool v = ii < end;
        n2_nl = v ? q.nl[ii] : 0u;
        n2_w = v ? q.w[ii] : 0u;
        n2_par = v ? q.parent[ii] : -1;
        n2_df = v ? q.dflag[ii] : 0x7FFFu;
        n2_info = v ? q.p0_info[ii] : 0u;
        n2_m0 = v ? q.p0_mask[ii] : 0ull;
        n2_po = v ? q.pair_ofs[ii] : 0u;
    };
    auto advance = [&]() {                                            // the batch two ahead becomes the next one; its second pairs are requested
        nx_nl = n2_nl; nx_w = n2_w; nx_par = n2_par; nx_df = n2_df; nx_info = n2_info; nx_m0 = n2_m0; nx_po = n2_po;
        nx_e1b = BNONE; nx_e1m = 0;
        if ((nx_info >> 16) > 1u) { nx_e1b = q.pair_blk[nx_po]; nx_e1m = q.pair_mask[nx_po]; }
    };
    fetch2(first); advance();
    if (first + WAVE < end) fetch2(first + WAVE); else { n2_nl = 0; n2_w = 0; n2_par = -1; n2_df = 0x7FFFu; n2_info = 0; n2_m0 = 0; n2_po = 0; }
    WaveArena A;
    arena_init(A, table, MODE != 0 ? 0u : q.tbits, q.n_keys, seg, lane);
    // direct mode: the lower triangle of tile (curX, curX) — rows 0..31 x cols 0..31, rows 32..63 x cols 0..31, rows 32..63 x cols 32..63
    k2_v16i c00 = {}, c10 = {}, c11 = {};
    uint32_t curX = BNONE, n_direct = 0;
    uint32_t flat_n = 0;                                             // flat mode: records of the slice so far (wave-uniform)
    auto dflush = [&]() {
        // one HBM atomic per non-zero cell (D layout of the MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).  Cell (rb + r, rb + c)
        // of the triangle sits at tri64(rb) + rb + [rb r + r (r - 1) / 2 + c]: a wave-uniform base and a 32-bit offset per lane
        const uint32_t half = lane >> 5, l31 = lane & 31u;
        const uint32_t rb = curX * q.bwidth;
        uint32_t* const Mb = q.M + (tri64((uint64_t)rb) + rb);
        const uint32_t lim = q.N - rb;                                  // rows of the block inside the matrix (masks never hold an id beyond N: belt and braces)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row0 = (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * half, row1 = row0 + 32u;
            const uint32_t o0 = rb * row0 + row0 * (row0 - 1u) / 2u + l31, o1 = rb * row1 + row1 * (row1 - 1u) / 2u + l31;
            const uint32_t v00 = (uint32_t)c00[r], v10 = (uint32_t)c10[r], v11 = (uint32_t)c11[r];
            if (v00 && l31 < row0 && row0 < lim) atomicAdd(Mb + o0, v00);
            if (v10 && row1 < lim) atomicAdd(Mb + o1, v10);
            if (v11 && l31 < row0 && row1 < lim) atomicAdd(Mb + o1 + 32u, v11);
            c00[r] = 0; c10[r] = 0; c11[r] = 0;
        }
        if (lane == 0 && q.touched) q.touched[tri32(curX) + curX] = 1;      // (all2all-sp scans only the tiles a call added to)
    };
    for (uint32_t base = first; base < end; base += WAVE) {
        const uint32_t idx = base + lane;
        const bool valid = idx < end;
        const uint32_t nl = nx_nl, w = nx_w, info = nx_info, df = nx_df, e1b = nx_e1b;
        const int32_t par = nx_par;
        const unsigned long long m0 = nx_m0, e1m = nx_e1m;
        if (base + WAVE < end) {
            advance();
            if (base + 2 * WAVE < end) fetch2(base + 2 * WAVE); else { n2_nl = 0; n2_w = 0; n2_par = -1; n2_df = 0x7FFFu; n2_info = 0; n2_m0 = 0; n2_po = 0; }
        }
        const uint32_t dep = df & 0x7FFFu;
        NSum S = locals(info, m0, e1b, e1m);
        if (valid && par >= 0 && par < (int32_t)base) {
            const ulonglong2 cm = chain_m[dep - 2u];
            S = nsum_merge(NSum{cm.x, cm.y, chain_b[dep - 2u]}, S);
        }
        // pointer doubling over the in-batch parents
        int pl = (valid && par >= (int32_t)base) ? (int)(par - (int32_t)base) : -1;
        while (__ballot(pl >= 0)) {
            const int src = pl >= 0 ? pl : (int)lane;
            const NSum o = nsum_shfl(S, src);
            if (pl >= 0) S = nsum_merge(o, S);
            const int npl = __shfl(pl, src, WAVE);
            pl = pl >= 0 ? npl : -1;
        }
        const bool wide = valid && ((S.bw >> 30) & 1u);
        const uint32_t w0 = S.bw & 0x7FFFu, w1 = (S.bw >> 15) & 0x7FFFu;
        const unsigned long long F0 = S.m0, F1 = w1 != BNONE ? S.m1 : 0ull;
        {
            const unsigned long long wb = __ballot(wide);
            if (lane == 0) q.widebits[base >> 6] = wb;
        }
        if (__ballot(wide)) {
            // a node with more blocks whose parent still has at most two takes the parent's (blocks, masks) along: the wide
            // kernel starts its list from them
            const int pl0 = (valid && par >= (int32_t)base) ? (int)(par - (int32_t)base) : (int)lane;
            NSum Pp = nsum_shfl(S, pl0);
            if (valid && par >= 0 && par < (int32_t)base) { const ulonglong2 cm = chain_m[dep - 2u]; Pp = NSum{cm.x, cm.y, chain_b[dep - 2u]}; }
            if (wide && par >= 0 && !((Pp.bw >> 30) & 1u)) {
                const uint32_t pb0 = Pp.bw & 0x7FFFu, pb1 = (Pp.bw >> 15) & 0x7FFFu;
                q.fn_mask[idx] = make_ulonglong2(Pp.m0, pb1 != BNONE ? Pp.m1 : 0ull);
                q.fn_blk[idx] = (pb0 == BNONE ? 0xFFFFu : pb0) | ((pb1 == BNONE ? 0xFFFFu : pb1) << 16);
            }
        }
        // ---- records (flat form): (w0, w0, F0), and with a second block (w1, w0, F1, F0) and (w1, w1, F1).  A diagonal
        // record needs two ids to have a pair.
        const bool act = valid && !wide && w != 0 && nl >= 2u && w0 != BNONE && idx >= q.emit_lo && idx < q.emit_hi;
        auto first_block_records = [&]() {
            // (w0, w0, F0): the lanes of a batch mostly share the block — one reservation per block in that block's open chunk.
            // (In wide mode, q.all_wide, these records take the wide pool as well.)
            bool d0 = act && __popcll(F0) >= 2;
            if (FLAT) {
                // the slice's first-block records, compacted in DFS order: two coalesced stores, no reservation, no chunk table
                const bool dd = d0 && w < 128u;                          // (an int8 operand holds a weight below 128; the others take the sort)
                const unsigned long long bm = __ballot(dd);
                if (dd) {
                    const uint32_t slot = first + flat_n + (uint32_t)__popcll(bm & lt_mask);
                    q.dmask[slot] = F0; q.dwx[slot] = w | (w0 << 8);
                }
                flat_n += (uint32_t)__popcll(bm);
                d0 = d0 && !dd;
            }
            if (DIRECT) {
                const bool dd = d0 && w < 128u;                          // (an int8 operand holds a weight below 128; nearly every weight is)
                const unsigned long long pend = __ballot(dd);
                if (pend) {
                    // ONE tile per wave: the block of the slice's first such record.  (Switching tiles where a slice crosses from one root child's
                    // subtree into the next would put a second write-back — 48 conditional atomics — into the loop body; the straddling
                    // slices are 2 - 6 % of all, and what they hold of the other block takes the records' way.)
                    if (curX == BNONE) curX = bcast(w0, (uint32_t)__builtin_ctzll(pend));
                    const bool mine = dd && w0 == curX;
                    n_direct += (uint32_t)__popcll(__ballot(mine));
                    if (!(q.dbg & 2u)) {
                    const TrConst trc = tr_const(lane);                      // (ten registers: made here, not kept across the batch)
                    const unsigned long long Rt = transpose64(mine ? F0 : 0ull, trc);        // lane r: bit k <=> node k of the batch has id r of the block
                    unsigned long long ra0, ra1;
                    {
                        const auto a = __builtin_amdgcn_permlane32_swap((uint32_t)Rt, (uint32_t)Rt, false, false);                 // [0]: the lower half's words everywhere, [1]: the upper half's
                        const auto b = __builtin_amdgcn_permlane32_swap((uint32_t)(Rt >> 32), (uint32_t)(Rt >> 32), false, false);
                        ra0 = ((unsigned long long)b[0] << 32) | a[0];
                        ra1 = ((unsigned long long)b[1] << 32) | a[1];
                    }
                    wbuf[wave][lane] = (unsigned char)(mine ? w : 0u);
                    lds_sync();
                    const uint32_t half = lane >> 5;
                    auto spread = [&](unsigned long long word, uint32_t shift, const unsigned long long* lut) -> k2_v4i {
                        const uint32_t f = (uint32_t)(word >> shift) & 0xFFFFu;
                        k2_v4i r;
                        const unsigned long long lo = lut[f & 0xFFu], hi = lut[f >> 8];
                        r[0] = (int)(uint32_t)lo; r[1] = (int)(uint32_t)(lo >> 32); r[2] = (int)(uint32_t)hi; r[3] = (int)(uint32_t)(hi >> 32);
                        return r;
                    };
    #pragma unroll 1
                    for (uint32_t kh = 0; kh < 2; ++kh) {
                        const uint32_t shift = 32u * kh + 16u * half;            // nodes 32 kh + 16 half .. + 15 of the batch
                        k2_v4i a0 = spread(ra0, shift, lut_ff), a1 = spread(ra1, shift, lut_ff);
                        const k2_v4i b0 = spread(ra0, shift, lut_01), b1 = spread(ra1, shift, lut_01);
                        const k2_v4i wv = *(const k2_v4i*)(wbuf[wave] + shift);
                        a0 &= wv; a1 &= wv;
                        c00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c00, 0, 0, 0);
                        c10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c10, 0, 0, 0);
                        c11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c11, 0, 0, 0);
                    }
                    lds_sync();
                    }
                    d0 = d0 && !mine;                                        // what is left takes the records' way
                }
            }
            if (MODE == 0 && !q.all_wide) {
                unsigned long long pend = __ballot(d0);
                while (pend) {
                    const uint32_t X0 = bcast(w0, (uint32_t)__builtin_ctzll(pend));
                    const bool mine = d0 && w0 == X0;
                    const unsigned long long bc = __ballot(mine);
                    const Resv r = arena_reserve(A, q.pool, X0, tri32(X0) + X0, (uint32_t)__popcll(bc), lane);
                    if (mine) rec_store_diag(q.pool, resv_slot(r, (uint32_t)__popcll(bc & lt_mask)), F0, w);
                    pend &= ~bc;
                }
            } else if (n_rows) {
                row_emit(A, q.pool, RT, d0, w0, F0, F0, w, tri32(w0) + w0, lane, lt_mask);
            } else {
                wide_emit(A, q.pool, d0, F0, F0, w, tri32(w0) + w0, lane, lt_mask);
            }
        };
        if (!DIRECT) first_block_records();
        // second blocks, (w1, w0, F1, F0) and (w1, w1, F1): the pairs differ from lane to lane, so these records go to the wide
        // pool in arrival order (one reservation for all lanes) and are grouped by the sort
        {
            const bool act2 = act && F1 != 0;
            if (__ballot(act2)) {
                if (n_rows) {
                    row_emit(A, q.pool, RT, act2, w1, F1, F0, w, tri32(w1) + w0, lane, lt_mask);
                    row_emit(A, q.pool, RT, act2 && __popcll(F1) >= 2, w1, F1, F1, w, tri32(w1) + w1, lane, lt_mask);
                } else {
                    wide_emit(A, q.pool, act2, F1, F0, w, tri32(w1) + w0, lane, lt_mask);
                    wide_emit(A, q.pool, act2 && __popcll(F1) >= 2, F1, F1, w, tri32(w1) + w1, lane, lt_mask);
                }
            }
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
            if (valid && dep < later) { chain_m[dep - 1u] = make_ulonglong2(S.m0, S.m1); chain_b[dep - 1u] = S.bw; }
            lds_sync();
        }
        // (direct mode: the matrix-core step comes last in the batch — nothing of the batch but F0, w0 and w is live beside its operands)
        if (DIRECT) first_block_records();
    }
    if (DIRECT && curX != BNONE && !(q.dbg & 1u)) dflush();
    if (DIRECT && lane == 0 && n_direct) atomicAdd(&q.direct_ctr[(seg % KMDB_SUBPOOLS) * 16u], n_direct);
    if (FLAT && lane == 0) { q.slice_cnt[seg] = flat_n; if (flat_n) atomicAdd(&q.direct_ctr[(seg % KMDB_SUBPOOLS) * 16u], flat_n); }
    arena_finish(A, q.pool, lane);
    if (n_rows) rowtab_finish(RT, q.pool, lane);
}

// ------------------------------------------------------------------------------------------
// K2d: the narrow kernel's first-block records, slice by slice
// ------------------------------------------------------------------------------------------
// Round 6.  The records (X, X, F0, w) of the nodes with at most two blocks are as clustered as records can be — a slice of the DFS stream is, but
// for 0.5 - 2 % of its records, ONE tile (profiles/r06_tile_locality.txt) — so they need no chunk table, no grouping and no tile in LDS: the narrow
// kernel leaves them compacted in DFS order (12 bytes per record, coalesced), and one wave per K2D_SLICES slices streams them through the
// matrix-core step of the apply kernels (bit transpose, byte spreading, six v_mfma_i32_32x32x32_i8 per 64 records: the lower triangle of the tile)
// with the tile in 48 accumulator registers, written back when the block changes: once per slice, nearly always.  Side stream, beside the wide
// kernel.  (The same step INSIDE the narrow kernel — k1n_kernel<1> — writes no record at all, but its 48 accumulators beside the narrow kernel's 95
// registers leave three waves per SIMD: the narrow kernel then takes 0.65 ms longer at C2 than this kernel costs beside the wide kernel:
// profiles/r06_j3, r06_j4.)
constexpr uint32_t K2D_SLICES = 2;
struct DParams {
    const unsigned long long* dmask;
    const uint32_t* dwx;
    const uint32_t* slice_cnt;
    uint32_t n_segs, nseg_nodes, slices;
    uint32_t* M;
    uint32_t N, bwidth;
    unsigned char* touched;
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k2d_kernel(const DParams q) {
    __shared__ unsigned long long lut_ff[256], lut_01[256];       // byte b -> its 8 bits spread over 8 bytes (0xFF / 0x01 where set)
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[4][64];
    {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((threadIdx.x >> i) & 1u) ? 0xFFull << (8 * i) : 0ull;
        lut_ff[threadIdx.x] = v;
        lut_01[threadIdx.x] = v & 0x0101010101010101ull;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t half = lane >> 5, l31 = lane & 31u;
    const uint32_t s0 = (blockIdx.x * 4u + wave) * q.slices;
    if (s0 >= q.n_segs) return;
    const uint32_t s1 = s0 + q.slices < q.n_segs ? s0 + q.slices : q.n_segs;
    k2_v16i c00 = {}, c10 = {}, c11 = {};
    uint32_t curX = BNONE;
    const TrConst trc = tr_const(lane);
    auto flush = [&]() {
        // one HBM atomic per non-zero cell (D layout of the MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).  Cell (rb + r, rb + c)
        // of the triangle sits at tri64(rb) + rb + [rb r + r (r - 1) / 2 + c]: a wave-uniform base and a 32-bit offset per lane
        const uint32_t rb = curX * q.bwidth;
        uint32_t* const Mb = q.M + (tri64((uint64_t)rb) + rb);
        const uint32_t lim = q.N - rb;                                  // rows of the block inside the matrix (masks never hold an id beyond N: belt and braces)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row0 = (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * half, row1 = row0 + 32u;
            const uint32_t o0 = rb * row0 + row0 * (row0 - 1u) / 2u + l31, o1 = rb * row1 + row1 * (row1 - 1u) / 2u + l31;
            const uint32_t v00 = (uint32_t)c00[r], v10 = (uint32_t)c10[r], v11 = (uint32_t)c11[r];
            if (v00 && l31 < row0 && row0 < lim) atomicAdd(Mb + o0, v00);
            if (v10 && row1 < lim) atomicAdd(Mb + o1, v10);
            if (v11 && l31 < row0 && row1 < lim) atomicAdd(Mb + o1 + 32u, v11);
            c00[r] = 0; c10[r] = 0; c11[r] = 0;
        }
        if (lane == 0 && q.touched) q.touched[tri32(curX) + curX] = 1;      // (all2all-sp scans only the tiles a call added to)
    };
    auto spread = [&](unsigned long long word, uint32_t shift, const unsigned long long* lut) -> k2_v4i {
        const uint32_t f = (uint32_t)(word >> shift) & 0xFFFFu;
        k2_v4i r;
        const unsigned long long lo = lut[f & 0xFFu], hi = lut[f >> 8];
        r[0] = (int)(uint32_t)lo; r[1] = (int)(uint32_t)(lo >> 32); r[2] = (int)(uint32_t)hi; r[3] = (int)(uint32_t)(hi >> 32);
        return r;
    };
    // the 64 records of a step, one per lane; the next step's are requested before this one is applied
    uint32_t s = s0, n = q.slice_cnt[s0], b = 0;
    auto skip_empty = [&]() { while (b >= n && s + 1u < s1) { ++s; n = q.slice_cnt[s]; b = 0; } };
    skip_empty();
    unsigned long long nm = 0; uint32_t nwx = 0;
    auto fetch = [&]() {
        nm = 0; nwx = 0;
        if (b + lane < n) { const size_t i = (size_t)s * q.nseg_nodes + b + lane; nm = q.dmask[i]; nwx = q.dwx[i]; }
    };
    if (b < n) fetch();
    for (;;) {
        // a step of 64 records — or, behind the last one, an empty step whose "block" differs from every tile: the ONE write-back site serves the
        // changes of block and the end (a second inlined copy of its 48 conditional atomics cost 60 registers)
        const bool cur = b < n;
        const unsigned long long m = cur ? nm : 0ull;
        const uint32_t wx = cur ? nwx : 0u;
        if (cur) {
            b += WAVE;
            skip_empty();
            if (b < n) fetch();
        }
        unsigned long long pend = __ballot(m != 0ull);
        for (;;) {
            const uint32_t X0 = pend ? bcast(wx >> 8, (uint32_t)__builtin_ctzll(pend)) : BNONE;
            if (X0 != curX) { if (curX != BNONE) flush(); curX = X0; }
            if (!pend) break;
            const bool mine = m != 0ull && (wx >> 8) == X0;
            pend &= ~__ballot(mine);
            const unsigned long long Rt = transpose64(mine ? m : 0ull, trc);        // lane r: bit k <=> record k of the step has id r of the block
            unsigned long long ra0, ra1;
            {
                const auto a = __builtin_amdgcn_permlane32_swap((uint32_t)Rt, (uint32_t)Rt, false, false);                 // [0]: the lower half's words everywhere, [1]: the upper half's
                const auto c = __builtin_amdgcn_permlane32_swap((uint32_t)(Rt >> 32), (uint32_t)(Rt >> 32), false, false);
                ra0 = ((unsigned long long)c[0] << 32) | a[0];
                ra1 = ((unsigned long long)c[1] << 32) | a[1];
            }
            wbuf[wave][lane] = (unsigned char)(mine ? (wx & 0xFFu) : 0u);
            lds_sync();
#pragma unroll 1
            for (uint32_t kh = 0; kh < 2; ++kh) {
                const uint32_t shift = 32u * kh + 16u * half;            // records 32 kh + 16 half .. + 15 of the step
                k2_v4i a0 = spread(ra0, shift, lut_ff), a1 = spread(ra1, shift, lut_ff);
                const k2_v4i b0 = spread(ra0, shift, lut_01), b1 = spread(ra1, shift, lut_01);
                const k2_v4i wv = *(const k2_v4i*)(wbuf[wave] + shift);
                a0 &= wv; a1 &= wv;
                c00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c00, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c11, 0, 0, 0);
            }
            lds_sync();
            if (!pend) break;                                    // (the step's records are in; a change of block is seen with the next step)
        }
        if (!cur) break;
    }
}

// ------------------------------------------------------------------------------------------
// wide list: compaction of the flagged nodes
// ------------------------------------------------------------------------------------------
__global__ void wide_count_kernel(const unsigned long long* __restrict__ bits, uint32_t n_words, uint32_t* __restrict__ cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) cnt[i] = (uint32_t)__popcll(bits[i]);
    else if (i == n_words) cnt[i] = 0;
}
__global__ void wide_expand_kernel(const unsigned long long* __restrict__ bits, const uint32_t* __restrict__ base, uint32_t n_words,
                                   uint32_t* __restrict__ widx, uint32_t cap, uint32_t* __restrict__ counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) counters[KCTR_NWIDE] = base[n_words];
    if (i >= n_words) return;
    unsigned long long b = bits[i];
    uint32_t o = base[i];
    while (b) {
        const uint32_t k = (uint32_t)__builtin_ctzll(b);
        b &= b - 1;
        if (o < cap) widx[o] = i * 64u + k;
        ++o;
    }
}

// ------------------------------------------------------------------------------------------
// K1w: nodes with more than two blocks
// ------------------------------------------------------------------------------------------
// Summary of a stretch of a root path as a list of (block, mask) entries: how many entries it has, its first and last block and
// the mask of its last entry.  Ids ascend along a root path, so two stretches can only share the block at their seam; stretches
// merge associatively (pointer doubling across lanes, scans along a path).
constexpr uint32_t BLK_NONE = 0xFFFFu;
struct LSum { unsigned long long lm; uint32_t c, fb, lb; };
__device__ __forceinline__ LSum lsum_none() { return LSum{0ull, 0u, BLK_NONE, BLK_NONE}; }
// A = the stretch closer to the root, B = the stretch below it
__device__ __forceinline__ LSum lsum_merge(const LSum& A, const LSum& B) {
    if (B.c == 0u) return A;
    if (A.c == 0u) return B;
    const bool seam = A.lb == B.fb;
    return LSum{(B.c == 1u && seam) ? (A.lm | B.lm) : B.lm, A.c + B.c - (seam ? 1u : 0u), A.fb, B.lb};
}
__device__ __forceinline__ LSum lsum_shfl(const LSum& s, int src) {
    return LSum{shfl64(s.lm, src), (uint32_t)__shfl((int)s.c, src, WAVE), (uint32_t)__shfl((int)s.fb, src, WAVE), (uint32_t)__shfl((int)s.lb, src, WAVE)};
}
__device__ __forceinline__ LSum lsum_shfl_up(const LSum& s, int d) {
    return LSum{shfl_up64(s.lm, d), (uint32_t)__shfl_up((int)s.c, d, WAVE), (uint32_t)__shfl_up((int)s.fb, d, WAVE), (uint32_t)__shfl_up((int)s.lb, d, WAVE)};
}
// the (blocks, masks) the narrow kernel left for a wide node whose parent has at most two blocks
__device__ __forceinline__ LSum lsum_of_fn(uint32_t fb, const ulonglong2& fm) {
    const uint32_t b0 = fb & 0xFFFFu, b1 = fb >> 16;
    if (b0 == BLK_NONE) return lsum_none();
    if (b1 == BLK_NONE) return LSum{fm.x, 1u, b0, b0};
    return LSum{fm.y, 2u, b0, b1};
}

// Second level above the block records.  A node whose full list has c >= min_blocks blocks would write c (c + 1) / 2 records; instead the wave
// that emits it takes a node index g, appends the node's c (g, block, mask) entries to an entry pool and sets bit g in the bitmap of every
// block of the list.  After the wide kernel: rank directories of the bitmaps, every entry placed in its block's list by its rank (no sort),
// and one job per tile (X, Y) that ANDs the two bitmaps, ranks the matches in both lists and feeds them, 64 at a time, to the apply step
// (l2_join_apply_kernel).  Prototype and measurements: profiles/r04_join_apply_probe.hip.
struct L2View {
    uint32_t on, min_blocks;
    uint32_t node_cap, W;          // node indices (a multiple of 64 * L2_SUB), bitmap words per block
    uint32_t ent_cap;              // entries (a multiple of L2_SUB)
    uint32_t* cursors;             // [2 * L2_SUB * 16]: node indices handed out per sub-range, then entries handed out per sub-range
    unsigned long long* bitmap;    // [blocks][W]
    uint32_t* ent_g;               // [ent_cap]
    uint16_t* ent_blk;             // [ent_cap], 0xFFFF: never written
    unsigned long long* ent_mask;  // [ent_cap]
    uint32_t* node_w;              // [node_cap]
};
struct WParams {
    const uint32_t* widx;          // the wide nodes, DFS order
    uint32_t n_wide;
    const uint32_t* nl;
    const int32_t* parent;
    const uint32_t* w;
    const uint16_t* dflag;
    const unsigned long long* widebits;
    const uint32_t* wide_base;     // [n_words + 1] wide nodes before every 64-node word of the DFS stream
    const unsigned long long* p0_mask;
    const uint32_t* p0_info;
    const uint32_t* pair_ofs;
    const uint16_t* pair_blk;
    const unsigned long long* pair_mask;
    const ulonglong2* fn_mask;
    const uint32_t* fn_blk;
    const uint32_t* seg_anc;       // [n_runs][chain_cap] root path of the first node of every run, root first (wrun_anc_kernel)
    const uint32_t* seg_anc_n;
    uint32_t run_nodes;            // wide nodes per run (a multiple of 64)
    uint32_t n_runs, n_waves;
    uint32_t* run_ctr;             // [K1W_CTRS * 16] next run of every class (run r belongs to class r % K1W_CTRS), a cache line apart
    uint32_t chain_cap, arena_cap, e_cap;
    uint32_t n_rows;               // row mode: block rows (0 otherwise)
    uint32_t emit_lo, emit_hi;
    PoolView pool;
    L2View l2;
};
constexpr int K1W_WAVES = 2;

constexpr uint32_t L2_MIN_BLOCKS = 24;     // nodes with that many blocks take the second level (measured at 10 000 samples: 11 -> 21.9 ms, 24 -> 20.0; KMDB_L2_MIN moves it)
constexpr uint32_t L2_NODE_GRAB = 16, L2_ENT_GRAB = 1024, L2_SUB = 16;   // node indices / entries a wave takes per device atomic, from one of L2_SUB cursors each
constexpr uint32_t K1W_QCAP = 64;          // inclusive record counts of the lanes of a batch (the owner search of the record-parallel emission)
constexpr uint32_t K1W_HEAVY = 11;         // a node with that many blocks (66 records and more) is emitted by the whole wave
constexpr uint32_t K1W_OXCAP = 96;         // further own pairs of a batch's nodes kept in LDS (with the queue and the node array trimmed the wave's LDS drops
                                           // below 10 KB at 200 blocks: 16 instead of 14 waves per CU — the kernel waits more than it issues)
constexpr uint32_t K1W_CTRS = 16;          // run counters: a wave takes its next run from the counter of its class
constexpr uint32_t K1W_ARENA_MIN = 256;    // entries of a wave's row arena at least (and always one full list: as many as there are blocks)
enum : uint32_t { WB_NONE = 0, WB_FN = 1, WB_LANE = 2, WB_CHAIN = 3 };
// LDS of one wave, carved from the dynamic allocation (the sizes depend on the database: blocks, depth of the tree)
struct K1WLds {
    unsigned long long* ent_mask;  // [arena_cap] rows of the batch's nodes, one contiguous run of entries per lane
    unsigned long long* e_mask;    // [e_cap] the chain list: the full list of the deepest node of the chain
    unsigned long long* ch_last;   // [chain_cap] per depth: the mask of the node's last entry (the chain list may hold more ids there)
    unsigned long long* own_m0;    // [64] first own pair of every lane's node
    unsigned long long* own_desc;  // [64] what a walk needs of the lane's node in ONE read: first own block | link << 16 (parent's lane, WB_LANE, or chain slot,
                                   //      WB_CHAIN) | ox << 32 (first copy of the further own pairs in ox_*, 0xFFFF: not copied) | own pairs << 48 | WB_* << 62
    uint32_t* ch_node;             // [chain_cap] the wide node of that depth on the current root path (0xFFFFFFFF: none)
    unsigned long long* ox_mask;   // [K1W_OXCAP] the batch's further own pairs (a copy: the walks read them once per descendant)
    uint32_t* own_po;              // [64] further own pairs: first entry in the pair pool
    uint32_t* st_w;                // [64]
    uint32_t* queue;               // [K1W_QCAP] (first 64: inclusive record counts of the lanes of a batch)
    uint16_t* ent_blk;             // [arena_cap]
    uint16_t* e_blk;               // [e_cap]
    uint16_t* ch_len;              // [chain_cap] entries of the node's list = a prefix of the chain list
    uint16_t* ox_blk;              // [K1W_OXCAP]
    uint16_t* st_start;            // [64] first arena entry of the lane's row
    uint16_t* st_pre;              // [64] entries of the lane's list that are NOT in its row: they are the first st_pre entries of the chain list
};
__host__ __device__ inline size_t k1w_core_bytes(uint32_t arena_cap, uint32_t e_cap, uint32_t chain_cap) {
    const size_t b = (size_t)8 * (arena_cap + e_cap + chain_cap + 2 * 64 + K1W_OXCAP) + (size_t)4 * (chain_cap + 2 * 64 + K1W_QCAP) +
                     (size_t)2 * (arena_cap + e_cap + chain_cap + 2 * 64 + K1W_OXCAP);
    return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t k1w_wave_bytes(uint32_t arena_cap, uint32_t e_cap, uint32_t chain_cap, uint32_t n_rows) {
    return k1w_core_bytes(arena_cap, e_cap, chain_cap) + rowtab_bytes(n_rows);
}
__device__ __forceinline__ K1WLds k1w_carve(unsigned char* p, uint32_t arena_cap, uint32_t e_cap, uint32_t chain_cap) {
    K1WLds L;
    L.ent_mask = (unsigned long long*)p;  L.e_mask = L.ent_mask + arena_cap;  L.ch_last = L.e_mask + e_cap;  L.own_m0 = L.ch_last + chain_cap;
    L.own_desc = L.own_m0 + 64;
    L.ox_mask = L.own_desc + 64;
    L.ch_node = (uint32_t*)(L.ox_mask + K1W_OXCAP);  L.own_po = L.ch_node + chain_cap;  L.st_w = L.own_po + 64;  L.queue = L.st_w + 64;
    L.ent_blk = (uint16_t*)(L.queue + K1W_QCAP);  L.e_blk = L.ent_blk + arena_cap;  L.ch_len = L.e_blk + e_cap;  L.st_start = L.ch_len + chain_cap;
    L.st_pre = L.st_start + 64;  L.ox_blk = L.st_pre + 64;
    return L;
}

// root path of the first node of every run of the wide-node kernel (root first), by climbing the parent links: one thread per run,
// inside the call (the wide list depends on the decoded ids)
__global__ void wrun_anc_kernel(const uint32_t* __restrict__ widx, uint32_t n_wide, uint32_t run_nodes, const int32_t* __restrict__ parent,
                                const uint16_t* __restrict__ dflag, uint32_t chain_cap, uint32_t* __restrict__ anc, uint32_t* __restrict__ anc_n) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if ((uint64_t)r * run_nodes >= n_wide) return;
    int32_t cur = parent[widx[(size_t)r * run_nodes]];
    uint32_t d = cur < 0 ? 0u : (uint32_t)(dflag[cur] & 0x7FFFu);
    if (d > chain_cap) d = 0;
    anc_n[r] = d;
    while (cur >= 0 && d) { anc[(size_t)r * chain_cap + (--d)] = (uint32_t)cur; cur = parent[cur]; }
}

// One lane per wide node, batches of 64 consecutive nodes of the (DFS-ordered) wide list.  A run = a few consecutive batches;
// the waves take runs as they go (the records per node differ by orders of magnitude — 3 blocks: 6
// records, 200 blocks: 20 100 — and heavy nodes sit together in the DFS order).
// A wide node's parent is wide as well or has at most two blocks.  Its list = the parent's list + its own pairs:
//   parent with <= 2 blocks: the (blocks, masks) the narrow kernel left in HBM;
//   wide parent inside the batch: that lane (the lengths resolve by pointer doubling, the rows are filled by walking the in-batch
//   parents);
//   wide parent before the batch: the CHAIN — the wide nodes on the root path of the previous batch's last node.  Their lists are
//   prefixes of one another (ids ascend along a root path), so the chain is ONE list, the deepest node's, plus per depth the
//   number of entries and the mask of the node's own last entry.  At the start of a run the chain is built from the root path of
//   the slice's first node (a table made at upload): the lists of all its wide ancestors in one scan.
// No node climbs parent links, and a list may have as many entries as there are blocks.
__global__ __launch_bounds__(WAVE * K1W_WAVES) void k1w_kernel(const WParams q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + wave;         // 1 or K1W_WAVES waves per workgroup, by the LDS a wave needs
    if (wid >= q.n_waves) return;
    unsigned char* wbase = lds_raw + k1w_wave_bytes(q.arena_cap, q.e_cap, q.chain_cap, q.n_rows) * wave;
    const K1WLds L = k1w_carve(wbase, q.arena_cap, q.e_cap, q.chain_cap);
    RowTab RT{nullptr, nullptr, 0u};
    if (q.n_rows) rowtab_init(RT, (uint32_t*)(wbase + k1w_core_bytes(q.arena_cap, q.e_cap, q.chain_cap)), q.n_rows, lane);
    auto iswide = [&](uint32_t y) -> bool { return (q.widebits[y >> 6] >> (y & 63u)) & 1ull; };
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    WaveArena A;
    arena_init(A, nullptr, 0u, 0u, wid + 64u, lane);
    uint32_t n_miss = 0;

    uint32_t l2_nodes = 0;
    uint32_t l2_gnext = 0, l2_gstock = 0, l2_enext = 0, l2_estock = 0;     // second level: the wave's stock of node indices and entries (wave-uniform)
    // record-parallel emission of the lanes in `on` (list of lane j: m entries from st_start[j]): a node with m blocks owns
    // m (m + 1) / 2 records (block pairs a >= b), one record per lane and step
    auto emit = [&](bool on, uint32_t m, uint32_t wv) {
        // a node with many blocks is taken by the whole wave: lane t builds pair t of the node
        {
            unsigned long long hb = __ballot(on && m >= K1W_HEAVY);
            while (hb) {
                const uint32_t j = (uint32_t)__builtin_ctzll(hb);
                hb &= hb - 1ull;
                const uint32_t mj = bcast(m, j), wj = bcast(wv, j);
                const uint32_t stj = L.st_start[j], prej = L.st_pre[j];
                // entry e of the node's list: the first prej from the chain list, the others from its row
                auto cmask = [&](uint32_t e) -> unsigned long long { return e < prej ? L.e_mask[e] : L.ent_mask[stj + e - prej]; };
                auto cblk = [&](uint32_t e) -> uint32_t { return e < prej ? (uint32_t)L.e_blk[e] : (uint32_t)L.ent_blk[stj + e - prej]; };
                if (q.l2.on && mj >= q.l2.min_blocks) {
                    // second level: the node's entries instead of its records
                    const uint32_t sub = wid % L2_SUB;
                    if (l2_gstock == 0u) {
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(&q.l2.cursors[sub * 16u], L2_NODE_GRAB);
                        b = bcast(b, 0);
                        // the grabs of the L2_SUB cursors interleave (grab n of cursor k = indices [(n L2_SUB + k) GRAB, + GRAB)): the indices in use
                        // stay dense from 0 on as long as the cursors advance alike, and the tile jobs scan the bitmaps only that far
                        const uint32_t g0 = ((b / L2_NODE_GRAB) * L2_SUB + sub) * L2_NODE_GRAB;
                        if (g0 + L2_NODE_GRAB <= q.l2.node_cap) { l2_gnext = g0; l2_gstock = L2_NODE_GRAB; }
                    }
                    if (l2_estock < mj) {
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(&q.l2.cursors[(L2_SUB + sub) * 16u], L2_ENT_GRAB);
                        b = bcast(b, 0);
                        if (b + L2_ENT_GRAB <= q.l2.ent_cap / L2_SUB) { l2_enext = sub * (q.l2.ent_cap / L2_SUB) + b; l2_estock = L2_ENT_GRAB; }
                        else l2_estock = 0u;
                    }
                    if (l2_gstock == 0u || l2_estock < mj) {             // out of indices or entries: the call is repeated with larger arrays
                        if (lane == 0) atomicOr(&q.pool.counters[KCTR_L2_OVERFLOW], 1u);
                        continue;
                    }
                    const uint32_t g = l2_gnext;
                    ++l2_gnext; --l2_gstock;
                    for (uint32_t e = lane; e < mj; e += WAVE) {
                        const uint32_t idx = l2_enext + e, blk = cblk(e);
                        q.l2.ent_g[idx] = g; q.l2.ent_blk[idx] = (uint16_t)blk; q.l2.ent_mask[idx] = cmask(e);
                        atomicOr(&q.l2.bitmap[(size_t)blk * q.l2.W + (g >> 6)], 1ull << (g & 63u));
                    }
                    l2_enext += mj; l2_estock -= mj;
                    if (lane == 0) q.l2.node_w[g] = wj;
                    ++l2_nodes;
                    continue;
                }
                const uint32_t Tj = mj * (mj + 1u) / 2u;
                for (uint32_t t0 = 0; t0 < Tj; t0 += WAVE) {
                    const uint32_t t = t0 + lane;
                    bool rec_on = false;
                    uint32_t stream = 0, X = 0;
                    unsigned long long FX = 0, FY = 0;
                    if (t < Tj) {
                        uint32_t a = (uint32_t)((__fsqrt_rn(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                        while (tri32(a) > t) --a;
                        while (tri32(a + 1u) <= t) ++a;
                        const uint32_t b = t - tri32(a);
                        FX = cmask(a); FY = cmask(b);
                        X = cblk(a);
                        const uint32_t Y = cblk(b);
                        rec_on = a != b || __popcll(FX) >= 2;
                        if (a == b) FY = FX;
                        stream = tri32(X) + Y;
                    }
                    if (q.n_rows) row_emit(A, q.pool, RT, rec_on, X, FX, FY, wj, stream, lane, lt_mask);
                    else wide_emit(A, q.pool, rec_on, FX, FY, wj, stream, lane, lt_mask);
                }
            }
            on = on && m < K1W_HEAVY;
        }
        // the others record by record: the records of the batch are numbered by a prefix sum over the lanes, lane t of a step finds
        // the owner of record t by a binary search in the prefix sums (LDS) and the pair (a, b) from the record's index inside the node.
        // (A queue of descriptors pushed by the owners cost a loop of up to 55 iterations per owner: 2/3 of the kernel's VALU work at
        // 10 000 samples.)
        const uint32_t myrec = on ? m * (m + 1u) / 2u : 0u;
        const uint32_t rincl = wave_incl_scan(myrec, lane);
        const uint32_t T = bcast(rincl, WAVE - 1);
        L.queue[lane] = rincl;
        L.st_w[lane] = wv;
        lds_sync();
        for (uint32_t t0 = 0; t0 < T; t0 += WAVE) {
            const uint32_t t = t0 + lane;
            bool rec_on = false, diag = false;
            uint32_t stream = 0, ww = 0, X = 0;
            unsigned long long FX = 0, FY = 0;
            if (t < T) {
                uint32_t own = 0;                               // the first lane whose inclusive sum exceeds t
#pragma unroll
                for (uint32_t sft = 32; sft >= 1u; sft >>= 1) if (L.queue[own + sft - 1u] <= t) own += sft;
                const uint32_t r0 = t - (own ? L.queue[own - 1u] : 0u);
                uint32_t a = (uint32_t)((__fsqrt_rn(8.0f * (float)r0 + 1.0f) - 1.0f) * 0.5f);
                while (tri32(a) > r0) --a;
                while (tri32(a + 1u) <= r0) ++a;
                const uint32_t b = r0 - tri32(a);
                const uint32_t st = L.st_start[own], pre = L.st_pre[own];
                FX = a < pre ? L.e_mask[a] : L.ent_mask[st + a - pre]; FY = b < pre ? L.e_mask[b] : L.ent_mask[st + b - pre];
                X = a < pre ? (uint32_t)L.e_blk[a] : (uint32_t)L.ent_blk[st + a - pre];
                const uint32_t Y = b < pre ? (uint32_t)L.e_blk[b] : (uint32_t)L.ent_blk[st + b - pre];
                ww = L.st_w[own];
                diag = a == b;
                rec_on = !diag || __popcll(FX) >= 2;               // a diagonal record needs two ids to have a pair
                stream = tri32(X) + Y;
            }
            // few streams: the step's records go to the wide pool in arrival order (one reservation for all lanes); many: to
            // the chunks of their block rows
            if (q.n_rows) row_emit(A, q.pool, RT, rec_on, X, FX, diag ? FX : FY, ww, stream, lane, lt_mask);
            else wide_emit(A, q.pool, rec_on, FX, diag ? FX : FY, ww, stream, lane, lt_mask);
        }
        lds_sync();
    };

    // The records per node differ by orders of magnitude (3 blocks: 6 records, 200 blocks: 20 100) and heavy nodes sit together
    // in the DFS order: runs dealt round-robin left some waves with several heavy runs (the kernel ended when the slowest wave did,
    // the average wave at half that time).  A wave takes its next run when it is done with one — from one of K1W_CTRS counters (a
    // single counter would be a hot address: same-address device atomics run at some ten million per second).
    for (;;) {
        uint32_t run = 0;
        const uint32_t n_cls = q.n_waves < K1W_CTRS ? q.n_waves : K1W_CTRS;       // classes of runs = counters in use (every class needs a wave)
        if (lane == 0) run = atomicAdd(&q.run_ctr[(wid % n_cls) * 16u], 1u);
        run = bcast(run, 0) * n_cls + wid % n_cls;
        if (run >= q.n_runs) break;
        const uint32_t kb = run * q.run_nodes;
        const uint32_t ke = q.n_wide - kb < q.run_nodes ? q.n_wide : kb + q.run_nodes;
        // ---- the chain at the start of the run: the wide nodes on the root path of the run's first node.  Wide nodes are a
        // suffix of a root path; the topmost one starts from its (narrow) parent's (blocks, masks).
        {
            for (uint32_t e = lane; e < q.chain_cap; e += WAVE) L.ch_node[e] = 0xFFFFFFFFu;
            const uint32_t d = q.seg_anc_n[run];
            LSum carry = lsum_none();
            bool started = false;
            for (uint32_t cb = 0; cb < d; cb += WAVE) {
                const uint32_t j = cb + lane;
                const bool on = j < d;
                const uint32_t a = on ? q.seg_anc[(size_t)run * q.chain_cap + j] : 0u;
                const unsigned long long wb = __ballot(on && iswide(a));
                if (!started && !wb) continue;
                const uint32_t t = started ? 0u : (uint32_t)__builtin_ctzll(wb);
                if (!started) {
                    const uint32_t at = bcast(a, t);
                    if (q.parent[at] >= 0) {
                        const uint32_t fb = q.fn_blk[at];
                        const ulonglong2 fm = q.fn_mask[at];
                        carry = lsum_of_fn(fb, fm);
                        if (lane == 0) {
                            if (carry.c >= 1u) { L.e_blk[0] = (uint16_t)(fb & 0xFFFFu); L.e_mask[0] = fm.x; }
                            if (carry.c >= 2u) { L.e_blk[1] = (uint16_t)(fb >> 16); L.e_mask[1] = fm.y; }
                        }
                    }
                    started = true;
                }
                const bool act = on && lane >= t;
                const uint32_t info = act ? q.p0_info[a] : 0u;
                const uint32_t np = info >> 16, b0 = info & 0xFFFFu;
                const unsigned long long m0 = act ? q.p0_mask[a] : 0ull;
                const uint32_t po = np > 1u ? q.pair_ofs[a] : 0u;
                uint32_t lb = b0;
                unsigned long long lm = m0;
                if (np > 1u) { lb = q.pair_blk[po + np - 2u]; lm = q.pair_mask[po + np - 2u]; }
                LSum S = np ? LSum{lm, np, b0, lb} : lsum_none();
                if (lane == t) S = lsum_merge(carry, S);
#pragma unroll
                for (int s = 1; s < WAVE; s <<= 1) {
                    const LSum o = lsum_shfl_up(S, s);
                    if (lane >= (uint32_t)s) S = lsum_merge(o, S);
                }
                LSum Pp = lsum_shfl_up(S, 1);
                if (lane == t) Pp = carry;
                const bool seam = act && np && Pp.c && Pp.lb == b0;
                const uint32_t idx0 = Pp.c - (seam ? 1u : 0u);
                if (act && np) {
                    if (!seam) { L.e_blk[idx0] = (uint16_t)b0; L.e_mask[idx0] = m0; }
                    for (uint32_t p = 1; p < np; ++p) { L.e_blk[idx0 + p] = q.pair_blk[po + p - 1u]; L.e_mask[idx0 + p] = q.pair_mask[po + p - 1u]; }
                }
                lds_sync();
                if (seam) atomicOr(&L.e_mask[idx0], m0);
                if (act) { L.ch_node[j] = a; L.ch_len[j] = (uint16_t)S.c; L.ch_last[j] = S.lm; }
                carry = lsum_shfl(S, WAVE - 1);
                lds_sync();
            }
            lds_sync();
        }
        // ---- the batches of the run
        for (uint32_t k0 = kb; k0 < ke; k0 += WAVE) {
            const uint32_t nv = ke - k0 < (uint32_t)WAVE ? ke - k0 : (uint32_t)WAVE;
            const bool valid = lane < nv;
            const uint32_t node = valid ? q.widx[k0 + lane] : 0u;
            const uint32_t nlv = valid ? q.nl[node] : 0u;
            const uint32_t wv = valid ? q.w[node] : 0u;
            const bool act = valid && wv != 0 && nlv >= 2u && node >= q.emit_lo && node < q.emit_hi;
            const int32_t par = valid ? q.parent[node] : -1;
            const uint32_t dep = valid ? (uint32_t)(q.dflag[node] & 0x7FFFu) : 0x7FFFu;
            const uint32_t info = valid ? q.p0_info[node] : 0u;
            const uint32_t np = info >> 16, b0 = info & 0xFFFFu;
            const unsigned long long m0 = valid ? q.p0_mask[node] : 0ull;
            // (every load that needs only `node` or `par` is requested here, whether its value will be used or not: the batch start is a chain of
            // dependent round trips — node, its fields, pair_ofs, the pairs; the parent's flags, then fn_* or wide_base — at four waves per SIMD.
            // pair_ofs of a node without further pairs and fn_* of a node with a wide parent were never written: what is read is not used.)
            const uint32_t po_raw = valid ? q.pair_ofs[node] : 0u;
            const uint32_t fnb_raw = valid ? q.fn_blk[node] : 0u;
            const ulonglong2 fnm_raw = valid ? q.fn_mask[node] : make_ulonglong2(0ull, 0ull);
            const uint32_t pw = (valid && par >= 0) ? (uint32_t)par >> 6 : 0u;
            const unsigned long long pwbits = (valid && par >= 0) ? q.widebits[pw] : 0ull;
            const uint32_t pwbase = (valid && par >= 0) ? q.wide_base[pw] : 0u;
            const uint32_t po = np > 1u ? po_raw : 0u;
            uint32_t lb = b0;
            unsigned long long lm = m0;
            if (np > 1u) { lb = q.pair_blk[po + np - 2u]; lm = q.pair_mask[po + np - 2u]; }
            // ---- where the parent's list comes from
            uint32_t base = WB_NONE, link = 0, pre0 = 0;
            LSum Pb = lsum_none();
            if (valid && par >= 0) {
                if (!((pwbits >> ((uint32_t)par & 63u)) & 1ull)) {
                    base = WB_FN;
                    Pb = lsum_of_fn(fnb_raw, fnm_raw);
                } else {
                    const uint32_t prank = pwbase + (uint32_t)__popcll(pwbits & ((1ull << ((uint32_t)par & 63u)) - 1ull));
                    if (prank >= k0) { base = WB_LANE; link = prank - k0; }
                    else {
                        base = WB_CHAIN; link = dep - 2u;
                        if (link < q.chain_cap && L.ch_node[link] == (uint32_t)par) {
                            const uint32_t cl = L.ch_len[link];
                            Pb = LSum{L.ch_last[link], cl, 0u, cl ? (uint32_t)L.e_blk[cl - 1u] : BLK_NONE};
                            pre0 = cl ? cl - 1u : 0u;              // all but the last entry of the chain node's list are taken from the chain list where they are
                        } else { base = WB_NONE; ++n_miss; }         // cannot happen (the engine treats it as an internal error)
                    }
                }
            }
            // ---- entries of every node's list: own stretch below the base, in-batch parents by pointer doubling
            LSum S = np ? LSum{lm, np, b0, lb} : lsum_none();
            if (base != WB_LANE) S = lsum_merge(Pb, S);
            // (pre: the shared prefix of the lane's TOPMOST in-batch ancestor — the lane whose base is not another lane)
            uint32_t pre = pre0, topl = lane;
            {
                int pl = base == WB_LANE ? (int)link : -1;
                while (__ballot(pl >= 0)) {
                    const int src = pl >= 0 ? pl : (int)lane;
                    const LSum o = lsum_shfl(S, src);
                    const int opl = __shfl(pl, src, WAVE);                // -1 once that lane's stretch reaches its base
                    const uint32_t opre = (uint32_t)__shfl((int)pre, src, WAVE);
                    const uint32_t otop = (uint32_t)__shfl((int)topl, src, WAVE);
                    if (pl >= 0) { S = lsum_merge(o, S); pl = opl; pre = opre; topl = otop; }
                }
            }
            // the (blocks, masks) the narrow kernel left for the topmost in-batch ancestor, fetched across the lanes NOW: a walk that ended in a
            // global load made every round of the batch wait for the record stores of the round before it (loads and stores return in order)
            const uint32_t tfnb = (uint32_t)__shfl((int)fnb_raw, (int)topl, WAVE);
            const unsigned long long tfnx = shfl64(fnm_raw.x, (int)topl), tfny = shfl64(fnm_raw.y, (int)topl);
            const uint32_t len = valid ? S.c : 0u;
            pre = valid ? pre : 0u;
            const uint32_t slen = len - pre;                               // entries of the lane's row
            L.own_m0[lane] = m0; L.own_po[lane] = po;
            {
                // the further own pairs, copied once: a node's pairs are read by every in-batch descendant's walk
                const uint32_t nx = np > 1u ? np - 1u : 0u;
                const uint32_t xincl = wave_incl_scan(nx, lane);
                const uint32_t x0 = xincl - nx;
                const bool fits = nx && xincl <= K1W_OXCAP;
                if (fits) for (uint32_t t = 0; t < nx; ++t) { L.ox_blk[x0 + t] = q.pair_blk[po + t]; L.ox_mask[x0 + t] = q.pair_mask[po + t]; }
                // (a walk over the in-batch parents reads one descriptor and one mask per node: the walks are chains of dependent LDS reads,
                // and five separate arrays made every step five of them — profiles/r05_j5: rows were 36 % of the kernel)
                L.own_desc[lane] = (unsigned long long)(b0 & 0xFFFFu) | ((unsigned long long)(link & 0xFFFFu) << 16) | ((unsigned long long)(fits ? x0 : 0xFFFFu) << 32) |
                                   ((unsigned long long)(np & 0x3FFFu) << 48) | ((unsigned long long)base << 62);
            }
            lds_sync();
            // ---- rows, as many lanes at a time as the arena holds, and their records.  Only the lanes that emit need a row, and
            // the last one (its list becomes the chain list).
            const bool need = valid && (act || lane == nv - 1u);
            uint32_t fin = 0, last_start = 0;
            while (fin < nv) {
                const uint32_t c = (lane >= fin && need) ? slen : 0u;
                const uint32_t incl = wave_incl_scan(c, lane);
                const unsigned long long over = __ballot(incl > q.arena_cap);
                uint32_t hi = over ? (uint32_t)__builtin_ctzll(over) : (uint32_t)WAVE;
                if (hi <= fin) { if (lane == 0) atomicOr(&q.pool.counters[KCTR_LIST_OVERFLOW], 1u); hi = fin + 1u; }     // a single list longer than the arena: cannot happen
                const bool on = need && lane >= fin && lane < hi && incl <= q.arena_cap;
                const uint32_t start = incl - c;
                if (on) {
                    // fill the row right to left: own pairs, the in-batch parents' pairs, then the base
                    // (the entry being built stays in registers until the next block starts: no read-modify-write of the row)
                    uint32_t pos = start + slen, cur = 0xFFFFFFFFu;
                    unsigned long long curm = 0;
                    auto rflush = [&]() { if (cur != 0xFFFFFFFFu) { --pos; L.ent_blk[pos] = (uint16_t)cur; L.ent_mask[pos] = curm; } };
                    auto rpush = [&](uint32_t blk, unsigned long long mask) {
                        if (blk == cur) curm |= mask;
                        else { rflush(); cur = blk; curm = mask; }
                    };
                    uint32_t y = lane;
                    for (;;) {
                        const unsigned long long dsc = L.own_desc[y];
                        const unsigned long long ym0 = L.own_m0[y];
                        const uint32_t ynp = (uint32_t)(dsc >> 48) & 0x3FFFu, yb = (uint32_t)(dsc >> 62), ylink = (uint32_t)(dsc >> 16) & 0xFFFFu;
                        if (ynp > 1u) {
                            const uint32_t yox = (uint32_t)(dsc >> 32) & 0xFFFFu;
                            const uint32_t ynx = ynp - 1u;                       // (a node has fewer than 2897 pairs: fewer than 2^22 block pairs)
                            if (yox != 0xFFFFu) for (uint32_t t = ynx; t-- > 0u;) rpush(L.ox_blk[yox + t], L.ox_mask[yox + t]);
                            else {
                                const uint32_t ypo = L.own_po[y];
                                for (uint32_t t = ynx; t-- > 0u;) rpush(q.pair_blk[ypo + t], q.pair_mask[ypo + t]);
                            }
                        }
                        if (ynp) rpush((uint32_t)dsc & 0xFFFFu, ym0);
                        if (yb == WB_LANE) { y = ylink; continue; }
                        if (yb == WB_FN) {                       // (y is the lane's topmost in-batch ancestor: topl)
                            if ((tfnb >> 16) != BLK_NONE) rpush(tfnb >> 16, tfny);
                            if ((tfnb & 0xFFFFu) != BLK_NONE) rpush(tfnb & 0xFFFFu, tfnx);
                        } else if (yb == WB_CHAIN) {
                            // the chain node's last entry (its mask may be the node's own: a deeper chain node can have more ids in that
                            // block); the entries before it stay in the chain list, shared by every lane that hangs below this node
                            const uint32_t cs = ylink;
                            const uint32_t cl = L.ch_len[cs];
                            if (cl) rpush(L.e_blk[cl - 1u], L.ch_last[cs]);
                        }
                        break;
                    }
                    rflush();
                }
                L.st_start[lane] = (uint16_t)start; L.st_pre[lane] = (uint16_t)pre;
                if (lane == nv - 1u && on) last_start = start;
                lds_sync();
                emit(on && act, len, wv);
                fin = hi;
            }
            // ---- the chain for the next batches: the lanes on the root path of the batch's last node, and that node's list
            {
                uint32_t mdep = dep;
#pragma unroll
                for (int sft = 1; sft < WAVE; sft <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_down((int)mdep, sft, WAVE);
                    if (lane + (uint32_t)sft < (uint32_t)WAVE) mdep = o < mdep ? o : mdep;
                }
                uint32_t later = (uint32_t)__shfl_down((int)mdep, 1, WAVE);
                if (lane == (uint32_t)WAVE - 1u) later = 0xFFFFFFFFu;
                if (valid && dep < later && dep - 1u < q.chain_cap) { L.ch_node[dep - 1u] = node; L.ch_len[dep - 1u] = (uint16_t)len; L.ch_last[dep - 1u] = S.lm; }
                // the last node's list becomes the chain list: its first entries are there already, its row follows them
                const uint32_t ll = bcast(slen, nv - 1u), ls = bcast(last_start, nv - 1u), lp = bcast(pre, nv - 1u);
                for (uint32_t e = lane; e < ll; e += WAVE) { L.e_blk[lp + e] = L.ent_blk[ls + e]; L.e_mask[lp + e] = L.ent_mask[ls + e]; }
                lds_sync();
            }
        }
    }
    arena_finish(A, q.pool, lane);
    if (q.n_rows) rowtab_finish(RT, q.pool, lane);
    if (lane == 0 && n_miss) atomicAdd(&q.pool.counters[KCTR_SLOW], n_miss);
    if (lane == 0 && l2_nodes) atomicAdd(&q.pool.counters[KCTR_L2_NODES], l2_nodes);
}

// ------------------------------------------------------------------------------------------
// K2: records -> matrix
// ------------------------------------------------------------------------------------------
struct K2Item {
    uint32_t X, Y, count;                  // block pair, wave steps (64 records each) of this stream in the window
    const uint32_t* ids;                   // chunk mode: chunk ids of the run (LDS), CH_REC / 64 steps per chunk
    const uint32_t* fills;                 //             records in every chunk (LDS)
    const WideRec* srec;                   // sorted mode: the run's records, n_rec of them, contiguous
    const uint32_t* skey;                  //              their key words (weight digits)
    uint32_t kbits, dbits;
    uint32_t pshift;                       // sorted mode, != 0: packed records (PoolView::pshift): weight digit in the column word, no key words
    uint32_t rect_cols;                    // 0: lower-triangular matrix of N samples; else a dense n_rows x rect_cols matrix (db2db), N = n_rows
    uint32_t n_rec;
    const unsigned char* rec;              // record pool
    const uint32_t* recw;
};
constexpr uint32_t CH_STEPS = CH_REC / 64u;
// the 64 records of step st, one per lane (lanes beyond the fill get an empty record)
template <bool SORTED>
__device__ __forceinline__ void k2_fetch(const K2Item& it, bool diag, uint32_t st, uint32_t lane, bool weighted, unsigned long long& R,
                                         unsigned long long& Cc, uint32_t& W) {
    R = 0; Cc = 0; W = 0;
    if (st >= it.count) return;
    if (SORTED) {
        const uint32_t p = st * 64u + lane;
        if (p < it.n_rec) {
            const WideRec r = it.srec[p];
            R = r.rows; Cc = r.cols;
            if (it.pshift) {
                const uint32_t f = (uint32_t)(r.cols >> it.pshift), pd = 64u - it.pshift - 2u < 16u ? 64u - it.pshift - 2u : 16u;
                const uint32_t sh = (f >> pd) * pd;
                Cc = r.cols & ((1ull << it.pshift) - 1ull);
                W = sh < 32u ? (f & ((1u << pd) - 1u)) << sh : 0u;
            } else
            W = weighted ? wide_weight(it.skey[p], it.kbits, it.dbits) : 1u;
        }
    } else {
        const uint32_t ci = st / CH_STEPS, j = (st % CH_STEPS) * 64u + lane;
        if (j < it.fills[ci]) {
            const uint32_t id = it.ids[ci];
            const unsigned char* base = it.rec + ((size_t)id << (CH_SHIFT + 4));
            if (diag) { R = ((const unsigned long long*)base)[j]; Cc = R; }
            else { const ulonglong2 rc = ((const ulonglong2*)base)[j]; R = rc.x; Cc = rc.y; }
            W = weighted ? it.recw[((size_t)id << CH_SHIFT) + j] : 1u;
        }
    }
}

// The accumulation of a run on the matrix cores: over the 64 records of a step,
//     cell(r, c) += sum_k  w_k [r in rows_k] * [c in cols_k]    =  (A B)(r, c),   A = 64 x 64 int8 (rows x records, weighted),
//                                                                                B = 64 x 64 int8 (records x cols, 0/1)
// as eight v_mfma_i32_32x32x32_i8 (operand layout probed in profiles/r01_mfma_i8_layout_probe.hip: lane l holds
// A[l & 31][16 (l >> 5) + j], B[16 (l >> 5) + j][l & 31], j < 16; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)).
// The bit matrices R^T / C^T of the step are parked in LDS; a lane turns 16 of their bits into 16 operand bytes with two
// reads of a 256-entry byte-spreading table and ANDs the weights in.  An int8 operand holds a weight below 128: a larger
// weight is split into base-128 digits, the run is accumulated once per digit that occurs (`digit`), and the digit's tile is
// merged shifted left by 7 * digit — exact in the matrix's uint32 wrap-around arithmetic.  Almost every weight is below 128
// (99.98 % at the benchmark database), so nearly every run takes one pass.  Returns (to every thread) the OR of the weights.

template <bool DIAG, bool SORTED>
__device__ __forceinline__ uint32_t k2_apply_mfma(const K2Item& it, uint32_t digit, uint32_t* acc,
                                                  unsigned char (*wbuf)[64], const unsigned long long* lut_ff, const unsigned long long* lut_01) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t half = lane >> 5, l31 = lane & 31u;
    k2_v16i c00 = {}, c01 = {}, c10 = {}, c11 = {};
    auto spread = [&](unsigned long long word, uint32_t shift, const unsigned long long* lut) -> k2_v4i {
        const uint32_t f = (uint32_t)(word >> shift) & 0xFFFFu;
        k2_v4i r;
        const unsigned long long lo = lut[f & 0xFFu], hi = lut[f >> 8];
        r[0] = (int)(uint32_t)lo; r[1] = (int)(uint32_t)(lo >> 32); r[2] = (int)(uint32_t)hi; r[3] = (int)(uint32_t)(hi >> 32);
        return r;
    };
    unsigned long long nR = 0, nC = 0;
    uint32_t nW = 0, wor = 0;
    const TrConst trc = tr_const(lane);
    // this wave's steps: every fourth one; in chunk mode the steps past a chunk's fill are skipped without a fetch
    auto next_step = [&](uint32_t st) -> uint32_t {
        if (!SORTED) while (st < it.count && (st % CH_STEPS) * 64u >= it.fills[st / CH_STEPS]) st += 4;
        return st;
    };
    uint32_t cur = next_step(wave);
    k2_fetch<SORTED>(it, DIAG, cur, lane, true, nR, nC, nW);
    while (cur < it.count) {
        const unsigned long long R = nR, C = nC;
        const uint32_t W = nW;
        cur = next_step(cur + 4);
        k2_fetch<SORTED>(it, DIAG, cur, lane, true, nR, nC, nW);
        wor |= W;
        // lane r now holds row r of the step's bit matrices R^T / C^T (bit k <=> record k has row / column r).  The operand layout wants
        // in every lane the rows l31 and 32 + l31: the lower half of the wave keeps its own word and fetches the upper half's, and
        // the other way round — one v_permlane32_swap per 32-bit word, nothing parked in LDS (round 3 wrote the 64 words to LDS and read
        // four back per lane; with the byte-spreading table that made 25 LDS instructions per step and a third of the LDS cycles lost to
        // bank conflicts).
        const unsigned long long Ct = transpose64(C, trc);
        const unsigned long long Rt = DIAG ? Ct : transpose64(R, trc);           // on the diagonal rows == cols
        auto halves = [&](unsigned long long w, unsigned long long& w0, unsigned long long& w1) {
            const auto a = __builtin_amdgcn_permlane32_swap((uint32_t)w, (uint32_t)w, false, false);                 // [0]: the lower half's words everywhere, [1]: the upper half's
            const auto b = __builtin_amdgcn_permlane32_swap((uint32_t)(w >> 32), (uint32_t)(w >> 32), false, false);
            w0 = ((unsigned long long)b[0] << 32) | a[0];
            w1 = ((unsigned long long)b[1] << 32) | a[1];
        };
        unsigned long long ra0, ra1, cb0, cb1;
        halves(Rt, ra0, ra1);
        if (DIAG) { cb0 = ra0; cb1 = ra1; } else halves(Ct, cb0, cb1);
        wbuf[wave][lane] = (unsigned char)((W >> (7u * digit)) & 127u);
        lds_sync();
#pragma unroll
        for (uint32_t kh = 0; kh < 2; ++kh) {
            const uint32_t shift = 32u * kh + 16u * half;            // records 32 kh + 16 half .. + 15 of the step
            k2_v4i a0 = spread(ra0, shift, lut_ff), a1 = spread(ra1, shift, lut_ff);
            const k2_v4i b0 = spread(cb0, shift, lut_01), b1 = spread(cb1, shift, lut_01);
            const k2_v4i wv = *(const k2_v4i*)(wbuf[wave] + shift);
            a0 &= wv; a1 &= wv;
            c00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c11, 0, 0, 0);
        }
        lds_sync();
    }
    // merge the four waves' tiles through the LDS block (on the diagonal only c < r)
    const uint32_t sh = 7u * digit;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t row0 = (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * half;
        const uint32_t v00 = (uint32_t)c00[r] << sh, v01 = (uint32_t)c01[r] << sh, v10 = (uint32_t)c10[r] << sh, v11 = (uint32_t)c11[r] << sh;
        if (v00 && (!DIAG || l31 < row0)) atomicAdd(&acc[row0 * 64 + l31], v00);
        if (v01 && (!DIAG || 32u + l31 < row0)) atomicAdd(&acc[row0 * 64 + 32u + l31], v01);
        if (v10 && (!DIAG || l31 < 32u + row0)) atomicAdd(&acc[(32u + row0) * 64 + l31], v10);
        if (v11 && (!DIAG || l31 < row0)) atomicAdd(&acc[(32u + row0) * 64 + 32u + l31], v11);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wor |= (uint32_t)__shfl_xor((int)wor, d, WAVE);
    return wor;
}

// one run (the records of one block pair inside a window): accumulate, once per weight digit that occurs, and flush the tile
template <bool SORTED>
__device__ __forceinline__ void k2_run(const K2Item& it, uint32_t* acc, uint32_t* wor_sh,
                                       unsigned char (*wbuf)[64], const unsigned long long* lut_ff, const unsigned long long* lut_01,
                                       uint32_t* __restrict__ M, uint32_t N, uint32_t bwidth, unsigned char* __restrict__ touched) {
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) acc[k] = 0;
    if (threadIdx.x == 0) { *wor_sh = 0; if (touched && !it.rect_cols) touched[tri32(it.X) + it.Y] = 1; }      // (the tile gets something: all2all-sp scans only such tiles)
    __syncthreads();
    for (uint32_t digit = 0; digit < 5; ++digit) {
        const uint32_t wor = (it.X == it.Y && !it.rect_cols) ? k2_apply_mfma<true, SORTED>(it, digit, acc, wbuf, lut_ff, lut_01)
                                          : k2_apply_mfma<false, SORTED>(it, digit, acc, wbuf, lut_ff, lut_01);
        if ((threadIdx.x & 63u) == 0 && wor) atomicOr(wor_sh, wor);
        __syncthreads();
        if (digit == 4u || (*wor_sh >> (7u * (digit + 1u))) == 0) break;           // no weight has a higher digit (digit 4 is the last: a shift by 35 is not defined)
        __syncthreads();
    }
    // one HBM atomic per non-zero cell of the block
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) {
        const uint32_t v = acc[k];
        if (!v) continue;
        const uint64_t row = (uint64_t)it.X * bwidth + (k >> 6), col = (uint64_t)it.Y * bwidth + (k & 63u);
        if (it.rect_cols) { if (row < N && col < it.rect_cols) atomicAdd(&M[row * it.rect_cols + col], v); }
        else if (row < N && col < row) atomicAdd(&M[tri64(row) + col], v);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// second level above the block records: bitmaps -> rank directories -> lists -> tile joins (L2View; profiles/r04_join_apply_probe.hip)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t stream_row(uint32_t s);
// one workgroup per block: exclusive prefix popcounts of its W bitmap words, and the length of its list
__global__ __launch_bounds__(256) void l2_ranks_kernel(const unsigned long long* __restrict__ B, uint32_t W, uint32_t* __restrict__ R, uint32_t* __restrict__ len) {
    __shared__ uint32_t part[256];
    const unsigned long long* b = B + (size_t)blockIdx.x * W;
    uint32_t* r = R + (size_t)blockIdx.x * W;
    const uint32_t per = (W + 255u) / 256u, lo = threadIdx.x * per < W ? threadIdx.x * per : W, hi = lo + per < W ? lo + per : W;
    uint32_t s = 0;
    for (uint32_t w = lo; w < hi; ++w) s += (uint32_t)__popcll(b[w]);
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t t = 0; t < 256; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; } len[blockIdx.x] = run; }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t w = lo; w < hi; ++w) { r[w] = run; run += (uint32_t)__popcll(b[w]); }
}
// list offsets; and loff[NB + 1] = bitmap words in use (the highest node index handed out, from the cursors' interleaved grabs)
__global__ void l2_offsets_kernel(const uint32_t* __restrict__ len, uint32_t NB, const uint32_t* __restrict__ cursors, uint32_t W, uint32_t* __restrict__ loff) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t x = 0; x < NB; ++x) { loff[x] = run; run += len[x]; }
        loff[NB] = run;
        uint32_t top = 0;
        for (uint32_t k = 0; k < L2_SUB; ++k) {
            const uint32_t c = cursors[k * 16u];                 // indices this cursor handed out (a multiple of the grab)
            if (c) { const uint32_t end = ((c / L2_NODE_GRAB - 1u) * L2_SUB + k + 1u) * L2_NODE_GRAB; top = end > top ? end : top; }
        }
        const uint32_t wu = (top + 63u) / 64u;
        loff[NB + 1] = wu < W ? wu : W;
    }
}
// an entry's place in its block's list is its node's rank in the block's bitmap: no sort
__global__ void l2_lists_kernel(const uint32_t* __restrict__ ent_g, const uint16_t* __restrict__ ent_blk, const unsigned long long* __restrict__ ent_mask,
                                const uint32_t* __restrict__ node_w, uint32_t n, uint32_t W, const unsigned long long* __restrict__ B, const uint32_t* __restrict__ R,
                                const uint32_t* __restrict__ loff, unsigned long long* __restrict__ L, uint32_t* __restrict__ Wt, const uint32_t* __restrict__ cursors) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // (the entry pool is sized with room to spare — four fifths of its slots were never handed out at 10 000 samples: a slot beyond its
    // sub-range's cursor is not even looked at)
    const uint32_t per_sub = n / L2_SUB, sub = i / per_sub;
    if (sub < L2_SUB && i - sub * per_sub >= cursors[(L2_SUB + sub) * 16u]) return;
    const uint32_t X = ent_blk[i];
    if (X == 0xFFFFu) return;                                 // a slot no wave wrote (the rest of a grab)
    const uint32_t g = ent_g[i];
    const unsigned long long word = B[(size_t)X * W + (g >> 6)];
    const uint32_t pos = loff[X] + R[(size_t)X * W + (g >> 6)] + (uint32_t)__popcll(word & ((1ull << (g & 63u)) - 1ull));
    L[pos] = ent_mask[i];
    Wt[pos] = node_w[g];
}
// One workgroup of four waves per tile (X, Y), X >= Y.  A lane ANDs one word of the two bitmaps (64 nodes), ranks every match in both lists
// and the wave queues (rank in L_X, rank in L_Y) in LDS, at most eight per lane and round; every 64 queued matches are one apply step — the
// masks and the weight gathered (the next step's while this one is applied), two bit transposes, byte spreading, eight MFMAs, as
// k2_apply_mfma does for 64 sorted records.  The queue's tail is drained by a last, empty round of the same loop (one call site of the step:
// inlined twice, the kernel ran at two waves per SIMD).  Weights of 128 and more: one join per base-128 digit that occurs.
constexpr uint32_t L2_WAVES = 4, L2_QCAP = 576;
__global__ __launch_bounds__(64 * L2_WAVES) __attribute__((amdgpu_waves_per_eu(3, 8)))
void l2_join_apply_kernel(const unsigned long long* __restrict__ B, const uint32_t* __restrict__ R, const unsigned long long* __restrict__ L, const uint32_t* __restrict__ Wt,
                          const uint32_t* __restrict__ loff, uint32_t w_stride, uint32_t nb_blocks, uint32_t* __restrict__ M, uint32_t N, uint32_t bwidth,
                          unsigned char* __restrict__ touched) {
    uint32_t W;
    __shared__ uint32_t q[L2_WAVES][L2_QCAP];
    __shared__ uint32_t q2[L2_WAVES][L2_QCAP];
    __shared__ unsigned long long lut_ff[256], lut_01[256];
    __shared__ uint32_t acc[64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[L2_WAVES][64];
    __shared__ uint32_t wor_sh;
    const uint32_t t = blockIdx.x;
    const uint32_t X = stream_row(t), Y = t - tri32(X);
    if (loff[X + 1] == loff[X] || loff[Y + 1] == loff[Y]) return;       // (uniform over the workgroup)
    W = loff[nb_blocks + 1u];                       // the bitmap words in use (l2_offsets_kernel); rows of the bitmaps stay w_stride apart
    const bool diag = X == Y;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t half = lane >> 5, l31 = lane & 31u;
    {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((threadIdx.x >> i) & 1u) ? 0xFFull << (8 * i) : 0ull;
        lut_ff[threadIdx.x] = v;
        lut_01[threadIdx.x] = v & 0x0101010101010101ull;
        for (uint32_t k = threadIdx.x; k < 64 * 64; k += 64 * L2_WAVES) acc[k] = 0;
        if (threadIdx.x == 0) wor_sh = 0;
        __syncthreads();
    }
    const unsigned long long* bx = B + (size_t)X * w_stride;
    const unsigned long long* by = B + (size_t)Y * w_stride;
    const uint32_t* rx = R + (size_t)X * w_stride;
    const uint32_t* ry = R + (size_t)Y * w_stride;
    const unsigned long long* lx = L + loff[X];
    const unsigned long long* ly = L + loff[Y];
    const uint32_t* wxl = Wt + loff[X];
    const TrConst trc = tr_const(lane);
    auto spread = [&](unsigned long long word, uint32_t shift, const unsigned long long* lut) -> k2_v4i {
        const uint32_t f = (uint32_t)(word >> shift) & 0xFFFFu;
        const unsigned long long lo = lut[f & 0xFFu], hi = lut[f >> 8];
        k2_v4i r;
        r[0] = (int)(uint32_t)lo; r[1] = (int)(uint32_t)(lo >> 32); r[2] = (int)(uint32_t)hi; r[3] = (int)(uint32_t)(hi >> 32);
        return r;
    };
    auto halves = [&](unsigned long long w, unsigned long long& w0, unsigned long long& w1) {
        const auto a = __builtin_amdgcn_permlane32_swap((uint32_t)w, (uint32_t)w, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap((uint32_t)(w >> 32), (uint32_t)(w >> 32), false, false);
        w0 = ((unsigned long long)b[0] << 32) | a[0];
        w1 = ((unsigned long long)b[1] << 32) | a[1];
    };
    for (uint32_t digit = 0; digit < 5; ++digit) {
        k2_v16i c00 = {}, c01 = {}, c10 = {}, c11 = {};
        uint32_t qn = 0, wor = 0;                             // qn wave-uniform
        auto fetch = [&](uint32_t base, uint32_t n, unsigned long long& Rm, unsigned long long& Cm, uint32_t& Wg) {
            Rm = 0; Cm = 0; Wg = 0;
            if (lane < n) {
                const uint32_t a = q[wave][base + lane], b = q2[wave][base + lane];
                Rm = lx[a]; Cm = ly[b]; Wg = wxl[a];
            }
        };
        auto apply = [&](unsigned long long Rm, unsigned long long Cm, uint32_t Wg) {
            wor |= Wg;
            const unsigned long long Ct = transpose64(Cm, trc);
            const unsigned long long Rt = transpose64(Rm, trc);
            unsigned long long ra0, ra1, cb0, cb1;
            halves(Rt, ra0, ra1);
            halves(Ct, cb0, cb1);
            wbuf[wave][lane] = (unsigned char)((Wg >> (7u * digit)) & 127u);
            lds_sync();
#pragma unroll
            for (uint32_t kh = 0; kh < 2; ++kh) {
                const uint32_t shift = 32u * kh + 16u * half;
                k2_v4i a0 = spread(ra0, shift, lut_ff), a1 = spread(ra1, shift, lut_ff);
                const k2_v4i b0 = spread(cb0, shift, lut_01), b1 = spread(cb1, shift, lut_01);
                const k2_v4i wv = *(const k2_v4i*)(wbuf[wave] + shift);
                a0 &= wv; a1 &= wv;
                c00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c11, 0, 0, 0);
            }
            lds_sync();
        };
        auto drain_steps = [&](bool all) {                    // the full steps of the queue; all: the partly filled last one too
            uint32_t head = 0;
            unsigned long long nR = 0, nC = 0; uint32_t nW = 0;
            auto avail = [&]() -> uint32_t { const uint32_t r = qn - head; return r >= 64u ? 64u : (all ? r : 0u); };
            if (avail()) fetch(0u, avail(), nR, nC, nW);
            while (avail()) {
                const unsigned long long Rm = nR, Cm = nC;
                const uint32_t Wg = nW;
                head += avail();
                if (avail()) fetch(head, avail(), nR, nC, nW);
                apply(Rm, Cm, Wg);
            }
            if (head) {
                const uint32_t rest = qn - head;
                uint32_t a = 0, b = 0;
                if (lane < rest) { a = q[wave][head + lane]; b = q2[wave][head + lane]; }
                lds_sync();
                if (lane < rest) { q[wave][lane] = a; q2[wave][lane] = b; }
                lds_sync();
                qn = rest;
            }
        };
        unsigned long long pwx = 0, pwy = 0; uint32_t pbx = 0, pby = 0;
        auto load_words = [&](uint32_t w0, unsigned long long& wx, unsigned long long& wy, uint32_t& bxr, uint32_t& byr) {
            const uint32_t w = w0 + lane;
            wx = 0; wy = 0; bxr = 0; byr = 0;
            if (w0 < W && w < W) { wx = bx[w]; wy = by[w]; bxr = rx[w]; byr = ry[w]; }
        };
        load_words(wave * 64u, pwx, pwy, pbx, pby);
        for (uint32_t w0 = wave * 64u;; w0 += 64u * L2_WAVES) {
            const bool last = w0 >= W;                        // one round past the bitmap: nothing to match, the queue's tail is drained
            const unsigned long long wx = pwx, wy = pwy;
            const uint32_t bxr = pbx, byr = pby;
            load_words(w0 + 64u * L2_WAVES, pwx, pwy, pbx, pby);
            unsigned long long m = wx & wy;
            do {
                const uint32_t left = (uint32_t)__popcll(m);
                const uint32_t k = left < 8u ? left : 8u;
                const uint32_t incl = wave_incl_scan(k, lane);
                const uint32_t total = bcast(incl, WAVE - 1);
                uint32_t pos = qn + incl - k;
                for (uint32_t j = 0; j < k; ++j) {
                    const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    const unsigned long long below = (1ull << bit) - 1ull;
                    q[wave][pos] = bxr + (uint32_t)__popcll(wx & below); q2[wave][pos] = byr + (uint32_t)__popcll(wy & below);
                    ++pos;
                }
                qn += total;
                lds_sync();
                drain_steps(last);
            } while (__ballot(m != 0ull));
            if (last) break;
        }
        // the four waves' tiles merged in LDS, shifted by the digit's weight (on the diagonal only c < r)
        const uint32_t sh = 7u * digit;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row0 = (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * half;
            const uint32_t v00 = (uint32_t)c00[r] << sh, v01 = (uint32_t)c01[r] << sh, v10 = (uint32_t)c10[r] << sh, v11 = (uint32_t)c11[r] << sh;
            if (v00 && (!diag || l31 < row0)) atomicAdd(&acc[row0 * 64 + l31], v00);
            if (v01 && (!diag || 32u + l31 < row0)) atomicAdd(&acc[row0 * 64 + 32u + l31], v01);
            if (v10 && (!diag || l31 < 32u + row0)) atomicAdd(&acc[(32u + row0) * 64 + l31], v10);
            if (v11 && (!diag || l31 < row0)) atomicAdd(&acc[(32u + row0) * 64 + 32u + l31], v11);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wor |= (uint32_t)__shfl_xor((int)wor, d, WAVE);
        if (lane == 0 && wor) atomicOr(&wor_sh, wor);
        __syncthreads();
        if (digit == 4u || (wor_sh >> (7u * (digit + 1u))) == 0) break;      // no weight has a higher digit (digit 4 is the last: a shift by 35 is not defined)
        __syncthreads();
    }
    // one HBM atomic per non-zero cell of the tile (the apply kernels add into the same matrix)
    if (threadIdx.x == 0 && touched) touched[t] = 1;
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 64 * L2_WAVES) {
        const uint32_t v = acc[k];
        if (!v) continue;
        const uint64_t row = (uint64_t)X * bwidth + (k >> 6), col = (uint64_t)Y * bwidth + (k & 63u);
        if (row < N && col < row) atomicAdd(&M[tri64(row) + col], v);
    }
}

constexpr uint32_t K2_WIN = 32;            // sorted chunks per workgroup at most; the launch picks 16 (few streams: measured better) or 32
constexpr int K2A_MIN_WAVES = 3;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K2A_MIN_WAVES, 8))) void k2_apply_kernel(const unsigned char* __restrict__ rec, const uint32_t* __restrict__ recw,
                                                       const uint32_t* __restrict__ sorted_key, const uint32_t* __restrict__ sorted_id,
                                                       const uint32_t* __restrict__ chunk_fill, uint32_t n_states, const uint32_t* __restrict__ n_chunks_ptr,
                                                       uint32_t* __restrict__ M, uint32_t N, uint32_t bwidth, uint32_t win, unsigned char* __restrict__ touched) {
    __shared__ uint32_t acc[64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[4][64];
    __shared__ unsigned long long lut_ff[256], lut_01[256];       // byte b -> its 8 bits spread over 8 bytes (0xFF / 0x01 where set)
    __shared__ uint32_t s_key[K2_WIN], s_id[K2_WIN], s_fill[K2_WIN];
    __shared__ uint32_t wor_sh;
    // the window: K2_WIN chunks of the grouped chunk table; its head [0, *n_chunks_ptr) holds the stream chunks
    if (threadIdx.x < win) {
        const uint32_t j = blockIdx.x * win + threadIdx.x;
        const uint32_t key = j < *n_chunks_ptr ? sorted_key[j] : n_states;
        const uint32_t id = key < n_states ? sorted_id[j] : 0u;
        s_key[threadIdx.x] = key; s_id[threadIdx.x] = id; s_fill[threadIdx.x] = key < n_states ? chunk_fill[id] : 0u;
    }
    {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((threadIdx.x >> i) & 1u) ? 0xFFull << (8 * i) : 0ull;
        lut_ff[threadIdx.x] = v;
        lut_01[threadIdx.x] = v & 0x0101010101010101ull;
    }
    __syncthreads();
    uint32_t a = 0;
    while (a < win && s_key[a] < n_states) {
        // one run of the window = the chunks of one stream
        const uint32_t key = s_key[a];
        uint32_t b = a + 1;
        while (b < win && s_key[b] == key) ++b;
        K2Item it;
        {
            const uint32_t bucket = key;
            uint32_t X = (uint32_t)((__fsqrt_rn(8.0f * (float)bucket + 1.0f) - 1.0f) * 0.5f);
            while (tri32(X) > bucket) --X;
            while (tri32(X + 1u) <= bucket) ++X;
            it.X = X; it.Y = bucket - tri32(X);
            it.count = (b - a) * CH_STEPS; it.ids = s_id + a; it.fills = s_fill + a; it.srec = nullptr; it.n_rec = 0; it.rec = rec; it.recw = recw; it.rect_cols = 0; it.pshift = 0;
        }
        k2_run<false>(it, acc, &wor_sh, wbuf, lut_ff, lut_01, M, N, bwidth, touched);
        a = b;
    }
}

// opened chunks = the head of the sorted chunk table
__global__ void count_chunks_kernel(const uint32_t* __restrict__ sorted_key, uint32_t pool_cap, uint32_t n_states, uint32_t* __restrict__ counters) {
    uint32_t lo = 0, hi = pool_cap;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted_key[mid] < n_states) lo = mid + 1; else hi = mid; }
    counters[KCTR_CHUNKS] = lo;
}
// slots of the wide pool in use = the busiest sub-pool's share of all
// chunks of the chunk pool in use = the busiest sub-pool's share of all (one wave: a maximum over the 256 cursors)
__global__ void pool_used_kernel(const uint32_t* __restrict__ sub_cursor, const uint32_t* __restrict__ direct_ctr, uint32_t* __restrict__ counters) {
    uint32_t mx = 0, nd = 0;
    for (uint32_t p = threadIdx.x; p < KMDB_SUBPOOLS; p += 64u) { mx = max(mx, sub_cursor[p * 16u]); nd += direct_ctr[p * 16u]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, WAVE)); nd += (uint32_t)__shfl_xor((int)nd, d, WAVE); }
    if (threadIdx.x == 0) { counters[KCTR_POOL_USED] = mx * KMDB_SUBPOOLS; counters[KCTR_DIRECT] = nd; }
}
__global__ void count_raw_kernel(const uint32_t* __restrict__ wsub_cursor, uint32_t* __restrict__ counters) {
    uint32_t mx = 0;
    for (uint32_t p = 0; p < KMDB_SUBPOOLS; ++p) mx = max(mx, wsub_cursor[p * 16u]);
    counters[KCTR_RAW] = mx * KMDB_SUBPOOLS;                    // chunk ids below this bound cover every written record
}

// Counting sort of the wide pool by stream: per-workgroup histograms in LDS (no device atomics), one exclusive scan over the
// table of counts, then every workgroup moves its records to its own range of every stream.  The order inside a stream is
// arbitrary (uint32 adds commute).  Two reads of the keys, one read and one write of the records per pass.
// Up to CS_MAX_KEYS streams: one pass, bins = streams.  More (10 000 samples at width 50 are 20 100 streams): two passes —
// first by block row X (bins = rows), then every row by itself, cut into chunks of whole tiles, bins = the row's X + 1 streams.
// A scatter straight to 20 100 destinations is bound by the number of write requests (measured: 25.8 ms for 495 M records, a
// tile holds less than one record per stream); the two staged passes write bursts (6.3 ms + the same again).
// More than CS_MAX_ROWS block rows (25 600 samples at width 50): rocprim's radix sort — a tile of 1024 records then holds about one
// record per bin of its row, the staged passes stop writing bursts (measured at 782 rows: 27.7 ms against 22.6 with rocprim).
constexpr uint32_t CS_MAX_KEYS = 2048, CS_BLOCKS = 2048, CS_MAX_ROWS = 512;
constexpr uint32_t CS_BLOCKS_ONE = 3072;   // workgroups of the one-pass sort (two full rounds on the chip measured better than 2048; the two-pass sort keeps 2048)
// block row X of stream s = tri32(X) + Y
__device__ __forceinline__ uint32_t stream_row(uint32_t s) {
    uint32_t X = (uint32_t)((__fsqrt_rn(8.0f * (float)s + 1.0f) - 1.0f) * 0.5f);
    while (tri32(X) > s) --X;
    while (tri32(X + 1u) <= s) ++X;
    return X;
}
// the second pass's view of the row-sorted records: where every row starts, its first workgroup, its first table entry
struct CsRows {
    const uint32_t* row_start;      // [NB + 1]
    const uint32_t* row_blk;        // [NB + 1] workgroups before the row (a row of n records has ceil(n / chunk))
    const uint32_t* row_tab;        // [NB + 1] table entries before the row (its (X + 1) streams x its workgroups)
    uint32_t NB, chunk;
};
enum { CS_BY_STREAM = 0, CS_BY_ROW = 1, CS_IN_ROW = 2 };
// what one workgroup sorts: records [lo, hi), bin of a key = row(key) or key - sub, count / offset of bin k at table[tab + k * stride]
struct CsJob { uint32_t lo, hi, n_bins, sub, tab, stride; };
__device__ __forceinline__ CsJob cs_job(int mode, uint32_t n, uint32_t n_keys, const CsRows& rows) {
    CsJob j{0u, 0u, n_keys, 0u, blockIdx.x, gridDim.x};
    if (mode != CS_IN_ROW) {
        // equal shares of whole 1024-record tiles; the shares in XCD-contiguous order (the grids are multiples of 8)
        const uint32_t vb = (gridDim.x & 7u) ? blockIdx.x : xcd_contiguous(blockIdx.x, gridDim.x);
        j.tab = vb;
        const uint32_t per_block = ((n + gridDim.x - 1u) / gridDim.x + 1023u) / 1024u * 1024u;
        const uint32_t lo = vb * per_block;
        if (lo < n) { j.lo = lo; j.hi = n - lo < per_block ? n : lo + per_block; }
        return j;
    }
    j.n_bins = 0;
    if (blockIdx.x >= rows.row_blk[rows.NB]) return j;
    uint32_t a = 0, b = rows.NB;                             // last row whose first workgroup is <= this one and that has workgroups
    while (b - a > 1u) { const uint32_t mid = (a + b) >> 1; if (rows.row_blk[mid] <= blockIdx.x) a = mid; else b = mid; }
    const uint32_t X = a, c = blockIdx.x - rows.row_blk[X];
    j.lo = rows.row_start[X] + c * rows.chunk;
    j.hi = rows.row_start[X + 1];
    if (j.hi - j.lo > rows.chunk) j.hi = j.lo + rows.chunk;
    j.n_bins = X + 1u; j.sub = tri32(X); j.tab = rows.row_tab[X] + c; j.stride = rows.row_blk[X + 1] - rows.row_blk[X];
    return j;
}
__device__ __forceinline__ uint32_t cs_bin(int mode, uint32_t key, uint32_t n_valid, uint32_t sub) {
    if (key >= n_valid) return 0xFFFFFFFFu;
    return mode == CS_BY_ROW ? stream_row(key) : key - sub;
}
// row tables of the second pass from the first pass's offsets O1[row][workgroup] (+ total)
__global__ void cs_rows_kernel(const uint32_t* __restrict__ O1, uint32_t NB, uint32_t chunk, uint32_t* __restrict__ row_start, uint32_t* __restrict__ row_blk,
                               uint32_t* __restrict__ row_tab) {
    for (uint32_t X = threadIdx.x; X <= NB; X += blockDim.x) row_start[X] = O1[(size_t)X * CS_BLOCKS];      // entry [NB][0] is the total
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t blk = 0, tab = 0;
        for (uint32_t X = 0; X < NB; ++X) {
            row_blk[X] = blk; row_tab[X] = tab;
            const uint32_t nch = (row_start[X + 1] - row_start[X] + chunk - 1u) / chunk;
            blk += nch; tab += nch * (X + 1u);
        }
        row_blk[NB] = blk; row_tab[NB] = tab;
    }
}
__global__ __launch_bounds__(256) void cs_hist_kernel(const uint32_t* __restrict__ wkey, uint32_t n, uint32_t n_valid, uint32_t n_keys, int mode, const CsRows rows,
                                                     uint32_t kmask, uint32_t* __restrict__ H) {
    extern __shared__ uint32_t cs_lds[];
    const CsJob job = cs_job(mode, n, n_keys, rows);
    for (uint32_t k = threadIdx.x; k < job.n_bins; k += blockDim.x) cs_lds[k] = 0;
    __syncthreads();
    // eight key words per thread requested together (the loop used to wait for one load per iteration)
    for (uint32_t i0 = job.lo; i0 < job.hi; i0 += 8u * blockDim.x) {
        uint32_t kw[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t i = i0 + j * blockDim.x + threadIdx.x;
            kw[j] = i < job.hi ? wkey[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t bin = kw[j] == 0xFFFFFFFFu ? 0xFFFFFFFFu : cs_bin(mode, kw[j] & kmask, n_valid, job.sub);
            if (bin < job.n_bins) atomicAdd(&cs_lds[bin], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < job.n_bins; k += blockDim.x) H[(size_t)job.tab + (size_t)k * job.stride] = cs_lds[k];
    if (mode != CS_IN_ROW && blockIdx.x == 0 && threadIdx.x == 0) H[(size_t)n_keys * gridDim.x] = 0;          // the scan leaves the total here
}
// The scatter stages tiles of CS_TILE records in LDS sorted by bin, so that the records of one stream leave the tile as one
// contiguous burst.  LDS: the staged records, their destinations, and three per-bin arrays (tile histogram = rank source,
// tile offsets, the workgroup's running global cursor).
constexpr uint32_t CS_TILE = 2048, CS_THREADS = 256;
__host__ __device__ inline size_t cs_scatter_lds(uint32_t n_keys, bool packed) { return (size_t)CS_TILE * (sizeof(WideRec) + (packed ? 4 : 8)) + (size_t)n_keys * 12 + CS_THREADS * 4 + 64; }
__global__ __launch_bounds__(CS_THREADS) void cs_scatter_kernel(const uint32_t* __restrict__ wkey, const WideRec* __restrict__ wrec, uint32_t n, uint32_t n_valid,
                                                         uint32_t n_keys, int mode, const CsRows rows, uint32_t kmask, const uint32_t* __restrict__ O,
                                                         uint32_t* __restrict__ swkey, WideRec* __restrict__ swrec, uint32_t packed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_raw[];
    WideRec* st_rec = (WideRec*)cs_raw;                                   // [CS_TILE]
    uint32_t* st_dst = (uint32_t*)(st_rec + CS_TILE);                     // [CS_TILE] global destination
    uint32_t* st_key = st_dst + CS_TILE;                                  // [CS_TILE]; not there for packed records: their key words (the stream only) stay behind
    uint32_t* hist = st_key + (packed ? 0u : CS_TILE);                    // [n_keys] records of the tile per bin
    uint32_t* toff = hist + n_keys;                                       // [n_keys] first staging position of the bin
    uint32_t* cursor = toff + n_keys;                                     // [n_keys] next global position of the bin for this workgroup
    uint32_t* part = cursor + n_keys;                                     // [CS_THREADS] scan scratch
    const CsJob job = cs_job(mode, n, n_keys, rows);
    const uint32_t nb = job.n_bins, lo = job.lo, hi = job.hi;
    for (uint32_t k = threadIdx.x; k < nb; k += CS_THREADS) cursor[k] = O[(size_t)job.tab + (size_t)k * job.stride];
    constexpr uint32_t PER = CS_TILE / CS_THREADS;
    const uint32_t kper = (nb + CS_THREADS - 1u) / CS_THREADS;                              // bins per thread in the scan
    // keys and records are fetched together, and the next tile's before this one is staged (one round trip per tile, under way early)
    uint32_t nkw[PER];
    WideRec nrec[PER];
    auto fetch = [&](uint32_t t0) {
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint32_t i = t0 + j * CS_THREADS + threadIdx.x;
            nkw[j] = 0xFFFFFFFFu; nrec[j] = WideRec{0ull, 0ull};
            if (i < hi) { nkw[j] = wkey[i]; nrec[j] = wrec[i]; }
        }
    };
    if (lo < hi) fetch(lo);
    for (uint32_t t0 = lo; t0 < hi; t0 += CS_TILE) {
        for (uint32_t k = threadIdx.x; k < nb; k += CS_THREADS) hist[k] = 0;
        __syncthreads();
        uint32_t key[PER], kw[PER], rank[PER];                     // bin, whole key word (stream + weight digit), rank in the tile
        WideRec rc[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            kw[j] = nkw[j]; rc[j] = nrec[j];
            key[j] = cs_bin(mode, kw[j] & kmask, n_valid, job.sub);
            rank[j] = key[j] < nb ? atomicAdd(&hist[key[j]], 1u) : 0u;
        }
        if (t0 + CS_TILE < hi) fetch(t0 + CS_TILE);
        __syncthreads();
        // exclusive scan of the tile histogram: kper consecutive bins per thread, then the partial sums
        uint32_t sum = 0;
        for (uint32_t k = threadIdx.x * kper; k < nb && k < (threadIdx.x + 1u) * kper; ++k) sum += hist[k];
        // the partial sums scanned inside the waves (DPP), the four wave totals through LDS: one barrier instead of sixteen
        const uint32_t sincl = wave_incl_scan(sum, threadIdx.x & 63u);
        if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = sincl;
        __syncthreads();
        uint32_t run = sincl - sum;
        for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) run += part[k];
        const uint32_t tile_n = part[0] + part[1] + part[2] + part[3];
        for (uint32_t k = threadIdx.x * kper; k < nb && k < (threadIdx.x + 1u) * kper; ++k) { toff[k] = run; run += hist[k]; }
        __syncthreads();
        // stage: record -> its bin's run inside the tile, with its global destination
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            if (key[j] < nb) {
                const uint32_t p = toff[key[j]] + rank[j];
                st_rec[p] = rc[j];
                st_dst[p] = cursor[key[j]] + rank[j];
                if (!packed) st_key[p] = kw[j];
            }
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nb; k += CS_THREADS) cursor[k] += hist[k];
        for (uint32_t p = threadIdx.x; p < tile_n; p += CS_THREADS) {
            const uint32_t d = st_dst[p];
            swrec[d] = st_rec[p];
            if (!packed) swkey[d] = st_key[p];
        }
        __syncthreads();
    }
}

// ---- the chunk table grouped by key (stream chunks [0, n_states), row chunks n_states + row; n_keys = never opened) without a general
// sort: histogram, scan, scatter.  Many chunks share few keys (a thousand-sample database has twenty diagonal streams), and
// same-address device atomics run at some ten million per second: a workgroup first groups ITS chunks by key in a small LDS hash
// table (LDS atomics), then does one device atomic per distinct key.  The order of the chunks inside a key is arbitrary.
constexpr uint32_t CT_THREADS = 1024, CT_SLOTS = 2048;
// every thread's key goes into the table; returns the slot and the thread's rank among the block's chunks of that key
__device__ __forceinline__ void ct_group(uint32_t key, bool on, uint32_t* t_key, uint32_t* t_cnt, uint32_t& slot, uint32_t& rank) {
    for (uint32_t e = threadIdx.x; e < CT_SLOTS; e += CT_THREADS) { t_key[e] = 0xFFFFFFFFu; t_cnt[e] = 0u; }
    __syncthreads();
    slot = 0; rank = 0;
    if (on) {
        uint32_t h = (key * 2654435761u) >> 21;                  // 11 bits
        for (;;) {
            const uint32_t prev = atomicCAS(&t_key[h], 0xFFFFFFFFu, key);
            if (prev == 0xFFFFFFFFu || prev == key) break;
            h = (h + 1u) & (CT_SLOTS - 1u);
        }
        slot = h;
        rank = atomicAdd(&t_cnt[h], 1u);
    }
    __syncthreads();
}
__global__ __launch_bounds__(CT_THREADS) void ct_hist_kernel(const uint32_t* __restrict__ chunk_key, uint32_t n, uint32_t n_keys, uint32_t* __restrict__ hist) {
    __shared__ uint32_t t_key[CT_SLOTS], t_cnt[CT_SLOTS];
    const uint32_t i = blockIdx.x * CT_THREADS + threadIdx.x;
    const uint32_t k = i < n ? chunk_key[i] : 0xFFFFFFFFu;
    uint32_t slot, rank;
    ct_group(k, k < n_keys, t_key, t_cnt, slot, rank);
    for (uint32_t e = threadIdx.x; e < CT_SLOTS; e += CT_THREADS)
        if (t_cnt[e]) atomicAdd(&hist[t_key[e]], t_cnt[e]);
}
__global__ __launch_bounds__(CT_THREADS) void ct_scatter_kernel(const uint32_t* __restrict__ chunk_key, uint32_t n, uint32_t n_keys, uint32_t* __restrict__ cursor,
                                                               uint32_t* __restrict__ sorted_key, uint32_t* __restrict__ sorted_id) {
    __shared__ uint32_t t_key[CT_SLOTS], t_cnt[CT_SLOTS];
    const uint32_t i = blockIdx.x * CT_THREADS + threadIdx.x;
    const uint32_t k = i < n ? chunk_key[i] : 0xFFFFFFFFu;
    const bool on = k < n_keys;
    uint32_t slot, rank;
    ct_group(k, on, t_key, t_cnt, slot, rank);
    // one reservation per distinct key of the block; the base replaces the count
    for (uint32_t e = threadIdx.x; e < CT_SLOTS; e += CT_THREADS)
        if (t_cnt[e]) t_cnt[e] = atomicAdd(&cursor[t_key[e]], t_cnt[e]);
    __syncthreads();
    if (on) {
        const uint32_t o = t_cnt[slot] + rank;
        sorted_key[o] = k; sorted_id[o] = i;
    }
}
// stream chunks in use, slots of the sorted chunk table in use
__global__ void ct_count_kernel(const uint32_t* __restrict__ ct_offs, uint32_t n_states, uint32_t* __restrict__ counters) {
    counters[KCTR_CHUNKS] = ct_offs[n_states];
}

// ---- many streams: the row chunks (key n_states + row) grouped by row.  Every chunk a wave opens costs one entry here; a device
// atomic per chunk on the row's counter was measured (the narrow kernel, whose neighbouring waves open chunks of the same rows: 1.7
// -> 4.5 ms), so: counting sort without device atomics — LDS histogram per workgroup of RG_THREADS chunk ids, one scan over
// [row][workgroup], scatter with LDS cursors.  Row X's chunks: row_ids[O[X * G] .. O[(X + 1) * G]), G = workgroups (O[NB * G] = all).
constexpr uint32_t RG_THREADS = 1024;
__global__ __launch_bounds__(RG_THREADS) void rg_hist_kernel(const uint32_t* __restrict__ chunk_key, uint32_t n, uint32_t n_states, uint32_t NB, uint32_t* __restrict__ H) {
    extern __shared__ uint32_t cs_lds[];
    for (uint32_t r = threadIdx.x; r < NB; r += RG_THREADS) cs_lds[r] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * RG_THREADS + threadIdx.x;
    const uint32_t r = (i < n ? chunk_key[i] : 0u) - n_states;
    if (i < n && r < NB) atomicAdd(&cs_lds[r], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < NB; k += RG_THREADS) H[(size_t)k * gridDim.x + blockIdx.x] = cs_lds[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) H[(size_t)NB * gridDim.x] = 0;          // the scan leaves the total here
}
__global__ __launch_bounds__(RG_THREADS) void rg_scatter_kernel(const uint32_t* __restrict__ chunk_key, uint32_t n, uint32_t n_states, uint32_t NB,
                                                               const uint32_t* __restrict__ O, uint32_t* __restrict__ row_ids) {
    extern __shared__ uint32_t cs_lds[];
    for (uint32_t k = threadIdx.x; k < NB; k += RG_THREADS) cs_lds[k] = O[(size_t)k * gridDim.x + blockIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * RG_THREADS + threadIdx.x;
    const uint32_t r = (i < n ? chunk_key[i] : 0u) - n_states;
    if (i < n && r < NB) row_ids[atomicAdd(&cs_lds[r], 1u)] = i;
}

// ---- many streams: the sort inside the block rows.  Row X's records sit in the chunks of its part of row_ids, cut into jobs of
// RS_JOB_CHUNKS chunks, one workgroup each; bins = the row's X + 1 streams.  Counting sort as above (LDS histograms per job, one
// scan over [row][stream][job], LDS-staged scatter), reading through the chunk list, writing the dense sorted arrays
// k2_sorted_kernel walks.
constexpr uint32_t RS_JOB_CHUNKS = 64;
struct RsRows { const uint32_t* O; const uint32_t* row_ids; uint32_t G, NB; const uint32_t* row_job; const uint32_t* row_tab; };
__global__ void rs_rows_kernel(const uint32_t* __restrict__ O, uint32_t G, uint32_t NB, uint32_t* __restrict__ row_job, uint32_t* __restrict__ row_tab,
                               uint32_t* __restrict__ counters) {
    __shared__ uint32_t s_job[1024], s_tab[1024];
    // jobs and table entries of every row, then block-wide exclusive scans (NB may be a few thousand)
    uint32_t cj = 0, ct = 0;
    for (uint32_t base = 0; base < NB; base += 1024) {
        const uint32_t X = base + threadIdx.x;
        uint32_t nj = 0;
        if (X < NB) nj = (O[(size_t)(X + 1u) * G] - O[(size_t)X * G] + RS_JOB_CHUNKS - 1u) / RS_JOB_CHUNKS;
        s_job[threadIdx.x] = nj; s_tab[threadIdx.x] = X < NB ? nj * (X + 1u) : 0u;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const uint32_t a = threadIdx.x >= d ? s_job[threadIdx.x - d] : 0u, b = threadIdx.x >= d ? s_tab[threadIdx.x - d] : 0u;
            __syncthreads();
            s_job[threadIdx.x] += a; s_tab[threadIdx.x] += b;
            __syncthreads();
        }
        if (X < NB) { row_job[X] = cj + s_job[threadIdx.x] - nj; row_tab[X] = ct + s_tab[threadIdx.x] - nj * (X + 1u); }
        cj += s_job[1023]; ct += s_tab[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) { row_job[NB] = cj; row_tab[NB] = ct; counters[KCTR_ROWJOBS] = cj; counters[KCTR_RAW] = O[(size_t)NB * G]; }
}
struct RsJob { uint32_t X, nj, cb, ce, tab; const uint32_t* ids; };

__device__ __forceinline__ bool rs_job(const RsRows& R, uint32_t job, uint32_t job_end, RsJob& j) {
    const uint32_t NB = R.NB;
    if (job >= job_end || job >= R.row_job[NB]) return false;
    uint32_t a = 0, b = NB;                                  // last row whose first job is <= this one (rows without jobs share their successor's first job)
    while (b - a > 1u) { const uint32_t mid = (a + b) >> 1; if (R.row_job[mid] <= job) a = mid; else b = mid; }
    const uint32_t c = job - R.row_job[a];
    j.X = a; j.nj = R.row_job[a + 1] - R.row_job[a];
    j.cb = R.O[(size_t)a * R.G] + c * RS_JOB_CHUNKS;
    const uint32_t re = R.O[(size_t)(a + 1u) * R.G];
    j.ce = re - j.cb > RS_JOB_CHUNKS ? j.cb + RS_JOB_CHUNKS : re;
    j.tab = R.row_tab[a] + c;
    j.ids = R.row_ids;
    return true;
}
template <uint32_t U>                                        // chunks whose key words a thread requests together (one round trip per U chunks)
__global__ __launch_bounds__(256) void rs_hist_kernel(const RsRows R, const uint32_t* __restrict__ chunk_fill, const uint32_t* __restrict__ recw, uint32_t kmask,
                                                     uint32_t* __restrict__ H) {
    extern __shared__ uint32_t cs_lds[];
    __shared__ uint32_t s_id[RS_JOB_CHUNKS], s_fill[RS_JOB_CHUNKS];
    RsJob j;
    if (!rs_job(R, blockIdx.x, 0xFFFFFFFFu, j)) return;
    const uint32_t nb = j.X + 1u, sub = tri32(j.X), nch = j.ce - j.cb;
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) cs_lds[k] = 0;
    // the job's chunk ids and fills first (one round trip), then the key words of four chunks at a time
    if (threadIdx.x < nch) { const uint32_t id = j.ids[j.cb + threadIdx.x]; s_id[threadIdx.x] = id; s_fill[threadIdx.x] = chunk_fill[id]; }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nch; c0 += U) {
        uint32_t kw[U];
#pragma unroll
        for (uint32_t jj = 0; jj < U; ++jj) {
            kw[jj] = 0xFFFFFFFFu;
            if (c0 + jj < nch && threadIdx.x < s_fill[c0 + jj]) kw[jj] = recw[((size_t)s_id[c0 + jj] << CH_SHIFT) + threadIdx.x];
        }
#pragma unroll
        for (uint32_t jj = 0; jj < U; ++jj) {
            const uint32_t bin = (kw[jj] & kmask) - sub;
            if (kw[jj] != 0xFFFFFFFFu && bin < nb) atomicAdd(&cs_lds[bin], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) H[(size_t)j.tab + (size_t)k * j.nj] = cs_lds[k];
}
// Per tile of CS_TILE records (four chunks): keys and records are fetched TOGETHER (the records wait in registers) and the next
// tile's are requested before this one is staged — the kernel used to pay four dependent round trips per tile (chunk id, fill, key,
// record), one workgroup-wide phase after the other.
__global__ __launch_bounds__(CS_THREADS) void rs_scatter_kernel(const RsRows R, const uint32_t* __restrict__ chunk_fill,
                                                         const uint32_t* __restrict__ recw, const WideRec* __restrict__ rec, uint32_t kmask,
                                                         const uint32_t* __restrict__ O, uint32_t cap, uint32_t* __restrict__ swkey, WideRec* __restrict__ swrec,
                                                         uint32_t* __restrict__ counters, uint32_t packed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_raw[];
    __shared__ uint32_t s_id[RS_JOB_CHUNKS], s_fill[RS_JOB_CHUNKS];
    RsJob job;
    // consecutive jobs write neighbouring stretches of every stream of their row (the lines at the seams are shared): workgroups are dealt
    // to the eight XCDs round-robin, so job = f(blockIdx) keeps every XCD on ONE contiguous range of jobs and the seams in one L2
    if (!rs_job(R, xcd_contiguous(blockIdx.x, gridDim.x), 0xFFFFFFFFu, job)) return;
    const uint32_t NB = R.NB;
    const uint32_t nb = job.X + 1u, sub = tri32(job.X), nch = job.ce - job.cb;
    WideRec* st_rec = (WideRec*)cs_raw;                                   // [CS_TILE]
    uint32_t* st_dst = (uint32_t*)(st_rec + CS_TILE);                     // [CS_TILE] global destination
    uint32_t* st_key = st_dst + CS_TILE;                                  // [CS_TILE]; not there for packed records (their key words stay behind: 8 KB
    uint32_t* hist = st_key + (packed ? 0u : CS_TILE);                    //   less LDS per workgroup, one more workgroup per CU)  [nb] records of the tile per bin
    uint32_t* toff = hist + NB;                                           // [nb] first staging position of the bin
    uint32_t* cursor = toff + NB;                                         // [nb] next global position of the bin for this workgroup
    uint32_t* part = cursor + NB;                                         // [CS_THREADS] scan scratch
    for (uint32_t k = threadIdx.x; k < nb; k += CS_THREADS) cursor[k] = O[(size_t)job.tab + (size_t)k * job.nj];
    if (threadIdx.x < nch) { const uint32_t id = job.ids[job.cb + threadIdx.x]; s_id[threadIdx.x] = id; s_fill[threadIdx.x] = chunk_fill[id]; }
    __syncthreads();
    constexpr uint32_t PER = CS_TILE / CH_REC;                            // chunks per tile: thread t takes slot t of each
    static_assert(CH_REC == CS_THREADS, "a chunk per pass of the workgroup");
    const uint32_t kper = (nb + CS_THREADS - 1u) / CS_THREADS;
    uint32_t nkw[PER];
    WideRec nrec[PER];
    auto fetch = [&](uint32_t c0) {
#pragma unroll
        for (uint32_t jj = 0; jj < PER; ++jj) {
            nkw[jj] = 0xFFFFFFFFu; nrec[jj] = WideRec{0ull, 0ull};
            if (c0 + jj < nch && threadIdx.x < s_fill[c0 + jj]) {
                const size_t slot = ((size_t)s_id[c0 + jj] << CH_SHIFT) + threadIdx.x;
                nkw[jj] = recw[slot]; nrec[jj] = rec[slot];
            }
        }
    };
    fetch(0);
    for (uint32_t c0 = 0; c0 < nch; c0 += PER) {
        for (uint32_t k = threadIdx.x; k < nb; k += CS_THREADS) hist[k] = 0;
        __syncthreads();
        uint32_t key[PER], kw[PER], rank[PER];
        WideRec rc[PER];
#pragma unroll
        for (uint32_t jj = 0; jj < PER; ++jj) {
            kw[jj] = nkw[jj]; rc[jj] = nrec[jj];
            key[jj] = kw[jj] != 0xFFFFFFFFu ? (kw[jj] & kmask) - sub : 0xFFFFFFFFu;
            rank[jj] = key[jj] < nb ? atomicAdd(&hist[key[jj]], 1u) : 0u;
        }
        if (c0 + PER < nch) fetch(c0 + PER);                     // the next tile's keys and records are under way while this one is staged
        __syncthreads();
        uint32_t sum = 0;
        for (uint32_t k = threadIdx.x * kper; k < nb && k < (threadIdx.x + 1u) * kper; ++k) sum += hist[k];
        // the partial sums scanned inside the waves (DPP), the four wave totals through LDS: one barrier instead of sixteen
        const uint32_t sincl = wave_incl_scan(sum, threadIdx.x & 63u);
        if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = sincl;
        __syncthreads();
        uint32_t run = sincl - sum;
        for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) run += part[k];
        const uint32_t tile_n = part[0] + part[1] + part[2] + part[3];
        for (uint32_t k = threadIdx.x * kper; k < nb && k < (threadIdx.x + 1u) * kper; ++k) { toff[k] = run; run += hist[k]; }
        __syncthreads();
#pragma unroll
        for (uint32_t jj = 0; jj < PER; ++jj) {
            if (key[jj] < nb) {
                const uint32_t pp = toff[key[jj]] + rank[jj];
                st_rec[pp] = rc[jj];
                st_dst[pp] = cursor[key[jj]] + rank[jj];
                if (!packed) st_key[pp] = kw[jj];
            }
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nb; k += CS_THREADS) cursor[k] += hist[k];
        for (uint32_t pp = threadIdx.x; pp < tile_n; pp += CS_THREADS) {
            const uint32_t d = st_dst[pp];
            if (d < cap) { swrec[d] = st_rec[pp]; if (!packed) swkey[d] = st_key[pp]; }      // (packed records carry their weight themselves)
            else atomicOr(&counters[KCTR_WIDE_OVERFLOW], 1u);            // the sorted arrays are too small: the call is repeated with larger ones
        }
        __syncthreads();
    }
}
__host__ __device__ inline size_t rs_scatter_lds(uint32_t NB, bool packed) { return (size_t)CS_TILE * (sizeof(WideRec) + (packed ? 4 : 8)) + (size_t)NB * 12 + CS_THREADS * 4 + 64; }

// K2 over records sorted by stream (dense mode): a window of K2S_WIN sorted positions, one 64 x 64 tile per run of equal streams
constexpr uint32_t K2S_WIN = 4096;
constexpr int K2S_MIN_WAVES = 3;           // waves per SIMD the compiler must leave room for (4: 128 VGPRs and 104 B of scratch per lane; 3: no scratch)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K2S_MIN_WAVES, 8))) void k2_sorted_kernel(const uint32_t* __restrict__ swkey, const WideRec* __restrict__ swrec,
                                                        uint32_t limit, const uint32_t* __restrict__ total_ptr, const uint32_t* __restrict__ lo_ptr,
                                                        uint32_t n_states, uint32_t kbits, uint32_t dbits, uint32_t* __restrict__ M, uint32_t N,
                                                        uint32_t bwidth, uint32_t rect_nbc, uint32_t rect_cols, unsigned char* __restrict__ touched) {
    __shared__ uint32_t acc[64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[4][64];
    __shared__ unsigned long long lut_ff[256], lut_01[256];
    __shared__ uint16_t bnd[K2S_WIN + 2];          // run starts inside the window, then the end
    __shared__ uint32_t wor_sh;
    __shared__ uint32_t tcount[256];
    const uint32_t total = total_ptr ? (*total_ptr < limit ? *total_ptr : limit) : limit;      // counting sort: the valid records; radix sort: all slots
    const uint32_t p0 = (lo_ptr ? *lo_ptr : 0u) + blockIdx.x * K2S_WIN;                        // (a band of block rows: [*lo_ptr, *total_ptr))
    const uint32_t kmask = (1u << kbits) - 1u;
    if (p0 >= total || (swkey[p0] & kmask) >= n_states) return;
    const uint32_t wend = total - p0 < K2S_WIN ? total - p0 : K2S_WIN;
    {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((threadIdx.x >> i) & 1u) ? 0xFFull << (8 * i) : 0ull;
        lut_ff[threadIdx.x] = v;
        lut_01[threadIdx.x] = v & 0x0101010101010101ull;
    }
    // run starts: every thread scans 32 consecutive positions; a block scan puts them in order
    constexpr uint32_t PER = K2S_WIN / 256;
    const uint32_t t0 = threadIdx.x * PER;
    uint32_t prev = (t0 == 0 || t0 >= wend) ? 0xFFFFFFFFu : (swkey[p0 + t0 - 1] & kmask);
    uint32_t mine = 0, flags = 0;
    uint32_t kk[PER];                                              // the thread's 32 keys: 8 loads of 16 bytes in flight together
    if (t0 + PER <= wend && (p0 & 3u) == 0u) {
#pragma unroll
        for (uint32_t v = 0; v < PER / 4; ++v) {
            const uint4 q4 = ((const uint4*)(swkey + p0 + t0))[v];
            kk[4 * v] = q4.x & kmask; kk[4 * v + 1] = q4.y & kmask; kk[4 * v + 2] = q4.z & kmask; kk[4 * v + 3] = q4.w & kmask;
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) kk[i] = t0 + i < wend ? (swkey[p0 + t0 + i] & kmask) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) {
        if (t0 + i < wend && kk[i] != prev) { flags |= 1u << i; ++mine; }      // the first never-written slot starts a last "run" that ends the loop below
        prev = kk[i];
    }
    const uint32_t mincl = wave_incl_scan(mine, threadIdx.x & 63u);
    if ((threadIdx.x & 63u) == 63u) tcount[threadIdx.x >> 6] = mincl;
    __syncthreads();
    {
        uint32_t o = mincl - mine;
        for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) o += tcount[k];
        for (uint32_t i = 0; i < PER; ++i) if ((flags >> i) & 1u) bnd[o++] = (uint16_t)(t0 + i);
    }
    const uint32_t nb = tcount[0] + tcount[1] + tcount[2] + tcount[3];
    if (threadIdx.x == 0) bnd[nb] = (uint16_t)wend;
    __syncthreads();
    for (uint32_t r = 0; r < nb; ++r) {
        const uint32_t a = bnd[r], b = bnd[r + 1];
        const uint32_t key = swkey[p0 + a] & kmask;
        if (key >= n_states) break;
        K2Item it;
        {
            const uint32_t bucket = key;
            if (rect_nbc) { it.X = bucket / rect_nbc; it.Y = bucket - it.X * rect_nbc; }      // db2db: stream = row block x column blocks + column block
            else { it.X = stream_row(bucket); it.Y = bucket - tri32(it.X); }
            it.rect_cols = rect_cols; it.pshift = 0;
            it.n_rec = b - a; it.count = (it.n_rec + 63u) / 64u; it.ids = nullptr; it.fills = nullptr; it.srec = swrec + p0 + a; it.skey = swkey + p0 + a; it.kbits = kbits; it.dbits = dbits; it.rec = nullptr; it.recw = nullptr;
        }
        k2_run<true>(it, acc, &wor_sh, wbuf, lut_ff, lut_01, M, N, bwidth, touched);
    }
}

// Many streams (row mode): the apply step over the sorted records takes a stream — or a part of K2J_REC records of a long one — per
// workgroup instead of a window of sorted positions: a tile is written back ONCE per K2J_REC records of its stream (north_star: "a single
// HBM write-back per tile"), where the windows of 4096 positions wrote every tile they met — 1.49 GB of atomics for the 0.2 GB matrix of
// 10 000 samples.  Where a stream starts is known from the counting sort's offsets: no run boundaries are searched.
constexpr uint32_t K2J_REC = 16384;
// (O = the sort's offsets [row][stream of the row][job of the row], exclusive sums: entry (X, Y, job 0) is where stream (X, Y) starts)
__global__ void k2j_starts_kernel(const RsRows R, const uint32_t* __restrict__ O, uint32_t n_states, uint32_t* __restrict__ start) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_states) return;
    if (s == n_states) { start[s] = O[R.row_tab[R.NB]]; return; }                         // all sorted records
    const uint32_t X = stream_row(s), Y = s - tri32(X);
    const uint32_t nj = R.row_job[X + 1] - R.row_job[X];                                   // a row without records has no jobs: its streams start where the next row does
    start[s] = O[(size_t)R.row_tab[X] + (size_t)Y * nj];
}
// few streams: O = the one-pass sort's offsets [stream][workgroup share] (+ the total): share 0 of a stream is where the stream starts
__global__ void k2j_starts_flat_kernel(const uint32_t* __restrict__ O, uint32_t n_states, uint32_t shares, uint32_t* __restrict__ start) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= n_states) start[s] = O[(size_t)s * shares];
}
// the jobs in any order: one reservation per workgroup of streams
__global__ __launch_bounds__(1024) void k2j_build_kernel(const uint32_t* __restrict__ start, uint32_t n_states, uint2* __restrict__ jobs, uint32_t cap,
                                                         uint32_t* __restrict__ counters) {
    __shared__ uint32_t wsum[16], base_sh;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t len = s < n_states ? start[s + 1] - start[s] : 0u;
    const uint32_t nj = (len + K2J_REC - 1u) / K2J_REC;
    const uint32_t incl = wave_incl_scan(nj, lane);
    if (lane == 63u) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < 16; ++k) { const uint32_t v = wsum[k]; wsum[k] = run; run += v; }
        base_sh = run ? atomicAdd(&counters[KCTR_K2JOBS], run) : 0u;
    }
    __syncthreads();
    uint32_t o = base_sh + wsum[wave] + incl - nj;
    for (uint32_t part = 0; part < nj; ++part, ++o) if (o < cap) jobs[o] = make_uint2(s, part);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K2S_MIN_WAVES, 8))) void k2_jobs_kernel(const uint32_t* __restrict__ swkey, const WideRec* __restrict__ swrec,
                                                        const uint32_t* __restrict__ start, const uint2* __restrict__ jobs, const uint32_t* __restrict__ n_jobs, uint32_t cap,
                                                        uint32_t limit, uint32_t pshift, uint32_t kbits, uint32_t dbits, uint32_t* __restrict__ M, uint32_t N, uint32_t bwidth, unsigned char* __restrict__ touched) {
    __shared__ uint32_t acc[64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[4][64];
    __shared__ unsigned long long lut_ff[256], lut_01[256];
    __shared__ uint32_t wor_sh;
    const uint32_t nj = *n_jobs < cap ? *n_jobs : cap;
    if (blockIdx.x >= nj) return;
    const uint2 job = jobs[blockIdx.x];
    {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((threadIdx.x >> i) & 1u) ? 0xFFull << (8 * i) : 0ull;
        lut_ff[threadIdx.x] = v;
        lut_01[threadIdx.x] = v & 0x0101010101010101ull;
    }
    // (limit = slots of the sorted arrays: a call whose pools were too small sorted only what fitted — it is repeated with larger pools, and
    // until then nothing beyond the arrays may be read)
    const uint32_t e = start[job.x + 1] < limit ? start[job.x + 1] : limit;
    uint32_t a = start[job.x] + job.y * K2J_REC;
    a = a < e ? a : e;
    const uint32_t b = e - a > K2J_REC ? a + K2J_REC : e;
    if (a == b) return;
    K2Item it;
    it.X = stream_row(job.x); it.Y = job.x - tri32(it.X); it.rect_cols = 0; it.pshift = pshift;
    it.n_rec = b - a; it.count = (it.n_rec + 63u) / 64u; it.ids = nullptr; it.fills = nullptr; it.srec = swrec + a; it.skey = swkey + a; it.kbits = kbits; it.dbits = dbits;
    it.rec = nullptr; it.recw = nullptr;
    k2_run<true>(it, acc, &wor_sh, wbuf, lut_ff, lut_01, M, N, bwidth, touched);
}

// ------------------------------------------------------------------------------------------
// upload-time: sampled estimate of the number of block records per candidate width
// ------------------------------------------------------------------------------------------
constexpr int EST_NW = 10;
struct EstParams {
    const uint2* k0in;
    const uint32_t* bitrel;
    const uint64_t* blkbase;
    const uint64_t* bits;
    const int32_t* parent;
    const uint32_t* nl;
    const uint32_t* w;
    uint32_t P, stride;
    uint32_t widths[EST_NW];
    uint32_t magics[EST_NW];
    unsigned long long* out;       // [EST_NW] records of the sampled nodes with <= 2 blocks, [EST_NW] of the others, [1] nodes sampled,
                                   // [EST_NW] the part of the others that comes from nodes with l2_min blocks or more (the second level writes no records for them)
    uint32_t l2_min;
};
// One thread per sampled node: climbs the root path, decodes every node's local ids (the only decoder run outside the
// call: it looks at one node in `stride`) and counts the blocks of the full list for every candidate width.
__global__ void width_estimate_kernel(const EstParams q) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i0 = (uint64_t)t * q.stride + (t * 2654435761u) % q.stride;
    const uint32_t node = (uint32_t)i0;
    const bool on = i0 < q.P && q.w[node] != 0 && q.nl[node] >= 2u;
    uint32_t nblk[EST_NW], lower_first[EST_NW];
#pragma unroll
    for (int c = 0; c < EST_NW; ++c) { nblk[c] = 0; lower_first[c] = 0xFFFFFFFFu; }
    int32_t y = on ? (int32_t)node : -1;
    while (y >= 0) {
        const uint2 km = q.k0in[y];
        const uint32_t l = kmdb_k0_l(km), last = kmdb_k0_last(km);
        if (l) {
            const uint64_t pos = q.blkbase[(uint32_t)y >> 8] + q.bitrel[y];
            uint32_t sum = 0;
            if (l > 1) { BitCursor c(q.bits, pos); for (uint32_t k = 0; k + 1 < l; ++k) sum += c.next(); }
            uint32_t id = last - sum;
            uint32_t fb[EST_NW], cb[EST_NW];
#pragma unroll
            for (int c = 0; c < EST_NW; ++c) { fb[c] = __umulhi(id, q.magics[c]); cb[c] = fb[c]; nblk[c] += 1; }
            if (l > 1) {
                BitCursor c2(q.bits, pos);
                for (uint32_t k = 0; k + 1 < l; ++k) {
                    id += c2.next();
#pragma unroll
                    for (int c = 0; c < EST_NW; ++c) { const uint32_t b = __umulhi(id, q.magics[c]); nblk[c] += b != cb[c]; cb[c] = b; }
                }
            }
#pragma unroll
            for (int c = 0; c < EST_NW; ++c) { nblk[c] -= cb[c] == lower_first[c]; lower_first[c] = fb[c]; }
        }
        y = q.parent[y];
    }
    // one atomic per wave and counter
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int c = 0; c < EST_NW; ++c) {
        // a node with at most two blocks sends one record to its first block's stream chunk; all other records take the wide pool
        const unsigned long long recs = (unsigned long long)nblk[c] * (nblk[c] + 1u) / 2u;
        unsigned long long rn = on && nblk[c] >= 1u && nblk[c] <= 2u ? 1ull : 0ull;
        unsigned long long rg = on ? recs - rn : 0ull;
        unsigned long long r2 = on && nblk[c] >= q.l2_min ? recs : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { rn += shfl64(rn, (int)(lane ^ (uint32_t)d)); rg += shfl64(rg, (int)(lane ^ (uint32_t)d)); r2 += shfl64(r2, (int)(lane ^ (uint32_t)d)); }
        if (lane == 0) { if (rn) atomicAdd(&q.out[c], rn); if (rg) atomicAdd(&q.out[EST_NW + c], rg); if (r2) atomicAdd(&q.out[2 * EST_NW + 1 + c], r2); }
    }
    const unsigned long long nb = __ballot(on);
    if (lane == 0 && nb) atomicAdd(&q.out[2 * EST_NW], (unsigned long long)__popcll(nb));
}

// upload-time: sampled estimate of the extra (block, mask) pairs K0 reserves (one node in `stride`, its own list only)
__global__ void pair_estimate_kernel(const uint2* __restrict__ k0in, const uint32_t* __restrict__ bitrel, const uint64_t* __restrict__ blkbase,
                                     const uint64_t* __restrict__ bits, uint32_t P, uint32_t stride, BlockMap bm, unsigned long long* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i0 = (uint64_t)t * stride + (t * 2654435761u) % stride;
    unsigned long long need = 0;
    if (i0 < P) {
        const uint2 km = k0in[i0];
        const uint32_t l = kmdb_k0_l(km), last = kmdb_k0_last(km);
        if (l > 1) {
            BitCursor c(bits, blkbase[(uint32_t)i0 >> 8] + bitrel[i0]);
            uint32_t span = 0;
            for (uint32_t k = 0; k + 1 < l; ++k) span += c.next();
            const uint32_t nb = bm.blk(last) - bm.blk(last - span);
            need = nb < l - 1u ? nb : l - 1u;
        }
    }
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) need += shfl64(need, (int)(lane ^ (uint32_t)d));
    if (lane == 0 && need) atomicAdd(out, need);
}

// v1 / new2all node arrays from the compact layout
__global__ void v1_arrays_kernel(const uint2* __restrict__ k0in, const uint32_t* __restrict__ bitrel, const uint64_t* __restrict__ blkbase,
                                 const uint32_t* __restrict__ nl, uint32_t P, uint4* __restrict__ meta, uint64_t* __restrict__ bitpos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint2 km = k0in[i];
    const uint32_t v = nl[i];
    meta[i] = make_uint4(v, kmdb_k0_l(km), kmdb_k0_last(km), kmdb_k0_bits(km));
    bitpos[i] = blkbase[i >> 8] + bitrel[i];
}

// weight digit bits of the wide pool's key word: what the stream bits and the two digit-index bits leave (four digits cover 32 bits)
inline uint32_t wide_digit_bits(int key_bits) { return (uint32_t)(32 - key_bits - 2); }
PoolView pool_view(const kmdb_db* db, bool dense) {
    return PoolView{db->counters, db->chunk_key, db->chunk_fill, db->rec, db->recw, db->sub_cursor, (uint32_t)(db->pool_cap / KMDB_SUBPOOLS),
                    (uint32_t)db->pool_cap, db->wkey, (WideRec*)db->wrec, db->wsub_cursor, (uint32_t)(db->wide_pool_cap / KMDB_SUBPOOLS), dense ? 1u : 0u,
                    (uint32_t)db->key_bits, wide_digit_bits(db->key_bits), db->row_mode ? 1u : 0u, db->n_states, db->rec_pshift};
}

void free_and_null(void** p) { if (*p) { (void)hipFree(*p); *p = nullptr; } }
#define FREE_NULL(x) free_and_null((void**)&(x))

constexpr uint32_t K1W_MAX_WAVES = 8192;
constexpr uint32_t K1W_RUN_NODES = 256;     // wide nodes per run of a wave (round-robin: heavy nodes sit together in the DFS order)
struct U32toU64 { __host__ __device__ unsigned long long operator()(uint32_t v) const { return v; } };
struct ValidKey { uint32_t n_states, kmask; __host__ __device__ uint32_t operator()(uint32_t k) const { return (k & kmask) < n_states ? 1u : 0u; } };

// Record slots a pool can index.  Slot numbers are 32-bit; the few-streams path hands slot counts to rocPRIM as int (2^31).  The many-streams
// path (row mode) is the engine's own kernels throughout: 2^32 minus a margin (the slot after the last chunk must not wrap to the
// "no open chunk" value 0 of the row tables, and the windows of the apply kernel round the record count up).
inline uint64_t pool_slot_limit(const kmdb_db* db) { return db->row_mode ? (1ull << 32) - (1ull << 20) : (1ull << 31); }

int alloc_record_pool(kmdb_db* db, uint64_t chunks) {
    FREE_NULL(db->chunk_key); FREE_NULL(db->chunk_fill); FREE_NULL(db->sorted_key); FREE_NULL(db->sorted_id);
    FREE_NULL(db->rec); FREE_NULL(db->recw); FREE_NULL(db->sort_tmp); FREE_NULL(db->rs_hist); FREE_NULL(db->rs_offs); FREE_NULL(db->rs_tmp);
    db->pool_cap = 0;
    chunks = (chunks + (uint64_t)KMDB_SUBPOOLS * ARENA_GRAB - 1) / ((uint64_t)KMDB_SUBPOOLS * ARENA_GRAB) * ((uint64_t)KMDB_SUBPOOLS * ARENA_GRAB);
    if (chunks >= pool_slot_limit(db) >> CH_SHIFT) return kmdb_set_error("kmdb: record pool would exceed its " + std::to_string(pool_slot_limit(db)) + " record slots");
    HIP_TRY(hipMalloc((void**)&db->chunk_key, chunks * 4));
    HIP_TRY(hipMalloc((void**)&db->chunk_fill, chunks * 4));
    HIP_TRY(hipMalloc((void**)&db->sorted_key, chunks * 4));
    HIP_TRY(hipMalloc((void**)&db->sorted_id, chunks * 4));
    HIP_TRY(hipMalloc((void**)&db->rec, (chunks << CH_SHIFT) * 16));
    HIP_TRY(hipMalloc((void**)&db->recw, (chunks << CH_SHIFT) * 4));
    {
        rocprim::transform_iterator<uint32_t*, U32toU64, unsigned long long> it(db->chunk_fill, U32toU64());
        HIP_TRY(prim::sum(nullptr, db->sort_tmp_bytes, it, (unsigned long long*)nullptr, (int)chunks, db->stream));
    }
    HIP_TRY(hipMalloc(&db->sort_tmp, std::max<size_t>(db->sort_tmp_bytes, 16)));
    if (db->row_mode) {
        // the row chunks grouped by row: [row][workgroup of RG_THREADS chunk ids] counts / offsets, the grouped ids
        FREE_NULL(db->rg_hist); FREE_NULL(db->rg_offs); FREE_NULL(db->rg_tmp); FREE_NULL(db->row_ids);
        db->rg_blocks = (uint32_t)((chunks + RG_THREADS - 1) / RG_THREADS);
        const size_t rg_ne = (size_t)db->NB * db->rg_blocks + 1;
        HIP_TRY(hipMalloc((void**)&db->rg_hist, rg_ne * 4));
        HIP_TRY(hipMalloc((void**)&db->rg_offs, rg_ne * 4));
        HIP_TRY(hipMalloc((void**)&db->row_ids, chunks * 4));
        HIP_TRY(prim::exclusive_sum(nullptr, db->rg_tmp_bytes, db->rg_hist, db->rg_offs, (int)rg_ne, db->stream));
        HIP_TRY(hipMalloc(&db->rg_tmp, std::max<size_t>(db->rg_tmp_bytes, 16)));
        // the sort inside the rows: at most chunks / RS_JOB_CHUNKS + NB jobs, every one with at most NB bins
        db->rs_entries = (chunks / RS_JOB_CHUNKS + db->NB + 1) * (uint64_t)db->NB + 1;
        if (db->rs_entries >= (1ull << 31)) return kmdb_set_error("kmdb: the table of the sort inside the block rows would exceed 2^31 entries");
        HIP_TRY(hipMalloc((void**)&db->rs_hist, db->rs_entries * 4));
        HIP_TRY(hipMalloc((void**)&db->rs_offs, db->rs_entries * 4));
        HIP_TRY(prim::exclusive_sum(nullptr, db->rs_tmp_bytes, db->rs_hist, db->rs_offs, (int)db->rs_entries, db->stream));
        HIP_TRY(hipMalloc(&db->rs_tmp, std::max<size_t>(db->rs_tmp_bytes, 16)));
    }
    db->pool_cap = chunks;
    return 0;
}
// few streams: the wide pool (records in arrival order + their stream keys) and the sorted copies; many streams (row mode): the
// sorted arrays only — the wide records travel through row chunks of the chunk pool
int alloc_wide_pool(kmdb_db* db, uint64_t chunks) {
    FREE_NULL(db->wkey); FREE_NULL(db->wrec); FREE_NULL(db->swkey); FREE_NULL(db->swrec); FREE_NULL(db->sort2_tmp);
    db->wide_pool_cap = 0; db->sorted_cap = 0;
    chunks = (chunks + (uint64_t)KMDB_SUBPOOLS * WIDE_GRAB - 1) / ((uint64_t)KMDB_SUBPOOLS * WIDE_GRAB) * ((uint64_t)KMDB_SUBPOOLS * WIDE_GRAB);
    const uint64_t slots = chunks << WCH_SHIFT;
    if (slots >= pool_slot_limit(db)) return kmdb_set_error("kmdb: wide record pool would exceed its " + std::to_string(pool_slot_limit(db)) + " record slots");
    HIP_TRY(hipMalloc((void**)&db->swkey, slots * 4));
    HIP_TRY(hipMalloc(&db->swrec, slots * sizeof(WideRec)));
    db->sorted_cap = slots;
    // jobs of the apply kernel: every stream at least one, a long one a job per K2J_REC records
    FREE_NULL(db->k2j_jobs);
    db->k2j_cap = slots / K2J_REC + db->n_states + 1;
    HIP_TRY(hipMalloc((void**)&db->k2j_jobs, db->k2j_cap * sizeof(uint2)));
    if (db->row_mode) {
        db->wide_pool_cap = chunks;
        return 0;
    }
    HIP_TRY(hipMalloc((void**)&db->wkey, slots * 4));
    HIP_TRY(hipMalloc(&db->wrec, slots * sizeof(WideRec)));
    if (!db->cs_hist) {
        // one pass: [stream][workgroup]
        const size_t ne = (size_t)db->n_states * CS_BLOCKS_ONE + 1;
        HIP_TRY(hipMalloc((void**)&db->cs_hist, ne * 4));
        HIP_TRY(hipMalloc((void**)&db->cs_offs, ne * 4));
        HIP_TRY(prim::exclusive_sum(nullptr, db->cs_tmp_bytes, db->cs_hist, db->cs_offs, (int)ne, db->stream));
        HIP_TRY(hipMalloc(&db->cs_tmp, std::max<size_t>(db->cs_tmp_bytes, 16)));
    }
    db->wide_pool_cap = chunks;
    return 0;
}
int alloc_pair_pool(kmdb_db* db, uint64_t entries) {
    FREE_NULL(db->pair_blk); FREE_NULL(db->pair_mask);
    entries = (entries + KMDB_PAIR_REGIONS - 1) / KMDB_PAIR_REGIONS * KMDB_PAIR_REGIONS;
    if (entries >= (1ull << 32)) return kmdb_set_error("kmdb: pair pool would exceed 2^32 entries");
    HIP_TRY(hipMalloc((void**)&db->pair_blk, entries * 2 + 64));
    HIP_TRY(hipMalloc((void**)&db->pair_mask, entries * 8 + 64));
    db->pair_cap = entries;
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// list index for new2all / db2db (engine_state.h): checkpoints of the long local lists
__global__ void ck_count_kernel(const uint2* __restrict__ k0in, uint32_t P, uint32_t* __restrict__ cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > P) return;
    const uint32_t l = i < P ? kmdb_k0_l(k0in[i]) : 0u;
    cnt[i] = l > KMDB_CK_IDS ? (l + KMDB_CK_IDS - 1u) / KMDB_CK_IDS : 0u;
}
__global__ void ck_fill_kernel(const uint2* __restrict__ k0in, const uint32_t* __restrict__ bitrel, const uint64_t* __restrict__ blkbase,
                               const uint64_t* __restrict__ bits, uint32_t P, const uint32_t* __restrict__ ck_ofs, uint64_t* __restrict__ ck_bit,
                               uint32_t* __restrict__ ck_id) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint2 km = k0in[i];
    const uint32_t l = kmdb_k0_l(km), last = kmdb_k0_last(km);
    if (l <= KMDB_CK_IDS) return;
    const uint64_t pos = blkbase[i >> 8] + bitrel[i];
    uint32_t sum = 0;
    { BitCursor c(bits, pos); for (uint32_t t = 0; t + 1 < l; ++t) sum += c.next(); }
    BitCursor c(bits, pos);
    uint32_t id = last - sum, o = ck_ofs[i];
    for (uint32_t t = 0; t < l; ++t) {
        if (t % KMDB_CK_IDS == 0) { ck_id[o] = id; ck_bit[o] = c.wi * 64u + c.s; ++o; }      // element t and the position of the code after it
        if (t + 1 < l) id += c.next();
    }
}

static int ensure_v1_impl(kmdb_db* db);
int kmdb_ensure_v1_arrays(kmdb_db* db) {
    if (db->v1_ready) return 0;
    if (ensure_v1_impl(db)) {
        // nothing half made is left behind: the next call starts over
        for (void** p : {(void**)&db->meta, (void**)&db->bitpos, (void**)&db->wprefix, (void**)&db->v1_counters, (void**)&db->segs, (void**)&db->v1_scan_tmp,
                         (void**)&db->ck_ofs, (void**)&db->ck_bit, (void**)&db->ck_id}) free_and_null(p);
        return 1;
    }
    db->v1_ready = true;
    return 0;
}
static int ensure_v1_impl(kmdb_db* db) {
    const uint64_t P = db->P;
    HIP_TRY(hipMalloc((void**)&db->meta, std::max<uint64_t>(P, 1) * sizeof(uint4)));
    HIP_TRY(hipMalloc((void**)&db->bitpos, std::max<uint64_t>(P, 1) * 8));
    HIP_TRY(hipMalloc((void**)&db->wprefix, (P + 1) * 4));
    HIP_TRY(hipMalloc((void**)&db->v1_counters, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(db->v1_counters, 0, 8 * sizeof(unsigned long long)));
    if (P) hipLaunchKernelGGL(v1_arrays_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, db->stream, db->k0in, db->bitrel, db->blkbase, db->nl,
                              (uint32_t)P, db->meta, db->bitpos);
    HIP_TRY(hipGetLastError());
    std::vector<Segment> segs;
    const uint64_t step = 2048;
    for (uint64_t f = 0; f < P; f += step) segs.push_back(Segment{(uint32_t)f, (uint32_t)std::min<uint64_t>(P, f + step)});
    while (segs.size() % WAVES_PER_BLOCK) segs.push_back(Segment{(uint32_t)P, (uint32_t)P});
    HIP_TRY(hipMalloc((void**)&db->segs, std::max<size_t>(segs.size(), 1) * sizeof(Segment)));
    if (!segs.empty()) HIP_TRY(hipMemcpy(db->segs, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice));
    db->n_segs = (uint32_t)segs.size();
    HIP_TRY(prim::exclusive_sum(nullptr, db->v1_scan_tmp_bytes, db->w, db->wprefix, (int)(P + 1)));
    HIP_TRY(hipMalloc(&db->v1_scan_tmp, std::max<size_t>(db->v1_scan_tmp_bytes, 16)));
    {
        const uint64_t P1 = P + 1;
        uint32_t* cnt = nullptr;
        void* tmp = nullptr;
        size_t tb = 0;
        HIP_TRY(hipMalloc((void**)&cnt, P1 * 4));
        HIP_TRY(hipMalloc((void**)&db->ck_ofs, P1 * 4));
        hipLaunchKernelGGL(ck_count_kernel, dim3((unsigned)((P1 + 255) / 256)), dim3(256), 0, db->stream, db->k0in, (uint32_t)P, cnt);
        HIP_TRY(prim::exclusive_sum(nullptr, tb, cnt, db->ck_ofs, (int)P1, db->stream));
        HIP_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
        HIP_TRY(prim::exclusive_sum(tmp, tb, cnt, db->ck_ofs, (int)P1, db->stream));
        uint32_t n_ck = 0;
        HIP_TRY(hipMemcpyAsync(&n_ck, db->ck_ofs + P, 4, hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        (void)hipFree(cnt); (void)hipFree(tmp);
        HIP_TRY(hipMalloc((void**)&db->ck_bit, std::max<uint64_t>(n_ck, 1) * 8));
        HIP_TRY(hipMalloc((void**)&db->ck_id, std::max<uint64_t>(n_ck, 1) * 4));
        if (P) hipLaunchKernelGGL(ck_fill_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, db->stream, db->k0in, db->bitrel, db->blkbase, db->bits,
                                  (uint32_t)P, db->ck_ofs, db->ck_bit, db->ck_id);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(db->stream));
    return 0;
}

// db2db (csrc/db2db.hip): records (row block, column block, row mask, column mask, weight digit in the key word) of the pairs of
// patterns that share k-mers, in arrival order in [0, nslots) with never-written slots keyed all-ones -> sorted by stream and
// accumulated into the dense n_rows x n_cols matrix M by the same sort and matrix-core kernels as the all2all's wide pool.
int kmdb_rect_sort_apply(hipStream_t st, uint32_t* wkey, void* wrec, uint32_t nslots, uint32_t nbr, uint32_t nbc, int key_bits, uint32_t* M, uint32_t n_rows,
                         uint32_t n_cols) {
    if (!nslots) return 0;
    const uint32_t n_states = nbr * nbc, kmask = (1u << key_bits) - 1u;
    uint32_t *swkey = nullptr, *hist = nullptr, *offs = nullptr;
    WideRec* swrec = nullptr;
    void* tmp = nullptr;
    auto cleanup = [&]() { for (void* p : {(void*)swkey, (void*)swrec, (void*)hist, (void*)offs, tmp}) if (p) (void)hipFree(p); };
#define RS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    RS_TRY(hipMalloc((void**)&swkey, (size_t)nslots * 4));
    RS_TRY(hipMalloc((void**)&swrec, (size_t)nslots * sizeof(WideRec)));
    const uint32_t* total_ptr = nullptr;
    if (n_states <= CS_MAX_KEYS) {
        const size_t ne = (size_t)n_states * CS_BLOCKS_ONE + 1;
        size_t tb = 0;
        RS_TRY(hipMalloc((void**)&hist, ne * 4));
        RS_TRY(hipMalloc((void**)&offs, ne * 4));
        RS_TRY(prim::exclusive_sum(nullptr, tb, hist, offs, (int)ne, st));
        RS_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
        const CsRows no_rows{};
        hipLaunchKernelGGL(cs_hist_kernel, dim3(CS_BLOCKS_ONE), dim3(256), n_states * 4, st, wkey, nslots, n_states, n_states, (int)CS_BY_STREAM, no_rows, kmask, hist);
        RS_TRY(prim::exclusive_sum(tmp, tb, hist, offs, (int)ne, st));
        RS_TRY(hipFuncSetAttribute((const void*)cs_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cs_scatter_lds(n_states, false)));
        hipLaunchKernelGGL(cs_scatter_kernel, dim3(CS_BLOCKS_ONE), dim3(CS_THREADS), cs_scatter_lds(n_states, false), st, wkey, (const WideRec*)wrec, nslots, n_states, n_states,
                           (int)CS_BY_STREAM, no_rows, kmask, offs, swkey, swrec, 0u);
        total_ptr = offs + (ne - 1);
    } else {
        size_t tb = 0;
        RS_TRY(prim::sort_pairs(nullptr, tb, wkey, swkey, (WideRec*)wrec, swrec, (int)nslots, 0, key_bits, st));
        RS_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
        RS_TRY(prim::sort_pairs(tmp, tb, wkey, swkey, (WideRec*)wrec, swrec, (int)nslots, 0, key_bits, st));
    }
    hipLaunchKernelGGL(k2_sorted_kernel, dim3((nslots + K2S_WIN - 1) / K2S_WIN), dim3(256), 0, st, swkey, (const WideRec*)swrec, nslots, total_ptr, (const uint32_t*)nullptr,
                       n_states, (uint32_t)key_bits, wide_digit_bits(key_bits), M, n_rows, 64u, nbc, n_cols, (unsigned char*)nullptr);
    RS_TRY(hipGetLastError());
    RS_TRY(hipStreamSynchronize(st));
#undef RS_TRY
    cleanup();
    return 0;
}

static int blocks_prepare_impl(kmdb_db* db);
// (the flag is set only when every allocation succeeded: a failed lazy preparation inside a call leaves nothing half made behind)
int kmdb_blocks_prepare(kmdb_db* db) {
    if (blocks_prepare_impl(db)) { kmdb_blocks_release(db); return 1; }
    db->blocks_prepared = true;
    return 0;
}
static int blocks_prepare_impl(kmdb_db* db) {
    const uint64_t N = db->N, P = db->P;
    db->fallback_reason.clear();
    if (N < 2 || P == 0) { db->fallback_reason = "fewer than two samples"; return 0; }
    if (!db->chain_ok) {
        db->fallback_reason = "a root path of " + std::to_string(db->max_depth) + " nodes exceeds the chain table (" + std::to_string(KMDB_CHAIN_MAX) + ")";
        return 0;
    }
    const bool verbose = getenv("KMDB_VERBOSE") != nullptr;
    auto t_mark = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {                        // where the preparation's time goes (hipMalloc of tens of GB is not free)
        if (!verbose) return;
        (void)hipStreamSynchronize(db->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kmdb] prepare: %-36s %.3f s\n", what, std::chrono::duration<double>(now - t_mark).count());
        t_mark = now;
    };
    // ---- block width: fewer sample ids per block than 64 pay off when the samples cluster (clades, species) in id
    // ranges that a 64-id grid would cut in two.  One node in `stride` is decoded along its whole root path and its
    // blocks are counted for every candidate at once; the candidate with the fewest records wins.
    const uint32_t cands[EST_NW] = {64, 60, 56, 52, 50, 48, 44, 40, 36, 32};
    uint32_t forced = 0;
    if (const char* e = getenv("KMDB_BLOCK_WIDTH")) forced = (uint32_t)strtoul(e, nullptr, 10);
    const uint32_t stride = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(1024, P / 65536));
    uint64_t est_n = 0, est_g = 0, est_l2 = 0;                  // records of the nodes with <= 2 blocks, of the others, and of those the part the second level replaces
    {
        EstParams q{};
        q.k0in = db->k0in; q.bitrel = db->bitrel; q.blkbase = db->blkbase; q.bits = db->bits; q.parent = db->parent; q.nl = db->nl; q.w = db->w;
        q.P = (uint32_t)P; q.stride = stride;
        for (int c = 0; c < EST_NW; ++c) { q.widths[c] = cands[c]; q.magics[c] = (uint32_t)((1ull << 32) / cands[c]) + 1u; }
        q.l2_min = getenv("KMDB_L2_MIN") ? std::max<uint32_t>(K1W_HEAVY, (uint32_t)atoi(getenv("KMDB_L2_MIN"))) : L2_MIN_BLOCKS;
        HIP_TRY(hipMalloc((void**)&q.out, (3 * EST_NW + 1) * 8));
        HIP_TRY(hipMemsetAsync(q.out, 0, (3 * EST_NW + 1) * 8, db->stream));
        const uint64_t nthreads = (P + stride - 1) / stride;
        hipLaunchKernelGGL(width_estimate_kernel, dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, db->stream, q);
        HIP_TRY(hipGetLastError());
        unsigned long long h[3 * EST_NW + 1];
        HIP_TRY(hipMemcpyAsync(h, q.out, sizeof h, hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        (void)hipFree(q.out);
        int best = 0;
        for (int c = 1; c < EST_NW; ++c) if (h[c] + h[EST_NW + c] < h[best] + h[EST_NW + best]) best = c;
        if (forced >= 32 && forced <= 64) {
            db->width = forced;
            best = 0;
            for (int c = 0; c < EST_NW; ++c) if (cands[c] >= forced) best = c;
        } else db->width = cands[best];
        est_n = h[best] * stride; est_g = h[EST_NW + best] * stride; est_l2 = h[2 * EST_NW + 1 + best] * stride;
        db->est_records = est_n + est_g;
        if (verbose) {
            fprintf(stderr, "[kmdb] width estimate (1 node in %u, %llu sampled):", stride, h[2 * EST_NW]);
            for (int c = 0; c < EST_NW; ++c) fprintf(stderr, " %u:%llu", cands[c], (h[c] + h[EST_NW + c]) * stride);
            fprintf(stderr, " -> width %u\n", db->width);
        }
    }
    db->NB = (uint32_t)((N + db->width - 1) / db->width);
    const uint64_t n_states = (uint64_t)db->NB * (db->NB + 1) / 2;          // streams = block pairs
    if (n_states + 1 >= (1ull << 22)) { db->fallback_reason = "too many block pairs"; return 0; }          // 22 stream bits + 8-bit weight digits in a key word
    static_assert((1ull << 22) < (uint64_t)BNONE * (BNONE + 1ull) / 2ull, "fewer than 2^22 block pairs keep the block index (< 2897) below BNONE and inside pair_blk's 16 bits");
    db->n_states = (uint32_t)n_states;
    if (k1w_wave_bytes(std::max<uint32_t>(K1W_ARENA_MIN, (db->NB + 2u + 63u) & ~63u), (db->NB + 2u + 3u) & ~3u, db->chain_cap, db->NB) > (size_t)(152u << 10)) {
        db->fallback_reason = "the lists of the wide-node kernel do not fit the LDS (" + std::to_string(db->NB) + " blocks, root paths of up to " +
                              std::to_string(db->max_depth) + " nodes)";
        return 0;
    }
    phase("block width estimate");
    // ---- working set
    HIP_TRY(hipMalloc((void**)&db->p0_mask, P * 8));
    HIP_TRY(hipMalloc((void**)&db->p0_info, P * 4));
    HIP_TRY(hipMalloc((void**)&db->pair_ofs, P * 4));
    HIP_TRY(hipMalloc((void**)&db->pair_cursor, (KMDB_PAIR_REGIONS + 1) * 16 * 4));
    HIP_TRY(hipMalloc((void**)&db->fn_mask, P * 16));
    HIP_TRY(hipMalloc((void**)&db->fn_blk, P * 4));
    const uint64_t n_words = (P + 63) / 64;
    HIP_TRY(hipMalloc((void**)&db->widebits, n_words * 8));
    HIP_TRY(hipMalloc((void**)&db->wide_cnt, (n_words + 1) * 4));
    HIP_TRY(hipMalloc((void**)&db->wide_base, (n_words + 1) * 4));
    HIP_TRY(prim::exclusive_sum(nullptr, db->scan_tmp_bytes, db->wide_cnt, db->wide_base, (int)(n_words + 1)));
    HIP_TRY(hipMalloc(&db->scan_tmp, std::max<size_t>(db->scan_tmp_bytes, 16)));
    HIP_TRY(hipMalloc((void**)&db->counters, KCTR_COUNT * 4));
    HIP_TRY(hipHostMalloc((void**)&db->h_counters, KCTR_COUNT * 4));
    phase("per-node arrays (hipMalloc)");
    {
        // extra pairs: sampled with the chosen width (K0 reserves min(l - 1, blocks spanned) per list that is not a single short run)
        unsigned long long* d_need = nullptr;
        HIP_TRY(hipMalloc((void**)&d_need, 8));
        HIP_TRY(hipMemsetAsync(d_need, 0, 8, db->stream));
        const BlockMap bm{db->width, (uint32_t)((1ull << 32) / db->width) + 1u};
        const uint64_t nthreads = (P + stride - 1) / stride;
        hipLaunchKernelGGL(pair_estimate_kernel, dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, db->stream, db->k0in, db->bitrel, db->blkbase, db->bits,
                           (uint32_t)P, stride, bm, d_need);
        unsigned long long h_need = 0;
        HIP_TRY(hipMemcpyAsync(&h_need, d_need, 8, hipMemcpyDeviceToHost, db->stream));
        HIP_TRY(hipStreamSynchronize(db->stream));
        (void)hipFree(d_need);
        const uint64_t est_pairs = h_need * stride;
        if (verbose) fprintf(stderr, "[kmdb] extra (block, mask) pairs, sampled: %llu\n", (unsigned long long)est_pairs);
        // (twice the estimate: three quarters of the pool are wave-private sub-pools, and a sub-pool that runs out sends its waves to the shared
        // rest behind ONE cursor — with 1.25 x the estimate the sub-pools were nearly full and the decode launches took 3.9 instead of 2.2 ms at C2,
        // profiles/r06_j6: same-address device atomics)
        if (alloc_pair_pool(db, std::max<uint64_t>(est_pairs * 2 + P / 4, (uint64_t)KMDB_PAIR_REGIONS * 64))) return 1;
    }
    phase("pair estimate + pair pool");
    HIP_TRY(hipMalloc((void**)&db->sub_cursor, KMDB_SUBPOOLS * 16 * 4));
    HIP_TRY(hipMalloc((void**)&db->wsub_cursor, KMDB_SUBPOOLS * 16 * 4));
    HIP_TRY(hipMalloc((void**)&db->run_ctr, K1W_CTRS * 16 * 4));
    HIP_TRY(hipMalloc((void**)&db->direct_ctr, KMDB_SUBPOOLS * 16 * 4));
    // Where the records go.  The narrow kernel's first-block diagonal records follow the clustering of the DFS stream: per-stream
    // chunks (an open chunk per block and wave), applied straight from the grouped chunk table.  The other records spread over
    // many streams, a few per stream and wave:
    //   few streams (<= CS_MAX_KEYS):  arrival order into the wide pool, ONE counting-sort pass by stream (measured: 5 x faster than
    //                                  per-stream reservations);
    //   many streams ("row mode"):     per-BLOCK-ROW chunks (an open chunk per row and wave, slots by LDS atomics), so the records
    //                                  arrive grouped by row and only the counting sort inside the rows remains (a scatter straight
    //                                  to 20 100 streams is bound by the number of write requests: measured 25.8 ms for 495 M records;
    //                                  grouping by row in a pass of its own cost another 7.5 ms).
    // With more streams than the narrow kernel's open-chunk table holds it sends its diagonal records the same way, if its chunks turn
    // out to be evicted nearly empty.
    db->dense_wide = true;
    db->dense_narrow = false;
    if (const char* e = getenv("KMDB_DENSE")) db->dense_narrow = atoi(e) >= 2;
    // Round 6: the narrow kernel's first-block records no longer travel through stream chunks (engine_state.h: k1n_mode; KMDB_K1N_MODE=0 / 1 / 2
    // for the A/B): compacted per slice and applied by k2d_kernel beside the wide kernel (2, default), or applied inside the narrow kernel (1).
    db->k1n_mode = 2;
    if (const char* e = getenv("KMDB_K1N_MODE")) if (*e) db->k1n_mode = std::max(0, std::min(2, atoi(e)));
    if (db->k1n_mode) db->dense_narrow = true;
    if (db->k1n_mode == 2) {
        const size_t slots = (size_t)db->n_nsegs * db->nseg_nodes;
        HIP_TRY(hipMalloc((void**)&db->dmask, std::max<size_t>(slots, 1) * 8));
        HIP_TRY(hipMalloc((void**)&db->dwx, std::max<size_t>(slots, 1) * 4));
        HIP_TRY(hipMalloc((void**)&db->slice_cnt, std::max<size_t>(db->n_nsegs, 1) * 4));
    }
    db->row_mode = db->n_states > CS_MAX_KEYS;
    if (const char* e = getenv("KMDB_ROW_MODE")) if (*e) db->row_mode = atoi(e) != 0;          // (tests: small databases through the many-streams path)
    // packed records: a block width of at most 54 leaves the column mask 10 spare bits and more — an 8-bit weight digit at least and the digit's
    // index — so the records of the many-streams path travel as 16 bytes through the sort and the apply step (KMDB_REC_PACKED=0: 16 + 4, A/B)
    // (round 5, later: the few-streams path as well — its one-pass sort moves 16 bytes per record and leaves the key words behind)
    db->rec_pshift = (db->width <= 54u && !(getenv("KMDB_REC_PACKED") && getenv("KMDB_REC_PACKED")[0] == '0')) ? db->width : 0u;
    // The second level (many streams, <= 256 blocks: see blocks_attempt) writes no records for the nodes with L2_MIN_BLOCKS blocks or more — 52 % of the
    // estimate at 10 000 samples.  Round 5 sized the pools from the estimate before it: 110 GB for 616 M records of 16 bytes.
    {
        const char* l2_env = getenv("KMDB_L2");
        if (db->row_mode && db->NB <= 256u && !(l2_env && l2_env[0] == '0')) {
            if (verbose) fprintf(stderr, "[kmdb] block records estimated: %llu, of them %llu from nodes the second level joins per tile instead\n",
                                 (unsigned long long)(est_n + est_g), (unsigned long long)est_l2);
            est_g -= std::min(est_g, est_l2);
        }
    }
    db->n_ckeys = db->n_states;                                  // keys of the grouped chunk table = the streams (row chunks sit in their rows' lists)
    {
        int key_bits = 1;
        while ((1ull << key_bits) <= (uint64_t)db->n_states + 1) ++key_bits;          // streams; all-ones = never written
        db->key_bits = key_bits;
    }
    HIP_TRY(hipMalloc((void**)&db->tile_touched, (size_t)db->n_states + 1));
    HIP_TRY(hipMalloc((void**)&db->ct_hist, ((size_t)db->n_ckeys + 2) * 4));
    HIP_TRY(hipMalloc((void**)&db->ct_offs, ((size_t)db->n_ckeys + 2) * 4));
    HIP_TRY(hipMalloc((void**)&db->ct_cursor, ((size_t)db->n_ckeys + 2) * 4));
    HIP_TRY(prim::exclusive_sum(nullptr, db->ct_tmp_bytes, db->ct_hist, db->ct_offs, (int)(db->n_ckeys + 1), db->stream));
    HIP_TRY(hipMalloc(&db->ct_tmp, std::max<size_t>(db->ct_tmp_bytes, 16)));
    // A database whose records do not fit one pass (2^31 record slots, or the memory left) is taken in SLICES of the pattern stream:
    // every slice emits the records of its own patterns only, the partial matrices add up in place (the engine's own version of
    // kmdb_opts.shard_index / shard_count; the reference blocks by `-buffer` the same way, similarity_calculator.cpp:290-325).
    {
        uint64_t slices = 1;
        const uint64_t slots = (est_n + est_g) * 3 / 2;
        const uint64_t usable = pool_slot_limit(db) / 4 * 3;                                            // three quarters of the slots a pool can index
        slices = std::max<uint64_t>(slices, (slots + usable - 1) / usable);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b)
            slices = std::max<uint64_t>(slices, (slots * 44 + free_b / 2) / std::max<size_t>(free_b * 6 / 10, 1));     // pool + sorted copy: 44 B per slot, 60 % of what is free
        if (const char* e = getenv("KMDB_SLICES")) if (*e) slices = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
        db->n_slices = (uint32_t)std::min<uint64_t>(slices, 1u << 16);
        if (verbose && db->n_slices > 1) fprintf(stderr, "[kmdb] %llu block records estimated: %u slices of the pattern stream per call\n",
                                                 (unsigned long long)(est_n + est_g), db->n_slices);
        est_n = est_n / db->n_slices + 1; est_g = est_g / db->n_slices + 1;
    }
    if (const char* e = getenv("KMDB_POOL_PERCENT")) if (*e) {
        // (tests: pools far too small, so that the first call takes the enlarge-and-repeat path on every one of them)
        const uint64_t pct = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
        est_n = est_n * pct / 100 + 1; est_g = est_g * pct / 100 + 1;
    }
    // waves of the wide-node kernel: in row mode every wave may end with an open chunk per block row
    db->k1w_waves = K1W_MAX_WAVES;
    if (db->row_mode) {
        db->k1w_waves = std::min<uint32_t>(db->k1w_waves, std::max<uint32_t>(256u, std::min<uint32_t>(4096u, (1u << 21) / std::max<uint32_t>(db->NB, 1u))));
        HIP_TRY(hipMalloc((void**)&db->rs_rows, (size_t)2 * (db->NB + 1) * 4));
    }
    HIP_TRY(hipMalloc((void**)&db->k2j_start, ((size_t)db->n_states + 2) * 4));
    // stream chunks: the narrow estimate at two thirds average fill, plus what the waves hold when they end (open chunks, an
    // unfinished grab); wide records: the wide estimate at two thirds (+ a grab / the open row chunks per wave)
    // (modes 1 and 2: of the narrow kernel's first-block records only the stragglers go to the pools — weights of 128 and more, in mode 1 also a
    // slice's nodes outside its tile: a sixteenth of the estimate is generous, and a pool that is too small is enlarged as ever)
    if (db->k1n_mode) { est_g += est_n / 16 + 1; est_n = 0; }
    if (db->row_mode) {
        // the estimate (one node in `stride`, exact per node) plus a quarter, plus what the waves hold when they end (open chunks, unfinished grabs); the
        // sorted copy holds the records themselves: the estimate plus an eighth.  A pool that turns out too small is enlarged and the call repeated.
        // (chunks, not records, are what runs out: every wave keeps an open chunk per block row it writes to — the wide kernel's waves all rows, a
        // slice of the narrow kernel the rows of its nodes' second blocks, up to 48 counted here — and leaves it partly filled.  Round 5's pools hid
        // that behind the records the second level no longer writes; with the estimate after the second level and 12 chunks per slice the first
        // call at 10 000 samples ran out and doubled its pool: profiles/r06_close_bench.err)
        if (alloc_record_pool(db, (est_n + est_g) * 5 / 4 / CH_REC + (uint64_t)db->n_nsegs * std::min<uint32_t>(db->NB, 48u) + (uint64_t)db->k1w_waves * (db->NB + ARENA_GRAB) + 4096)) return 1;
        if (alloc_wide_pool(db, (est_n + est_g) * 9 / 8 / WCH_REC + 4096)) return 1;
    } else {
        if (alloc_record_pool(db, est_n * 3 / 2 / CH_REC + (db->dense_narrow ? 0u : (uint64_t)db->n_nsegs * 8) + 4096)) return 1;
        if (alloc_wide_pool(db, est_g * 3 / 2 / WCH_REC + (uint64_t)(K1W_MAX_WAVES + 64) * (WIDE_GRAB + 2) + (uint64_t)db->n_nsegs * (WIDE_GRAB / 2) + 1024)) return 1;
    }
    if (verbose) fprintf(stderr, "[kmdb] prepare: record pools %.2f GB\n", kmdb_blocks_device_bytes(db) / 1e9);
    phase("record pools (hipMalloc)");
    return 0;
}

void kmdb_blocks_release(kmdb_db* db) {
    FREE_NULL(db->p0_mask); FREE_NULL(db->p0_info); FREE_NULL(db->pair_ofs); FREE_NULL(db->pair_blk); FREE_NULL(db->pair_mask);
    FREE_NULL(db->pair_cursor); FREE_NULL(db->fn_mask); FREE_NULL(db->fn_blk); FREE_NULL(db->widebits); FREE_NULL(db->wide_cnt);
    FREE_NULL(db->wide_base); FREE_NULL(db->widx); FREE_NULL(db->wrun_anc); FREE_NULL(db->wrun_anc_n); FREE_NULL(db->chunk_key); FREE_NULL(db->chunk_fill); FREE_NULL(db->sorted_key);
    FREE_NULL(db->sorted_id); FREE_NULL(db->chunk_iota); FREE_NULL(db->sort_tmp);
    FREE_NULL(db->wkey); FREE_NULL(db->wrec); FREE_NULL(db->swkey); FREE_NULL(db->swrec); FREE_NULL(db->sort2_tmp); FREE_NULL(db->wsub_cursor); FREE_NULL(db->cs_hist); FREE_NULL(db->cs_offs); FREE_NULL(db->cs_rows); FREE_NULL(db->cs_tmp);
    FREE_NULL(db->rec); FREE_NULL(db->recw); FREE_NULL(db->counters); FREE_NULL(db->scan_tmp); FREE_NULL(db->sub_cursor);
    FREE_NULL(db->ct_hist); FREE_NULL(db->ct_offs); FREE_NULL(db->ct_cursor); FREE_NULL(db->ct_tmp); FREE_NULL(db->rs_rows); FREE_NULL(db->rs_hist); FREE_NULL(db->rs_offs);
    FREE_NULL(db->run_ctr); FREE_NULL(db->direct_ctr); FREE_NULL(db->dmask); FREE_NULL(db->dwx); FREE_NULL(db->slice_cnt); FREE_NULL(db->rs_tmp); FREE_NULL(db->rg_hist); FREE_NULL(db->rg_offs); FREE_NULL(db->rg_tmp); FREE_NULL(db->row_ids); FREE_NULL(db->k2j_start); FREE_NULL(db->k2j_jobs); FREE_NULL(db->tile_touched);
    FREE_NULL(db->l2_cursors); FREE_NULL(db->l2_bitmap); FREE_NULL(db->l2_rank); FREE_NULL(db->l2_len); FREE_NULL(db->l2_loff); FREE_NULL(db->l2_ent_g);
    FREE_NULL(db->l2_node_w); FREE_NULL(db->l2_list_w); FREE_NULL(db->l2_ent_blk); FREE_NULL(db->l2_ent_mask); FREE_NULL(db->l2_list_mask);
    db->l2_node_cap = 0; db->l2_ent_cap = 0;
    if (db->h_counters) { (void)hipHostFree(db->h_counters); db->h_counters = nullptr; }
    db->pool_cap = 0; db->pair_cap = 0; db->wide_cap = 0;
}

uint64_t kmdb_blocks_device_bytes(const kmdb_db* db) {
    if (!db->counters) return 0;
    return db->P * (8 + 4 + 4 + 16 + 4) + (db->dmask ? (uint64_t)db->n_nsegs * db->nseg_nodes * 12 : 0u) + db->pair_cap * 10 + ((db->pool_cap << CH_SHIFT) * 20) + db->pool_cap * 16 + db->wide_cap * 4 + (db->P / 64) * 16 +
           (db->wide_pool_cap << WCH_SHIFT) * (db->row_mode ? 20 : 40) + db->rs_entries * 8 + (uint64_t)db->n_ckeys * 12 +
           (uint64_t)db->NB * (db->l2_node_cap / 64u) * 12 + (uint64_t)db->l2_ent_cap * 26 + (uint64_t)db->l2_node_cap * 4;
}

namespace {

// second level (L2View): arrays for node_cap node indices and ent_cap entries; frees what was there
int alloc_l2(kmdb_db* db, uint32_t node_cap, uint32_t ent_cap) {
    FREE_NULL(db->l2_cursors); FREE_NULL(db->l2_bitmap); FREE_NULL(db->l2_rank); FREE_NULL(db->l2_len); FREE_NULL(db->l2_loff); FREE_NULL(db->l2_ent_g);
    FREE_NULL(db->l2_node_w); FREE_NULL(db->l2_list_w); FREE_NULL(db->l2_ent_blk); FREE_NULL(db->l2_ent_mask); FREE_NULL(db->l2_list_mask);
    node_cap = (node_cap + 64u * L2_SUB - 1u) / (64u * L2_SUB) * (64u * L2_SUB);
    ent_cap = (ent_cap + L2_ENT_GRAB * L2_SUB - 1u) / (L2_ENT_GRAB * L2_SUB) * (L2_ENT_GRAB * L2_SUB);
    const size_t W = node_cap / 64u, NB = db->NB;
    HIP_TRY(hipMalloc((void**)&db->l2_cursors, 2 * L2_SUB * 16 * 4));
    HIP_TRY(hipMalloc((void**)&db->l2_bitmap, NB * W * 8));
    HIP_TRY(hipMalloc((void**)&db->l2_rank, NB * W * 4));
    HIP_TRY(hipMalloc((void**)&db->l2_len, NB * 4));
    HIP_TRY(hipMalloc((void**)&db->l2_loff, (NB + 2) * 4));
    HIP_TRY(hipMalloc((void**)&db->l2_ent_g, (size_t)ent_cap * 4));
    HIP_TRY(hipMalloc((void**)&db->l2_ent_blk, (size_t)ent_cap * 2));
    HIP_TRY(hipMalloc((void**)&db->l2_ent_mask, (size_t)ent_cap * 8));
    HIP_TRY(hipMalloc((void**)&db->l2_list_mask, (size_t)ent_cap * 8));
    HIP_TRY(hipMalloc((void**)&db->l2_list_w, (size_t)ent_cap * 4));
    HIP_TRY(hipMalloc((void**)&db->l2_node_w, (size_t)node_cap * 4));
    db->l2_node_cap = node_cap; db->l2_ent_cap = ent_cap;
    return 0;
}

// one attempt of the whole pipeline; *retry is set when a pool was too small (it has been enlarged)
// decode: run K0 (the first slice of a call does; what it leaves — the local (block, mask) pairs of every node — serves all slices)
int blocks_attempt(kmdb_db* db, uint32_t* M, uint32_t emit_lo, uint32_t emit_hi, bool decode, hipStream_t st, bool* retry) {
    *retry = false;
    // KMDB_SYNC_DEBUG: wait after every stage and name the one that failed (debugging aid, no effect on results)
    static const bool sync_debug = getenv("KMDB_SYNC_DEBUG") != nullptr;
    auto stage = [&](const char* name) -> int {
        if (!sync_debug) return 0;
        const hipError_t e = hipStreamSynchronize(st);
        fprintf(stderr, "[kmdb] stage %-14s %s\n", name, e == hipSuccess ? "ok" : hipGetErrorString(e));
        return e == hipSuccess ? 0 : kmdb_set_error(std::string("stage ") + name + ": " + hipGetErrorString(e));
    };
    const uint32_t P = (uint32_t)db->P;
    const BlockMap bm{db->width, (uint32_t)((1ull << 32) / db->width) + 1u};
    const uint32_t n_words = (P + 63) / 64;
    const uint32_t pool_cap = (uint32_t)db->pool_cap;
    const bool row_mode = db->row_mode;
    const uint32_t n_ckeys = db->n_ckeys;
    HIP_TRY(hipMemsetAsync(db->counters, 0, KCTR_COUNT * 4, st));
    if (decode) HIP_TRY(hipMemsetAsync(db->pair_cursor, 0, (KMDB_PAIR_REGIONS + 1) * 16 * 4, st));
    HIP_TRY(hipMemsetAsync(db->sub_cursor, 0, KMDB_SUBPOOLS * 16 * 4, st));
    HIP_TRY(hipMemsetAsync(db->wsub_cursor, 0, KMDB_SUBPOOLS * 16 * 4, st));
    HIP_TRY(hipMemsetAsync(db->run_ctr, 0, K1W_CTRS * 16 * 4, st));
    HIP_TRY(hipMemsetAsync(db->direct_ctr, 0, KMDB_SUBPOOLS * 16 * 4, st));
    // Sizes of the launches: measured by the previous call on this handle.  What the narrow kernel produces repeats exactly; where
    // the wide kernel's records land depends on which wave took which run, so the chunk counts behind it vary a little from call
    // to call: those launches get some slack, every kernel takes the true counts from device memory, and the call is repeated with
    // upper bounds if a true count exceeded its launch (checked at the end).
    auto with_slack = [](uint32_t v) -> uint32_t { return v + v / 8u + 1024u; };
    const uint32_t raw_launch = db->have_counts ? (uint32_t)std::min<uint64_t>(with_slack(db->last_n_raw), row_mode ? 0xFFFFFFFFull : db->wide_pool_cap) : 0u;
    if (!row_mode) {
        // never-written slots of the wide pool sort last; only the part the previous call used has to be reset
        const uint64_t wslots = db->have_counts ? (uint64_t)raw_launch << WCH_SHIFT : db->wide_pool_cap << WCH_SHIFT;
        HIP_TRY(hipMemsetAsync(db->wkey, 0xFF, wslots * 4, st));
    }
    HIP_TRY(hipMemsetAsync(db->chunk_fill, 0, (size_t)pool_cap * 4, st));
    HIP_TRY(hipMemsetAsync(db->ct_hist, 0, ((size_t)n_ckeys + 2) * 4, st));
    hipLaunchKernelGGL(fill_u32_kernel, dim3((pool_cap + 255) / 256), dim3(256), 0, st, db->chunk_key, pool_cap, 0xFFFFFFFFu);      // never opened
    if (stage("init")) return 1;
    // ---- K0  (the two launches one after the other: both are bound by instruction issue, side by side on two streams they take
    // exactly as long — measured, profiles/r03_k0side_ab.sh)
    if (decode) {
        K0Params q{};
        q.k0in = db->k0in; q.bitrel = db->bitrel; q.blkbase = db->blkbase; q.bits = db->bits; q.perm = nullptr; q.P = P; q.short_ids = db->short_max_ids; q.bm = bm;
        q.p0_mask = db->p0_mask; q.p0_info = db->p0_info; q.pair_ofs = db->pair_ofs; q.pair_blk = db->pair_blk; q.pair_mask = db->pair_mask;
        // sub-pools of the pair pool: enough of them that their cursors are not hot, few enough that one wave's need fits a share
        uint32_t nreg = 1;
        while (nreg < KMDB_PAIR_REGIONS && (uint64_t)nreg * 16384 < P) nreg <<= 1;
        // three quarters of the pool in sub-pools, the rest shared
        q.pair_cursor = db->pair_cursor; q.n_regions = nreg; q.region_cap = (uint32_t)(db->pair_cap * 3 / 4 / nreg);
        q.spill_cap = (uint32_t)(db->pair_cap - (uint64_t)q.region_cap * nreg); q.counters = db->counters;
        hipLaunchKernelGGL((k0_decode_kernel<false>), dim3((P + 255) / 256), dim3(256), 0, st, q);
        if (db->n_long) {
            q.perm = db->long_nodes; q.P = db->n_long;
            hipLaunchKernelGGL((k0_decode_kernel<true>), dim3((db->n_long + 255) / 256), dim3(256), 0, st, q);
        }
        HIP_TRY(hipGetLastError());
    }
    if (stage("decode")) return 1;
    HIP_TRY(hipEventRecord(db->ev_k[0], st));
    // ---- K1n
    {
        NParams q{};
        q.nl = db->nl; q.parent = db->parent; q.w = db->w; q.dflag = db->dflag; q.seg_anc = db->nseg_anc; q.seg_anc_n = db->nseg_anc_n;
        q.p0_mask = db->p0_mask; q.p0_info = db->p0_info; q.pair_ofs = db->pair_ofs; q.pair_blk = db->pair_blk; q.pair_mask = db->pair_mask;
        q.fn_mask = db->fn_mask; q.fn_blk = db->fn_blk; q.widebits = db->widebits;
        q.P = P; q.nseg_nodes = db->nseg_nodes; q.n_segs = db->n_nsegs; q.chain_cap = db->chain_cap;
        q.emit_lo = emit_lo; q.emit_hi = emit_hi; q.tbits = db->dense_narrow ? 0u : arena_table_bits(db->NB); q.n_keys = db->NB;
        q.all_wide = db->dense_narrow ? 1u : 0u;
        q.pool = pool_view(db, db->dense_narrow);
        // a wave's chain table grows with the depth of the tree: as many waves per workgroup as 144 KB of LDS hold
        const size_t wave_lds = k1n_wave_bytes(q.chain_cap, q.tbits, q.n_keys, row_mode ? db->NB : 0u);
        const uint32_t waves = (uint32_t)std::max<size_t>(1, std::min<size_t>(K1N_WAVES, (144u << 10) / wave_lds));
        const size_t lds = wave_lds * waves;
        q.dbg = getenv("KMDB_K1N_DBG") ? (uint32_t)atoi(getenv("KMDB_K1N_DBG")) : 0u;
        q.M = M; q.N = (uint32_t)db->N; q.bwidth = db->width; q.touched = db->tile_touched; q.direct_ctr = db->direct_ctr;
        q.dmask = db->dmask; q.dwx = db->dwx; q.slice_cnt = db->slice_cnt;
        // (mode 1 holds 48 accumulator registers beside the narrow kernel's 95: three waves per SIMD without spills, or four with 84 bytes of
        // scratch per lane — KMDB_K1N_MINW=4, A/B)
        static const int minw = getenv("KMDB_K1N_MINW") ? atoi(getenv("KMDB_K1N_MINW")) : 3;
        const void* fn = db->k1n_mode == 2 ? (const void*)k1n_kernel<2, 4> : db->k1n_mode == 1 ? (minw >= 4 ? (const void*)k1n_kernel<1, 4> : (const void*)k1n_kernel<1, 3>) : (const void*)k1n_kernel<0, 4>;
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const dim3 grid((q.n_segs + waves - 1) / waves), block(WAVE * waves);
        if (db->k1n_mode == 2) hipLaunchKernelGGL((k1n_kernel<2, 4>), grid, block, lds, st, q);
        else if (db->k1n_mode == 0) hipLaunchKernelGGL((k1n_kernel<0, 4>), grid, block, lds, st, q);
        else if (minw >= 4) hipLaunchKernelGGL((k1n_kernel<1, 4>), grid, block, lds, st, q);
        else hipLaunchKernelGGL((k1n_kernel<1, 3>), grid, block, lds, st, q);
        HIP_TRY(hipGetLastError());
    }
    if (stage("narrow emit")) return 1;
    HIP_TRY(hipEventRecord(db->ev_k[1], st));
    hipStream_t s2 = db->stream2;
    // The chunk table grouped by key (on stream `cs`), then on the side stream the stream chunks applied.  Few streams: right here,
    // next to the wide kernel (which writes no chunks).  Many streams: after the wide kernel (it writes the row chunks), next to
    // the sort inside the rows.
    auto group_and_apply_chunks = [&](hipStream_t cs) -> int {
        hipLaunchKernelGGL(ct_hist_kernel, dim3((pool_cap + CT_THREADS - 1) / CT_THREADS), dim3(CT_THREADS), 0, cs, db->chunk_key, pool_cap, n_ckeys, db->ct_hist);
        size_t tb = db->ct_tmp_bytes;
        HIP_TRY(prim::exclusive_sum(db->ct_tmp, tb, db->ct_hist, db->ct_offs, (int)(n_ckeys + 1), cs));
        HIP_TRY(hipMemcpyAsync(db->ct_cursor, db->ct_offs, ((size_t)n_ckeys + 1) * 4, hipMemcpyDeviceToDevice, cs));
        hipLaunchKernelGGL(ct_scatter_kernel, dim3((pool_cap + CT_THREADS - 1) / CT_THREADS), dim3(CT_THREADS), 0, cs, db->chunk_key, pool_cap, n_ckeys, db->ct_cursor, db->sorted_key, db->sorted_id);
        hipLaunchKernelGGL(ct_count_kernel, dim3(1), dim3(1), 0, cs, db->ct_offs, db->n_states, db->counters);
        if (cs != s2) { HIP_TRY(hipEventRecord(db->ev_side[0], cs)); HIP_TRY(hipStreamWaitEvent(s2, db->ev_side[0], 0)); }
        {
            rocprim::transform_iterator<uint32_t*, U32toU64, unsigned long long> it(db->chunk_fill, U32toU64());
            HIP_TRY(prim::sum(db->sort_tmp, db->sort_tmp_bytes, it, (unsigned long long*)(db->counters + KCTR_RECORDS), (int)pool_cap, s2));
        }
        const uint32_t win = db->n_states <= CS_MAX_KEYS ? 16u : 32u;      // (8 / 32 / 64 chunks at few streams: within the boxes' noise, profiles/r05_j5)
        uint32_t grid = (pool_cap + win - 1) / win;
        if (db->have_counts) grid = std::min(grid, (db->last_n_chunks + win - 1) / win);
        if (grid)
            hipLaunchKernelGGL(k2_apply_kernel, dim3(grid), dim3(256), 0, s2, db->rec, db->recw, db->sorted_key, db->sorted_id, db->chunk_fill, db->n_states,
                               db->ct_offs + db->n_states, M, (uint32_t)db->N, db->width, win, db->tile_touched);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(db->ev_side[1], s2));
        if (sync_debug) { const hipError_t e = hipStreamSynchronize(s2); fprintf(stderr, "[kmdb] stage %-14s %s\n", "chunk apply", e == hipSuccess ? "ok" : hipGetErrorString(e)); }
        return 0;
    };
    auto apply_slices = [&]() -> int {
        // mode 2: the slices' first-block records through k2d_kernel, on the side stream beside the wide kernel
        HIP_TRY(hipEventRecord(db->ev_side[0], st));
        HIP_TRY(hipStreamWaitEvent(s2, db->ev_side[0], 0));
        DParams d{};
        d.dmask = db->dmask; d.dwx = db->dwx; d.slice_cnt = db->slice_cnt; d.n_segs = db->n_nsegs; d.nseg_nodes = db->nseg_nodes;
        static const uint32_t sl = getenv("KMDB_K2D_SLICES") ? (uint32_t)std::max(1, atoi(getenv("KMDB_K2D_SLICES"))) : K2D_SLICES;
        d.slices = sl; d.M = M; d.N = (uint32_t)db->N; d.bwidth = db->width; d.touched = db->tile_touched;
        const uint32_t waves = (db->n_nsegs + sl - 1u) / sl;
        if (waves) hipLaunchKernelGGL(k2d_kernel, dim3((waves + 3u) / 4u), dim3(256), 0, s2, d);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(db->ev_side[1], s2));               // (the call's end waits for the side stream; many streams: recorded again behind the chunk table's work)
        if (sync_debug) { const hipError_t e = hipStreamSynchronize(s2); fprintf(stderr, "[kmdb] stage %-14s %s\n", "slice apply", e == hipSuccess ? "ok" : hipGetErrorString(e)); }
        return 0;
    };
    // (the slices' apply kernel starts right behind the narrow kernel, beside the wide list's small kernels — scan, expand, root paths of the runs,
    // which then take 0.28 ms at C2 instead of their own 0.05.  KMDB_K2D_EARLY=0 starts it with the wide kernel: the wide stage gains 0.1 ms, the
    // sort beside which the kernel then ends loses them — profiles/r06_j13: 6.70 / 6.98 against 6.86 / 6.93 ms at C2, 18.45 against 19.11 at 10 000 samples)
    static const bool k2d_early = !(getenv("KMDB_K2D_EARLY") && getenv("KMDB_K2D_EARLY")[0] == '0');
    if (db->k1n_mode == 2 && k2d_early && apply_slices()) return 1;
    if (!row_mode) {
        if (!db->dense_narrow) {
            HIP_TRY(hipEventRecord(db->ev_side[0], st));
            HIP_TRY(hipStreamWaitEvent(s2, db->ev_side[0], 0));
            if (group_and_apply_chunks(s2)) return 1;
        } else HIP_TRY(hipEventRecord(db->ev_side[1], s2));        // (no stream chunk is ever opened: nothing to group; the side stream holds the slices' apply kernel or nothing)
    }
    // ---- wide list
    hipLaunchKernelGGL(wide_count_kernel, dim3((n_words + 1 + 255) / 256), dim3(256), 0, st, db->widebits, n_words, db->wide_cnt);
    HIP_TRY(prim::exclusive_sum(db->scan_tmp, db->scan_tmp_bytes, db->wide_cnt, db->wide_base, (int)(n_words + 1), st));
    uint32_t n_wide;
    if (db->have_counts && db->wide_cap >= db->last_n_wide) n_wide = db->last_n_wide;     // deterministic per database; checked at the end of the call
    else {
        HIP_TRY(hipMemcpyAsync(db->h_counters, db->wide_base + n_words, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_wide = db->h_counters[0];
        if (n_wide > db->wide_cap) {
            FREE_NULL(db->widx); FREE_NULL(db->wrun_anc); FREE_NULL(db->wrun_anc_n);
            HIP_TRY(hipMalloc((void**)&db->widx, (size_t)n_wide * 4));
            HIP_TRY(hipMalloc((void**)&db->wrun_anc, ((size_t)n_wide / 64 + 1) * db->chain_cap * 4));
            HIP_TRY(hipMalloc((void**)&db->wrun_anc_n, ((size_t)n_wide / 64 + 1) * 4));
            db->wide_cap = n_wide;
        }
    }
    if (!db->widx) { HIP_TRY(hipMalloc((void**)&db->widx, 4)); db->wide_cap = std::max<uint64_t>(db->wide_cap, 1); }
    hipLaunchKernelGGL(wide_expand_kernel, dim3((n_words + 255) / 256), dim3(256), 0, st, db->widebits, db->wide_base, n_words, db->widx,
                       (uint32_t)db->wide_cap, db->counters);
    if (db->k1n_mode == 2 && !k2d_early && apply_slices()) return 1;
    // ---- K1w
    if (n_wide) {
        WParams q{};
        q.widx = db->widx; q.n_wide = n_wide; q.nl = db->nl; q.parent = db->parent; q.w = db->w; q.widebits = db->widebits;
        q.dflag = db->dflag; q.wide_base = db->wide_base;
        q.p0_mask = db->p0_mask; q.p0_info = db->p0_info; q.pair_ofs = db->pair_ofs; q.pair_blk = db->pair_blk; q.pair_mask = db->pair_mask;
        q.fn_mask = db->fn_mask; q.fn_blk = db->fn_blk; q.emit_lo = emit_lo; q.emit_hi = emit_hi; q.pool = pool_view(db, db->dense_wide);
            // ---- second level: many streams, few enough blocks that a tile job's bitmap scan stays small (tiles x nodes: quadratic in the
        // blocks), KMDB_L2=0 switches it off, KMDB_L2_MIN=c moves the threshold
        {
            const char* l2_env = getenv("KMDB_L2");
            const char* l2_min_env = getenv("KMDB_L2_MIN");
            db->l2_on = row_mode && db->NB <= 256u && !(l2_env && l2_env[0] == '0');
            db->l2_min_blocks = l2_min_env ? std::max<uint32_t>(K1W_HEAVY, (uint32_t)atoi(l2_min_env)) : L2_MIN_BLOCKS;
            if (db->l2_on && !db->l2_bitmap) {
                // a first guess from the wide list (at 10 000 samples 2 % of the wide nodes have 24 blocks or more); a call that runs out doubles both
                // (KMDB_POOL_PERCENT, tests: the smallest arrays there are, so that the doubling path runs)
                const bool tiny = getenv("KMDB_POOL_PERCENT") != nullptr;
                if (alloc_l2(db, tiny ? 1u : std::max<uint32_t>(64u * L2_SUB * 4u, n_wide / 16u), tiny ? 1u : std::max<uint32_t>(L2_ENT_GRAB * L2_SUB * 4u, n_wide / 16u * 48u))) return 1;
            }
            q.l2 = L2View{};
            if (db->l2_on) {
                q.l2.on = 1u; q.l2.min_blocks = db->l2_min_blocks; q.l2.node_cap = db->l2_node_cap; q.l2.W = db->l2_node_cap / 64u; q.l2.ent_cap = db->l2_ent_cap;
                q.l2.cursors = db->l2_cursors; q.l2.bitmap = db->l2_bitmap; q.l2.ent_g = db->l2_ent_g; q.l2.ent_blk = db->l2_ent_blk; q.l2.ent_mask = db->l2_ent_mask;
                q.l2.node_w = db->l2_node_w;
                HIP_TRY(hipMemsetAsync(db->l2_cursors, 0, 2 * L2_SUB * 16 * 4, st));
                HIP_TRY(hipMemsetAsync(db->l2_bitmap, 0, (size_t)db->NB * q.l2.W * 8, st));
                HIP_TRY(hipMemsetAsync(db->l2_ent_blk, 0xFF, (size_t)db->l2_ent_cap * 2, st));
            }
        }
        q.run_nodes = K1W_RUN_NODES;
        if (const char* e = getenv("KMDB_K1W_RUN")) q.run_nodes = std::max<uint32_t>(64u, (uint32_t)atoi(e) / 64u * 64u);
        q.n_runs = (n_wide + q.run_nodes - 1) / q.run_nodes;
        // root paths of the runs' first nodes (the table is sized with the wide list, for runs of 64 nodes at least)
        hipLaunchKernelGGL(wrun_anc_kernel, dim3((q.n_runs + 63) / 64), dim3(64), 0, st, db->widx, n_wide, q.run_nodes, db->parent, db->dflag, db->chain_cap,
                           db->wrun_anc, db->wrun_anc_n);
        q.seg_anc = db->wrun_anc; q.seg_anc_n = db->wrun_anc_n;
        
        // LDS of a wave: rows of a batch (at least one full list: as many entries as there are blocks), the chain list, the chain
        q.chain_cap = db->chain_cap; q.e_cap = (db->NB + 2u + 3u) & ~3u; q.arena_cap = std::max<uint32_t>(K1W_ARENA_MIN, (db->NB + 2u + 63u) & ~63u);
        q.n_rows = row_mode ? db->NB : 0u;
        const size_t wave_lds = k1w_wave_bytes(q.arena_cap, q.e_cap, q.chain_cap, q.n_rows);
        const uint32_t waves = wave_lds * K1W_WAVES <= (size_t)(64u << 10) ? (uint32_t)K1W_WAVES : 1u;
        HIP_TRY(hipFuncSetAttribute((const void*)k1w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wave_lds * waves)));
        // As many waves as the chip holds at once, each with an equal share of the runs (dealt round-robin): with more, the waves
        // beyond the first round start when the first ones end, and all take equally long — 1.6 rounds cost 2 (measured at 10 000
        // samples: 4096 waves at 10 per CU took 8.8 ms, of which every wave ran 4.4).
        if (!db->k1w_slots) {
            int per_cu = 0, dev = 0;
            hipDeviceProp_t prop;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k1w_kernel, (int)(WAVE * waves), wave_lds * waves));
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            db->k1w_slots = (uint32_t)std::max(1, per_cu) * waves * (uint32_t)std::max(1, prop.multiProcessorCount);
        }
        q.n_waves = std::min<uint32_t>(std::min<uint32_t>(db->k1w_slots, db->k1w_waves), q.n_runs);
        q.run_ctr = db->run_ctr;
        if (const char* e = getenv("KMDB_K1W_WAVES")) q.n_waves = std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(db->k1w_waves, q.n_runs), (uint32_t)atoi(e)));
        if (!db->have_counts && getenv("KMDB_VERBOSE"))
            fprintf(stderr, "[kmdb] wide kernel: %u wide nodes in %u runs, %u waves (%u fit the chip, %zu B of LDS per wave, %u per workgroup)\n", n_wide, q.n_runs, q.n_waves,
                    db->k1w_slots, wave_lds, waves);
        hipLaunchKernelGGL(k1w_kernel, dim3((q.n_waves + waves - 1) / waves), dim3(WAVE * waves), wave_lds * waves, st, q);
        if (db->l2_on) {
            // rank directories, list offsets, lists, and the tile joins adding into M (a tile whose blocks have no list leaves at once)
            const uint32_t NB = db->NB, W = db->l2_node_cap / 64u;
            hipLaunchKernelGGL(l2_ranks_kernel, dim3(NB), dim3(256), 0, st, db->l2_bitmap, W, db->l2_rank, db->l2_len);
            hipLaunchKernelGGL(l2_offsets_kernel, dim3(1), dim3(64), 0, st, db->l2_len, NB, db->l2_cursors, W, db->l2_loff);
            hipLaunchKernelGGL(l2_lists_kernel, dim3((db->l2_ent_cap + 255u) / 256u), dim3(256), 0, st, db->l2_ent_g, db->l2_ent_blk, db->l2_ent_mask, db->l2_node_w,
                               db->l2_ent_cap, W, db->l2_bitmap, db->l2_rank, db->l2_loff, db->l2_list_mask, db->l2_list_w, db->l2_cursors);
            hipLaunchKernelGGL(l2_join_apply_kernel, dim3(NB * (NB + 1u) / 2u), dim3(64 * L2_WAVES), 0, st, db->l2_bitmap, db->l2_rank, db->l2_list_mask, db->l2_list_w,
                               db->l2_loff, W, NB, M, (uint32_t)db->N, db->width, db->tile_touched);
        }
    }
    HIP_TRY(hipGetLastError());
    if (stage("wide emit")) return 1;
    HIP_TRY(hipEventRecord(db->ev_k[2], st));
    const uint32_t kmask = (1u << db->key_bits) - 1u;
    uint32_t jobs_launched = 0, k2jobs_launched = 0;
    if (row_mode) {
        // ---- many streams: chunk table grouped (stream chunks -> side stream), then the sort inside the block rows and its apply
        if (group_and_apply_chunks(st)) return 1;
        const uint32_t NB = db->NB;
        uint32_t* row_job = db->rs_rows, *row_tab = db->rs_rows + (NB + 1);
        {
            // the row chunks grouped by row
            const size_t rg_ne = (size_t)NB * db->rg_blocks + 1;
            hipLaunchKernelGGL(rg_hist_kernel, dim3(db->rg_blocks), dim3(RG_THREADS), NB * 4, st, db->chunk_key, pool_cap, db->n_states, NB, db->rg_hist);
            size_t tbg = db->rg_tmp_bytes;
            HIP_TRY(prim::exclusive_sum(db->rg_tmp, tbg, db->rg_hist, db->rg_offs, (int)rg_ne, st));
            hipLaunchKernelGGL(rg_scatter_kernel, dim3(db->rg_blocks), dim3(RG_THREADS), NB * 4, st, db->chunk_key, pool_cap, db->n_states, NB, db->rg_offs, db->row_ids);
        }
        hipLaunchKernelGGL(rs_rows_kernel, dim3(1), dim3(1024), 0, st, db->rg_offs, db->rg_blocks, NB, row_job, row_tab, db->counters);
        const RsRows R{db->rg_offs, db->row_ids, db->rg_blocks, NB, row_job, row_tab};
        // jobs and table entries: measured by the previous call, else their upper bounds (workgroups beyond the last job leave at once)
        const uint32_t jobs_bound = (uint32_t)(pool_cap / RS_JOB_CHUNKS + NB + 1);
        const uint32_t jobs = db->have_counts ? std::min<uint32_t>(jobs_bound, with_slack(db->last_n_rowjobs)) : jobs_bound;
        jobs_launched = jobs;
        const size_t ne = std::min<size_t>(db->rs_entries, (size_t)jobs * NB + 1);
        HIP_TRY(hipMemsetAsync(db->rs_hist, 0, ne * 4, st));
        {
            // (4, round 4's, against 8 / 16 at 10 000 samples: apply step 6.08 / 5.88 / 5.85 ms, profiles/r05_j14 — the kernel waits for its loads)
            const char* ue = getenv("KMDB_RSH_UNROLL");
            const int u = ue ? atoi(ue) : 16;
            if (u == 16) hipLaunchKernelGGL(rs_hist_kernel<16>, dim3(jobs), dim3(256), NB * 4, st, R, db->chunk_fill, db->recw, kmask, db->rs_hist);
            else if (u == 8) hipLaunchKernelGGL(rs_hist_kernel<8>, dim3(jobs), dim3(256), NB * 4, st, R, db->chunk_fill, db->recw, kmask, db->rs_hist);
            else hipLaunchKernelGGL(rs_hist_kernel<4>, dim3(jobs), dim3(256), NB * 4, st, R, db->chunk_fill, db->recw, kmask, db->rs_hist);
        }
        size_t tb = db->rs_tmp_bytes;
        HIP_TRY(prim::exclusive_sum(db->rs_tmp, tb, db->rs_hist, db->rs_offs, (int)ne, st));
        const uint32_t* total_ptr = db->rs_offs + (ne - 1);
        HIP_TRY(hipMemcpyAsync(db->counters + KCTR_WIDE_RECORDS, total_ptr, 4, hipMemcpyDeviceToDevice, st));
        // (Sorting and applying bands of block rows side by side on two streams was measured at 10 000 samples: 2 / 4 / 8 bands 18.4 / 17.0 /
        // 18.7 ms against 16.5 for one — the two kernels do not complement each other.)
        HIP_TRY(hipFuncSetAttribute((const void*)rs_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs_scatter_lds(NB, db->rec_pshift != 0)));
        hipLaunchKernelGGL(rs_scatter_kernel, dim3((jobs + 7u) & ~7u), dim3(CS_THREADS), rs_scatter_lds(NB, db->rec_pshift != 0), st, R, db->chunk_fill, db->recw, (const WideRec*)db->rec, kmask,
                           db->rs_offs, (uint32_t)db->sorted_cap, db->swkey, (WideRec*)db->swrec, db->counters, db->rec_pshift ? 1u : 0u);
        // the sorted records applied stream by stream (a long stream in parts of K2J_REC records): one write-back of a tile per job
        hipLaunchKernelGGL(k2j_starts_kernel, dim3((db->n_states + 1u + 255u) / 256u), dim3(256), 0, st, R, db->rs_offs, db->n_states, db->k2j_start);
        hipLaunchKernelGGL(k2j_build_kernel, dim3((db->n_states + 1023u) / 1024u), dim3(1024), 0, st, db->k2j_start, db->n_states, db->k2j_jobs, (uint32_t)db->k2j_cap, db->counters);
        k2jobs_launched = db->have_counts ? db->last_n_k2jobs : (uint32_t)db->k2j_cap;       // (the streams' lengths repeat exactly from call to call)
        if (k2jobs_launched)
            hipLaunchKernelGGL(k2_jobs_kernel, dim3(k2jobs_launched), dim3(256), 0, st, db->swkey, (const WideRec*)db->swrec, db->k2j_start, db->k2j_jobs, db->counters + KCTR_K2JOBS,
                               (uint32_t)db->k2j_cap, (uint32_t)db->sorted_cap, db->rec_pshift, (uint32_t)db->key_bits, wide_digit_bits(db->key_bits), M, (uint32_t)db->N, db->width, db->tile_touched);
        HIP_TRY(hipGetLastError());
        if (stage("row sort+apply")) return 1;
    } else {
        hipLaunchKernelGGL(count_raw_kernel, dim3(1), dim3(1), 0, st, db->wsub_cursor, db->counters);
        // the wide pool: records sorted by stream (the sort moves the 16-byte records with their key words), one tile per run
        uint32_t n_raw;
        if (db->have_counts) n_raw = raw_launch;
        else {
            HIP_TRY(hipMemcpyAsync(db->h_counters, db->counters, KCTR_COUNT * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            n_raw = std::min<uint32_t>(db->h_counters[KCTR_RAW], (uint32_t)db->wide_pool_cap);
        }
        if (n_raw) {
            const uint32_t nslots = (uint32_t)((uint64_t)n_raw << WCH_SHIFT);
            const CsRows no_rows{};
            const size_t ne = (size_t)db->n_states * CS_BLOCKS_ONE + 1;
            hipLaunchKernelGGL(cs_hist_kernel, dim3(CS_BLOCKS_ONE), dim3(256), db->n_states * 4, st, db->wkey, nslots, db->n_states, db->n_states, (int)CS_BY_STREAM,
                               no_rows, kmask, db->cs_hist);
            size_t tb = db->cs_tmp_bytes;
            HIP_TRY(prim::exclusive_sum(db->cs_tmp, tb, db->cs_hist, db->cs_offs, (int)ne, st));
            const bool packed = db->rec_pshift != 0;
            HIP_TRY(hipFuncSetAttribute((const void*)cs_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cs_scatter_lds(db->n_states, packed)));
            hipLaunchKernelGGL(cs_scatter_kernel, dim3(CS_BLOCKS_ONE), dim3(CS_THREADS), cs_scatter_lds(db->n_states, packed), st, db->wkey, (const WideRec*)db->wrec, nslots,
                               db->n_states, db->n_states, (int)CS_BY_STREAM, no_rows, kmask, db->cs_offs, db->swkey, (WideRec*)db->swrec, packed ? 1u : 0u);
            const uint32_t* total_ptr = db->cs_offs + (ne - 1);
            HIP_TRY(hipMemcpyAsync(db->counters + KCTR_WIDE_RECORDS, total_ptr, 4, hipMemcpyDeviceToDevice, st));
            if (getenv("KMDB_K2_WINDOWS") && !packed) {
                // (A/B: round 4's apply step over windows of 4096 sorted positions, run boundaries searched in the key words)
                const uint32_t g2 = (nslots + K2S_WIN - 1) / K2S_WIN;
                hipLaunchKernelGGL(k2_sorted_kernel, dim3(g2), dim3(256), 0, st, db->swkey, (const WideRec*)db->swrec, nslots, total_ptr, (const uint32_t*)nullptr,
                                   db->n_states, (uint32_t)db->key_bits, wide_digit_bits(db->key_bits), M, (uint32_t)db->N, db->width, 0u, 0u, db->tile_touched);
            } else {
                // the sorted records applied stream by stream, as on the many-streams path: the streams start where the sort's offsets say (no run
                // boundaries searched, no key word read for packed records), a tile is written back once per K2J_REC records of its stream
                hipLaunchKernelGGL(k2j_starts_flat_kernel, dim3((db->n_states + 1u + 255u) / 256u), dim3(256), 0, st, db->cs_offs, db->n_states, CS_BLOCKS_ONE, db->k2j_start);
                hipLaunchKernelGGL(k2j_build_kernel, dim3((db->n_states + 1023u) / 1024u), dim3(1024), 0, st, db->k2j_start, db->n_states, db->k2j_jobs, (uint32_t)db->k2j_cap, db->counters);
                k2jobs_launched = db->have_counts ? db->last_n_k2jobs : (uint32_t)db->k2j_cap;
                if (k2jobs_launched)
                    hipLaunchKernelGGL(k2_jobs_kernel, dim3(k2jobs_launched), dim3(256), 0, st, db->swkey, (const WideRec*)db->swrec, db->k2j_start, db->k2j_jobs, db->counters + KCTR_K2JOBS,
                                       (uint32_t)db->k2j_cap, (uint32_t)db->sorted_cap, db->rec_pshift, (uint32_t)db->key_bits, wide_digit_bits(db->key_bits), M, (uint32_t)db->N, db->width, db->tile_touched);
            }
            HIP_TRY(hipGetLastError());
        }
        if (stage("sorted apply")) return 1;
    }
    HIP_TRY(hipStreamWaitEvent(st, db->ev_side[1], 0));          // the side stream's tiles are in the matrix too
    hipLaunchKernelGGL(pool_used_kernel, dim3(1), dim3(64), 0, st, db->sub_cursor, db->direct_ctr, db->counters);
    HIP_TRY(hipEventRecord(db->ev_k[3], st));
    // ---- what the call found
    HIP_TRY(hipMemcpyAsync(db->h_counters, db->counters, KCTR_COUNT * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t* c = db->h_counters;
    if (c[KCTR_LIST_OVERFLOW] || c[KCTR_SLOW]) {
        db->fallback_reason = "internal: the wide-node kernel lost a list (chain miss " + std::to_string(c[KCTR_SLOW]) + ", arena overflow " + std::to_string(c[KCTR_LIST_OVERFLOW]) + ")";
        return 0;
    }
    if (c[KCTR_L2_OVERFLOW]) {
        if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] second level out of node indices or entries (%u / %u): doubling\n", db->l2_node_cap, db->l2_ent_cap);
        if (alloc_l2(db, db->l2_node_cap * 2u, db->l2_ent_cap * 2u)) return 1;
        db->have_counts = false; *retry = true;
        return 0;
    }
    if (c[KCTR_PAIR_OVERFLOW] || c[KCTR_POOL_OVERFLOW] || c[KCTR_WIDE_OVERFLOW]) {
        const bool verbose = getenv("KMDB_VERBOSE") != nullptr;
        if (c[KCTR_PAIR_OVERFLOW]) {
            if (verbose) fprintf(stderr, "[kmdb] pair pool too small (%llu entries): doubling\n", (unsigned long long)db->pair_cap);
            if (alloc_pair_pool(db, db->pair_cap * 2)) return 1;
        }
        if (c[KCTR_WIDE_OVERFLOW]) {
            const uint64_t want = db->wide_pool_cap * 2;
            if ((want << WCH_SHIFT) >= pool_slot_limit(db)) {
                if (db->n_slices >= (1u << 16)) { db->fallback_reason = "more block records than a pool can index from the nodes with many blocks in a 65536th of the patterns"; return 0; }
                db->n_slices *= 2;                                  // the pools stay as they are: half as many patterns per pass
                if (verbose) fprintf(stderr, "[kmdb] wide record pool at its limit: %u slices of the pattern stream per call\n", db->n_slices);
                db->have_counts = false; *retry = true;
                return 0;
            }
            if (verbose) fprintf(stderr, "[kmdb] wide record pool too small (%llu chunks): doubling\n", (unsigned long long)db->wide_pool_cap);
            if (alloc_wide_pool(db, want)) return 1;
        }
        if (c[KCTR_POOL_OVERFLOW]) {
            uint64_t want = db->pool_cap * 2;
            if (!row_mode && !db->dense_narrow && db->n_states > (1u << ST_MAX_BITS) && (db->pool_cap << CH_SHIFT) > 8 * (db->est_records + (1u << 22))) {
                // the estimate is long covered: the narrow kernel's chunks are being evicted nearly empty.  Wide pool for it as well.
                db->dense_narrow = true;
                if (verbose) fprintf(stderr, "[kmdb] record chunks of the narrow kernel are evicted nearly empty: its records go through the sort too\n");
                if (alloc_wide_pool(db, db->wide_pool_cap + db->est_records * 5 / 4 / WCH_REC + (uint64_t)db->n_nsegs * (WIDE_GRAB + 2))) return 1;
                want = 0;
            } else if (row_mode && !db->dense_narrow && db->NB > (1u << ST_MAX_BITS) && (db->pool_cap << CH_SHIFT) > 8 * (db->est_records + (1u << 22))) {
                // the same with row chunks: the diagonal records join the rows
                db->dense_narrow = true;
                if (verbose) fprintf(stderr, "[kmdb] record chunks of the narrow kernel are evicted nearly empty: its records go through the row chunks too\n");
                if (alloc_wide_pool(db, db->wide_pool_cap + db->est_records * 5 / 4 / WCH_REC)) return 1;
                want = 0;
            } else if ((want << CH_SHIFT) >= pool_slot_limit(db)) {
                if (db->n_slices >= (1u << 16)) {
                    db->fallback_reason = "more block records than a pool can index in a 65536th of the patterns (" + std::to_string(db->n_states) + " streams, " + std::to_string(db->pool_cap) + " chunks were not enough)";
                    return 0;
                }
                db->n_slices *= 2;
                if (verbose) fprintf(stderr, "[kmdb] record pool at its limit: %u slices of the pattern stream per call\n", db->n_slices);
                db->have_counts = false; *retry = true;
                return 0;
            }
            if (want) {
                if (verbose) fprintf(stderr, "[kmdb] record pool too small (%llu chunks): %llu\n", (unsigned long long)db->pool_cap, (unsigned long long)want);
                if (alloc_record_pool(db, want)) return 1;
            }
        }
        db->have_counts = false; *retry = true;
        return 0;
    }
    if (db->have_counts && (c[KCTR_NWIDE] != db->last_n_wide || c[KCTR_CHUNKS] != db->last_n_chunks ||
                            (row_mode ? (c[KCTR_ROWJOBS] > jobs_launched || c[KCTR_WIDE_RECORDS] != db->last_n_sorted || c[KCTR_K2JOBS] > k2jobs_launched)
                                      : (c[KCTR_RAW] > raw_launch || c[KCTR_K2JOBS] > k2jobs_launched)))) {
        // the exact counts cannot differ for an unchanged database and emit range, and the varying ones stay inside their slack in
        // practice; if not: redo the call with upper bounds
        if (getenv("KMDB_VERBOSE"))
            fprintf(stderr, "[kmdb] a launch was sized too small (wide nodes %u / %u, chunks %u / %u, row jobs %u / %u, sorted %u / %u, apply jobs %u / %u, raw %u / %u): the call is repeated\n",
                    c[KCTR_NWIDE], db->last_n_wide, c[KCTR_CHUNKS], db->last_n_chunks, c[KCTR_ROWJOBS], jobs_launched, c[KCTR_WIDE_RECORDS], db->last_n_sorted, c[KCTR_K2JOBS],
                    k2jobs_launched, c[KCTR_RAW], raw_launch);
        db->have_counts = false; *retry = true;
        return 0;
    }
    // Headroom: where the wide kernel's records land depends on which wave took which run, so a pool that was just large enough this time may
    // be too small next time (a warm call would then take the enlarge-and-repeat path: seen once in the tests with pools of 1 %).  A pool
    // more than 85 % in use grows by a quarter now, while nothing runs.
    if ((uint64_t)c[KCTR_POOL_USED] * 100 > db->pool_cap * 85) {
        if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] record pool %u of %llu chunks in use: enlarged by a quarter\n", c[KCTR_POOL_USED], (unsigned long long)db->pool_cap);
        const uint64_t want = db->pool_cap + db->pool_cap / 4;
        if (((want + (uint64_t)KMDB_SUBPOOLS * ARENA_GRAB) << CH_SHIFT) < pool_slot_limit(db) && alloc_record_pool(db, want)) return 1;
    }
    if (!row_mode && (uint64_t)c[KCTR_RAW] * 100 > db->wide_pool_cap * 85) {
        if (getenv("KMDB_VERBOSE")) fprintf(stderr, "[kmdb] wide record pool %u of %llu chunks in use: enlarged by a quarter\n", c[KCTR_RAW], (unsigned long long)db->wide_pool_cap);
        const uint64_t want = db->wide_pool_cap + db->wide_pool_cap / 4;
        if (((want + (uint64_t)KMDB_SUBPOOLS * WIDE_GRAB) << WCH_SHIFT) < pool_slot_limit(db) && alloc_wide_pool(db, want)) return 1;
    }
    if (!db->have_counts && getenv("KMDB_VERBOSE"))
        fprintf(stderr, "[kmdb] record pool: %u of %llu chunks in use (busiest sub-pool x sub-pools); wide pool / sorted arrays: %llu slots\n", c[KCTR_POOL_USED],
                (unsigned long long)db->pool_cap, (unsigned long long)db->sorted_cap);
    db->last_n_wide = c[KCTR_NWIDE]; db->last_n_chunks = c[KCTR_CHUNKS]; db->last_n_raw = c[KCTR_RAW]; db->last_n_slow = c[KCTR_SLOW];
    db->last_n_rowjobs = c[KCTR_ROWJOBS]; db->last_n_sorted = c[KCTR_WIDE_RECORDS]; db->last_l2_nodes = c[KCTR_L2_NODES]; db->last_n_k2jobs = c[KCTR_K2JOBS];
    if (db->l2_on && !db->have_counts && getenv("KMDB_VERBOSE"))
        fprintf(stderr, "[kmdb] second level: %u nodes with %u blocks or more joined per tile (indices for %u, entries for %u)\n", c[KCTR_L2_NODES], db->l2_min_blocks,
                db->l2_node_cap, db->l2_ent_cap);
    db->last_records = ((uint64_t)c[KCTR_RECORDS] | ((uint64_t)c[KCTR_RECORDS_HI] << 32)) + (row_mode ? 0u : c[KCTR_WIDE_RECORDS]);
    db->last_n_direct = c[KCTR_DIRECT];
    return 0;
}

}  // namespace

// an attempt that fails half way may have work under way on the side streams: nothing of it may outlive the call (the caller frees M)
static void blocks_join_side_streams(kmdb_db* db) {
    if (db->stream2) (void)hipStreamSynchronize(db->stream2);
}

int kmdb_blocks_run(kmdb_db* db, uint32_t* M, uint32_t emit_lo, uint32_t emit_hi, hipStream_t st) {
    const uint64_t cells = db->N * (db->N - 1) / 2;
    for (int round = 0; round < 64; ++round) {
        if (db->tile_touched) HIP_TRY(hipMemsetAsync(db->tile_touched, 0, (size_t)db->n_states + 1, st));      // the tiles this call adds to (all its slices)
        // the slices of [emit_lo, emit_hi) one after the other, adding into the same matrix; a pass that had to enlarge a pool (or
        // asked for more slices) leaves a partial sum behind: everything again from a zeroed matrix
        const uint32_t S = std::max<uint32_t>(1u, db->n_slices);
        bool again = false;
        uint64_t records = 0;
        // the launch sizes a call measures belong to one emit range: kept per slice, so that a sliced database reaches the warm path too
        if (db->slice_counts.size() != S) db->slice_counts.assign(S, kmdb_db::SliceCounts{});
        bool decoded = false;
        db->last_call_sized = false;
        for (uint32_t k = 0; k < S && !again; ++k) {
            const uint32_t lo = emit_lo + (uint32_t)((uint64_t)(emit_hi - emit_lo) * k / S), hi = emit_lo + (uint32_t)((uint64_t)(emit_hi - emit_lo) * (k + 1) / S);
            if (lo == hi) continue;
            kmdb_db::SliceCounts& sc = db->slice_counts[k];
            db->have_counts = sc.valid && sc.lo == lo && sc.hi == hi;
            if (db->have_counts) { db->last_n_wide = sc.n_wide; db->last_n_chunks = sc.n_chunks; db->last_n_raw = sc.n_raw; db->last_n_rowjobs = sc.n_rowjobs; db->last_n_sorted = sc.n_sorted; db->last_n_k2jobs = sc.n_k2jobs; }
            db->last_emit_lo = lo; db->last_emit_hi = hi;
            if (!db->have_counts) db->last_call_sized = true;
            bool retry = false;
            if (blocks_attempt(db, M, lo, hi, !decoded, st, &retry)) { blocks_join_side_streams(db); return 1; }
            decoded = true;
            if (!db->fallback_reason.empty()) return 0;
            if (retry) { again = true; for (auto& c : db->slice_counts) c.valid = false; }
            else {
                sc = kmdb_db::SliceCounts{true, lo, hi, db->last_n_wide, db->last_n_chunks, db->last_n_raw, db->last_n_rowjobs, db->last_n_sorted, db->last_n_k2jobs};
                records += db->last_records;
            }
        }
        if (!again) { db->last_records = records; db->have_counts = true; return 0; }
        HIP_TRY(hipMemsetAsync(M, 0, cells * 4, st));
    }
    return kmdb_set_error("kmdb_blocks_run: the record pools did not converge");
}
