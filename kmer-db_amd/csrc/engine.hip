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
#include "kmdb_amd.h"
#include "kmdb_internal.h"
#include "engine_internal.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
int kmdb_set_error(const std::string& msg) { g_last_error = msg; return 1; }
extern "C" const char* kmdb_last_error(void) { return g_last_error.c_str(); }
extern "C" int kmdb_abi_version(void) { return KMDB_ABI_VERSION; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return kmdb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

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

// ------------------------------------------------------------------------------------------
// resident database
// ------------------------------------------------------------------------------------------
struct Segment { uint32_t first, end; };

struct kmdb_db {
    int device = 0;
    uint64_t N = 0, P = 0;
    uint4* meta = nullptr;          // {n, l, last_id, nbits} per node, DFS order
    uint64_t* bitpos = nullptr;     // absolute bit offset of the node's gamma stream
    int32_t* parent = nullptr;      // DFS index of the parent, -1 for roots
    uint32_t* w = nullptr;          // on-disk num_kmers truncated to u32, P+1 entries (last = 0)
    uint32_t* sub_end = nullptr;    // DFS index one past the node's subtree
    uint32_t* wprefix = nullptr;    // P+1, exclusive scan of w (recomputed by every call)
    uint64_t* bits = nullptr;
    uint64_t n_bit_words = 0;
    Segment* segs = nullptr;            // equal-COST slices (tree-form updates) for the v1 scatter kernels
    uint32_t n_segs = 0;
    Segment* rsegs = nullptr;           // equal-NODE-COUNT slices for the block-record emit kernels
    uint32_t n_rsegs = 0;
    void* scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    uint32_t* stack_scratch = nullptr;  // global kernel: per-wave id stacks
    size_t stack_scratch_words = 0;
    unsigned long long* counters = nullptr;   // [0] tile flushes
    // hashtables (new2all)
    uint64_t n_buckets = 0;
    uint64_t* bucket_offset = nullptr;
    uint64_t* slots = nullptr;
    uint32_t* pid2dfs = nullptr;    // original pattern id -> DFS index
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    kmdb_stats stats{};
    uint32_t kmer_length = 0;
    // v2 (block record) pipeline state, built at upload when the database qualifies
    bool b2_ready = false;
    uint32_t b2_maxn_pad = 0, b2_dec_cap = 0, b2_nctr = 0, b2_n_items = 0;
    uint32_t b2_width = 64;             // sample ids per block (<= 64), chosen at upload
    uint32_t* b2_table = nullptr;       // [n_segs][nctr] record bases
    unsigned long long* b2_rec_rows = nullptr;   // [total]
    ulonglong2* b2_rec_rc = nullptr;    // [total]
    uint32_t* b2_rec_w = nullptr;       // [total]
    void* b2_items = nullptr;           // B2Item[n_items]
    uint64_t b2_total = 0;
    hipEvent_t ev_k2[2] = {nullptr, nullptr};
    double k1_ms = 0, k2_ms = 0;
    // v3 front half (K0 decode + batch-parallel K1)
    bool b3_ready = false;
    uint32_t b3_nbw = 0;
    uint32_t* b3_perm = nullptr;        // the nodes with long local lists, longest first
    uint32_t b3_n_long = 0, b3_short_max = 32, b3_chain_cap = 64;
    uint32_t* b3_nl = nullptr;          // n | l << 16 per node
    uint32_t* b3_pair_ofs = nullptr;    // [P+1] CSR of the pairs beyond a node's first
    unsigned long long* b3_p0_mask = nullptr;   // [P] first pair, inline
    uint16_t* b3_p0_info = nullptr;     // [P] block | npairs << 8
    uint8_t* b3_pair_blk = nullptr;
    unsigned long long* b3_pair_mask = nullptr;
    uint32_t* b3_seg_anc = nullptr;     // [n_segs][B3_CHAIN]
    uint32_t* b3_seg_anc_n = nullptr;
    uint64_t b3_total_pairs = 0;
    hipEvent_t ev_k0 = nullptr;
    double k0_ms = 0;
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
namespace {

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int DEC_CAP = 1024;       // decoded local ids buffered per wave per batch

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// make this wave's earlier LDS / global writes visible to its other lanes (same CU: the
// workgroup-scope fence is enough, no cache maintenance involved)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ordering point for data that lives in LDS only: the LDS pipeline executes one wave's DS
// instructions in order, so only the compiler has to be kept from moving accesses across it.
// (wave_sync() also drains outstanding global stores, which costs microseconds per call.)
__device__ __forceinline__ void lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src_lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_lane);
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)v, d, WAVE);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

// Gamma streams are MSB-first in little-endian uint64 words (reference src/elias_gamma.h:113-125).
// BitCursor keeps three consecutive words in registers: a code (<= 63 bits) is extracted from
// c0:c1 with shifts only, and the word two ahead is fetched when the cursor crosses a word
// boundary, so the decode loop has no load on its dependency chain.  The bit array carries four
// padding words.
struct BitCursor {
    const uint64_t* __restrict__ bits;
    uint64_t wi;
    uint64_t c0, c1, c2;
    uint32_t s;                                    // bit offset inside c0
    __device__ __forceinline__ BitCursor(const uint64_t* __restrict__ b, uint64_t pos) : bits(b) {
        wi = pos >> 6;
        s = (uint32_t)pos & 63u;
        c0 = bits[wi]; c1 = bits[wi + 1]; c2 = bits[wi + 2];
    }
    // one Elias-gamma value: (L-1) ones, a zero, (L-1) low bits (reference src/elias_gamma.h:104-128)
    __device__ __forceinline__ uint32_t next() {
        const uint64_t win = s ? ((c0 << s) | (c1 >> (64u - s))) : c0;
        uint32_t ones = (uint32_t)__clzll((long long)~win);
        ones = ones > 31u ? 31u : ones;             // a valid code has at most 31 leading ones
        const uint32_t low = (uint32_t)((win << ones) >> (63u - ones));
        s += 2u * ones + 1u;
        if (s >= 64u) {
            s -= 64u;
            ++wi;
            c0 = c1; c1 = c2; c2 = bits[wi + 2];
        }
        return low | (1u << ones);
    }
};

// Same decoder with a deeper look-ahead (PF words in flight): for the threads that walk long streams
// alone, where the single look-ahead word of BitCursor leaves a full memory latency per 64 bits.
template <int PF>
struct BitCursorDeep {
    const uint64_t* __restrict__ bits;
    uint64_t wi;
    uint64_t c[PF + 2];
    uint32_t s;
    __device__ __forceinline__ BitCursorDeep(const uint64_t* __restrict__ b, uint64_t pos) : bits(b) {
        wi = pos >> 6;
        s = (uint32_t)pos & 63u;
#pragma unroll
        for (int k = 0; k < PF + 2; ++k) c[k] = bits[wi + k];
    }
    __device__ __forceinline__ uint32_t next() {
        const uint64_t win = s ? ((c[0] << s) | (c[1] >> (64u - s))) : c[0];
        uint32_t ones = (uint32_t)__clzll((long long)~win);
        ones = ones > 31u ? 31u : ones;
        const uint32_t low = (uint32_t)((win << ones) >> (63u - ones));
        s += 2u * ones + 1u;
        if (s >= 64u) {
            s -= 64u;
            ++wi;
#pragma unroll
            for (int k = 0; k < PF + 1; ++k) c[k] = c[k + 1];
            c[PF + 1] = bits[wi + PF + 1];
        }
        return low | (1u << ones);
    }
};

// Decode the l local ids of one node into out[0..l) (ascending).  pattern_t::decodeSamples
// (reference src/pattern.cpp:99-109): l-1 gamma-coded deltas in append order, last id explicit.
template <class T>
__device__ __forceinline__ void decode_node(const uint64_t* __restrict__ bits, uint64_t pos, uint32_t l, uint32_t last, T* out) {
    if (l == 0) return;
    if (l > 1) {
        BitCursor cur(bits, pos);
        uint32_t sum = 0;
        for (uint32_t i = 0; i + 1 < l; ++i) {
            const uint32_t d = cur.next();
            out[i] = (T)d;
            sum += d;
        }
        uint32_t id = last - sum;
        for (uint32_t i = 0; i + 1 < l; ++i) {
            const uint32_t d = (uint32_t)out[i];
            out[i] = (T)id;
            id += d;
        }
    }
    out[l - 1] = (T)last;
}

// Sample ids are grouped into blocks of `width` (<= 64) consecutive ids; the width is chosen per database
// at upload (a narrower block that matches the cluster structure of the samples means fewer block records).
struct BlockMap {
    uint32_t width, magic;                          // magic = floor(2^32 / width) + 1: exact division for ids < 2^16
    __host__ __device__ __forceinline__ uint32_t blk(uint32_t id) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return __umulhi(id, magic);
#else
        return id / width;
#endif
    }
    __host__ __device__ __forceinline__ uint32_t bit(uint32_t id, uint32_t b) const { return id - b * width; }
};

// decode_node plus, per id, the running bit mask of the ids of the same block seen so far
// in this node ("cum"): the block-record kernel needs it per stack position.
__device__ __forceinline__ void decode_node_cum(const uint64_t* __restrict__ bits, uint64_t pos, uint32_t l, uint32_t last,
                                                uint16_t* out, unsigned long long* cum, const BlockMap bm) {
    if (l == 0) return;
    uint32_t id = last;
    if (l > 1) {
        BitCursor cur(bits, pos);
        uint32_t sum = 0;
        for (uint32_t i = 0; i + 1 < l; ++i) {
            const uint32_t d = cur.next();
            out[i] = (uint16_t)d;
            sum += d;
        }
        id = last - sum;
    }
    uint32_t curblk = 0xFFFFFFFFu;
    unsigned long long acc = 0;
    for (uint32_t i = 0; i < l; ++i) {
        const uint32_t d = (i + 1 < l) ? (uint32_t)out[i] : 0u;
        const uint32_t blk = bm.blk(id);
        if (blk != curblk) { curblk = blk; acc = 0; }
        acc |= 1ull << bm.bit(id, blk);
        out[i] = (uint16_t)id;
        cum[i] = acc;
        id += d;
    }
}

__device__ __forceinline__ uint64_t tri64(uint64_t a) { return a * (a - 1) / 2; }

struct A2AParams {
    const uint4* meta;
    const uint64_t* bitpos;
    const int32_t* parent;
    const uint32_t* sub_end;
    const uint32_t* wprefix;
    const uint64_t* bits;
    const Segment* segs;
    uint32_t seg_begin, seg_end;
    uint32_t* M;                    // N(N-1)/2 lower-triangular matrix in HBM
    uint32_t* stack_scratch;        // global kernel only
    uint32_t stack_stride;          // words per wave
    unsigned long long* counters;
    uint32_t dbg;                   // timing experiments only: 2 = skip scatter, 4 = skip flush, 8 = skip mapping+scatter
};

// rebuild the id stack for the ancestors of `first` by walking parent links
template <class T>
__device__ __forceinline__ void init_stack(const A2AParams& p, uint32_t first, T* stack, uint32_t lane) {
    int32_t cur = p.parent[first];
    while (cur >= 0) {
        uint4 m = p.meta[cur];
        if (lane == 0) decode_node<T>(p.bits, p.bitpos[cur], m.y, m.z, stack + (m.x - m.y));
        cur = p.parent[cur];
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------
// generic kernel: stack in global scratch, HBM atomics
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void a2a_global_kernel(A2AParams p) {
    __shared__ uint32_t dec_all[WAVES_PER_BLOCK][DEC_CAP];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg = p.seg_begin + blockIdx.x * WAVES_PER_BLOCK + wave;
    if (seg >= p.seg_end) return;
    uint32_t* dec = dec_all[wave];
    uint32_t* stack = p.stack_scratch + (size_t)(blockIdx.x * WAVES_PER_BLOCK + wave) * p.stack_stride;
    const Segment sg = p.segs[seg];
    const uint32_t first = __builtin_amdgcn_readfirstlane(sg.first);
    const uint32_t end = __builtin_amdgcn_readfirstlane(sg.end);
    if (first >= end) return;
    init_stack<uint32_t>(p, first, stack, lane);

    for (uint32_t base = first; base < end;) {
        const uint32_t i = base + lane;
        const bool valid = i < end;
        uint4 m = valid ? p.meta[i] : make_uint4(0, 0, 0, 0);
        const uint64_t bp = valid ? p.bitpos[i] : 0;
        const uint32_t W = valid ? (p.wprefix[p.sub_end[i]] - p.wprefix[i]) : 0u;
        const uint32_t l = m.y;
        const uint32_t incl = wave_incl_scan(l, lane);
        const unsigned long long fit = __ballot(valid && incl <= (uint32_t)DEC_CAP);
        uint32_t cnt = fit == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fit);
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        const uint32_t off = incl - l;
        if (cnt == 0) {
            // a single node with more than DEC_CAP local ids: decode straight into the stack
            const uint32_t n0 = bcast(m.x, 0), l0 = bcast(m.y, 0), last0 = bcast(m.z, 0), W0 = bcast(W, 0);
            const uint64_t bp0 = ((uint64_t)bcast((uint32_t)(bp >> 32), 0) << 32) | bcast((uint32_t)bp, 0);
            const uint32_t top = n0 - l0;
            if (lane == 0) decode_node<uint32_t>(p.bits, bp0, l0, last0, stack + top);
            wave_sync();
            if (W0 != 0) {
                for (uint32_t t = top; t < n0; ++t) {
                    const uint64_t rb = tri64(stack[t]);
                    for (uint32_t u = lane; u < t; u += WAVE) atomicAdd(&p.M[rb + stack[u]], W0);
                }
            }
            wave_sync();
            base += 1;
            continue;
        }
        if (lane < cnt) decode_node<uint32_t>(p.bits, bp, l, m.z, dec + off);
        wave_sync();
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t nj = bcast(m.x, j), lj = bcast(m.y, j), oj = bcast(off, j), Wj = bcast(W, j);
            const uint32_t top = nj - lj;
            for (uint32_t k = lane; k < lj; k += WAVE) stack[top + k] = dec[oj + k];
            wave_sync();
            if (Wj != 0) {
                for (uint32_t t = top; t < nj; ++t) {
                    const uint64_t rb = tri64(stack[t]);
                    for (uint32_t u = lane; u < t; u += WAVE) atomicAdd(&p.M[rb + stack[u]], Wj);
                }
            }
            wave_sync();
        }
        base += cnt;
    }
}

// ------------------------------------------------------------------------------------------
// direct kernel: stack + decode buffer in LDS (small per-wave footprint -> high occupancy),
// every update is an HBM/L2 atomic
// ------------------------------------------------------------------------------------------
constexpr int DIRECT_WAVES = 8;
template <int NCAP>
__global__ __launch_bounds__(WAVE * DIRECT_WAVES) void a2a_direct_kernel(A2AParams p) {
    __shared__ uint16_t dec_all[DIRECT_WAVES][DEC_CAP];
    __shared__ uint16_t stack_all[DIRECT_WAVES][NCAP];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg = p.seg_begin + blockIdx.x * DIRECT_WAVES + wave;
    if (seg >= p.seg_end) return;
    uint16_t* dec = dec_all[wave];
    uint16_t* stack = stack_all[wave];
    const Segment sg = p.segs[seg];
    const uint32_t first = __builtin_amdgcn_readfirstlane(sg.first);
    const uint32_t end = __builtin_amdgcn_readfirstlane(sg.end);
    if (first >= end) return;
    init_stack<uint16_t>(p, first, stack, lane);
    for (uint32_t base = first; base < end;) {
        const uint32_t i = base + lane;
        const bool valid = i < end;
        uint4 m = valid ? p.meta[i] : make_uint4(0, 0, 0, 0);
        const uint64_t bp = valid ? p.bitpos[i] : 0;
        const uint32_t W = valid ? (p.wprefix[p.sub_end[i]] - p.wprefix[i]) : 0u;
        const uint32_t l = m.y;
        const uint32_t incl = wave_incl_scan(l, lane);
        const unsigned long long fit = __ballot(valid && incl <= (uint32_t)DEC_CAP);
        uint32_t cnt = fit == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fit);
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        const uint32_t off = incl - l;
        uint32_t nproc = cnt;
        if (cnt == 0) {
            const uint32_t n0 = bcast(m.x, 0), l0 = bcast(m.y, 0), last0 = bcast(m.z, 0);
            const uint64_t bp0 = ((uint64_t)bcast((uint32_t)(bp >> 32), 0) << 32) | bcast((uint32_t)bp, 0);
            if (lane == 0) decode_node<uint16_t>(p.bits, bp0, l0, last0, stack + (n0 - l0));
            nproc = 1;
        } else if (lane < cnt) {
            decode_node<uint16_t>(p.bits, bp, l, m.z, dec + off);
        }
        wave_sync();
        for (uint32_t j = 0; j < nproc; ++j) {
            const uint32_t nj = bcast(m.x, j), lj = bcast(m.y, j), oj = bcast(off, j), Wj = bcast(W, j);
            const uint32_t top = nj - lj;
            if (cnt != 0)
                for (uint32_t k = lane; k < lj; k += WAVE) stack[top + k] = dec[oj + k];
            wave_sync();
            if (Wj != 0 && nj > 1 && !(p.dbg & 2)) {
                for (uint32_t t = top; t < nj; ++t) {
                    const uint64_t rb = tri64(stack[t]);
                    for (uint32_t u = lane; u < t; u += WAVE) atomicAdd(&p.M[rb + stack[u]], Wj);
                }
            }
            wave_sync();
        }
        base += nproc;
    }
}

// ------------------------------------------------------------------------------------------
// tile kernel: everything wave-private in LDS
// ------------------------------------------------------------------------------------------
template <int S, int NCAP>
struct WaveLds {
    uint32_t tile[S * (S - 1) / 2];
    uint32_t dec[DEC_CAP];
    uint16_t rstack[NCAP];          // real sample ids, root -> current node
    uint8_t cstack[NCAP];           // compact indices of the same entries (valid below `cvalid`)
    uint8_t map[NCAP];              // sample id -> compact index, 0xFF = none
    uint16_t rid[S];                // compact index -> sample id
};

template <int S, int NCAP>
__device__ __forceinline__ void tile_flush(WaveLds<S, NCAP>& L, uint32_t& ns, uint32_t* __restrict__ M, uint32_t lane,
                                           unsigned long long* counters) {
    for (uint32_t i = 1; i < ns; ++i) {
        const uint32_t x = L.rid[i];
        const uint32_t rowoff = i * (i - 1) / 2;
        for (uint32_t j = lane; j < i; j += WAVE) {
            const uint32_t v = L.tile[rowoff + j];
            if (v) {
                const uint32_t y = L.rid[j];
                const uint32_t hi = x > y ? x : y, lo = x > y ? y : x;
                atomicAdd(&M[tri64(hi) + lo], v);
                L.tile[rowoff + j] = 0;
            }
        }
    }
    for (uint32_t k = lane; k < ns; k += WAVE) L.map[L.rid[k]] = 0xFF;
    if (lane == 0 && ns) atomicAdd(&counters[0], 1ull);
    ns = 0;
    wave_sync();
}

template <int S, int NCAP>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK) void a2a_tile_kernel(A2AParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    using LDS = WaveLds<S, NCAP>;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg = p.seg_begin + blockIdx.x * WAVES_PER_BLOCK + wave;
    if (seg >= p.seg_end) return;
    LDS& L = reinterpret_cast<LDS*>(lds_raw)[wave];
    for (uint32_t k = lane; k < (uint32_t)(S * (S - 1) / 2); k += WAVE) L.tile[k] = 0;
    for (uint32_t k = lane; k < (uint32_t)NCAP; k += WAVE) L.map[k] = 0xFF;
    wave_sync();

    const Segment sg = p.segs[seg];
    const uint32_t first = __builtin_amdgcn_readfirstlane(sg.first);
    const uint32_t end = __builtin_amdgcn_readfirstlane(sg.end);
    if (first >= end) return;
    init_stack<uint16_t>(p, first, L.rstack, lane);
    uint32_t ns = 0;        // compact indices in use
    uint32_t cvalid = 0;    // cstack[0..cvalid) is valid for the current epoch
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (uint32_t base = first; base < end;) {
        const uint32_t i = base + lane;
        const bool valid = i < end;
        uint4 m = valid ? p.meta[i] : make_uint4(0, 0, 0, 0);
        const uint64_t bp = valid ? p.bitpos[i] : 0;
        const uint32_t W = valid ? (p.wprefix[p.sub_end[i]] - p.wprefix[i]) : 0u;
        const uint32_t l = m.y;
        const uint32_t incl = wave_incl_scan(l, lane);
        const unsigned long long fit = __ballot(valid && incl <= (uint32_t)DEC_CAP);
        uint32_t cnt = fit == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fit);
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        const uint32_t off = incl - l;
        uint32_t nproc = cnt;
        if (cnt == 0) {
            // one node with > DEC_CAP local ids (possible only when N > DEC_CAP): decode into rstack
            const uint32_t n0 = bcast(m.x, 0), l0 = bcast(m.y, 0), last0 = bcast(m.z, 0);
            const uint64_t bp0 = ((uint64_t)bcast((uint32_t)(bp >> 32), 0) << 32) | bcast((uint32_t)bp, 0);
            if (lane == 0) decode_node<uint16_t>(p.bits, bp0, l0, last0, L.rstack + (n0 - l0));
            nproc = 1;
        } else if (lane < cnt) {
            decode_node<uint32_t>(p.bits, bp, l, m.z, L.dec + off);
        }
        wave_sync();
        for (uint32_t j = 0; j < nproc; ++j) {
            const uint32_t nj = bcast(m.x, j), lj = bcast(m.y, j), oj = bcast(off, j), Wj = bcast(W, j);
            const uint32_t top = nj - lj;
            if (cnt != 0)
                for (uint32_t k = lane; k < lj; k += WAVE) L.rstack[top + k] = (uint16_t)L.dec[oj + k];
            cvalid = cvalid < top ? cvalid : top;
            wave_sync();
            if (Wj == 0 || nj < 2) continue;
            if (p.dbg & 8) continue;
            if (nj > (uint32_t)S) {
                if (p.dbg & 16) continue;
                // list longer than the tile side: straight to HBM
                for (uint32_t t = top; t < nj; ++t) {
                    const uint64_t rb = tri64(L.rstack[t]);
                    for (uint32_t u = lane; u < t; u += WAVE) atomicAdd(&p.M[rb + L.rstack[u]], Wj);
                }
                wave_sync();
                continue;
            }
            // --- make sure stack entries [cvalid, nj) have compact indices ------------------
            uint32_t need = 0;
            for (uint32_t c0 = cvalid; c0 < nj; c0 += WAVE) {
                const uint32_t pos = c0 + lane;
                const bool isnew = pos < nj && L.map[L.rstack[pos]] == 0xFF;
                need += (uint32_t)__popcll(__ballot(isnew));
            }
            if (ns + need > (uint32_t)S) {
                if (p.dbg & 4) { for (uint32_t k = lane; k < ns; k += WAVE) L.map[L.rid[k]] = 0xFF; ns = 0; wave_sync(); } else
                tile_flush<S, NCAP>(L, ns, p.M, lane, p.counters);
                cvalid = 0;
            }
            for (uint32_t c0 = cvalid; c0 < nj; c0 += WAVE) {
                const uint32_t pos = c0 + lane;
                const bool act = pos < nj;
                const uint32_t id = act ? L.rstack[pos] : 0;
                uint32_t ci = act ? L.map[id] : 0;
                const bool isnew = act && ci == 0xFF;
                const unsigned long long bal = __ballot(isnew);
                if (isnew) {
                    ci = ns + (uint32_t)__popcll(bal & lt_mask);
                    L.map[id] = (uint8_t)ci;
                    L.rid[ci] = (uint16_t)id;
                }
                if (act) L.cstack[pos] = (uint8_t)ci;
                ns += (uint32_t)__popcll(bal);
            }
            cvalid = nj;
            wave_sync();
            // --- scatter-add: rows = local ids, columns = everything before them -------------
            // nj <= S <= 128: the whole compact list lives in two registers per lane
            const uint32_t c0v = lane < nj ? L.cstack[lane] : 0u;
            const uint32_t c1v = (S > 64 && lane + 64 < nj) ? L.cstack[lane + 64] : 0u;
            const uint32_t t0v = c0v * (c0v - 1) / 2;       // garbage for c=0 is never used as a base with c<r false... guarded below
            const uint32_t t1v = c1v * (c1v - 1) / 2;
            if (!(p.dbg & 2))
            for (uint32_t t = top; t < nj; ++t) {
                const uint32_t r = t < 64 ? bcast(c0v, t) : bcast(c1v, t - 64);
                const uint32_t rr = r * (r - 1) / 2;
                if (lane < t) {
                    const uint32_t idx = c0v < r ? rr + c0v : t0v + r;
                    atomicAdd(&L.tile[idx], Wj);
                }
                if (S > 64 && lane + 64 < t) {
                    const uint32_t idx = c1v < r ? rr + c1v : t1v + r;
                    atomicAdd(&L.tile[idx], Wj);
                }
            }
            wave_sync();
        }
        wave_sync();
        base += nproc;
    }
    tile_flush<S, NCAP>(L, ns, p.M, lane, p.counters);
}

// ------------------------------------------------------------------------------------------
// v2 pipeline: block records + wavefront ballot / popcount accumulation
//
// The N x N matrix is cut into 64 x 64 blocks (X, Y), X >= Y, by sample-id range.  Because a
// pattern's id list is ascending, the ids that fall into one block are a contiguous run of
// stack positions, so the whole update of a node
//        for every local id a (row), every earlier id b (column):  M[a][b] += W
// factors into a few BLOCK RECORDS  (X, Y, rowmask, colmask, W):
//        M[64X + r][64Y + c] += W   for r in rowmask, c in colmask (and c < r when X == Y).
// K1 (b2_emit_kernel) walks the DFS stream exactly like the v1 kernels (lane-per-node gamma
//   decode, one id stack per wave) but keeps, per stack position, the running bit mask of the
//   ids of the same block ("cum"), and writes records instead of touching the matrix.
//   Records go straight to their final, bucket-grouped position: the per-(segment, bucket)
//   record counts are a pure function of the database and are tabulated once at upload
//   (count mode of the same kernel), so no atomics and no sort are needed at run time.
// K2 (b2_apply_kernel) gives one workgroup a chunk of one bucket and a 64 x 64 uint32
//   accumulator in LDS.  Records with W == 1 and many rows (the bulk: unique k-mer patterns)
//   are reduced 64 at a time with ballots: R_r = ballot(row r in record j), C^T by a 64 x 64
//   bit transpose across lanes, cell(r, c) += popcount(R_r & C^T_c) — one LDS add per cell per
//   64 records instead of one per record.  The other records are applied row by row with the
//   column mask as the lane mask.  The accumulator is written back with one HBM atomic per
//   non-zero cell.
// ------------------------------------------------------------------------------------------
// block records, struct-of-arrays, one slot per record: rows always; cols only for off-diagonal
// buckets (on the diagonal cols == rows); w only for the heavy class (class 0 records have w == 1)
struct B2Recs { unsigned long long* rows; ulonglong2* rc; uint32_t* w; };   // rows: diagonal buckets; rc = {rows, cols}: others
struct B2Item { uint32_t X, Y, cls, begin, end; };

constexpr int B2_WAVES = 4;

struct B2Params {
    A2AParams a;
    uint32_t maxn_pad;            // stack capacity (multiple of 64)
    uint32_t dec_cap;             // decoded ids per batch (>= maxn_pad)
    uint32_t nctr;                // 2 * number of buckets
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
            const uint32_t cls = Wj == 1u ? 0u : 1u;
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
                const uint32_t b = (bX * (bX + 1) / 2 + bY) * 2 + cls;
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
constexpr int B3_CHAIN = 192;      // max root-path length (in nodes) the chain table can hold (slot ids are bytes)

template <bool COUNT, bool LONG>
__global__ __launch_bounds__(256) void b3_decode_kernel(const uint4* __restrict__ meta, const uint64_t* __restrict__ bitpos,
                                                        const uint64_t* __restrict__ bits, const uint32_t* __restrict__ perm,
                                                        uint32_t P, uint32_t short_max, BlockMap bm, unsigned long long* __restrict__ p0_mask, uint16_t* __restrict__ p0_info,
                                                        uint32_t* __restrict__ pair_ofs, uint8_t* __restrict__ pair_blk,
                                                        unsigned long long* __restrict__ pair_mask) {
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
        // DFS-order launch: the 256 nodes of the block are re-dealt to its threads by decreasing list length
        // (counting sort in LDS), so every wave decodes streams of similar length while all global accesses
        // of the block stay inside its own 256-node window
        __shared__ uint32_t bins[64];
        __shared__ uint16_t order[256];
        if (threadIdx.x < 64) bins[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t lt = t < P ? meta[t].y : 0u;
        const uint32_t key = (t < P && lt <= short_max) ? lt : 0u;      // 0: nothing to decode here
        atomicAdd(&bins[short_max - key], 1u);                           // bin 0 = longest
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (uint32_t b = 0; b <= short_max; ++b) { const uint32_t c = bins[b]; bins[b] = run; run += c; }
        }
        __syncthreads();
        order[atomicAdd(&bins[short_max - key], 1u)] = (uint16_t)threadIdx.x;
        __syncthreads();
        i = blockIdx.x * blockDim.x + order[threadIdx.x];
        if (i >= P) return;
    }
    const uint4 m = meta[i];
    const uint32_t l = m.y;
    if (!perm && l > short_max) return;
    uint32_t out = 0;
    uint32_t npairs = 0, blk0 = 0;
    unsigned long long mask0 = 0;
    if (l && !LONG) {
        // short lists: ONE decode pass.  The deltas are parked in LDS ([k][thread], conflict-free) and the ids
        // are rebuilt from the explicit last id downwards (pattern_t::decodeSamples does the same subtraction,
        // reference src/pattern.cpp:104-107); blocks therefore come out in DESCENDING order: the lowest block is the
        // node's inline first pair, the others fill its CSR range from the top.
        __shared__ uint16_t dl[32 * 256];
        if (l > 1) {
            BitCursor c1(bits, bitpos[i]);
            for (uint32_t k = 0; k + 1 < l; ++k) dl[k * 256 + threadIdx.x] = (uint16_t)c1.next();
        }
        uint32_t id = m.z;
        uint32_t curblk = bm.blk(id);
        unsigned long long acc = 0;
        uint32_t top = 0;
        if (!COUNT) top = pair_ofs[i + 1];                 // one past this node's extra pairs
        for (uint32_t k = l; k-- > 0;) {
            const uint32_t blk = bm.blk(id);
            if (blk != curblk) {
                if (!COUNT) { --top; pair_blk[top] = (uint8_t)curblk; pair_mask[top] = acc; }
                ++npairs;
                curblk = blk; acc = 0;
            }
            acc |= 1ull << bm.bit(id, blk);
            if (k) id -= dl[(k - 1) * 256 + threadIdx.x];
        }
        blk0 = curblk; mask0 = acc;
        ++npairs;
    } else if (l) {
        uint32_t id = m.z;
        using Cursor = BitCursorDeep<6>;
        if (l > 1) {
            Cursor c1(bits, bitpos[i]);
            uint32_t sum = 0;
            for (uint32_t k = 0; k + 1 < l; ++k) sum += c1.next();
            id = m.z - sum;
        }
        Cursor c2(bits, bitpos[i]);
        uint32_t curblk = bm.blk(id);
        unsigned long long acc = 0;
        for (uint32_t k = 0; k < l; ++k) {
            const uint32_t blk = bm.blk(id);
            if (blk != curblk) {
                if (npairs == 0) { blk0 = curblk; mask0 = acc; if (!COUNT) out = pair_ofs[i]; }
                else { if (!COUNT) { pair_blk[out] = (uint8_t)curblk; pair_mask[out] = acc; } ++out; }
                ++npairs;
                curblk = blk; acc = 0;
            }
            acc |= 1ull << bm.bit(id, blk);
            if (k + 1 < l) id += c2.next();
        }
        if (npairs == 0) { blk0 = curblk; mask0 = acc; }
        else if (!COUNT) { pair_blk[out] = (uint8_t)curblk; pair_mask[out] = acc; }
        ++npairs;
    }
    if (COUNT) pair_ofs[i] = npairs ? npairs - 1 : 0;
    else { p0_mask[i] = mask0; p0_info[i] = (uint16_t)(blk0 | (npairs << 8)); }
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
};

__host__ __device__ inline size_t b3_wave_bytes(uint32_t nbw, uint32_t maxn_pad, uint32_t nctr, uint32_t chain_cap) {
    size_t b = (size_t)chain_cap * nbw * 8 + (size_t)nctr * 4 + (size_t)chain_cap * 4 * 2 + (maxn_pad + 64);
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
    uint32_t lane;
    unsigned long long lt_mask;
};

struct B3Lane {                    // one node per lane
    bool valid;
    uint32_t n, l, w, info, idx;
    int32_t par;
    unsigned long long m0;
};

// One batch of 64 consecutive DFS nodes.  Every lane keeps the full-list masks of its node in W 64-bit
// registers.  W == NBW with IDENT: register s is word s.  Otherwise the batch only touches ku <= W distinct
// words (the usual case: one or two clades) and register s holds word wl[s] — the work then does not grow
// with the number of blocks of the matrix.
template <int W, bool IDENT, int NBW, bool EMIT>
__device__ __forceinline__ void b3_batch(const B3Params& q, const B3Ctx& C, const B3Lane& L, uint32_t base, uint32_t end,
                                         uint32_t U, const uint32_t (&wl)[W], uint32_t ku, uint32_t rootslot, uint32_t inh) {
    const uint32_t lane = C.lane;
    auto slot_of_word = [&](uint32_t wd) -> uint32_t { return IDENT ? wd : (uint32_t)__popc(U & ((1u << wd) - 1u)); };
    unsigned long long F[W];
    const uint32_t b0 = L.info & 0xFFu, np = L.info >> 8;
    {
        const uint32_t s0 = slot_of_word(b0);
#pragma unroll
        for (int s = 0; s < W; ++s) F[s] = (np != 0 && s0 == (uint32_t)s) ? L.m0 : 0ull;
        if (__ballot(np > 1)) {
            const uint32_t po = np > 1 ? q.pair_ofs[L.idx] : 0u;
            uint32_t mx = np > 1 ? np - 1 : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, d, WAVE); mx = o > mx ? o : mx; }
            mx = __builtin_amdgcn_readfirstlane(mx);
            for (uint32_t t = 0; t < mx; ++t) {
                if (t + 1 < np) {
                    const uint32_t sb = slot_of_word(q.pair_blk[po + t]);
                    const unsigned long long mk = q.pair_mask[po + t];
#pragma unroll
                    for (int s = 0; s < W; ++s) F[s] |= (sb == (uint32_t)s) ? mk : 0ull;
                }
            }
        }
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
    // ---- records: flat form, one per pair of non-empty words X >= Y of patterns with w > 0
    const bool act = L.valid && L.w != 0 && L.n >= 2 && !(q.dbg & 512u);
    const bool heavy = L.w != 1u;
    uint32_t mywl = 0;               // word of register (lane >> 1): lane 2*sY + c owns the counter of combo (X, Y, c)
    if (IDENT) mywl = lane >> 1;
    else {
#pragma unroll
        for (int s = 0; s < W; ++s) if ((lane >> 1) == (uint32_t)s) mywl = wl[s];
    }
    uint32_t Ua = act ? 1u : 0u;     // does any lane emit at all
    if (!__ballot(Ua != 0)) goto chain_update;
#pragma unroll
    for (int sX = 0; sX < W; ++sX) {
        if (!IDENT && (uint32_t)sX >= ku) break;
        const uint32_t X = IDENT ? (uint32_t)sX : wl[sX];
        const bool ax = act && F[sX] != 0;
        if (!__ballot(ax)) continue;
        uint32_t mycnt = 0;
#pragma unroll
        for (int sY = 0; sY <= sX; ++sY) {
            const bool a = ax && F[sY] != 0;
            const uint32_t c0 = (uint32_t)__popcll(__ballot(a && !heavy));
            const uint32_t c1 = (uint32_t)__popcll(__ballot(a && heavy));
            if (lane == (uint32_t)(2 * sY)) mycnt = c0;
            if (lane == (uint32_t)(2 * sY + 1)) mycnt = c1;
        }
        uint32_t mybase = 0;
        if (mycnt) mybase = atomicAdd(&C.ctr[(X * (X + 1) / 2 + mywl) * 2 + (lane & 1u)], mycnt);
        if (EMIT && !(q.dbg & 256u)) {
#pragma unroll
            for (int sY = 0; sY <= sX; ++sY) {
                const bool a = ax && F[sY] != 0;
                const unsigned long long bal = __ballot(a && !heavy), bah = __ballot(a && heavy);
                if (!(bal | bah)) continue;
                const uint32_t base0 = bcast(mybase, 2 * sY), base1 = bcast(mybase, 2 * sY + 1);
                if (a) {
                    const uint32_t slot = heavy ? base1 + (uint32_t)__popcll(bah & C.lt_mask) : base0 + (uint32_t)__popcll(bal & C.lt_mask);
                    if (sX != sY) q.rec.rc[slot] = make_ulonglong2(F[sX], F[sY]); else q.rec.rows[slot] = F[sX];
                    if (heavy) q.rec.w[slot] = L.w;
                }
            }
        }
    }
chain_update:
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
    }
}

constexpr int B3_K = 8;            // word registers of the compact path

template <int NBW, bool EMIT>
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
    C.lane = lane;
    C.lt_mask = (1ull << lane) - 1ull;
    uint32_t* my_table = q.table + (size_t)seg * q.nctr;
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
    uint32_t nx_nl = 0, nx_w = 0, nx_info = 0;
    int32_t nx_par = -1;
    unsigned long long nx_m0 = 0;
    auto fetch = [&](uint32_t b0) {
        const uint32_t ii = b0 + lane;
        const bool v = ii < end;
        nx_nl = v ? q.nl[ii] : 0u;
        nx_w = v ? q.w[ii] : 0u;
        nx_par = v ? q.parent[ii] : -1;
        nx_info = v ? q.p0_info[ii] : 0u;
        nx_m0 = v ? q.p0_mask[ii] : 0ull;
    };
    fetch(first);
    const bool prof = (q.dbg & 32u) != 0;
    unsigned long long t_all = 0, t0 = 0;
    unsigned long long n_compact = 0, n_mid = 0, n_full = 0;
    for (uint32_t base = first; base < end; base += WAVE) {
        if (prof) t0 = __builtin_amdgcn_s_memtime();
        B3Lane L;
        L.idx = base + lane;
        L.valid = L.idx < end;
        L.n = nx_nl & 0xFFFFu; L.l = nx_nl >> 16;
        L.w = nx_w; L.par = nx_par; L.info = nx_info; L.m0 = nx_m0;
        if (base + WAVE < end) fetch(base + WAVE);
        // words this batch touches: own local words + the words inherited from a parent before the batch
        const uint32_t np = L.info >> 8;
        uint32_t lw = np ? (1u << (L.info & 0xFFu)) : 0u;
        if (__ballot(np > 1)) {
            const uint32_t po = np > 1 ? q.pair_ofs[L.idx] : 0u;
            for (uint32_t t = 0; t + 1 < np; ++t) lw |= 1u << q.pair_blk[po + t];
        }
        uint32_t rootslot = 0xFFFFFFFFu, inh = 0;
        if (L.valid && L.par >= 0 && L.par < (int32_t)base) {
            rootslot = C.slot_of_n[L.n - L.l];
            inh = C.chain_nz[rootslot];
        }
        uint32_t U = L.valid ? (lw | inh) : 0u;
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
            b3_batch<B3_K, false, NBW, EMIT>(q, C, L, base, end, U, wl, ku, rootslot, inh);
            ++n_mid;
        } else {
            uint32_t wl[NBW];
#pragma unroll
            for (int s = 0; s < NBW; ++s) wl[s] = (uint32_t)s;
            b3_batch<NBW, true, NBW, EMIT>(q, C, L, base, end, U, wl, (uint32_t)NBW, rootslot, inh);
            ++n_full;
        }
        if (prof) t_all += __builtin_amdgcn_s_memtime() - t0;
    }
    if (prof && lane == 0) {
        atomicAdd(&q.counters[1], t_all); atomicAdd(&q.counters[2], n_compact);
        atomicAdd(&q.counters[3], n_mid); atomicAdd(&q.counters[4], n_full);
    }
    if (!EMIT) {
        lds_sync();
        for (uint32_t k = lane; k < q.nctr; k += WAVE) my_table[k] = C.ctr[k];
    }
}

// 64 x 64 bit-matrix transpose across the lanes of a wave: lane i holds row i on entry, column i on exit
__device__ __forceinline__ unsigned long long transpose64(unsigned long long x, uint32_t lane) {
    const unsigned long long masks[6] = {0x00000000FFFFFFFFull, 0x0000FFFF0000FFFFull, 0x00FF00FF00FF00FFull,
                                         0x0F0F0F0F0F0F0F0Full, 0x3333333333333333ull, 0x5555555555555555ull};
    int s = 32;
#pragma unroll
    for (int k = 0; k < 6; ++k, s >>= 1) {
        const unsigned long long m = masks[k];
        const uint32_t plo = (uint32_t)__shfl_xor((int)(uint32_t)x, s, WAVE);
        const uint32_t phi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), s, WAVE);
        const unsigned long long pv = ((unsigned long long)phi << 32) | plo;
        x = (lane & (uint32_t)s) ? ((x & ~m) | ((pv >> s) & m)) : ((x & m) | ((pv & m) << s));
    }
    return x;
}

__global__ __launch_bounds__(256) void b2_apply_kernel(const B2Recs rec, const B2Item* __restrict__ items,
                                                       uint32_t* __restrict__ M, uint32_t N, uint32_t dbg, uint32_t bwidth) {
    __shared__ uint32_t acc[64 * 64];
    __shared__ unsigned long long rtbuf[4][64];
    const B2Item it = items[blockIdx.x];
    if ((dbg & 64u) && it.cls == 0) return;
    if ((dbg & 128u) && it.cls == 1) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) acc[k] = 0;
    __syncthreads();
    const bool diag = it.X == it.Y;
    // the next group's records are fetched while the current group is reduced
    unsigned long long nR = 0, nC = 0;
    uint32_t nW = 0;
    auto fetch = [&](uint32_t g) {
        const uint32_t j = g + lane;
        nR = 0; nC = 0; nW = 0;
        if (j < it.end) {
            if (diag) { nR = rec.rows[j]; nC = nR; }
            else { const ulonglong2 rc = rec.rc[j]; nR = rc.x; nC = rc.y; }
            nW = it.cls ? rec.w[j] : 1u;
        }
    };
    fetch(it.begin + wave * 64);
    for (uint32_t g0 = it.begin + wave * 64; g0 < it.end; g0 += 256) {
        const unsigned long long R = nR, C = nC;
        const uint32_t W = nW;
        if (g0 + 256 < it.end) fetch(g0 + 256);
        // lane c: bit j of Ct = record j contains column c
        const unsigned long long Ct = transpose64(C, lane);
        if (dbg & 2048u) { if ((uint32_t)Ct + (uint32_t)R + W == 0x12345u) acc[lane] = 1; continue; }
        if (it.cls == 0) {
            // every record has weight 1: cell(r, c) += number of records that contain row r and column c
            // R^T goes through LDS: row r's record mask is then a broadcast read instead of a ballot
            // (a ballot writes an SGPR pair that the next VALU must wait for; 64 of them serialise the loop)
            rtbuf[wave][lane] = transpose64(R, lane);
            lds_sync();
            if (!diag) {
#pragma unroll 8
                for (uint32_t r = 0; r < bwidth; ++r) {             // rows >= the block width never occur
                    const uint32_t c = (uint32_t)__popcll(Ct & rtbuf[wave][r]);
                    if (c) atomicAdd(&acc[r * 64 + lane], c);
                }
            } else {
#pragma unroll 8
                for (uint32_t r = 0; r < bwidth; ++r) {
                    const uint32_t c = lane < r ? (uint32_t)__popcll(Ct & rtbuf[wave][r]) : 0u;
                    if (c) atomicAdd(&acc[r * 64 + lane], c);
                }
            }
            lds_sync();
        } else {
            // general weights: the same count per bit plane of w, scaled by 2^plane.  Planes 0..3 are
            // kept in scalar registers (weights are usually small); higher planes are rare.
            uint32_t wor = W;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) wor |= (uint32_t)__shfl_xor((int)wor, d, WAVE);
            wor = __builtin_amdgcn_readfirstlane(wor);
            const unsigned long long W0 = __ballot((W & 1u) != 0), W1 = __ballot((W & 2u) != 0);
            const unsigned long long W2 = __ballot((W & 4u) != 0), W3 = __ballot((W & 8u) != 0);
            const uint32_t whigh = wor >> 4;
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                const unsigned long long Rr = __ballot(((R >> r) & 1ull) != 0);
                if (!Rr) continue;
                const unsigned long long base = Ct & Rr;
                uint32_t c = (uint32_t)__popcll(base & W0) + ((uint32_t)__popcll(base & W1) << 1) +
                             ((uint32_t)__popcll(base & W2) << 2) + ((uint32_t)__popcll(base & W3) << 3);
                for (uint32_t wb = whigh; wb; wb &= wb - 1) {
                    const uint32_t b = 4u + (uint32_t)__builtin_ctz(wb);
                    c += (uint32_t)__popcll(base & __ballot(((W >> b) & 1u) != 0)) << b;
                }
                if (c && !(diag && lane >= (uint32_t)r)) atomicAdd(&acc[r * 64 + lane], c);
            }
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 256) {
        const uint32_t v = acc[k];
        if (!v) continue;
        const uint64_t row = (uint64_t)it.X * bwidth + (k >> 6), col = (uint64_t)it.Y * bwidth + (k & 63u);
        if (!(dbg & 1024u)) atomicAdd(&M[tri64(row) + col], v);
    }
}

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

struct BitWriter {
    std::vector<uint64_t>& words;
    uint64_t pos = 0;
    explicit BitWriter(std::vector<uint64_t>& w) : words(w) {}
    // append the first `nbits` bits (MSB-first) of src
    void append(const uint64_t* src, uint32_t nbits) {
        uint64_t need = (pos + nbits + 63) / 64 + 1;
        if (words.size() < need) words.resize(std::max<uint64_t>(need, words.size() * 2), 0);
        uint32_t done = 0;
        while (done < nbits) {
            uint32_t take = std::min<uint32_t>(64, nbits - done);
            uint64_t chunk = src[done >> 6];                     // done is always a multiple of 64 here
            if (take < 64) chunk &= ~0ull << (64 - take);
            uint32_t s = (uint32_t)(pos & 63);
            words[pos >> 6] |= chunk >> s;
            if (s && take > 64 - s) words[(pos >> 6) + 1] |= chunk << (64 - s);
            pos += take;
            done += take;
        }
    }
};

template <class T>
int dev_upload(T** dst, const T* src, size_t n) {
    size_t bytes = std::max<size_t>(1, n) * sizeof(T);
    HIP_TRY(hipMalloc((void**)dst, bytes));
    if (n) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace


// ------------------------------------------------------------------------------------------
// v2 pipeline: host side
// ------------------------------------------------------------------------------------------
namespace {

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

template <int NBW, bool EMIT>
int b3_launch_emit_t(kmdb_db* db, hipStream_t st, uint32_t dbg) {
    B3Params q{};
    q.nl = db->b3_nl; q.parent = db->parent; q.w = db->w; q.segs = db->rsegs;
    q.seg_anc = db->b3_seg_anc; q.seg_anc_n = db->b3_seg_anc_n;
    q.p0_mask = db->b3_p0_mask; q.p0_info = db->b3_p0_info;
    q.pair_ofs = db->b3_pair_ofs; q.pair_blk = db->b3_pair_blk; q.pair_mask = db->b3_pair_mask;
    q.n_segs = db->n_rsegs; q.maxn_pad = db->b2_maxn_pad; q.nctr = db->b2_nctr; q.chain_cap = db->b3_chain_cap;
    q.table = db->b2_table; q.rec = B2Recs{db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w}; q.dbg = dbg; q.counters = db->counters;
    const size_t lds = b3_wave_bytes(NBW, q.maxn_pad, q.nctr, q.chain_cap) * B3_WAVES;
    HIP_TRY(hipFuncSetAttribute((const void*)b3_emit_kernel<NBW, EMIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (EMIT && getenv("KMDB_VERBOSE")) {
        static bool once = false;
        if (!once) {
            once = true;
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)b3_emit_kernel<NBW, EMIT>, WAVE * B3_WAVES, lds);
            fprintf(stderr, "[kmdb] emit kernel: NBW %d, width %u, chain_cap %u, LDS/block %zu B, resident blocks/CU %d, segments %u\n",
                    NBW, db->b2_width, q.chain_cap, lds, nb, db->n_rsegs);
        }
    }
    const uint32_t blocks = (db->n_rsegs + B3_WAVES - 1) / B3_WAVES;
    if (blocks) hipLaunchKernelGGL((b3_emit_kernel<NBW, EMIT>), dim3(blocks), dim3(WAVE * B3_WAVES), lds, st, q);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool EMIT>
int b3_launch_emit(kmdb_db* db, hipStream_t st, uint32_t dbg = 0) {
    if (db->b3_nbw <= 8) return b3_launch_emit_t<8, EMIT>(db, st, dbg);
    if (db->b3_nbw <= 16) return b3_launch_emit_t<16, EMIT>(db, st, dbg);
    if (db->b3_nbw <= 20) return b3_launch_emit_t<20, EMIT>(db, st, dbg);
    if (db->b3_nbw <= 24) return b3_launch_emit_t<24, EMIT>(db, st, dbg);
    return b3_launch_emit_t<32, EMIT>(db, st, dbg);
}

template <bool COUNT>
int b3_launch_decode(kmdb_db* db, hipStream_t st) {
    const uint32_t P = (uint32_t)db->P;
    const BlockMap bm{db->b2_width, (uint32_t)((1ull << 32) / db->b2_width) + 1u};
    if (P && !getenv("KMDB_SKIP_K0A"))
        hipLaunchKernelGGL((b3_decode_kernel<COUNT, false>), dim3((P + 255) / 256), dim3(256), 0, st, db->meta, db->bitpos, db->bits,
                           (const uint32_t*)nullptr, P, db->b3_short_max, bm, db->b3_p0_mask, db->b3_p0_info, db->b3_pair_ofs,
                           db->b3_pair_blk, db->b3_pair_mask);
    if (db->b3_n_long && !getenv("KMDB_SKIP_K0B"))
        hipLaunchKernelGGL((b3_decode_kernel<COUNT, true>), dim3((db->b3_n_long + 255) / 256), dim3(256), 0, st, db->meta, db->bitpos,
                           db->bits, (const uint32_t*)db->b3_perm, db->b3_n_long, db->b3_short_max, bm, db->b3_p0_mask, db->b3_p0_info,
                           db->b3_pair_ofs, db->b3_pair_blk, db->b3_pair_mask);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Decide whether the database qualifies for the block-record pipeline and, if so, tabulate the
// per-(segment, bucket) record counts with the count mode of the emit kernel (layout metadata:
// a pure function of the database, like CSR row pointers), turn them into record bases and cut
// the buckets into work items for the apply kernel.
void b2_release_width(kmdb_db* db) {
    void* ptrs[] = {db->b2_table, db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w, db->b2_items, db->b3_pair_ofs,
                    db->b3_pair_blk, db->b3_pair_mask, db->b3_p0_mask, db->b3_p0_info};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    db->b2_table = nullptr; db->b2_rec_rows = nullptr; db->b2_rec_rc = nullptr; db->b2_rec_w = nullptr; db->b2_items = nullptr;
    db->b3_pair_ofs = nullptr; db->b3_pair_blk = nullptr; db->b3_pair_mask = nullptr; db->b3_p0_mask = nullptr;
    db->b3_p0_info = nullptr;
    db->b2_ready = db->b3_ready = false;
}

// Everything of the block-record pipeline that depends on the block width: run the count modes of the
// kernels (layout metadata: a pure function of the database, like CSR row pointers), turn the
// per-(segment, bucket) record counts into record bases and cut the buckets into work items for the
// apply kernel.  *fits is false when the width cannot be used.
int b2_prepare_width(kmdb_db* db, uint32_t width, uint32_t max_n, bool chain_ok, bool* fits) {
    *fits = false;
    const uint64_t N = db->N, P = db->P;
    const uint32_t NB = (uint32_t)((N + width - 1) / width);
    if (NB > 32) return 0;
    db->b2_width = width;
    db->b2_maxn_pad = std::max<uint32_t>(64, (max_n + 63) / 64 * 64);
    db->b2_dec_cap = std::max<uint32_t>(512, db->b2_maxn_pad);
    db->b2_nctr = NB * (NB + 1) / 2 * 2;
    if (b2_lds_per_wave(db->b2_maxn_pad, db->b2_dec_cap, db->b2_nctr) * B2_WAVES > 160 * 1024) return 0;
    const size_t tbl = (size_t)db->n_rsegs * db->b2_nctr;
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
        if (b3_launch_emit<false>(db, db->stream)) return 1;
    } else {
        if (b2_launch_emit<false>(db, 0, db->n_rsegs, 0, db->stream)) return 1;
    }
    HIP_TRY(hipStreamSynchronize(db->stream));
    std::vector<uint32_t> counts(tbl);
    HIP_TRY(hipMemcpy(counts.data(), db->b2_table, tbl * 4, hipMemcpyDeviceToHost));
    // bucket-major record order: all records of (bucket, class) c are contiguous, segment by segment
    std::vector<uint32_t> bases(tbl);
    std::vector<uint64_t> cstart(db->b2_nctr + 1, 0);
    uint64_t run = 0;
    for (uint32_t c = 0; c < db->b2_nctr; ++c) {
        cstart[c] = run;
        for (uint32_t sgi = 0; sgi < db->n_rsegs; ++sgi) {
            bases[(size_t)sgi * db->b2_nctr + c] = (uint32_t)run;
            run += counts[(size_t)sgi * db->b2_nctr + c];
        }
    }
    cstart[db->b2_nctr] = run;
    if (run >= (1ull << 32)) return 0;                    // record index must fit 32 bits
    db->b2_total = run;
    std::vector<B2Item> items;
    uint32_t CH = 8192;
    if (const char* e = getenv("KMDB_K2_CHUNK")) CH = std::max<uint32_t>(256, (uint32_t)strtoul(e, nullptr, 10));
    for (uint32_t X = 0, c = 0; X < NB; ++X)
        for (uint32_t Y = 0; Y <= X; ++Y)
            for (uint32_t cls = 0; cls < 2; ++cls, ++c)
                for (uint64_t b = cstart[c]; b < cstart[c + 1]; b += CH)
                    items.push_back({X, Y, cls, (uint32_t)b, (uint32_t)std::min<uint64_t>(b + CH, cstart[c + 1])});
    HIP_TRY(hipMemcpy(db->b2_table, bases.data(), tbl * 4, hipMemcpyHostToDevice));
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
int b2_prepare(kmdb_db* db, uint32_t max_n, bool chain_ok, const std::vector<uint32_t>& perm, const std::vector<uint32_t>& nl,
               const std::vector<uint32_t>& seg_anc, const std::vector<uint32_t>& seg_anc_n) {
    const uint64_t N = db->N, P = db->P;
    if (N < 2 || P == 0 || N > 2048 || max_n > 1024) return 0;
    if (chain_ok) {
        db->b3_n_long = (uint32_t)perm.size();
        HIP_TRY(hipMalloc((void**)&db->b3_perm, std::max<size_t>(perm.size(), 1) * 4));
        if (!perm.empty()) HIP_TRY(hipMemcpy(db->b3_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&db->b3_nl, P * 4));
        HIP_TRY(hipMemcpy(db->b3_nl, nl.data(), P * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&db->b3_seg_anc, std::max<size_t>(seg_anc.size(), 1) * 4));
        HIP_TRY(hipMalloc((void**)&db->b3_seg_anc_n, std::max<size_t>(seg_anc_n.size(), 1) * 4));
        if (!seg_anc.empty()) HIP_TRY(hipMemcpy(db->b3_seg_anc, seg_anc.data(), seg_anc.size() * 4, hipMemcpyHostToDevice));
        if (!seg_anc_n.empty()) HIP_TRY(hipMemcpy(db->b3_seg_anc_n, seg_anc_n.data(), seg_anc_n.size() * 4, hipMemcpyHostToDevice));
    }
    uint32_t forced = 0;
    if (const char* e = getenv("KMDB_BLOCK_WIDTH")) forced = (uint32_t)strtoul(e, nullptr, 10);
    std::vector<uint32_t> cands = {64, 60, 56, 52, 50, 48, 44, 40, 36, 32};
    if (forced >= 8 && forced <= 64) cands = {forced};
    uint32_t best_w = 0;
    uint64_t best_cost = ~0ull;
    for (uint32_t wd : cands) {
        bool fits = false;
        if (b2_prepare_width(db, wd, max_n, chain_ok, &fits)) { b2_release_width(db); return 1; }
        if (fits) {
            // records dominate K1/K2; wider per-lane register sets (more blocks) make K1 a little dearer
            const uint64_t cost = db->b2_total + P * (db->b3_nbw > 16 ? (db->b3_nbw - 16) : 0) / 64;
            if (cost < best_cost) { best_cost = cost; best_w = wd; }
        }
        b2_release_width(db);
    }
    if (!best_w) return 0;
    bool fits = false;
    if (b2_prepare_width(db, best_w, max_n, chain_ok, &fits)) { b2_release_width(db); return 1; }
    if (!fits) b2_release_width(db);
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

    // ---- DFS-ordered arrays, bit-packed streams, cost model ----------------------------------
    std::vector<uint4> meta(P);
    std::vector<uint64_t> bitpos(P);
    std::vector<int32_t> parent(P);
    std::vector<uint32_t> w(P + 1, 0);
    std::vector<uint64_t> bits;
    bits.reserve(v->n_data_words / 4 + 16);
    BitWriter bw(bits);
    std::vector<uint64_t> cost_prefix(P + 1, 0);
    uint64_t alg_bytes = 0, tree_updates = 0, sum_pairs = 0;
    uint32_t max_n = 0;
    for (uint64_t i = 0; i < P; ++i) {
        const uint32_t pid = order[i];
        const uint32_t n = v->num_samples[pid], l = v->num_local[pid], nb = v->num_bits[pid];
        if (l > n || n > N) return kmdb_set_error("kmdb_db_upload: inconsistent pattern header");
        meta[i] = make_uint4(n, l, v->last_sample_id[pid], nb);
        max_n = std::max(max_n, n);
        bitpos[i] = bw.pos;
        if (nb) bw.append(v->data + v->data_offset[pid], nb);
        const int64_t par = v->parent_id[pid];
        parent[i] = par < 0 ? -1 : (int32_t)dfs_of[par];
        w[i] = (uint32_t)v->num_kmers[pid];
        const uint64_t upd = (uint64_t)(n - l) * l + (uint64_t)l * (l ? l - 1 : 0) / 2;
        tree_updates += upd;
        sum_pairs += (uint64_t)v->num_kmers[pid] * ((uint64_t)n * (n ? n - 1 : 0) / 2);
        alg_bytes += 40 + (uint64_t)((nb + 127) / 128) * 16;
        // per-node cost in "wave instructions": decode share + one scatter instruction per 64 columns per row
        uint64_t rows_cost = 0;
        if (l) {
            // sum over t in [n-l, n) of (t/64 + 1)
            for (uint32_t blk = (n - l) / 64; blk <= (n - 1) / 64; ++blk) {
                uint32_t lo = std::max<uint32_t>(n - l, blk * 64), hi = std::min<uint32_t>(n, blk * 64 + 64);
                rows_cost += (uint64_t)(hi - lo) * (blk + 1);
            }
        }
        cost_prefix[i + 1] = cost_prefix[i] + 4 + l / 2 + rows_cost * 2;
    }
    bits.resize((bw.pos + 63) / 64 + 16, 0);           // zero padding words for the cursors' look-ahead
    alg_bytes += 4ull * (N ? N * (N - 1) / 2 : 0);

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
        // K0: nodes with more than 32 local ids (a few percent) are decoded by a second launch, longest first
        const uint32_t SHORT_MAX = 32;
        std::vector<uint32_t> perm, nl(P);
        {
            std::vector<uint32_t> cntl(max_n + 2, 0);
            uint64_t nlong = 0;
            for (uint64_t i = 0; i < P; ++i) {
                nl[i] = meta[i].x | (meta[i].y << 16);
                if (meta[i].y > SHORT_MAX) { ++cntl[max_n - meta[i].y + 1]; ++nlong; }
            }
            for (uint32_t b = 1; b < cntl.size(); ++b) cntl[b] += cntl[b - 1];
            perm.resize(nlong);
            for (uint64_t i = 0; i < P; ++i)
                if (meta[i].y > SHORT_MAX) perm[cntl[max_n - meta[i].y]++] = (uint32_t)i;
        }
        // segments of the emit kernels.  Measured: the equal-cost slices of the scatter model (long multi-clade
        // lists weigh more) also balance the emit kernel better than equal node counts do, so they are reused.
        std::vector<Segment> rsegs(segs.begin(), segs.end());
        if (dev_upload(&db->rsegs, rsegs.data(), rsegs.size())) { kmdb_db_free(db); return 1; }
        db->n_rsegs = (uint32_t)rsegs.size();
        std::vector<uint32_t> seg_anc, seg_anc_n(rsegs.size(), 0);
        const bool chain_ok = max_depth <= (uint32_t)B3_CHAIN;
        db->b3_chain_cap = std::min<uint32_t>(B3_CHAIN, std::max<uint32_t>(8, (max_depth + 7) / 8 * 8));
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
        if (b2_prepare(db, max_n, chain_ok, perm, nl, seg_anc, seg_anc_n)) { kmdb_db_free(db); return 1; }
    }
    db->stats.device_bytes += db->b2_total * 20 + (uint64_t)db->n_rsegs * db->b2_nctr * 4;
    *out = db;
    return 0;
}

extern "C" void kmdb_db_free(kmdb_db* db) {
    if (!db) return;
    (void)hipSetDevice(db->device);
    void* ptrs[] = {db->meta, db->bitpos, db->parent, db->w, db->sub_end, db->wprefix, db->bits, db->segs, db->rsegs, db->scan_tmp,
                    db->stack_scratch, db->counters, db->bucket_offset, db->slots, db->pid2dfs, db->b2_table, db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w,
                    db->b2_items, db->b3_perm, db->b3_pair_ofs, db->b3_pair_blk, db->b3_pair_mask, db->b3_seg_anc,
                    db->b3_seg_anc_n, db->b3_p0_mask, db->b3_p0_info, db->b3_nl};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& e : db->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : db->ev_k2) if (e) (void)hipEventDestroy(e);
    if (db->ev_k0) (void)hipEventDestroy(db->ev_k0);
    if (db->stream) (void)hipStreamDestroy(db->stream);
    delete db;
}

void kmdb_engine_get(kmdb_db* db, kmdb_engine_view* o) {
    o->device = db->device; o->N = db->N; o->P = db->P;
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

template <int S, int NCAP>
int launch_tile(kmdb_db* db, const A2AParams& p, uint32_t blocks, hipStream_t st) {
    const size_t lds = sizeof(WaveLds<S, NCAP>) * WAVES_PER_BLOCK;
    HIP_TRY(hipFuncSetAttribute((const void*)a2a_tile_kernel<S, NCAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((a2a_tile_kernel<S, NCAP>), dim3(blocks), dim3(WAVE * WAVES_PER_BLOCK), lds, st, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

// enqueue the whole dense pipeline on `st`; M is device memory of N(N-1)/2 uint32
int run_dense(kmdb_db* db, uint32_t* M, const kmdb_opts* opts, hipStream_t st) {
    const uint64_t N = db->N, P = db->P;
    const uint64_t cells = N ? N * (N - 1) / 2 : 0;
    uint32_t shard_index = opts ? opts->shard_index : 0, shard_count = opts && opts->shard_count ? opts->shard_count : 1;
    if (shard_index >= shard_count) return kmdb_set_error("kmdb_all2all: shard_index >= shard_count");
    const uint32_t seg_begin = (uint32_t)((uint64_t)db->n_segs * shard_index / shard_count);
    const uint32_t seg_end = (uint32_t)((uint64_t)db->n_segs * (shard_index + 1) / shard_count);
    const bool force_global = opts && (opts->flags & KMDB_FLAG_FORCE_GLOBAL_ATOMICS);

    HIP_TRY(hipEventRecord(db->ev[0], st));
    if (cells) HIP_TRY(hipMemsetAsync(M, 0, cells * 4, st));
    HIP_TRY(hipMemsetAsync(db->counters, 0, 8 * sizeof(unsigned long long), st));
    // subtree weights (reference similarity_calculator.cpp:64-72): exclusive scan of w in DFS order
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(db->scan_tmp, db->scan_tmp_bytes, db->w, db->wprefix, (int)(P + 1), st));

    A2AParams p{};
    p.meta = db->meta; p.bitpos = db->bitpos; p.parent = db->parent; p.sub_end = db->sub_end;
    p.wprefix = db->wprefix; p.bits = db->bits; p.segs = db->segs;
    p.seg_begin = seg_begin; p.seg_end = seg_end; p.M = M; p.counters = db->counters;
    p.dbg = opts ? (opts->flags >> 8) : 0;
    const uint32_t nseg = seg_end - seg_begin;
    const uint32_t blocks = (nseg + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    HIP_TRY(hipEventRecord(db->ev[1], st));
    const bool force_direct = opts && (opts->flags & 2u);
    const bool force_v1 = opts && (opts->flags & 4u);
    db->k1_ms = db->k2_ms = -1;
    if (db->b2_ready && shard_count == 1 && !force_global && !force_direct && !force_v1 && cells) {
        // v2: emit block records (K1), then ballot/popcount accumulate per 64 x 64 block (K2)
        const bool seq_emit = !db->b3_ready || (opts && (opts->flags & 8u));
        if (seq_emit) {
            if (b2_launch_emit<true>(db, 0, db->n_rsegs, p.dbg, st)) return 1;
            HIP_TRY(hipEventRecord(db->ev_k0, st));
            db->k0_ms = -1;
        } else {
            if (b3_launch_decode<false>(db, st)) return 1;
            HIP_TRY(hipEventRecord(db->ev_k0, st));
            if (b3_launch_emit<true>(db, st, p.dbg)) return 1;
            db->k0_ms = 0;
        }
        HIP_TRY(hipEventRecord(db->ev_k2[0], st));
        if (db->b2_n_items && !(p.dbg & 2))
            hipLaunchKernelGGL(b2_apply_kernel, dim3(db->b2_n_items), dim3(256), 0, st,
                               B2Recs{db->b2_rec_rows, db->b2_rec_rc, db->b2_rec_w}, (const B2Item*)db->b2_items, M, (uint32_t)N, p.dbg, db->b2_width);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(db->ev_k2[1], st));
        HIP_TRY(hipEventRecord(db->ev[2], st));
        db->k1_ms = 0;
        return 0;
    }
    if (blocks && cells && force_direct && N <= 4096) {
        const uint32_t dblocks = (nseg + DIRECT_WAVES - 1) / DIRECT_WAVES;
        if (N <= 1024) hipLaunchKernelGGL((a2a_direct_kernel<1024>), dim3(dblocks), dim3(WAVE * DIRECT_WAVES), 0, st, p);
        else hipLaunchKernelGGL((a2a_direct_kernel<4096>), dim3(dblocks), dim3(WAVE * DIRECT_WAVES), 0, st, p);
        HIP_TRY(hipGetLastError());
    } else if (blocks && cells) {
        if (!force_global && N <= 1024) {
            if (launch_tile<120, 1024>(db, p, blocks, st)) return 1;
        } else if (!force_global && N <= 4096) {
            if (launch_tile<96, 4096>(db, p, blocks, st)) return 1;
        } else {
            const size_t stride = (N + 63) / 64 * 64;
            const size_t words = (size_t)blocks * WAVES_PER_BLOCK * stride;
            if (db->stack_scratch_words < words) {
                if (db->stack_scratch) (void)hipFree(db->stack_scratch);
                db->stack_scratch = nullptr; db->stack_scratch_words = 0;
                HIP_TRY(hipMalloc((void**)&db->stack_scratch, words * 4));
                db->stack_scratch_words = words;
            }
            p.stack_scratch = db->stack_scratch;
            p.stack_stride = (uint32_t)stride;
            hipLaunchKernelGGL(a2a_global_kernel, dim3(blocks), dim3(WAVE * WAVES_PER_BLOCK), 0, st, p);
            HIP_TRY(hipGetLastError());
        }
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
        fprintf(stderr, "[kmdb prof] K1 wave-cycles (memtime ticks), phases 1..4: %llu %llu %llu %llu\n", c[1], c[2], c[3], c[4]);
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
