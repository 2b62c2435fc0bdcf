// device_common.h — device-side helpers shared by the kernels of libkmdb_amd.so: wave-level primitives,
// the Elias-gamma stream reader, node decoding, block mapping.  Everything is internal to each
// translation unit (anonymous namespace).
#pragma once
#include "engine_state.h"

#include <type_traits>

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

// Run-aware reader of the same streams.  Most deltas of a local list are 1 (code "0": the samples of a
// cluster are consecutive ids), so the decoder consumes a whole run of "0" codes with one count-leading-zeros
// and only walks code by code through the larger deltas.  NW stream words live in registers and are all fetched
// together: a load inside the decode loop would put a full memory latency on every word the cursor crosses
// (the loop is divergent, so some lane crosses a word in nearly every iteration).  RELOAD = false: the stream
// is known to end inside the first NW words.  RELOAD = true: when only one word is left, the next NW - 1 are
// fetched in one go.
template <int NW, bool RELOAD>
struct RunCursor {
    const uint64_t* __restrict__ bits;
    uint64_t wi;
    uint64_t c[NW];
    uint32_t s;                                    // bit offset inside c[0]
    uint32_t valid;                                // words of c[] that hold stream data
    __device__ __forceinline__ RunCursor(const uint64_t* __restrict__ b, uint64_t pos) : bits(b) {
        wi = pos >> 6;
        s = (uint32_t)pos & 63u;
        valid = NW;
#pragma unroll
        for (int k = 0; k < NW; ++k) c[k] = bits[wi + k];
    }
    __device__ __forceinline__ void advance(uint32_t nbits) {          // nbits <= 64
        s += nbits;
        if (s >= 64u) {
            s -= 64u;
            ++wi;
#pragma unroll
            for (int k = 0; k + 1 < NW; ++k) c[k] = c[k + 1];
            if (RELOAD) {
                if (--valid == 1u) {
#pragma unroll
                    for (int k = 1; k < NW; ++k) c[k] = bits[wi + k];
                    valid = NW;
                }
            } else c[NW - 1] = 0;
        }
    }
    __device__ __forceinline__ uint64_t window() const { return s ? ((c[0] << s) | (c[1] >> (64u - s))) : c[0]; }
    // number of consecutive "0" codes (deltas of 1) at the cursor, at most `limit`; consumes them
    __device__ __forceinline__ uint32_t zeros(uint32_t limit) {
        const uint64_t win = window();
        uint32_t z = win ? (uint32_t)__clzll((long long)win) : 64u;
        z = z < limit ? z : limit;
        if (z) advance(z);
        return z;
    }
    // One step of the run-aware decoder: z consecutive "0" codes (deltas of 1, at most `limit`) and, when it is
    // complete inside the 64-bit window, the code that follows them (value >= 2; 0 = none taken).  One window
    // extraction serves both.
    __device__ __forceinline__ void step(uint32_t limit, uint32_t& z, uint32_t& v) {
        const uint64_t win = window();
        z = win ? (uint32_t)__clzll((long long)win) : 64u;
        z = z < limit ? z : limit;
        v = 0;
        uint32_t used = z;
        if (z < limit && z < 64u) {
            const uint64_t rest = win << z;                          // starts with a 1 bit
            uint32_t ones = (uint32_t)__clzll((long long)~rest);
            ones = ones > 31u ? 31u : ones;
            const uint32_t len = 2u * ones + 1u;
            if (z + len <= 64u) {
                v = (uint32_t)((rest << ones) >> (63u - ones)) | (1u << ones);
                used += len;
            }
        }
        advance(used);
    }
    // one code that is known to start with a 1 bit (value >= 2)
    __device__ __forceinline__ uint32_t big() {
        const uint64_t win = window();
        uint32_t ones = (uint32_t)__clzll((long long)~win);
        ones = ones > 31u ? 31u : ones;
        const uint32_t low = (uint32_t)((win << ones) >> (63u - ones));
        advance(2u * ones + 1u);
        return low | (1u << ones);
    }
};

// The same reader on 32-bit units: ids are below 2^16, so a gamma code has at most 31 bits and one 32-bit window holds a run
// of "0" codes and, when it fits behind them, the code that follows — with v_alignbit_b32 and a 32-bit count-leading-zeros
// instead of 64-bit shifts (the decode loop is bound by VALU issue).  Unit u of the stream (MSB-first inside little-endian
// uint64 words: a word's high half comes first) is the uint32 at index u ^ 1.  NU units are fetched together before the loop;
// RELOAD: when two are left, the next NU - 2 are fetched in one go.
template <int NU, bool RELOAD>
struct RunCursor32 {
    const uint32_t* __restrict__ b32;
    uint64_t u0;                                   // stream unit held in c[0]
    uint32_t c[NU + 1];
    uint32_t s;                                    // bit offset inside c[0]
    uint32_t left;                                 // units of c[] that hold stream data (RELOAD)
    __device__ __forceinline__ RunCursor32(const uint64_t* __restrict__ b, uint64_t pos) : b32((const uint32_t*)b) {
        u0 = pos >> 5;
        s = (uint32_t)pos & 31u;
#pragma unroll
        for (int k = 0; k < NU; ++k) c[k] = b32[(u0 + k) ^ 1ull];
        c[NU] = 0;
        left = NU;
    }
    // the units are already in registers (the caller fetched them with the units of other streams: c[0 .. NU) filled by it)
    struct Preloaded {};
    __device__ __forceinline__ RunCursor32(Preloaded, uint64_t pos) : b32(nullptr) {
        u0 = pos >> 5;
        s = (uint32_t)pos & 31u;
        c[NU] = 0;
        left = NU;
    }
    __device__ __forceinline__ void shift_unit() {
#pragma unroll
        for (int k = 0; k < NU; ++k) c[k] = c[k + 1];
        ++u0;
        if (RELOAD) {
            if (--left == 2u) {
#pragma unroll
                for (int k = 2; k < NU; ++k) c[k] = b32[(u0 + k) ^ 1ull];
                left = NU;
            }
        }
    }
    // One step of the run-aware decoder: z consecutive "0" codes (deltas of 1, at most `limit`) and, when it is complete inside
    // the 32-bit window, the code that follows them (value >= 2; 0 = none taken).
    __device__ __forceinline__ void step(uint32_t limit, uint32_t& z, uint32_t& v) {
        const uint32_t win = s ? __builtin_amdgcn_alignbit(c[0], c[1], 32u - s) : c[0];
        z = win ? (uint32_t)__clz((int)win) : 32u;
        z = z < limit ? z : limit;
        v = 0;
        uint32_t used = z;
        if (z < limit && z < 32u) {
            const uint32_t rest = win << z;                          // starts with a 1 bit
            uint32_t ones = (uint32_t)__clz((int)~rest);
            if (ones <= 15u) {                                       // a delta below 2^16: the code has at most 31 bits
                const uint32_t len = 2u * ones + 1u;
                if (z + len <= 32u) {
                    v = ((rest << ones) >> (31u - ones)) | (1u << ones);
                    used += len;
                }
            } else if (z == 0u) {
                // A delta of 2^16 and more (collections beyond 65 536 samples: ids take KMDB_ID_BITS = 20 bits) — a code of 33 to 39 bits, read
                // through a 64-bit window over three units.  (Round 5 clamped the count of leading ones to 15 here: such a delta — rare, it
                // needs a list that jumps over 65 536 ids at once — was decoded wrongly.)  Behind zeros the code waits for the next step, which
                // starts at it.
                const unsigned long long w64 = (((unsigned long long)c[0] << 32) | c[1]) << s | (s ? (unsigned long long)(c[2] >> (32u - s)) : 0ull);
                uint32_t o64 = (uint32_t)__clzll((long long)~w64);
                o64 = o64 > 31u ? 31u : o64;
                v = (uint32_t)((w64 << o64) >> (63u - o64)) | (1u << o64);
                used = 2u * o64 + 1u;
            }
        }
        s += used;
        while (s >= 32u) { s -= 32u; shift_unit(); }
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
    uint32_t width, magic;                          // magic = floor(2^32 / width) + 1: __umulhi(id, magic) == id / width for every id < 2^32 / width,
                                                    // i.e. below 2^26 at width <= 64 — far above KMDB_MAX_SAMPLES = 2^20 (static_assert below)
    __host__ __device__ __forceinline__ uint32_t blk(uint32_t id) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return __umulhi(id, magic);
#else
        return id / width;
#endif
    }
    __host__ __device__ __forceinline__ uint32_t bit(uint32_t id, uint32_t b) const { return id - b * width; }
};

static_assert((uint64_t)KMDB_MAX_SAMPLES * 64u <= (1ull << 32), "BlockMap::blk: the magic division is exact only for ids below 2^32 / width");

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


}  // namespace
