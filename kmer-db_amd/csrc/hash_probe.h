// hash_probe.h — hash_map_lp::find (reference src/hashmap_lp.h:308-333, hash :53-64, item :71-74, empty :78) on the slot-exact tables the
// engine keeps in HBM: murmur3 fmix32(key) & (capacity - 1), linear probing until the key or an empty slot (val == INT32_MAX).
// Shared by the new2all probe (new2all.hip) and the db2db probe (db2db.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

__device__ __forceinline__ uint32_t kmdb_fmix32(uint32_t h) {        // murmur3 finaliser (src/hashmap_lp.h:53-64)
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// the value stored with key kk in the table [off, off + cap) (cap a power of two), INT32_MAX if the key is not there.
// A probe walks consecutive slots: FOUR slots (32 bytes, the sector the first one sits in) are fetched per step, so a miss at load
// factor 0.8 — thirteen slots on average — is three or four dependent round trips to HBM instead of thirteen (the probe kernels spent
// 89 % of their wave time waiting, at 21 % active lanes: every wave waits for its longest probe).
__device__ __forceinline__ int32_t kmdb_probe(const uint64_t* __restrict__ slots, uint64_t off, uint64_t cap, uint32_t kk) {
    const uint64_t mask = cap - 1;
    uint64_t h = (uint64_t)kmdb_fmix32(kk) & mask;
    if (cap >= 4 && !(off & 1ull)) {
        uint64_t g = h & ~3ull;
        uint32_t first = (uint32_t)(h & 3ull);
        for (uint64_t seen = 0; seen < cap + 4; seen += 4) {            // (a full table — a corrupt file — ends the probe too)
            const ulonglong2 a = *(const ulonglong2*)(slots + off + g), b = *(const ulonglong2*)(slots + off + g + 2);
            const uint64_t s4[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) {
                if (t < first) continue;
                const int32_t val = (int32_t)(s4[t] >> 32);
                if (val == 0x7fffffff) return val;                     // an empty slot ends the probe (src/hashmap_lp.h:78)
                if ((uint32_t)s4[t] == kk) return val;
            }
            first = 0;
            g = (g + 4) & mask;
        }
        return 0x7fffffff;
    }
    for (uint64_t step = 0; step < cap; ++step) {
        const uint64_t it = slots[off + h];
        const int32_t val = (int32_t)(it >> 32);
        if (val == 0x7fffffff || (uint32_t)it == kk) return val;
        h = (h + 1) & mask;
    }
    return 0x7fffffff;
}

}  // namespace
