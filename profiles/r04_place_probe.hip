// r04_place_probe.hip — could the wide records be PLACED per tile as they are emitted (one slot reservation per record on its tile's counter),
// instead of going through per-block-row chunks and the sort inside the rows (rs_hist + rs_scatter: 7.1 ms for 515 M records at 10 000 samples)?
// The product avoids hot device atomics by design (same-address atomics run at a few million per second: profiles/README.md, round 2); here
// the addresses are the 20 100 tile counters of a 10 000-sample collection, hit at random by every lane.  Timing probe, standalone:
//   (1) the reservations alone (atomicAdd returning the slot), (2) reservation + the 16-byte record stored at its tile's tail,
//   (3) the same with the lanes of a wave that hit the same tile combined first (one atomic per distinct tile and wave), (4) a plain
//   streaming store of the same bytes for scale.
//   hipcc --offload-arch=gfx950 -O3 profiles/r04_place_probe.hip -o /tmp/place_probe && /tmp/place_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// records of a wave step come from a few nodes: the 64 lanes' tiles are (X_a, X_b) pairs of ONE node's ~12 blocks => mostly distinct tiles,
// and neighbouring waves work on neighbouring nodes with other block sets: modelled as independent uniform tiles
template <int MODE>
__global__ __launch_bounds__(256) void place_kernel(uint32_t* __restrict__ ctr, const uint32_t* __restrict__ base, ulonglong2* __restrict__ rec, uint64_t n, uint32_t tiles) {
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = i0; i < n; i += stride) {
        const uint32_t h = mix((uint32_t)i * 2654435761u + (uint32_t)(i >> 32));
        const uint32_t t = (uint32_t)(((uint64_t)h * tiles) >> 32);
        if (MODE == 4) { rec[i] = ulonglong2{i, (unsigned long long)h}; continue; }
        uint32_t slot;
        if (MODE == 3) {
            // lanes with the same tile: the lowest one reserves for all of them
            const unsigned long long same = __match_any_sync(~0ull, t);
            const uint32_t lane = threadIdx.x & 63u;
            const uint32_t leader = (uint32_t)__builtin_ctzll(same), rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            uint32_t b = 0;
            if (lane == leader) b = atomicAdd(&ctr[t], (uint32_t)__popcll(same));
            slot = (uint32_t)__shfl((int)b, (int)leader, 64) + rank;
        } else slot = atomicAdd(&ctr[t], 1u);
        if (MODE >= 2) rec[(uint64_t)base[t] + slot] = ulonglong2{i, (unsigned long long)h};
        else if (slot == 0xFFFFFFFFu) rec[0] = ulonglong2{0, 0};
    }
}

int main() {
    const uint32_t tiles = 20100;
    const uint64_t n = 515ull << 20;                         // 540 M records
    // exact per-tile counts of the generator (so that every tile's region is exactly as long as it gets filled)
    std::vector<uint32_t> cnt(tiles, 0), base(tiles + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t x = (uint32_t)i * 2654435761u + (uint32_t)(i >> 32);
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        ++cnt[(uint32_t)(((uint64_t)x * tiles) >> 32)];
    }
    for (uint32_t t = 0; t < tiles; ++t) base[t + 1] = base[t] + cnt[t];
    uint32_t *dctr, *dbase; ulonglong2* drec;
    CK(hipMalloc(&dctr, tiles * 4)); CK(hipMalloc(&dbase, (tiles + 1) * 4)); CK(hipMalloc(&drec, n * 16));
    CK(hipMemcpy(dbase, base.data(), (tiles + 1) * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best[5] = {0, 1e9f, 1e9f, 1e9f, 1e9f};
    auto run = [&](auto kern, int mode) -> int {
        for (int rep = 0; rep < 3; ++rep) {
            float ms;
            CK(hipMemset(dctr, 0, tiles * 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(256 * 16), dim3(256), 0, 0, dctr, dbase, drec, n, tiles);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); best[mode] = std::min(best[mode], ms);
        }
        return 0;
    };
    if (run(place_kernel<1>, 1) || run(place_kernel<2>, 2) || run(place_kernel<3>, 3)) return 1;
    // the placement is a permutation: every tile's counter ends at its count
    std::vector<uint32_t> got(tiles);
    CK(hipMemcpy(got.data(), dctr, tiles * 4, hipMemcpyDeviceToHost));
    const bool ok = got == cnt;
    if (run(place_kernel<4>, 4)) return 1;
    printf("%.0f M records on %u tile counters: reservations alone %.2f ms (%.0f G/s); reservation + 16-byte record at the tile's tail %.2f ms; "
           "with same-tile lanes of a wave combined %.2f ms; plain streaming store of the same bytes %.2f ms (%.0f GB/s); counters %s\n",
           n / 1e6, tiles, best[1], n / best[1] / 1e6, best[2], best[3], best[4], n * 16 / best[4] / 1e6, ok ? "end at the tiles' counts" : "WRONG");
    return ok ? 0 : 1;
}
