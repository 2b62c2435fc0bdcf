// r04_join_probe.hip — what would the "second level above the block records"  cost?  A standalone timing probe, not
// part of the product.  The many-block nodes of a 10 000-sample collection (k-mers conserved from the root genome in a fifth of the 200 clades:
// 0.6 % of the emitting nodes, a third of the 515 M block records) would no longer be expanded into c (c + 1) / 2 records each, written, sorted and
// read back; instead every node writes its c (block, mask) entries once, and a tile job (X, Y) JOINS the two blocks' entry lists:
//     B_X     bitmap over the G many-block nodes: node g holds block X                     (G bits per block)
//     R_X     popcount of B_X's words before word w                                        (rank directory)
//     L_X     the masks of the nodes that hold X, in g order
// A lane ANDs one word of B_X and B_Y (64 nodes), ranks every match in both lists, and the wave queues (rank_X, rank_Y) pairs in LDS; every 64
// queued matches are one step of the apply kernel: the two masks are gathered and — here — folded into a checksum where the product would run its
// 64-record MFMA step.  The probe measures the join + gather alone, on synthetic nodes with the sizes measured by r04_record_stats.py.
//   hipcc --offload-arch=gfx950 -O3 profiles/r04_join_probe.hip -o /tmp/join_probe && /tmp/join_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr uint32_t WAVES = 4;
constexpr uint32_t QCAP = 576;                 // queued matches per wave: fewer than 64 left over + at most eight per lane and round

template <bool GATHER>
__global__ __launch_bounds__(64 * WAVES) void join_kernel(const unsigned long long* __restrict__ B, const uint32_t* __restrict__ R, const unsigned long long* __restrict__ L,
                                                          const uint32_t* __restrict__ loff, uint32_t NB, uint32_t W, unsigned long long* __restrict__ out,
                                                          unsigned long long* __restrict__ n_match) {
    __shared__ uint32_t q[WAVES][QCAP];        // rank_X | rank_Y << 16 is not enough (lists hold up to G entries): two arrays
    __shared__ uint32_t q2[WAVES][QCAP];
    // tile (X, Y), X >= Y, from the linear block index
    uint32_t t = blockIdx.x, X = 0;
    while ((X + 1) * (X + 2) / 2 <= t) ++X;    // (a handful of iterations' worth of integer work per workgroup; the product would use a table)
    const uint32_t Y = t - X * (X + 1) / 2;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long* bx = B + (size_t)X * W;
    const unsigned long long* by = B + (size_t)Y * W;
    const uint32_t* rx = R + (size_t)X * W;
    const uint32_t* ry = R + (size_t)Y * W;
    const unsigned long long* lx = L + loff[X];
    const unsigned long long* ly = L + loff[Y];
    unsigned long long acc = 0, cnt = 0;
    uint32_t qn = 0;                           // wave-uniform
    auto step = [&](uint32_t base, uint32_t n) {          // one apply step: lanes < n take a queued match each
        if (lane < n) {
            if (GATHER) {
                const unsigned long long mx = lx[q[wave][base + lane]], my = ly[q2[wave][base + lane]];
                acc += (mx ^ (my * 0x9E3779B97F4A7C15ull)) + 1ull;
            } else acc += (unsigned long long)q[wave][base + lane] * 3ull + q2[wave][base + lane];     // the join alone: ranks only, no mask gathered
        }
    };
    auto drain_full_steps = [&]() {
        uint32_t head = 0;
        while (qn - head >= 64u) { step(head, 64u); head += 64u; }
        if (head) {                                           // fewer than 64 are left: they move to the front
            const uint32_t rest = qn - head;
            uint32_t a = 0, b = 0;
            if (lane < rest) { a = q[wave][head + lane]; b = q2[wave][head + lane]; }
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) { q[wave][lane] = a; q2[wave][lane] = b; }
            __builtin_amdgcn_wave_barrier();
            qn = rest;
        }
    };
    for (uint32_t w0 = wave * 64u; w0 < W; w0 += 64u * WAVES) {
        const uint32_t w = w0 + lane;
        unsigned long long m = 0, wx = 0, wy = 0;
        uint32_t bxr = 0, byr = 0;
        if (w < W) { wx = bx[w]; wy = by[w]; m = wx & wy; bxr = rx[w]; byr = ry[w]; }
        cnt += (uint32_t)__popcll(m);
        // a word's matches enter the queue at most eight per lane and round (a diagonal tile, X == Y, matches every node of the list: 13 per word)
        while (__ballot(m != 0ull)) {
            const uint32_t left = (uint32_t)__popcll(m);
            const uint32_t k = left < 8u ? left : 8u;
            uint32_t incl = k;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += v; }
            const uint32_t total = __shfl(incl, 63, 64);
            uint32_t pos = qn + incl - k;
            for (uint32_t j = 0; j < k; ++j) {
                const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const unsigned long long below = (1ull << bit) - 1ull;
                q[wave][pos] = bxr + (uint32_t)__popcll(wx & below); q2[wave][pos] = byr + (uint32_t)__popcll(wy & below);
                ++pos;
            }
            if (qn + total > QCAP) { if (lane == 0) atomicAdd(n_match + 1, 1ull); }     // (cannot happen: fewer than 64 + 512)
            qn += total;
            __builtin_amdgcn_wave_barrier();
            drain_full_steps();
        }
    }
    if (qn) step(0u, qn);
    // tile checksum
#pragma unroll
    for (int d = 32; d; d >>= 1) { acc += __shfl_down(acc, d, 64); cnt += __shfl_down(cnt, d, 64); }
    if (lane == 0) { atomicAdd(out + t, acc); atomicAdd(n_match, cnt); }
}

int main() {
    const uint32_t G = 262144, NB = 200, C = 40, W = G / 64;
    std::mt19937_64 rng(12345);
    std::vector<unsigned long long> B((size_t)NB * W, 0);
    std::vector<std::vector<unsigned long long>> lists(NB);
    std::vector<uint32_t> perm(NB);
    for (uint32_t g = 0; g < G; ++g) {
        for (uint32_t i = 0; i < NB; ++i) perm[i] = i;
        for (uint32_t i = 0; i < C; ++i) { std::swap(perm[i], perm[i + rng() % (NB - i)]); }
        for (uint32_t i = 0; i < C; ++i) {
            const uint32_t X = perm[i];
            B[(size_t)X * W + g / 64] |= 1ull << (g & 63);
        }
    }
    // lists in g order
    std::vector<uint32_t> R((size_t)NB * W), loff(NB + 1, 0);
    std::vector<unsigned long long> L;
    for (uint32_t X = 0; X < NB; ++X) {
        loff[X] = (uint32_t)L.size();
        uint32_t run = 0;
        for (uint32_t w = 0; w < W; ++w) {
            R[(size_t)X * W + w] = run;
            unsigned long long m = B[(size_t)X * W + w];
            run += (uint32_t)__builtin_popcountll(m);
            while (m) { m &= m - 1; L.push_back(rng() & ((1ull << 50) - 1)); }
        }
    }
    loff[NB] = (uint32_t)L.size();
    const uint32_t tiles = NB * (NB + 1) / 2;
    // reference checksums of a few tiles on the host
    auto host_tile = [&](uint32_t X, uint32_t Y, unsigned long long& cnt) {
        unsigned long long acc = 0; cnt = 0;
        for (uint32_t w = 0; w < W; ++w) {
            const unsigned long long wx = B[(size_t)X * W + w], wy = B[(size_t)Y * W + w];
            unsigned long long m = wx & wy;
            while (m) {
                const int bit = __builtin_ctzll(m); m &= m - 1;
                const unsigned long long below = (1ull << bit) - 1ull;
                const unsigned long long mx = L[loff[X] + R[(size_t)X * W + w] + __builtin_popcountll(wx & below)];
                const unsigned long long my = L[loff[Y] + R[(size_t)Y * W + w] + __builtin_popcountll(wy & below)];
                acc += (mx ^ (my * 0x9E3779B97F4A7C15ull)) + 1ull; ++cnt;
            }
        }
        return acc;
    };
    unsigned long long *dB, *dL, *dout, *dn; uint32_t *dR, *doff;
    CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dL, L.size() * 8)); CK(hipMalloc(&dR, R.size() * 4)); CK(hipMalloc(&doff, loff.size() * 4));
    CK(hipMalloc(&dout, (size_t)tiles * 8)); CK(hipMalloc(&dn, 16));
    CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(doff, loff.data(), loff.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, best_join = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        float ms;
        CK(hipMemset(dout, 0, (size_t)tiles * 8)); CK(hipMemset(dn, 0, 16));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(join_kernel<false>, dim3(tiles), dim3(64 * WAVES), 0, 0, dB, dR, dL, doff, NB, W, dout, dn);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); best_join = std::min(best_join, ms);
        CK(hipMemset(dout, 0, (size_t)tiles * 8)); CK(hipMemset(dn, 0, 16));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(join_kernel<true>, dim3(tiles), dim3(64 * WAVES), 0, 0, dB, dR, dL, doff, NB, W, dout, dn);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    std::vector<unsigned long long> out(tiles); unsigned long long n[2];
    CK(hipMemcpy(out.data(), dout, (size_t)tiles * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(n, dn, 16, hipMemcpyDeviceToHost));
    int bad = 0;
    const uint32_t probe[5][2] = {{0, 0}, {7, 3}, {199, 0}, {199, 199}, {100, 57}};
    unsigned long long some = 0;
    for (auto& p : probe) {
        unsigned long long c; const unsigned long long h = host_tile(p[0], p[1], c);
        some = c;
        if (h != out[p[0] * (p[0] + 1) / 2 + p[1]]) ++bad;
    }
    printf("G %u many-block nodes x %u of %u blocks: %u tiles, %.1f M matches (%.0f per tile; host count of the last probed tile %llu), lists %.1f MB, bitmaps %.1f MB\n",
           G, C, NB, tiles, n[0] / 1e6, (double)n[0] / tiles, some, L.size() * 8 / 1e6, B.size() * 8 / 1e6);
    printf("join alone (bitmaps, ranks, queue): %.3f ms; join + gather of both masks: %.3f ms best of 5 (%.1f G matches/s); queue overflows %llu; probed tiles %s\n",
           best_join, best, n[0] / best / 1e6, n[1], bad ? "DIFFER" : "equal the host's");
    return bad ? 1 : 0;
}
