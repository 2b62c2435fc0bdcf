// r04_join_apply_probe.hip — the second level above the block records as ONE kernel: the tile join of
// r04_join_probe.hip feeding the product's apply step (csrc/a2a_blocks.hip, k2_apply_mfma: two 64 x 64 bit transposes across the wave, byte
// spreading through the LDS table, eight v_mfma_i32_32x32x32_i8 per 64 matches) and writing every finished 64 x 64 tile once.  Standalone,
// synthetic many-block nodes (r04_record_stats.py's sizes), weights 1..3; the tiles of a few block pairs are recomputed on the host from the
// definition  cell(r, c) += w [r in rows] [c in cols].
//   hipcc --offload-arch=gfx950 -O3 profiles/r04_join_apply_probe.hip -o /tmp/join_apply && /tmp/join_apply
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

__device__ __forceinline__ void lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
// 64 x 64 bit-matrix transpose across the lanes of a wave (csrc/a2a_blocks.hip, transpose64: permlane32_swap, then five stages of fetch /
// v_alignbit / v_bfi per 32-bit word)
struct TrConst { uint32_t amt[5], msk[5]; };
__device__ __forceinline__ TrConst tr_const(uint32_t lane) {
    TrConst c;
    const uint32_t m[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t s = 16u >> k;
        const bool up = (lane & s) != 0;
        c.amt[k] = up ? s : 32u - s;
        c.msk[k] = up ? m[k] : ~m[k];
    }
    return c;
}
__device__ __forceinline__ uint32_t tr_fetch(uint32_t v, int k, bool up16) {
    if (k == 0) {
        const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return up16 ? a[0] : a[1];
    }
    if (k == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false);
    if (k == 2) {
        const int a = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);
        return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)v, 0x114, 0xF, 0xA, false);
    }
    if (k == 3) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned long long transpose64(unsigned long long x, const TrConst& c) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    }
    const bool up16 = c.msk[0] == 0x0000FFFFu;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t pl = tr_fetch(lo, k, up16), ph = tr_fetch(hi, k, up16);
        const uint32_t rl = __builtin_amdgcn_alignbit(pl, pl, c.amt[k]), rh = __builtin_amdgcn_alignbit(ph, ph, c.amt[k]);
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(lo) : "v"(c.msk[k]), "v"(rl));
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(hi) : "v"(c.msk[k]), "v"(rh));
    }
    return ((unsigned long long)hi << 32) | lo;
}
typedef int k2_v4i __attribute__((ext_vector_type(4)));
typedef int k2_v16i __attribute__((ext_vector_type(16)));

// APPLY = false: the join and the gathers alone (a checksum in place of the tile); true: the whole kernel
// PF: the gathers of the next step and the bitmap words of the next round are under way while a step is applied
// (one call site of the apply step — the tail of the queue is drained by a last, empty round of the same loop — and at least four waves per
// SIMD asked of the register allocator: the first edition inlined the step twice and ran at two.)
template <bool APPLY, bool PF, int MINW>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(MINW, 8))) void join_apply_kernel(const unsigned long long* __restrict__ B, const uint32_t* __restrict__ R, const unsigned long long* __restrict__ L,
                                                                const unsigned char* __restrict__ Wt, const uint32_t* __restrict__ loff, uint32_t NB, uint32_t W,
                                                                uint32_t* __restrict__ tiles_out, unsigned long long* __restrict__ sums, unsigned long long* __restrict__ n_match) {
    __shared__ uint32_t q[WAVES][QCAP];
    __shared__ uint32_t q2[WAVES][QCAP];
    __shared__ unsigned long long lut_ff[256], lut_01[256];       // byte b -> its 8 bits spread over 8 bytes (0xFF / 0x01 where set)
    __shared__ uint32_t acc[64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[WAVES][64];
    uint32_t t = blockIdx.x, X = 0;
    while ((X + 1) * (X + 2) / 2 <= t) ++X;
    const uint32_t Y = t - X * (X + 1) / 2;
    const bool diag = X == Y;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t half = lane >> 5, l31 = lane & 31u;
    if (APPLY) {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((threadIdx.x >> i) & 1u) ? 0xFFull << (8 * i) : 0ull;
        lut_ff[threadIdx.x] = v;
        lut_01[threadIdx.x] = v & 0x0101010101010101ull;
        for (uint32_t k = threadIdx.x; k < 64 * 64; k += 64 * WAVES) acc[k] = 0;
        __syncthreads();
    }
    const unsigned long long* bx = B + (size_t)X * W;
    const unsigned long long* by = B + (size_t)Y * W;
    const uint32_t* rx = R + (size_t)X * W;
    const uint32_t* ry = R + (size_t)Y * W;
    const unsigned long long* lx = L + loff[X];
    const unsigned long long* ly = L + loff[Y];
    const unsigned char* wx8 = Wt + loff[X];
    unsigned long long csum = 0, cnt = 0;
    uint32_t qn = 0;                           // wave-uniform
    k2_v16i c00 = {}, c01 = {}, c10 = {}, c11 = {};
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
    auto fetch = [&](uint32_t base, uint32_t n, unsigned long long& Rm, unsigned long long& Cm, uint32_t& Wg) {
        Rm = 0; Cm = 0; Wg = 0;
        if (lane < n) {
            const uint32_t a = q[wave][base + lane], b = q2[wave][base + lane];
            Rm = lx[a]; Cm = ly[b]; Wg = wx8[a];
        }
    };
    auto apply = [&](unsigned long long Rm, unsigned long long Cm, uint32_t Wg) {      // 64 block records (X, Y, rows, cols, weight), one per lane
        if (!APPLY) { csum += (Rm ^ (Cm * 0x9E3779B97F4A7C15ull)) + Wg; return; }
        const unsigned long long Ct = transpose64(Cm, trc);
        const unsigned long long Rt = transpose64(Rm, trc);
        unsigned long long ra0, ra1, cb0, cb1;
        halves(Rt, ra0, ra1);
        halves(Ct, cb0, cb1);
        wbuf[wave][lane] = (unsigned char)(Wg & 127u);
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
        if (PF && avail()) fetch(0u, avail(), nR, nC, nW);
        while (avail()) {
            unsigned long long Rm, Cm; uint32_t Wg;
            const uint32_t n = avail();
            if (PF) { Rm = nR; Cm = nC; Wg = nW; } else fetch(head, n, Rm, Cm, Wg);
            head += n;
            if (PF && avail()) fetch(head, avail(), nR, nC, nW);
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
    if (PF) load_words(wave * 64u, pwx, pwy, pbx, pby);
    for (uint32_t w0 = wave * 64u;; w0 += 64u * WAVES) {
        const bool last = w0 >= W;                            // one round past the bitmap: nothing to match, the queue's tail is drained
        unsigned long long wx, wy;
        uint32_t bxr, byr;
        if (PF) { wx = pwx; wy = pwy; bxr = pbx; byr = pby; load_words(w0 + 64u * WAVES, pwx, pwy, pbx, pby); }
        else load_words(w0, wx, wy, bxr, byr);
        unsigned long long m = wx & wy;
        cnt += (uint32_t)__popcll(m);
        do {
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
            qn += total;
            lds_sync();
            drain_steps(last);
        } while (__ballot(m != 0ull));
        if (last) break;
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if (lane == 0) atomicAdd(n_match, cnt);
    if (!APPLY) {
#pragma unroll
        for (int d = 32; d; d >>= 1) csum += __shfl_down(csum, d, 64);
        if (lane == 0) atomicAdd(sums + t, csum);
        return;
    }
    // the four waves' tiles merged in LDS (on the diagonal only c < r), then the tile written once
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t row0 = (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * half;
        const uint32_t v00 = (uint32_t)c00[r], v01 = (uint32_t)c01[r], v10 = (uint32_t)c10[r], v11 = (uint32_t)c11[r];
        if (v00 && (!diag || l31 < row0)) atomicAdd(&acc[row0 * 64 + l31], v00);
        if (v01 && (!diag || 32u + l31 < row0)) atomicAdd(&acc[row0 * 64 + 32u + l31], v01);
        if (v10 && (!diag || l31 < 32u + row0)) atomicAdd(&acc[(32u + row0) * 64 + l31], v10);
        if (v11 && (!diag || l31 < row0)) atomicAdd(&acc[(32u + row0) * 64 + 32u + l31], v11);
    }
    __syncthreads();
    uint32_t* dst = tiles_out + (size_t)t * 4096;
    for (uint32_t k = threadIdx.x; k < 64 * 64; k += 64 * WAVES) dst[k] = acc[k];
}

// ---- the structures built on the device from what the wide kernel would write for a many-block node: its c entries (node index g, block,
// mask) in arrival order and its weight.  No sort: an entry's place in its block's list is its rank in the block's bitmap.
__global__ void build_bitmaps_kernel(const uint32_t* __restrict__ ent_g, const uint16_t* __restrict__ ent_blk, uint32_t n, uint32_t W, unsigned long long* __restrict__ B) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicOr(&B[(size_t)ent_blk[i] * W + (ent_g[i] >> 6)], 1ull << (ent_g[i] & 63u));
}
// one workgroup of 256 threads per block: exclusive prefix popcounts of its W bitmap words, and the list length
__global__ __launch_bounds__(256) void build_ranks_kernel(const unsigned long long* __restrict__ B, uint32_t W, uint32_t* __restrict__ R, uint32_t* __restrict__ len) {
    __shared__ uint32_t part[256];
    const unsigned long long* b = B + (size_t)blockIdx.x * W;
    uint32_t* r = R + (size_t)blockIdx.x * W;
    const uint32_t per = (W + 255u) / 256u, lo = threadIdx.x * per, hi = lo + per < W ? lo + per : W;
    uint32_t s = 0;
    for (uint32_t w = lo; w < hi; ++w) s += (uint32_t)__popcll(b[w]);
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t t = 0; t < 256; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; } len[blockIdx.x] = run; }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t w = lo; w < hi; ++w) { r[w] = run; run += (uint32_t)__popcll(b[w]); }
}
__global__ void build_offsets_kernel(const uint32_t* __restrict__ len, uint32_t NB, uint32_t* __restrict__ loff) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { uint32_t run = 0; for (uint32_t x = 0; x < NB; ++x) { loff[x] = run; run += len[x]; } loff[NB] = run; }
}
__global__ void build_lists_kernel(const uint32_t* __restrict__ ent_g, const uint16_t* __restrict__ ent_blk, const unsigned long long* __restrict__ ent_mask,
                                   const unsigned char* __restrict__ wnode, uint32_t n, uint32_t W, const unsigned long long* __restrict__ B, const uint32_t* __restrict__ R,
                                   const uint32_t* __restrict__ loff, unsigned long long* __restrict__ L, unsigned char* __restrict__ Wt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = ent_g[i], X = ent_blk[i];
    const unsigned long long word = B[(size_t)X * W + (g >> 6)];
    const uint32_t pos = loff[X] + R[(size_t)X * W + (g >> 6)] + (uint32_t)__popcll(word & ((1ull << (g & 63u)) - 1ull));
    L[pos] = ent_mask[i];
    Wt[pos] = wnode[g];
}

int main(int argc, char** argv) {
    // node classes "count:blocks" on the command line (default: 262 144 nodes with 40 of the 200 blocks each); the classes are interleaved
    // over the node indices as the DFS order would mix them
    std::vector<std::pair<uint32_t, uint32_t>> classes;
    for (int a = 1; a < argc; ++a) { unsigned n = 0, c = 0; if (sscanf(argv[a], "%u:%u", &n, &c) == 2 && n && c) classes.emplace_back(n, c); }
    if (classes.empty()) classes.emplace_back(262144u, 40u);
    uint32_t G = 0;
    for (auto& cl : classes) G += cl.first;
    G = (G + 63u) & ~63u;
    const uint32_t NB = 200, W = G / 64;
    std::mt19937_64 rng(12345);
    std::vector<unsigned long long> B((size_t)NB * W, 0);
    std::vector<uint32_t> perm(NB);
    std::vector<uint32_t> cls_left;
    for (auto& cl : classes) cls_left.push_back(cl.first);
    uint64_t n_left = 0; for (auto v : cls_left) n_left += v;
    for (uint32_t g = 0; g < G && n_left; ++g) {
        uint64_t pick = rng() % n_left; uint32_t k = 0;
        while (pick >= cls_left[k]) { pick -= cls_left[k]; ++k; }
        --cls_left[k]; --n_left;
        const uint32_t C = std::min(classes[k].second, NB);
        for (uint32_t i = 0; i < NB; ++i) perm[i] = i;
        for (uint32_t i = 0; i < C; ++i) std::swap(perm[i], perm[i + rng() % (NB - i)]);
        for (uint32_t i = 0; i < C; ++i) B[(size_t)perm[i] * W + g / 64] |= 1ull << (g & 63);
    }
    std::vector<unsigned char> wnode(G);
    for (uint32_t g = 0; g < G; ++g) wnode[g] = (unsigned char)(1 + g % 3);
    std::vector<uint32_t> R((size_t)NB * W), loff(NB + 1, 0);
    std::vector<unsigned long long> L;
    std::vector<unsigned char> Wt;
    std::vector<uint32_t> ent_g; std::vector<uint16_t> ent_blk; std::vector<unsigned long long> ent_mask;      // the entries as the wide kernel would write them: node by node
    for (uint32_t X = 0; X < NB; ++X) {
        loff[X] = (uint32_t)L.size();
        uint32_t run = 0;
        for (uint32_t w = 0; w < W; ++w) {
            R[(size_t)X * W + w] = run;
            unsigned long long m = B[(size_t)X * W + w];
            run += (uint32_t)__builtin_popcountll(m);
            while (m) { const int bit = __builtin_ctzll(m); m &= m - 1; L.push_back(rng() & rng() & ((1ull << 50) - 1)); Wt.push_back(wnode[w * 64 + bit]);
                        ent_g.push_back(w * 64 + bit); ent_blk.push_back((uint16_t)X); ent_mask.push_back(L.back()); }
        }
    }
    loff[NB] = (uint32_t)L.size();
    const uint32_t tiles = NB * (NB + 1) / 2;
    auto host_tile = [&](uint32_t X, uint32_t Y, std::vector<uint32_t>& tile) {
        tile.assign(4096, 0);
        for (uint32_t w = 0; w < W; ++w) {
            const unsigned long long wx = B[(size_t)X * W + w], wy = B[(size_t)Y * W + w];
            unsigned long long m = wx & wy;
            while (m) {
                const int bit = __builtin_ctzll(m); m &= m - 1;
                const unsigned long long below = (1ull << bit) - 1ull;
                const uint32_t a = loff[X] + R[(size_t)X * W + w] + __builtin_popcountll(wx & below);
                const uint32_t b = loff[Y] + R[(size_t)Y * W + w] + __builtin_popcountll(wy & below);
                const unsigned long long rows = L[a], cols = L[b];
                for (unsigned long long rr = rows; rr; rr &= rr - 1) {
                    const int r = __builtin_ctzll(rr);
                    for (unsigned long long cc = cols; cc; cc &= cc - 1) {
                        const int c = __builtin_ctzll(cc);
                        if (X != Y || c < r) tile[r * 64 + c] += Wt[a];
                    }
                }
            }
        }
    };
    unsigned long long *dB, *dL, *dsum, *dn; uint32_t *dR, *doff, *dtiles; unsigned char* dW;
    CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dL, L.size() * 8)); CK(hipMalloc(&dR, R.size() * 4)); CK(hipMalloc(&doff, loff.size() * 4));
    CK(hipMalloc(&dW, Wt.size())); CK(hipMalloc(&dtiles, (size_t)tiles * 4096 * 4)); CK(hipMalloc(&dsum, (size_t)tiles * 8)); CK(hipMalloc(&dn, 16));
    CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(doff, loff.data(), loff.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, Wt.data(), Wt.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- the same structures built on the device from the entries (shuffled into node order first: the host loop above listed them block by block)
    float build_ms = 0;
    {
        const uint32_t n = (uint32_t)ent_g.size();
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ent_g[a] != ent_g[b] ? ent_g[a] < ent_g[b] : ent_blk[a] < ent_blk[b]; });
        std::vector<uint32_t> eg(n); std::vector<uint16_t> eb(n); std::vector<unsigned long long> em(n);
        for (uint32_t i = 0; i < n; ++i) { eg[i] = ent_g[order[i]]; eb[i] = ent_blk[order[i]]; em[i] = ent_mask[order[i]]; }
        uint32_t *deg, *dR2, *dlen, *doff2; uint16_t* deb; unsigned long long *dem, *dB2, *dL2; unsigned char *dwn, *dW2;
        CK(hipMalloc(&deg, n * 4)); CK(hipMalloc(&deb, n * 2)); CK(hipMalloc(&dem, (size_t)n * 8)); CK(hipMalloc(&dwn, G));
        CK(hipMalloc(&dB2, B.size() * 8)); CK(hipMalloc(&dR2, R.size() * 4)); CK(hipMalloc(&dlen, NB * 4)); CK(hipMalloc(&doff2, (NB + 1) * 4));
        CK(hipMalloc(&dL2, (size_t)n * 8)); CK(hipMalloc(&dW2, n));
        CK(hipMemcpy(deg, eg.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(deb, eb.data(), n * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dem, em.data(), (size_t)n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dwn, wnode.data(), G, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            CK(hipMemsetAsync(dB2, 0, B.size() * 8, 0));
            hipLaunchKernelGGL(build_bitmaps_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, deg, deb, n, W, dB2);
            hipLaunchKernelGGL(build_ranks_kernel, dim3(NB), dim3(256), 0, 0, dB2, W, dR2, dlen);
            hipLaunchKernelGGL(build_offsets_kernel, dim3(1), dim3(64), 0, 0, dlen, NB, doff2);
            hipLaunchKernelGGL(build_lists_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, deg, deb, dem, dwn, n, W, dB2, dR2, doff2, dL2, dW2);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&build_ms, e0, e1));
        }
        std::vector<unsigned long long> B2(B.size()), L2(n); std::vector<uint32_t> R2(R.size()), off2(NB + 1); std::vector<unsigned char> W2(n);
        CK(hipMemcpy(B2.data(), dB2, B.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(R2.data(), dR2, R.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(off2.data(), doff2, (NB + 1) * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(L2.data(), dL2, (size_t)n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(W2.data(), dW2, n, hipMemcpyDeviceToHost));
        const bool same = B2 == B && R2 == R && off2 == loff && L2 == L && W2 == Wt;
        printf("structures built on the device from %u entries in node order (bitmaps by atomicOr, rank directories, lists placed by rank: no sort): %.3f ms, %s\n",
               n, build_ms, same ? "equal the host's" : "DIFFER from the host's");
        if (!same) return 1;
    }
    float best_full = 1e9f, best_join = 1e9f, best_pf = 1e9f, best_join_pf = 1e9f;
    auto timed = [&](auto kern, float& best) -> int {
        float ms;
        CK(hipMemset(dsum, 0, (size_t)tiles * 8)); CK(hipMemset(dn, 0, 16));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WAVES), 0, 0, dB, dR, dL, dW, doff, NB, W, dtiles, dsum, dn);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        return 0;
    };
    float b3 = 1e9f, b3pf = 1e9f, b4 = 1e9f, b2pf = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        if (timed(join_apply_kernel<false, false, 4>, best_join)) return 1;
        if (timed(join_apply_kernel<false, true, 4>, best_join_pf)) return 1;
        if (timed(join_apply_kernel<true, false, 2>, best_full)) return 1;
        if (timed(join_apply_kernel<true, true, 2>, b2pf)) return 1;
        if (timed(join_apply_kernel<true, false, 3>, b3)) return 1;
        if (timed(join_apply_kernel<true, false, 4>, b4)) return 1;
        if (timed(join_apply_kernel<true, true, 4>, best_pf)) return 1;
        if (timed(join_apply_kernel<true, true, 3>, b3pf)) return 1;          // (last: its tiles are the ones checked)
    }
    printf("whole kernel by waves per SIMD asked of the register allocator: 2: %.3f ms (prefetch %.3f), 3: %.3f (prefetch %.3f), 4: %.3f (prefetch %.3f, spills)\n", best_full, b2pf, b3, b3pf, b4, best_pf);
    best_pf = std::min(std::min(best_pf, b3pf), b2pf);
    CK(hipGetLastError());
    unsigned long long n[2];
    CK(hipMemcpy(n, dn, 16, hipMemcpyDeviceToHost));
    int bad = 0;
    const uint32_t probe[6][2] = {{0, 0}, {7, 3}, {199, 0}, {199, 199}, {100, 57}, {150, 149}};
    std::vector<uint32_t> want, got(4096);
    unsigned long long cells = 0;
    for (auto& p : probe) {
        host_tile(p[0], p[1], want);
        CK(hipMemcpy(got.data(), dtiles + (size_t)(p[0] * (p[0] + 1) / 2 + p[1]) * 4096, 4096 * 4, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < 4096; ++k) { if (want[k] != got[k]) ++bad; cells += want[k]; }
    }
    printf("%u many-block nodes (", G);
    for (auto& cl : classes) printf(" %u x %u blocks", cl.first, cl.second);
    printf(" ) of %u blocks, weights 1..3: %u tiles, %.1f M matches (block records), lists %.1f MB, bitmaps %.1f MB\n", NB, tiles, n[0] / 1e6, L.size() * 8 / 1e6, B.size() * 8 / 1e6);
    printf("join + gather alone: %.3f ms (with prefetch %.3f); join + gather + transposes + MFMA accumulation + one write per tile: %.3f ms, with the next step's gathers "
           "and the next round's bitmap words under way %.3f ms, best of 5 (%.1f G records/s); 6 probed tiles (%llu cell updates) %s\n", best_join, best_join_pf, best_full, best_pf,
           n[0] / best_pf / 1e6, cells, bad ? "DIFFER from the definition" : "equal the definition on the host");
    if (bad) printf("  differing cells: %d\n", bad);
    return bad ? 1 : 0;
}
