// r04_transpose_probe.hip — 64 x 64 bit-matrix transpose across a wave: the round-3 version (permlane swaps, ds_bpermute shuffles, 64-bit
// masks) against the round-4 one (one cross-lane fetch + v_alignbit + v_bfi per 32-bit word and stage; the fetches of the stages
// 16 / 8 / 4 on the LDS crossbar (ds_swizzle) or by DPP).  Checks both against the definition on random matrices and times them.
//   hipcc --offload-arch=gfx950 -O3 -o r04_transpose_probe profiles/r04_transpose_probe.hip && ./r04_transpose_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned long long transpose64_v1(unsigned long long x, uint32_t lane) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    }
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
        if (s == 2) {
            plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, 0x4E, 0xF, 0xF, false);
            phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), 0x4E, 0xF, 0xF, false);
        } else if (s == 1) {
            plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, 0xB1, 0xF, 0xF, false);
            phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), 0xB1, 0xF, 0xF, false);
        } else {
            plo = (uint32_t)__shfl_xor((int)(uint32_t)x, s, 64);
            phi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), s, 64);
        }
        const unsigned long long pv = ((unsigned long long)phi << 32) | plo;
        x = (lane & (uint32_t)s) ? ((x & ~m) | ((pv >> s) & m)) : ((x & m) | ((pv & m) << s));
    }
    return x;
}

// per-lane constants of the stages below 32: rotate amount and keep-mask of every stage (loop invariant in the callers)
struct TrConst { uint32_t amt[5], msk[5]; };
__device__ __forceinline__ TrConst tr_const(uint32_t lane) {
    TrConst c;
    const uint32_t m[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t s = 16u >> k;
        const bool up = (lane & s) != 0;
        c.amt[k] = up ? s : 32u - s;          // the partner's word rotated right by s (upper lane of the pair) or left by s
        c.msk[k] = up ? m[k] : ~m[k];         // bits taken from the partner
    }
    return c;
}
template <int MODE>
__device__ __forceinline__ uint32_t tr_fetch(uint32_t v, int k) {
    // partner = lane ^ (16 >> k)
    if (k == 0) return MODE == 0 ? (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F) : (uint32_t)__shfl_xor((int)v, 16, 64);
    if (k == 1) return MODE == 0 ? (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x201F) : (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false);   // row_ror:8
    if (k == 2) {
        if (MODE == 0) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);
        const int a = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);          // row_shl:4 into the banks 0 and 2: lane i <- i + 4
        return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)v, 0x114, 0xF, 0xA, false);       // row_shr:4 into the banks 1 and 3: lane i <- i - 4
    }
    if (k == 3) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);               // quad_perm [1,0,3,2]
}
template <int MODE>
__device__ __forceinline__ unsigned long long transpose64_v2(unsigned long long x, const TrConst& c) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t pl = tr_fetch<MODE>(lo, k), ph = tr_fetch<MODE>(hi, k);
        const uint32_t rl = __builtin_amdgcn_alignbit(pl, pl, c.amt[k]), rh = __builtin_amdgcn_alignbit(ph, ph, c.amt[k]);
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(lo) : "v"(c.msk[k]), "v"(rl));      // (mask & partner) | (~mask & own)
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(hi) : "v"(c.msk[k]), "v"(rh));
    }
    return ((unsigned long long)hi << 32) | lo;
}

template <int V>
__global__ void k(const unsigned long long* in, unsigned long long* out, int reps) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long x = in[i];
    const TrConst c = tr_const(lane);
    for (int r = 0; r < reps; ++r) {
        if (V == 1) x = transpose64_v1(x, lane);
        else if (V == 2) x = transpose64_v2<0>(x, c);
        else x = transpose64_v2<1>(x, c);
        if (r + 1 < reps) x ^= (unsigned long long)r * 0x9E3779B97F4A7C15ull;      // keep the chain from folding
    }
    out[i] = x;
}

int main() {
    const int waves = 1 << 16, n = waves * 64;
    std::vector<unsigned long long> h(n), o(n);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
    unsigned long long *din, *dout;
    hipMalloc(&din, n * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int V = 1; V <= 3; ++V) {
        if (V == 1) hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, din, dout, 1);
        if (V == 2) hipLaunchKernelGGL(k<2>, dim3(n / 256), dim3(256), 0, 0, din, dout, 1);
        if (V == 3) hipLaunchKernelGGL(k<3>, dim3(n / 256), dim3(256), 0, 0, din, dout, 1);
        hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 256; ++w)
            for (int r = 0; r < 64; ++r) {
                unsigned long long e = 0;
                for (int c = 0; c < 64; ++c) e |= ((h[w * 64 + c] >> r) & 1ull) << c;
                if (o[w * 64 + r] != e) ++bad;
            }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 200;
        hipEventRecord(e0);
        if (V == 1) hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, din, dout, reps);
        if (V == 2) hipLaunchKernelGGL(k<2>, dim3(n / 256), dim3(256), 0, 0, din, dout, reps);
        if (V == 3) hipLaunchKernelGGL(k<3>, dim3(n / 256), dim3(256), 0, 0, din, dout, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("transpose v%d (%s): %d wrong rows of 16384, %.3f ms for %d transposes of %d waves = %.1f G wave-transposes/s\n", V,
               V == 1 ? "round 3" : V == 2 ? "alignbit + bfi, ds_swizzle" : "alignbit + bfi, DPP / bpermute", bad, ms, reps, waves, (double)reps * waves / ms / 1e6);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
