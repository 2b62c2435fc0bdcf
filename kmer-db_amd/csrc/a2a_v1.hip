// a2a_v1.hip — tree-form scatter kernels (the first generation of the dense all2all path; today the A/B
// reference of the block-record pipeline behind KMDB_FLAG_FORCE_*, and the announced fallback for the
// databases it cannot take: root paths beyond its chain table).
//
// All three walk the DFS stream like the reference walks its pattern blocks
// (reference src/similarity_calculator.cpp:110-241) — lanes decode 64 nodes' gamma streams in parallel
// (lane-per-node), the wave replays them in order on one stack of ids, and for every local id (row) adds
// the subtree weight to the cells of all earlier ids (columns): the GPU form of row_add
// (src/simd/row_add_avx2.cpp:30-124).
//   a2a_tile_kernel   updates go to a wave-private lower-triangular LDS tile over compact sample indices,
//                     written back with one HBM atomic per non-zero cell when the index space overflows
//   a2a_direct_kernel stack in LDS, every update an HBM atomic (measured ceiling: 1.6e11 updates/s)
//   a2a_global_kernel stack in global scratch, any N
#include "device_common.h"

namespace {
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
            if (Wj != 0 && nj > 1) {
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
            if (nj > (uint32_t)S) {
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


template <int S, int NCAP>
int launch_tile(kmdb_db* db, const A2AParams& p, uint32_t blocks, hipStream_t st) {
    const size_t lds = sizeof(WaveLds<S, NCAP>) * WAVES_PER_BLOCK;
    HIP_TRY(hipFuncSetAttribute((const void*)a2a_tile_kernel<S, NCAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((a2a_tile_kernel<S, NCAP>), dim3(blocks), dim3(WAVE * WAVES_PER_BLOCK), lds, st, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

int kmdb_v1_run(kmdb_db* db, uint32_t* M, uint32_t seg_begin, uint32_t seg_end, uint32_t flags, hipStream_t st) {
    const uint64_t N = db->N;
    const uint64_t cells = N ? N * (N - 1) / 2 : 0;
    A2AParams p{};
    p.meta = db->meta; p.bitpos = db->bitpos; p.parent = db->parent; p.sub_end = db->sub_end;
    p.wprefix = db->wprefix; p.bits = db->bits; p.segs = db->segs;
    p.seg_begin = seg_begin; p.seg_end = seg_end; p.M = M; p.counters = db->v1_counters;
    const uint32_t nseg = seg_end - seg_begin;
    const uint32_t blocks = (nseg + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const bool force_global = (flags & KMDB_FLAG_FORCE_GLOBAL_ATOMICS) != 0;
    const bool force_direct = (flags & KMDB_FLAG_FORCE_DIRECT) != 0;
    if (!blocks || !cells) return 0;
    if (force_direct && N <= 4096) {
        const uint32_t dblocks = (nseg + DIRECT_WAVES - 1) / DIRECT_WAVES;
        if (N <= 1024) hipLaunchKernelGGL((a2a_direct_kernel<1024>), dim3(dblocks), dim3(WAVE * DIRECT_WAVES), 0, st, p);
        else hipLaunchKernelGGL((a2a_direct_kernel<4096>), dim3(dblocks), dim3(WAVE * DIRECT_WAVES), 0, st, p);
        HIP_TRY(hipGetLastError());
    } else if (!force_global && N <= 1024) {
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
    return 0;
}
