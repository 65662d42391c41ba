// block_utils.h -- wave64 / 256-thread workgroup helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gh {

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(v, off, 64);
        if (lane >= off) v += y;
    }
    return v;
}

// exclusive scan of one int per thread over a 256-thread block; `total` = block sum.
// s_w: 4 ints of LDS scratch.  Contains two barriers.
__device__ __forceinline__ int block_excl_scan256(int v, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = wave_incl_scan(v);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int t = s_w[i];
        if (i < w) base += t;
    }
    total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return base + incl - v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// Sort a[0..n) (64-bit items, all distinct except ~0 padding) ascending in place with a
// 256-thread block: every thread ranks its items by counting smaller ones (broadcast LDS
// reads, no bank conflicts), one barrier, then scatters.  O(n^2/256) compares but only two
// barriers -- much faster than a barrier-per-stage bitonic network for n <= ~1024.
// The n items must be pairwise distinct (ours are (key, position) pairs); n <= MAXI * NTHREADS.
template <int NTHREADS, int MAXI>
__device__ __forceinline__ void block_rank_sort(unsigned long long* a, int n) {
    unsigned long long it[MAXI];
    int rk[MAXI];
    const int nu = (n + NTHREADS - 1) / NTHREADS;   // items per thread actually in use (uniform)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MAXI; u++) {
        const int i = threadIdx.x + NTHREADS * u;
        it[u] = i < n ? a[i] : ~0ull;
        rk[u] = 0;
    }
    // broadcast reads (same address in every lane), 8 issued before the first compare: hipcc
    // does not pipeline this loop by itself and it would pay one LDS latency per item
    // a wave whose first thread holds no item has nothing to rank (n << NTHREADS: most of a 1024-thread block).
    // Only for the big blocks: in the 256-thread callers the test cost k_rerank_topk 302 -> 376 us per 16384 queries
    // (the compiler stopped overlapping the loop with the exact-distance gathers in front of it).
    int j = 0;
    if (NTHREADS > 256 && (int)(threadIdx.x & ~63u) >= n) j = n;
    for (; j + 8 <= n; j += 8) {
        unsigned long long x[8];
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = a[j + e];
#pragma unroll
        for (int e = 0; e < 8; e++) {
#pragma unroll
            for (int u = 0; u < MAXI; u++)
                if (u < nu) rk[u] += (x[e] < it[u]) ? 1 : 0;
        }
    }
    for (; j < n; j++) {
        const unsigned long long x = a[j];
#pragma unroll
        for (int u = 0; u < MAXI; u++)
            if (u < nu) rk[u] += (x < it[u]) ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MAXI; u++)
        if (threadIdx.x + NTHREADS * u < n) a[rk[u]] = it[u];
    __syncthreads();
}


// ---- 64-item wave sorts ----
// the value of lane ^ STRIDE.  Strides 1, 2, 4, 8 stay inside a row of 16 lanes: DPP moves on the vector ALU (quad_perm for 1 and 2;
// for 4 and 8 a row shift left for the lanes whose partner is above and a row shift right for the others, each written under its
// bank mask) -- 18 of a 64-item bitonic sort's 21 stages; __shfl_xor is two ds_bpermute_b32 through the LDS crossbar per 64-bit value,
// and 168 of those per query were what k_select_final's waves queued for (round 6)
template <int STRIDE>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v) {
    if constexpr (STRIDE == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (STRIDE == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (STRIDE == 4) {
        const int a = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);   // row_shl:4 -> banks 0, 2 (lanes 0-3, 8-11)
        return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)v, 0x114, 0xF, 0xA, false);     // row_shr:4 -> banks 1, 3
    } else if constexpr (STRIDE == 8) {
        const int a = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x108, 0xF, 0x3, false);   // row_shl:8 -> banks 0, 1
        return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)v, 0x118, 0xF, 0xC, false);     // row_shr:8 -> banks 2, 3
    } else {
        return (uint32_t)__shfl_xor((int)v, STRIDE, 64);
    }
}
template <int STRIDE>
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long x) {
    const uint32_t lo = lane_xor32<STRIDE>((uint32_t)x), hi = lane_xor32<STRIDE>((uint32_t)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_stage(unsigned long long& x, int lane) {
    const unsigned long long y = lane_xor64<STRIDE>(x);
    const bool keep_min = ((lane & STRIDE) == 0) == ((lane & SIZE) == 0);
    x = ((x < y) == keep_min) ? x : y;
}
template <int SIZE, int STRIDE, int N>
__device__ __forceinline__ void bitonic_size(unsigned long long (&x)[N], int lane) {
    if constexpr (STRIDE > 0) {
#pragma unroll
        for (int r = 0; r < N; r++) bitonic_stage<SIZE, STRIDE>(x[r], lane);
        bitonic_size<SIZE, (STRIDE >> 1), N>(x, lane);
    }
}
template <int SIZE, int N>
__device__ __forceinline__ void bitonic_all(unsigned long long (&x)[N], int lane) {
    if constexpr (SIZE <= 64) {
        bitonic_size<SIZE, (SIZE >> 1), N>(x, lane);
        bitonic_all<SIZE * 2, N>(x, lane);
    }
}

// ascending bitonic sort of 64 items, one per lane
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long x) {
    unsigned long long a[1] = {x};
    bitonic_all<2, 1>(a, (int)(threadIdx.x & 63));
    return a[0];
}
// N independent sorts, stage by stage: the 2 N cross-lane moves of a stage are in flight together (one sort after the other is a
// chain of 21 N dependent LDS-crossbar round trips: 20 k of k_select_final's 40 k cycles per query, s_memtime, round 6)
template <int N>
__device__ __forceinline__ void wave_sort64_multi(unsigned long long (&x)[N]) {
    bitonic_all<2, N>(x, (int)(threadIdx.x & 63));
}

}  // namespace gh
