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

}  // namespace gh
