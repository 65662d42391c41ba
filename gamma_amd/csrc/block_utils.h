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

}  // namespace gh
