// rerank_dev.h -- exact fvec_L2sqr / fvec_inner_product of one (query, raw vector) pair by EIGHT
// consecutive threads, which play the eight AVX lane accumulators of the reference's kernels
// (faiss:utils/distances_simd.cpp:366-437; device_math.h fvec_dist is the one-thread form): thread l8
// runs the k-ascending fma chain over elements l8, l8 + 8, ...; then s[l] = acc[l+4] + acc[l], the
// 4-lane and masked tails, (s0+s1)+(s2+s3).  All 8 threads of a group call it together (shuffles);
// the result is valid on thread l8 == 0.  live == false: no loads, result unspecified.
#pragma once
#include <hip/hip_runtime.h>

namespace gh {

template <bool L2>
__device__ __forceinline__ float rerank_dist8(const float* __restrict__ xq, const float* __restrict__ v, int d,
                                              int l, bool live) {
    const int d8 = d & ~7;
    float a = 0.f;
    if (live) {
        // 16 row elements (and 16 query elements) are requested before the dependent fma chain
        // starts: the chain is sequential by construction, the loads need not be
        int i = l;
        for (; i + 15 * 8 < d8; i += 16 * 8) {
            float vv[16], xx[16];
#pragma unroll
            for (int u = 0; u < 16; u++) vv[u] = v[i + 8 * u];
#pragma unroll
            for (int u = 0; u < 16; u++) xx[u] = xq[i + 8 * u];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                if (L2) {
                    const float t = xx[u] - vv[u];
                    a = __builtin_fmaf(t, t, a);
                } else {
                    a = __builtin_fmaf(xx[u], vv[u], a);
                }
            }
        }
        for (; i < d8; i += 8) {
            if (L2) {
                const float t = xq[i] - v[i];
                a = __builtin_fmaf(t, t, a);
            } else {
                a = __builtin_fmaf(xq[i], v[i], a);
            }
        }
    }
    float s = __shfl_down(a, 4, 8) + a;   // s[l] = acc[l+4] + acc[l] for l < 4
    int rem = d - d8, i = d8;
    if (live && rem >= 4) {
        if (l < 4) {
            if (L2) {
                const float t = xq[i + l] - v[i + l];
                s = __builtin_fmaf(t, t, s);
            } else {
                s = __builtin_fmaf(xq[i + l], v[i + l], s);
            }
        }
        i += 4;
        rem -= 4;
    }
    if (live && l < rem) {
        if (L2) {
            const float t = xq[i + l] - v[i + l];
            s = __builtin_fmaf(t, t, s);
        } else {
            s = __builtin_fmaf(xq[i + l], v[i + l], s);
        }
    }
    const float t01 = s + __shfl_down(s, 1, 8);   // lane 0: s0+s1, lane 2: s2+s3
    return t01 + __shfl_down(t01, 2, 8);          // lane 0: (s0+s1)+(s2+s3)
}

}  // namespace gh
