// device_math.h -- fp32 primitives that reproduce, operation for operation, the summation
// order of the reference's faiss build (faiss:utils/distances_simd.cpp as compiled by
// gcc -O3 -mavx2 -mfma; see oracle/gamma_oracle.c for the CPU restatement that is pinned
// bit-for-bit against the real library).  Compiled with -ffp-contract=off: every fused
// multiply-add is an explicit __builtin_fmaf, IEEE round-to-nearest, denormals kept
// (hipcc default float_denorm_mode_32 = 3), so results are bit-identical to the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gh {

__device__ __forceinline__ float hsum4(float s0, float s1, float s2, float s3) {
    return (s0 + s1) + (s2 + s3);
}

// fvec_L2sqr / fvec_inner_product (AVX path, :366-437): 8 lane accumulators, fused;
// s[l] = acc[l+4] + acc[l]; 4-lane tail fused; masked tail fused; (s0+s1)+(s2+s3).
template <bool L2, typename PX, typename PY>
__device__ __forceinline__ float fvec_dist(PX x, PY y, int d) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 8 <= d; i += 8) {
#pragma unroll
        for (int l = 0; l < 8; l++) {
            if (L2) {
                float t = x[i + l] - y[i + l];
                acc[l] = __builtin_fmaf(t, t, acc[l]);
            } else {
                acc[l] = __builtin_fmaf(x[i + l], y[i + l], acc[l]);
            }
        }
    }
    float s[4];
#pragma unroll
    for (int l = 0; l < 4; l++) s[l] = acc[l + 4] + acc[l];
    int rem = d - i;
    if (rem >= 4) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            if (L2) {
                float t = x[i + l] - y[i + l];
                s[l] = __builtin_fmaf(t, t, s[l]);
            } else {
                s[l] = __builtin_fmaf(x[i + l], y[i + l], s[l]);
            }
        }
        i += 4;
        rem -= 4;
    }
#pragma unroll
    for (int l = 0; l < 3; l++) {
        if (l < rem) {
            if (L2) {
                float t = x[i + l] - y[i + l];
                s[l] = __builtin_fmaf(t, t, s[l]);
            } else {
                s[l] = __builtin_fmaf(x[i + l], y[i + l], s[l]);
            }
        }
    }
    return hsum4(s[0], s[1], s[2], s[3]);
}

// fvec_norm_L2sqr (SSE, :159-176): 4 lanes fused; masked tail NOT fused (as built).
template <typename PX>
__device__ __forceinline__ float fvec_norm_L2sqr(PX x, int d) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 4 <= d; i += 4) {
#pragma unroll
        for (int l = 0; l < 4; l++) a[l] = __builtin_fmaf(x[i + l], x[i + l], a[l]);
    }
#pragma unroll
    for (int l = 0; l < 3; l++)
        if (l < d - i) a[l] = a[l] + x[i + l] * x[i + l];
    return hsum4(a[0], a[1], a[2], a[3]);
}

// one row of fvec_inner_products_ny / fvec_L2sqr_ny (:207-345): special forms for
// d in {1,2,4,8,12} (second 4-block product rounded, first and third fused -- as built),
// otherwise the AVX fvec_* above.
template <bool L2, typename PX, typename PY>
__device__ __forceinline__ float fvec_ny_row(PX x, PY y, int d) {
    auto op = [](float a, float b) -> float {
        if (L2) {
            float t = a - b;
            return t * t;
        }
        return a * b;
    };
    auto opf = [](float a, float b, float c) -> float {
        if (L2) {
            float t = a - b;
            return __builtin_fmaf(t, t, c);
        }
        return __builtin_fmaf(a, b, c);
    };
    switch (d) {
        case 1:
            return op(x[0], y[0]);
        case 2:
            return op(x[0], y[0]) + op(x[1], y[1]);
        case 4:
            return hsum4(op(x[0], y[0]), op(x[1], y[1]), op(x[2], y[2]), op(x[3], y[3]));
        case 8: {
            float a[4];
#pragma unroll
            for (int l = 0; l < 4; l++) a[l] = op(x[4 + l], y[4 + l]);
#pragma unroll
            for (int l = 0; l < 4; l++) a[l] = opf(x[l], y[l], a[l]);
            return hsum4(a[0], a[1], a[2], a[3]);
        }
        case 12: {
            float a[4];
#pragma unroll
            for (int l = 0; l < 4; l++) a[l] = op(x[4 + l], y[4 + l]);
#pragma unroll
            for (int l = 0; l < 4; l++) a[l] = opf(x[l], y[l], a[l]);
#pragma unroll
            for (int l = 0; l < 4; l++) a[l] = opf(x[8 + l], y[8 + l], a[l]);
            return hsum4(a[0], a[1], a[2], a[3]);
        }
        default:
            return fvec_dist<L2>(x, y, d);
    }
}

// s[i] = a[i] of lane ^ 1 + a[i]: four v_add_f32 with a DPP operand (quad_perm [1, 0, 3, 2]).  __shfl_xor goes through the
// LDS crossbar (ds_bpermute + a wait); a DPP move + add written in C++ becomes moves and PACKED adds, which cannot take a
// DPP operand (4 moves + 3 copies + 2 packed adds where 4 instructions do).  The
// leading s_nop covers the VALU-write -> DPP-read hazard for inputs written just before (the assembler text is opaque to
// the hazard recognizer).  Every lane of the wave must be active.
__device__ __forceinline__ void add_xor1_x4(float a0, float a1, float a2, float a3, float& s0, float& s1, float& s2,
                                            float& s3) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}

// float <-> order-preserving uint32 key.  ascending key == ascending float (-0 < +0).
__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

}  // namespace gh
