// gemm.hip -- a2, GEMM form (faiss:utils/distances.cpp:215-296): squared norms and ||x||^2 + ||y||^2 - 2 x.y on the fp32
// matrix pipe (v_mfma_f32_32x32x2_f32: a k-ascending fma chain per element = what the compiled sgemm_ sums up to K = 384;
// two half chains for 384 < K <= 768, kernels.h gemm_k_split).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <utility>

#include "block_utils.h"
#include "device_math.h"
#include "filter_dev.h"
#include "kernels.h"
#include "rerank_dev.h"
#include "scan_dev.h"

namespace gh {

// ------------------------------------------------------------------------------------
// a2 (GEMM form, faiss:utils/distances.cpp:215-296): squared norms in the SSE order.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_norms(const float* __restrict__ y, int64_t n, int d,
                                                   float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = fvec_norm_L2sqr(y + i * d, d);
}
void launch_row_norms(hipStream_t s, const float* y, int64_t n, int d, float* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, n, d, out);
}

// dis = (xn + yn) - 2*ip, clamped at 0; ip = k-ordered single-accumulator fmaf chain.
// This is the exact value the fp32 MFMA path produces (v_mfma_f32_*_f32 accumulates as a
// k-ordered fmaf chain); this VALU kernel is the correctness reference for it and the
// fallback for shapes the MFMA kernel does not tile.
__global__ __launch_bounds__(256) void k_l2_gemmform_valu(const float* __restrict__ x, int nq, int d,
                                                          const float* __restrict__ y, int64_t ny,
                                                          const float* __restrict__ xn,
                                                          const float* __restrict__ yn,
                                                          float* __restrict__ out, int64_t ld_out,
                                                          int q_per_block, int ksplit) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= ny) return;
    const int q0 = blockIdx.y * q_per_block;
    const int q1 = min(nq, q0 + q_per_block);
    const float* yr = y + row * d;
    const float ynr = yn[row];
    for (int q = q0; q < q1; q++) {
        const float* xq = x + (int64_t)q * d;
        float ip = 0.f;
        for (int t = 0; t < (ksplit ? ksplit : d); t++) ip = __builtin_fmaf(xq[t], yr[t], ip);
        if (ksplit) {   // the compiled sgemm_'s second K block (gemm_k_split): its own chain, added once
            float ip2 = 0.f;
            for (int t = ksplit; t < d; t++) ip2 = __builtin_fmaf(xq[t], yr[t], ip2);
            ip = ip + ip2;
        }
        float dis = (xn[q] + ynr) - 2.f * ip;
        if (dis < 0.f) dis = 0.f;
        out[(int64_t)q * ld_out + row] = dis;
    }
}

// MFMA version: C[q][c] tile 64x64 per wave-quad; v_mfma_f32_32x32x2_f32 (exact fp32,
// k-ordered fmaf chain => bit-identical to k_l2_gemmform_valu).  Block = 256 threads =
// 4 waves, each wave owns a 32x32 output tile of a 64x64 block tile; A (queries) and B
// (centroids) k-slabs of 32 are staged through LDS.
// lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; C/D: col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ksplit > 0 (gemm_k_split: 384 < d <= 768): the K dimension in the two blocks the compiled reference's sgemm_ sums it
// in -- [0, ksplit) and [ksplit, d), each chain from zero, added once.
template <int KS>
__global__ __launch_bounds__(256) void k_l2_gemmform_mfma(const float* __restrict__ x, int nq, int d,
                                                          const float* __restrict__ y, int ny,
                                                          const float* __restrict__ xn,
                                                          const float* __restrict__ yn,
                                                          float* __restrict__ out, int64_t ld_out, int ksplit) {
    // Whole-K slabs of 128 in LDS (2 x 64 x 129 floats = 66 KB, 2 blocks / CU): all global
    // loads of a slab are issued back to back (float4, 16 per thread and operand), then each
    // wave runs 64 dependent MFMAs uninterrupted.  Row stride 129 dwords: the fragment reads
    // (row = lane & 31, fixed k) hit 32 distinct banks.
    constexpr int LD = KS + 1, NIT = KS / 16, SEG = KS / 32;   // float4 slots per thread and operand; 32-float segments per row
    extern __shared__ float s_gemm[];
    float* sA = s_gemm;            // [64][LD]
    float* sB = s_gemm + 64 * LD;  // [64][LD]
    __shared__ float s_xn[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wq = w >> 1, wc = w & 1;
    const int q_base = blockIdx.y * 64, c_base = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    // fused query norms (fvec_norm_L2sqr order): thread (row = tid >> 2, lane4 = tid & 3)
    float nacc = 0.f;
    const bool vec4 = (d & 3) == 0;
    // 64 rows x 32 float4 per operand = 2048 float4, 8 per thread.  Loads are UNCONDITIONAL on clamped
    // addresses (a branch per load would make hipcc wait for each one); out-of-range lanes are zeroed
    // when the slab is written to LDS.  The NEXT slab is requested before the MFMAs of the current one
    // (d = 768: six slabs per tile, their global latency used to be exposed once per slab).
    // slot (it) of a thread: one wave instruction covers 8 rows x 32 floats (8 lanes per 128-byte row segment,
    // coalesced); with row stride 129 the four scalar LDS stores of such an instruction hit 32 distinct banks per
    // half wave (a whole row per instruction would be 4-way conflicted)
    auto slot_r = [&](int it) { return (((w * NIT + it) / SEG) << 3) + (lane >> 3); };
    auto slot_c = [&](int it) { return (((w * NIT + it) % SEG) << 5) + ((lane & 7) << 2); };
    float4 va[NIT], vb[NIT];
    auto gload = [&](int k0, int ke) {
        const int kw = min(KS, ke - k0);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int r = slot_r(it), c4 = slot_c(it);
            const int q = min(q_base + r, nq - 1), cc = min(c_base + r, ny - 1);
            const int c4c = min(c4, kw - 4);
            va[it] = *reinterpret_cast<const float4*>(x + (int64_t)q * d + k0 + c4c);
            vb[it] = *reinterpret_cast<const float4*>(y + (int64_t)cc * d + k0 + c4c);
        }
    };
    f32x16 tot;
    const int nseg = ksplit > 0 ? 2 : 1;
    if (vec4) gload(0, ksplit > 0 ? ksplit : d);
    for (int seg = 0; seg < nseg; seg++) {
    const int kb = seg ? ksplit : 0, ke = (seg == 0 && ksplit > 0) ? ksplit : d;
    for (int k0 = kb; k0 < ke; k0 += KS) {
        const int kw = min(KS, ke - k0);
        if (vec4) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int r = slot_r(it), c4 = slot_c(it);
                const bool oka = c4 < kw && q_base + r < nq, okb = c4 < kw && c_base + r < ny;
                float* pa = sA + r * LD + c4;
                float* pb = sB + r * LD + c4;
                pa[0] = oka ? va[it].x : 0.f; pa[1] = oka ? va[it].y : 0.f;
                pa[2] = oka ? va[it].z : 0.f; pa[3] = oka ? va[it].w : 0.f;
                pb[0] = okb ? vb[it].x : 0.f; pb[1] = okb ? vb[it].y : 0.f;
                pb[2] = okb ? vb[it].z : 0.f; pb[3] = okb ? vb[it].w : 0.f;
            }
        } else {
            for (int it = 0; it < KS / 4; it++) {
                const int e = it * 256 + tid;
                const int r = e / KS, c = e % KS;
                const int q = q_base + r, cc = c_base + r;
                sA[r * LD + c] = (q < nq && c < kw) ? x[(int64_t)q * d + k0 + c] : 0.f;
                sB[r * LD + c] = (cc < ny && c < kw) ? y[(int64_t)cc * d + k0 + c] : 0.f;
            }
        }
        __syncthreads();
        if (vec4) {   // uniform: the next slab of this K block, or the first of the second block
            if (k0 + KS < ke) gload(k0 + KS, ke);
            else if (seg + 1 < nseg) gload(ksplit, d);
        }
        if (!xn) {
            const float* row = sA + (tid >> 2) * LD;
            const int l4 = tid & 3;
            for (int i = 0; i < kw; i += 4) {
                const float xv = row[i + l4];
                // full 4-blocks are fused; the masked tail block is mul + add (as built)
                if (k0 + i + 4 <= d) nacc = __builtin_fmaf(xv, xv, nacc);
                else nacc = nacc + xv * xv;
            }
        }
        // K advances in order: each MFMA consumes k, k+1 (lane >> 5 selects which); the zero
        // pad beyond d contributes fma(0, 0, acc) == acc
        const float* fa = sA + (wq * 32 + (lane & 31)) * LD + (lane >> 5);
        const float* fb = sB + (wc * 32 + (lane & 31)) * LD + (lane >> 5);
        // chunks of 16 k = 8 MFMAs: 16 fragment reads are issued first, then the dependent
        // MFMA chain runs while the next chunk's reads are in flight
        const int nch = (kw + 15) >> 4;
        for (int ch = 0; ch < nch; ch++) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a[u] = fa[ch * 16 + 2 * u];
                b[u] = fb[ch * 16 + 2 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (seg + 1 < nseg) {   // C = P1; the second block accumulates from zero
        tot = acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
    }
    }
    if (nseg == 2) {        // C += P2
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = tot[i] + acc[i];
    }
    if (!xn) {
        // (a0 + a1) + (a2 + a3) inside each 4-lane group
        const float t01 = nacc + __shfl_down(nacc, 1, 4);
        const float nn = t01 + __shfl_down(t01, 2, 4);
        if ((tid & 3) == 0) s_xn[tid >> 2] = nn;
        __syncthreads();
    }
    // epilogue: dis = (xn + yn) - 2*ip, clamp
    const int col = c_base + wc * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int lr = wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int row = q_base + lr;
        if (row < nq && col < ny) {
            const float xnr = xn ? xn[row] : s_xn[lr];
            float dis = (xnr + yn[col]) - 2.f * acc[r];
            if (dis < 0.f) dis = 0.f;
            out[(int64_t)row * ld_out + col] = dis;
        }
    }
}

// Long rows (d > 128, e.g. 768-dimensional embeddings): a 128 x 128 tile per workgroup, 64 x 64 per wave as four
// 32 x 32 accumulators, K slabs of 32 staged in LDS.  Per MFMA half as many floats are staged and half as many
// fragments read as with the 64 x 64 tile above, and four workgroups fit a CU (34 KB of LDS), so one's staging
// overlaps the others' MFMAs.  Every accumulator still receives its k in ascending order: the same chain.
// Query norms come from their own pass (xn != nullptr), d % 4 == 0.
// SPLIT (gemm_k_split: 384 < d <= 768): K in the two blocks of the compiled sgemm_, [0, ksplit) and [ksplit, d) -- the
// first block's sums wait in a second accumulator set while the second block runs, then the two are added once.
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_l2_gemmform_big(const float* __restrict__ x, int nq, int d,
                                                         const float* __restrict__ y, int ny,
                                                         const float* __restrict__ xn,
                                                         const float* __restrict__ yn,
                                                         float* __restrict__ out, int64_t ld_out, int ksplit) {
    constexpr int KS = 32, LD = KS + 1, NIT = 4;   // 128 rows x 8 float4 per operand = 4 per thread
    __shared__ float sA[128 * LD];
    __shared__ float sB[128 * LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wq = w >> 1, wc = w & 1;
    const int q_base = blockIdx.y * 128, c_base = blockIdx.x * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // a wave instruction covers 8 rows x 32 floats: coalesced 128-byte segments, conflict-free scalar LDS stores
    auto slot_r = [&](int it) { return ((w * NIT + it) << 3) + (lane >> 3); };
    const int c4 = (lane & 7) << 2;
    float4 va[NIT], vb[NIT];
    auto gload = [&](int k0, int ke) {
        const int c4c = min(c4, ke - k0 - 4);   // clamped address; out-of-range lanes are zeroed when written to LDS
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int r = slot_r(it);
            va[it] = *reinterpret_cast<const float4*>(x + (int64_t)min(q_base + r, nq - 1) * d + k0 + c4c);
            vb[it] = *reinterpret_cast<const float4*>(y + (int64_t)min(c_base + r, ny - 1) * d + k0 + c4c);
        }
    };
    f32x16 tot[2][2];
    constexpr int NSEG = SPLIT ? 2 : 1;
    gload(0, SPLIT ? ksplit : d);
    const float* fa = sA + (wq * 64 + (lane & 31)) * LD + (lane >> 5);
    const float* fb = sB + (wc * 64 + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
    for (int seg = 0; seg < NSEG; seg++) {
    const int kb = seg ? ksplit : 0, ke = (SPLIT && seg == 0) ? ksplit : d;
    for (int k0 = kb; k0 < ke; k0 += KS) {
        const int kw = min(KS, ke - k0);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int r = slot_r(it);
            const bool oka = c4 < kw && q_base + r < nq, okb = c4 < kw && c_base + r < ny;
            float* pa = sA + r * LD + c4;
            float* pb = sB + r * LD + c4;
            pa[0] = oka ? va[it].x : 0.f; pa[1] = oka ? va[it].y : 0.f;
            pa[2] = oka ? va[it].z : 0.f; pa[3] = oka ? va[it].w : 0.f;
            pb[0] = okb ? vb[it].x : 0.f; pb[1] = okb ? vb[it].y : 0.f;
            pb[2] = okb ? vb[it].z : 0.f; pb[3] = okb ? vb[it].w : 0.f;
        }
        __syncthreads();
        // uniform; lands while the 64 MFMAs below run (the next slab of this K block, or the first of the second block)
        if (k0 + KS < ke) gload(k0 + KS, ke);
        else if (SPLIT && seg == 0) gload(ksplit, d);
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {   // 8 k pairs per chunk: fragment reads first, then 32 MFMAs
            float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0[u] = fa[ch * 16 + 2 * u];
                a1[u] = fa[32 * LD + ch * 16 + 2 * u];
                b0[u] = fb[ch * 16 + 2 * u];
                b1[u] = fb[32 * LD + ch * 16 + 2 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b1[u], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b0[u], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (SPLIT && seg == 0) {   // C = P1; the second block accumulates from zero
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                tot[i][j] = acc[i][j];
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
            }
    }
    }
    if (SPLIT) {               // C += P2
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = tot[i][j][r] + acc[i][j][r];
    }
    // epilogue: dis = (xn + yn) - 2*ip, clamp
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = c_base + wc * 64 + j * 32 + (lane & 31);
            const float ync = yn[min(col, ny - 1)];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = q_base + wq * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < nq && col < ny) {
                    float dis = (xn[row] + ync) - 2.f * acc[i][j][r];
                    if (dis < 0.f) dis = 0.f;
                    out[(int64_t)row * ld_out + col] = dis;
                }
            }
        }
    }
}

// Strip variant for d <= 128 (one K slab): a workgroup keeps its 64-query tile in LDS and walks
// over `tps` consecutive 64-centroid tiles.  The next centroid tile is fetched into registers
// while the current one feeds the MFMAs, so global latency is paid once per workgroup instead
// of once per tile, and the query tile (and its norms) is loaded once per strip.  Per output
// element the accumulation is the same k-ascending fma chain as above.
__global__ __launch_bounds__(256) void k_l2_gemmform_strip(const float* __restrict__ x, int nq, int d,
                                                           const float* __restrict__ y, int ny,
                                                           const float* __restrict__ xn,
                                                           const float* __restrict__ yn,
                                                           float* __restrict__ out, int64_t ld_out,
                                                           int tps) {
    constexpr int KS = 128, LD = KS + 1;
    extern __shared__ float s_gemm[];
    float* sA = s_gemm;            // [64][LD]
    float* sB = s_gemm + 64 * LD;  // [64][LD]
    __shared__ float s_xn[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wq = w >> 1, wc = w & 1;
    const int q_base = blockIdx.y * 64;
    const int ntiles = (ny + 63) >> 6;
    const int t0 = blockIdx.x * tps, t1 = min(ntiles, t0 + tps);
    const int kw = d;   // <= KS, multiple of 4
    // thread's 8 float4 slots of a 64 x 128 tile.  One wave instruction covers 8 rows x 32
    // floats (8 lanes per 128-byte row segment, coalesced); with row stride 129 the four scalar
    // LDS stores of such an instruction hit 32 distinct banks per half wave (a whole row per
    // instruction would be 4-way conflicted).  combo = w*8 + it: row block combo>>2, segment combo&3
    auto slot_r = [&](int it) { return (((w * 8 + it) >> 2) << 3) + (lane >> 3); };
    auto slot_c = [&](int it) { return (((w * 8 + it) & 3) << 5) + ((lane & 7) << 2); };
    float4 va[8], vb[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int r = slot_r(it), c4c = min(slot_c(it), kw - 4);
        va[it] = *reinterpret_cast<const float4*>(x + (int64_t)min(q_base + r, nq - 1) * d + c4c);
        vb[it] = *reinterpret_cast<const float4*>(y + (int64_t)min(t0 * 64 + r, ny - 1) * d + c4c);
    }
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int r = slot_r(it), c4 = slot_c(it);
        const bool oka = c4 < kw && q_base + r < nq, okb = c4 < kw && t0 * 64 + r < ny;
        float* pa = sA + r * LD + c4;
        float* pb = sB + r * LD + c4;
        pa[0] = oka ? va[it].x : 0.f; pa[1] = oka ? va[it].y : 0.f;
        pa[2] = oka ? va[it].z : 0.f; pa[3] = oka ? va[it].w : 0.f;
        pb[0] = okb ? vb[it].x : 0.f; pb[1] = okb ? vb[it].y : 0.f;
        pb[2] = okb ? vb[it].z : 0.f; pb[3] = okb ? vb[it].w : 0.f;
    }
    __syncthreads();
    if (!xn) {
        // fused query norms (fvec_norm_L2sqr order): thread (row = tid >> 2, lane4 = tid & 3);
        // d % 4 == 0 here, so every 4-block is a fused one
        const float* row = sA + (tid >> 2) * LD;
        const int l4 = tid & 3;
        float nacc = 0.f;
        for (int i = 0; i < kw; i += 4) {
            const float xv = row[i + l4];
            nacc = __builtin_fmaf(xv, xv, nacc);
        }
        const float t01 = nacc + __shfl_down(nacc, 1, 4);
        const float nn = t01 + __shfl_down(t01, 2, 4);
        if ((tid & 3) == 0) s_xn[tid >> 2] = nn;
        __syncthreads();
    }
    float xnr[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int lr = wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        xnr[r] = xn ? xn[min(q_base + lr, nq - 1)] : s_xn[lr];
    }
    const float* fa = sA + (wq * 32 + (lane & 31)) * LD + (lane >> 5);
    const float* fb = sB + (wc * 32 + (lane & 31)) * LD + (lane >> 5);
    const int nch = (kw + 15) >> 4;
    for (int t = t0; t < t1; t++) {
        const bool more = t + 1 < t1;   // uniform
        if (more) {
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int r = slot_r(it), c4c = min(slot_c(it), kw - 4);
                vb[it] = *reinterpret_cast<const float4*>(y + (int64_t)min((t + 1) * 64 + r, ny - 1) * d + c4c);
            }
        }
        const int col = t * 64 + wc * 32 + (lane & 31);
        const float ync = yn[min(col, ny - 1)];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
        // fragment reads of chunk ch+1 are issued before the 8 dependent MFMAs of chunk ch
        float a0[8], b0[8], a1[8], b1[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a0[u] = fa[2 * u];
            b0[u] = fb[2 * u];
        }
        for (int ch = 0; ch < nch; ch += 2) {
            if (ch + 1 < nch) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    a1[u] = fa[(ch + 1) * 16 + 2 * u];
                    b1[u] = fb[(ch + 1) * 16 + 2 * u];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
            if (ch + 2 < nch) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    a0[u] = fa[(ch + 2) * 16 + 2 * u];
                    b0[u] = fb[(ch + 2) * 16 + 2 * u];
                }
            }
            if (ch + 1 < nch) {
#pragma unroll
                for (int u = 0; u < 8; u++)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);
            }
        }
        // epilogue: dis = (xn + yn) - 2*ip, clamp
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int lr = wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int row = q_base + lr;
            if (row < nq && col < ny) {
                float dis = (xnr[r] + ync) - 2.f * acc[r];
                if (dis < 0.f) dis = 0.f;
                out[(int64_t)row * ld_out + col] = dis;
            }
        }
        if (more) {
            __syncthreads();   // every wave is done with sB
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int r = slot_r(it), c4 = slot_c(it);
                const bool okb = c4 < kw && (t + 1) * 64 + r < ny;
                float* pb = sB + r * LD + c4;
                pb[0] = okb ? vb[it].x : 0.f; pb[1] = okb ? vb[it].y : 0.f;
                pb[2] = okb ? vb[it].z : 0.f; pb[3] = okb ? vb[it].w : 0.f;
            }
            __syncthreads();
        }
    }
}

void launch_l2_gemmform(hipStream_t s, const float* x, int nq, int d, const float* y, int64_t ny,
                        const float* xn, const float* yn, float* out, int64_t ld_out,
                        bool use_mfma) {
    if (nq <= 0 || ny <= 0) return;
    const int ksplit = gemm_k_split(d);   // the K blocks of the compiled reference's sgemm_ (kernels.h)
    // the 32x32x2 MFMA consumes k in pairs with a zero pad for odd d: fma(0,0,acc) == acc
    // exactly, so any d is bit-safe
    if (use_mfma) {
        dim3 grid((unsigned)((ny + 63) / 64), (unsigned)((nq + 63) / 64));
        constexpr size_t lds = 2 * 64 * 129 * sizeof(float);  // 66 KB > the 64 KB default cap
        static std::atomic<uint64_t> attr_set{0};   // the attribute is per device (several handles / a group in one process)
        if (first_call_on_device(attr_set)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_l2_gemmform_mfma<128>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        static const bool no_strip = getenv("GAMMA_HIP_NO_GEMM_STRIP") != nullptr;
        if (d <= 128 && (d & 3) == 0 && !no_strip) {
            static std::atomic<uint64_t> attr2{0};
            if (first_call_on_device(attr2)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_l2_gemmform_strip),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            }
            // tiles per strip: amortise the query tile, but keep >= 512 workgroups
            const int ntiles = (int)((ny + 63) / 64);
            int tps = 8;
            while (tps > 1 && (int64_t)grid.y * ((ntiles + tps - 1) / tps) < 512) tps >>= 1;
            dim3 g2((unsigned)((ntiles + tps - 1) / tps), grid.y);
            hipLaunchKernelGGL(k_l2_gemmform_strip, g2, dim3(256), lds, s, x, nq, d, y, (int)ny, xn, yn, out,
                               ld_out, tps);
            return;
        }
        // xn == nullptr: query norms are computed inside the kernel from the staged tile
        static const bool no_big = getenv("GAMMA_HIP_NO_GEMM_BIG") != nullptr;
        if (xn && d > 128 && (d & 3) == 0 && nq >= 256 && !no_big) {
            const dim3 gb((unsigned)((ny + 127) / 128), (unsigned)((nq + 127) / 128));
            if (ksplit) hipLaunchKernelGGL(k_l2_gemmform_big<true>, gb, dim3(256), 0, s, x, nq, d, y, (int)ny, xn, yn, out, ld_out, ksplit);
            else hipLaunchKernelGGL(k_l2_gemmform_big<false>, gb, dim3(256), 0, s, x, nq, d, y, (int)ny, xn, yn, out, ld_out, 0);
            return;
        }
        static const int ks_env = getenv("GAMMA_HIP_GEMM_KS") ? atoi(getenv("GAMMA_HIP_GEMM_KS")) : 128;
        if (ks_env == 64)
            hipLaunchKernelGGL(k_l2_gemmform_mfma<64>, grid, dim3(256), 2 * 64 * 65 * sizeof(float), s, x, nq, d, y, (int)ny, xn, yn,
                               out, ld_out, ksplit);
        else if (ks_env == 32)
            hipLaunchKernelGGL(k_l2_gemmform_mfma<32>, grid, dim3(256), 2 * 64 * 33 * sizeof(float), s, x, nq, d, y, (int)ny, xn, yn,
                               out, ld_out, ksplit);
        else
            hipLaunchKernelGGL(k_l2_gemmform_mfma<128>, grid, dim3(256), lds, s, x, nq, d, y, (int)ny, xn, yn,
                               out, ld_out, ksplit);
    } else {
        const int64_t row_blocks = (ny + 255) / 256;
        int q_per_block = 8;
        dim3 grid((unsigned)row_blocks, (unsigned)((nq + q_per_block - 1) / q_per_block));
        hipLaunchKernelGGL(k_l2_gemmform_valu, grid, dim3(256), 0, s, x, nq, d, y, ny, xn, yn, out,
                           ld_out, q_per_block, ksplit);
    }
}

}  // namespace gh
