// flat_mfma.hip -- flat search (GammaFLATIndex::Search, index/impl/gamma_index_flat.cc:118-300) on the matrix pipe.
//
// The reference scores every row exactly (fvec_L2sqr / fvec_inner_product) and keeps a k-heap.  With a running bound
// (gamma_hip_search.cpp, flat_search_device_locked: the k-th best so far bounds what can still enter the heap) only
// ~k rows per query and pass matter; everything else is work spent proving "not among them".  That proof does not need
// the reference's arithmetic -- it needs a lower bound on the exact value.  So:
//
//   k_flat_filter   for every (query, row) of the pass an APPROXIMATE inner product on the bf16 matrix pipe
//                   (v_mfma_f32_32x32x16_bf16, 16 x the fp32 rate): x = xh + xl + rx with xh = bf16(x), xl = bf16(x - xh),
//                   |rx| <= 2^-18 |x|, and x.y ~ xh.yh + xh.yl + xl.yh -- three products, fp32 accumulation.  A row
//                   survives when the approximate distance, minus a PROVEN error margin, is within the query's bound.
//                   Survivors (row ids) go to a per-query list: about k per query and pass.
//   k_flat_exact    the survivors' EXACT distance in the reference's operation order (rerank_dist8 = fvec_L2sqr /
//                   fvec_inner_product), the validity filter, the score window, the exact test against the bound, and the
//                   append to the query's candidate list -- item for item what k_pairwise_lds<.., EMIT> appends, so
//                   k_flat_compact, the tie log and every result are unchanged.
//
// Margin (D <= 128; norms xn = |x|^2, yn = |y|^2 in fp32):
//   dropped products  |xl.yl + rx.y + x.ry| <= 3.1 * 2^-18 |x||y|          (Cauchy-Schwarz over the elements)
//   fp32 accumulation of 3 D exact products        <= 3 D 2^-24 |x||y|  = 2.3e-5 |x||y| at D = 128
//   => |ip~ - ip| <= 3.5e-5 |x||y| <= 1.75e-5 (xn + yn);  the fp32 norms add <= 2.4e-6 (xn + yn), the exact path's own
//   rounding (<= 25 roundings of partial sums <= d) <= 3e-6 (xn + yn):
//   L2:  |d~ - d_exact| <= 4.1e-5 (xn + yn)    margin used: c (xn + yn),     c = 2^-13 = 1.2e-4
//   IP:  |ip~ - ip_exact| <= 3.7e-5 |x||y|     margin used: c sqrt(xn yn),   c = 2^-13
// A row passes iff  ip~ >= thr(q, r) = a_q + g_q h_r  (one fma and one compare per element in the MFMA epilogue):
//   L2:  d~ - c (xn + yn) <= tau   <=>   ip~ >= ((1 - c) xn - tau) / 2 + (1 - c) yn / 2        (g = 1)
//   IP:  ip~ + c sqrt(xn yn) >= tau  <=>  ip~ >= tau - c sqrt(xn) sqrt(yn)
//
// The k index of the products may be permuted freely as long as queries and rows use the SAME permutation (a dot product
// does not care): MFMA slot (k step kk, half kh, t) holds element kh * D/2 + 8 kk + t, so that a lane's share of a row is
// one contiguous half row (whole cache lines, each fetched once by one lane).
// Shape of k_flat_filter: a workgroup (4 waves; two of them share a CU, so that one's epilogue and row loads run beside the
// other's MFMAs) owns 128 rows for ALL queries of the call.  Wave w keeps the bf16 hi / lo
// fragments of its 64 rows in REGISTERS for the whole launch (128 VGPRs at D = 128: the rows are read from HBM once per
// pass and converted once); queries stream through LDS in tiles of 64, pre-converted once per call into the exact LDS
// image (k_flat_prep_queries), double-buffered, one barrier per tile.  Per tile and wave: 48 MFMAs (64 rows x 32 queries
// x 128 x 3 products), 16 conflict-free 1 KB fragment reads, a 32-element epilogue.  C2 (1 M x 128, 1024 queries):
// 825 GFLOP of bf16 products instead of 134 G exact sub + fma pairs on the vector ALU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "device_math.h"
#include "filter_dev.h"
#include "block_utils.h"
#include "kernels.h"
#include "rerank_dev.h"

namespace gh {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float ff32x16 __attribute__((ext_vector_type(16)));

constexpr float FM_C = 1.220703125e-4f;   // 2^-13, see the margin above
constexpr int FM_QT = 64;                 // queries per LDS tile
constexpr int FM_ROWS = 128;              // rows per workgroup (4 waves: 2 row groups x 2 query halves)
constexpr int FM_NT = 256;                // threads per workgroup

// two floats -> two bf16 (round to nearest even), lo in bits 0..15
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 8 floats -> (hi, lo) bf16x8: hi = bf16(f), lo = bf16(f - hi)
__device__ __forceinline__ void split_bf16x8(const float* f, uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h[i] = cvt_pk_bf16(f[2 * i], f[2 * i + 1]);
        const float h0 = __uint_as_float(h[i] << 16), h1 = __uint_as_float(h[i] & 0xffff0000u);
        l[i] = cvt_pk_bf16(f[2 * i] - h0, f[2 * i + 1] - h1);   // the differences are exact in fp32
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// bytes of the LDS image of one 32-query block: [part hi | lo][k step of 16][k half of 8][query 0..31][8 bf16]
__host__ __device__ constexpr int fm_mt_bytes(int D) { return 2 * (D / 16) * 2 * 32 * 16; }

// queries -> bf16 hi / lo in the filter's LDS image, padded with zero queries to a multiple of 64.
// One thread per (query, group of 8 elements).
template <int D>
__global__ __launch_bounds__(256) void k_flat_prep_queries(const float* __restrict__ x, int nq, int nq_pad,
                                                           char* __restrict__ out) {
    constexpr int KK = D / 16;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nq_pad * (D / 8)) return;
    const int q = idx / (D / 8), kg = idx % (D / 8);
    float f[8];
#pragma unroll
    for (int t = 0; t < 8; t++) f[t] = q < nq ? x[(int64_t)q * D + kg * 8 + t] : 0.f;
    uint4 hi, lo;
    split_bf16x8(f, hi, lo);
    // elements 8 kg .. 8 kg + 7 = slot (kk, kh): kh * D/2 + 8 kk
    const int mt = q >> 5, i = q & 31, kh = kg / KK, kk = kg % KK;
    char* base = out + (int64_t)mt * fm_mt_bytes(D);
    *reinterpret_cast<uint4*>(base + ((((0 * KK + kk) * 2 + kh) * 32 + i) * 16)) = hi;
    *reinterpret_cast<uint4*>(base + ((((1 * KK + kk) * 2 + kh) * 32 + i) * 16)) = lo;
}

// per pass: the queries' thresholds a_q (and g_q for the inner product), padded to whole tiles -- bnd[2][nq_pad]
template <bool L2>
__global__ __launch_bounds__(256) void k_flat_bounds(const float* __restrict__ xn, const uint32_t* __restrict__ tau, int nq,
                                                     int nq_pad, float* __restrict__ bnd, float cm) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_pad) return;
    float av = INFINITY, gv = 0.f;   // a padding query never passes
    if (q < nq) {
        const uint32_t t = tau[q];
        const float tq = t >= 0xff800000u ? (L2 ? 3.402823466e+38f : -3.402823466e+38f) : key2f(L2 ? t : ~t);
        const float x2 = xn[q];
        if (L2) {
            av = 0.5f * ((1.f - cm) * x2 - tq);
        } else {
            av = tq;
            gv = -cm * __builtin_sqrtf(x2);
        }
    }
    bnd[q] = av;
    bnd[nq_pad + q] = gv;
}

struct FlatFilterArgs {
    const char* qimg;          // k_flat_prep_queries
    const float* bnd;          // k_flat_bounds: a_q [nq_pad] | g_q [nq_pad]
    int nq, nq_pad;
    const float* y;            // rows of the pass
    int64_t ny;
    int64_t row_base;          // store row of y[0]
    uint2* pairs;              // survivors (query, store row), any order
    int* npairs;               // appended so far (may exceed cap: overflow)
    int cap;
};
constexpr int FM_WLIST = 256;   // survivors a wave collects in LDS before it reserves room in the global list
constexpr int FM_NSUB = 32;     // the global pair list is FM_NSUB segments, each with a counter on a cache line of its own
                                // (workgroup b appends to segment b % FM_NSUB): reservations do not queue on ONE address
constexpr int FM_CTR_STRIDE = 32;   // ints between the segments' counters

template <bool L2, int D>
__global__ __launch_bounds__(FM_NT) void k_flat_filter(FlatFilterArgs a) {
    constexpr int KK = D / 16;
    constexpr int TILE = 2 * fm_mt_bytes(D);   // bytes of a 64-query tile
    constexpr int LPT = TILE / (FM_NT * 16);   // 1 KB pieces per wave and tile
    static_assert(TILE % (FM_NT * 16) == 0, "tile staged by whole rounds of the workgroup");
    extern __shared__ __attribute__((aligned(16))) char s_fm[];
    char* s_tile = s_fm;                                                  // [2][TILE]
    float* s_a = reinterpret_cast<float*>(s_fm + 2 * TILE);              // [2][64]
    float* s_g = s_a + 2 * FM_QT;                                         // [2][64]  (inner product only)
    uint2* s_list = reinterpret_cast<uint2*>(s_g + 2 * FM_QT);            // [4 waves][FM_WLIST]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rg = w >> 1, mh = w & 1, kh = lane >> 5, j = lane & 31;
    const int64_t r0 = (int64_t)blockIdx.x * FM_ROWS + rg * 64;
    // ---- this wave's 64 rows: bf16 hi / lo B fragments in registers, norms ----
    bf16x8 bh[2][KK], bl[2][KK];
    float hrow[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int64_t r = r0 + nt * 32 + j;
        const bool live = r < a.ny;
        const float* yp = a.y + (live ? r : 0) * D + (D / 2) * kh;   // this lane's half row, contiguous
        float ss = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            const float4 v0 = *reinterpret_cast<const float4*>(yp + 8 * kk);
            const float4 v1 = *reinterpret_cast<const float4*>(yp + 8 * kk + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int t = 0; t < 8; t++) ss = __builtin_fmaf(f[t], f[t], ss);
            uint4 hi, lo;
            split_bf16x8(f, hi, lo);
            bh[nt][kk] = __builtin_bit_cast(bf16x8, hi);
            bl[nt][kk] = __builtin_bit_cast(bf16x8, lo);
        }
        const float yn = ss + __shfl_xor(ss, 32, 64);
        // L2: h = (1 - c) yn / 2; inner product: h = sqrt(yn).  A row past the end never passes.
        hrow[nt] = !live ? (L2 ? INFINITY : -INFINITY) : (L2 ? 0.5f * (1.f - FM_C) * yn : __builtin_sqrtf(yn));
    }
    const int nit = a.nq_pad / FM_QT;
    // a tile goes global -> LDS directly (LDS-DMA, 16 bytes per lane, no staging registers): the image in memory IS the LDS
    // image, wave w copies 1 KB pieces w * LPT .. w * LPT + LPT - 1.  Issued at the start of the iteration BEFORE the one
    // that reads it, into the buffer every wave left behind at the previous barrier; the barrier at the end of the
    // iteration waits for it.
    auto stage = [&](int it, int b) {
        const char* src = a.qimg + (int64_t)it * TILE + (w * LPT) * 1024 + lane * 16;
        char* dst = s_tile + b * TILE + (w * LPT) * 1024;
#pragma unroll
        for (int u = 0; u < LPT; u++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + u * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 0);
        if (w == 0) {   // the tile's 64 thresholds: 4 bytes per lane
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bnd + it * FM_QT + lane),
                                             (__attribute__((address_space(3))) void*)(s_a + b * FM_QT), 4, 0, 0);
            if (!L2)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bnd + a.nq_pad + it * FM_QT + lane),
                                                 (__attribute__((address_space(3))) void*)(s_g + b * FM_QT), 4, 0, 0);
        }
    };
    uint2* wl = s_list + w * FM_WLIST;
    int wcnt = 0;   // wave-uniform
    const int sub = (int)(blockIdx.x & (FM_NSUB - 1));
    int* seg_ctr = a.npairs + sub * FM_CTR_STRIDE;
    uint2* seg = a.pairs + (int64_t)sub * a.cap;   // a.cap: capacity of ONE segment
    auto flush = [&]() {
        // the wave's list -> the global pair list: one atomic per flush
        int base = 0;
        if (lane == 0) base = atomicAdd(seg_ctr, wcnt);
        base = __builtin_amdgcn_readfirstlane(base);
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < wcnt; i += 64)
            if (base + i < a.cap) seg[base + i] = wl[i];
        __builtin_amdgcn_wave_barrier();
        wcnt = 0;
    };
    stage(0, 0);
    __syncthreads();
    for (int it = 0; it < nit; it++) {
        const int b = it & 1;
        // this tile's thresholds into registers BEFORE the next tile's LDS-DMA is issued (hipcc drains the DMA queue in
        // front of an LDS read it cannot prove independent)
        const float* pa = s_a + b * FM_QT + mh * 32 + 4 * kh;
        const float* pg = s_g + b * FM_QT + mh * 32 + 4 * kh;
        float4 av4[4], gv4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            av4[g4] = *reinterpret_cast<const float4*>(pa + 8 * g4);
            gv4[g4] = L2 ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4*>(pg + 8 * g4);
        }
        if (it + 1 < nit) stage(it + 1, b ^ 1);   // in flight during the MFMAs
        ff32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;
        const char* tb = s_tile + b * TILE + mh * fm_mt_bytes(D) + (kh * 32 + j) * 16;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(tb + ((0 * KK + kk) * 2) * 512);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(tb + ((1 * KK + kk) * 2) * 512);
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[nt][kk], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[nt][kk], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[nt][kk], acc[nt], 0, 0, 0);
            }
        }
        // ---- epilogue: C[i][j], i = (reg & 3) + 8 (reg >> 2) + 4 kh the query, j the row.  One compare per element
        //      straight into a scalar mask; survivors (about one per wave and tile) go to the wave's own LDS list -- no
        //      atomics, the count is wave-uniform ----
        unsigned long long m[2][16], any = 0ull;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float aa = e == 0 ? av4[g4].x : e == 1 ? av4[g4].y : e == 2 ? av4[g4].z : av4[g4].w;
                    const float gg = e == 0 ? gv4[g4].x : e == 1 ? gv4[g4].y : e == 2 ? gv4[g4].z : gv4[g4].w;
                    const float t = L2 ? aa + hrow[nt] : __builtin_fmaf(gg, hrow[nt], aa);
                    m[nt][4 * g4 + e] = __ballot(acc[nt][4 * g4 + e] >= t);
                    any |= m[nt][4 * g4 + e];
                }
        if (any) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned long long mm = m[nt][r];
                    if (mm) {   // scalar branch
                        const int n = __popcll(mm);
                        if (wcnt + n > FM_WLIST) flush();
                        if ((mm >> lane) & 1ull) {
                            // the lane's query: i = (r & 3) + 8 (r >> 2) + 4 kh
                            const uint32_t q = (uint32_t)(it * FM_QT + mh * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh);
                            wl[wcnt + __popcll(mm & ((1ull << lane) - 1ull))] =
                                    make_uint2(q, (uint32_t)(a.row_base + r0 + nt * 32 + j));
                        }
                        wcnt += n;
                    }
                }
        }
        __syncthreads();   // every wave is through buffer b; the next tile's LDS-DMA and bounds have landed
    }
    // what the waves still hold goes out with ONE reservation per workgroup: the counter is a single address, a returning
    // atomic per wave (31 k per pass at C2) queued the finishing workgroups behind each other
    __shared__ int s_wcnt[FM_NT / 64];
    __shared__ int s_wbase;
    if (lane == 0) s_wcnt[w] = wcnt;
    __syncthreads();
    if (tid == 0) {
        int total = 0;
#pragma unroll
        for (int i = 0; i < FM_NT / 64; i++) total += s_wcnt[i];
        s_wbase = total > 0 ? atomicAdd(seg_ctr, total) : 0;
    }
    __syncthreads();
    int base = s_wbase;
    for (int i = 0; i < w; i++) base += s_wcnt[i];
    for (int i = lane; i < wcnt; i += 64)
        if (base + i < a.cap) seg[base + i] = wl[i];
}

// the survivors' exact distances, 8 threads per survivor (the 8 AVX lane accumulators of the reference's kernels)
template <bool L2>
__global__ __launch_bounds__(256) void k_flat_exact(const uint2* __restrict__ pairs, const int* __restrict__ npairs, int cap,
                                                    const float* __restrict__ x, int d, const float* __restrict__ store,
                                                    FilterDesc filt, int use_filter, float min_score, float max_score,
                                                    FlatEmit em, int* __restrict__ overflow) {
    const int l8 = threadIdx.x & 7, g = threadIdx.x >> 3;
    // the list is FM_NSUB segments of `cap` slots (k_flat_filter): their counts, prefix-summed, index the survivors
    __shared__ int s_off[FM_NSUB + 1];
    __shared__ int s_over;
    if (threadIdx.x == 0) s_over = 0;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x < FM_NSUB ? npairs[threadIdx.x * FM_CTR_STRIDE] : 0;
        if (c > cap) s_over = 1;   // more survivors than a segment holds (no bound yet, or adversarial data)
        const int incl = wave_incl_scan(c);
        if (threadIdx.x < FM_NSUB) s_off[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) s_off[0] = 0;
    }
    __syncthreads();
    if (s_over) {   // the caller redoes the call
        if (threadIdx.x == 0 && blockIdx.x == 0) *overflow = 1;
        return;
    }
    const int n_all = s_off[FM_NSUB];
    const float sentinel = L2 ? INFINITY : -INFINITY;
    for (int s0 = blockIdx.x * 32; s0 < n_all; s0 += gridDim.x * 32) {
        const int s = s0 + g;
        const bool live = s < n_all;
        int sg = 0;   // last segment with s_off[sg] <= s
#pragma unroll
        for (int step = FM_NSUB / 2; step >= 1; step >>= 1)
            if (s_off[sg + step] <= s) sg += step;
        const uint2 pr = live ? pairs[(int64_t)sg * cap + (s - s_off[sg])] : make_uint2(0u, 0u);
        const int q = (int)pr.x;
        const int64_t row = (int64_t)pr.y;
        float dis = rerank_dist8<L2>(x + (int64_t)q * d, store + row * d, d, l8, live);
        if (live && l8 == 0) {
            const uint32_t t = em.tau[q];
            const float tau_q = t >= 0xff800000u ? (L2 ? 3.402823466e+38f : -3.402823466e+38f) : key2f(L2 ? t : ~t);
            const bool valid = !use_filter || is_valid_doc(filt, row);
            if (!valid || !(dis <= max_score && dis >= min_score)) dis = sentinel;
            if (L2 ? dis <= tau_q : dis >= tau_q) {
                const uint32_t kk = f2key(dis);
                const uint32_t key = L2 ? kk : ~kk;
                const int slot = atomicAdd(&em.cnt[(int64_t)q * em.cstride], 1);
                if (slot < em.cap) em.cand[(int64_t)q * em.cap + slot] = ((unsigned long long)key << 32) | (unsigned)row;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Long rows (128 < d <= 1536, d % 32 == 0; round 5: C5's d = 768 ran the exact vector kernel at ~1.3 TFLOP/s).  The rows'
// fragments no longer fit the registers for a whole launch, so the roles are swapped: a workgroup keeps ONE block of 32
// QUERIES -- its bf16 hi / lo image, d x 128 bytes of LDS -- for the whole launch and streams rows: 256 rows per step (64
// per wave), K in steps of 16 with the lane's 8 floats of each of its two rows loaded, split into hi / lo and fed to the same
// three products; the row's norm falls out of the same loads.  Rows are re-read once per query block (L2 / Infinity Cache:
// the query blocks walk the same row slice together).  Same survivors' list, same exact stage as the short-row kernel.
// Margin: |dropped products| <= 3.1 * 2^-18 |x||y|, fp32 accumulation of 3 d exact products <= 3 d 2^-24 |x||y| (1.83e-4 at
// d = 1024), the fp32 norms <= d 2^-24 (xn + yn), the exact path's own roundings <= (d / 8 + 4) 2^-24 (2 (xn + yn)):
//   d <= 512:  |d~ - d_exact| <= 1.4e-4 (xn + yn)   margin 2^-12 = 2.44e-4
//   d <= 1024: |d~ - d_exact| <= 2.7e-4 (xn + yn)   margin 2^-11 = 4.88e-4       (inner product: the same constants on |x||y|)
//   d <= 1536: |d~ - d_exact| <= 4.0e-4 (xn + yn)   margin 2^-10 = 9.77e-4
// Beyond 1024 the image of a query block (d x 128 bytes) no longer fits the LDS: its hi half stays there (d x 64 bytes, 96 KB
// at d = 1536), the lo fragments -- one of the three products' operands -- are read from the L2 beside the rows.
// ------------------------------------------------------------------------------------
float flat_filter_margin(int d) { return d <= 128 ? FM_C : (d <= 512 ? 2.44140625e-4f : (d <= 1024 ? 4.8828125e-4f : 9.765625e-4f)); }

__global__ __launch_bounds__(256) void k_flat_prep_queries_rt(const float* __restrict__ x, int nq, int nq_pad, int D,
                                                              char* __restrict__ out) {
    const int KK = D / 16;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nq_pad * (D / 8)) return;
    const int q = idx / (D / 8), kg = idx % (D / 8);
    float f[8];
#pragma unroll
    for (int t = 0; t < 8; t++) f[t] = q < nq ? x[(int64_t)q * D + kg * 8 + t] : 0.f;
    uint4 hi, lo;
    split_bf16x8(f, hi, lo);
    const int mt = q >> 5, i = q & 31, kh = kg / KK, kk = kg % KK;
    char* base = out + (int64_t)mt * fm_mt_bytes(D);
    *reinterpret_cast<uint4*>(base + ((((0 * KK + kk) * 2 + kh) * 32 + i) * 16)) = hi;
    *reinterpret_cast<uint4*>(base + ((((1 * KK + kk) * 2 + kh) * 32 + i) * 16)) = lo;
}

template <bool L2, bool LO_L2>   // LO_L2: d > 1024, only the hi half of the image fits the LDS (a kernel of its own: the choice costs the others nothing)
__global__ __launch_bounds__(FM_NT) void k_flat_filter_big(FlatFilterArgs a, int D, float c_margin) {
    const int KK = D / 16;
    const int IMG = fm_mt_bytes(D);
    extern __shared__ __attribute__((aligned(16))) char s_fm[];
    constexpr bool lo_in_l2 = LO_L2;
    const int IMG_LDS = lo_in_l2 ? IMG / 2 : IMG;
    char* s_img = s_fm;                                                   // [IMG_LDS] the block's 32 queries
    uint2* s_list = reinterpret_cast<uint2*>(s_fm + IMG_LDS);             // [4 waves][FM_WLIST]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int kh = lane >> 5, j = lane & 31;
    const int mt = blockIdx.x;
    // the image: plain copy, once
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.qimg + (int64_t)mt * IMG);
        uint4* dst = reinterpret_cast<uint4*>(s_img);
        for (int i = tid; i < IMG_LDS / 16; i += FM_NT) dst[i] = src[i];
    }
    // this lane's 16 queries: i = (reg & 3) + 8 (reg >> 2) + 4 kh
    float av[16], gv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int qi = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        av[r] = a.bnd[qi];
        gv[r] = L2 ? 1.f : a.bnd[a.nq_pad + qi];
    }
    __syncthreads();
    uint2* wl = s_list + w * FM_WLIST;
    int wcnt = 0;   // wave-uniform
    const int sub = (int)((blockIdx.x + blockIdx.y * gridDim.x) & (FM_NSUB - 1));
    int* seg_ctr = a.npairs + sub * FM_CTR_STRIDE;
    uint2* seg = a.pairs + (int64_t)sub * a.cap;
    auto flush = [&]() {
        int base = 0;
        if (lane == 0) base = atomicAdd(seg_ctr, wcnt);
        base = __builtin_amdgcn_readfirstlane(base);
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < wcnt; i += 64)
            if (base + i < a.cap) seg[base + i] = wl[i];
        __builtin_amdgcn_wave_barrier();
        wcnt = 0;
    };
    // row slice of this workgroup: whole steps of 256 rows
    const int64_t steps = (a.ny + 255) / 256, per = (steps + gridDim.y - 1) / gridDim.y;
    const int64_t s_lo = (int64_t)blockIdx.y * per, s_hi = min(steps, s_lo + per);
    const char* tb = s_img + (kh * 32 + j) * 16;
    const char* tg = a.qimg + (int64_t)mt * IMG + (kh * 32 + j) * 16;    // the same fragment in the image in memory
    for (int64_t st = s_lo; st < s_hi; st++) {
        const int64_t r0 = st * 256 + w * 64;
        const float* yp[2];
        bool live[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int64_t r = r0 + nt * 32 + j;
            live[nt] = r < a.ny;
            yp[nt] = a.y + (live[nt] ? r : 0) * D + (D / 2) * kh;
        }
        ff32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;
        float ss[2] = {0.f, 0.f};
        float4 v[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            v[nt][0] = *reinterpret_cast<const float4*>(yp[nt]);
            v[nt][1] = *reinterpret_cast<const float4*>(yp[nt] + 4);
        }
        for (int kk = 0; kk < KK; kk++) {
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const float f[8] = {v[nt][0].x, v[nt][0].y, v[nt][0].z, v[nt][0].w, v[nt][1].x, v[nt][1].y, v[nt][1].z, v[nt][1].w};
#pragma unroll
                for (int t = 0; t < 8; t++) ss[nt] = __builtin_fmaf(f[t], f[t], ss[nt]);
                uint4 hi, lo;
                split_bf16x8(f, hi, lo);
                bh[nt] = __builtin_bit_cast(bf16x8, hi);
                bl[nt] = __builtin_bit_cast(bf16x8, lo);
            }
            if (kk + 1 < KK) {   // the next step's floats, in flight during this step's products
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    v[nt][0] = *reinterpret_cast<const float4*>(yp[nt] + 8 * (kk + 1));
                    v[nt][1] = *reinterpret_cast<const float4*>(yp[nt] + 8 * (kk + 1) + 4);
                }
            }
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(tb + ((0 * KK + kk) * 2) * 512);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>((lo_in_l2 ? tg : tb) + ((1 * KK + kk) * 2) * 512);
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[nt], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[nt], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[nt], acc[nt], 0, 0, 0);
            }
        }
        float hrow[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const float yn = ss[nt] + __shfl_xor(ss[nt], 32, 64);
            hrow[nt] = !live[nt] ? (L2 ? INFINITY : -INFINITY) : (L2 ? 0.5f * (1.f - c_margin) * yn : __builtin_sqrtf(yn));
        }
        unsigned long long m[2][16], any = 0ull;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float t = L2 ? av[r] + hrow[nt] : __builtin_fmaf(gv[r], hrow[nt], av[r]);
                m[nt][r] = __ballot(acc[nt][r] >= t);
                any |= m[nt][r];
            }
        if (any) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned long long mm = m[nt][r];
                    if (mm) {   // scalar branch
                        const int n = __popcll(mm);
                        if (wcnt + n > FM_WLIST) flush();
                        if ((mm >> lane) & 1ull) {
                            const uint32_t q = (uint32_t)(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh);
                            wl[wcnt + __popcll(mm & ((1ull << lane) - 1ull))] = make_uint2(q, (uint32_t)(a.row_base + r0 + nt * 32 + j));
                        }
                        wcnt += n;
                    }
                }
        }
    }
    if (wcnt > 0) flush();
}

bool flat_filter_supported(int nq, int d, int64_t ny) {
    static const bool off = getenv("GAMMA_HIP_NO_FLAT_MFMA") != nullptr;
    static const bool off_big = getenv("GAMMA_HIP_NO_FLAT_MFMA_BIG") != nullptr;
    const bool small = d == 128 || d == 96 || d == 64 || d == 32;
    const bool big = !off_big && d > 128 && d <= 1536 && d % 32 == 0;   // k_flat_filter_big: the query block's image is d x 128 bytes (LDS; beyond 1024 half of it)
    return !off && (small || big) && nq >= 64 && ny >= 4096;
}
int64_t flat_filter_pair_cap(int nq) { return (int64_t)nq * 1024; }   // ~k survivors per query and pass are expected (all segments)
int flat_filter_counter_bytes() { return FM_NSUB * FM_CTR_STRIDE * (int)sizeof(int); }
size_t flat_filter_query_image_bytes(int nq, int d) {
    const int nq_pad = (nq + FM_QT - 1) / FM_QT * FM_QT;
    return (size_t)(nq_pad / 32) * fm_mt_bytes(d);
}

void launch_flat_prep_queries(hipStream_t s, const float* x, int nq, int d, void* image) {
    if (nq <= 0) return;
    const int nq_pad = (nq + FM_QT - 1) / FM_QT * FM_QT;
    const int n = nq_pad * (d / 8);
    char* out = static_cast<char*>(image);
#define GH_PREP(DD) hipLaunchKernelGGL((k_flat_prep_queries<DD>), dim3((n + 255) / 256), dim3(256), 0, s, x, nq, nq_pad, out)
    switch (d) {
        case 128: GH_PREP(128); break;
        case 96: GH_PREP(96); break;
        case 64: GH_PREP(64); break;
        case 32: GH_PREP(32); break;
        default:   // long rows (flat_filter_supported: d % 32 == 0)
            hipLaunchKernelGGL(k_flat_prep_queries_rt, dim3((n + 255) / 256), dim3(256), 0, s, x, nq, nq_pad, d, out);
            break;
    }
#undef GH_PREP
}

template <bool L2, int D>
static void launch_filter_t(hipStream_t s, const FlatFilterArgs& a) {
    constexpr size_t lds = 2 * 2 * (size_t)fm_mt_bytes(D) + 4 * FM_QT * sizeof(float) + (FM_NT / 64) * FM_WLIST * sizeof(uint2);
    static std::atomic<uint64_t> attr{0};   // per device
    if (first_call_on_device(attr))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_flat_filter<L2, D>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL((k_flat_filter<L2, D>), dim3((unsigned)((a.ny + FM_ROWS - 1) / FM_ROWS)), dim3(FM_NT), lds, s, a);
}

size_t flat_filter_bounds_bytes(int nq) { return (size_t)2 * ((nq + FM_QT - 1) / FM_QT * FM_QT) * sizeof(float); }

void launch_flat_filter(hipStream_t s, bool l2, int d, const void* qimage, const float* xn, const uint32_t* tau, float* bounds,
                        int nq, const float* y, int64_t ny, int64_t row_base, void* pairs, int* npairs, int64_t cap) {
    if (nq <= 0 || ny <= 0) return;
    FlatFilterArgs a;
    a.qimg = static_cast<const char*>(qimage);
    a.nq = nq;
    a.nq_pad = (nq + FM_QT - 1) / FM_QT * FM_QT;
    const float cm = flat_filter_margin(d);
    if (l2) hipLaunchKernelGGL((k_flat_bounds<true>), dim3((a.nq_pad + 255) / 256), dim3(256), 0, s, xn, tau, nq, a.nq_pad, bounds, cm);
    else hipLaunchKernelGGL((k_flat_bounds<false>), dim3((a.nq_pad + 255) / 256), dim3(256), 0, s, xn, tau, nq, a.nq_pad, bounds, cm);
    a.bnd = bounds;
    a.y = y;
    a.ny = ny;
    a.row_base = row_base;
    a.pairs = static_cast<uint2*>(pairs);
    a.npairs = npairs;
    a.cap = (int)std::min<int64_t>(cap / FM_NSUB, INT32_MAX);   // per segment
#define GH_FILT(DD)                                   \
    do {                                              \
        if (l2) launch_filter_t<true, DD>(s, a);      \
        else launch_filter_t<false, DD>(s, a);        \
    } while (0)
    switch (d) {
        case 128: GH_FILT(128); break;
        case 96: GH_FILT(96); break;
        case 64: GH_FILT(64); break;
        case 32: GH_FILT(32); break;
        default: {   // long rows: one workgroup per (block of 32 queries, slice of the rows)
            const size_t lds = (size_t)fm_mt_bytes(d) / (d > 1024 ? 2 : 1) + (FM_NT / 64) * FM_WLIST * sizeof(uint2);
            static std::atomic<uint64_t> attr{0};   // per device
            if (first_call_on_device(attr)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_flat_filter_big<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_flat_filter_big<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_flat_filter_big<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_flat_filter_big<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
            }
            const int qb = a.nq_pad / 32;
            const int64_t steps = (a.ny + 255) / 256;
            const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(steps, (2048 + qb - 1) / qb));
            if (d > 1024) {
                if (l2) hipLaunchKernelGGL((k_flat_filter_big<true, true>), dim3(qb, slices), dim3(FM_NT), lds, s, a, d, cm);
                else hipLaunchKernelGGL((k_flat_filter_big<false, true>), dim3(qb, slices), dim3(FM_NT), lds, s, a, d, cm);
            } else {
                if (l2) hipLaunchKernelGGL((k_flat_filter_big<true, false>), dim3(qb, slices), dim3(FM_NT), lds, s, a, d, cm);
                else hipLaunchKernelGGL((k_flat_filter_big<false, false>), dim3(qb, slices), dim3(FM_NT), lds, s, a, d, cm);
            }
            break;
        }
    }
#undef GH_FILT
}

void launch_flat_exact(hipStream_t s, bool l2, const void* pairs, const int* npairs, int64_t cap, const float* x, int nq, int d,
                       const float* store, const FilterDesc& filt, float min_score, float max_score, const FlatEmit& em,
                       int* overflow) {
    if (nq <= 0) return;
    const int use_filter = (filt.del_bitmap || filt.has_range || filt.n_field > 0 || filt.n_term > 0 || filt.vid2doc) ? 1 : 0;
    const dim3 grid(1024);
    const uint2* pp = static_cast<const uint2*>(pairs);
    const int icap = (int)std::min<int64_t>(cap / FM_NSUB, INT32_MAX);   // per segment
    if (l2)
        hipLaunchKernelGGL((k_flat_exact<true>), grid, dim3(256), 0, s, pp, npairs, icap, x, d, store, filt, use_filter, min_score,
                           max_score, em, overflow);
    else
        hipLaunchKernelGGL((k_flat_exact<false>), grid, dim3(256), 0, s, pp, npairs, icap, x, d, store, filt, use_filter,
                           min_score, max_score, em, overflow);
}

}  // namespace gh
