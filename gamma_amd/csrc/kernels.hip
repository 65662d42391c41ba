// kernels.hip -- hand-written gfx950 (MI355X, wave64) kernels for the Gamma retrieval
// hot path: exact pairwise distances, PQ lookup-table build, IVFPQ inverted-list scan with
// the LUT in LDS, radix/bitonic k-selection, exact re-rank.  Arithmetic order follows
// device_math.h so distances are bit-identical to the reference's faiss-CPU path.
//
// Reference stages: a2 coarse quantizer, a4 LUT build, a5/a6 list scan,
// a7 top-k, a8 validity, a9 re-rank, a10 flat.
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
// a2/a10: exact pairwise distances, one database row per thread held in registers.
//   out[q][row] = fvec_L2sqr / fvec_inner_product (x_q, y_row), reference op order.
// The row (D floats) is read from HBM once per block and reused for every query of the
// block's query range; the query vector is wave-uniform and comes in through scalar
// loads.  grid = (ceil(ny/256), ceil(nq/q_per_block)).
// ------------------------------------------------------------------------------------
template <bool L2, int D, bool FILTER>
__device__ __forceinline__ void pairwise_rowreg_body(int bx, int by, const float* __restrict__ x, int nq,
                                                     const float* __restrict__ y, int64_t ny,
                                                     float* __restrict__ out, int64_t ld_out,
                                                     int q_per_block, const FilterDesc& filt,
                                                     float min_score, float max_score,
                                                     float sentinel, int64_t row_base) {
    const int64_t row = (int64_t)bx * 256 + threadIdx.x;
    const int q0 = by * q_per_block;
    const int q1 = min(nq, q0 + q_per_block);
    float yr[D];
    const bool live = row < ny;
    if (live) {
        const float4* yp = reinterpret_cast<const float4*>(y + row * D);
#pragma unroll
        for (int i = 0; i < D / 4; i++) {
            float4 v = yp[i];
            yr[4 * i + 0] = v.x;
            yr[4 * i + 1] = v.y;
            yr[4 * i + 2] = v.z;
            yr[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < D; i++) yr[i] = 0.f;
    }
    bool valid = live;
    if (FILTER && live) valid = is_valid_doc(filt, row_base + row);
    for (int q = q0; q < q1; q++) {
        const float* xq = x + (int64_t)q * D;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < D; i += 8) {
#pragma unroll
            for (int l = 0; l < 8; l++) {
                if (L2) {
                    float t = xq[i + l] - yr[i + l];
                    acc[l] = __builtin_fmaf(t, t, acc[l]);
                } else {
                    acc[l] = __builtin_fmaf(xq[i + l], yr[i + l], acc[l]);
                }
            }
        }
        float dis = hsum4(acc[4] + acc[0], acc[5] + acc[1], acc[6] + acc[2], acc[7] + acc[3]);
        if (FILTER) {
            if (!valid || !(dis <= max_score && dis >= min_score)) dis = sentinel;
        }
        if (live) out[(int64_t)q * ld_out + row] = dis;
    }
}
template <bool L2, int D, bool FILTER>
__global__ __launch_bounds__(256) void k_pairwise_rowreg(const float* __restrict__ x, int nq,
                                                         const float* __restrict__ y, int64_t ny,
                                                         float* __restrict__ out, int64_t ld_out,
                                                         int q_per_block, FilterDesc filt,
                                                         float min_score, float max_score,
                                                         float sentinel, int64_t row_base) {
    pairwise_rowreg_body<L2, D, FILTER>(blockIdx.x, blockIdx.y, x, nq, y, ny, out, ld_out, q_per_block, filt, min_score,
                                        max_score, sentinel, row_base);
}

// Same contract, queries staged in LDS and broadcast (one ds_read_b128 feeds a whole wave), packed
// fp32 math.  k_pairwise_rowreg pulls every query vector through the scalar cache once per wave
// (512 B per wave and query at d = 128) and that path, not the VALU, is its limit; here a tile of
// PW_QT queries is read once per workgroup and the sub / fma pairs run as v_pk_add_f32 /
// v_pk_fma_f32 on two of the eight lane accumulators at a time (IEEE per component: same bits).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int PW_QT = 32;
// EMIT (flat search with a running bound, gamma_hip_search.cpp flat_search_device_locked): instead of writing
// the nq x ny distance slab, append (key << 32 | row id) of every distance within the query's current
// bound tau[q] to the query's candidate list (cand[q][cap], cnt[q]; an atomic per survivor, and there
// are about k per query and chunk).  Keys order like the distances ("smaller is better").
template <bool L2, int D, bool FILTER, bool EMIT = false>
__global__ __launch_bounds__(256) void k_pairwise_lds(const float* __restrict__ x, int nq,
                                                      const float* __restrict__ y, int64_t ny,
                                                      float* __restrict__ out, int64_t ld_out,
                                                      int q_per_block, FilterDesc filt, float min_score,
                                                      float max_score, float sentinel, int64_t row_base,
                                                      FlatEmit em) {
    // Two threads per database row: the even one owns AVX lanes 0-3 (elements 8i .. 8i+3), the odd
    // one lanes 4-7, D/2 row values each -- half the registers of a whole row per thread, so
    // several waves fit per SIMD.  Per lane the accumulation order is untouched; the pair's sums
    // meet in s[l] = acc[l+4] + acc[l] through one shuffle.
    __shared__ float4 s_x[PW_QT * D / 4];
    // EMIT: the queries' bounds AS DISTANCES.  key <= tau (keys order like the distances; neither side is ever -0 or NaN
    // here) is dis <= key2f(tau) for L2 and dis >= key2f(~tau) for the inner product; a bound at or beyond the sentinel's
    // key (none yet) becomes the largest finite value, so the one comparison also keeps the sentinel (+-inf) out
    __shared__ float s_tau[PW_QT];
    const int half = threadIdx.x & 1;
    const int64_t row = (int64_t)blockIdx.x * 128 + (threadIdx.x >> 1);
    const int q0 = blockIdx.y * q_per_block;
    const int q1 = min(nq, q0 + q_per_block);
    f32x2 yr[D / 4];
    const bool live = row < ny;
    {
        const float4* yp = reinterpret_cast<const float4*>(y + (live ? row : 0) * D) + half;
#pragma unroll
        for (int i = 0; i < D / 8; i++) {
            const float4 v = yp[2 * i];
            yr[2 * i] = f32x2{v.x, v.y};
            yr[2 * i + 1] = f32x2{v.z, v.w};
        }
    }
    bool valid = live;
    if (FILTER && live) valid = is_valid_doc(filt, row_base + row);
    for (int qt = q0; qt < q1; qt += PW_QT) {
        const int nqt = min(PW_QT, q1 - qt);
        __syncthreads();   // the previous tile has been consumed
        for (int e = threadIdx.x; e < nqt * (D / 4); e += 256)
            s_x[e] = reinterpret_cast<const float4*>(x + (int64_t)qt * D)[e];
        if (EMIT && threadIdx.x < nqt) {
            const uint32_t t = em.tau[qt + threadIdx.x];
            s_tau[threadIdx.x] = t >= 0xff800000u ? (L2 ? 3.402823466e+38f : -3.402823466e+38f) : key2f(L2 ? t : ~t);
        }
        __syncthreads();
        for (int qi = 0; qi < nqt; qi++) {
            const float4* xq = s_x + qi * (D / 4) + half;
            const float tau_q = EMIT ? s_tau[qi] : 0.f;   // requested before the row's math, used behind it
            f32x2 acc[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
            for (int i0 = 0; i0 < D / 8; i0 += 8) {
                // 8 reads (two distinct addresses per wave: broadcast), then 16 packed sub/fma pairs;
                // the barrier keeps hipcc from hoisting every read of the query to the top
                float4 xa[8];
#pragma unroll
                for (int u = 0; u < 8; u++) xa[u] = (i0 + u) < D / 8 ? xq[2 * (i0 + u)] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u;
                    if (i < D / 8) {
                        const f32x2 xv[2] = {f32x2{xa[u].x, xa[u].y}, f32x2{xa[u].z, xa[u].w}};
#pragma unroll
                        for (int l = 0; l < 2; l++) {
                            if (L2) {
                                const f32x2 t = xv[l] - yr[2 * i + l];
                                acc[l] = __builtin_elementwise_fma(t, t, acc[l]);
                            } else {
                                acc[l] = __builtin_elementwise_fma(xv[l], yr[2 * i + l], acc[l]);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // the chunk's 8 LDS reads back to back ...
                __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);  // ... then its packed math
                __builtin_amdgcn_sched_barrier(0);
            }
            // even thread: lanes 0..3, odd thread: lanes 4..7;  s[l] = acc[l+4] + acc[l]
            float s0, s1, s2, s3;
            add_xor1_x4(acc[0].x, acc[0].y, acc[1].x, acc[1].y, s0, s1, s2, s3);
            float dis = hsum4(s0, s1, s2, s3);
            if (FILTER) {
                if (!valid || !(dis <= max_score && dis >= min_score)) dis = sentinel;
            }
            if (EMIT) {
                if (live && half == 0) {
                    if (L2 ? dis <= tau_q : dis >= tau_q) {
                        const uint32_t kk = f2key(dis);
                        const uint32_t key = L2 ? kk : ~kk;
                        const int slot = atomicAdd(&em.cnt[qt + qi], 1);
                        if (slot < em.cap)
                            em.cand[(int64_t)(qt + qi) * em.cap + slot] =
                                    ((unsigned long long)key << 32) | (unsigned)(row_base + row);
                    }
                }
            } else {
                if (live && half == 0) out[(int64_t)(qt + qi) * ld_out + row] = dis;
            }
        }
    }
}

// generic-d fallback: same contract, row streamed from memory per query.
template <bool L2, bool FILTER>
__global__ __launch_bounds__(256) void k_pairwise_generic(const float* __restrict__ x, int nq, int d,
                                                          const float* __restrict__ y, int64_t ny,
                                                          float* __restrict__ out, int64_t ld_out,
                                                          int q_per_block, FilterDesc filt,
                                                          float min_score, float max_score,
                                                          float sentinel, int64_t row_base) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= ny) return;
    const int q0 = blockIdx.y * q_per_block;
    const int q1 = min(nq, q0 + q_per_block);
    const float* yr = y + row * d;
    bool valid = true;
    if (FILTER) valid = is_valid_doc(filt, row_base + row);
    for (int q = q0; q < q1; q++) {
        float dis = fvec_dist<L2>(x + (int64_t)q * d, yr, d);
        if (FILTER) {
            if (!valid || !(dis <= max_score && dis >= min_score)) dis = sentinel;
        }
        out[(int64_t)q * ld_out + row] = dis;
    }
}

// which shapes the EMIT variant covers (the others keep the distance slab)
bool pairwise_can_emit(int nq, int d, int64_t ny) {
    static const bool no_lds = getenv("GAMMA_HIP_NO_PAIRWISE_LDS") != nullptr;
    static const bool no_emit = getenv("GAMMA_HIP_NO_FLAT_BOUND") != nullptr;
    if (no_lds || no_emit || nq < 8 || ny < 1024) return false;
    return d == 128 || d == 96 || d == 64 || d == 32 || d == 16;
}

// queries per workgroup: enough workgroups to fill 256 CUs a few times over; a workgroup's rows stay
// in registers across its whole query range
static int pairwise_q_per_block(int nq, int64_t row_blocks) {
    int q_per_block = nq;
    const int64_t want_blocks = 2048;
    if (row_blocks < want_blocks) {
        int splits = (int)((want_blocks + row_blocks - 1) / row_blocks);
        q_per_block = (nq + splits - 1) / splits;
        if (q_per_block < 8) q_per_block = nq < 8 ? nq : 8;
    }
    return q_per_block;
}

template <bool L2>
static void launch_pairwise_emit_t(hipStream_t s, const float* x, int nq, int d, const float* y, int64_t ny,
                                   const FilterDesc& filt, float min_score, float max_score, float sentinel,
                                   int64_t row_base, const FlatEmit& em) {
    const int q_per_block = pairwise_q_per_block(nq, (ny + 255) / 256);
    const dim3 grid((unsigned)((ny + 127) / 128), (unsigned)((nq + q_per_block - 1) / q_per_block));
#define GH_EMIT(DD)                                                                                         \
    hipLaunchKernelGGL((k_pairwise_lds<L2, DD, true, true>), grid, dim3(256), 0, s, x, nq, y, ny, nullptr, 0, \
                       q_per_block, filt, min_score, max_score, sentinel, row_base, em)
    switch (d) {
        case 128: GH_EMIT(128); break;
        case 96: GH_EMIT(96); break;
        case 64: GH_EMIT(64); break;
        case 32: GH_EMIT(32); break;
        case 16: GH_EMIT(16); break;
        default: abort();   // pairwise_can_emit
    }
#undef GH_EMIT
}

void launch_pairwise_emit(hipStream_t s, bool l2, const float* x, int nq, int d, const float* y, int64_t ny,
                          const FilterDesc& filt, float min_score, float max_score, int64_t row_base,
                          const FlatEmit& em) {
    if (ny <= 0 || nq <= 0) return;
    if (l2) launch_pairwise_emit_t<true>(s, x, nq, d, y, ny, filt, min_score, max_score, INFINITY, row_base, em);
    else launch_pairwise_emit_t<false>(s, x, nq, d, y, ny, filt, min_score, max_score, -INFINITY, row_base, em);
}

template <bool L2, bool FILTER>
static void launch_pairwise_t(hipStream_t s, const float* x, int nq, int d, const float* y,
                              int64_t ny, float* out, int64_t ld_out, const FilterDesc& filt,
                              float min_score, float max_score, float sentinel,
                              int64_t row_base) {
    if (ny <= 0 || nq <= 0) return;
    const int64_t row_blocks = (ny + 255) / 256;
    const int q_per_block = pairwise_q_per_block(nq, row_blocks);
    dim3 grid((unsigned)row_blocks, (unsigned)((nq + q_per_block - 1) / q_per_block));
    static const bool no_lds = getenv("GAMMA_HIP_NO_PAIRWISE_LDS") != nullptr;
    // a workgroup that sees fewer than a few queries gains nothing from staging them
    const bool use_lds = !no_lds && q_per_block >= 8;
#define GH_ROWREG(DD)                                                                                  \
    do {                                                                                               \
        if (use_lds)                                                                                   \
            hipLaunchKernelGGL((k_pairwise_lds<L2, DD, FILTER>), dim3((unsigned)((ny + 127) / 128), grid.y), \
                               dim3(256), 0, s, x, nq, y, ny, out, ld_out, q_per_block, filt, min_score, \
                               max_score, sentinel, row_base, FlatEmit{});                             \
        else                                                                                           \
            hipLaunchKernelGGL((k_pairwise_rowreg<L2, DD, FILTER>), grid, dim3(256), 0, s, x, nq, y, ny, \
                               out, ld_out, q_per_block, filt, min_score, max_score, sentinel, row_base); \
    } while (0)
    switch (d) {
        case 128: GH_ROWREG(128); break;
        case 96: GH_ROWREG(96); break;
        case 64: GH_ROWREG(64); break;
        case 32: GH_ROWREG(32); break;
        case 16: GH_ROWREG(16); break;
        default:
            hipLaunchKernelGGL((k_pairwise_generic<L2, FILTER>), grid, dim3(256), 0, s, x, nq, d, y,
                               ny, out, ld_out, q_per_block, filt, min_score, max_score, sentinel,
                               row_base);
    }
#undef GH_ROWREG
}

void launch_pairwise(hipStream_t s, bool l2, const float* x, int nq, int d, const float* y,
                     int64_t ny, float* out, int64_t ld_out) {
    FilterDesc f{};
    if (l2)
        launch_pairwise_t<true, false>(s, x, nq, d, y, ny, out, ld_out, f, 0, 0, 0, 0);
    else
        launch_pairwise_t<false, false>(s, x, nq, d, y, ny, out, ld_out, f, 0, 0, 0, 0);
}

// The numeric-column and term clauses of a request, evaluated once per DOCUMENT into one bit each -- the form in
// which range filters arrive (table/range_query_result.h).  A large batch tests a document's clauses once per scored
// code of that document; with the bit in hand the scan reads one bit of an L2-resident map instead of a column value
// (and a term list) at a random address.  Same predicate code (filter_dev.h), so the same answer by construction.
__global__ __launch_bounds__(256) void k_filter_bitmap(FilterDesc f, int64_t nbits, unsigned long long* __restrict__ out) {
    const int64_t doc = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = doc < nbits && is_valid_doc(f, doc);
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && (doc >> 6) < ((nbits + 63) >> 6)) out[doc >> 6] = m;
}
void launch_filter_bitmap(hipStream_t s, const FilterDesc& f, int64_t nbits, uint8_t* out) {
    if (nbits <= 0) return;
    hipLaunchKernelGGL(k_filter_bitmap, dim3((unsigned)((nbits + 255) / 256)), dim3(256), 0, s, f, nbits,
                       reinterpret_cast<unsigned long long*>(out));
}

// Per-call compaction of the inverted lists under the call's validity predicate (large filtered batches): a
// document passes or fails whatever the query, so every list is cut down ONCE to the entries that pass -- same
// region of a shadow arena, same order, new length -- and the batch then runs the UNFILTERED pipeline over the
// shadow lists: no id reads, no predicate and no ADC work for entries that could never be returned (at 10 %
// selectivity nine tenths of the scan).  One workgroup per list; positions by ballot + prefix, stable.
__global__ __launch_bounds__(256) void k_compact_lists(const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
                                                       const uint8_t* __restrict__ codes, const int64_t* __restrict__ ids,
                                                       int code_size, const FilterDesc* __restrict__ ftab,
                                                       uint8_t* __restrict__ out_codes, int64_t* __restrict__ out_ids,
                                                       int* __restrict__ out_len, const float* __restrict__ sums,
                                                       float* __restrict__ out_sums) {
    __shared__ int s_wtot[4];
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t off = list_off[l];
    const int len = list_len[l];
    const FilterDesc& filt = ftab[0];
    int run = 0;
    for (int j0 = 0; j0 < len; j0 += 256) {   // uniform
        const int j = j0 + tid;
        int64_t id = -1;
        bool ok = false;
        if (j < len) {
            id = ids[off + j];
            ok = id >= 0 && is_valid_doc(filt, id);   // bit 63: superseded slot (realtime_mem_data.h:26)
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) s_wtot[w] = __popcll(m);
        __syncthreads();
        int base = run, tot = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int t = s_wtot[i];
            if (i < w) base += t;
            tot += t;
        }
        if (ok) {
            const int64_t dst = off + base + __popcll(m & ((1ull << lane) - 1ull));
            out_ids[dst] = id;
            if (out_sums) out_sums[dst] = sums[off + j];   // the filter pass of the scan reads them beside the codes
            const uint8_t* src = codes + (off + j) * code_size;
            uint8_t* d8 = out_codes + dst * code_size;
            if ((code_size & 15) == 0) {
                for (int b = 0; b < code_size; b += 16) *reinterpret_cast<uint4*>(d8 + b) = *reinterpret_cast<const uint4*>(src + b);
            } else if ((code_size & 7) == 0) {
                for (int b = 0; b < code_size; b += 8) *reinterpret_cast<uint2*>(d8 + b) = *reinterpret_cast<const uint2*>(src + b);
            } else {
                for (int b = 0; b < code_size; b++) d8[b] = src[b];
            }
        }
        run += tot;
        __syncthreads();   // s_wtot is rewritten by the next round
    }
    if (tid == 0) out_len[l] = run;
}
void launch_compact_lists(hipStream_t s, const int64_t* list_off, const int* list_len, int nlist, const uint8_t* codes,
                          const int64_t* ids, int code_size, const FilterDesc* ftab, uint8_t* out_codes, int64_t* out_ids,
                          int* out_len, const float* sums, float* out_sums) {
    if (nlist <= 0) return;
    hipLaunchKernelGGL(k_compact_lists, dim3(nlist), dim3(256), 0, s, list_off, list_len, codes, ids, code_size, ftab,
                       out_codes, out_ids, out_len, sums, out_sums);
}

void launch_pairwise_filtered(hipStream_t s, bool l2, const float* x, int nq, int d,
                              const float* y, int64_t ny, float* out, int64_t ld_out,
                              const FilterDesc& filt, float min_score, float max_score,
                              int64_t row_base) {
    if (l2)
        launch_pairwise_t<true, true>(s, x, nq, d, y, ny, out, ld_out, filt, min_score, max_score,
                                      INFINITY, row_base);
    else
        launch_pairwise_t<false, true>(s, x, nq, d, y, ny, out, ld_out, filt, min_score, max_score,
                                       -INFINITY, row_base);
}

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

// ------------------------------------------------------------------------------------
// a4: per-query inner-product table  st2[q][m][j] = <x_q,m , c_mj>
// (ProductQuantizer::compute_inner_prod_table, faiss:impl/ProductQuantizer.cpp:518-531)
// block = 256 threads = the 256 centroids of one sub-quantizer; grid = (M, nq / IPT_QB).
// ------------------------------------------------------------------------------------
constexpr int IPT_QB = 8;   // queries per workgroup: the centroid row stays in registers
__device__ __forceinline__ void pq_ip_table_body(int m, int q0, const float* __restrict__ x, int nq, int d, int M,
                                                 int dsub, const float* __restrict__ pqc, float* __restrict__ out) {
    const int j = threadIdx.x;
    const float* c = pqc + ((int64_t)m * 256 + j) * dsub;            // per-lane row
#pragma unroll
    for (int u = 0; u < IPT_QB; u++) {
        const int q = q0 + u;
        if (q < nq) {                                                // uniform
            const float* xs = x + (int64_t)q * d + m * dsub;         // wave-uniform
            out[((int64_t)q * M + m) * 256 + j] = fvec_ny_row<false>(xs, c, dsub);
        }
    }
}
__global__ __launch_bounds__(256) void k_pq_ip_table(const float* __restrict__ x, int nq, int d, int M,
                                                     int dsub, const float* __restrict__ pqc,
                                                     float* __restrict__ out) {
    pq_ip_table_body(blockIdx.x, blockIdx.y * IPT_QB, x, nq, d, M, dsub, pqc, out);
}
// Small batches (nq <= 16): the exact coarse distances (k_pairwise_rowreg, one query range) and the queries'
// inner-product tables are independent of each other and each is a dozen workgroups: one launch, roles by block.
__global__ __launch_bounds__(256) void k_small_coarse_ip(const float* __restrict__ x, int nq, int D, const float* __restrict__ cc,
                                                         int nlist, float* __restrict__ mat, int row_blocks, int M,
                                                         const float* __restrict__ pqc, float* __restrict__ st2,
                                                         int* __restrict__ zero_me) {
    if (zero_me && blockIdx.x == 0 && threadIdx.x == 0) *zero_me = 0;   // the next launch's work-list counter
    if ((int)blockIdx.x < row_blocks) {
        // eight threads per centroid = the eight AVX lane accumulators of fvec_L2sqr (rerank_dev.h): 32 centroids per
        // workgroup, coalesced 32-byte pieces, 128 workgroups at nlist 4096 instead of 16 threads-per-row ones
        const int l = threadIdx.x & 7, row = (int)blockIdx.x * 32 + (threadIdx.x >> 3);
        const bool live = row < nlist;
        const float* yr = cc + (int64_t)(live ? row : 0) * D;
        for (int q = 0; q < nq; q++) {
            const float dis = rerank_dist8<true>(x + (int64_t)q * D, yr, D, l, live);
            if (l == 0 && live) mat[(int64_t)q * nlist + row] = dis;
        }
    } else {
        for (int q0 = 0; q0 < nq; q0 += IPT_QB) pq_ip_table_body((int)blockIdx.x - row_blocks, q0, x, nq, D, M, D / M, pqc, st2);
    }
}
bool launch_small_coarse_ip(hipStream_t s, const float* x, int nq, int d, const float* cc, int nlist, float* mat, int M,
                            const float* pqc, float* st2, int* zero_me) {
    if (nq <= 0 || nq > 2 * IPT_QB || M < 0 || (M > 0 && d % M)) return false;   // M = 0 (IVFFLAT): no query tables
    const int rb = (nlist + 31) / 32;
    hipLaunchKernelGGL(k_small_coarse_ip, dim3(rb + M), dim3(256), 0, s, x, nq, d, cc, nlist, mat, rb, M, pqc, st2, zero_me);
    return true;
}
// Large batches: the kernel is all stores (16 KB of table per query).  A thread keeps FOUR consecutive centroids of its
// sub-quantizer in registers and writes their four products as one 16-byte store; a workgroup (4 sub-quantizers x 64
// lanes) covers 4 KB of consecutive table per query, for IPT4_QB queries.  Same fvec_inner_products_ny arithmetic.
constexpr int IPT4_QB = 32;
template <int DSUB>
__global__ __launch_bounds__(256) void k_pq_ip_table4(const float* __restrict__ x, int nq, int d, int M,
                                                      const float* __restrict__ pqc, float* __restrict__ out) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float c[4][DSUB];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int t = 0; t < DSUB; t++) c[r][t] = pqc[((int64_t)m * 256 + 4 * lane + r) * DSUB + t];
    const int q0 = blockIdx.y * IPT4_QB;
    for (int u = 0; u < IPT4_QB; u++) {
        const int q = q0 + u;
        if (q >= nq) break;                                          // uniform
        const float* xs = x + (int64_t)q * d + m * DSUB;             // wave-uniform
        float xv[DSUB];
#pragma unroll
        for (int t = 0; t < DSUB; t++) xv[t] = xs[t];
        float4 o;
        o.x = fvec_ny_row<false>(xv, c[0], DSUB);
        o.y = fvec_ny_row<false>(xv, c[1], DSUB);
        o.z = fvec_ny_row<false>(xv, c[2], DSUB);
        o.w = fvec_ny_row<false>(xv, c[3], DSUB);
        *reinterpret_cast<float4*>(out + ((int64_t)q * M + m) * 256 + 4 * lane) = o;
    }
}
void launch_pq_ip_table(hipStream_t s, const float* x, int nq, int d, int M, const float* pqc,
                        float* out) {
    if (nq <= 0) return;
    static const bool no4 = getenv("GAMMA_HIP_NO_IP_TABLE4") != nullptr;
    const int dsub = d / M;
    if (!no4 && nq >= 256 && (M & 3) == 0 && (dsub == 4 || dsub == 8 || dsub == 12)) {
        const dim3 grid(M / 4, (nq + IPT4_QB - 1) / IPT4_QB);
        if (dsub == 4) hipLaunchKernelGGL(k_pq_ip_table4<4>, grid, dim3(256), 0, s, x, nq, d, M, pqc, out);
        else if (dsub == 8) hipLaunchKernelGGL(k_pq_ip_table4<8>, grid, dim3(256), 0, s, x, nq, d, M, pqc, out);
        else hipLaunchKernelGGL(k_pq_ip_table4<12>, grid, dim3(256), 0, s, x, nq, d, M, pqc, out);
        return;
    }
    hipLaunchKernelGGL(k_pq_ip_table, dim3(M, (nq + IPT_QB - 1) / IPT_QB), dim3(256), 0, s, x, nq, d, M,
                       d / M, pqc, out);
}

// precomputed table T2[l][m][j] = ||c_mj||^2 + 2 <centroid_l,m , c_mj>
// (faiss:IndexIVFPQ.cpp:453-479: r_norms via fvec_norm_L2sqr, fvec_madd with bf = 2)
__global__ __launch_bounds__(256) void k_precompute_table(const float* __restrict__ cc, int d, int M,
                                                          int dsub, const float* __restrict__ pqc,
                                                          float* __restrict__ out) {
    const int m = blockIdx.x, l = blockIdx.y, j = threadIdx.x;
    const float* xs = cc + (int64_t)l * d + m * dsub;
    const float* c = pqc + ((int64_t)m * 256 + j) * dsub;
    float ip = fvec_ny_row<false>(xs, c, dsub);
    float rn = fvec_norm_L2sqr(c, dsub);
    out[((int64_t)l * M + m) * 256 + j] = __builtin_fmaf(2.0f, ip, rn);
}
void launch_precompute_table(hipStream_t s, const float* cc, int nlist, int d, int M,
                             const float* pqc, float* out) {
    hipLaunchKernelGGL(k_precompute_table, dim3(M, nlist), dim3(256), 0, s, cc, d, M, d / M, pqc, out);
}

// ------------------------------------------------------------------------------------
// per-query exclusive prefix of probed-list lengths -> where each (query, probe) pair
// writes its distances.  grid = nq, block = 256.  Also masks lists not owned by this
// shard (length 0) and accumulates the algorithmic scan-byte counter.
// ------------------------------------------------------------------------------------
constexpr int QO_BINS = 4096, QO_BATCH = 8;   // query order (below)
__global__ __launch_bounds__(256) void k_pair_offsets(const int* __restrict__ probe_list, int nq, int P,
                                                      const int* __restrict__ list_len,
                                                      const uint8_t* __restrict__ list_mask,
                                                      int nlist, int* __restrict__ pair_off,
                                                      int* __restrict__ q_total,
                                                      unsigned long long* __restrict__ scan_codes,
                                                      const int64_t* __restrict__ list_off,
                                                      int64_t* __restrict__ pair_base, PairZero z) {
    // one wave per query: the scan is a wave-shuffle prefix sum, no barriers
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq) return;
    // the per-call state the scan / selection / tie flags start from zero, cleared here instead of by four fills
    if (lane == 0) {
        if (z.bytes) z.bytes[q] = 0;
        if (z.words) z.words[q] = 0ull;
        if (q == 0) {
            if (z.count_a) *z.count_a = 0;
            if (z.count_b) *z.count_b = 0;
        }
        // the histogram pass of the grid-wide query order (k_qo_scan / k_qo_scatter below): key of the query's nearest list
        if (z.qo_bins) {
            const int l0 = probe_list[(int64_t)q * P];
            const int r = (l0 >= 0 && l0 < nlist) ? z.qo_rank[l0] : 0;
            const int key = (int)((int64_t)r * QO_BINS / nlist);
            z.qo_key[q] = key;
            atomicAdd(&z.qo_bins[key], 1);
        }
    }
    int running = 0;
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int p = p0 + lane;
        int len = 0;
        int64_t lbase = 0;
        if (p < P) {
            const int l = probe_list[(int64_t)q * P + p];
            if (l >= 0 && l < nlist && (!list_mask || list_mask[l])) {
                len = list_len[l];
                lbase = list_off[l];
            }
        }
        const int incl = wave_incl_scan(len);
        if (p < P) {
            pair_off[(int64_t)q * (P + 1) + p] = running + incl - len;
            if (pair_base) pair_base[(int64_t)q * P + p] = lbase;   // arena offset of the pair's list
        }
        running += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
        pair_off[(int64_t)q * (P + 1) + P] = running;
        q_total[q] = running;
    }
}

// Sharded search: a shard owns ~1/W of a query's probed lists.  Move the owned, non-empty ones to
// the front of the query's probe list (stable, coarse distances move along) so that the scan's
// probe groups are dense again and the first group can bound the local top-recall_num.
// One wave per query, 64 probes per pass.  Entries behind the owned ones become -1.
__global__ __launch_bounds__(256) void k_compact_probes(const int* __restrict__ probe_in,
                                                        const float* __restrict__ cdis_in, int nq, int P,
                                                        const int* __restrict__ list_len,
                                                        const uint8_t* __restrict__ list_mask, int nlist,
                                                        int* __restrict__ probe_out,
                                                        float* __restrict__ cdis_out) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq) return;
    int nown = 0;
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int p = p0 + lane;
        int l = -1;
        float cd = 0.f;
        if (p < P) {
            l = probe_in[(int64_t)q * P + p];
            cd = cdis_in[(int64_t)q * P + p];
        }
        const bool own = p < P && l >= 0 && l < nlist && (!list_mask || list_mask[l]) && list_len[l] > 0;
        const unsigned long long bal = __ballot(own);
        if (own) {
            const int at = nown + __popcll(bal & ((1ull << lane) - 1ull));
            probe_out[(int64_t)q * P + at] = l;
            cdis_out[(int64_t)q * P + at] = cd;
        }
        nown += __popcll(bal);
    }
    for (int p = nown + lane; p < P; p += 64) {
        probe_out[(int64_t)q * P + p] = -1;
        cdis_out[(int64_t)q * P + p] = 0.f;
    }
}
void launch_compact_probes(hipStream_t s, const int* probe_in, const float* cdis_in, int nq, int P,
                           const int* list_len, const uint8_t* list_mask, int nlist, int* probe_out,
                           float* cdis_out) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_compact_probes, dim3((nq + 3) / 4), dim3(256), 0, s, probe_in, cdis_in, nq, P, list_len,
                       list_mask, nlist, probe_out, cdis_out);
}

// profiling only: algorithmic scan volume of a batch = sum of the per-query candidate counts
__global__ __launch_bounds__(256) void k_sum_totals(const int* __restrict__ q_total, int nq,
                                                    unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long s_part[4];
    unsigned long long t = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) t += (unsigned long long)q_total[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}
void launch_pair_offsets(hipStream_t s, const int* probe_list, int nq, int P, const int* list_len,
                         const uint8_t* list_mask, int nlist, int* pair_off, int* q_total,
                         unsigned long long* scan_codes, const int64_t* list_off, int64_t* pair_base,
                         const PairZero* zero) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_pair_offsets, dim3((nq + 3) / 4), dim3(256), 0, s, probe_list, nq, P, list_len,
                       list_mask, nlist, pair_off, q_total, scan_codes, list_off, pair_base, zero ? *zero : PairZero());
    if (scan_codes)
        hipLaunchKernelGGL(k_sum_totals, dim3(std::min(64, (nq + 255) / 256)), dim3(256), 0, s, q_total, nq,
                           scan_codes);
}
void launch_sum_totals(hipStream_t s, const int* q_total, int nq, unsigned long long* acc) {
    if (nq > 0 && acc)
        hipLaunchKernelGGL(k_sum_totals, dim3(std::min(64, (nq + 255) / 256)), dim3(256), 0, s, q_total, nq, acc);
}

// ------------------------------------------------------------------------------------
// Query order for the scan (speed only, results do not depend on it): counting sort of the
// queries by list_rank[nearest list] scaled to QO_BINS bins.  list_rank is a spatial order
// of the coarse centroids (recursive principal-axis bisection, host side).  One workgroup;
// keys are fetched QO_BATCH at a time so the strided loads overlap.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_query_order(const int* __restrict__ probe_list, int nq, int P,
                                                      const int* __restrict__ list_rank, int nlist,
                                                      int* __restrict__ qkey, int* __restrict__ qperm) {
    __shared__ int s_bin[QO_BINS];
    __shared__ int s_part[32];
    const int tid = threadIdx.x;
    for (int b = tid; b < QO_BINS; b += 1024) s_bin[b] = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024 * QO_BATCH) {
        int l[QO_BATCH], r[QO_BATCH];
#pragma unroll
        for (int u = 0; u < QO_BATCH; u++) {
            const int q = q0 + u * 1024 + tid;
            l[u] = q < nq ? probe_list[(int64_t)q * P] : -1;
        }
#pragma unroll
        for (int u = 0; u < QO_BATCH; u++) r[u] = (l[u] >= 0 && l[u] < nlist) ? list_rank[l[u]] : 0;
#pragma unroll
        for (int u = 0; u < QO_BATCH; u++) {
            const int q = q0 + u * 1024 + tid;
            if (q < nq) {
                const int key = (int)((int64_t)r[u] * QO_BINS / nlist);
                qkey[q] = key;
                atomicAdd(&s_bin[key], 1);
            }
        }
    }
    __syncthreads();
    // exclusive scan of the bins: QO_BINS / 1024 bins per thread
    int v[QO_BINS / 1024], sum = 0;
#pragma unroll
    for (int u = 0; u < QO_BINS / 1024; u++) {
        v[u] = s_bin[tid * (QO_BINS / 1024) + u];
        sum += v[u];
    }
    // block-wide exclusive prefix of `sum`: wave scans + one scan of the 16 wave totals
    const int incl = wave_incl_scan(sum);
    if ((tid & 63) == 63) s_part[tid >> 6] = incl;
    __syncthreads();
    if (tid < 64) {
        const int t = tid < 16 ? s_part[tid] : 0;
        const int ti = wave_incl_scan(t);
        if (tid < 16) s_part[16 + tid] = ti - t;   // exclusive prefix of the wave totals
    }
    __syncthreads();
    int run = s_part[16 + (tid >> 6)] + incl - sum;
#pragma unroll
    for (int u = 0; u < QO_BINS / 1024; u++) {
        s_bin[tid * (QO_BINS / 1024) + u] = run;
        run += v[u];
    }
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024 * QO_BATCH) {
        int key[QO_BATCH];
#pragma unroll
        for (int u = 0; u < QO_BATCH; u++) {
            const int q = q0 + u * 1024 + tid;
            key[u] = q < nq ? qkey[q] : 0;
        }
#pragma unroll
        for (int u = 0; u < QO_BATCH; u++) {
            const int q = q0 + u * 1024 + tid;
            if (q < nq) qperm[atomicAdd(&s_bin[key[u]], 1)] = q;
        }
    }
}
// large batches (sharded search: W x 8192 queries): the same counting sort over the whole grid, bins in
// global memory -- key + histogram, scan of the QO_BINS bins, scatter.  The order inside a bin is whatever
// the atomics give; the order only steers scheduling.
__global__ __launch_bounds__(256) void k_qo_hist(const int* __restrict__ probe_list, int nq, int P,
                                                 const int* __restrict__ list_rank, int nlist,
                                                 int* __restrict__ qkey, int* __restrict__ bins) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int l = probe_list[(int64_t)q * P];
    const int r = (l >= 0 && l < nlist) ? list_rank[l] : 0;
    const int key = (int)((int64_t)r * QO_BINS / nlist);
    qkey[q] = key;
    atomicAdd(&bins[key], 1);
}
// bins -> cursor (exclusive prefix); the bins are left zero for the next call's histogram
__global__ __launch_bounds__(1024) void k_qo_scan(int* __restrict__ bins, int* __restrict__ cursor) {
    __shared__ int s_part[32];
    const int tid = threadIdx.x;
    int v[QO_BINS / 1024], sum = 0;
#pragma unroll
    for (int u = 0; u < QO_BINS / 1024; u++) {
        v[u] = bins[tid * (QO_BINS / 1024) + u];
        bins[tid * (QO_BINS / 1024) + u] = 0;
        sum += v[u];
    }
    const int incl = wave_incl_scan(sum);
    if ((tid & 63) == 63) s_part[tid >> 6] = incl;
    __syncthreads();
    if (tid < 64) {
        const int t = tid < 16 ? s_part[tid] : 0;
        const int ti = wave_incl_scan(t);
        if (tid < 16) s_part[16 + tid] = ti - t;
    }
    __syncthreads();
    int run = s_part[16 + (tid >> 6)] + incl - sum;
#pragma unroll
    for (int u = 0; u < QO_BINS / 1024; u++) {
        cursor[tid * (QO_BINS / 1024) + u] = run;
        run += v[u];
    }
}
__global__ __launch_bounds__(256) void k_qo_scatter(const int* __restrict__ qkey, int nq, int* __restrict__ bins,
                                                    int* __restrict__ qperm) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < nq) qperm[atomicAdd(&bins[qkey[q]], 1)] = q;
}
int query_order_bins() { return QO_BINS; }
bool query_order_grid(int nq) { return nq > 8192; }   // one workgroup takes 20 us for 8192 queries and grows linearly
// bins: 2 * QO_BINS ints (histogram | cursors), the histogram zero on entry and left zero; hist_done: the pair-offset
// kernel has filled qkey and the histogram (PairZero::qo_*)
void launch_query_order(hipStream_t s, const int* probe_list, int nq, int P, const int* list_rank,
                        int nlist, int* qkey, int* qperm, int* bins, bool hist_done) {
    if (nq <= 0) return;
    if (query_order_grid(nq) && bins) {
        if (!hist_done)
            hipLaunchKernelGGL(k_qo_hist, dim3((nq + 255) / 256), dim3(256), 0, s, probe_list, nq, P, list_rank, nlist, qkey, bins);
        hipLaunchKernelGGL(k_qo_scan, dim3(1), dim3(1024), 0, s, bins, bins + QO_BINS);
        hipLaunchKernelGGL(k_qo_scatter, dim3((nq + 255) / 256), dim3(256), 0, s, qkey, nq, bins + QO_BINS, qperm);
        return;
    }
    hipLaunchKernelGGL(k_query_order, dim3(1), dim3(1024), 0, s, probe_list, nq, P, list_rank, nlist, qkey,
                       qperm);
}

// ------------------------------------------------------------------------------------
// dis0 of the inner-product scan: <x_q, centroid_l> for every (query, probe) pair, in
// fvec_inner_product order (gamma_index_ivfpq.h:216-230; device_math.h fvec_dist<false>): eight
// threads per pair play the eight AVX lanes, each a k-ascending fma chain over its elements,
// then s[j] = acc[j+4] + acc[j], the 4-lane and masked tails, (s0+s1)+(s2+s3).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pair_ip(const float* __restrict__ x, const float* __restrict__ cc,
                                                 const int* __restrict__ probe_list, int64_t npairs, int P,
                                                 int d, int nlist, float* __restrict__ out) {
    const int64_t pair = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int l8 = threadIdx.x & 7, l4 = l8 & 3;
    if (pair >= npairs) return;   // groups of 8 lanes leave together
    const int l = probe_list[pair];
    float res = 0.f;
    if (l >= 0 && l < nlist) {    // uniform inside the group
        const float* xq = x + (pair / P) * d;
        const float* c = cc + (int64_t)l * d;
        const int nblk = d >> 3;
        float a = 0.f;
        int b = 0;
        for (; b + 8 <= nblk; b += 8) {   // 16 loads in flight, then the chain
            float xv[8], cv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                xv[u] = xq[(b + u) * 8 + l8];
                cv[u] = c[(b + u) * 8 + l8];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) a = __builtin_fmaf(xv[u], cv[u], a);
        }
        for (; b < nblk; b++) a = __builtin_fmaf(xq[b * 8 + l8], c[b * 8 + l8], a);
        const int base = (threadIdx.x & 63) & ~7;
        float sv = __shfl(a, base + l4 + 4, 64) + __shfl(a, base + l4, 64);   // s[l4], on lanes l4 and l4 + 4
        int i0 = nblk * 8, rem = d - i0;
        if (rem >= 4) {
            sv = __builtin_fmaf(xq[i0 + l4], c[i0 + l4], sv);
            i0 += 4;
            rem -= 4;
        }
        if (l4 < rem) sv = __builtin_fmaf(xq[i0 + l4], c[i0 + l4], sv);   // rem <= 3
        const float s0 = __shfl(sv, base, 64), s1 = __shfl(sv, base + 1, 64), s2 = __shfl(sv, base + 2, 64),
                    s3 = __shfl(sv, base + 3, 64);
        res = hsum4(s0, s1, s2, s3);
    }
    if (l8 == 0) out[pair] = res;
}
void launch_pair_ip(hipStream_t s, const float* x, const float* cc, const int* probe_list, int nq, int P, int d,
                    int nlist, float* out) {
    const int64_t npairs = (int64_t)nq * P;
    if (npairs <= 0) return;
    hipLaunchKernelGGL(k_pair_ip, dim3((unsigned)((npairs + 31) / 32)), dim3(256), 0, s, x, cc, probe_list, npairs,
                       P, d, nlist, out);
}

// ------------------------------------------------------------------------------------
// a4+a5+a6+a8: IVFPQ list scan, one workgroup per (query, probe) pair.
//   LUT (M x 256 fp32) built in LDS:  L2: lut = T2[list] + (-2) * st2[q]  (fvec_madd)
//                                     IP: lut = st2[q]
//   dis0: L2 = coarse distance; IP = <x_q, centroid> in fvec_inner_product order.
//   per code j:  skip if ids[j] bit 63 / !IsValid;  dis = dis0; for m: dis += lut[m][code[m]]
//   (sequential fp32 adds, gamma_index_ivfpq.h:591-597).  Distances go to the pair's slot
//   range in out; filtered entries get the sentinel.
// Codes are AoS [len][M] exactly as the reference stores them; a 16-byte code is one
// dwordx4 load per lane, so a wave reads 1 KiB contiguous.
// ------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int SCAN_STAGE = 256;                  // survivors staged in LDS per workgroup
constexpr int SCAN_SLICE = 1024;                 // candidate slice of one workgroup (global); a producer needs recall_num + one histogram bin
constexpr int SCAN_BATCH = 64;                   // queries per XCD by which producers run ahead

// amdgpu_num_sgpr(96): 8 waves per SIMD need <= 96 SGPRs each (800 per SIMD); the FILT variant
// would otherwise take 100 and lose one of the eight resident workgroups per CU
// IPF (sharded search with every probe of a query in ONE workgroup): the query's table <x_q,m , c_mj> is
// computed here from the PQ codebook (128 KB, L2 resident; `st2` then points at it) instead of being
// written to HBM by k_pq_ip_table and read back -- with W shards that table is W x 16 KB per query of
// traffic that does not shrink with the shard, and each of its entries would be read exactly once.
// CF (L2, FILT, MT 16 / 32, large batches): the consumer groups of a query with a bound run a FILTER pass without the
// per-list table -- see "filter pass" in the body.
constexpr int SCAN_CF_CAP = 768;   // filter-pass candidates staged per workgroup (8 bytes each)
template <bool L2, int MT, bool FILT, bool IPF = false, bool UNITS = false, bool CF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void k_ivfpq_scan_pair(
        const float* __restrict__ x, int nq, int d, int M, int P, int G, const int* __restrict__ probe_list,
        const float* __restrict__ coarse_dis, const float* __restrict__ cc,
        const float* __restrict__ st2, const float* __restrict__ T2,
        const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
        const uint8_t* __restrict__ list_mask, int nlist, const uint8_t* __restrict__ codes,
        const int64_t* __restrict__ ids, const int* __restrict__ pair_off, int64_t q_stride,
        float* __restrict__ out, const FilterDesc* __restrict__ ftab, const int* __restrict__ qfil, int need_ids,
        float sentinel, const int* __restrict__ qperm, int pg_lo, int pg_cnt, int sparse, ScanBound sb,
        const int* __restrict__ rq_list, const int* __restrict__ rq_count, int chunk_len) {
    // UNITS (small batches over long lists, G = 1, no bound; a variant of its own so that the bulk kernel carries
    // none of its state): rq_list is a work list of
    // (query << 20 | probe << 13 | chunk) units written by k_small_coarse_select, one per chunk_len codes of a
    // (query, probe) pair, walked by a fixed grid -- one query's 64 long lists are then a few hundred pieces of
    // even size instead of 64 workgroups that run for as long as the longest list takes.
    // This launch covers probe groups [pg_lo, pg_lo + pg_cnt) of every query.
    // FILT (pg_lo = 0, pg_cnt >= 2): threshold pre-filter.  The workgroup of a query's FIRST probe
    // group (its nearest lists) ends by bounding the query's recall_num-th best distance from above
    // with the candidates it has just scored, and publishes that bound; the workgroups of the other
    // groups append every candidate within the bound, as a (key, position) item, to the query's
    // short survivor list.  The exact top-recall_num is then selected from the first group's
    // candidates within the bound plus a few hundred survivors, instead of all ~10^4 candidates
    // (select.hip, k_select_final).
    // One workgroup scans G consecutive probes of one query: the query's 16 KB table st2 is
    // read ONCE into registers (MT per thread) and reused for the G list-specific LUTs, so
    // the per-pair table traffic drops from 2 x M KB to (1 + 1/G) x M KB.
    extern __shared__ float s_lut[];  // M*256
    // XCD-aware placement (speed only): block b runs on XCD b % 8 with its own L2, so all PGN
    // workgroups of one query are given block ids with the same residue -- the query's table
    // st2[q] is then fetched from HBM/MALL once per XCD and served from that L2 afterwards.
    // With qperm (queries sorted by the spatial rank of their nearest list, k_query_order) XCD x
    // takes the x-th contiguous eighth of that order, in order: concurrently running queries
    // probe overlapping lists, so the 16 KB T2 rows they stream are mostly L2 hits as well.
    const int xcd = blockIdx.x & 7;
    int slot = blockIdx.x >> 3, pg, qslot;
    if (FILT) {
        // Block order inside an XCD (slot = XCD-local index): producers (group 0, they publish
        // the bounds) run one batch of SCAN_BATCH queries AHEAD of the consumers (other groups):
        //   P(0) | P(1) C(0) | P(2) C(1) | ...
        // Workgroups are dispatched in id order, so a consumer only ever waits for a producer that
        // is already resident or finished -- and it starts >= SCAN_BATCH * pg_cnt dispatches after
        // its producer, by when the bound is normally there.  The query's table st2[q] is still in
        // this XCD's L2 when its consumers arrive.
        const int nq8 = (nq + 7) >> 3;
        if (slot < SCAN_BATCH) {
            pg = 0;
            qslot = slot;
        } else {
            const int s2 = slot - SCAN_BATCH, period = SCAN_BATCH * pg_cnt;
            const int t = s2 / period, r = s2 % period;
            if (r < SCAN_BATCH) {
                pg = 0;
                qslot = (t + 1) * SCAN_BATCH + r;
            } else {
                const int i = r - SCAN_BATCH;
                pg = 1 + i % (pg_cnt - 1);
                qslot = t * SCAN_BATCH + i / (pg_cnt - 1);
            }
        }
        if (qslot >= nq8) return;
    } else {
        pg = pg_lo + slot % pg_cnt;
        qslot = slot / pg_cnt;
    }
    const bool repair = !FILT && rq_list != nullptr;
    int q = 0;
    if (repair) {
        // repair launch (launch_ivfpq_scan_repair): a fixed grid walks the (query, probe group) items of the
        // queries k_select_final could not finish from their survivor slices -- consumer groups with a bound
        // do not store distances (finish() below) -- and scores those groups again, storing everything
    } else if (qperm) {
        const int qi = xcd * ((nq + 7) >> 3) + qslot;
        if (qi >= nq) return;
        q = qperm[qi];
    } else {
        q = qslot * 8 + xcd;
        if (q >= nq) return;
    }
    // ALL of the kernel's LDS is the dynamic buffer, the LUT first: its LDS address is then the constant 0 and a
    // gather address is just (code byte << 2) + an immediate offset (one VALU op per look-up instead of two)
    unsigned long long* s_stage = reinterpret_cast<unsigned long long*>(s_lut + M * 256);   // [SCAN_STAGE]
    int& s_nstage = *reinterpret_cast<int*>(s_stage + SCAN_STAGE);
    uint32_t& s_tau = *(reinterpret_cast<uint32_t*>(s_stage + SCAN_STAGE) + 1);
    uint32_t* s_red = reinterpret_cast<uint32_t*>(s_stage + SCAN_STAGE) + 2;                  // [12]
    int& s_ncand = *(reinterpret_cast<int*>(s_stage + SCAN_STAGE) + 14);                       // CF: staged candidates
    uint2* s_cand = reinterpret_cast<uint2*>(reinterpret_cast<int*>(s_stage + SCAN_STAGE) + 16);  // CF: [SCAN_CF_CAP]
    int cbase = 0;     // unit mode: first code of the unit within its list
    int lut_q = -1;    // unit mode, inner product: the query whose table is in LDS
    int lut_pair = -1; // unit mode, L2: the (query, probe) pair whose table is in LDS
    auto body = [&](const int q, const int pg) {
    // validity predicates of THIS query: entry qfil[q] of the call's filter table (one entry unless the
    // call is a combined batch of requests with their own filters); only read when need_ids
    const FilterDesc& filt = ftab[(need_ids && qfil) ? qfil[q] : 0];
    const int lane = threadIdx.x & 63;
    // Survivors are staged in LDS (one LDS atomic per wave and iteration) and flushed to the
    // query's list with ONE global atomic per workgroup; a returning global atomic per wave
    // iteration would put ~1 us of latency into the scan loop.  All lanes of a wave call append().
    uint32_t tauq = 0xffffffffu;
    float tau_f = sentinel;   // the bound as a distance: a candidate survives iff it is not worse than tau_f
    bool bound_on = false;
    // producer (pg == 0): range and count of its valid distances.  Kept as floats (one min, one max per code
    // instead of a key conversion and two compare-selects), turned into keys once at the end.
    float g_fmn = INFINITY, g_fmx = -INFINITY;
    int g_nv = 0;
    auto within = [&](float val) -> bool { return L2 ? val <= tau_f : val >= tau_f; };
    auto append = [&](bool keep, float val, int pos) {
        const unsigned long long bal = __ballot(keep);
        if (bal) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_nstage, __popcll(bal));
            base = __shfl(base, 0, 64);
            if (keep) {
                const int at = base + __popcll(bal & ((1ull << lane) - 1ull));
                const unsigned long long item = ((unsigned long long)dis_key<L2>(val) << 32) | (unsigned)pos;
                if (at < SCAN_STAGE) s_stage[at] = item;
                else if (at < SCAN_SLICE)   // staging full (rare): the slot number is already unique
                    sb.surv[((int64_t)q * pg_cnt + pg) * SCAN_SLICE + at] = item;
            }
        }
    };
    // every consumer workgroup owns one fixed slice of its query's survivor list: no global
    // atomics, the count (> SCAN_SLICE = overflowed) is a plain store
    auto flush = [&]() {   // whole workgroup
        __syncthreads();
        const int n = s_nstage;
        const int64_t slice = (int64_t)q * pg_cnt + pg;
        if (threadIdx.x == 0) sb.gcnt[(int64_t)q * sb.cnt_stride + pg] = n;
        for (int i = threadIdx.x; i < min(n, SCAN_STAGE); i += 256) sb.surv[slice * SCAN_SLICE + i] = s_stage[i];
    };
    // (CF: the LAST group takes every probe behind the ones before it -- its table is the query's, not a list's, so
    //  one workgroup per query serves all consumer probes: one table write, one slice)
    // (long lists: several consumer groups of sb.cf_span probes each, so that no group's candidates outgrow its stage)
    const int cfs = CF ? sb.cf_span : 0;
    const int p_begin = (CF && cfs > 0 && pg > 0) ? G + (pg - 1) * cfs : pg * G;
    const int p_end = CF ? (pg == 0 ? min(P, G) : (cfs > 0 ? min(P, p_begin + cfs) : P)) : min(P, p_begin + G);
    const int tid = threadIdx.x;
    const int msz = M * 256;
    // LDS byte address of this wave's 256-byte segment of a LUT row (lut_store)
    const uint32_t lut_m0 = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)s_lut + 256u * (uint32_t)(tid >> 6));
    // nothing to scan in this group (a shard owns ~1/W of the probed lists): leave before the
    // 16 KB query table is fetched (sparse = sharded search only: the check costs two dependent
    // scalar loads per probe).  Producers always go on: they must publish.
    if (sparse && (!FILT || pg > 0)) {
        bool any = false;
        for (int p = p_begin; p < p_end; p++) {
            const int l = probe_list[q * P + p];
            if (l >= 0 && l < nlist && (!list_mask || list_mask[l]) && list_len[l] > 0) any = true;
        }
        if (!any) {   // uniform
            if (FILT && threadIdx.x == 0) sb.gcnt[(int64_t)q * sb.cnt_stride + pg] = 0;
            return;
        }
    }
    const float* st2q = st2 + (int64_t)q * msz;
    float s2r[MT > 0 ? MT : 1];
    if (IPF && MT > 0) {
        // same arithmetic as k_pq_ip_table: one fvec_inner_products_ny row per (m, code word)
        const int dsub = d / M;
        const float* xq = x + (int64_t)q * d;
#pragma unroll
        for (int i = 0; i < MT; i++) s2r[i] = fvec_ny_row<false>(xq + i * dsub, st2 + ((int64_t)i * 256 + tid) * dsub, dsub);
    } else if (MT > 0 && (!UNITS || (L2 ? q * P + pg != lut_pair : q != lut_q))) {
#pragma unroll
        for (int i = 0; i < MT; i++) s2r[i] = st2q[tid + 256 * i];
    }
    if (!L2 && (!UNITS || q != lut_q)) {   // inner product: the LUT is the query table itself, list independent
        if (UNITS) lut_q = q;
        if (MT > 0) {
            lut_store_begin(lut_m0);
            lut_store_rows<MT>([&](int i) { return s2r[i]; }, std::make_integer_sequence<int, (MT > 0 ? MT : 1)>{});
            lut_store_done();
        } else {
            for (int e = tid; e < msz; e += 256) s_lut[e] = st2q[e];
        }
    }
    if constexpr (CF) {
        if (pg > 0) {   // (uniform) largest |entry| of the query's table: one word per wave, read behind the barrier below
            float mxv = 0.f;
#pragma unroll
            for (int i = 0; i < MT; i++) mxv = fmaxf(mxv, fabsf(s2r[i]));
            const uint32_t wmx = __reduce_max_sync(~0ull, __float_as_uint(mxv));   // non-negative floats order as integers
            if (lane == 0) s_red[tid >> 6] = wmx;
        }
    }
    if (FILT) {   // placed after the table loads were issued: their latency and this one overlap
        if (threadIdx.x == 0) {
            s_nstage = 0;
            if (CF) s_ncand = 0;
            if (pg > 0) {   // wait for this query's bound (published by its group-0 workgroup)
                // ONE relaxed 64-bit word carries (state << 32 | bound): no acquire/release fence is
                // needed (nothing else the producer wrote is read here), and agent-scope fences
                // would write back / invalidate the L2 this kernel lives on
                int spins = 0;
                unsigned long long word;
                // (bounded: dispatch order is not a contract -- if the producer has not published within ~2e6
                //  cycles the group goes on without a bound and the query takes the unfiltered selection)
                while ((word = __hip_atomic_load(&sb.ready[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull &&
                       ++spins < (1 << 11))
                    __builtin_amdgcn_s_sleep(16);
                if (word != 0ull) {
                    s_tau = (word >> 32) == 1ull ? (uint32_t)word : 0xffffffffu;
                } else {   // never expected: give the query to the unfiltered selection instead of hanging
                    s_tau = 0xffffffffu;
                    s_nstage = SCAN_SLICE + 1;
                }
            }
        }
        __syncthreads();
        if (pg > 0) {
            tauq = s_tau;
            bound_on = tauq < KEY_SENTINEL;   // otherwise the query takes the unfiltered selection
            tau_f = key2f(L2 ? tauq : ~tauq);
        }
    }
    if constexpr (CF) {
        if (pg > 0 && bound_on) {   // (uniform)
            // ---- filter pass (L2 consumers with a bound) ----------------------------------------------------------
            // Half of the regular loop's instructions build the per-list table T2[l] - 2 ip[q] (4096 entries for
            // lists of a few hundred codes).  Here the LUT is the QUERY's table ip[q] alone, written once per
            // workgroup -- no per-list build, no barriers in the probe loop -- and a code is tested on
            //     f = (dis0 + s_j) - 2 sum_m ip[q][m][c_m],     s_j = sum_m T2[l][m][c_m]  (kept beside the code, 4 bytes),
            // which differs from the reference's value  dis0 + sum_m fma(-2, ip, T2)  (sequential) only by rounding:
            // every one of the < 50 roundings of either evaluation is at most 2^-24 times a partial sum, and every
            // partial sum is bounded by S = |dis0| + sum_m max_c |T2[l][m][c]| + 2 sum_m max_c |ip[q][m][c]|, so
            // |f - exact| <= 50 * 2^-24 * S.  A code passes when f <= tau + 2^-17 S (a margin 2.5 times that, the
            // second term of S taken as 32 times the largest |entry| of the query's table).  The few that pass
            // (about as many as end up in the slice) get the EXACT value afterwards -- table entries fetched from
            // the L2-resident T2 row, fma and adds in the reference's order -- and the slice receives what the
            // regular loop would have put there: same keys, same positions.
            lut_store_begin(lut_m0);
            lut_store_rows<MT>([&](int i) { return s2r[i]; }, std::make_integer_sequence<int, (MT > 0 ? MT : 1)>{});
            lut_store_done();
            const float qmax = __uint_as_float(max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
            // The table does not depend on the list, so nothing in this loop needs the workgroup in step: every WAVE
            // takes whole lists of the group (next one from a counter in LDS), 64 codes per step, the next step's codes
            // and sums requested before the current step's gathers -- four independent latency chains per workgroup
            // instead of one, and no barrier until the candidates are complete.
            int& s_next = *reinterpret_cast<int*>(s_cand + SCAN_CF_CAP);
            if (tid == 0) s_next = 0;
            __syncthreads();   // the LUT and the list counter are in place
            const int ng = p_end - p_begin;
            for (;;) {
                int r = 0;
                if (lane == 0) r = atomicAdd(&s_next, 1);
                r = __builtin_amdgcn_readfirstlane(r);
                if (r >= ng) break;
                const int p = p_begin + r, pair = q * P + p;
                const int l = probe_list[pair];
                if (l < 0 || l >= nlist) continue;            // uniform per wave
                if (list_mask && !list_mask[l]) continue;
                const int len = list_len[l];
                if (len <= 0) continue;
                const int64_t off = list_off[l];
                const uint8_t* lc = codes + off * MT;
                const float* ls = sb.sums + off;
                const int64_t* lid = ids + off;
                const float dis0 = coarse_dis[pair];
                const int pbase = pair_off[(int64_t)q * (P + 1) + p];
                const float S = fabsf(dis0) + sb.t2max[l] + 32.f * qmax;
                float thr = __builtin_fmaf(S, 1.f / 131072.f, tau_f);
                thr += fabsf(thr) * 1.2e-7f;   // the threshold's own rounding
                uint4 cn[MT / 16];
                float sn;
                {
                    const int jc = min(lane, len - 1);
                    const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)jc * MT);
#pragma unroll
                    for (int u = 0; u < MT / 16; u++) cn[u] = cp[u];
                    sn = ls[jc];
                }
                for (int j0 = 0; j0 < len; j0 += 64) {
                    const int j = j0 + lane;
                    uint32_t cw[MT / 4];
#pragma unroll
                    for (int u = 0; u < MT / 16; u++) {
                        cw[4 * u] = cn[u].x; cw[4 * u + 1] = cn[u].y; cw[4 * u + 2] = cn[u].z; cw[4 * u + 3] = cn[u].w;
                    }
                    const float sj = sn;
                    if (j0 + 64 < len) {   // (uniform) the next step's codes and sums, in flight during this step's gathers
                        const int jc = min(j + 64, len - 1);
                        const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)jc * MT);
#pragma unroll
                        for (int u = 0; u < MT / 16; u++) cn[u] = cp[u];
                        sn = ls[jc];
                    }
                    bool ok = j < len;
                    if (need_ids) {
                        const int64_t id = lid[min(j, len - 1)];
                        ok = ok && id >= 0;
                        if (ok) ok = is_valid_doc(filt, id);
                    }
                    float t[MT];
#pragma unroll
                    for (int m = 0; m < MT; m++) t[m] = lut_gather(cw[m >> 2], m & 3, m);
                    __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the adds
                    float g4[4] = {t[0], t[1], t[2], t[3]};   // four independent chains: the order is free here
#pragma unroll
                    for (int m = 4; m < MT; m++) g4[m & 3] += t[m];
                    const float g = (g4[0] + g4[1]) + (g4[2] + g4[3]);
                    const float f = __builtin_fmaf(-2.f, g, dis0 + sj);
                    const bool cand = ok && f <= thr;
                    const unsigned long long bal = __ballot(cand);
                    if (bal) {   // uniform per wave
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_ncand, __popcll(bal));
                        base = __shfl(base, 0, 64);
                        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                        if (cand && slot < SCAN_CF_CAP) s_cand[slot] = make_uint2((uint32_t)(pbase + j), (uint32_t)p);
                    }
                }
            }
            __syncthreads();
            const int nc = s_ncand;
            if (nc > SCAN_CF_CAP) {   // (uniform) more candidates than the stage holds: the query takes the unfiltered path
                if (tid == 0) s_nstage = SCAN_SLICE + 1;
            } else {
                for (int c0 = 0; c0 < nc; c0 += 256) {   // uniform trip count: append() ballots
                    const int c = c0 + tid;
                    bool keep = false;
                    float dis = 0.f;
                    int pos = 0;
                    if (c < nc) {
                        const uint2 cd = s_cand[c];
                        pos = (int)cd.x;
                        const int p = (int)cd.y, pair = q * P + p;
                        const int l = probe_list[pair];
                        const int j = pos - pair_off[(int64_t)q * (P + 1) + p];
                        const uint8_t* cj = codes + (list_off[l] + j) * MT;
                        const float* t2 = T2 + (int64_t)l * msz;
                        uint32_t cw[MT / 4];
#pragma unroll
                        for (int u = 0; u < MT / 16; u++) {
                            const uint4 cv = reinterpret_cast<const uint4*>(cj)[u];
                            cw[4 * u] = cv.x; cw[4 * u + 1] = cv.y; cw[4 * u + 2] = cv.z; cw[4 * u + 3] = cv.w;
                        }
                        dis = coarse_dis[pair];
#pragma unroll
                        for (int m0 = 0; m0 < MT; m0 += 8) {   // eight table entries in flight at a time
                            float a[8];
#pragma unroll
                            for (int m = 0; m < 8; m++) a[m] = t2[(m0 + m) * 256 + ((cw[(m0 + m) >> 2] >> (8 * (m & 3))) & 255u)];
#pragma unroll
                            for (int m = 0; m < 8; m++)   // the regular loop's table entry and its adds, in the reference's order
                                dis += __builtin_fmaf(-2.0f, s_lut[(m0 + m) * 256 + ((cw[(m0 + m) >> 2] >> (8 * (m & 3))) & 255u)], a[m]);
                        }
                        keep = dis <= tau_f;
                    }
                    append(keep, dis, pos);
                }
            }
            flush();
            return;
        }
    }
    if (!L2) __syncthreads();   // the LUT (written once per query) is complete; L2 rebuilds it per list
    for (int p = p_begin; p < p_end; p++) {
        const int pair = q * P + p;
        const int l = probe_list[pair];
        if (l < 0 || l >= nlist) continue;            // uniform
        if (list_mask && !list_mask[l]) continue;
        int len = list_len[l];
        int64_t off = list_off[l];
        if (UNITS) {
            len = min(len - cbase, chunk_len);
            off += cbase;
        }
        if (len <= 0) continue;
        const uint8_t* lc = codes + off * M;
        // the first 256 codes are requested BEFORE the T2 row: both latencies overlap, and lists of
        // up to 256 codes (most of them) never wait for their codes after the LUT is ready
        // compiled code widths: any multiple of 8 bytes up to 64; a code is NLD loads of LW dwords
        constexpr bool PRE = MT > 0 && MT % 8 == 0 && MT <= 64;
        constexpr int LW = (MT % 16 == 0) ? 4 : 2, NLD = PRE ? MT / (4 * LW) : 1;
        typedef uint32_t cvec_t __attribute__((ext_vector_type(LW)));
        // (issued as inline asm: hipcc sinks an ordinary load down to its first use, behind both
        // barriers; the matching s_waitcnt is placed by hand where the codes are consumed)
        cvec_t cfirst[NLD];
        if (PRE) {
            const uint8_t* cp0 = lc + (int64_t)min(tid, len - 1) * (PRE ? MT : 16);
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                const uint8_t* a = cp0 + 4 * LW * u;
                if constexpr (LW == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cfirst[u]) : "v"(a) : "memory");
                else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(cfirst[u]) : "v"(a) : "memory");
            }
        }
        if (L2 && (!UNITS || pair != lut_pair)) {   // (uniform)
            if (UNITS) lut_pair = pair;
            __syncthreads();   // the previous list's gathers are finished
            const float* t2 = T2 + (int64_t)l * msz;
            if (MT > 0) {
                float tv[MT > 0 ? MT : 1];
#pragma unroll
                for (int i = 0; i < MT; i++) tv[i] = t2[tid + 256 * i];   // MT loads in flight
                lut_store_begin(lut_m0);
                lut_store_rows<MT>([&](int i) { return __builtin_fmaf(-2.0f, s2r[i], tv[i]); },
                                   std::make_integer_sequence<int, (MT > 0 ? MT : 1)>{});
                lut_store_done();
            } else {
                for (int e = tid; e < msz; e += 256) s_lut[e] = __builtin_fmaf(-2.0f, st2q[e], t2[e]);
            }
            __syncthreads();
        }
        // L2: the coarse distance; IP: <x_q, centroid_l>, computed per pair by k_pair_ip (a chain of d/8
        // dependent fmas per AVX lane has no place inside this loop)
        const float dis0 = coarse_dis[pair];
        const int64_t* lid = ids + off;
        const int pbase = pair_off[(int64_t)q * (P + 1) + p] + (UNITS ? cbase : 0);
        float* o = out + (int64_t)q * q_stride + pbase;
        // store + what the pre-filter tracks about a scored code
        // Distances are stored where something reads them: the first group's (its producer's histogram, the
        // unfiltered selection, the tie replay) and those of a group without a bound.  A consumer with a bound
        // keeps only its survivors; if k_select_final cannot finish the query from the slices (a slice
        // overflowed, > 256 equal keys at the cut) the repair launch scores the group again with stores.
        const bool store = !FILT || pg == 0 || !bound_on || sb.store_all;
        auto finish = [&](int j, bool ok, float dis) -> float {
            const float val = ok ? dis : sentinel;
            if (store) o[j] = val;
            if (FILT && pg == 0) {
                const bool valid = val != sentinel;
                g_fmn = fminf(g_fmn, valid ? val : INFINITY);
                g_fmx = fmaxf(g_fmx, valid ? val : -INFINITY);
                g_nv += valid ? 1 : 0;
            }
            return val;
        };
        // one code: validity, ADC (gathers issued together, adds in reference order), store
        auto do_code = [&](int j, const uint32_t* cw) -> float {
            // ids are read only when something can reject an entry (delete bit, range filter,
            // superseded slot); otherwise 8 of the 28 bytes per candidate stay in HBM
            bool ok = true;
            if (need_ids) {
                const int64_t id = lid[j];
                ok = id >= 0;  // bit 63 = kDelIdxMask (realtime_mem_data.h:26)
                if (ok) ok = is_valid_doc(filt, id);
            }
            float dis = dis0;
            if (PRE) {
                float t[PRE ? MT : 1];
#pragma unroll
                for (int m = 0; m < (PRE ? MT : 1); m++) t[m] = lut_gather(cw[m >> 2], m & 3, m);
                __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the add chain
#pragma unroll
                for (int m = 0; m < (PRE ? MT : 1); m++) dis += t[m];   // sequential, reference order
            } else {
                const uint8_t* cj = lc + (int64_t)j * M;
                for (int m = 0; m < M; m++) dis += s_lut[m * 256 + cj[m]];
            }
            return finish(j, ok, dis);
        };
        // uniform trip counts: append() ballots.  First 256 codes: already in registers.
        if (PRE) {
#pragma unroll
            for (int u = 0; u < NLD; u++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cfirst[u]) : : "memory");
        }
        {
            float val = sentinel;
            if (tid < len) {
                uint32_t cw[PRE ? MT / 4 : 1];
                if (PRE) {
#pragma unroll
                    for (int u = 0; u < NLD; u++)
#pragma unroll
                        for (int i = 0; i < LW; i++) cw[LW * u + i] = cfirst[u][i];
                }
                val = do_code(tid, cw);
            }
            if (FILT && bound_on) append(within(val), val, pbase + tid);
        }
        if constexpr (MT == 64) {
            // 64-byte codes: the 64 KB LUT leaves two workgroups per CU (2 waves per SIMD), so each
            // thread scores TWO codes per iteration -- 128 LDS gathers in flight, two independent
            // add chains -- instead of relying on other waves to cover its latency
            int j0 = 256;
            for (; j0 + 256 < len; j0 += 512) {   // both halves hold codes (uniform)
                const int ja = j0 + tid, jb = ja + 256;
                const bool ina = true, inb = jb < len;
                uint32_t cwa[16], cwb[16];
                {
                    const uint4* pa = reinterpret_cast<const uint4*>(lc + (int64_t)min(ja, len - 1) * 64);
                    const uint4* pb = reinterpret_cast<const uint4*>(lc + (int64_t)min(jb, len - 1) * 64);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint4 a = pa[u], b = pb[u];
                        cwa[4 * u] = a.x; cwa[4 * u + 1] = a.y; cwa[4 * u + 2] = a.z; cwa[4 * u + 3] = a.w;
                        cwb[4 * u] = b.x; cwb[4 * u + 1] = b.y; cwb[4 * u + 2] = b.z; cwb[4 * u + 3] = b.w;
                    }
                }
                bool oka = true, okb = true;
                if (need_ids) {
                    const int64_t ida = lid[min(ja, len - 1)], idb = lid[min(jb, len - 1)];
                    oka = ida >= 0;
                    if (oka) oka = is_valid_doc(filt, ida);
                    okb = idb >= 0;
                    if (okb) okb = is_valid_doc(filt, idb);
                }
                float ta[64], tb[64];
#pragma unroll
                for (int m = 0; m < 64; m++) ta[m] = lut_gather(cwa[m >> 2], m & 3, m);
#pragma unroll
                for (int m = 0; m < 64; m++) tb[m] = lut_gather(cwb[m >> 2], m & 3, m);
                __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the add chains
                float da = dis0, db = dis0;
#pragma unroll
                for (int m = 0; m < 64; m++) {       // each chain sequential, reference order
                    da += ta[m];
                    db += tb[m];
                }
                const float vala = ina ? finish(ja, oka, da) : sentinel;
                const float valb = inb ? finish(jb, okb, db) : sentinel;
                if (FILT && bound_on) {
                    append(within(vala), vala, pbase + ja);
                    append(within(valb), valb, pbase + jb);
                }
            }
            if (j0 < len) {   // at most 256 codes left: one per thread
                const int j = j0 + tid;
                float val = sentinel;
                if (j < len) {
                    uint32_t cw[16];
                    const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)j * 64);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint4 cv = cp[u];
                        cw[4 * u] = cv.x; cw[4 * u + 1] = cv.y; cw[4 * u + 2] = cv.z; cw[4 * u + 3] = cv.w;
                    }
                    val = do_code(j, cw);
                }
                if (FILT && bound_on) append(within(val), val, pbase + j);
            }
        } else {
            for (int j0 = 256; j0 < len; j0 += 256) {
                const int j = j0 + tid;
                float val = sentinel;
                if (j < len) {
                    uint32_t cw[PRE ? MT / 4 : 1];
                    if (PRE) {
                        const cvec_t* cp = reinterpret_cast<const cvec_t*>(lc + (int64_t)j * (PRE ? MT : 16));
#pragma unroll
                        for (int u = 0; u < NLD; u++) {
                            const cvec_t cv = cp[u];
#pragma unroll
                            for (int i = 0; i < LW; i++) cw[LW * u + i] = cv[i];
                        }
                    }
                    val = do_code(j, cw);
                }
                if (FILT && bound_on) append(within(val), val, pbase + j);
            }
        }
    }
    if (FILT && pg > 0) flush();   // also without a bound: the slice count must be written (0)
    if (FILT && pg == 0) {
        // ---- producer: bound of this query's K-th best from its first probe group ----
        // 256-bin histogram of the group's valid keys over [min, max]; tau = upper edge of the bin
        // holding the K-th smallest.  At least K candidates are <= tau, hence the whole final top-K.
        __syncthreads();   // this workgroup's distance stores are visible to all its threads
        int* hist = reinterpret_cast<int*>(s_stage);   // staging has not been used yet
        const int n0 = pair_off[(int64_t)q * (P + 1) + min(G, P)];
        const float* o0 = out + (int64_t)q * q_stride;
        // float range -> key range (a zero may carry either sign: take the widest pair of keys)
        const float fmn = g_fmn == 0.f ? -0.f : g_fmn, fmx = g_fmx == 0.f ? 0.f : g_fmx;
        uint32_t mn = g_nv ? (L2 ? dis_key<L2>(fmn) : dis_key<L2>(fmx)) : 0xffffffffu;
        uint32_t mx = g_nv ? (L2 ? dis_key<L2>(fmx) : dis_key<L2>(fmn)) : 0u;
        int nv = g_nv;
        mn = wave_min_u32(mn);
        mx = wave_max_u32(mx);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nv += __shfl_xor(nv, off, 64);
        if (lane == 0) {
            s_red[threadIdx.x >> 6] = mn;
            s_red[4 + (threadIdx.x >> 6)] = mx;
            s_red[8 + (threadIdx.x >> 6)] = (uint32_t)nv;
        }
        hist[threadIdx.x] = 0;
        __syncthreads();
        mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
        nv = (int)(s_red[8] + s_red[9] + s_red[10] + s_red[11]);
        uint32_t tau = 0xffffffffu;
        if (nv >= sb.K) {   // uniform
            const uint32_t range = mx - mn;
            const int sh = range >= 256u ? (32 - __clz((int)range)) - 8 : 0;   // (range >> sh) < 256
            for (int i0 = 0; i0 < n0; i0 += 256 * 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = o0[min(i0 + u * 256 + (int)threadIdx.x, n0 - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t key = dis_key<L2>(t[u]);
                    if (i0 + u * 256 + (int)threadIdx.x < n0 && key < KEY_SENTINEL)
                        atomicAdd(&hist[(key - mn) >> sh], 1);
                }
            }
            __syncthreads();
            if (threadIdx.x < 64) {   // wave 0: scan of the 256 bins, 4 per lane
                int c[4], c4 = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    c[u] = hist[lane * 4 + u];
                    c4 += c[u];
                }
                const int incl = wave_incl_scan(c4);
                int run = incl - c4;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (run < sb.K && sb.K <= run + c[u]) {
                        unsigned long long edge = (unsigned long long)mn +
                                                  (((unsigned long long)(lane * 4 + u) + 1ull) << sh) - 1ull;
                        if (edge > (unsigned long long)mx) edge = mx;
                        s_tau = (uint32_t)edge;
                    }
                    run += c[u];
                }
            }
            __syncthreads();
            tau = s_tau;
        }
        if (threadIdx.x == 0)
            __hip_atomic_store(&sb.ready[q], tau < KEY_SENTINEL ? ((1ull << 32) | tau) : (2ull << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the group's own candidates within the bound become its survivor slice (slice 0), like a
        // consumer's: k_select_final then reads a few hundred items per query and never the distance
        // buffer (one wave walking a long first group -- 24 k candidates at C4 -- was the slow part)
        if (tau < KEY_SENTINEL) {   // uniform
            __syncthreads();        // the histogram (aliasing the staging area) has been read
            for (int i0 = 0; i0 < n0; i0 += 256 * 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = o0[min(i0 + u * 256 + (int)threadIdx.x, n0 - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int idx = i0 + u * 256 + (int)threadIdx.x;
                    append(idx < n0 && dis_key<L2>(t[u]) <= tau, t[u], idx);
                }
            }
        }
        flush();   // without a bound: count 0
    }
    };   // body
    if (!repair) {
        body(q, pg);
    } else if (UNITS) {
        // a workgroup takes a contiguous run of units: consecutive chunks of one list share its LUT
        const int nu = *rq_count, per = (nu + (int)gridDim.x - 1) / (int)gridDim.x;
        const int w0 = (int)blockIdx.x * per, w1 = min(nu, w0 + per);
        for (int w = w0; w < w1; w++) {
            const uint32_t u = (uint32_t)rq_list[w];
            cbase = (int)(u & 8191u) * chunk_len;
            body((int)(u >> 20), (int)((u >> 13) & 127u));
            __syncthreads();   // the LUT of this unit has been consumed
        }
    } else {
        const int nrq = *rq_count;
        for (int w = blockIdx.x; w / pg_cnt < nrq; w += gridDim.x) {
            body(rq_list[w / pg_cnt], pg_lo + w % pg_cnt);
            __syncthreads();   // the LUT of this item has been consumed
        }
    }
}

int scan_slice_cap() { return SCAN_SLICE; }
bool scan_cf_applies(bool l2, int M, int P, int G, bool have_sums, bool store_all) {
    return l2 && have_sums && !store_all && (M == 16 || M == 32) && P > G;
}

int scan_group_size(int nq, int P, int G0) {
    // probes per workgroup: amortise the query table, but keep >= ~4096 workgroups in flight
    static const int g_env = getenv("GAMMA_HIP_SCAN_G") ? atoi(getenv("GAMMA_HIP_SCAN_G")) : 0;
    int G = g_env > 0 ? g_env : G0;
    while (G > 1 && (int64_t)nq * ((P + G - 1) / G) < 4096) G >>= 1;
    return std::max(1, std::min(G, P));
}

void launch_ivfpq_scan_pair(hipStream_t s, bool l2, const float* x, int nq, int d, int M, int P,
                            const int* probe_list, const float* coarse_dis, const float* cc,
                            const float* st2, const float* T2, const int64_t* list_off,
                            const int* list_len, const uint8_t* list_mask, int nlist,
                            const uint8_t* codes, const int64_t* ids, const int* pair_off,
                            int64_t q_stride, float* out, const FilterDesc* ftab, const int* qfil, int need_ids,
                            const int* qperm, int G, int pg_lo, int pg_cnt, int sparse, const ScanBound* bound,
                            const float* pqc_fused, const int* rq_list, const int* rq_count, int chunk_len, int max_units) {
    if (nq <= 0 || pg_cnt <= 0) return;
    if (chunk_len > 0 && (bound || pqc_fused || !rq_list || G != 1 || pg_lo != 0 || pg_cnt != P || max_units < 1)) abort();
    if (pqc_fused) {   // the table is computed inside the kernel (IPF): one workgroup per query, M 16 / 32
        if (!bound || pg_cnt != 1 || (M != 16 && M != 32)) abort();
        st2 = pqc_fused;
    }
    // LUT | survivor staging | a few words (see the kernel)
    size_t lds = (size_t)M * 256 * sizeof(float) + SCAN_STAGE * sizeof(unsigned long long) + 16 * sizeof(int);
    dim3 grid((unsigned)(8 * (int64_t)((nq + 7) / 8) * pg_cnt));
    if (bound) {   // P(0) | P(t+1) C(t) ...: whole batches, see the kernel
        const int64_t nq8 = (nq + 7) / 8, nb = (nq8 + SCAN_BATCH - 1) / SCAN_BATCH;
        grid.x = (unsigned)(8 * (SCAN_BATCH + nb * SCAN_BATCH * pg_cnt));
    }
    ScanBound sb = {nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
    if (bound) sb = *bound;
    // filter pass for the consumers of a bounded L2 scan: needs the per-code sums (sb.sums) and survivor-only consumers
    const bool cf = bound && l2 && !pqc_fused && pg_cnt > 1 && sb.sums && sb.t2max && !sb.store_all && (M == 16 || M == 32);
    if (rq_list) {   // repair launch: a fixed grid loops over the flagged (query, group) items
        if (bound || pqc_fused) abort();
        grid.x = (unsigned)std::min<int64_t>((int64_t)nq * pg_cnt, 2048);
    }
    if (chunk_len > 0) {   // as many workgroups as are resident at once (LDS: the LUT), no more than there can be units
        const int per_cu = std::max(1, std::min(8, (int)(160 * 1024 / (lds + 1024))));
        grid.x = (unsigned)std::min<int64_t>(max_units, 256 * per_cu);
    }
    if (cf) lds += SCAN_CF_CAP * sizeof(uint2) + 16;
#define GH_SCAN(LL, MT, FF)                                                                       \
    GH_SCAN4(LL, MT, FF, false)
#define GH_SCAN4(LL, MT, FF, II) GH_SCAN5(LL, MT, FF, II, false)
#define GH_SCAN5(LL, MT, FF, II, UU)                                                                       \
    hipLaunchKernelGGL((k_ivfpq_scan_pair<LL, MT, FF, II, UU>), grid, dim3(256), lds, s, x, nq, d, M, P, G,     \
                       probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, \
                       ids, pair_off, q_stride, out, ftab, qfil, need_ids, LL ? INFINITY : -INFINITY, qperm,   \
                       pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len)
#define GH_SCAN_M(LL, FF)                       \
    do {                                        \
        if (M == 16) GH_SCAN(LL, 16, FF);       \
        else if (M == 32) GH_SCAN(LL, 32, FF);  \
        else if (M == 64) GH_SCAN(LL, 64, FF);  \
        else if (M == 8) GH_SCAN(LL, 8, FF);    \
        else if (M == 24) GH_SCAN(LL, 24, FF);  \
        else if (M == 48) GH_SCAN(LL, 48, FF);  \
        else GH_SCAN(LL, 0, FF);                \
    } while (0)
    if (chunk_len > 0) {
#define GH_SCAN_U(LL)                                           \
    do {                                                        \
        if (M == 16) GH_SCAN5(LL, 16, false, false, true);      \
        else if (M == 32) GH_SCAN5(LL, 32, false, false, true); \
        else if (M == 64) GH_SCAN5(LL, 64, false, false, true); \
        else if (M == 8) GH_SCAN5(LL, 8, false, false, true);   \
        else GH_SCAN5(LL, 0, false, false, true);               \
    } while (0)
        if (l2) GH_SCAN_U(true);
        else GH_SCAN_U(false);
#undef GH_SCAN_U
    } else if (pqc_fused) {
        if (l2 && M == 16) GH_SCAN4(true, 16, true, true);
        else if (l2) GH_SCAN4(true, 32, true, true);
        else if (M == 16) GH_SCAN4(false, 16, true, true);
        else GH_SCAN4(false, 32, true, true);
    } else if (cf) {
#define GH_SCAN_CF(MT)                                                                                                  \
    hipLaunchKernelGGL((k_ivfpq_scan_pair<true, MT, true, false, false, true>), grid, dim3(256), lds, s, x, nq, d, M, P, G, \
                       probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off,  \
                       q_stride, out, ftab, qfil, need_ids, INFINITY, qperm, pg_lo, pg_cnt, sparse, sb, rq_list,         \
                       rq_count, chunk_len)
        if (M == 16) GH_SCAN_CF(16);
        else GH_SCAN_CF(32);
#undef GH_SCAN_CF
    } else if (bound) {
        if (l2) GH_SCAN_M(true, true);
        else GH_SCAN_M(false, true);
    } else {
        if (l2) GH_SCAN_M(true, false);
        else GH_SCAN_M(false, false);
    }
#undef GH_SCAN_M
#undef GH_SCAN
#undef GH_SCAN4
#undef GH_SCAN5
}

// ------------------------------------------------------------------------------------
// positions in a query's candidate segment -> vector ids (KnnSearchResults::add stores
// ids[j], gamma_index_ivfpq.h:363-369).  grid = nq.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_map_candidates(const int* __restrict__ pos, int R, int P,
                                                        const int* __restrict__ probe_list,
                                                        const int* __restrict__ pair_off,
                                                        const int64_t* __restrict__ list_off,
                                                        const int64_t* __restrict__ ids,
                                                        int64_t* __restrict__ cand_ids,
                                                        const uint8_t* __restrict__ only) {
    const int q = blockIdx.x;
    if (only && !only[q]) return;   // rows k_select_final has already mapped
    const int* off = pair_off + (int64_t)q * (P + 1);
    for (int r = threadIdx.x; r < R; r += 256) {
        const int ps = pos[(int64_t)q * R + r];
        int64_t id = -1;
        if (ps >= 0) {
            // last p with off[p] <= ps
            int lo = 0, hi = P - 1;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (off[mid] <= ps) lo = mid; else hi = mid - 1;
            }
            const int l = probe_list[(int64_t)q * P + lo];
            id = ids[list_off[l] + (ps - off[lo])] & 0x7fffffffffffffffLL;
        }
        cand_ids[(int64_t)q * R + r] = id;
    }
}
void launch_map_candidates(hipStream_t s, const int* pos, int nq, int R, int P,
                           const int* probe_list, const int* pair_off, const int64_t* list_off,
                           const int64_t* ids, int64_t* cand_ids, const uint8_t* only) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_map_candidates, dim3(nq), dim3(256), 0, s, pos, R, P, probe_list, pair_off,
                       list_off, ids, cand_ids, only);
}

// ------------------------------------------------------------------------------------
// a9: exact re-rank distances (compute_dis, gamma_index_ivfpq.cc:642-680).  8 threads
// per candidate = the 8 lane accumulators of fvec_L2sqr / fvec_inner_product; the
// cross-lane reduction mirrors extractf128 + add + 2x haddps.  grid = nq, block = 256
// (32 candidates in flight).  Out-of-window scores and empty slots get the sentinel.
// ------------------------------------------------------------------------------------
template <bool L2>
__global__ __launch_bounds__(256) void k_rerank_dist(const float* __restrict__ x, int d,
                                                     const float* __restrict__ raw, int64_t nraw,
                                                     const int64_t* __restrict__ cand_ids, int R,
                                                     float min_score, float max_score,
                                                     float* __restrict__ out) {
    const int q = blockIdx.x;
    const int l = threadIdx.x & 7, g = threadIdx.x >> 3;
    const float* xq = x + (int64_t)q * d;
    const float sentinel = L2 ? INFINITY : -INFINITY;
    for (int r0 = blockIdx.y * 32; r0 < R; r0 += gridDim.y * 32) {
        const int r = r0 + g;
        int64_t id = -1;
        if (r < R) id = cand_ids[(int64_t)q * R + r];
        const bool live = id >= 0 && id < nraw;
        float dis = rerank_dist8<L2>(xq, raw + (live ? id : 0) * d, d, l, live);
        if (l == 0 && r < R) {
            if (!live || !(dis <= max_score && dis >= min_score)) dis = sentinel;
            out[(int64_t)q * R + r] = dis;
        }
    }
}
void launch_rerank_dist(hipStream_t s, bool l2, const float* x, int nq, int d, const float* raw,
                        int64_t nraw, const int64_t* cand_ids, int R, float min_score,
                        float max_score, float* out) {
    if (nq <= 0) return;
    const int gy = (R + 31) / 32;   // 32 candidates (8 lanes each) per workgroup
    if (l2)
        hipLaunchKernelGGL((k_rerank_dist<true>), dim3(nq, gy), dim3(256), 0, s, x, d, raw, nraw, cand_ids,
                           R, min_score, max_score, out);
    else
        hipLaunchKernelGGL((k_rerank_dist<false>), dim3(nq, gy), dim3(256), 0, s, x, d, raw, nraw,
                           cand_ids, R, min_score, max_score, out);
}

// ------------------------------------------------------------------------------------
// a9, fused: exact re-rank distances + top-k + output in ONE kernel (compute_dis with
// has_rank, gamma_index_ivfpq.cc:646-680).  One workgroup per query: the R exact distances
// become (key, candidate rank) items in LDS, a block rank sort orders them -- equal exact
// distances keep the ADC order of the candidates -- and the first k go out with their ids
// (empty slots: -1 / heap neutral).  R <= 1024.
// ------------------------------------------------------------------------------------
template <bool L2>
__global__ __launch_bounds__(256) void k_rerank_topk(const float* __restrict__ x, int d,
                                                     const float* __restrict__ raw, int64_t nraw,
                                                     const int64_t* __restrict__ cand_ids, int R, int k,
                                                     float min_score, float max_score, float neutral,
                                                     float* __restrict__ distances,
                                                     int64_t* __restrict__ labels, int nq,
                                                     const int* __restrict__ qperm, TieFlags tf) {
    __shared__ unsigned long long s_it[1024];
    __shared__ int64_t s_id[1024];
    __shared__ int s_tie;
    // With the scan's query order (qperm: queries sorted by the spatial rank of their nearest list) XCD x takes
    // the x-th eighth of that order: queries running together share candidates (a batch references every raw
    // row ~3 times), so their rows are served by that XCD's L2 instead of HBM.  Results do not depend on it.
    int q = blockIdx.x;
    if (qperm) {
        const int qi = (blockIdx.x & 7) * ((nq + 7) >> 3) + (blockIdx.x >> 3);
        if (qi >= nq) return;
        q = qperm[qi];
    } else if (q >= nq) {
        return;
    }
    const int l = threadIdx.x & 7, g = threadIdx.x >> 3;
    const float* xq = x + (int64_t)q * d;
    const float sentinel = L2 ? INFINITY : -INFINITY;
    // all candidate ids first (one coalesced pass): the row gathers below then start without a
    // dependent id load in front of each of them
    for (int r = threadIdx.x; r < R; r += 256) s_id[r] = cand_ids[(int64_t)q * R + r];
    __syncthreads();
    for (int r0 = 0; r0 < R; r0 += 32) {
        const int r = r0 + g;
        int64_t id = -1;
        if (r < R) id = s_id[r];
        const bool live = id >= 0 && id < nraw;
        float dis = rerank_dist8<L2>(xq, raw + (live ? id : 0) * d, d, l, live);
        if (l == 0 && r < R) {
            if (!live || !(dis <= max_score && dis >= min_score)) dis = sentinel;
            const uint32_t key = L2 ? f2key(dis) : ~f2key(dis);
            s_it[r] = ((unsigned long long)key << 32) | (unsigned)r;
        }
    }
    if (tf.list && threadIdx.x == 0) s_tie = tf.cut ? tf.cut[q] : 0;
    block_rank_sort<256, 4>(s_it, R);   // R distinct items (the rank field differs)
    if (tf.list) {
        // exact ties (ties.hip): two of the first k+1 exact distances equal -- their order, or which of them
        // stays inside the k, is decided by the reference's heaps -- or the top-R cut went through a tie
        for (int i = threadIdx.x; i < k && i + 1 < R; i += 256) {
            const uint32_t ka = (uint32_t)(s_it[i] >> 32), kb = (uint32_t)(s_it[i + 1] >> 32);
            if (ka == kb && ka != (L2 ? f2key(sentinel) : ~f2key(sentinel))) s_tie = 1;   // benign race: same value
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_tie) {
            tf.list[atomicAdd(tf.count, 1)] = q;
            if (tf.stats) atomicAdd(tf.stats + 2, 1ull);
        }
    }
    for (int i = threadIdx.x; i < k; i += 256) {
        float val = neutral;
        int64_t id = -1;
        if (i < R) {
            const unsigned long long it = s_it[i];
            const uint32_t key = (uint32_t)(it >> 32);
            const float dv = key2f(L2 ? key : ~key);
            if (dv != sentinel) {
                val = dv;
                id = s_id[(uint32_t)it];
            }
        }
        distances[(int64_t)q * k + i] = val;
        labels[(int64_t)q * k + i] = id;
    }
}
void launch_rerank_topk(hipStream_t s, bool l2, const float* x, int nq, int d, const float* raw,
                        int64_t nraw, const int64_t* cand_ids, int R, int k, float min_score,
                        float max_score, float neutral, float* distances, int64_t* labels, const int* qperm,
                        const TieFlags* ties) {
    if (nq <= 0) return;
    const dim3 grid((unsigned)(8 * ((nq + 7) / 8)));
    const TieFlags tf = ties ? *ties : TieFlags{};
    if (l2)
        hipLaunchKernelGGL((k_rerank_topk<true>), grid, dim3(256), 0, s, x, d, raw, nraw, cand_ids,
                           R, k, min_score, max_score, neutral, distances, labels, nq, qperm, tf);
    else
        hipLaunchKernelGGL((k_rerank_topk<false>), grid, dim3(256), 0, s, x, d, raw, nraw, cand_ids,
                           R, k, min_score, max_score, neutral, distances, labels, nq, qperm, tf);
}

// ------------------------------------------------------------------------------------
// IVFFLAT list scan (GammaIVFFlatScanner1::scan_codes, index/impl/gamma_index_ivfflat.h:52-75): the reference's
// lists hold the vectors themselves; here a list holds vector ids and the rows come from the raw store (the same
// floats).  One workgroup per (query, probe) pair, eight threads per list entry = the eight AVX lane accumulators
// of fvec_L2sqr / fvec_inner_product (rerank_dev.h).  Entries with bit 63, filtered docs and scores outside the
// window get the sentinel; one fp32 per entry into the query's slab at the pair's offset.
// ------------------------------------------------------------------------------------
template <bool L2>
__global__ __launch_bounds__(256) void k_ivfflat_scan(const float* __restrict__ x, int d, int P,
                                                      const int* __restrict__ pair_off,
                                                      const int64_t* __restrict__ pair_base,
                                                      const int64_t* __restrict__ ids,
                                                      const float* __restrict__ raw, int64_t nraw, int64_t q_stride,
                                                      float* __restrict__ out, const FilterDesc* __restrict__ ftab,
                                                      int need_filter, float min_score, float max_score) {
    const int q = blockIdx.x / P, p = blockIdx.x - q * P;
    const int off = pair_off[(int64_t)q * (P + 1) + p], len = pair_off[(int64_t)q * (P + 1) + p + 1] - off;
    if (len <= 0) return;   // uniform
    const int64_t base = pair_base[(int64_t)q * P + p];
    const float* xq = x + (int64_t)q * d;
    const int l8 = threadIdx.x & 7, g = threadIdx.x >> 3;
    const float sentinel = L2 ? INFINITY : -INFINITY;
    for (int j0 = 0; j0 < len; j0 += 32) {
        const int j = j0 + g;
        int64_t id = -1;
        if (j < len) id = ids[base + j];
        const int64_t vid = id & 0x7fffffffffffffffLL;
        bool live = j < len && id >= 0 && vid < nraw;   // id < 0: bit 63, superseded by an Update
        if (need_filter && live) live = is_valid_doc(ftab[0], vid);
        float dis = rerank_dist8<L2>(xq, raw + (live ? vid : 0) * d, d, l8, live);
        if (l8 == 0 && j < len) {
            if (!live || !(dis <= max_score && dis >= min_score)) dis = sentinel;
            out[(int64_t)q * q_stride + off + j] = dis;
        }
    }
}
void launch_ivfflat_scan(hipStream_t s, bool l2, const float* x, int nq, int d, int P, const int* pair_off,
                         const int64_t* pair_base, const int64_t* ids, const float* raw, int64_t nraw, int64_t q_stride,
                         float* out, const FilterDesc* ftab, int need_filter, float min_score, float max_score) {
    if (nq <= 0 || P <= 0) return;
    const dim3 grid((unsigned)((int64_t)nq * P));
    if (l2)
        hipLaunchKernelGGL((k_ivfflat_scan<true>), grid, dim3(256), 0, s, x, d, P, pair_off, pair_base, ids, raw, nraw,
                           q_stride, out, ftab, need_filter, min_score, max_score);
    else
        hipLaunchKernelGGL((k_ivfflat_scan<false>), grid, dim3(256), 0, s, x, d, P, pair_off, pair_base, ids, raw, nraw,
                           q_stride, out, ftab, need_filter, min_score, max_score);
}

// final outputs from a top-k selection over re-ranked (or flat) candidates:
//   labels = src_ids ? src_ids[q][pos] : id_base + pos ; empty -> -1 / heap neutral
__global__ __launch_bounds__(256) void k_finalize_topk(const float* __restrict__ sel_vals,
                                                       const int* __restrict__ sel_pos, int k,
                                                       const int64_t* __restrict__ src_ids,
                                                       int64_t src_stride, int64_t id_base,
                                                       float neutral,
                                                       float* __restrict__ distances,
                                                       int64_t* __restrict__ labels, int n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * k) return;
    const int q = (int)(i / k);
    const int ps = sel_pos[i];
    if (ps < 0) {
        distances[i] = neutral;
        labels[i] = -1;
    } else {
        distances[i] = sel_vals[i];
        labels[i] = src_ids ? src_ids[(int64_t)q * src_stride + ps] : id_base + (int64_t)ps;
    }
}
void launch_finalize_topk(hipStream_t s, const float* sel_vals, const int* sel_pos, int nq, int k,
                          const int64_t* src_ids, int64_t src_stride, int64_t id_base,
                          float neutral, float* distances, int64_t* labels) {
    if (nq <= 0 || k <= 0) return;
    int64_t tot = (int64_t)nq * k;
    hipLaunchKernelGGL(k_finalize_topk, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, sel_vals,
                       sel_pos, k, src_ids, src_stride, id_base, neutral, distances, labels, nq);
}

// has_rank == false (gamma_index_ivfpq.cc:681-696): candidates are already sorted by ADC
// distance; copy the first k whose score is inside the window.  grid = nq.
__global__ __launch_bounds__(256) void k_finalize_norank(const float* __restrict__ cand_dis,
                                                         const int64_t* __restrict__ cand_ids, int R,
                                                         int k, float min_score, float max_score,
                                                         float neutral, float* __restrict__ distances,
                                                         int64_t* __restrict__ labels, TieFlags tf) {
    __shared__ int s_w[4];
    __shared__ int s_tie;
    const int q = blockIdx.x;
    int running = 0;
    if (tf.list && threadIdx.x == 0) s_tie = tf.cut ? tf.cut[q] : 0;
    for (int r0 = 0; r0 < R && running < k; r0 += 256) {
        const int r = r0 + threadIdx.x;
        float dis = 0.f;
        int64_t id = -1;
        if (r < R) {
            dis = cand_dis[(int64_t)q * R + r];
            id = cand_ids[(int64_t)q * R + r];
        }
        const int flag = (id != -1 && dis <= max_score && dis >= min_score) ? 1 : 0;
        int tot;
        const int ex = block_excl_scan256(flag, s_w, tot);
        const int slot = running + ex;
        if (flag && slot < k) {
            distances[(int64_t)q * k + slot] = dis;
            labels[(int64_t)q * k + slot] = id;
            // exact ties (ties.hip): an entry that is taken and its successor at the same ADC distance -- their
            // order, or which of them is the k-th, is whatever heap_reorder of the reference's R-heap leaves
            if (tf.list && r + 1 < R && cand_ids[(int64_t)q * R + r + 1] != -1 &&
                cand_dis[(int64_t)q * R + r + 1] == dis)
                s_tie = 1;
        }
        running += tot;
    }
    if (tf.list) {
        __syncthreads();
        if (threadIdx.x == 0 && s_tie) {
            tf.list[atomicAdd(tf.count, 1)] = q;
            if (tf.stats) atomicAdd(tf.stats + 2, 1ull);
        }
    }
    for (int i = min(running, k) + threadIdx.x; i < k; i += 256) {
        distances[(int64_t)q * k + i] = neutral;
        labels[(int64_t)q * k + i] = -1;
    }
}
void launch_finalize_norank(hipStream_t s, const float* cand_dis, const int64_t* cand_ids, int nq,
                            int R, int k, float min_score, float max_score, float neutral,
                            float* distances, int64_t* labels, const TieFlags* ties) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_finalize_norank, dim3(nq), dim3(256), 0, s, cand_dis, cand_ids, R, k,
                       min_score, max_score, neutral, distances, labels, ties ? *ties : TieFlags{});
}

// ------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------
__global__ void k_pos_to_i32(const int* __restrict__ pos, int* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = pos[i];
}

// [shard][nq][R] -> [nq][shard*R] for the sharded merge
__global__ __launch_bounds__(256) void k_gather_shards(const float* __restrict__ all_dis,
                                                       const int64_t* __restrict__ all_ids,
                                                       int nshards, int nq, int R,
                                                       float* __restrict__ dis,
                                                       int64_t* __restrict__ ids, float sentinel) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)nshards * nq * R;
    if (i >= tot) return;
    const int r = (int)(i % R);
    const int64_t t = i / R;
    const int q = (int)(t % nq);
    const int sh = (int)(t / nq);
    const int64_t id = all_ids[i];
    const int64_t o = ((int64_t)q * nshards + sh) * R + r;
    dis[o] = id < 0 ? sentinel : all_dis[i];
    ids[o] = id;
}
void launch_gather_shards(hipStream_t s, const float* all_dis, const int64_t* all_ids, int nshards,
                          int nq, int R, float* dis, int64_t* ids, float sentinel) {
    int64_t tot = (int64_t)nshards * nq * R;
    if (tot <= 0) return;
    hipLaunchKernelGGL(k_gather_shards, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, all_dis,
                       all_ids, nshards, nq, R, dis, ids, sentinel);
}

// out_ids[q][r] = pos<0 ? -1 : src_ids[q][pos]
__global__ __launch_bounds__(256) void k_take_ids(const int* __restrict__ pos,
                                                  const int64_t* __restrict__ src_ids,
                                                  int64_t src_stride, int R, int64_t n,
                                                  int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t q = i / R;
    const int ps = pos[i];
    out[i] = ps < 0 ? -1 : src_ids[q * src_stride + ps];
}
void launch_take_ids(hipStream_t s, const int* pos, const int64_t* src_ids, int64_t src_stride,
                     int nq, int R, int64_t* out) {
    int64_t n = (int64_t)nq * R;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_take_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, src_ids,
                       src_stride, R, n, out);
}

// coarse result packing: selected positions are the centroid ids
__global__ void k_i32_copy_check(const int* __restrict__ in, int* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// set / clear bits of the delete bitmap
__global__ void k_bitmap_set(uint8_t* __restrict__ bm, const int64_t* __restrict__ docids, int64_t n,
                             int64_t nbits, int value) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t id = docids[i];
    if (id < 0 || id >= nbits) return;
    unsigned int* w = reinterpret_cast<unsigned int*>(bm) + (id >> 5);
    unsigned int m = 1u << (id & 31);  // little-endian: bit (id&7) of byte id>>3
    if (value) atomicOr(w, m); else atomicAnd(w, ~m);
}
void launch_bitmap_set(hipStream_t s, uint8_t* bm, const int64_t* docids, int64_t n, int64_t nbits,
                       int value) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_bitmap_set, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bm, docids, n,
                       nbits, value);
}

// mark an inverted-list entry as superseded (ids[pos] |= kDelIdxMask)
__global__ void k_mark_moved(int64_t* __restrict__ ids, int64_t pos) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ids[pos] |= (int64_t)(1ULL << 63);
}
void launch_mark_moved(hipStream_t s, int64_t* ids, int64_t pos) {
    hipLaunchKernelGGL(k_mark_moved, dim3(1), dim3(64), 0, s, ids, pos);
}

// Arena repack (gamma_hip_store.cpp, arena_repack): every list's live entries move from (old arrays, old offset)
// to (new arrays, new offset).  grid = (nlist, chunks); the code bytes move as dwords when M % 4 == 0.
__global__ __launch_bounds__(256) void k_repack_lists(const uint8_t* __restrict__ oc, const int64_t* __restrict__ oi,
                                                      uint8_t* __restrict__ nc, int64_t* __restrict__ ni,
                                                      const int64_t* __restrict__ old_off,
                                                      const int64_t* __restrict__ new_off,
                                                      const int* __restrict__ len, int M) {
    const int l = blockIdx.x;
    const int n = len[l];
    const int64_t a = old_off[l], b = new_off[l];
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) ni[b + i] = oi[a + i];
    if ((M & 3) == 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(oc + a * M);
        uint32_t* dst = reinterpret_cast<uint32_t*>(nc + b * M);
        const int64_t nw = (int64_t)n * (M >> 2);
        for (int64_t i = blockIdx.y * 256 + threadIdx.x; i < nw; i += gridDim.y * 256) dst[i] = src[i];
    } else {
        const int64_t nb = (int64_t)n * M;
        for (int64_t i = blockIdx.y * 256 + threadIdx.x; i < nb; i += gridDim.y * 256) nc[b * M + i] = oc[a * M + i];
    }
}
void launch_repack_lists(hipStream_t s, const uint8_t* oc, const int64_t* oi, uint8_t* nc, int64_t* ni,
                         const int64_t* old_off, const int64_t* new_off, const int* len, int nlist, int M,
                         int max_len) {
    if (nlist <= 0) return;
    const int chunks = std::max(1, std::min(64, (max_len + 1023) / 1024));
    hipLaunchKernelGGL(k_repack_lists, dim3(nlist, chunks), dim3(256), 0, s, oc, oi, nc, ni, old_off, new_off, len, M);
}

// ------------------------------------------------------------------------------------
// Per-code table sums of the L2 scan's filter pass (k_ivfpq_scan_pair<.., CF>): sums[pos] = sum_m T2[list][m][code[m]]
// (sequential fp32 adds from 0).  The value is only ever used inside a bound with a margin that covers its rounding,
// so any fixed order would do.
//   ranges: range r = n[r] entries of list list_no[r] from arena position pos[r]  (grid = (ranges, chunks))
//   lists : every entry of every list at its current extent                      (grid = (nlist, chunks))
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void code_sums_span(const float* __restrict__ t2l, const uint8_t* __restrict__ codes, int M,
                                               int64_t pos, int n, float* __restrict__ sums) {
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
        const uint8_t* c = codes + (pos + i) * M;
        float acc = 0.f;
        for (int m = 0; m < M; m++) acc += t2l[m * 256 + c[m]];
        sums[pos + i] = acc;
    }
}
__global__ __launch_bounds__(256) void k_code_sums_ranges(const float* __restrict__ T2, const uint8_t* __restrict__ codes, int M,
                                                          const int* __restrict__ list_no, const int64_t* __restrict__ pos,
                                                          const int* __restrict__ n, float* __restrict__ sums) {
    const int r = blockIdx.x;
    code_sums_span(T2 + (int64_t)list_no[r] * M * 256, codes, M, pos[r], n[r], sums);
}
__global__ __launch_bounds__(256) void k_code_sums_lists(const float* __restrict__ T2, const uint8_t* __restrict__ codes, int M,
                                                         const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
                                                         float* __restrict__ sums) {
    const int l = blockIdx.x;
    code_sums_span(T2 + (int64_t)l * M * 256, codes, M, list_off[l], list_len[l], sums);
}
__global__ __launch_bounds__(256) void k_code_sums_one(const float* __restrict__ t2l, const uint8_t* __restrict__ codes, int M,
                                                       int64_t pos, int n, float* __restrict__ sums) {
    code_sums_span(t2l, codes, M, pos, n, sums);
}
void launch_code_sums_one(hipStream_t s, const float* T2, const uint8_t* codes, int M, int list_no, int64_t pos, int n,
                          float* sums) {
    if (n <= 0) return;
    const int chunks = std::max(1, std::min(64, (n + 1023) / 1024));
    hipLaunchKernelGGL(k_code_sums_one, dim3(1, chunks), dim3(256), 0, s, T2 + (int64_t)list_no * M * 256, codes, M, pos, n, sums);
}
void launch_code_sums_ranges(hipStream_t s, const float* T2, const uint8_t* codes, int M, const int* list_no, const int64_t* pos,
                             const int* n, int nranges, int max_n, float* sums) {
    if (nranges <= 0) return;
    const int chunks = std::max(1, std::min(64, (max_n + 1023) / 1024));
    hipLaunchKernelGGL(k_code_sums_ranges, dim3(nranges, chunks), dim3(256), 0, s, T2, codes, M, list_no, pos, n, sums);
}
void launch_code_sums_lists(hipStream_t s, const float* T2, const uint8_t* codes, int M, const int64_t* list_off,
                            const int* list_len, int nlist, int max_len, float* sums) {
    if (nlist <= 0) return;
    const int chunks = std::max(1, std::min(64, (max_len + 1023) / 1024));
    hipLaunchKernelGGL(k_code_sums_lists, dim3(nlist, chunks), dim3(256), 0, s, T2, codes, M, list_off, list_len, sums);
}
// t2max[l] = sum_m max_c |T2[l][m][c]|: bounds every partial sum of a code's table entries (the filter's margin)
__global__ __launch_bounds__(256) void k_t2_rowmax(const float* __restrict__ T2, int M, float* __restrict__ t2max) {
    __shared__ float s_w[4];
    const int l = blockIdx.x, tid = threadIdx.x;
    const float* t = T2 + (int64_t)l * M * 256;
    float tot = 0.f;
    for (int m = 0; m < M; m++) {
        float v = fabsf(t[m * 256 + tid]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
        __syncthreads();
        if ((tid & 63) == 0) s_w[tid >> 6] = v;
        __syncthreads();
        tot += fmaxf(fmaxf(s_w[0], s_w[1]), fmaxf(s_w[2], s_w[3]));
    }
    if (tid == 0) t2max[l] = tot;
}
void launch_t2_rowmax(hipStream_t s, const float* T2, int nlist, int M, float* t2max) {
    if (nlist > 0) hipLaunchKernelGGL(k_t2_rowmax, dim3(nlist), dim3(256), 0, s, T2, M, t2max);
}

// ------------------------------------------------------------------------------------
// k-means update (gamma_hip_train.cpp; compute_centroids, faiss:Clustering.cpp:138-208): cluster c = the points
// order[seg[c] .. seg[c + 1]) in ascending point order; one float accumulator per (cluster, dimension) adds them in that
// order, then c[j] *= 1 / count.  An empty cluster's centroid is zero (the host re-seeds it, split_clusters).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_centroid_update(const float* __restrict__ x, int d, const int* __restrict__ order,
                                                         const int* __restrict__ seg, float* __restrict__ centroids,
                                                         float* __restrict__ hassign) {
    const int c = blockIdx.x;
    const int b = seg[c], e = seg[c + 1];
    const float cnt = (float)(e - b);
    for (int j = threadIdx.x; j < d; j += 128) {
        float acc = 0.f;
        for (int m = b; m < e; m++) acc += x[(int64_t)order[m] * d + j];
        if (e > b) {
            const float norm = 1 / cnt;
            acc *= norm;
        }
        centroids[(int64_t)c * d + j] = acc;
    }
    if (threadIdx.x == 0) hassign[c] = cnt;
}
void launch_centroid_update(hipStream_t s, const float* x, int d, const int* order, const int* seg, int k, float* centroids,
                            float* hassign) {
    if (k > 0) hipLaunchKernelGGL(k_centroid_update, dim3(k), dim3(128), 0, s, x, d, order, seg, centroids, hassign);
}

// ------------------------------------------------------------------------------------
// a12 (Add path): residual + PQ encode.  assign comes from the coarse kernels + select.
//   code[m] = argmin_j fvec_L2sqr_ny(residual_m, c_mj)   (strict <, first minimum,
//   faiss:impl/ProductQuantizer.cpp:321-348).  grid = (M, n), block = 256 = ksub.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pq_encode(const float* __restrict__ x, int d, int M, int dsub,
                                                   const int* __restrict__ assign,
                                                   const float* __restrict__ cc,
                                                   const float* __restrict__ pqc,
                                                   uint8_t* __restrict__ codes) {
    __shared__ float s_res[64];
    __shared__ unsigned long long s_best[4];
    const int m = blockIdx.x, i = blockIdx.y, j = threadIdx.x;
    const int l = assign[i];
    if (j < dsub) {
        float xv = x[(int64_t)i * d + m * dsub + j];
        s_res[j] = l < 0 ? 0.f : xv - cc[(int64_t)l * d + m * dsub + j];
    }
    __syncthreads();
    const float* c = pqc + ((int64_t)m * 256 + j) * dsub;
    float dis = fvec_ny_row<true>(s_res, c, dsub);
    if (!(dis < 1e20f)) dis = INFINITY;  // reference never picks dis >= 1e20 (mindis init)
    // argmin with first-index tie rule: min over (key(dis), j)
    unsigned long long item = ((unsigned long long)f2key(dis) << 32) | (unsigned)j;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(item, off, 64);
        if (o < item) item = o;
    }
    if ((j & 63) == 0) s_best[j >> 6] = item;
    __syncthreads();
    if (j == 0) {
        unsigned long long b = s_best[0];
        for (int w = 1; w < 4; w++) if (s_best[w] < b) b = s_best[w];
        int best = (int)(uint32_t)b;
        if (key2f((uint32_t)(b >> 32)) == INFINITY) best = 0;  // idxm initial value
        codes[(int64_t)i * M + m] = (uint8_t)best;
    }
}
void launch_pq_encode(hipStream_t s, const float* x, int64_t n, int d, int M, const int* assign,
                      const float* cc, const float* pqc, uint8_t* codes) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pq_encode, dim3(M, (unsigned)n), dim3(256), 0, s, x, d, M, d / M, assign, cc,
                       pqc, codes);
}

}  // namespace gh
