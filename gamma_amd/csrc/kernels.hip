// kernels.hip -- exact pairwise distances (a2 for nq < 20 / coarse_mode 0, a10 flat: the first row chunk and small calls) in
// the reference's operation order (device_math.h), the per-document filter bitmap and the per-call list compaction under
// a filter.  The other stages' kernels: gemm.hip (GEMM-form coarse distances), tables.hip (query tables, pair offsets,
// query order), scan.hip (the IVFPQ list scan), rerank.hip (candidates -> results), store_kernels.hip (writers, training),
// select.hip, coarse.hip, ties.hip, ivfflat.hip, flat_mfma.hip.
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

namespace {
thread_local const char* g_launch_refused = nullptr;
}
void launch_refused(const char* what) {
    if (!g_launch_refused) g_launch_refused = what;
}
const char* launch_refused_take() {
    const char* w = g_launch_refused;
    g_launch_refused = nullptr;
    return w;
}


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
                        const int slot = atomicAdd(&em.cnt[(int64_t)(qt + qi) * em.cstride], 1);
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
        default: launch_refused("launch_pairwise_emit: row length without an emitting kernel (pairwise_can_emit)"); return;
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

}  // namespace gh
