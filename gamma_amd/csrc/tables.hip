// tables.hip -- a4 / a5 front end of the list scan: the per-query inner-product tables (fvec_inner_products_ny order), the
// precomputed table T2, slab offsets of the (query, probe) pairs, the compacted probe lists of a shard, the spatial query
// order, <x, centroid> of the inner-product scan.
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
    // (the sub-vector of the NEXT query is requested before this query's products: one scalar-load latency per query
    //  was most of the kernel -- 73.6 us for 268 MB of stores that a plain fill writes in 39)
    float xn[DSUB];
    {
        const float* xs = x + (int64_t)min(q0, nq - 1) * d + m * DSUB;   // wave-uniform
#pragma unroll
        for (int t = 0; t < DSUB; t++) xn[t] = xs[t];
    }
    for (int u = 0; u < IPT4_QB; u++) {
        const int q = q0 + u;
        if (q >= nq) break;                                          // uniform
        float xv[DSUB];
#pragma unroll
        for (int t = 0; t < DSUB; t++) xv[t] = xn[t];
        {
            const float* xs = x + (int64_t)min(q + 1, nq - 1) * d + m * DSUB;
#pragma unroll
            for (int t = 0; t < DSUB; t++) xn[t] = xs[t];
        }
        float4 o;
        o.x = fvec_ny_row<false>(xv, c[0], DSUB);
        o.y = fvec_ny_row<false>(xv, c[1], DSUB);
        o.z = fvec_ny_row<false>(xv, c[2], DSUB);
        o.w = fvec_ny_row<false>(xv, c[3], DSUB);
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v ov = {o.x, o.y, o.z, o.w};
        __builtin_nontemporal_store(ov, reinterpret_cast<f4v*>(out + ((int64_t)q * M + m) * 256 + 4 * lane));
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
                                                        float* __restrict__ cdis_out, int P_out) {
    // P_out <= P: rows of the OUTPUT (two-phase shard search: the longest run of owned probes any query of the batch has,
    // measured first -- k_max_local_total -- so that the scan behind sees a search of P_out probes, all of them dense)
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
            if (at < P_out) {   // (never false when P_out came from k_max_local_total of this batch)
                probe_out[(int64_t)q * P_out + at] = l;
                cdis_out[(int64_t)q * P_out + at] = cd;
            }
        }
        nown += __popcll(bal);
    }
    for (int p = nown + lane; p < P_out; p += 64) {
        probe_out[(int64_t)q * P_out + p] = -1;
        cdis_out[(int64_t)q * P_out + p] = 0.f;
    }
}
void launch_compact_probes(hipStream_t s, const int* probe_in, const float* cdis_in, int nq, int P,
                           const int* list_len, const uint8_t* list_mask, int nlist, int* probe_out,
                           float* cdis_out, int P_out) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_compact_probes, dim3((nq + 3) / 4), dim3(256), 0, s, probe_in, cdis_in, nq, P, list_len,
                       list_mask, nlist, probe_out, cdis_out, P_out > 0 ? std::min(P_out, P) : P);
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
// list shard over a supplied assignment: the longest candidate row any query of the batch gets from the lists this
// handle scans (its slab stride; the host reads it back before it sizes the chunks of the call)
__global__ __launch_bounds__(256) void k_max_local_total(const int* __restrict__ probe_list, int nq, int P,
                                                         const int* __restrict__ list_len,
                                                         const uint8_t* __restrict__ list_mask, int nlist,
                                                         int* __restrict__ out_max) {
    // out_max[0]: the longest candidate row; out_max[1]: the most owned, non-empty probes any query has (the rows'
    // length after k_compact_probes).  One wave per query, the workgroup's four results combined in LDS: ONE pair of
    // atomics per workgroup (a wave looping over eight queries behind dependent loads took 200 us per 65536 queries)
    __shared__ int s_t[4], s_n[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int best = 0, bestn = 0;
    for (int q = blockIdx.x * 4 + wv; q < nq; q += gridDim.x * 4) {
        int t = 0, n = 0;
        for (int p = lane; p < P; p += 64) {
            const int l = probe_list[(int64_t)q * P + p];
            if (l >= 0 && l < nlist && (!list_mask || list_mask[l])) {
                const int len = list_len[l];
                t += max(len, 0);
                n += len > 0 ? 1 : 0;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            t += __shfl_xor(t, off, 64);
            n += __shfl_xor(n, off, 64);
        }
        best = max(best, t);
        bestn = max(bestn, n);
    }
    if (lane == 0) {
        s_t[wv] = best;
        s_n[wv] = bestn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        best = max(max(s_t[0], s_t[1]), max(s_t[2], s_t[3]));
        bestn = max(max(s_n[0], s_n[1]), max(s_n[2], s_n[3]));
        if (best > 0) atomicMax(out_max, best);
        if (bestn > 0) atomicMax(out_max + 1, bestn);
    }
}

// Two-phase shard search (round 6): the bound a shard's producers published per query, as a float the caller reduces
// across the shards (min for L2, max for inner product: every shard's value bounds the GLOBAL recall_num-th best from
// the wrong side at worst, so the tightest of them does too), and the reduced value back into the ready words the
// consumers read.  A query without a bound travels as +inf / -inf.
template <bool L2>
__global__ __launch_bounds__(256) void k_bound_export(const unsigned long long* __restrict__ ready, int nq, float* __restrict__ out) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const unsigned long long w = ready[q];
    const uint32_t key = (uint32_t)w;
    out[q] = (w >> 32) == 1ull ? key2f(L2 ? key : ~key) : (L2 ? INFINITY : -INFINITY);
}
template <bool L2>
__global__ __launch_bounds__(256) void k_bound_import(const float* __restrict__ in, int nq, unsigned long long* __restrict__ ready) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const float v = in[q];
    const uint32_t key = L2 ? f2key(v) : ~f2key(v);
    // (never looser than the shard's own bound: the reduction included it)
    ready[q] = (v == v && key < KEY_SENTINEL) ? ((1ull << 32) | key) : (2ull << 32);
}
void launch_bound_export(hipStream_t s, bool l2, const unsigned long long* ready, int nq, float* out) {
    if (nq <= 0) return;
    if (l2) hipLaunchKernelGGL(k_bound_export<true>, dim3((nq + 255) / 256), dim3(256), 0, s, ready, nq, out);
    else hipLaunchKernelGGL(k_bound_export<false>, dim3((nq + 255) / 256), dim3(256), 0, s, ready, nq, out);
}
void launch_bound_import(hipStream_t s, bool l2, const float* in, int nq, unsigned long long* ready) {
    if (nq <= 0) return;
    if (l2) hipLaunchKernelGGL(k_bound_import<true>, dim3((nq + 255) / 256), dim3(256), 0, s, in, nq, ready);
    else hipLaunchKernelGGL(k_bound_import<false>, dim3((nq + 255) / 256), dim3(256), 0, s, in, nq, ready);
}
// st2 rows of the queries in a device list (the repair launch of a call whose tables were computed on the fly)
__global__ __launch_bounds__(256) void k_pq_ip_table_rows(const float* __restrict__ x, int d, int M, const float* __restrict__ pqc,
                                                          float* __restrict__ out, const int* __restrict__ rq_list,
                                                          const int* __restrict__ rq_count) {
    const int n = *rq_count, dsub = d / M, j = threadIdx.x;
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int q = rq_list[w];
        for (int m = 0; m < M; m++)
            out[((int64_t)q * M + m) * 256 + j] = fvec_ny_row<false>(x + (int64_t)q * d + m * dsub, pqc + ((int64_t)m * 256 + j) * dsub, dsub);
    }
}
void launch_pq_ip_table_rows(hipStream_t s, const float* x, int d, int M, const float* pqc, float* out, const int* rq_list,
                             const int* rq_count) {
    hipLaunchKernelGGL(k_pq_ip_table_rows, dim3(1024), dim3(256), 0, s, x, d, M, pqc, out, rq_list, rq_count);
}
__global__ __launch_bounds__(256) void k_bound_combine(float* __restrict__ acc, const float* __restrict__ in, int n, int take_max) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] = take_max ? fmaxf(acc[i], in[i]) : fminf(acc[i], in[i]);
}
void launch_bound_combine(hipStream_t s, float* acc, const float* in, int n, int take_max) {
    if (n > 0) hipLaunchKernelGGL(k_bound_combine, dim3((n + 255) / 256), dim3(256), 0, s, acc, in, n, take_max);
}
__global__ __launch_bounds__(256) void k_fill_f32(float* __restrict__ p, int n, float v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_f32(hipStream_t s, float* p, int n, float v) {
    if (n > 0) hipLaunchKernelGGL(k_fill_f32, dim3((n + 255) / 256), dim3(256), 0, s, p, n, v);
}
void launch_max_local_total(hipStream_t s, const int* probe_list, int nq, int P, const int* list_len,
                            const uint8_t* list_mask, int nlist, int* out_max) {
    (void)hipMemsetAsync(out_max, 0, 2 * sizeof(int), s);
    if (nq > 0)
        hipLaunchKernelGGL(k_max_local_total, dim3(std::min(16384, (nq + 3) / 4)), dim3(256), 0, s, probe_list, nq, P, list_len,
                           list_mask, nlist, out_max);
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

}  // namespace gh
