// ties.hip -- the reference's behaviour inside EXACT distance ties (gamma_hip_set_exact_ties).
//
// Everywhere else in this library a top-k is "the k smallest (distance, scan position) pairs".  The
// reference keeps its candidates in faiss binary heaps (faiss:utils/Heap.h:46-131), and which of several
// candidates at exactly the same fp32 distance survive a boundary -- and in which order equal distances
// come out -- is whatever the heap's sift order leaves:
//   * R-heap of the list scan: KnnSearchResults::add, `if (cmp(top, dis)) heap_replace_top`
//     (index/impl/gamma_index_ivfpq.h:363-369), candidates in probe order then list order;
//   * compute_dis with has_rank (index/impl/gamma_index_ivfpq.cc:646-680): the R-heap is NOT sorted
//     first -- its ARRAY order is the order in which the exact distances enter the k-heap through
//     heap_pop + heap_push; then heap_reorder;
//   * compute_dis without rank (:681-696): heap_reorder of the R-heap, first k inside the score window.
// The array a heap ends with depends on its whole insertion history, so the only way to reproduce it is
// to replay it.  That is what k_tie_replay does, for the few queries the regular kernels flag:
//   - the top-R cut goes through a group of equal ADC distances (flag from the selection kernels), or
//   - two of the first k+1 final distances are equal (flag from k_rerank_topk / k_finalize_norank).
// One workgroup per flagged query.  The candidate stream is rebuilt in scan order from what the scan
// left behind: the ADC distances of the query's first probe group (always stored) followed by the
// survivor slices of the other groups -- a candidate outside its slice was above the bound tau, and the
// reference's heap top was already <= tau when it came by, so it could not have entered the heap -- or,
// for a query without a bound, the whole distance slab.  Wave 0 walks the stream (64 candidates per
// step, ballot of `top > dis`) and sifts every accepted candidate into a heap kept in LDS with the
// reference's own sift code; then the exact distances of the heap's R entries are computed in ARRAY
// order and pushed through the k-heap the same way.
//
// Inner product: the reference's CMin heap on v is the CMax heap on -v (negation is exact), so one
// heap implementation serves both metrics; values are negated on the way in and out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "block_utils.h"
#include "device_math.h"
#include "kernels.h"
#include "rerank_dev.h"
#include "tie_dev.h"

namespace gh {

int tie_replay_max_k() { return TR_MAXK; }
int tie_replay_max_probes() { return TR_MAXP; }
size_t tie_replay_lds_bytes(int R, int k, int P) { return tie_replay_lds_bytes_(R, k, P); }

template <bool L2, int STG>
__global__ __launch_bounds__(256) void k_tie_replay(TieReplayArgs a) {
    extern __shared__ __attribute__((aligned(16))) char s_tie_lds[];
    const int nflag = min(*a.count, a.nq);
    for (int fi = blockIdx.x; fi < nflag; fi += gridDim.x)
        tie_replay_query<L2, 256, TR_SLAB, STG>(a, a.list[fi], s_tie_lds, (a.dbg && fi == 0) ? a.dbg : nullptr,
                                                 a.compact_rows ? fi : -1);
}

void launch_tie_replay(hipStream_t s, bool l2, const TieReplayArgs& a0) {
    if (a0.nq <= 0) return;
    TieReplayArgs a = a0;
    static const bool want_dbg = getenv("GAMMA_HIP_TIE_DBG") != nullptr;
    static unsigned long long* dbg = nullptr;
    static int shown = 0;
    if (want_dbg) {
        if (!dbg) (void)hipMalloc((void**)&dbg, 64);
        if (shown++ % 10 == 9) {
            unsigned long long h[8];
            int n = 0;
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
            (void)hipMemcpy(&n, a.count, sizeof(int), hipMemcpyDeviceToHost);
            fprintf(stderr, "tie replay (10 ns ticks, previous call): stage %llu slab walk %llu slices %llu ids+exact %llu k-heap %llu out %llu; slab %llu candidates, %llu taken; %d queries flagged now\n",
                    h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] >> 32, h[7] & 0xffffffffull, n);
        }
        a.dbg = dbg;
    }
    const int grid = std::min(a.nq, 1024);
    if (a.slice_cap > 2048) abort();   // callers gate on this
    if (a.slice_cap > TR_STAGE) {      // long slices (flat search: the candidate lists of the running bound)
        const size_t lds = tie_replay_lds_bytes_(a.R, a.k, a.P, TR_SLAB, 2048);
        if (l2) hipLaunchKernelGGL((k_tie_replay<true, 2048>), dim3(grid), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_tie_replay<false, 2048>), dim3(grid), dim3(256), lds, s, a);
        return;
    }
    const size_t lds = tie_replay_lds_bytes(a.R, a.k, a.P);
    if (l2) hipLaunchKernelGGL((k_tie_replay<true, TR_STAGE>), dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((k_tie_replay<false, TR_STAGE>), dim3(grid), dim3(256), lds, s, a);
}

// ------------------------------------------------------------------------------------
// Does the top-K cut of a query go through a group of equal distances?  (unfiltered selection paths:
// their kernels return exactly K items.)  One wave per query: count the slab entries equal to the K-th
// selected value, compare with how many of them were selected.  only != nullptr: rows with only[q] != 0.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flag_cut_ties(const float* __restrict__ slab, int64_t q_stride,
                                                       const int* __restrict__ q_total, int nq, int K,
                                                       const float* __restrict__ sel_vals,
                                                       const int* __restrict__ sel_pos,
                                                       const uint8_t* __restrict__ only,
                                                       uint8_t* __restrict__ tflag, int fixed_n, int inside) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq || (only && !only[q])) return;
    if (inside) {   // two equal values among the selected K (coarse quantizer: the order of the probes)
        bool eq = false;
        for (int r = lane; r + 1 < K; r += 64)
            eq |= sel_pos[(int64_t)q * K + r + 1] >= 0 && sel_vals[(int64_t)q * K + r] == sel_vals[(int64_t)q * K + r + 1];
        if (__ballot(eq)) {
            if (lane == 0) tflag[q] = 1;
            return;
        }
    }
    if (sel_pos[(int64_t)q * K + K - 1] < 0) return;   // fewer than K candidates: nothing was cut
    const float vk = sel_vals[(int64_t)q * K + K - 1];
    int in_sel = 0, in_all = 0;
    for (int r = lane; r < K; r += 64) in_sel += sel_vals[(int64_t)q * K + r] == vk ? 1 : 0;
    const float* v = slab + (int64_t)q * q_stride;
    const int n = q_total ? q_total[q] : fixed_n;
    for (int j0 = 0; j0 < n; j0 += 64 * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = v[min(j0 + u * 64 + lane, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) in_all += (j0 + u * 64 + lane < n && t[u] == vk) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        in_sel += __shfl_xor(in_sel, off, 64);
        in_all += __shfl_xor(in_all, off, 64);
    }
    if (lane == 0 && in_all > in_sel) tflag[q] = 1;
}
void launch_flag_cut_ties(hipStream_t s, const float* slab, int64_t q_stride, const int* q_total, int nq, int K,
                          const float* sel_vals, const int* sel_pos, const uint8_t* only, uint8_t* tflag, int fixed_n,
                          int inside) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_flag_cut_ties, dim3((nq + 3) / 4), dim3(256), 0, s, slab, q_stride, q_total, nq, K,
                       sel_vals, sel_pos, only, tflag, fixed_n, inside);
}

// ------------------------------------------------------------------------------------
// flat search, chunked paths under exact ties (gamma_hip_search.cpp): D1/I1 [nq][k + 1] sorted on (distance, row id).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flat_take_flag(const float* __restrict__ D1, const int64_t* __restrict__ I1, int nq,
                                                        int k, float* __restrict__ distances, int64_t* __restrict__ labels,
                                                        int* __restrict__ list, int* __restrict__ count,
                                                        unsigned long long* __restrict__ tie_stats) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float* dv = D1 + (int64_t)q * (k + 1);
    const int64_t* iv = I1 + (int64_t)q * (k + 1);
    bool eq = false;
    for (int i = lane; i < k; i += 64) {
        distances[(int64_t)q * k + i] = dv[i];
        labels[(int64_t)q * k + i] = iv[i];
        eq |= iv[i] >= 0 && iv[i + 1] >= 0 && dv[i] == dv[i + 1];
    }
    if (__ballot(eq) && lane == 0) {
        list[atomicAdd(count, 1)] = q;
        if (tie_stats) atomicAdd(tie_stats + 2, 1ull);
    }
}
void launch_flat_take_flag(hipStream_t s, const float* D1, const int64_t* I1, int nq, int k, float* distances, int64_t* labels,
                           int* list, int* count, unsigned long long* tie_stats) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_flat_take_flag, dim3((nq + 3) / 4), dim3(256), 0, s, D1, I1, nq, k, distances, labels, list, count,
                       tie_stats);
}

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ x, const int* __restrict__ list, int d,
                                                     float* __restrict__ out) {
    const int64_t src = list[blockIdx.x];
    for (int j = threadIdx.x; j < d; j += 256) out[(int64_t)blockIdx.x * d + j] = x[src * d + j];
}
void launch_gather_rows(hipStream_t s, const float* x, const int* list, int n, int d, float* out) {
    if (n > 0) hipLaunchKernelGGL(k_gather_rows, dim3(n), dim3(256), 0, s, x, list, d, out);
}

}  // namespace gh
