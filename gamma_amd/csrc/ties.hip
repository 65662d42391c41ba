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
#include "reservoir_dev.h"
#include "tie_dev.h"

namespace gh {

int tie_replay_max_k() { return TR_MAXK_BIG; }
int tie_small_max_k() { return TR_MAXK; }
int tie_replay_max_probes() { return TR_MAXP; }
size_t tie_replay_lds_bytes(int R, int k, int P) { return tie_replay_lds_bytes_(R, k, P); }

template <bool L2, int STG, int MAXK = TR_MAXK>
__global__ __launch_bounds__(256) void k_tie_replay(TieReplayArgs a) {
    extern __shared__ __attribute__((aligned(16))) char s_tie_lds[];
    const int nflag = min(*a.count, a.nq);
    for (int fi = blockIdx.x; fi < nflag; fi += gridDim.x)
        tie_replay_query<L2, 256, TR_SLAB, STG, MAXK>(a, a.list[fi], s_tie_lds, (a.dbg && fi == 0) ? a.dbg : nullptr,
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
    if (a.slice_cap > 2048) {   // callers gate on this
        launch_refused("launch_tie_replay: survivor slices of more than 2048 items");
        return;
    }
    if (a.R > TR_MAXK || a.k > TR_MAXK) {
        // heaps beyond 1024 entries (up to the 4096 the ABI accepts): the same replay with a sort buffer of 4096 items --
        // up to 155 KB of LDS, one workgroup per CU; a sequential heap_reorder of 4096 entries takes milliseconds, which
        // is what a request for thousands of results with a tie at its cut costs
        const size_t lds = tie_replay_lds_bytes_(a.R, a.k, a.P, TR_SLAB, TR_MAXK_BIG);
        static std::atomic<uint64_t> attr{0};   // per device
        if (first_call_on_device(attr)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tie_replay<true, TR_MAXK_BIG, TR_MAXK_BIG>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tie_replay<false, TR_MAXK_BIG, TR_MAXK_BIG>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
        }
        if (l2) hipLaunchKernelGGL((k_tie_replay<true, TR_MAXK_BIG, TR_MAXK_BIG>), dim3(grid), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_tie_replay<false, TR_MAXK_BIG, TR_MAXK_BIG>), dim3(grid), dim3(256), lds, s, a);
        return;
    }
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

// flagged[q] = cut[q] | extra[q] -> list / count (the paths whose final kernels do not list the flagged queries themselves:
// recall_num beyond 1024)
__global__ __launch_bounds__(256) void k_tie_list(const uint8_t* __restrict__ cut, const uint8_t* __restrict__ extra, int nq,
                                                  int* __restrict__ list, int* __restrict__ count,
                                                  unsigned long long* __restrict__ tie_stats) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    if ((cut && cut[q]) || (extra && extra[q])) {
        list[atomicAdd(count, 1)] = q;
        if (tie_stats) atomicAdd(tie_stats + 2, 1ull);
    }
}
void launch_tie_list(hipStream_t s, const uint8_t* cut, const uint8_t* extra, int nq, int* list, int* count,
                     unsigned long long* tie_stats) {
    if (nq > 0) hipLaunchKernelGGL(k_tie_list, dim3((nq + 255) / 256), dim3(256), 0, s, cut, extra, nq, list, count, tie_stats);
}

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ x, const int* __restrict__ list, int d,
                                                     float* __restrict__ out) {
    const int64_t src = list[blockIdx.x];
    for (int j = threadIdx.x; j < d; j += 256) out[(int64_t)blockIdx.x * d + j] = x[src * d + j];
}
void launch_gather_rows(hipStream_t s, const float* x, const int* list, int n, int d, float* out) {
    if (n > 0) hipLaunchKernelGGL(k_gather_rows, dim3(n), dim3(256), 0, s, x, list, d, out);
}

// ------------------------------------------------------------------------------------
// Exact ties across list shards (several GPUs, gamma_hip_group.cpp / DESIGN.md 7).  The owner of a query slice merges
// the shards' top-R tables; a query whose result a tie can change is replayed over the stream the reference's scanner
// saw -- which is spread over the shards: every shard EXPORTS, for the flagged queries, the distances and ids of the
// probed lists it owns in list order (k_shard_export, from the slab of an unbounded scan), the owner assembles the
// exports probe by probe (k_merge_streams) and runs the ordinary replay over the assembled rows.
// ------------------------------------------------------------------------------------
// does the merged top-R cut of a slice query go through a group of equal distances?  One wave per query: entries equal
// to the R-th merged value in all W tables against those that made it into the merged table.  A shard's own cut may
// have dropped members of the group: its table ends at that value and its own selection says the cut went through a tie
// (shard_flags [W][nq], gamma_hip_ivfpq_shard_cut_flags; without them every such table counts).
__global__ __launch_bounds__(256) void k_flag_merge_cut(const float* __restrict__ all_dis, int W, int nq, int R, int q0, int nql,
                                                        const float* __restrict__ merged, const int64_t* __restrict__ merged_ids,
                                                        uint8_t* __restrict__ tcut, const uint8_t* __restrict__ shard_flags) {
    const int lane = threadIdx.x & 63;
    const int ql = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ql >= nql) return;
    if (lane == 0) tcut[ql] = 0;
    if (merged_ids[(int64_t)ql * R + R - 1] < 0) return;   // fewer than R candidates in all: nothing was cut
    const float vk = merged[(int64_t)ql * R + R - 1];
    int in_m = 0, in_all = 0, shard_cut = 0;
    for (int r = lane; r < R; r += 64) in_m += merged[(int64_t)ql * R + r] == vk ? 1 : 0;
    for (int w = 0; w < W; w++) {
        const float* row = all_dis + ((int64_t)w * nq + q0 + ql) * R;
        for (int r = lane; r < R; r += 64) in_all += row[r] == vk ? 1 : 0;
        // the shard's own cut went through a tie (its flag; without flags: assumed) and ended at this very value
        if (lane == 0 && row[R - 1] == vk && (!shard_flags || shard_flags[(int64_t)w * nq + q0 + ql])) shard_cut = 1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        in_m += __shfl_xor(in_m, off, 64);
        in_all += __shfl_xor(in_all, off, 64);
    }
    shard_cut = __shfl(shard_cut, 0, 64);
    if (lane == 0 && (in_all > in_m || shard_cut)) tcut[ql] = 1;
}
void launch_flag_merge_cut(hipStream_t s, const float* all_dis, int W, int nq, int R, int q0, int nql, const float* merged,
                           const int64_t* merged_ids, uint8_t* tcut, const uint8_t* shard_flags) {
    if (nql > 0)
        hipLaunchKernelGGL(k_flag_merge_cut, dim3((nql + 3) / 4), dim3(256), 0, s, all_dis, W, nq, R, q0, nql, merged,
                           merged_ids, tcut, shard_flags);
}

// one workgroup per exported query f: off[f][p] = start of probe p's entries in the row (probes of lists this shard
// does not own are empty); vals = the slab row of the unbounded scan (same order: the shard's compacted probe list keeps
// the order of the original one), ids = the vector ids of those entries
__global__ __launch_bounds__(256) void k_shard_export(const int32_t* __restrict__ probe, int P, const int* __restrict__ list_len,
                                                      const int64_t* __restrict__ list_off, const uint8_t* __restrict__ list_mask,
                                                      int nlist, const int64_t* __restrict__ ids, const float* __restrict__ slab,
                                                      int64_t q_stride, int64_t stride, float* __restrict__ vals,
                                                      int64_t* __restrict__ out_ids, int32_t* __restrict__ off) {
    __shared__ int s_off[TR_MAXP + 1];
    __shared__ int64_t s_base[TR_MAXP];
    const int f = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int at = 0;
        for (int pi = 0; pi < P; pi++) {
            const int l = probe[(int64_t)f * P + pi];
            const bool mine = l >= 0 && l < nlist && (!list_mask || list_mask[l]);
            s_off[pi] = at;
            s_base[pi] = mine ? list_off[l] : 0;
            at += mine ? list_len[l] : 0;
        }
        s_off[P] = at;
    }
    __syncthreads();
    for (int pi = tid; pi <= P; pi += 256) off[(int64_t)f * (P + 1) + pi] = s_off[pi];
    const int n = (int)min((int64_t)s_off[P], stride);
    const float* row = slab + (int64_t)f * q_stride;
    for (int j = tid; j < n; j += 256) {
        int lo = 0, hi = P - 1;
        while (lo < hi) {   // last p with off[p] <= j
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= j) lo = mid; else hi = mid - 1;
        }
        vals[(int64_t)f * stride + j] = row[j];
        out_ids[(int64_t)f * stride + j] = ids[s_base[lo] + (j - s_off[lo])] & 0x7fffffffffffffffLL;
    }
}
// the longest export row of nf queries on this shard: sum of the lengths of the probed lists it owns
__global__ __launch_bounds__(256) void k_shard_export_rows(const int32_t* __restrict__ probe, int nf, int P,
                                                           const int* __restrict__ list_len, const uint8_t* __restrict__ list_mask,
                                                           int nlist, int* __restrict__ max_entries) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    int tot = 0;
    for (int pi = 0; pi < P; pi++) {
        const int l = probe[(int64_t)f * P + pi];
        if (l >= 0 && l < nlist && (!list_mask || list_mask[l])) tot += list_len[l];
    }
    atomicMax(max_entries, tot);
}
void launch_shard_export_rows(hipStream_t s, const int32_t* probe, int nf, int P, const int* list_len, const uint8_t* list_mask,
                              int nlist, int* max_entries) {
    if (nf > 0)
        hipLaunchKernelGGL(k_shard_export_rows, dim3((nf + 255) / 256), dim3(256), 0, s, probe, nf, P, list_len, list_mask, nlist,
                           max_entries);
}
void launch_shard_export(hipStream_t s, const int32_t* probe, int nf, int P, const int* list_len, const int64_t* list_off,
                         const uint8_t* list_mask, int nlist, const int64_t* ids, const float* slab, int64_t q_stride,
                         int64_t stride, float* vals, int64_t* out_ids, int32_t* off) {
    if (nf > 0)
        hipLaunchKernelGGL(k_shard_export, dim3(nf), dim3(256), 0, s, probe, P, list_len, list_off, list_mask, nlist, ids, slab,
                           q_stride, stride, vals, out_ids, off);
}

// one workgroup per flagged query f: the W exports [W][nf][stride] / [W][nf][P + 1] -> ONE row in probe order
// of mstride entries (m_off[f][P + 1], m_base[f][p] = f * mstride + m_off[f][p]: the replay's pair_off / pair_base over m_ids)
__global__ __launch_bounds__(256) void k_merge_streams(int W, int nf, int P, int64_t stride, int64_t mstride, const float* __restrict__ vals,
                                                       const int64_t* __restrict__ ids, const int32_t* __restrict__ off,
                                                       float* __restrict__ m_vals, int64_t* __restrict__ m_ids,
                                                       int32_t* __restrict__ m_off, int64_t* __restrict__ m_base, float sentinel) {
    __shared__ int s_moff[TR_MAXP + 1];
    __shared__ int s_src[TR_MAXP];    // the shard that holds probe p's list (or -1)
    __shared__ int s_soff[TR_MAXP];   // where its entries start in that shard's row
    const int f = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int at = 0;
        for (int pi = 0; pi < P; pi++) {
            int src = -1, so = 0, len = 0;
            for (int w = 0; w < W; w++) {
                const int32_t* o = off + ((int64_t)w * nf + f) * (P + 1);
                const int lw = o[pi + 1] - o[pi];
                if (lw > 0) {
                    src = w;
                    so = o[pi];
                    len = lw;
                }
            }
            if ((int64_t)at + len > mstride) len = (int)max((int64_t)0, mstride - at);   // cannot happen: mstride = W x stride
            s_moff[pi] = at;
            s_src[pi] = src;
            s_soff[pi] = so;
            at += len;
        }
        s_moff[P] = at;
    }
    __syncthreads();
    for (int pi = tid; pi <= P; pi += 256) m_off[(int64_t)f * (P + 1) + pi] = s_moff[pi];
    for (int pi = tid; pi < P; pi += 256) m_base[(int64_t)f * P + pi] = (int64_t)f * mstride + s_moff[pi];
    const int n = s_moff[P];
    for (int j = tid; j < n; j += 256) {
        int lo = 0, hi = P - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_moff[mid] <= j) lo = mid; else hi = mid - 1;
        }
        const int w = s_src[lo];
        const int64_t at = ((int64_t)w * nf + f) * stride + s_soff[lo] + (j - s_moff[lo]);
        m_vals[(int64_t)f * mstride + j] = w >= 0 ? vals[at] : sentinel;
        m_ids[(int64_t)f * mstride + j] = w >= 0 ? ids[at] : -1;
    }
}
void launch_merge_streams(hipStream_t s, int W, int nf, int P, int64_t stride, int64_t mstride, const float* vals, const int64_t* ids,
                          const int32_t* off, float* m_vals, int64_t* m_ids, int32_t* m_off, int64_t* m_base, float sentinel) {
    if (nf > 0)
        hipLaunchKernelGGL(k_merge_streams, dim3(nf), dim3(256), 0, s, W, nf, P, stride, mstride, vals, ids, off, m_vals, m_ids,
                           m_off, m_base, sentinel);
}

// rows of `words` 32-bit words: out[i] = src[list[i]]
__global__ __launch_bounds__(256) void k_gather_words(const uint32_t* __restrict__ src, const int* __restrict__ list, int words,
                                                      uint32_t* __restrict__ out) {
    const int64_t r = list[blockIdx.x];
    for (int j = threadIdx.x; j < words; j += 256) out[(int64_t)blockIdx.x * words + j] = src[r * words + j];
}
void launch_gather_words(hipStream_t s, const void* src, const int* list, int n, int words, void* out) {
    if (n > 0)
        hipLaunchKernelGGL(k_gather_words, dim3(n), dim3(256), 0, s, static_cast<const uint32_t*>(src), list, words,
                           static_cast<uint32_t*>(out));
}


// ------------------------------------------------------------------------------------
// Test hook (gamma_hip_debug_heap_stream): one stream of keys through ONE heap with each form of the sifts, so that the
// device's heaps can be compared with the oracle's (= the compiled library's) entry for entry.
//   op 0: heap_replace_top stream through the pipelined HeapWalk          -> the heap ARRAY, then heap_reorder
//   op 1: heap_pop + heap_push stream through ParHeap (all lanes per sift) -> likewise
//   op 2: heap_pop + heap_push stream through the sequential forms
//   op 3: heap_replace_top stream through ParHeap's sift
//   op 4: the stream through faiss's ReservoirTopN (n >= 1; out_arr = out_sorted = to_result's output)
// out_arr: the array when the stream is through (k entries: value, payload); out_sorted: after heap_reorder.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_debug_heap_stream(int op, int k, int n, const float* __restrict__ vals,
                                                          uint2* __restrict__ out_arr, uint2* __restrict__ out_sorted) {
    __shared__ __attribute__((aligned(16))) uint2 h[TR_MAXK + 2];
    const int lane = threadIdx.x;
    if (op == 4) {   // faiss's ReservoirTopN (reservoir_dev.h): there is no array to show, only to_result's output
        __shared__ float s_v[reservoir_capacity(TR_MAXK)];
        __shared__ int s_i[reservoir_capacity(TR_MAXK)];
        (void)reservoir_row(vals, n, k, s_v, s_i, h);
        for (int i = lane; i < k; i += 64) {
            out_arr[i] = h[1 + i];
            out_sorted[i] = h[1 + i];
        }
        return;
    }
    heap_fill(h, k, lane, 64);
    __builtin_amdgcn_wave_barrier();
    HeapWalk w;
    w.begin(h, k);
    float top = kHeapFltMax;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const float dv = j < n ? vals[j] : INFINITY;
        if (op == 0) {
            w.accept(j < n, dv, j);
            continue;
        }
        unsigned long long m = __ballot(top > dv);
        while (m) {
            const int l = (int)__ffsll((long long)m) - 1;
            const float val = hw_readlane_f(dv, l);
            if (op == 1) {
                const float root = par_heap_pop(h, k);
                top = par_heap_push(h, k, val, (unsigned)(j0 + l)) ? val : root;
            } else if (op == 2) {
                heap_pop_seq(h, k);
                heap_push_seq(h, k, val, (unsigned)(j0 + l));
                top = hs_f(h[1].x);
            } else {
                top = par_heap_replace_top(h, k, val, (unsigned)(j0 + l));
            }
            const unsigned long long above = l >= 63 ? 0ull : (~0ull << (l + 1));
            m = __ballot(top > dv) & above;
        }
    }
    if (op == 0) w.drain();
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < k; i += 64) out_arr[i] = h[1 + i];
    __builtin_amdgcn_wave_barrier();
    if (op == 2) heap_reorder_seq(h, k);
    else par_heap_reorder(h, k);
    for (int i = lane; i < k; i += 64) out_sorted[i] = h[1 + i];
}
void launch_debug_heap_stream(hipStream_t s, int op, int k, int n, const float* vals, uint2* out_arr, uint2* out_sorted) {
    hipLaunchKernelGGL(k_debug_heap_stream, dim3(1), dim3(64), 0, s, op, k, n, vals, out_arr, out_sorted);
}

}  // namespace gh
