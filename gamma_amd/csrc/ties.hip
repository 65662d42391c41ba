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

#include "block_utils.h"
#include "device_math.h"
#include "kernels.h"
#include "rerank_dev.h"

namespace gh {

namespace {

constexpr float kFltMax = 3.402823466e+38f;

// ---- faiss:utils/Heap.h, CMax (cmp(a, b) = a > b), 1-based arrays v1 / id1 (= array - 1) in LDS.  Called
//      by ONE lane; plain sequential code. ----
__device__ __forceinline__ void lh_sift_down(float* v1, int* id1, int k, float val, int id) {
    int i = 1;
    for (;;) {
        const int i1 = i << 1, i2 = i1 + 1;
        if (i1 > k) break;
        const float c1 = v1[i1];
        const float c2 = i2 <= k ? v1[i2] : 0.f;
        if (i2 == k + 1 || c1 > c2) {
            if (val > c1) break;
            v1[i] = c1;
            id1[i] = id1[i1];
            i = i1;
        } else {
            if (val > c2) break;
            v1[i] = c2;
            id1[i] = id1[i2];
            i = i2;
        }
    }
    v1[i] = val;
    id1[i] = id;
}
// heap_pop: the last element sifts down from the root; slot k is left stale (Heap.h:46-72)
__device__ __forceinline__ void lh_pop(float* v1, int* id1, int k) { lh_sift_down(v1, id1, k, v1[k], id1[k]); }
// heap_push into slot k (Heap.h:77-100)
__device__ __forceinline__ void lh_push(float* v1, int* id1, int k, float val, int id) {
    int i = k;
    while (i > 1) {
        const int f = i >> 1;
        if (!(val > v1[f])) break;
        v1[i] = v1[f];
        id1[i] = id1[f];
        i = f;
    }
    v1[i] = val;
    id1[i] = id;
}
// heap_reorder (Heap.h:300-330) on 0-based arrays v / id: sorted best first, (FLT_MAX, -1) padded.
// Returns the number of real entries.
__device__ __forceinline__ int lh_reorder(float* v, int* id, int k) {
    int ii = 0;
    for (int i = 0; i < k; i++) {
        const float val = v[0];
        const int idv = id[0];
        lh_pop(v - 1, id - 1, k - i);
        v[k - ii - 1] = val;
        id[k - ii - 1] = idv;
        if (idv != -1) ii++;
    }
    for (int i = 0; i < ii; i++) {   // memmove to the front (ascending copy is safe: dst < src)
        v[i] = v[k - ii + i];
        id[i] = id[k - ii + i];
    }
    for (int i = ii; i < k; i++) {
        v[i] = kFltMax;
        id[i] = -1;
    }
    return ii;
}

}  // namespace

constexpr int TR_MAXK = 256;     // heap sizes the replay covers (recall_num and k)
constexpr int TR_MAXP = 256;     // probes per query
constexpr int TR_STAGE = 1024;   // survivor items sorted per round (= the scan's slice capacity)

int tie_replay_max_k() { return TR_MAXK; }
int tie_replay_max_probes() { return TR_MAXP; }

template <bool L2>
__global__ __launch_bounds__(256) void k_tie_replay(TieReplayArgs a) {
    __shared__ float s_hv[TR_MAXK];               // R-heap, array order
    __shared__ int s_hp[TR_MAXK];                 //   position in the query's segment, -1 = empty
    __shared__ float s_kv[TR_MAXK];               // k-heap
    __shared__ int s_ki[TR_MAXK];                 //   slot of the R-heap array the entry came from
    __shared__ int64_t s_id[TR_MAXK];             // vector id of R-heap slot j (array order)
    __shared__ float s_ex[TR_MAXK];               // exact distance of slot j
    __shared__ unsigned long long s_it[TR_STAGE]; // survivors of some slices, sorted by position
    __shared__ int s_off[TR_MAXP + 1];
    __shared__ int64_t s_base[TR_MAXP];
    __shared__ float s_top;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int R = a.R, k = a.k, P = a.P;
    const int nflag = min(*a.count, a.nq);
    for (int fi = blockIdx.x; fi < nflag; fi += gridDim.x) {
        const int q = a.list[fi];
        __syncthreads();   // the previous query's LDS state has been consumed
        for (int i = tid; i <= P; i += 256) s_off[i] = a.pair_off[(int64_t)q * (P + 1) + i];
        for (int i = tid; i < P; i += 256) s_base[i] = a.pair_base[(int64_t)q * P + i];
        for (int i = tid; i < R; i += 256) {
            s_hv[i] = kFltMax;   // heap_heapify: (neutral, -1)
            s_hp[i] = -1;
        }
        if (tid == 0) s_top = kFltMax;
        __syncthreads();
        // ---- the candidate stream, in scan order ----
        // bounded query (the scan published a bound and no slice overflowed): first probe group from the
        // slab, the other groups from their survivor slices; otherwise the whole slab
        bool sliced = false;
        if (a.ready) {
            sliced = (a.ready[q] >> 32) == 1ull;
            for (int s = 0; s < a.nsl && sliced; s++)
                if (a.gcnt[(int64_t)q * a.nsl + s] > a.slice_cap) sliced = false;   // uniform
        }
        const int ntot = s_off[P];
        const int n_slab = sliced ? s_off[min(a.G, P)] : ntot;
        const float* slab = a.slab + (int64_t)q * a.q_stride;
        // accept() : one block of <= 64 candidates held one per lane of wave 0 (dv = value in "smaller is
        // better" form, ps = position); every candidate the heap's top does not beat is skipped in one
        // ballot, the others are sifted in one at a time, in order
        auto accept = [&](bool ok, float dv, int ps) {
            float top = s_top;
            int from = 0;
            for (;;) {
                unsigned long long m = __ballot(ok && top > dv);
                m &= from >= 64 ? 0ull : (~0ull << from);
                if (m == 0ull) break;
                const int l = (int)__ffsll((long long)m) - 1;
                const float val = __shfl(dv, l, 64);
                const int pv = __shfl(ps, l, 64);
                if (lane == 0) {
                    lh_sift_down(s_hv - 1, s_hp - 1, R, val, pv);   // heap_replace_top
                    s_top = s_hv[0];
                }
                __builtin_amdgcn_wave_barrier();
                top = s_top;
                from = l + 1;
            }
        };
        if (wv == 0) {
            // four blocks of the slab in flight ahead of the walk
            for (int j0 = 0; j0 < n_slab; j0 += 256) {
                float t[4];
#pragma unroll
                for (int u = 0; u < 4; u++) t[u] = slab[min(j0 + u * 64 + lane, n_slab - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + u * 64 + lane;
                    if (j0 + u * 64 < n_slab) accept(j < n_slab, L2 ? t[u] : -t[u], j);   // filtered entries: +inf
                }
            }
        }
        if (sliced) {
            // slices 1.. in order (slice s holds positions of probe group s only, so slices are ordered among
            // themselves); several short slices share one sorting round
            int s = 1;
            while (s < a.nsl) {
                __syncthreads();   // s_it free again
                int n = 0, s_end = s;
                while (s_end < a.nsl) {
                    const int c = a.gcnt[(int64_t)q * a.nsl + s_end];
                    if (n + c > TR_STAGE) break;
                    const unsigned long long* src = a.surv + ((int64_t)q * a.nsl + s_end) * a.slice_cap;
                    for (int i = tid; i < c; i += 256) {
                        const unsigned long long it = src[i];   // (key << 32 | position)
                        s_it[n + i] = (it << 32) | (it >> 32);  // -> (position << 32 | key)
                    }
                    n += c;
                    s_end++;
                }
                block_rank_sort<256, TR_STAGE / 256>(s_it, n);   // positions are distinct
                if (wv == 0) {
                    for (int j0 = 0; j0 < n; j0 += 64) {
                        const unsigned long long it = s_it[min(j0 + lane, n - 1)];
                        const uint32_t key = (uint32_t)it;
                        const float val = key2f(L2 ? key : ~key);
                        accept(j0 + lane < n, L2 ? val : -val, (int)(uint32_t)(it >> 32));
                    }
                }
                s = s_end;
            }
        }
        __syncthreads();
        // ---- the R-heap is final: array order in s_hv / s_hp.  Positions -> vector ids. ----
        auto pos_to_id = [&](int ps) -> int64_t {
            if (ps < 0) return -1;
            int lo = 0, hi = P - 1;
            while (lo < hi) {   // last p with off[p] <= ps
                const int mid = (lo + hi + 1) >> 1;
                if (s_off[mid] <= ps) lo = mid; else hi = mid - 1;
            }
            return a.ids[s_base[lo] + (ps - s_off[lo])] & 0x7fffffffffffffffLL;
        };
        for (int j = tid; j < R; j += 256) s_id[j] = pos_to_id(s_hp[j]);
        __syncthreads();
        float* od = a.distances + (int64_t)q * k;
        int64_t* ol = a.labels + (int64_t)q * k;
        if (a.has_rank) {
            // exact distances in array order: 8 threads per candidate = the 8 AVX lane accumulators
            // (same arithmetic as k_rerank_topk)
            const int l8 = tid & 7, g = tid >> 3;
            const float* xq = a.x + (int64_t)q * a.d;
            for (int j0 = 0; j0 < R; j0 += 32) {
                const int j = j0 + g;
                const int64_t id = j < R ? s_id[j] : -1;
                const bool live = id >= 0 && id < a.nraw;
                const float dis = rerank_dist8<L2>(xq, a.raw + (live ? id : 0) * a.d, a.d, l8, live);
                if (l8 == 0 && j < R) {
                    const bool ok = live && dis <= a.max_score && dis >= a.min_score;   // IsSimilarScoreValid
                    s_ex[j] = ok ? (L2 ? dis : -dis) : INFINITY;   // +inf never beats the heap's top
                }
            }
            for (int i = tid; i < k; i += 256) {
                s_kv[i] = kFltMax;
                s_ki[i] = -1;
            }
            __syncthreads();
            if (wv == 0) {
                float top = kFltMax;
                for (int j0 = 0; j0 < R; j0 += 64) {
                    const int j = j0 + lane;
                    const float dv = j < R ? s_ex[j] : INFINITY;
                    int from = 0;
                    for (;;) {
                        unsigned long long m = __ballot(top > dv);
                        m &= from >= 64 ? 0ull : (~0ull << from);
                        if (m == 0ull) break;
                        const int l = (int)__ffsll((long long)m) - 1;
                        const float val = __shfl(dv, l, 64);
                        if (lane == 0) {   // heap_pop + heap_push (gamma_index_ivfpq.cc:664-676)
                            lh_pop(s_kv - 1, s_ki - 1, k);
                            lh_push(s_kv - 1, s_ki - 1, k, val, j0 + l);
                            s_top = s_kv[0];
                        }
                        __builtin_amdgcn_wave_barrier();
                        top = s_top;
                        from = l + 1;
                    }
                }
                if (lane == 0) lh_reorder(s_kv, s_ki, k);
            }
            __syncthreads();
            for (int i = tid; i < k; i += 256) {
                const int j = s_ki[i];
                od[i] = j < 0 ? a.neutral : (L2 ? s_kv[i] : -s_kv[i]);
                ol[i] = j < 0 ? -1 : s_id[j];
            }
            __syncthreads();
        }
        // the recall-stage table in the reference's order (heap_reorder of the R-heap): stage output of the
        // call, and the result itself without rank
        if (tid == 0) lh_reorder(s_hv, s_hp, R);
        __syncthreads();
        for (int j = tid; j < R; j += 256) {
            const int ps = s_hp[j];
            const int64_t id = pos_to_id(ps);
            s_id[j] = id;
            a.cand_dis[(int64_t)q * R + j] = ps < 0 ? (L2 ? INFINITY : -INFINITY) : (L2 ? s_hv[j] : -s_hv[j]);
            a.cand_ids[(int64_t)q * R + j] = id;
        }
        __syncthreads();
        if (!a.has_rank && wv == 0) {
            // first k entries inside the score window (gamma_index_ivfpq.cc:681-696)
            int taken = 0;
            for (int j0 = 0; j0 < R && taken < k; j0 += 64) {
                const int j = j0 + lane;
                const float dis = j < R ? (L2 ? s_hv[j] : -s_hv[j]) : 0.f;
                const bool ok = j < R && s_hp[j] >= 0 && dis <= a.max_score && dis >= a.min_score;
                const unsigned long long bal = __ballot(ok);
                const int slot = taken + __popcll(bal & ((1ull << lane) - 1ull));
                if (ok && slot < k) {
                    od[slot] = dis;
                    ol[slot] = s_id[j];
                }
                taken += __popcll(bal);
            }
            for (int i = min(taken, k) + lane; i < k; i += 64) {
                od[i] = a.neutral;
                ol[i] = -1;
            }
        }
    }
}

void launch_tie_replay(hipStream_t s, bool l2, const TieReplayArgs& a) {
    if (a.nq <= 0) return;
    const int grid = std::min(a.nq, 1024);
    if (l2) hipLaunchKernelGGL((k_tie_replay<true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_tie_replay<false>), dim3(grid), dim3(256), 0, s, a);
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
                                                       uint8_t* __restrict__ tflag) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq || (only && !only[q])) return;
    if (sel_pos[(int64_t)q * K + K - 1] < 0) return;   // fewer than K candidates: nothing was cut
    const float vk = sel_vals[(int64_t)q * K + K - 1];
    int in_sel = 0, in_all = 0;
    for (int r = lane; r < K; r += 64) in_sel += sel_vals[(int64_t)q * K + r] == vk ? 1 : 0;
    const float* v = slab + (int64_t)q * q_stride;
    const int n = q_total[q];
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
                          const float* sel_vals, const int* sel_pos, const uint8_t* only, uint8_t* tflag) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_flag_cut_ties, dim3((nq + 3) / 4), dim3(256), 0, s, slab, q_stride, q_total, nq, K,
                       sel_vals, sel_pos, only, tflag);
}

}  // namespace gh
