// reservoir_dev.h -- faiss's ReservoirTopN (faiss:impl/ResultHandler.h:131-187) replayed on the device, CMax form: what
// knn_L2sqr collects the coarse assignment through from 100 probes on (faiss:utils/distances.cpp:341-358) instead of
// HeapResultHandler.  With distinct keys both return the same sorted list; inside exact ties WHICH tied centroids are
// probed and in which order is the doing of the reservoir: candidates below the threshold are appended to an array of
// (2 K + 15) & ~15 slots; a full array is shrunk by partition_fuzzy_median3 (faiss:utils/partitioning.cpp:119-215:
// bisection on sampled medians of three, then a stable compaction that keeps the first few entries equal to the
// threshold) to between K and (capacity + K) / 2 entries; to_result pushes the first K entries into a heap,
// heap_addn's the rest and reorders.  The array a reservoir ends with depends on its whole history, so a row whose
// selection can be changed by ties is replayed in full, one wave per row: the 64 lanes test 64 candidates per ballot,
// count / sample / compact 64 slots per step; the heap phases are heap_dev.h's ParHeap (all lanes per sift).
// oracle/gamma_oracle.c (go_reservoir_stream) is the CPU restatement, pinned against the compiled library.
#pragma once
#include "heap_dev.h"

namespace gh {

constexpr int RV_MIN_K = 100;   // distance_compute_min_k_reservoir (faiss:utils/distances.cpp:306)
__host__ __device__ constexpr int reservoir_capacity(int K) { return (2 * K + 15) & ~15; }   // ResultHandler.h:211

__device__ __forceinline__ float rv_median3(float a, float b, float c) {   // partitioning.cpp:29-40
    if (a > b) {
        const float t = a;
        a = b;
        b = t;
    }
    if (c > b) return b;
    if (c > a) return c;
    return a;
}

// partition_fuzzy_median3<CMax> on (sv, si)[0..n): q_min = K, q_max.  Returns the new threshold, *q_out entries stay.
// All 64 lanes call it together; every result is wave-uniform.
__device__ __forceinline__ float rv_partition_fuzzy(float* sv, int* si, int n, int q_min, int q_max, int* q_out) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    float thresh_inf = -kHeapFltMax, thresh_sup = kHeapFltMax;
    float thresh = rv_median3(hs_f(__float_as_uint(sv[0])), hs_f(__float_as_uint(sv[n / 2])), hs_f(__float_as_uint(sv[n - 1])));
    int n_lt = 0, n_eq = 0, q = 0;
    for (int it = 0; it < 200; it++) {
        n_lt = n_eq = 0;   // count_lt_and_eq
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const float v = i < n ? sv[i] : 0.f;
            n_lt += __popcll(__ballot(i < n && thresh > v));
            n_eq += __popcll(__ballot(i < n && !(thresh > v) && v == thresh));
        }
        if (n_lt <= q_min) {
            if (n_lt + n_eq >= q_min) {
                q = q_min;
                break;
            }
            thresh_inf = thresh;
        } else if (n_lt <= q_max) {
            q = n_lt;
            break;
        } else {
            thresh_sup = thresh;
        }
        // sample_threshold_median3: the first three values strictly between the bounds, visited at (i * 6700417) % n
        float val3[3] = {0.f, 0.f, 0.f};
        int vi = 0;
        for (int i0 = 0; i0 < n && vi < 3; i0 += 64) {
            const unsigned i = (unsigned)(i0 + lane);
            const bool in = (int)i < n;
            const float v = in ? sv[(unsigned)(((unsigned long long)i * 6700417ull) % (unsigned long long)n)] : 0.f;
            unsigned long long m = __ballot(in && v > thresh_inf && thresh_sup > v);
            while (m && vi < 3) {
                const int l = __ffsll((long long)m) - 1;
                const float pv = hw_readlane_f(v, l);
                if (vi == 0) val3[0] = pv;
                else if (vi == 1) val3[1] = pv;
                else val3[2] = pv;
                vi++;
                m &= m - 1;
            }
        }
        const float new_thresh = vi == 3 ? rv_median3(val3[0], val3[1], val3[2]) : (vi != 0 ? val3[0] : thresh_inf);
        if (new_thresh == thresh_inf) break;   // nothing between the bounds
        thresh = new_thresh;
    }
    int n_eq_1 = q - n_lt;
    if (n_eq_1 < 0) {   // more than q entries at the lower bound
        q = q_min;
        thresh = nextafterf(thresh, -INFINITY);   // C::Crev::nextafter
        n_eq_1 = q;
    }
    // compress_array: stable; the first n_eq_1 entries equal to thresh stay
    int wp = 0, eq_seen = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n;
        const float v = in ? sv[i] : 0.f;
        const int id = in ? si[i] : 0;
        const bool lt = in && thresh > v;
        const bool eq = in && !lt && v == thresh;
        const unsigned long long m_eq = __ballot(eq);
        const bool keep = lt || (eq && eq_seen + __popcll(m_eq & below) < n_eq_1);
        const unsigned long long m_keep = __ballot(keep);
        __builtin_amdgcn_wave_barrier();   // every lane holds its slot before any slot of this step is overwritten
        if (keep) {
            const int p = wp + __popcll(m_keep & below);
            sv[p] = v;
            si[p] = id;
        }
        wp += __popcll(m_keep);
        eq_seen += __popcll(m_eq);
        __builtin_amdgcn_wave_barrier();
    }
    *q_out = q;
    return thresh;
}

// One row of n keys (v[j], id j) through ReservoirTopN<CMax>(K, capacity) and to_result: h[1..K] = the K kept entries
// sorted best first as (key bits, id), (FLT_MAX, -1) padded; returns the number of real ones.  sv / si: capacity slots;
// h: K + 2 entries, 16-byte aligned.  One wave; all lanes call it together.
__device__ __forceinline__ int reservoir_row(const float* __restrict__ v, int n, int K, float* sv, int* si, uint2* h) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int cap = reservoir_capacity(K), q_max = (cap + K) / 2;
    int cnt = 0;
    float thr = kHeapFltMax;
    float nx = v[min(lane, n - 1)];
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const float dv = nx;
        nx = v[min(j + 64, n - 1)];
        bool pend = j < n;
        for (;;) {
            const unsigned long long m = __ballot(pend && thr > dv);   // ReservoirTopN::add: `if (threshold > val)`
            if (!m) break;
            if (cnt == cap) {
                // shrink_fuzzy, then the candidate that found the array full is appended whatever the new threshold
                thr = rv_partition_fuzzy(sv, si, cap, K, q_max, &cnt);
                const int f = __ffsll((long long)m) - 1;
                if (lane == f) {
                    sv[cnt] = dv;
                    si[cnt] = j;
                    pend = false;
                }
                cnt++;
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            const int room = cap - cnt, r = __popcll(m & below);
            if (((m >> lane) & 1ull) && r < room) {
                sv[cnt + r] = dv;
                si[cnt + r] = j;
                pend = false;
            }
            cnt += min(room, __popcll(m));
            __builtin_amdgcn_wave_barrier();
        }
    }
    // to_result (ResultHandler.h:172-186)
    const int m0 = min(cnt, K);
    for (int j = 0; j < m0; j++) (void)par_heap_push(h, j + 1, hs_f(__float_as_uint(sv[j])), hs_u((unsigned)si[j]));
    if (cnt < K) {
        const int real = cnt > 0 ? par_heap_reorder(h, cnt) : 0;
        for (int i = cnt + lane; i < K; i += 64) h[1 + i] = make_uint2(__float_as_uint(kHeapFltMax), 0xffffffffu);
        __builtin_amdgcn_wave_barrier();
        return real;
    }
    for (int j = K; j < cnt; j++) {   // heap_addn (Heap.h:247-260)
        const float x = hs_f(__float_as_uint(sv[j]));
        if (hs_f(h[1].x) > x) (void)par_heap_replace_top(h, K, x, hs_u((unsigned)si[j]));
    }
    return par_heap_reorder(h, K);
}

}  // namespace gh
