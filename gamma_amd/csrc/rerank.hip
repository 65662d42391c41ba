// rerank.hip -- a9: candidate positions -> ids, exact re-rank (compute_dis), the IVFFLAT pair scan, result finalisation.
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
                                                     float* __restrict__ out, const int32_t* __restrict__ slot, int64_t nslot) {
    const int q = blockIdx.x;
    const int l = threadIdx.x & 7, g = threadIdx.x >> 3;
    const float* xq = x + (int64_t)q * d;
    const float sentinel = L2 ? INFINITY : -INFINITY;
    for (int r0 = blockIdx.y * 32; r0 < R; r0 += gridDim.y * 32) {
        const int r = r0 + g;
        int64_t id = -1;
        if (r < R) id = cand_ids[(int64_t)q * R + r];
        if (slot) id = (id >= 0 && id < nslot) ? (int64_t)slot[id] : -1;   // sharded raw store: the row of this vector HERE
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
                        float max_score, float* out, const int32_t* slot, int64_t nslot) {
    if (nq <= 0) return;
    const int gy = (R + 31) / 32;   // 32 candidates (8 lanes each) per workgroup
    if (l2)
        hipLaunchKernelGGL((k_rerank_dist<true>), dim3(nq, gy), dim3(256), 0, s, x, d, raw, nraw, cand_ids,
                           R, min_score, max_score, out, slot, nslot);
    else
        hipLaunchKernelGGL((k_rerank_dist<false>), dim3(nq, gy), dim3(256), 0, s, x, d, raw, nraw,
                           cand_ids, R, min_score, max_score, out, slot, nslot);
}

// Exact distances of the entries of an exported candidate stream that can still be members of the recall_num-heap: ADC value
// within the query's bound and vector held on this shard.  One workgroup per exported row: the entries that qualify are
// compacted 256 at a time (ballot), then scored 32 at a time, 8 lanes each (rerank_dist8: the arithmetic of k_rerank_topk).
template <bool L2>
__global__ __launch_bounds__(256) void k_export_exact(const float* __restrict__ xf, int d, const float* __restrict__ raw,
                                                      const int32_t* __restrict__ slot, int64_t nslot, const float* __restrict__ vals,
                                                      const int64_t* __restrict__ ids, int64_t stride, const int32_t* __restrict__ off,
                                                      int P, const float* __restrict__ bound, float* __restrict__ ex) {
    __shared__ int s_pos[256];
    __shared__ int s_row[256];
    __shared__ int s_n;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l8 = tid & 7, g = tid >> 3;
    const int n = (int)min((int64_t)off[(int64_t)f * (P + 1) + P], stride);
    const float b = bound[f];
    const float* xq = xf + (int64_t)f * d;
    const float qnan = __uint_as_float(0x7fc00000u);
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + tid;
        bool want = false;
        int row = -1;
        if (j < n) {
            const float v = vals[(int64_t)f * stride + j];
            const int64_t id = ids[(int64_t)f * stride + j];
            // (a filtered entry carries the sentinel: never within a finite bound; NaN bound = every entry)
            const bool in = (b != b) || (L2 ? v <= b : v >= b);
            row = (id >= 0 && id < nslot) ? slot[id] : -1;
            want = in && row >= 0 && fabsf(v) != INFINITY;
            if (!want) ex[(int64_t)f * stride + j] = qnan;
        }
        if (tid == 0) s_n = 0;
        __syncthreads();
        const unsigned long long bal = __ballot(want);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (want) {
            const int at = base + __popcll(bal & ((1ull << lane) - 1ull));
            s_pos[at] = j;
            s_row[at] = row;
        }
        __syncthreads();
        const int m = s_n;
        for (int c0 = 0; c0 < m; c0 += 32) {   // uniform trip count
            const int c = c0 + g;
            const bool live = c < m;
            const float dis = rerank_dist8<L2>(xq, raw + (int64_t)(live ? s_row[c] : 0) * d, d, l8, live);
            if (l8 == 0 && live) ex[(int64_t)f * stride + s_pos[c]] = dis;
        }
        __syncthreads();
        (void)wv;
    }
}
void launch_export_exact(hipStream_t s, bool l2, const float* xf, int nf, int d, const float* raw, const int32_t* slot, int64_t nslot,
                         const float* vals, const int64_t* ids, int64_t stride, const int32_t* off, int P, const float* bound, float* ex) {
    if (nf <= 0) return;
    if (l2) hipLaunchKernelGGL((k_export_exact<true>), dim3(nf), dim3(256), 0, s, xf, d, raw, slot, nslot, vals, ids, stride, off, P, bound, ex);
    else hipLaunchKernelGGL((k_export_exact<false>), dim3(nf), dim3(256), 0, s, xf, d, raw, slot, nslot, vals, ids, stride, off, P, bound, ex);
}

// The exact distance that travelled with each candidate of the merged table: every shard's table row is sorted by ADC value, a
// candidate's (ADC value, id) pair sits in exactly one of them -- lower bound on the value, then the run of equal values.
template <bool L2>
__global__ __launch_bounds__(256) void k_lookup_exact(const float* __restrict__ all_dis, const int64_t* __restrict__ all_ids,
                                                      const float* __restrict__ all_exact, int W, int nq, int R, int q0, int nql,
                                                      const float* __restrict__ cand_dis, const int64_t* __restrict__ cand_ids,
                                                      float* __restrict__ cand_exact) {
    const int ql = blockIdx.x;
    if (ql >= nql) return;
    const float sentinel = L2 ? INFINITY : -INFINITY;
    for (int r = threadIdx.x; r < R; r += 256) {
        const int64_t id = cand_ids[(int64_t)ql * R + r];
        const float v = cand_dis[(int64_t)ql * R + r];
        float out = sentinel;
        if (id >= 0) {
            bool found = false;
            for (int w = 0; w < W && !found; w++) {
                const int64_t rb = ((int64_t)w * nq + q0 + ql) * R;
                int lo = 0, hi = R;   // first position whose value is not better than v (rows are best first)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const float t = all_dis[rb + mid];
                    const bool before = all_ids[rb + mid] >= 0 && (L2 ? t < v : t > v);
                    if (before) lo = mid + 1; else hi = mid;
                }
                for (int j = lo; j < R && all_ids[rb + j] >= 0 && all_dis[rb + j] == v; j++)
                    if (all_ids[rb + j] == id) {
                        out = all_exact[rb + j];
                        found = true;
                        break;
                    }
            }
        }
        cand_exact[(int64_t)ql * R + r] = out;
    }
}
void launch_lookup_exact(hipStream_t s, bool l2, const float* all_dis, const int64_t* all_ids, const float* all_exact, int W, int nq, int R,
                         int q0, int nql, const float* cand_dis, const int64_t* cand_ids, float* cand_exact) {
    if (nql <= 0) return;
    if (l2) hipLaunchKernelGGL((k_lookup_exact<true>), dim3(nql), dim3(256), 0, s, all_dis, all_ids, all_exact, W, nq, R, q0, nql, cand_dis, cand_ids, cand_exact);
    else hipLaunchKernelGGL((k_lookup_exact<false>), dim3(nql), dim3(256), 0, s, all_dis, all_ids, all_exact, W, nq, R, q0, nql, cand_dis, cand_ids, cand_exact);
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
    auto put = [&](int r, bool live, float dis) {
        if (l == 0 && r < R) {
            if (!live || !(dis <= max_score && dis >= min_score)) dis = sentinel;
            const uint32_t key = L2 ? f2key(dis) : ~f2key(dis);
            s_it[r] = ((unsigned long long)key << 32) | (unsigned)r;
        }
    };
    if (d == 128 && nraw > 0) {   // (nraw == 0: row 0 of the reserved range may not be mapped yet -- the predicated path below never touches it)
        // d = 128 (C3, C4): the lane's 16 query elements stay in registers for all candidates (they were half of the loads), and TWO
        // candidates per group of 8 lanes are in flight -- the rows are 512 bytes from all over the raw store, the workgroup's seven
        // rounds of dependent row loads were what it waited for.  Same arithmetic as rerank_dist8 (rerank_dev.h): the lane's fma chain
        // over its elements l, l + 8, .., then (a[l] + a[l + 4]), (s0 + s1) + (s2 + s3).
        float xx[16];
#pragma unroll
        for (int u = 0; u < 16; u++) xx[u] = xq[l + 8 * u];
        for (int r0 = 0; r0 < R; r0 += 64) {
            const int ra = r0 + g, rb = r0 + 32 + g;
            const int64_t ida = ra < R ? s_id[ra] : -1, idb = rb < R ? s_id[rb] : -1;
            const bool la = ida >= 0 && ida < nraw, lb = idb >= 0 && idb < nraw;
            const float* va = raw + (la ? ida : 0) * 128 + l;
            const float* vb = raw + (lb ? idb : 0) * 128 + l;
            float fa[16], fb[16];
#pragma unroll
            for (int u = 0; u < 16; u++) fa[u] = va[8 * u];
#pragma unroll
            for (int u = 0; u < 16; u++) fb[u] = vb[8 * u];
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                if (L2) {
                    const float ta = xx[u] - fa[u], tb = xx[u] - fb[u];
                    a = __builtin_fmaf(ta, ta, a);
                    b = __builtin_fmaf(tb, tb, b);
                } else {
                    a = __builtin_fmaf(xx[u], fa[u], a);
                    b = __builtin_fmaf(xx[u], fb[u], b);
                }
            }
            if (!la) a = 0.f;
            if (!lb) b = 0.f;
            const float sa = __shfl_down(a, 4, 8) + a, sb_ = __shfl_down(b, 4, 8) + b;
            const float ta = sa + __shfl_down(sa, 1, 8), tb = sb_ + __shfl_down(sb_, 1, 8);
            put(ra, la, ta + __shfl_down(ta, 2, 8));
            put(rb, lb, tb + __shfl_down(tb, 2, 8));
        }
    } else {
        for (int r0 = 0; r0 < R; r0 += 32) {
            const int r = r0 + g;
            int64_t id = -1;
            if (r < R) id = s_id[r];
            const bool live = id >= 0 && id < nraw;
            put(r, live, rerank_dist8<L2>(xq, raw + (live ? id : 0) * d, d, l, live));
        }
    }
    if (tf.list && threadIdx.x == 0) s_tie = tf.cut ? tf.cut[q] : 0;
    // Only the first k + 1 items are read below (the k results; whether two of the first k + 1 exact distances are equal).  Up to
    // k + 1 = 64: every 64-item run of the R items is sorted by a wave (DPP bitonic stages), the first k + 1 items of a run get
    // their rank in the whole -- their index in the run + the number of smaller items in every other run (binary searches) --
    // and the items of rank < k + 1 go to their place: the same first k + 1 entries the full rank sort leaves.  (The full sort
    // ranks all R items against all R: 200 LDS reads and 600 vector instructions per thread, ~100 of the kernel's 283 us at C3.)
    const int K1 = min(k + 1, R);
    if (K1 <= 64) {   // (uniform)
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int nruns = (R + 63) >> 6;   // <= 16
        __syncthreads();   // every distance is in s_it
        for (int run = w; run < nruns; run += 4) {
            const int idx = run * 64 + lane;
            s_it[idx] = wave_sort64(idx < R ? s_it[idx] : ~0ull);   // (padding sorts last; s_it holds 1024 items)
        }
        __syncthreads();
        unsigned long long xr[4];
        int rkk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int run = w + 4 * u;
            xr[u] = ~0ull;
            rkk[u] = K1;
            if (run < nruns) {   // (uniform)
                const unsigned long long x = s_it[run * 64 + lane];
                int rank = lane;
                if (lane < K1 && x != ~0ull) {
                    for (int o = 0; o < nruns; o++) {
                        if (o == run) continue;
                        const unsigned long long* ro = s_it + o * 64;
                        int lo = 0, n2 = 64;
#pragma unroll
                        for (int st = 0; st < 7; st++) {   // number of items of run o smaller than x
                            if (n2 > 0) {
                                const int half = n2 >> 1;
                                if (ro[lo + half] < x) {
                                    lo += half + 1;
                                    n2 -= half + 1;
                                } else {
                                    n2 = half;
                                }
                            }
                        }
                        rank += lo;
                        if (rank >= K1) break;
                    }
                    xr[u] = x;
                    rkk[u] = rank;
                }
            }
        }
        __syncthreads();   // every search has read the runs
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (rkk[u] < K1) s_it[rkk[u]] = xr[u];
        __syncthreads();
    } else {
        block_rank_sort<256, 4>(s_it, R);   // R distinct items (the rank field differs)
    }
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

}  // namespace gh
