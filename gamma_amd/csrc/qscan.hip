// qscan.hip -- fused IVFPQ list scan + k-selection, one 512-thread workgroup per query
// (or per (query, probe range) for small batches) on gfx950.
//
// Replaces, for d = M*dsub <= 128, the chain  LUT build -> per-pair scan -> distance slab in
// HBM -> k-select -> id mapping  (a4..a8 of SURVEY.md §8a;
// gamma_index_ivfpq.h:184-257,575-601,363-369) by ONE kernel whose only HBM traffic is the
// PQ codes (+ ids when a filter needs them) of the probed lists and dsub floats of coarse
// centroid per thread and probe:
//
//   * 512/M threads share sub-quantizer `mm`; each keeps 256*M/512 consecutive PQ centroids of
//     it in registers (d/2 floats), with ||c||^2 and the per-query entry st2 = <x_mm, c>;
//   * per probe the LUT entries  T2 = ||c||^2 + 2<centroid_l,mm , c>  (faiss precompute_table,
//     IndexIVFPQ.cpp:461-479) and  lut = T2 + (-2)*st2  (fvec_madd) are RECOMPUTED from those
//     registers in the reference's exact operation order -- the 64 MB precomputed table and
//     the per-query 16 KB st2 table are never read; the LUT is double-buffered in LDS;
//   * the first codes of a list are prefetched before the LUT arithmetic, so HBM latency hides
//     under ~100 VALU instructions; codes are scanned with LDS gathers and sequential fp32
//     adds (gamma_index_ivfpq.h:591-597);
//   * a candidate survives only if it beats the running threshold tau (the recall_num-th best
//     so far); survivors go to an LDS buffer by wave-aggregated atomics and the buffer is
//     compacted by a bucket select before it could overflow;
//   * the final recall_num survivors are bitonic-sorted on (distance key, scan position) --
//     equal distances in scan order, the deterministic counterpart of the reference heap --
//     and written with their vector ids.
// Inner product: the LUT is st2 itself (list independent), written once per query.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <stdio.h>

#include <algorithm>
#include <vector>

#include "block_utils.h"
#include "device_math.h"
#include "filter_dev.h"
#include "kernels.h"

namespace gh {

namespace {
constexpr int NT = 512;        // threads per workgroup (8 waves)
constexpr int NW = NT / 64;
constexpr int QS_CAPS = 2048;  // survivor buffer entries (16 KB)
constexpr int QS_NB = 2048;    // compaction histogram buckets
constexpr int IPT = QS_CAPS / NT;   // buffer items per thread
constexpr int BPT = QS_NB / NT;     // histogram bins per thread

template <bool SMALLEST>
__device__ __forceinline__ uint32_t qs_key(float v) {
    uint32_t k = f2key(v);
    return SMALLEST ? k : ~k;
}
template <bool SMALLEST>
__device__ __forceinline__ float qs_unkey(uint32_t k) {
    return key2f(SMALLEST ? k : ~k);
}

// exclusive scan over the NT threads of the block; s_w: NW ints.  Two barriers.
__device__ __forceinline__ int qs_excl_scan(int v, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int t = s_w[i];
        if (i < w) base += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return base + incl - v;
}

__device__ __forceinline__ void qs_bitonic(unsigned long long* a, int npad) {
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (npad >> 1); t += NT) {
                const int lo = ((t / stride) * stride << 1) + (t % stride);
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                unsigned long long x = a[lo], y = a[hi];
                if ((x > y) == asc) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t qs_block_max(uint32_t v, uint32_t* s_red) {
    v = wave_max_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) r = max(r, s_red[i]);
    return r;
}
__device__ __forceinline__ uint32_t qs_block_min(uint32_t v, uint32_t* s_red) {
    v = wave_min_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NW; i++) r = min(r, s_red[i]);
    return r;
}

// Keep the K smallest (key, pos) items of s_items[0..n) in s_items[0..K) (any order) and
// return the largest kept key.  Block-wide; K < n <= QS_CAPS, K <= 1024.
__device__ uint32_t qs_compact(unsigned long long* s_items, int n, int K, int* s_hist, int* s_w,
                               uint32_t* s_red, int* s_misc) {
    const int tid = threadIdx.x;
    unsigned long long it[IPT];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int i = tid + NT * j;
        it[j] = i < n ? s_items[i] : ~0ull;
        if (i < n) {
            const uint32_t key = (uint32_t)(it[j] >> 32);
            mn = key < mn ? key : mn;
            mx = key > mx ? key : mx;
        }
    }
    for (int i = tid; i < QS_NB; i += NT) s_hist[i] = 0;
    if (tid < 8) s_misc[tid] = 0;
    mn = qs_block_min(mn, s_red);
    mx = qs_block_max(mx, s_red);   // (barriers inside also publish the zeroed histogram)
    const uint32_t range = mx - mn;
    const int s = range >= (uint32_t)QS_NB ? (32 - __clz((int)range)) - 11 : 0;
#pragma unroll
    for (int j = 0; j < IPT; j++)
        if (tid + NT * j < n) atomicAdd(&s_hist[((uint32_t)(it[j] >> 32) - mn) >> s], 1);
    __syncthreads();
    int cb = 0;
#pragma unroll
    for (int j = 0; j < BPT; j++) cb += s_hist[tid * BPT + j];
    int tot;
    const int ex = qs_excl_scan(cb, s_w, tot);
    if (ex < K && K <= ex + cb) {
        int run = ex;
#pragma unroll
        for (int j = 0; j < BPT; j++) {
            const int c = s_hist[tid * BPT + j];
            if (run < K && K <= run + c) {
                s_misc[0] = tid * BPT + j;
                s_misc[1] = run;
                s_misc[2] = c;
            }
            run += c;
        }
    }
    __syncthreads();
    const uint32_t B = (uint32_t)s_misc[0];
    const int below = s_misc[1], cnt = s_misc[2];
    const int need = K - below;
    if (cnt == need) {
        // every member of bucket B is kept
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            if (tid + NT * j < n && (((uint32_t)(it[j] >> 32) - mn) >> s) <= B)
                s_items[atomicAdd(&s_misc[3], 1)] = it[j];
        }
        __syncthreads();
    } else if (cnt <= 1024) {
        int cpad = 2;
        while (cpad < cnt) cpad <<= 1;
        unsigned long long* cand = s_items + (QS_CAPS - cpad);  // disjoint from [0, below)
#pragma unroll
        for (int j = 0; j < IPT; j++) {
            if (tid + NT * j < n) {
                const uint32_t b = ((uint32_t)(it[j] >> 32) - mn) >> s;
                if (b < B) s_items[atomicAdd(&s_misc[3], 1)] = it[j];
                else if (b == B) cand[atomicAdd(&s_misc[4], 1)] = it[j];
            }
        }
        for (int i = cnt + tid; i < cpad; i += NT) cand[i] = ~0ull;
        qs_bitonic(cand, cpad);
        for (int i = tid; i < need; i += NT) s_items[below + i] = cand[i];
        __syncthreads();
    } else {
        // degenerate: > 1024 items share the threshold bucket -> sort everything
#pragma unroll
        for (int j = 0; j < IPT; j++) s_items[tid + NT * j] = it[j];
        qs_bitonic(s_items, QS_CAPS);
    }
    uint32_t kmx = 0u;
    for (int i = tid; i < K; i += NT) {
        const uint32_t key = (uint32_t)(s_items[i] >> 32);
        kmx = key > kmx ? key : kmx;
    }
    return qs_block_max(kmx, s_red);
}

template <int M>
struct CodeReg {
    uint32_t w[M / 4];
};

template <int M>
__device__ __forceinline__ CodeReg<M> load_code(const uint8_t* p) {
    CodeReg<M> c;
    if (M == 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
    } else if (M == 32) {
        const uint4 v0 = reinterpret_cast<const uint4*>(p)[0], v1 = reinterpret_cast<const uint4*>(p)[1];
        c.w[0] = v0.x; c.w[1] = v0.y; c.w[2] = v0.z; c.w[3] = v0.w;
        c.w[4] = v1.x; c.w[5] = v1.y; c.w[6] = v1.z; c.w[7] = v1.w;
    } else {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        c.w[0] = v.x; c.w[1] = v.y;
    }
    return c;
}
}  // namespace

template <bool L2, int M, int DSUB>
__global__ __launch_bounds__(NT, 2) void k_ivfpq_qscan(
        const float* __restrict__ x, int nq, int P, int PG, const int* __restrict__ probe_list,
        const float* __restrict__ dis0_arr, const float* __restrict__ cc,
        const float* __restrict__ pqc, const int64_t* __restrict__ list_off,
        const int* __restrict__ list_len, const uint8_t* __restrict__ list_mask, int nlist,
        const uint8_t* __restrict__ codes, const int64_t* __restrict__ ids, FilterDesc filt,
        int need_ids, int R, int Rpad, float* __restrict__ out_dis, int64_t* __restrict__ out_ids,
        int* __restrict__ q_total, int dbg, unsigned long long* __restrict__ dbg_out) {
    constexpr int D = M * DSUB;
    // optional phase timing (dbg & 32): s_memtime deltas accumulated by every thread (uniform)
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = 0;
    auto TICK = [&](int slot) {
        if (dbg & 32) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tacc[slot] += now - tlast;
            tlast = now;
        }
    };
    if (dbg & 32) tlast = __builtin_amdgcn_s_memtime();
    constexpr int TPM = NT / M;       // threads per sub-quantizer
    constexpr int CPT = 256 / TPM;    // consecutive centroids per thread
    extern __shared__ unsigned long long s_dyn[];
    unsigned long long* s_items = s_dyn;                                  // QS_CAPS
    float* s_lut = reinterpret_cast<float*>(s_dyn + QS_CAPS);             // 2 x M*256
    int* s_hist = reinterpret_cast<int*>(s_lut + 2 * M * 256);            // QS_NB
    int* s_poff = s_hist + QS_NB;                                         // P+1
    int* s_pl = s_poff + (P + 1);                                         // P   list number
    float* s_d0 = reinterpret_cast<float*>(s_pl + P);                     // P   dis0
    // P   list start (entries); 8-byte aligned: QS_CAPS*8 + 2*M*1024 + QS_NB*4 + (3P+1)*4 -> pad
    long long* s_off = reinterpret_cast<long long*>(
            (reinterpret_cast<uintptr_t>(s_d0 + P) + 7) & ~(uintptr_t)7);
    float* s_cl = reinterpret_cast<float*>(                                // P*D coarse centroids
            (reinterpret_cast<uintptr_t>(s_off + P) + 15) & ~(uintptr_t)15);
    __shared__ int s_w[NW];
    __shared__ uint32_t s_red[NW];
    __shared__ int s_misc[8];
    __shared__ int s_cnt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = blockIdx.y;

    // ---- per-thread PQ state, query independent: loaded ONCE per (persistent) workgroup ----
    const int mm = tid / TPM, c_base = (tid % TPM) * CPT;
    float cen[CPT][DSUB];
    {
        // CPT*DSUB contiguous floats per thread: float4 loads
        const float4* c4 = reinterpret_cast<const float4*>(pqc + ((int64_t)mm * 256 + c_base) * DSUB);
        float4 t[CPT * DSUB / 4];
#pragma unroll
        for (int i = 0; i < CPT * DSUB / 4; i++) t[i] = c4[i];
#pragma unroll
        for (int i = 0; i < CPT * DSUB / 4; i++) {
            float* dst = &cen[0][0] + 4 * i;
            dst[0] = t[i].x; dst[1] = t[i].y; dst[2] = t[i].z; dst[3] = t[i].w;
        }
    }
    float rn[CPT];
    if (L2) {
#pragma unroll
        for (int i = 0; i < CPT; i++) rn[i] = fvec_norm_L2sqr(cen[i], DSUB);
    }
    const int my_lut = mm * 256 + c_base;   // CPT consecutive LUT entries

  // persistent loop over queries: grid.x workgroups stride through the batch
  TICK(0);   // codebook load
  for (int q = blockIdx.x; q < nq; q += gridDim.x) {
    const float* xq = x + (int64_t)q * D;
    const int* plist = probe_list + (int64_t)q * P;
    float xm[DSUB];
#pragma unroll
    for (int k = 0; k < DSUB; k++) xm[k] = xq[mm * DSUB + k];
    float st2[CPT];
#pragma unroll
    for (int i = 0; i < CPT; i++) st2[i] = fvec_ny_row<false>(xm, cen[i], DSUB);
    if (!L2) {
#pragma unroll
        for (int i = 0; i < CPT; i++) s_lut[my_lut + i] = st2[i];
    }
    // ---- probe metadata -> LDS (list number, dis0) and scan positions = exclusive prefix of
    // the probed list lengths, so the probe loop below never chases pointers through HBM ----
    {
        int running = 0;
        for (int p0 = 0; p0 < P; p0 += NT) {
            const int p = p0 + tid;
            int len = 0, l = -1;
            if (p < P) {
                l = plist[p];
                if (l >= 0 && l < nlist && (!list_mask || list_mask[l])) len = list_len[l];
                s_pl[p] = l;
                s_d0[p] = dis0_arr[(int64_t)q * P + p];
                s_off[p] = len > 0 ? (long long)list_off[l] : 0;
            }
            int tot;
            const int ex = qs_excl_scan(len, s_w, tot);
            if (p < P) s_poff[p] = running + ex;
            running += tot;
        }
        if (tid == 0) {
            s_poff[P] = running;
            s_cnt = 0;
            if (q_total && g == 0) q_total[q] = running;
        }
    }
    __syncthreads();

    // ---- coarse-centroid slices of every probed list -> LDS, one coalesced burst ----
    if (L2) {
        const float4* cc4 = reinterpret_cast<const float4*>(cc);
        float4* cl4 = reinterpret_cast<float4*>(s_cl);
        for (int e = tid; e < P * (D / 4); e += NT) {
            const int pp = e / (D / 4), c4 = e - pp * (D / 4);
            const int l = s_pl[pp];
            const bool live = s_poff[pp + 1] != s_poff[pp];
            cl4[e] = live ? cc4[(int64_t)l * (D / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
    }

    TICK(1);   // per-query prologue: st2, metadata, centroid staging
    uint32_t tau = 0xffffffffu;
    int buf = 0;
    const int per = (P + PG - 1) / PG;
    const int p_begin = g * per, p_end = min(P, p_begin + per);
    auto next_nonempty = [&](int p) {
        while (p < p_end && s_poff[p + 1] == s_poff[p]) p++;
        return min(p, p_end);
    };
    // first-pass codes (and ids) of a list, issued up to three lists ahead of their use
    auto first_loads = [&](int p, CodeReg<M>& c, int64_t& id) {
        if (p < p_end) {
            const int len = s_poff[p + 1] - s_poff[p];
            const int64_t e = s_off[p] + min(tid, len - 1);
            c = load_code<M>(codes + e * M);
            id = need_ids ? ids[e] : 0;
        }
    };
    auto do_probe = [&](int p, CodeReg<M>& slot_code, int64_t& slot_id, int p_refill) {
        const int len = s_poff[p + 1] - s_poff[p];
        const float dis0 = s_d0[p];
        const int pos0 = s_poff[p];
        const int64_t off = s_off[p];
        const uint8_t* lc = codes + off * M;
        const int64_t* lid = ids + off;
        const float* lut = s_lut;
        if (L2) {
            float clm[DSUB];                          // slice mm of the coarse centroid (LDS)
            const float* cl = s_cl + p * D + mm * DSUB;
#pragma unroll
            for (int k = 0; k < DSUB; k++) clm[k] = cl[k];
            float v[CPT];
            if (dbg & 2) {
#pragma unroll
                for (int i = 0; i < CPT; i++) v[i] = clm[i % DSUB];
            } else
#pragma unroll
            for (int i = 0; i < CPT; i++) {
                const float ipc = fvec_ny_row<false>(clm, cen[i], DSUB);
                const float t2 = __builtin_fmaf(2.0f, ipc, rn[i]);   // precomputed-table entry
                v[i] = __builtin_fmaf(-2.0f, st2[i], t2);            // fvec_madd(T2, -2, st2)
            }
            // double buffer: the other half may still be read by waves finishing the last list
            float* wl = s_lut + buf * (M * 256) + my_lut;
#pragma unroll
            for (int i = 0; i < CPT; i += 4)
                *reinterpret_cast<float4*>(wl + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
            lut = s_lut + buf * (M * 256);
            buf ^= 1;
        }
        TICK(2);   // LUT arithmetic + LDS write
        CodeReg<M> nxt_code;
        int64_t nxt_id = 0;
        for (int j0 = 0; j0 < len; j0 += NT) {
            const int clen = min(NT, len - j0);
            if (!(dbg & 8)) __syncthreads();  // LUT visible (first pass) / appends of the previous pass settled
            const int cnt = s_cnt;
            if (!(dbg & 8)) __syncthreads();  // everyone has read s_cnt before the next appends: cnt is uniform
            TICK(3);   // the two barriers
            if (cnt + clen > QS_CAPS) {
                tau = qs_compact(s_items, cnt, R, s_hist, s_w, s_red, s_misc);
                if (tid == 0) s_cnt = R;
                __syncthreads();
                TICK(4);   // compaction
            }
            const int j = j0 + tid;
            const bool live = j < len;
            const CodeReg<M> code = j0 == 0 ? slot_code : nxt_code;
            const int64_t id = j0 == 0 ? slot_id : nxt_id;
            if (j0 + NT < len) {   // long list: next pass of the same list
                const int jn = min(j + NT, len - 1);
                nxt_code = load_code<M>(lc + (int64_t)jn * M);
                nxt_id = need_ids ? lid[jn] : 0;
            }
            bool ok = live;
            if (need_ids) ok = live && id >= 0 && is_valid_doc(filt, id);
            float dis = dis0;
            if (dbg & 1) {
                dis += __uint_as_float((code.w[0] & 0xffff) | 0x3f800000);
            } else
            {
                // issue all M gathers before the first add: hipcc otherwise schedules them
                // one or two at a time and the chain pays M LDS latencies
                float t[M];
#pragma unroll
                for (int m = 0; m < M; m++)
                    t[m] = lut[m * 256 + ((code.w[m >> 2] >> ((m & 3) * 8)) & 255)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < M; m++) dis += t[m];   // sequential, reference order
            }
            if (j0 == 0) first_loads(p_refill, slot_code, slot_id);   // slot is free again
            if (dbg & 32) asm volatile("" :: "v"(dis));
            TICK(5);   // code wait + gathers
            const uint32_t key = qs_key<L2>(dis);
            const bool pass = ok && key <= tau && !(dbg & 4);
            const unsigned long long bal = __ballot(pass);
            if (bal) {
                const int nw = __popcll(bal);
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_cnt, nw);
                base = __shfl(base, 0, 64);
                if (pass) {
                    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
                    s_items[base + rank] = ((unsigned long long)key << 32) | (unsigned)(pos0 + j);
                }
            }
            TICK(6);   // append
        }
    };
    {
        int pa = next_nonempty(p_begin), pb = next_nonempty(pa + 1), pc = next_nonempty(pb + 1);
        CodeReg<M> cA, cB, cC;
        int64_t iA = 0, iB = 0, iC = 0;
        first_loads(pa, cA, iA);
        first_loads(pb, cB, iB);
        first_loads(pc, cC, iC);
        while (pa < p_end) {
            const int pd = next_nonempty(pc + 1);
            do_probe(pa, cA, iA, pd);
            if (pb >= p_end) break;
            const int pe = next_nonempty(pd + 1);
            do_probe(pb, cB, iB, pe);
            if (pc >= p_end) break;
            const int pf = next_nonempty(pe + 1);
            do_probe(pc, cC, iC, pf);
            pa = pd;
            pb = pe;
            pc = pf;
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    int nres = cnt;
    if (cnt > R) {
        (void)qs_compact(s_items, cnt, R, s_hist, s_w, s_red, s_misc);
        nres = R;
    }
    __syncthreads();
    for (int i = nres + tid; i < Rpad; i += NT) s_items[i] = ~0ull;
    qs_bitonic(s_items, Rpad);
    // ---- write recall_num results: distance + vector id (pos -> probe -> list entry) ----
    const float sentinel = L2 ? INFINITY : -INFINITY;
    float* od = out_dis + ((int64_t)q * PG + g) * R;
    int64_t* oi = out_ids + ((int64_t)q * PG + g) * R;
    for (int r = tid; r < R; r += NT) {
        const unsigned long long it = s_items[r];
        float val = sentinel;
        int64_t id = -1;
        if (it != ~0ull) {
            const int ps = (int)(uint32_t)it;
            int lo = 0, hi = P - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_poff[mid] <= ps) lo = mid; else hi = mid - 1;
            }
            id = ids[s_off[lo] + (ps - s_poff[lo])] & 0x7fffffffffffffffLL;
            val = qs_unkey<L2>((uint32_t)(it >> 32));
        }
        od[r] = val;
        oi[r] = id;
    }
    __syncthreads();   // LDS state is reused by the next query of this workgroup
    TICK(7);   // epilogue: final compaction, sort, id lookup
  }
  if ((dbg & 32) && dbg_out && tid == 0 && blockIdx.y == 0) {
      for (int i = 0; i < 8; i++) dbg_out[(size_t)blockIdx.x * 8 + i] = tacc[i];
  }
}

// dis0 of the inner-product scan: <x_q, centroid_l> in fvec_inner_product order
// (precompute_list_tables_IP, gamma_index_ivfpq.h:216-230).  One thread per (query, probe).
__global__ __launch_bounds__(256) void k_ip_dis0(const float* __restrict__ x, int d, int64_t npairs, int P,
                                                 const int* __restrict__ probe_list,
                                                 const float* __restrict__ cc, int nlist,
                                                 float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const int l = probe_list[i];
    float v = 0.f;
    if (l >= 0 && l < nlist) v = fvec_dist<false>(x + (i / P) * d, cc + (int64_t)l * d, d);
    out[i] = v;
}

void launch_ip_dis0(hipStream_t s, const float* x, int nq, int d, int P, const int* probe_list,
                    const float* cc, int nlist, float* out) {
    const int64_t n = (int64_t)nq * P;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ip_dis0, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, d, n, P,
                       probe_list, cc, nlist, out);
}

bool qscan_supported(int d, int M, int R, int P) {
    const int dsub = d / M;
    if (d > 128 || R > 1024 || P > 64) return false;
    return (M == 16 && (dsub == 8 || dsub == 4)) || (M == 32 && dsub == 4) ||
           (M == 8 && (dsub == 8 || dsub == 4 || dsub == 16));
}

template <bool L2, int M, int DSUB>
static void launch_qscan_t(hipStream_t s, const float* x, int nq, int P, int PG, const int* probe_list,
                           const float* dis0, const float* cc, const float* pqc,
                           const int64_t* list_off, const int* list_len, const uint8_t* list_mask,
                           int nlist, const uint8_t* codes, const int64_t* ids, const FilterDesc& filt,
                           int need_ids, int R, float* out_dis, int64_t* out_ids, int* q_total) {
    const int Rpad = select_kpad(R);
    static const int dbg = getenv("GAMMA_HIP_QSCAN_DBG") ? atoi(getenv("GAMMA_HIP_QSCAN_DBG")) : 0;
    static unsigned long long* dbg_buf = nullptr;
    if ((dbg & 32) && !dbg_buf) (void)hipMalloc((void**)&dbg_buf, 8 * 8 * 4096);
    const size_t lds = (size_t)QS_CAPS * 8 + (size_t)2 * M * 256 * 4 + (size_t)QS_NB * 4 +
                       (size_t)(3 * P + 1) * 4 + 8 + (size_t)P * 8 + 16 + (size_t)P * M * DSUB * 4;
    auto kern = k_ivfpq_qscan<L2, M, DSUB>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // persistent: one workgroup per CU (register file) walks the queries
    const int gx = std::min(nq, std::max(1, 256 / PG));
    hipLaunchKernelGGL(kern, dim3(gx, PG), dim3(NT), lds, s, x, nq, P, PG, probe_list, dis0, cc, pqc,
                       list_off, list_len, list_mask, nlist, codes, ids, filt, need_ids, R, Rpad,
                       out_dis, out_ids, q_total, dbg, dbg_buf);
    if (dbg & 32) {
        static int shown = 0;
        if (shown++ == 8) {   // one steady-state launch
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> hb((size_t)gx * 8);
            (void)hipMemcpy(hb.data(), dbg_buf, hb.size() * 8, hipMemcpyDeviceToHost);
            const char* nm[8] = {"codebook", "q-prologue", "lut", "barriers", "compact", "gather", "append", "epilogue"};
            for (int i = 0; i < 8; i++) {
                double sum = 0, mx = 0;
                for (int w = 0; w < gx; w++) { double v = (double)hb[(size_t)w * 8 + i]; sum += v; mx = std::max(mx, v); }
                fprintf(stderr, "qscan phase %-10s avg %.0f max %.0f ticks per workgroup\n", nm[i], sum / gx, mx);
            }
        }
    }
}

void launch_ivfpq_qscan(hipStream_t s, bool l2, const float* x, int nq, int d, int M, int P, int PG,
                        const int* probe_list, const float* dis0, const float* cc, const float* pqc,
                        const int64_t* list_off, const int* list_len, const uint8_t* list_mask,
                        int nlist, const uint8_t* codes, const int64_t* ids, const FilterDesc& filt,
                        int need_ids, int R, float* out_dis, int64_t* out_ids, int* q_total) {
    if (nq <= 0) return;
    const int dsub = d / M;
#define GH_QS(MM, DS)                                                                               \
    do {                                                                                            \
        if (l2) launch_qscan_t<true, MM, DS>(s, x, nq, P, PG, probe_list, dis0, cc, pqc, list_off,  \
                                             list_len, list_mask, nlist, codes, ids, filt,          \
                                             need_ids, R, out_dis, out_ids, q_total);               \
        else launch_qscan_t<false, MM, DS>(s, x, nq, P, PG, probe_list, dis0, cc, pqc, list_off,    \
                                           list_len, list_mask, nlist, codes, ids, filt, need_ids,  \
                                           R, out_dis, out_ids, q_total);                           \
    } while (0)
    if (M == 16 && dsub == 8) GH_QS(16, 8);
    else if (M == 16 && dsub == 4) GH_QS(16, 4);
    else if (M == 32 && dsub == 4) GH_QS(32, 4);
    else if (M == 8 && dsub == 8) GH_QS(8, 8);
    else if (M == 8 && dsub == 4) GH_QS(8, 4);
    else if (M == 8 && dsub == 16) GH_QS(8, 16);
#undef GH_QS
}

}  // namespace gh
