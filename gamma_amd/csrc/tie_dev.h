// tie_dev.h -- one query replayed through the reference's heaps (device code shared by k_tie_replay, ties.hip, and the
// small-batch chain's tail kernel, select.hip).  See ties.hip for what is replayed and why.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_utils.h"
#include "device_math.h"
#include "heap_dev.h"
#include "kernels.h"
#include "rerank_dev.h"

namespace gh {

constexpr int TR_MAXK = 1024;    // heap sizes the in-kernel replays of the small-batch chains cover (recall_num and k)
constexpr int TR_MAXK_BIG = 4096;   // k_tie_replay: every recall_num / k the ABI accepts (its <.., 4096, 4096> variant beyond 1024)
constexpr int TR_MAXP = 1024;    // probes per query
constexpr int TR_STAGE = 1024;   // survivor items sorted per round (= the scan's slice capacity)
constexpr int TR_SLAB = 2048;    // candidates of the slab part brought into LDS per round (with it a C3 replay needs 21 KB of LDS: a
                                 // workgroup fits a CU beside two of the coarse kernel, whose stage a deferred replay runs behind)


// dynamic LDS of one replayed query
struct TieLds {
    uint2* hR;                  // R-heap: (value, position in the query's slab), array order
    uint2* hK;                  // k-heap: (exact distance, slot of the R-heap array)
    int64_t* id;                // vector id of R-heap slot j
    float* ex;                  // exact distance of slot j
    float* slab;                // staged candidates
    unsigned long long* it;     // sort buffer
    int* off;                   // [P + 1]
    int64_t* base;              // [P]
};
__host__ __device__ inline size_t tie_align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t tie_replay_lds_bytes_(int R, int k, int P, int slab = TR_SLAB, int stage = TR_STAGE) {
    return tie_align16((size_t)(R + 2) * 8) + tie_align16((size_t)(k + 2) * 8) + tie_align16((size_t)R * 8) +
           tie_align16((size_t)R * 4) + (size_t)slab * 4 + (size_t)stage * 8 + tie_align16((size_t)(P + 1) * 4) +
           tie_align16((size_t)P * 8);
}
__device__ __forceinline__ TieLds tie_carve(char* p, int R, int k, int P, int slab, int stage = TR_STAGE) {
    TieLds L;
    L.hR = reinterpret_cast<uint2*>(p);
    p += tie_align16((size_t)(R + 2) * 8);
    L.hK = reinterpret_cast<uint2*>(p);
    p += tie_align16((size_t)(k + 2) * 8);
    L.id = reinterpret_cast<int64_t*>(p);
    p += tie_align16((size_t)R * 8);
    L.ex = reinterpret_cast<float*>(p);
    p += tie_align16((size_t)R * 4);
    L.slab = reinterpret_cast<float*>(p);
    p += (size_t)slab * 4;
    L.it = reinterpret_cast<unsigned long long*>(p);
    p += (size_t)stage * 8;
    L.off = reinterpret_cast<int*>(p);
    p += tie_align16((size_t)(P + 1) * 4);
    L.base = reinterpret_cast<int64_t*>(p);
    return L;
}

// One query, replayed the way the reference runs it.  Called by all NT threads of a workgroup (NT a multiple
// of 64, <= 1024); `lds` = tie_replay_lds_bytes_(R, k, P, SLAB) bytes, 16-byte aligned, free for this call.
template <bool L2, int NT, int SLAB = TR_SLAB, int STG = TR_STAGE, int MAXK = TR_MAXK>
__device__ __forceinline__ void tie_replay_query(const TieReplayArgs& a, int q, char* lds, unsigned long long* dbg,
                                                 int slab_row = -1) {
#define GH_TT(i) do { if (dbg && threadIdx.x == 0) dbg[i] = wall_clock64(); } while (0)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int R = a.R, k = a.k, P = a.P;
    const TieLds L = tie_carve(lds, R, k, P, SLAB, STG);
    __syncthreads();   // the LDS is free
    GH_TT(0);
    if (a.pair_off) {   // (compact rows: the tables of the slab_row-th flagged query)
        const int64_t qr = slab_row >= 0 ? slab_row : q;
        for (int i = tid; i <= P; i += NT) L.off[i] = a.pair_off[qr * (P + 1) + i];
        for (int i = tid; i < P; i += NT) L.base[i] = a.pair_base[qr * P + i];
    }
    heap_fill(L.hR, R, tid, NT);   // heap_heapify: (neutral, -1)
    if (a.has_rank) heap_fill(L.hK, k, tid, NT);
    __syncthreads();
    // ---- the candidate stream, in scan order ----
    // bounded query (the scan published a bound and no slice overflowed): first probe group from the
    // slab, the other groups from their survivor slices; otherwise the whole slab
    bool sliced = a.always_sliced != 0;
    if (a.ready) {
        sliced = (a.ready[q] >> 32) == 1ull;
        for (int s = 0; s < a.nsl && sliced; s++)
            if (a.gcnt[(int64_t)q * a.nsl + s] > a.slice_cap) sliced = false;   // uniform
    }
    const int ntot = a.pair_off ? L.off[P] : a.fixed_n;
    const int n_slab = sliced ? (a.pair_off ? L.off[min(a.G, P)] : min(a.G, ntot)) : ntot;
    const float* slab = a.slab + (int64_t)(slab_row >= 0 ? slab_row : q) * a.q_stride;
    GH_TT(1);
    HeapWalk w;
    w.begin(L.hR, R);
    // IVFFLAT / flat scanners: `if (C::cmp(simi[0], dis)) { heap_pop; heap_push }` (gamma_index_ivfflat.h:52-75,
    // gamma_index_flat.cc:118-300) -- one candidate at a time, 64 compared with the top per step.  The IVFPQ scanner:
    // heap_replace_top, pipelined (HeapWalk).  take(): one block of <= 64 candidates in stream order, wave 0.
    float ptop = kHeapFltMax;
    // (heap_pop + heap_push cannot be pipelined the way heap_replace_top is: the next pop starts from slot k, where this
    //  push ended.  What can be cut is the cost of a level: up to 63 entries the heap lives in ONE register -- node i in
    //  lane i, a level is two readlanes instead of an LDS round trip (~135 ns per accepted candidate at k = 10); the array
    //  goes to LDS once, when the stream is through.  With two registers (k <= 127) the lane / register selects cost what the
    //  LDS round trip does: measured at k = 100, 0.80 ms for a 16384-row slab against 0.74 -- not used.)
    // From 16 entries on (flat search at k = 100): the sifts run with all 64 lanes (ParHeap, heap_dev.h) -- three LDS round
    // trips and no loop over levels per accepted candidate: 450 ns per pop + push up to 128 entries, 590 up to 256, against
    // 450 / 590 / 680 ns in one register at 10 / 32 / 63 entries and 1020 ns sequentially at 100 (tools/exp/heap_bench.hip).
    const bool reg_heap = a.pop_push && R <= 15;
    const bool par_heap = a.pop_push && !reg_heap && R <= kParHeapMaxK;
    RegHeap<1> rh;
    rh.fill();
    auto take = [&](bool valid, float dv, int pay) {
        if (!a.pop_push) {
            w.accept(valid, dv, pay);
            return;
        }
        if (!valid) dv = INFINITY;
        unsigned long long m = __ballot(ptop > dv);
        while (m) {
            const int l = (int)__ffsll((long long)m) - 1;
            const float val = hw_readlane_f(dv, l);
            if (reg_heap) {
                rh.pop(R);
                rh.push(R, val, (unsigned)hw_readlane_i(pay, l));
                ptop = rh.top();
            } else if (par_heap) {
                const float root = par_heap_pop(L.hR, R);
                ptop = par_heap_push(L.hR, R, val, (unsigned)hw_readlane_i(pay, l)) ? val : root;
            } else {
                heap_pop_seq(L.hR, R);
                heap_push_seq(L.hR, R, val, (unsigned)hw_readlane_i(pay, l));
                ptop = hs_f(L.hR[1].x);
            }
            const unsigned long long above = l >= 63 ? 0ull : (~0ull << (l + 1));
            m = __ballot(ptop > dv) & above;
        }
    };
    {
        // The slab part of the stream, SLAB candidates at a time: all threads bring a chunk into LDS (the next one is in
        // flight in registers meanwhile), every wave compares its share of the chunk's 64-candidate groups with the root
        // value the heap had when the chunk began -- the root only ever decreases, so a group without a candidate below
        // it holds nothing the heap would take -- and wave 0 walks the groups that are left, in order.
        static_assert(SLAB % NT == 0 && SLAB / 64 <= 64, "chunk = whole rounds of the workgroup, at most 64 groups");
        constexpr int PER = SLAB / NT;
        unsigned long long* gmask = L.it;                        // [SLAB / 64], the sort buffer is idle here
        float* s_top = reinterpret_cast<float*>(L.it + SLAB / 64);
        const float far = L2 ? INFINITY : -INFINITY;
        float pre[PER];
        auto issue = [&](int c0) {
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int j = c0 + u * NT + tid;
                pre[u] = j < n_slab ? slab[j] : far;
            }
        };
        if (n_slab > 0) issue(0);
        if (tid == 0) *s_top = kHeapFltMax;
        for (int c0 = 0; c0 < n_slab; c0 += SLAB) {
            const int cn = min(SLAB, n_slab - c0);
            __syncthreads();   // wave 0 is through the previous chunk
#pragma unroll
            for (int u = 0; u < PER; u++) L.slab[u * NT + tid] = L2 ? pre[u] : -pre[u];   // filtered entries: +inf
            if (c0 + SLAB < n_slab) issue(c0 + SLAB);
            __syncthreads();
            const float t0 = *s_top;
            for (int g = wv; g * 64 < cn; g += NT / 64) {
                const unsigned long long m = __ballot(t0 > L.slab[g * 64 + lane]);   // beyond cn: +inf
                if (lane == 0) gmask[g] = m;
            }
            __syncthreads();
            if (wv == 0) {
                const int ng = (cn + 63) >> 6;
                unsigned long long gm = __ballot(lane < ng && gmask[min(lane, ng - 1)] != 0ull);
                while (gm) {
                    const int g = (int)__ffsll((long long)gm) - 1;
                    gm &= gm - 1ull;
                    const int j = g * 64 + lane;
                    take(j < cn, L.slab[j], c0 + j);
                }
                if (lane == 0) *s_top = a.pop_push ? ptop : w.top;
            }
        }
        __syncthreads();
    }
    GH_TT(2);
    if (dbg && threadIdx.x == 0) dbg[7] = (unsigned long long)(unsigned)w.nin | ((unsigned long long)(unsigned)n_slab << 32);
    if (sliced) {
        // slices 1.. in order (slice s holds positions of probe group s only, so slices are ordered among
        // themselves); several short slices share one sorting round
        int s = a.slice0_all ? 0 : 1;
        while (s < a.nsl) {
            __syncthreads();   // the sort buffer is free again
            int n = 0, s_end = s;
            while (s_end < a.nsl) {
                const int c = min(a.gcnt[(int64_t)q * a.nsl + s_end], STG);
                if (n + c > STG) break;
                const unsigned long long* src = a.surv + ((int64_t)q * a.nsl + s_end) * a.slice_cap;
                for (int i = tid; i < c; i += NT) {
                    const unsigned long long it = src[i];   // (key << 32 | position)
                    L.it[n + i] = (it << 32) | (it >> 32);  // -> (position << 32 | key)
                }
                n += c;
                s_end++;
            }
            block_rank_sort<NT, STG / NT>(L.it, n);   // positions are distinct
            if (wv == 0) {
                for (int j0 = 0; j0 < n; j0 += 64) {
                    const unsigned long long it = L.it[min(j0 + lane, n - 1)];
                    const uint32_t key = (uint32_t)it;
                    const float val = key2f(L2 ? key : ~key);
                    // (slice0_all: the first group's survivors are in the slice too -- they went through with the slab part)
                    take(j0 + lane < n && (int)(uint32_t)(it >> 32) >= n_slab, L2 ? val : -val, (int)(uint32_t)(it >> 32));
                }
            }
            s = s_end;
        }
    }
    if (wv == 0 && !a.pop_push) w.drain();
    if (wv == 0 && reg_heap) rh.dump(L.hR, R);
    __syncthreads();
    GH_TT(3);
    // ---- the R-heap is final: array order in hR[1..R].  Positions -> vector ids. ----
    auto pos_to_id = [&](int ps) -> int64_t {
        if (ps < 0) return -1;
        if (!a.pair_off) return ps;
        int lo = 0, hi = P - 1;
        while (lo < hi) {   // last p with off[p] <= ps
            const int mid = (lo + hi + 1) >> 1;
            if (L.off[mid] <= ps) lo = mid; else hi = mid - 1;
        }
        return a.ids[L.base[lo] + (ps - L.off[lo])] & 0x7fffffffffffffffLL;
    };
    for (int j = tid; j < R; j += NT) L.id[j] = pos_to_id((int)L.hR[1 + j].y);
    __syncthreads();
    float* od = a.distances + (int64_t)q * k;
    int64_t* ol = a.labels + (int64_t)q * k;
    if (a.has_rank) {
        // exact distances in array order: 8 threads per candidate = the 8 AVX lane accumulators
        // (same arithmetic as k_rerank_topk)
        const int l8 = tid & 7, g = tid >> 3;
        const float* xq = a.x + (int64_t)q * a.d;
        for (int j0 = 0; j0 < R; j0 += NT / 8) {
            const int j = j0 + g;
            const int64_t id = j < R ? L.id[j] : -1;
            bool live = id >= 0 && (a.ex_slab != nullptr || id < a.nraw);
            float dis;
            if (a.ex_slab) {   // (uniform) raw vectors sharded with their lists: the distance travelled with the entry
                const int ps = j < R ? (int)L.hR[1 + j].y : -1;
                dis = (live && ps >= 0) ? a.ex_slab[(int64_t)(slab_row >= 0 ? slab_row : q) * a.q_stride + ps] : 0.f;
                if (live && dis != dis) {   // a member whose owner did not export its distance: never silent
                    if (l8 == 0 && a.ex_missing) atomicAdd(a.ex_missing, 1);
                    live = false;
                }
            } else {
                dis = rerank_dist8<L2>(xq, a.raw + (live ? id : 0) * a.d, a.d, l8, live);
            }
            if (l8 == 0 && j < R) {
                const bool ok = live && dis <= a.max_score && dis >= a.min_score;   // IsSimilarScoreValid
                L.ex[j] = ok ? (L2 ? dis : -dis) : INFINITY;   // +inf never beats the heap's top
            }
        }
        __syncthreads();
        GH_TT(4);
        if (wv == 0 && k <= 15) {
            // the k-heap in registers: heap_pop + heap_push per accepted candidate (gamma_index_ivfpq.cc:664-676)
            RegHeap<1> kh;
            kh.fill();
            float top = kHeapFltMax;
            for (int j0 = 0; j0 < R; j0 += 64) {
                const int j = j0 + lane;
                const float dv = j < R ? L.ex[j] : INFINITY;
                unsigned long long m = __ballot(top > dv);
                while (m) {
                    const int l = (int)__ffsll((long long)m) - 1;
                    kh.pop(k);
                    kh.push(k, hw_readlane_f(dv, l), (unsigned)(j0 + l));
                    top = kh.top();
                    const unsigned long long above = l >= 63 ? 0ull : (~0ull << (l + 1));
                    m = __ballot(top > dv) & above;
                }
            }
            const int real = kh.reorder_pops(k);
            kh.dump(L.hK, k);
            heap_reorder_tail(L.hK, k, real);
        } else if (wv == 0) {
            float top = kHeapFltMax;
            for (int j0 = 0; j0 < R; j0 += 64) {
                const int j = j0 + lane;
                const float dv = j < R ? L.ex[j] : INFINITY;
                unsigned long long m = __ballot(top > dv);
                while (m) {
                    const int l = (int)__ffsll((long long)m) - 1;
                    const float val = hw_readlane_f(dv, l);
                    if (k <= kParHeapMaxK) {   // heap_pop + heap_push (gamma_index_ivfpq.cc:664-676), all lanes per sift
                        const float root = par_heap_pop(L.hK, k);
                        top = par_heap_push(L.hK, k, val, (unsigned)(j0 + l)) ? val : root;
                    } else {
                        heap_pop_seq(L.hK, k);
                        heap_push_seq(L.hK, k, val, (unsigned)(j0 + l));
                        top = hs_f(L.hK[1].x);
                    }
                    const unsigned long long above = l >= 63 ? 0ull : (~0ull << (l + 1));
                    m = __ballot(top > dv) & above;
                }
            }
            par_heap_reorder(L.hK, k);
        }
        __syncthreads();
        GH_TT(5);
        for (int i = tid; i < k; i += NT) {
            const uint2 e = L.hK[1 + i];
            const int j = (int)e.y;
            const float v = __uint_as_float(e.x);
            od[i] = j < 0 ? a.neutral : (L2 ? v : -v);
            ol[i] = j < 0 ? -1 : L.id[j];
        }
        // the recall-stage table: the heap's entries, best first; equal distances in scan order (the order of the
        // regular kernels' table -- the reference never sorts its R-heap on this path)
        __syncthreads();
        for (int j = tid; j < R; j += NT) {
            const uint2 e = L.hR[1 + j];
            // (empty slots: distinct items that sort last -- block_rank_sort needs pairwise distinct items)
            L.it[j] = (int)e.y < 0 ? (0xffffffff80000000ull | (unsigned)j)
                                   : (((unsigned long long)f2key(__uint_as_float(e.x)) << 32) | e.y);
        }
        block_rank_sort<NT, (MAXK + NT - 1) / NT>(L.it, R);   // (L.it holds max(STG, MAXK) items: tie_replay_lds_bytes_)
        for (int j = tid; j < R; j += NT) {
            const unsigned long long it = L.it[j];
            const bool empty = (uint32_t)(it >> 32) == 0xffffffffu;
            const float v = key2f((uint32_t)(it >> 32));
            a.cand_dis[(int64_t)q * R + j] = empty ? (L2 ? INFINITY : -INFINITY) : (L2 ? v : -v);
            a.cand_ids[(int64_t)q * R + j] = empty ? -1 : pos_to_id((int)(uint32_t)it);
        }
        GH_TT(6);
    } else {
        // without rank: heap_reorder of the R-heap is the result (gamma_index_ivfpq.cc:681-696)
        if (wv == 0) par_heap_reorder(L.hR, R);
        __syncthreads();
        GH_TT(4);
        for (int j = tid; j < R; j += NT) {
            const uint2 e = L.hR[1 + j];
            const int ps = (int)e.y;
            const int64_t id = pos_to_id(ps);
            const float v = __uint_as_float(e.x);
            L.id[j] = id;
            L.ex[j] = L2 ? v : -v;
            a.cand_dis[(int64_t)q * R + j] = ps < 0 ? (L2 ? INFINITY : -INFINITY) : (L2 ? v : -v);
            a.cand_ids[(int64_t)q * R + j] = id;
        }
        __syncthreads();
        if (wv == 0) {
            // first k entries inside the score window
            int taken = 0;
            for (int j0 = 0; j0 < R && taken < k; j0 += 64) {
                const int j = j0 + lane;
                const float dis = j < R ? L.ex[j] : 0.f;
                const bool ok = j < R && L.id[j] >= 0 && dis <= a.max_score && dis >= a.min_score;
                const unsigned long long bal = __ballot(ok);
                const int slot = taken + __popcll(bal & ((1ull << lane) - 1ull));
                if (ok && slot < k) {
                    od[slot] = dis;
                    ol[slot] = L.id[j];
                }
                taken += __popcll(bal);
            }
            for (int i = min(taken, k) + lane; i < k; i += 64) {
                od[i] = a.neutral;
                ol[i] = -1;
            }
        }
        GH_TT(5);
    }
#undef GH_TT
}

}  // namespace gh
