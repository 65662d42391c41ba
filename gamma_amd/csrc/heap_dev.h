// heap_dev.h -- faiss's binary heaps (faiss:utils/Heap.h:46-131,300-330) replayed on the device, CMax form
// (cmp(a, b) = a > b: the heap of the k SMALLEST values, its largest at the root; the inner-product CMin heap on
// v is this heap on -v, negation is exact).
//
// The array a heap ends with depends on its whole history, so reproducing the reference inside exact ties means
// replaying every accepted candidate -- a chain of dependent sifts that no amount of lanes shortens by itself.
// What the 64 lanes CAN do:
//   * skip: `if (top > dis)` of 64 candidates at a time is one ballot (HeapWalk::accept);
//   * pipeline: heap_replace_top of candidate n+1 may start as soon as candidate n has left the two levels
//     below the root.  Lanes 0..7 are operation slots; every tick() moves all operations in flight one level
//     down (one 16-byte LDS read of the two children, one 8-byte write), a new operation enters every second
//     tick.  An operation two levels behind its predecessor reads nodes the predecessor wrote one tick earlier,
//     never nodes it is about to write, so the heap array goes through exactly the states of the sequential
//     code.  One candidate costs two LDS round trips instead of one per level (8 for recall_num = 200).
// The heap lives in LDS as h[i] = (value bits, payload), node i 1-based as in faiss after its `bh_val--`;
// h[0] is a dump slot for idle lanes; the array needs K + 2 entries and 16-byte alignment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gh {

constexpr float kHeapFltMax = 3.402823466e+38f;

// heap_heapify on an empty heap: (neutral, -1) everywhere (Heap.h:195-217)
__device__ __forceinline__ void heap_fill(uint2* h, int K, int tid, int nthreads) {
    for (int i = tid; i < K + 2; i += nthreads) h[i] = make_uint2(__float_as_uint(kHeapFltMax), 0xffffffffu);
}

__device__ __forceinline__ float hw_readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(l)));
}
__device__ __forceinline__ int hw_readlane_i(int v, int l) {
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l));
}

// One wave.  All 64 lanes call every member together.
struct HeapWalk {
    char* hb;    // the heap array as bytes: node i at hb + 8 i
    int K8, KR;  // 8 K; byte offset of the last aligned child pair that may be read (clamp of the read address)
    float top;   // value at the root (wave-uniform)
    int nin;     // operations started (wave-uniform); slot of the next one = nin & 7
    // the operation slot of this lane (lanes 0..7; the other lanes stay idle): byte offset of its node (0 = idle),
    // the value / payload it carries, the value it wrote last
    int nb, pay;
    float val, wv;

    __device__ __forceinline__ void begin(uint2* h_, int K_) {
        hb = reinterpret_cast<char*>(h_);
        K8 = K_ * 8;
        KR = (K_ & ~1) * 8;
        top = __uint_as_float(h_[1].x);
        nin = 0;
        nb = 0;
        pay = 0;
        val = 0.f;
        wv = 0.f;
    }
    // every operation in flight goes one level down (heap_replace_top's loop body, Heap.h:110-127); an idle lane
    // reads the pair (h[0], h[1]) and writes the dump slot h[0]
    __device__ __forceinline__ void tick() {
        const int cb = nb << 1;                               // byte offset of the children pair (i1 = 2 i)
        const uint4 c = *reinterpret_cast<const uint4*>(hb + min(cb, KR));
        const float v1 = __uint_as_float(c.x), v2 = __uint_as_float(c.z);
        const bool pick1 = !(cb < K8) || v1 > v2;             // i2 == k + 1 || cmp(val[i1], val[i2])
        const float cv = pick1 ? v1 : v2;
        const unsigned cp = pick1 ? c.y : c.w;
        const bool stop = !(cb <= K8) || val > cv || nb == 0; // i1 > k, or cmp(val, child): the value stays here
        wv = stop ? val : cv;
        *reinterpret_cast<uint2*>(hb + nb) = make_uint2(__float_as_uint(wv), stop ? (unsigned)pay : cp);
        nb = stop ? 0 : (pick1 ? cb : cb + 8);
    }
    // heap_replace_top(val, payload)
    __device__ __forceinline__ void replace_top(float v, int p) {
        const int slot = nin & 7;
        const bool me = (int)(threadIdx.x & 63) == slot;
        nb = me ? 8 : nb;
        val = me ? v : val;
        pay = me ? p : pay;
        nin++;
        tick();
        top = hw_readlane_f(wv, slot);   // what the root holds now
        tick();
    }
    // one block of <= 64 candidates in stream order, one per lane: `if (cmp(top, dis)) heap_replace_top`
    // (dv: smaller is better; filtered entries carry +inf and never beat the top).  The top only ever decreases, so a
    // candidate that has lost against it once need not be looked at again.
    __device__ __forceinline__ void accept(bool ok, float dv, int p) {
        unsigned long long m = __ballot(ok && top > dv);
        while (m) {
            const int l = (int)__ffsll((long long)m) - 1;
            replace_top(hw_readlane_f(dv, l), hw_readlane_i(p, l));
            m &= m - 1ull;
            m &= __ballot(top > dv);
        }
    }
    // let the operations in flight finish (a heap of K nodes has at most 32 - clz(K) levels)
    __device__ __forceinline__ void drain() {
        const int levels = 32 - __clz(K8 > 8 ? K8 >> 3 : 1);
        for (int t = 0; t < levels; t++) tick();
        __builtin_amdgcn_wave_barrier();
    }
};

// ---- sequential forms, for the short phases (k-heap of compute_dis, heap_reorder).  Every lane of the calling
//      wave executes the same accesses (LDS broadcast); values are pulled into SGPRs so that the branches are
//      scalar. ----
__device__ __forceinline__ float hs_f(unsigned x) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)x)); }
__device__ __forceinline__ unsigned hs_u(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }

// the sift of heap_pop / heap_replace_top over k nodes
__device__ __forceinline__ void heap_sift_down_seq(uint2* h, int k, float val, unsigned pay) {
    int i = 1;
    for (;;) {
        const int i1 = i << 1;
        if (i1 > k) break;
        const uint4 c = *reinterpret_cast<const uint4*>(h + i1);
        const float v1 = hs_f(c.x), v2 = hs_f(c.z);
        const bool pick1 = i1 == k || v1 > v2;
        const float cv = pick1 ? v1 : v2;
        if (val > cv) break;
        h[i] = make_uint2(__float_as_uint(cv), pick1 ? hs_u(c.y) : hs_u(c.w));
        i = pick1 ? i1 : i1 + 1;
    }
    h[i] = make_uint2(__float_as_uint(val), pay);
}
// heap_pop (Heap.h:46-72): the last element sifts down from the root; slot k keeps its stale copy
__device__ __forceinline__ void heap_pop_seq(uint2* h, int k) {
    const uint2 last = h[k];
    heap_sift_down_seq(h, k, hs_f(last.x), hs_u(last.y));
}
// heap_push into slot k (Heap.h:77-100)
__device__ __forceinline__ void heap_push_seq(uint2* h, int k, float val, unsigned pay) {
    int i = k;
    while (i > 1) {
        const int f = i >> 1;
        const uint2 pf = h[f];
        if (!(val > hs_f(pf.x))) break;
        h[i] = make_uint2(hs_u(pf.x), hs_u(pf.y));
        i = f;
    }
    h[i] = make_uint2(__float_as_uint(val), pay);
}
// heap_reorder (Heap.h:300-330): sorted best first into h[1..k], (FLT_MAX, -1) padded; returns the number of real
// entries (payload != -1)
// heap_reorder, second half: the ii real entries sit at the end of h[1..k]; memmove to the front, 64 entries per step (a
// step reads before it writes, and writes below what the next step reads: dst <= src), then the padding
__device__ __forceinline__ int heap_reorder_tail(uint2* h, int k, int ii) {
    const int lane = threadIdx.x & 63;
    if (ii < k) {
        for (int i0 = 0; i0 < ii; i0 += 64) {
            const int i = i0 + lane;
            uint2 t = make_uint2(0u, 0u);
            if (i < ii) t = h[1 + k - ii + i];
            if (i < ii) h[1 + i] = t;
        }
        for (int i = ii + lane; i < k; i += 64) h[1 + i] = make_uint2(__float_as_uint(kHeapFltMax), 0xffffffffu);
    }
    __builtin_amdgcn_wave_barrier();
    return ii;
}
__device__ __forceinline__ int heap_reorder_seq(uint2* h, int k) {
    int ii = 0;
    for (int i = 0; i < k; i++) {
        const uint2 r = h[1];
        const float val = hs_f(r.x);
        const unsigned idv = hs_u(r.y);
        heap_pop_seq(h, k - i);
        h[k - ii] = make_uint2(__float_as_uint(val), idv);   // 0-based slot k - ii - 1
        if (idv != 0xffffffffu) ii++;
    }
    return heap_reorder_tail(h, k, ii);
}

// ---- the same sequential forms with the heap in REGISTERS: node i (1-based) in lane i & 63 of register i >> 6, read
//      with v_readlane and written by the owning lane -- a level of a sift costs a few scalar instructions instead of an
//      LDS round trip.  For the heaps that take their candidates one at a time (heap_pop + heap_push: the k-heap of
//      compute_dis, the IVFFLAT / flat scanners' heap).  K <= 64 NREG - 1.  All lanes of the wave call every member.
template <int NREG>
struct RegHeap {
    unsigned kv[NREG], pv[NREG];
    __device__ __forceinline__ void fill() {
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            kv[r] = __float_as_uint(kHeapFltMax);
            pv[r] = 0xffffffffu;
        }
    }
    __device__ __forceinline__ unsigned getk(int i) const {
        const int l = __builtin_amdgcn_readfirstlane(i & 63), rr = __builtin_amdgcn_readfirstlane(i >> 6);
        unsigned v = (unsigned)__builtin_amdgcn_readlane((int)kv[0], l);
#pragma unroll
        for (int r = 1; r < NREG; r++) {
            const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)kv[r], l);
            v = rr == r ? t : v;
        }
        return v;
    }
    __device__ __forceinline__ unsigned getp(int i) const {
        const int l = __builtin_amdgcn_readfirstlane(i & 63), rr = __builtin_amdgcn_readfirstlane(i >> 6);
        unsigned v = (unsigned)__builtin_amdgcn_readlane((int)pv[0], l);
#pragma unroll
        for (int r = 1; r < NREG; r++) {
            const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)pv[r], l);
            v = rr == r ? t : v;
        }
        return v;
    }
    __device__ __forceinline__ void set(int i, unsigned kb, unsigned pb) {
        const int l = __builtin_amdgcn_readfirstlane(i & 63), rr = __builtin_amdgcn_readfirstlane(i >> 6);
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            const bool me = rr == r && (int)(threadIdx.x & 63) == l;
            kv[r] = me ? kb : kv[r];
            pv[r] = me ? pb : pv[r];
        }
    }
    __device__ __forceinline__ float top() const { return __uint_as_float(getk(1)); }
    // heap_sift_down_seq over the first n nodes
    __device__ __forceinline__ void sift_down(int n, float val, unsigned pay) {
        int i = 1;
        for (;;) {
            const int i1 = i << 1;
            if (i1 > n) break;
            const float v1 = __uint_as_float(getk(i1)), v2 = __uint_as_float(getk(i1 + 1));   // i1 + 1 <= 64 NREG - 1
            const bool pick1 = i1 == n || v1 > v2;
            const float cv = pick1 ? v1 : v2;
            if (val > cv) break;
            const int c = pick1 ? i1 : i1 + 1;
            set(i, __float_as_uint(cv), getp(c));
            i = c;
        }
        set(i, __float_as_uint(val), pay);
    }
    __device__ __forceinline__ void pop(int n) { sift_down(n, __uint_as_float(getk(n)), getp(n)); }
    __device__ __forceinline__ void push(int n, float val, unsigned pay) {
        int i = n;
        while (i > 1) {
            const int f = i >> 1;
            const unsigned fk = getk(f);
            if (!(val > __uint_as_float(fk))) break;
            set(i, fk, getp(f));
            i = f;
        }
        set(i, __float_as_uint(val), pay);
    }
    // heap_reorder's pops (the array then goes to LDS for heap_reorder_tail); returns the number of real entries
    __device__ __forceinline__ int reorder_pops(int k) {
        int ii = 0;
        for (int i = 0; i < k; i++) {
            const unsigned rk = getk(1), rp = getp(1);
            pop(k - i);
            set(k - ii, rk, rp);
            if (rp != 0xffffffffu) ii++;
        }
        return ii;
    }
    // h[i] -> node i, i = 1..k (an LDS heap taken over, e.g. for heap_reorder's pops)
    __device__ __forceinline__ void load(const uint2* h, int k) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            const int i = r * 64 + lane;
            const uint2 e = (i >= 1 && i <= k) ? h[i] : make_uint2(__float_as_uint(kHeapFltMax), 0xffffffffu);
            kv[r] = e.x;
            pv[r] = e.y;
        }
    }
    // node i -> h[i], i = 1..k
    __device__ __forceinline__ void dump(uint2* h, int k) const {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int r = 0; r < NREG; r++) {
            const int i = r * 64 + lane;
            if (i >= 1 && i <= k) h[i] = make_uint2(kv[r], pv[r]);
        }
        __builtin_amdgcn_wave_barrier();
    }
};

// ---- the sifts with all 64 lanes (LDS heap of k <= 128 NPL entries: lane l holds the internal nodes l + 1 + 64 t).
//      A sift cannot be cut short -- but which way it goes does not depend on what is sifted: from every internal node
//      the walk continues to its LARGER child (Heap.h:56-70), and the sifted value only decides where it stops.  So one
//      step does every node at once: each lane reads the two children of its node(s) (one 16-byte LDS read); two ballots
//      say for every node which child is larger and whether the sifted value would go on below it; "the walk reaches
//      node i" = every step from the root towards i is the one the walk takes, an AND along i's ancestors, evaluated for
//      all nodes in three doubling rounds (ballot + shift, no memory, no loop over levels -- a scalar walk over the masks
//      costs ~100 cycles a level on this chip); the lanes of the nodes the walk passes write their larger child's entry
//      into their own node in one store.  heap_push: the ancestors of slot k are k >> 1, k >> 2, ..: lane j reads the
//      j-th, one ballot says how far the new value rises, the lanes below that store their ancestor one level down.
//      Measured (tools/exp/heap_bench.hip, one wave, k = 100): pop + push 1018 ns sequential, 768 with a scalar walk,
//      450 with the doubling rounds (590 at k = 200 against 1150).  The heap array goes through exactly the states of the sequential code (same
//      moves, same final slot).  All lanes of ONE wave call every member together.
template <int NPL>   // 1 or 2
struct ParHeap {
    static_assert(NPL == 1 || NPL == 2, "masks of 64 NPL nodes");
    struct Mask {
        unsigned long long w[NPL];
        // bit of node n (1-based); node 0 = "above the root": true
        __device__ __forceinline__ bool at(int n) const {
            if (n < 1) return true;
            const unsigned long long m = (NPL == 2 && n > 64) ? w[NPL - 1] : w[0];
            return ((m >> ((n - 1) & 63)) & 1ull) != 0ull;
        }
    };
    static __device__ __forceinline__ Mask vote(const bool (&b)[NPL]) {
        Mask m;
#pragma unroll
        for (int t = 0; t < NPL; t++) m.w[t] = __ballot(b[t]);
        return m;
    }
    // the sift of heap_pop / heap_replace_top over nodes 1..k with (val, pay) entering at the root; returns the root's value
    static __device__ __forceinline__ float sift_down(uint2* h, int k, float val, unsigned pay) {
        const int lane = threadIdx.x & 63;
        unsigned cv[NPL], cp[NPL];
        bool in[NPL], p1[NPL], le[NPL], g[NPL];
#pragma unroll
        for (int t = 0; t < NPL; t++) {
            const int i = lane + 64 * t + 1;
            in[t] = 2 * i <= k;
            const uint4 c = in[t] ? *reinterpret_cast<const uint4*>(h + 2 * i) : make_uint4(0u, 0u, 0u, 0u);
            const float v1 = __uint_as_float(c.x), v2 = __uint_as_float(c.z);
            p1[t] = in[t] && (2 * i == k || v1 > v2);
            const float v = p1[t] ? v1 : v2;
            cv[t] = __float_as_uint(v);
            cp[t] = p1[t] ? c.y : c.w;
            le[t] = in[t] && !(val > v);   // the sifted value goes on below this node
        }
        const Mask P1 = vote(p1), LE = vote(le);
        // g(i): the walk, IF it reaches i's parent, steps to i
#pragma unroll
        for (int t = 0; t < NPL; t++) {
            const int i = lane + 64 * t + 1, p = i >> 1;
            g[t] = i == 1 || (LE.at(p) && P1.at(p) == ((i & 1) == 0));
        }
        // reaches(i) = g(i) & g(i >> 1) & g(i >> 2) & ..: three doubling rounds cover 8 levels (i <= 255)
        Mask G = vote(g);
#pragma unroll
        for (int t = 0; t < NPL; t++) g[t] = g[t] && G.at((lane + 64 * t + 1) >> 1);
        G = vote(g);
#pragma unroll
        for (int t = 0; t < NPL; t++) g[t] = g[t] && G.at((lane + 64 * t + 1) >> 2);
        G = vote(g);
        bool mv[NPL];
#pragma unroll
        for (int t = 0; t < NPL; t++) mv[t] = g[t] && G.at((lane + 64 * t + 1) >> 4) && le[t];   // passed: the child moves up
        const Mask MV = vote(mv);
        // where the value lands: below the deepest node passed (nodes on the walk grow with depth), or at the root
        int cur = 1;
        if (NPL == 2 && MV.w[NPL - 1] != 0ull) {
            const int m = 128 - __builtin_clzll(MV.w[NPL - 1]);
            cur = 2 * m + (P1.at(m) ? 0 : 1);
        } else if (MV.w[0] != 0ull) {
            const int m = 64 - __builtin_clzll(MV.w[0]);
            cur = 2 * m + (P1.at(m) ? 0 : 1);
        }
        const float root = (MV.w[0] & 1ull) ? __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)cv[0], 0)) : val;
#pragma unroll
        for (int t = 0; t < NPL; t++)
            if (mv[t]) h[lane + 64 * t + 1] = make_uint2(cv[t], cp[t]);
        if (lane == 0) h[cur] = make_uint2(__float_as_uint(val), pay);
        __builtin_amdgcn_wave_barrier();
        return root;
    }
    // heap_pop over k nodes (Heap.h:46-72); returns the root's value afterwards
    static __device__ __forceinline__ float pop(uint2* h, int k) {
        const uint2 last = h[k];
        return sift_down(h, k, hs_f(last.x), hs_u(last.y));
    }
    // heap_push into slot k (Heap.h:77-100); true when the new entry became the root
    static __device__ __forceinline__ bool push(uint2* h, int k, float val, unsigned pay) {
        const int lane = threadIdx.x & 63;
        const int a = lane < 31 ? k >> (lane + 1) : 0;   // the (lane + 1)-th ancestor of slot k
        const uint2 e = a >= 1 ? h[a] : make_uint2(0u, 0u);
        const unsigned long long up = __ballot(a >= 1 && val > __uint_as_float(e.x));
        const int t = (int)__ffsll((long long)~up) - 1;   // ancestors the value passes: 0 .. t - 1
        if (lane < t) h[k >> lane] = e;
        const int slot = k >> t;
        if (lane == 0) h[slot] = make_uint2(__float_as_uint(val), pay);
        __builtin_amdgcn_wave_barrier();
        return slot == 1;
    }
    // heap_reorder (Heap.h:300-330) with parallel pops
    static __device__ __forceinline__ int reorder(uint2* h, int k) {
        int ii = 0;
        for (int i = 0; i < k; i++) {
            const uint2 r = h[1];
            const float val = hs_f(r.x);
            const unsigned idv = hs_u(r.y);
            (void)pop(h, k - i);
            h[k - ii] = make_uint2(__float_as_uint(val), idv);   // 0-based slot k - ii - 1
            __builtin_amdgcn_wave_barrier();
            if (idv != 0xffffffffu) ii++;
        }
        return heap_reorder_tail(h, k, ii);
    }
};
// by heap size (k <= 256; beyond: the sequential forms -- eight nodes per lane lose to them)
constexpr int kParHeapMaxK = 256;
__device__ __forceinline__ float par_heap_pop(uint2* h, int k) {
    if (k <= 128) return ParHeap<1>::pop(h, k);
    if (k <= 256) return ParHeap<2>::pop(h, k);
    heap_pop_seq(h, k);
    return hs_f(h[1].x);
}
__device__ __forceinline__ int par_heap_reorder(uint2* h, int k) {
    if (k <= 128) return ParHeap<1>::reorder(h, k);
    if (k <= 256) return ParHeap<2>::reorder(h, k);
    return heap_reorder_seq(h, k);
}
__device__ __forceinline__ bool par_heap_push(uint2* h, int k, float val, unsigned pay) { return ParHeap<1>::push(h, k, val, pay); }
// the sift of heap_replace_top over k nodes
__device__ __forceinline__ float par_heap_replace_top(uint2* h, int k, float val, unsigned pay) {
    if (k <= 128) return ParHeap<1>::sift_down(h, k, val, pay);
    if (k <= 256) return ParHeap<2>::sift_down(h, k, val, pay);
    heap_sift_down_seq(h, k, val, pay);
    return hs_f(h[1].x);
}

}  // namespace gh
