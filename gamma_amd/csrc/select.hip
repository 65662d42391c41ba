// select.hip -- k-selection for gfx950 (a7: the device counterpart of faiss's binary heap,
// faiss:utils/Heap.h:103-131, used by KnnSearchResults::add, gamma_index_ivfpq.h:363-369).
//
// One 256-thread workgroup per row segment.  Keys are order-preserving uint32 images of the
// fp32 distances.  Algorithm (K smallest keys of n):
//   1. min / max key of the segment (wave shuffles + LDS)
//   2. 2048-bucket histogram over [min, max] (LDS atomics), block scan -> bucket B holding
//      the K-th key; everything in lower buckets is selected outright
//   3. (only if bucket B is still crowded) repeat 2 inside B with 2048 finer buckets
//   4. one collect pass: lower buckets -> result, bucket B -> small candidate list in LDS
//   5. bitonic-sort the candidates on (key, position), take what is still needed, then
//      bitonic-sort the K results on (key, position)
// Equal distances therefore come out in scan order (probe order, then list order) -- the
// deterministic counterpart of the reference heap, which keeps the same multiset and differs
// only in the order / membership inside exact ties.
// Rows that fit (n <= 256*NPT) keep their keys in registers, so steps 1-4 never re-read memory.
// SMALLEST=true: K smallest (CMax heap, L2); false: K largest (CMin heap, IP).
// Sentinels (+inf / -inf) mark filtered entries and come back as pos = -1.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "block_utils.h"
#include "heap_dev.h"
#include "reservoir_dev.h"
#include "tie_dev.h"
#include "device_math.h"
#include "kernels.h"

namespace gh {

namespace {
constexpr int NB = 2048;   // histogram buckets per level
constexpr int CAP = 1024;  // candidate capacity for the threshold bucket

template <bool SMALLEST>
__device__ __forceinline__ uint32_t sel_key(float v) {
    uint32_t k = f2key(v);
    return SMALLEST ? k : ~k;
}

__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* a, int npad) {
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (npad >> 1); t += 256) {
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
}  // namespace

template <bool SMALLEST, int NPT>
__global__ __launch_bounds__(256) void k_select2(const float* __restrict__ vals, int64_t seg_stride,
                                                 const int* __restrict__ seg_len, int fixed_len, int K,
                                                 int Kpad, float* __restrict__ out_vals,
                                                 int* __restrict__ out_pos,
                                                 const uint8_t* __restrict__ only) {
    if (only && !only[blockIdx.x]) return;         // rows another kernel has already selected
    extern __shared__ unsigned long long s_dyn[];  // items[Kpad] | cand[CAP]
    unsigned long long* s_items = s_dyn;
    unsigned long long* s_cand = s_dyn + Kpad;
    __shared__ int s_hist[NB];
    __shared__ int s_w[4];
    __shared__ uint32_t s_red[8];
    __shared__ int s_misc[8];
    const int tid = threadIdx.x;
    const int seg = blockIdx.x;
    const int n = seg_len ? seg_len[seg] : fixed_len;
    const float* v = vals + (int64_t)seg * seg_stride;

    uint32_t kreg[NPT > 0 ? NPT : 1];
    if (NPT > 0) {
#pragma unroll
        for (int j = 0; j < NPT; j++) {
            const int i = tid + 256 * j;
            kreg[j] = i < n ? sel_key<SMALLEST>(v[i]) : 0xffffffffu;
        }
    }
    // visit every (key, position) this thread owns
    auto for_each = [&](auto&& body) {
        if (NPT > 0) {
#pragma unroll
            for (int j = 0; j < NPT; j++) {
                const int i = tid + 256 * j;
                if (i < n) body(kreg[j], i);
            }
        } else {
            // streaming rows: 8 independent loads in flight per thread before any use
            int i0 = tid;
            for (; i0 + 7 * 256 < n; i0 += 8 * 256) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = v[i0 + u * 256];
#pragma unroll
                for (int u = 0; u < 8; u++) body(sel_key<SMALLEST>(t[u]), i0 + u * 256);
            }
            for (; i0 < n; i0 += 256) body(sel_key<SMALLEST>(v[i0]), i0);
        }
    };

    for (int i = tid; i < Kpad; i += 256) s_items[i] = ~0ull;
    if (tid < 8) s_misc[tid] = 0;
    __syncthreads();

    if (n <= K) {
        for_each([&](uint32_t key, int i) { s_items[i] = ((unsigned long long)key << 32) | (unsigned)i; });
    } else {
        // ---- 1. min / max ----
        uint32_t mn = 0xffffffffu, mx = 0u;
        for_each([&](uint32_t key, int) {
            mn = key < mn ? key : mn;
            mx = key > mx ? key : mx;
        });
        mn = wave_min_u32(mn);
        mx = wave_max_u32(mx);
        if ((tid & 63) == 0) {
            s_red[tid >> 6] = mn;
            s_red[4 + (tid >> 6)] = mx;
        }
        __syncthreads();
        mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
        // ---- 2./3. bucket histogram levels ----
        uint32_t lo = mn;
        const uint32_t range = mx - mn;
        int s = range >= (uint32_t)NB ? (32 - __clz((int)range)) - 11 : 0;  // (range >> s) < NB
        uint32_t nbins = (range >> s) + 1;                                   // <= NB
        int kk = K;      // still needed among keys >= lo
        int cnt = 0;     // population of the threshold bucket
        for (;;) {
            for (int i = tid; i < NB; i += 256) s_hist[i] = 0;
            __syncthreads();
            for_each([&](uint32_t key, int) {
                if (key >= lo) {
                    const uint32_t b = (key - lo) >> s;
                    if (b < nbins) atomicAdd(&s_hist[b], 1);
                }
            });
            __syncthreads();
            // each thread owns 8 consecutive bins
            int c8 = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) c8 += s_hist[tid * 8 + j];
            int tot;
            const int ex = block_excl_scan256(c8, s_w, tot);
            if (ex < kk && kk <= ex + c8) {
                int run = ex;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int c = s_hist[tid * 8 + j];
                    if (run < kk && kk <= run + c) {
                        s_misc[0] = tid * 8 + j;
                        s_misc[1] = run;
                        s_misc[2] = c;
                    }
                    run += c;
                }
            }
            __syncthreads();
            const int B = s_misc[0];
            kk -= s_misc[1];
            cnt = s_misc[2];
            lo += (uint32_t)B << s;
            __syncthreads();
            if (cnt <= CAP || s == 0) break;
            // refine inside bucket B: window [lo, lo + 2^s) split into 2^(s - s2) <= NB bins
            const int s2 = s > 11 ? s - 11 : 0;
            nbins = 1u << (s - s2);
            s = s2;
        }
        // threshold bucket = [lo, lo + 2^s); kk of its cnt members are needed
        const int n_less = K - kk;
        const uint32_t width_m1 = s >= 32 ? 0xffffffffu : ((1u << s) - 1u);
        if (cnt <= CAP) {
            for (int i = tid; i < CAP; i += 256) s_cand[i] = ~0ull;
            __syncthreads();
            for_each([&](uint32_t key, int i) {
                if (key < lo) {
                    const int slot = atomicAdd(&s_misc[3], 1);
                    s_items[slot] = ((unsigned long long)key << 32) | (unsigned)i;
                } else if (key - lo <= width_m1) {
                    const int slot = atomicAdd(&s_misc[4], 1);
                    s_cand[slot] = ((unsigned long long)key << 32) | (unsigned)i;
                }
            });
            __syncthreads();
            if (cnt == kk) {
                for (int i = tid; i < cnt; i += 256) s_items[n_less + i] = s_cand[i];
            } else {
                block_rank_sort<256, 4>(s_cand, cnt);   // cnt <= CAP distinct items
                for (int i = tid; i < kk; i += 256) s_items[n_less + i] = s_cand[i];
            }
        } else {
            // s == 0: more than CAP copies of one key straddle the boundary; keep the first kk
            // in scan order (ordered block scan; rare)
            int running = 0;
            for (int i0 = 0; i0 < n; i0 += 256) {
                const int i = i0 + tid;
                uint32_t key = 0xffffffffu;
                const bool in = i < n;
                if (in) key = sel_key<SMALLEST>(v[i]);
                if (in && key < lo) {
                    const int slot = atomicAdd(&s_misc[3], 1);
                    s_items[slot] = ((unsigned long long)key << 32) | (unsigned)i;
                }
                const int flag = (in && key == lo) ? 1 : 0;
                int tot;
                const int ex = block_excl_scan256(flag, s_w, tot);
                const int rank = running + ex;
                if (flag && rank < kk) s_items[n_less + rank] = ((unsigned long long)key << 32) | (unsigned)i;
                running += tot;
            }
        }
    }
    __syncthreads();
    {
        const int nres = n < K ? n : K;               // s_items[0..nres) hold distinct items
        if (nres <= 2048) {
            block_rank_sort<256, 8>(s_items, nres);   // slots >= nres stay ~0
        } else {
            bitonic_sort_lds(s_items, Kpad);
        }
    }
    const float sentinel = SMALLEST ? INFINITY : -INFINITY;
    for (int r = tid; r < K; r += 256) {
        const unsigned long long it = s_items[r];
        float val = sentinel;
        int pos = -1;
        if (it != ~0ull) {
            pos = (int)(uint32_t)it;
            val = v[pos];
            if (val == sentinel) pos = -1;
        }
        out_vals[(int64_t)seg * K + r] = val;
        out_pos[(int64_t)seg * K + r] = pos;
    }
}

// ------------------------------------------------------------------------------------
// Streaming variant for long rows (n > 4096, K <= 1024): ONE pass over the row.  Elements
// are visited in chunks of 2048; an element is kept only if it beats the running threshold
// tau (the K-th best seen so far), survivors go to an LDS buffer by wave-aggregated atomics,
// and the buffer is compacted (bucket select, keeps the K best, tightens tau) whenever the
// next chunk could overflow it.  The first chunk -- the closest probed lists, which hold most
// of the final result -- acts as the sample that sets tau.  Same result as k_select2:
// the K smallest (key, position) pairs, sorted.
// ------------------------------------------------------------------------------------
namespace {
// survivor buffer entries: CAPS = 2048 (16 KB -> 6 workgroups / CU) up to K = 1024, 4096 up to K = 2048 (round 5: full-size C5 needs
// recall_num 1200 for the metric's recall bar, and k_select2 took 5.4 ms of its 31 ms step)
constexpr int ST_CHUNK = 1024;         // elements per threshold check

// keep the K smallest items of buf[0..n) in buf[0..K); returns the largest kept key
template <int ST_CAPS>
__device__ uint32_t st_compact(unsigned long long* buf, int n, int K, int* s_hist, int* s_w,
                               uint32_t* s_red, int* s_misc) {
    constexpr int ST_IPT = ST_CAPS / 256;  // buffer items per thread during compaction
    const int tid = threadIdx.x;
    unsigned long long it[ST_IPT];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int j = 0; j < ST_IPT; j++) {
        const int i = tid + 256 * j;
        it[j] = i < n ? buf[i] : ~0ull;
        if (i < n) {
            const uint32_t key = (uint32_t)(it[j] >> 32);
            mn = key < mn ? key : mn;
            mx = key > mx ? key : mx;
        }
    }
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    if ((tid & 63) == 0) {
        s_red[tid >> 6] = mn;
        s_red[4 + (tid >> 6)] = mx;
    }
    for (int i = tid; i < NB; i += 256) s_hist[i] = 0;
    if (tid < 8) s_misc[tid] = 0;
    __syncthreads();
    mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const uint32_t range = mx - mn;
    const int s = range >= (uint32_t)NB ? (32 - __clz((int)range)) - 11 : 0;
#pragma unroll
    for (int j = 0; j < ST_IPT; j++)
        if (tid + 256 * j < n) atomicAdd(&s_hist[((uint32_t)(it[j] >> 32) - mn) >> s], 1);
    __syncthreads();
    int c8 = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) c8 += s_hist[tid * 8 + j];
    int tot;
    const int ex = block_excl_scan256(c8, s_w, tot);
    if (ex < K && K <= ex + c8) {
        int run = ex;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = s_hist[tid * 8 + j];
            if (run < K && K <= run + c) {
                s_misc[0] = tid * 8 + j;
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
#pragma unroll
        for (int j = 0; j < ST_IPT; j++)
            if (tid + 256 * j < n && (((uint32_t)(it[j] >> 32) - mn) >> s) <= B)
                buf[atomicAdd(&s_misc[3], 1)] = it[j];
        __syncthreads();
    } else if (cnt <= 1024) {
        int cpad = 2;
        while (cpad < cnt) cpad <<= 1;
        unsigned long long* cand = buf + (ST_CAPS - cpad);   // disjoint from [0, below): K <= ST_CAPS / 2, cpad <= 1024
#pragma unroll
        for (int j = 0; j < ST_IPT; j++) {
            if (tid + 256 * j < n) {
                const uint32_t b = ((uint32_t)(it[j] >> 32) - mn) >> s;
                if (b < B) buf[atomicAdd(&s_misc[3], 1)] = it[j];
                else if (b == B) cand[atomicAdd(&s_misc[4], 1)] = it[j];
            }
        }
        block_rank_sort<256, 4>(cand, cnt);
        for (int i = tid; i < need; i += 256) buf[below + i] = cand[i];
        __syncthreads();
    } else {
#pragma unroll
        for (int j = 0; j < ST_IPT; j++) buf[tid + 256 * j] = it[j];
        bitonic_sort_lds(buf, ST_CAPS);
    }
    uint32_t kmx = 0u;
    for (int i = tid; i < K; i += 256) {
        const uint32_t key = (uint32_t)(buf[i] >> 32);
        kmx = key > kmx ? key : kmx;
    }
    kmx = wave_max_u32(kmx);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = kmx;
    __syncthreads();
    return max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
}
}  // namespace

template <bool SMALLEST, int ST_CAPS = 2048>
__global__ __launch_bounds__(256) void k_select_stream(const float* __restrict__ vals, int64_t seg_stride,
                                                       const int* __restrict__ seg_len, int fixed_len,
                                                       int K, int Kpad, float* __restrict__ out_vals,
                                                       int* __restrict__ out_pos,
                                                       const uint8_t* __restrict__ only) {
    if (only && !only[blockIdx.x]) return;         // rows another kernel has already selected
    __shared__ unsigned long long s_buf[ST_CAPS];
    __shared__ int s_hist[NB];
    __shared__ int s_w[4];
    __shared__ uint32_t s_red[8];
    __shared__ int s_misc[8];
    __shared__ int s_cnt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int seg = blockIdx.x;
    const int n = seg_len ? seg_len[seg] : fixed_len;
    const float* v = vals + (int64_t)seg * seg_stride;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    uint32_t tau = 0xffffffffu;
    // 16-byte loads: lane holds 4 consecutive elements, 2 float4 per thread and chunk of 2048
    // (the row base is 16-byte aligned: seg_stride % 4 == 0 is required by the launcher).
    // Reads past n inside the last float4 group are masked by the i < n test; reads past the
    // row are avoided by clamping the group index.
    const float4* v4 = reinterpret_cast<const float4*>(v);
    const int n4 = (n + 3) >> 2;               // float4 groups that contain an element
    // two chunks (of 1024 = 256 lanes x float4) are kept in flight ahead of the one in use
    float4 t4[3];
#pragma unroll
    for (int u = 0; u < 3; u++) t4[u] = v4[min(tid + 256 * u, n4 - 1)];
    for (int base = 0; base < n; base += ST_CHUNK) {
        const float4 cur = t4[0];
        t4[0] = t4[1];
        t4[1] = t4[2];
        if (base + 3 * ST_CHUNK < n) t4[2] = v4[min((base + 3 * ST_CHUNK) / 4 + tid, n4 - 1)];
        // one LDS atomic per wave per chunk: ballots of the 4 slots are prefix-summed in SGPRs
        const float t[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t key[4];
        unsigned long long bal[4];
        int wtot = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + 4 * tid + u;
            key[u] = sel_key<SMALLEST>(t[u]);
            bal[u] = __ballot(i < n && key[u] <= tau);
            wtot += __popcll(bal[u]);
        }
        if (wtot) {
            int b0 = 0;
            if (lane == 0) b0 = atomicAdd(&s_cnt, wtot);
            b0 = __shfl(b0, 0, 64);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if ((bal[u] >> lane) & 1ull)
                    s_buf[b0 + __popcll(bal[u] & ((1ull << lane) - 1ull))] =
                            ((unsigned long long)key[u] << 32) | (unsigned)(base + 4 * tid + u);
                b0 += __popcll(bal[u]);
            }
        }
        __syncthreads();
        const int cnt = s_cnt;
        __syncthreads();
        if (cnt > ST_CAPS - ST_CHUNK && cnt > K) {     // uniform
            tau = st_compact<ST_CAPS>(s_buf, cnt, K, s_hist, s_w, s_red, s_misc);
            if (tid == 0) s_cnt = K;
            __syncthreads();
        }
    }
    int cnt = s_cnt;
    if (cnt > K) {
        (void)st_compact<ST_CAPS>(s_buf, cnt, K, s_hist, s_w, s_red, s_misc);
        cnt = K;
    }
    __syncthreads();
    block_rank_sort<256, ST_CAPS / 512>(s_buf, cnt);   // cnt <= K <= ST_CAPS / 2 distinct items
    for (int i = cnt + tid; i < K; i += 256) s_buf[i] = ~0ull;
    __syncthreads();
    const float sentinel = SMALLEST ? INFINITY : -INFINITY;
    for (int r = tid; r < K; r += 256) {
        const unsigned long long it = s_buf[r];
        float val = sentinel;
        int pos = -1;
        if (it != ~0ull) {
            pos = (int)(uint32_t)it;
            val = v[pos];
            if (val == sentinel) pos = -1;
        }
        out_vals[(int64_t)seg * K + r] = val;
        out_pos[(int64_t)seg * K + r] = pos;
    }
}

// ------------------------------------------------------------------------------------
// Wave-per-row variant for small K (K <= 64: the coarse quantizer's top-nprobe, 1-NN
// assignment).  No workgroup barriers: a 256-thread workgroup is four independent waves, one
// row each.  A row is visited in chunks of 64 lanes x SW_NPL registers:
//   1. every lane takes the minimum of its SW_NPL keys; the K-th smallest of the 64 lane minima
//      is an upper bound tau of the chunk's K-th smallest key (K lanes hold a key <= tau)
//   2. keys <= tau (about 1.4 K of them on unstructured data) are appended to a per-wave LDS
//      list behind the running top-K, lane offsets from a wave prefix sum
//   3. wave rank sort of the list on (key, position), first K kept
// If the bound is loose (structured rows, mass ties) the list would overflow; the chunk's
// top-K is then extracted exactly from the registers, K rounds of wave arg-min.
// Same result as k_select2: the K smallest (key, position) pairs, sorted.
// ------------------------------------------------------------------------------------
namespace {
constexpr int SW_CAP = 512;   // list entries per wave (4 KB)
constexpr int SW_NPL = 64;    // keys per lane and chunk (chunk = 64 * SW_NPL elements)

// buf[0..c) distinct items -> the min(c, K) smallest, sorted, in buf[0..); wave-synchronous
__device__ __forceinline__ void wave_rank_take(unsigned long long* buf, int c, int K) {
    const int lane = threadIdx.x & 63;
    constexpr int MAXI = SW_CAP / 64;
    unsigned long long it[MAXI];
    int rk[MAXI];
    const int nu = (c + 63) >> 6;   // uniform
#pragma unroll
    for (int u = 0; u < MAXI; u++) {
        const int i = lane + 64 * u;
        it[u] = i < c ? buf[i] : ~0ull;
        rk[u] = 0;
    }
    int j = 0;
    for (; j + 8 <= c; j += 8) {
        unsigned long long x[8];
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = buf[j + e];
#pragma unroll
        for (int e = 0; e < 8; e++) {
#pragma unroll
            for (int u = 0; u < MAXI; u++)
                if (u < nu) rk[u] += (x[e] < it[u]) ? 1 : 0;
        }
    }
    for (; j < c; j++) {
        const unsigned long long x = buf[j];
#pragma unroll
        for (int u = 0; u < MAXI; u++)
            if (u < nu) rk[u] += (x < it[u]) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();   // every read above is issued before the scatter below
#pragma unroll
    for (int u = 0; u < MAXI; u++)
        if (lane + 64 * u < c && rk[u] < K) buf[rk[u]] = it[u];
    __builtin_amdgcn_wave_barrier();
}
}  // namespace

// the K smallest (key, position) of v[0..n) -> buf[0..return value), sorted; one wave, buf: SW_CAP items of LDS
template <bool SMALLEST, int NPL>
__global__ __launch_bounds__(256) void k_select_wave(const float* __restrict__ vals, int64_t seg_stride,
                                                     const int* __restrict__ seg_len, int fixed_len,
                                                     int nseg, int K, float* __restrict__ out_vals,
                                                     int* __restrict__ out_pos,
                                                     uint8_t* __restrict__ tie_flag = nullptr) {
    __shared__ unsigned long long s_buf[4][SW_CAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int seg = blockIdx.x * 4 + w;
    if (seg >= nseg) return;   // whole wave; the kernel has no workgroup barrier
    const int n = seg_len ? seg_len[seg] : fixed_len;
    const float* v = vals + (int64_t)seg * seg_stride;
    unsigned long long* buf = s_buf[w];
    int run = 0;               // buf[0..run): running top, sorted
    for (int base = 0; base < n; base += 64 * NPL) {
        uint32_t key[NPL];
        // slot j of this lane is element base + j*64 + lane; it exists iff j*64 < nrem
        const int nrem = n - base - lane;
        if (base + 64 * NPL <= n) {      // full chunk (uniform): one base address, constant offsets
            const float* pv = v + base + lane;
#pragma unroll
            for (int j = 0; j < NPL; j++) key[j] = sel_key<SMALLEST>(pv[j * 64]);
        } else {
            constexpr int LB = NPL < 16 ? NPL : 16;
#pragma unroll
            for (int j0 = 0; j0 < NPL; j0 += LB) {   // LB unconditional (clamped) loads in flight
                float t[LB];
#pragma unroll
                for (int u = 0; u < LB; u++) t[u] = v[min(base + (j0 + u) * 64 + lane, n - 1)];
#pragma unroll
                for (int u = 0; u < LB; u++)
                    key[j0 + u] = (j0 + u) * 64 < nrem ? sel_key<SMALLEST>(t[u]) : 0xffffffffu;
            }
        }
        uint32_t tau;
        if (K <= 32 || NPL < 2) {   // uniform
            uint32_t m = key[0];
#pragma unroll
            for (int j = 1; j < NPL; j++) m = key[j] < m ? key[j] : m;
            // rank of this lane's minimum among the 64 (ties by lane): K-th smallest = tau
            int rk = 0;
#pragma unroll
            for (int l = 0; l < 64; l++) {
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)m, l);
                rk += (o < m || (o == m && l < lane)) ? 1 : 0;
            }
            const unsigned long long who = __ballot(rk == K - 1);
            tau = (uint32_t)__shfl((int)m, (int)__ffsll((long long)who) - 1, 64);
        } else {
            // K in (32, 64]: the K-th of 64 lane minima is no bound (at K = 64 it is the LARGEST of them, a quarter of
            // the chunk passes and the exact extraction below runs for every chunk).  Two minima per lane, over the
            // even and the odd slots: the K-th smallest of the 128 sits where the K/2-th of 64 would -- ~1.4 K items pass.
            uint32_t m0 = key[0], m1 = key[1];
#pragma unroll
            for (int j = 2; j < NPL; j += 2) m0 = key[j] < m0 ? key[j] : m0;
#pragma unroll
            for (int j = 3; j < NPL; j += 2) m1 = key[j] < m1 ? key[j] : m1;
            int rk0 = 0, rk1 = 0;   // ranks among the 128, ties by (group, lane)
#pragma unroll
            for (int l = 0; l < 64; l++) {
                const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)m0, l);
                const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)m1, l);
                rk0 += ((o0 < m0 || (o0 == m0 && l < lane)) ? 1 : 0) + (o1 < m0 ? 1 : 0);
                rk1 += (o0 <= m1 ? 1 : 0) + ((o1 < m1 || (o1 == m1 && l < lane)) ? 1 : 0);
            }
            const unsigned long long who0 = __ballot(rk0 == K - 1), who1 = __ballot(rk1 == K - 1);
            tau = who0 ? (uint32_t)__shfl((int)m0, (int)__ffsll((long long)who0) - 1, 64)
                       : (uint32_t)__shfl((int)m1, (int)__ffsll((long long)who1) - 1, 64);
        }
        if (run == K) {
            const uint32_t kth = (uint32_t)(buf[K - 1] >> 32);
            tau = kth < tau ? kth : tau;
        }
        int c = 0;
#pragma unroll
        for (int j = 0; j < NPL; j++) c += (key[j] <= tau && j * 64 < nrem) ? 1 : 0;
        const int incl = wave_incl_scan(c);
        const int tot = __shfl(incl, 63, 64);
        if (run + tot <= SW_CAP) {
            int off = run + incl - c;
#pragma unroll
            for (int j = 0; j < NPL; j++)
                if (key[j] <= tau && j * 64 < nrem)
                    buf[off++] = ((unsigned long long)key[j] << 32) | (unsigned)(base + j * 64 + lane);
            __builtin_amdgcn_wave_barrier();
            // tie_flag (coarse quantizer, rows of one chunk): one rank more is kept -- is the (K+1)-th key equal
            // to the K-th?  Then WHICH of the tied entries the reference keeps is decided by its heap
            // (k_coarse_heap_fix redoes the row the way the heap does).
            const bool want_flag = tie_flag && base == 0 && n <= 64 * NPL;
            wave_rank_take(buf, run + tot, want_flag ? K + 1 : K);
            if (want_flag) {
                // two equal keys among the K + 1 smallest: which of them is probed (tie at the cut), or in which
                // order their lists are scanned (tie inside), is the heap's doing
                const int have = min(run + tot, K + 1);
                bool eq = false;
                for (int i = lane; i + 1 < have; i += 64)
                    eq |= (uint32_t)(buf[i] >> 32) == (uint32_t)(buf[i + 1] >> 32);
                const unsigned long long any = __ballot(eq);
                if (lane == 0) tie_flag[seg] = any ? 1 : 0;
            }
            run = min(run + tot, K);
        } else {
            if (tie_flag && base == 0 && n <= 64 * NPL && lane == 0) tie_flag[seg] = 1;   // not known here: redo the row
            // exact extraction from the registers: K rounds of (lane arg-min, wave arg-min)
            unsigned long long rm = 0;   // bit j: slot j already taken
            int got = 0;
            for (; got < K; got++) {
                uint32_t bk = 0xffffffffu;
                int bj = -1;
#pragma unroll
                for (int j = 0; j < NPL; j++) {
                    const bool ok = !((rm >> j) & 1ull) && j * 64 < nrem && (bj < 0 || key[j] < bk);
                    bk = ok ? key[j] : bk;
                    bj = ok ? j : bj;
                }
                unsigned long long item = bj < 0 ? ~0ull
                                                 : (((unsigned long long)bk << 32) | (unsigned)(base + bj * 64 + lane));
                unsigned long long best = item;
#pragma unroll
                for (int off2 = 32; off2 > 0; off2 >>= 1) {
                    const unsigned long long o = __shfl_xor(best, off2, 64);
                    best = o < best ? o : best;
                }
                if (best == ~0ull) break;          // chunk exhausted (uniform)
                if (item == best) rm |= 1ull << bj; // positions are distinct: one lane
                if (lane == 0) buf[run + got] = best;
            }
            __builtin_amdgcn_wave_barrier();
            wave_rank_take(buf, run + got, K);
            run = min(run + got, K);
        }
    }
    const float sentinel = SMALLEST ? INFINITY : -INFINITY;
    for (int r = lane; r < K; r += 64) {
        float val = sentinel;
        int pos = -1;
        if (r < run) {
            pos = (int)(uint32_t)buf[r];
            val = v[pos];
            if (val == sentinel) pos = -1;
        }
        out_vals[(int64_t)seg * K + r] = val;
        out_pos[(int64_t)seg * K + r] = pos;
    }
}

// ------------------------------------------------------------------------------------
// Threshold pre-filter of the list scan (kernels.hip, k_ivfpq_scan_pair<.., FILT>).
//
// The bound itself is computed inside the scan kernel by the workgroup of each query's first probe
// group (kernels.hip); ready[q] = 1 when a bound exists, 2 when the group held fewer than K valid
// candidates (that query then takes the unfiltered selection kernel).
//
// k_select_final: one wave per query turns the survivor list written by the filtered scan
// (a few hundred (key, position) items) into the sorted top-K: histogram cut down to <= 256
// items, four 64-item bitonic sorts in registers (wave shuffles), rank merge by binary search.
// Lists that overflowed, or cuts that cannot get below 256 items (mass ties), set flag = 1.
// ------------------------------------------------------------------------------------
namespace {
constexpr int SF_CAND = 384;   // candidates of one query held in LDS (first group within the bound + slices);
                               // sized so that 8 workgroups (= all 2048 of an 8192-query batch) are resident per CU
}

template <bool SMALLEST, int PMAX, int NR = 4>   // PMAX: 64 or 128 probes per query; NR: sorted runs of 64 (K <= 64 NR: 4, or 8 for
                                                  // recall_num 257 ... 512, round 6 -- the workgroup-per-query kernel took 0.9 ms there against 0.1)
__device__ __forceinline__ void select_final_body(const unsigned long long* __restrict__ surv,
                                                      const int* __restrict__ gcnt, int nslices,
                                                      int slice_cap,
                                                      const unsigned long long* __restrict__ ready,
                                                      const int* __restrict__ pair_off, int P, int nq, int K,
                                                      const int64_t* __restrict__ pair_base,
                                                      const int64_t* __restrict__ ids,
                                                      uint8_t* __restrict__ flag,
                                                      float* __restrict__ out_vals,
                                                      int* __restrict__ out_pos,
                                                      int64_t* __restrict__ out_ids,
                                                      uint8_t* __restrict__ cut_tie,
                                                      unsigned long long* __restrict__ tie_stats,
                                                      int* __restrict__ rq_list, int* __restrict__ rq_count,
                                                      unsigned long long* __restrict__ bound_stat) {
    __shared__ int s_hist[4][256];
    constexpr int KEEP = 64 * NR, SFC = NR == 4 ? SF_CAND : 96 * NR;
    __shared__ unsigned long long s_cand[4][SFC];   // candidates, later the <= KEEP kept ones (in place)
    __shared__ int s_off[4][PMAX + 8];
    __shared__ int64_t s_base[4][PMAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (bound_stat && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(bound_stat + 1, (unsigned long long)nq);
    if (q >= nq) return;
    const unsigned long long word = ready[q];
    // slice counts of the consumer workgroups (nslices <= 64: one per lane)
    const int my_cnt = lane < nslices ? gcnt[(int64_t)q * nslices + lane] : 0;
    // left to the unfiltered selection kernel: no bound (every group stored its distances), or a slice overflowed
    // (groups with a bound did not store: the repair launch scores them again first, rq_list)
    if ((word >> 32) != 1ull) {
        if (lane == 0) {
            flag[q] = 1;
            if (bound_stat) atomicAdd(bound_stat, 1ull);
        }
        return;
    }
    auto slice_ptr = [&](int g) -> const unsigned long long* { return surv + ((int64_t)q * nslices + g) * slice_cap; };
    if (__ballot(my_cnt > slice_cap)) {
        if (lane == 0) {
            flag[q] = 1;
            if (rq_list) rq_list[atomicAdd(rq_count, 1)] = q;
            if (bound_stat) atomicAdd(bound_stat, 1ull);
        }
        return;
    }
    int* hist = s_hist[w];
    unsigned long long* cand = s_cand[w];
    unsigned long long* runs = cand;
    const int* goff = pair_off + (int64_t)q * (P + 1);
    int* off = s_off[w];        // this query's pair offsets (P + 1 <= PMAX + 1 entries, see the launcher)
    for (int i = lane; i <= P; i += 64) off[i] = goff[i];
    int64_t* lbase = s_base[w];   // arena offset of each probed list (k_pair_offsets): no probe_list ->
    for (int i = lane; i < P; i += 64) lbase[i] = pair_base[(int64_t)q * P + i];   // list_off chain at the end
    // ---- gather the candidate set into LDS: the survivor slices (slice 0 = the first probe group's own
    //      candidates within the bound, appended by its producer workgroup) ----
    int c = 0;
    uint32_t mn = 0xffffffffu, mx = 0u;
    if (nslices == 2) {
        // two slices (the filter-pass launches: the producer's own candidates + ONE consumer's): the first four 64-item blocks of
        // both requested together -- a consumer's slice holds ~230 items, and each further block used to be a load of its own
        // behind the previous one's use
        const int c0 = __shfl(my_cnt, 0, 64), c1 = __shfl(my_cnt, 1, 64);
        const unsigned long long* s0 = slice_ptr(0);
        const unsigned long long* s1 = slice_ptr(1);
        unsigned long long t0[4], t1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            t0[u] = s0[min(64 * u + lane, max(c0 - 1, 0))];
            t1[u] = s1[min(64 * u + lane, max(c1 - 1, 0))];
        }
        auto take = [&](const unsigned long long* sg, const unsigned long long (&t)[4], int cg) {
            for (int i0 = 0; i0 < cg; i0 += 64) {
                if (i0 + lane < cg) {
                    const unsigned long long item = i0 < 256 ? t[(i0 >> 6) & 3] : sg[i0 + lane];
                    const uint32_t key = (uint32_t)(item >> 32);
                    if (c + i0 + lane < SFC) cand[c + i0 + lane] = item;
                    mn = key < mn ? key : mn;
                    mx = key > mx ? key : mx;
                }
            }
            c += cg;
        };
        take(s0, t0, c0);
        take(s1, t1, c1);
    } else
    for (int g0 = 0; g0 < nslices; g0 += 8) {   // the first 64 items of 8 slices in flight at once
        int cg[8];
        unsigned long long t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            cg[u] = g0 + u < nslices ? __shfl(my_cnt, min(g0 + u, 63), 64) : 0;
            const unsigned long long* sg = slice_ptr(min(g0 + u, nslices - 1));
            t[u] = sg[min(lane, max(cg[u] - 1, 0))];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned long long* sg = slice_ptr(min(g0 + u, nslices - 1));
            for (int i0 = 0; i0 < cg[u]; i0 += 64) {
                if (i0 + lane < cg[u]) {
                    const unsigned long long item = i0 == 0 ? t[u] : sg[i0 + lane];
                    const uint32_t key = (uint32_t)(item >> 32);
                    if (c + i0 + lane < SFC) cand[c + i0 + lane] = item;
                    mn = key < mn ? key : mn;
                    mx = key > mx ? key : mx;
                }
            }
            c += cg[u];
        }
    }
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    __builtin_amdgcn_wave_barrier();
    // later passes visit the candidates in LDS, or -- when they did not fit (loose bound, rare) --
    // stream them again from memory; body(valid, key, item) is called with a uniform trip count
    auto for_each = [&](auto&& body) {
        if (c <= SFC) {
            for (int i0 = 0; i0 < c; i0 += 64) {
                const unsigned long long item = i0 + lane < c ? cand[i0 + lane] : ~0ull;
                body(i0 + lane < c, (uint32_t)(item >> 32), item);
            }
            return;
        }
        for (int g = 0; g < nslices; g++) {
            const int cg = __shfl(my_cnt, g, 64);
            const unsigned long long* sg = slice_ptr(g);
            for (int i0 = 0; i0 < cg; i0 += 64) {
                const unsigned long long item = sg[min(i0 + lane, cg - 1)];
                body(i0 + lane < cg, (uint32_t)(item >> 32), item);
            }
        }
    };
    // ---- cut to <= 256 items: keep keys <= cutoff, cutoff from (nested) 256-bin histograms ----
    uint32_t cutoff = 0xffffffffu;
    if (c > KEEP) {
        uint32_t lo = mn;
        const uint32_t range = mx - mn;
        int s = range >= 256u ? (32 - __clz((int)range)) - 8 : 0;
        int below = 0;                   // items with key < lo (all kept)
        for (;;) {
#pragma unroll
            for (int u = 0; u < 4; u++) hist[lane * 4 + u] = 0;
            __builtin_amdgcn_wave_barrier();
            for_each([&](bool ok, uint32_t key, unsigned long long) {
                if (ok && key >= lo && ((key - lo) >> s) < 256u) atomicAdd(&hist[(key - lo) >> s], 1);
            });
            __builtin_amdgcn_wave_barrier();
            int cc[4], c4 = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                cc[u] = hist[lane * 4 + u];
                c4 += cc[u];
            }
            const int incl = wave_incl_scan(c4);
            int run = below + incl - c4, b = -1, kept = 0, before = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (b < 0 && run < K && K <= run + cc[u]) {
                    b = lane * 4 + u;
                    before = run;
                    kept = run + cc[u];
                }
                run += cc[u];
            }
            const unsigned long long who = __ballot(b >= 0);
            const int src_lane = (int)__ffsll((long long)who) - 1;
            b = __shfl(b, src_lane, 64);
            kept = __shfl(kept, src_lane, 64);
            before = __shfl(before, src_lane, 64);
            __builtin_amdgcn_wave_barrier();
            if (kept <= KEEP || s == 0) {
                if (kept > KEEP) {        // > 256 copies of one key around the K-th: rare, unfiltered path
                    if (lane == 0) {
                        flag[q] = 1;
                        if (rq_list) rq_list[atomicAdd(rq_count, 1)] = q;
                        if (bound_stat) atomicAdd(bound_stat, 1ull);
                    }
                    return;
                }
                const unsigned long long edge = (unsigned long long)lo + (((unsigned long long)b + 1ull) << s) - 1ull;
                cutoff = edge > 0xffffffffull ? 0xffffffffu : (uint32_t)edge;
                break;
            }
            below = before;
            lo += (uint32_t)b << s;
            s = s > 8 ? s - 8 : 0;
        }
    }
    if (lane == 0) flag[q] = 0;
    // ---- compact the kept items into runs[0..m), m <= 256 ----
    // (in place when the candidates sit in LDS: each lane holds its item before the wave writes,
    //  and the write index never passes the read index)
    int m = 0;
    for_each([&](bool ok, uint32_t key, unsigned long long item) {
        const bool keep = ok && key <= cutoff;
        const unsigned long long bal = __ballot(keep);
        __builtin_amdgcn_wave_barrier();
        if (keep) runs[m + __popcll(bal & ((1ull << lane) - 1ull))] = item;
        m += __popcll(bal);
    });
    __builtin_amdgcn_wave_barrier();
    // ---- four sorted runs of 64, then rank merge ----
    unsigned long long x[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) x[r] = (r * 64 + lane < m) ? runs[r * 64 + lane] : ~0ull;
    __builtin_amdgcn_wave_barrier();
    wave_sort64_multi<NR>(x);
#pragma unroll
    for (int r = 0; r < NR; r++) runs[r * 64 + lane] = x[r];
    __builtin_amdgcn_wave_barrier();
    const float sentinel = SMALLEST ? INFINITY : -INFINITY;
    int rk[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int rank = lane;
#pragma unroll
        for (int o = 0; o < NR; o++) {
            if (o == r) continue;
            // number of items of run o smaller than x[r]
            const unsigned long long* ro = runs + o * 64;
            int lo2 = 0, n2 = 64;
#pragma unroll
            for (int st = 0; st < 7; st++) {
                if (n2 > 0) {
                    const int half = n2 >> 1;
                    if (ro[lo2 + half] < x[r]) {
                        lo2 += half + 1;
                        n2 -= half + 1;
                    } else {
                        n2 = half;
                    }
                }
            }
            rank += lo2;
        }
        rk[r] = rank;
    }
    // the K winners in rank order (LDS), then one coalesced pass: rank r = lane + 64 i
    __builtin_amdgcn_wave_barrier();   // all binary searches have read the runs
#pragma unroll
    for (int r = 0; r < NR; r++)
        if (x[r] != ~0ull && rk[r] < KEEP) runs[rk[r]] = x[r];   // ranks are a permutation of 0..m-1
    __builtin_amdgcn_wave_barrier();
    // exact ties: every candidate at the K-th key is among the m kept items (key <= cutoff), so the cut went
    // through a tie group iff the item of rank K carries the key of rank K - 1
    if (cut_tie && lane == 0) {
        const bool tie = m > K && (uint32_t)(runs[K] >> 32) == (uint32_t)(runs[K - 1] >> 32);
        cut_tie[q] = tie ? 1 : 0;
        if (tie && tie_stats) atomicAdd(tie_stats + 1, 1ull);
    }
    // position in the query's segment -> vector id (as k_map_candidates): last p with off[p] <= ps.
    // Branch-free on clamped values so the four dependent load chains run side by side.
    const int nres = min(m, K);
    unsigned long long it[NR];
    int ps[NR], pp[NR];
    int64_t idv[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        it[r] = runs[min(lane + 64 * r, max(nres - 1, 0))];
        ps[r] = (int)(uint32_t)it[r];
        int lo = 0;
#pragma unroll
        for (int step = PMAX / 2; step >= 1; step >>= 1) {   // P <= PMAX
            const int mid = lo + step;
            if (mid < P && off[min(mid, P)] <= ps[r]) lo = mid;
        }
        pp[r] = lo;
    }
    // (no survivor at all -- a shard that holds none of the query's probed lists: nothing to look up)
#pragma unroll
    for (int r = 0; r < NR; r++) idv[r] = nres > 0 ? ids[lbase[pp[r]] + (ps[r] - off[pp[r]])] : -1;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int rank = lane + 64 * r;
        if (rank < nres) {
            const uint32_t key = (uint32_t)(it[r] >> 32);
            out_vals[(int64_t)q * K + rank] = key2f(SMALLEST ? key : ~key);
            out_pos[(int64_t)q * K + rank] = ps[r];
            out_ids[(int64_t)q * K + rank] = idv[r] & 0x7fffffffffffffffLL;
        }
    }
    for (int r = m + lane; r < K; r += 64) {   // fewer than K candidates in all (the widest bound, scan.hip)
        out_vals[(int64_t)q * K + r] = sentinel;
        out_pos[(int64_t)q * K + r] = -1;
        out_ids[(int64_t)q * K + r] = -1;
    }
}

// the kernels: four runs (recall_num <= 256) under a 64-register budget -- eight waves per SIMD: the kernel is a chain of dependent
// loads per wave (bound word + counts -> slices -> ids), what it gains from is waves in flight (93.8 against 96.5 us at C3) -- and
// eight runs (recall_num <= 512: 99 registers) without one
#define GH_SF_PARAMS                                                                                                          \
    const unsigned long long *__restrict__ surv, const int *__restrict__ gcnt, int nslices, int slice_cap,                    \
        const unsigned long long *__restrict__ ready, const int *__restrict__ pair_off, int P, int nq, int K,                 \
        const int64_t *__restrict__ pair_base, const int64_t *__restrict__ ids, uint8_t *__restrict__ flag,                   \
        float *__restrict__ out_vals, int *__restrict__ out_pos, int64_t *__restrict__ out_ids, uint8_t *__restrict__ cut_tie, \
        unsigned long long *__restrict__ tie_stats, int *__restrict__ rq_list, int *__restrict__ rq_count,                    \
        unsigned long long *__restrict__ bound_stat
#define GH_SF_ARGS                                                                                                                   \
    surv, gcnt, nslices, slice_cap, ready, pair_off, P, nq, K, pair_base, ids, flag, out_vals, out_pos, out_ids, cut_tie, tie_stats, \
        rq_list, rq_count, bound_stat
// (PMAX = 128: the pair tables in LDS leave seven waves per SIMD whatever the registers -- the budget still holds, the compiler says so)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <bool SMALLEST, int PMAX, int NR = 4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_select_final(GH_SF_PARAMS) {
    static_assert(NR == 4, "the 64-register variant");
    select_final_body<SMALLEST, PMAX, 4>(GH_SF_ARGS);
}
#pragma clang diagnostic pop
template <bool SMALLEST, int PMAX>
__global__ __launch_bounds__(256) void k_select_final8(GH_SF_PARAMS) {
    select_final_body<SMALLEST, PMAX, 8>(GH_SF_ARGS);
}
#undef GH_SF_PARAMS
#undef GH_SF_ARGS

// ------------------------------------------------------------------------------------
// k_select_final_wg: the same selection for 256 < K <= 1024 (round 5: the bounded scan's gate was recall_num <= 256, and the
// configurations that need a long short-list -- full-size C5 reaches recall@10 0.95 at recall_num ~1000 -- ran on the
// unfiltered path).  One WORKGROUP per query: the survivor slices (slice_cap up to 2048 items each, a few thousand items in
// all) are streamed from memory for nested 256-bin histograms of the keys until the cut keeps <= SFW_KEEP items, those are
// compacted into LDS, rank-sorted on the whole (key, position) item, and the first K go out.  Same results, flags, repair
// list and counters as k_select_final.
// ------------------------------------------------------------------------------------
namespace {
constexpr int SFW_KEEP = 2048;
}
template <bool SMALLEST>
__global__ __launch_bounds__(256) void k_select_final_wg(const unsigned long long* __restrict__ surv, const int* __restrict__ gcnt,
                                                         int nslices, int slice_cap, const unsigned long long* __restrict__ ready,
                                                         const int* __restrict__ pair_off, int P, int nq, int K,
                                                         const int64_t* __restrict__ pair_base, const int64_t* __restrict__ ids,
                                                         uint8_t* __restrict__ flag, float* __restrict__ out_vals,
                                                         int* __restrict__ out_pos, int64_t* __restrict__ out_ids,
                                                         uint8_t* __restrict__ cut_tie, unsigned long long* __restrict__ tie_stats,
                                                         int* __restrict__ rq_list, int* __restrict__ rq_count,
                                                         unsigned long long* __restrict__ bound_stat) {
    __shared__ unsigned long long s_it[SFW_KEEP];
    __shared__ int s_hist[256];
    __shared__ int s_off[128 + 8];
    __shared__ int64_t s_base[128];
    __shared__ int s_w[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = blockIdx.x;
    if (bound_stat && q == 0 && tid == 0) atomicAdd(bound_stat + 1, (unsigned long long)nq);
    const unsigned long long word = ready[q];
    if ((word >> 32) != 1ull) {   // no bound: every group stored its distances, the unfiltered selection takes the query
        if (tid == 0) {
            flag[q] = 1;
            if (bound_stat) atomicAdd(bound_stat, 1ull);
        }
        return;
    }
    auto give_up = [&]() {   // a slice overflowed / a mass tie: the repair launch scores the consumer groups again, with stores
        if (tid == 0) {
            flag[q] = 1;
            if (rq_list) rq_list[atomicAdd(rq_count, 1)] = q;
            if (bound_stat) atomicAdd(bound_stat, 1ull);
        }
    };
    const int* cnt = gcnt + (int64_t)q * nslices;
    int c = 0;
    bool over = false;
    for (int g = 0; g < nslices; g++) {   // uniform
        const int n = cnt[g];
        over |= n > slice_cap;
        c += n;
    }
    if (over) {
        give_up();
        return;
    }
    const int* goff = pair_off + (int64_t)q * (P + 1);
    for (int i = tid; i <= P; i += 256) s_off[i] = goff[i];
    for (int i = tid; i < P; i += 256) s_base[i] = pair_base[(int64_t)q * P + i];
    // body(valid, item) for every survivor, uniform trip counts per slice
    auto for_each = [&](auto&& body) {
        for (int g = 0; g < nslices; g++) {
            const int n = cnt[g];
            const unsigned long long* sg = surv + ((int64_t)q * nslices + g) * slice_cap;
            for (int i0 = 0; i0 < n; i0 += 256) {
                const int i = i0 + tid;
                body(i < n, sg[min(i, n - 1)]);
            }
        }
    };
    // ---- the range of the keys ----
    uint32_t mn = 0xffffffffu, mx = 0u;
    for_each([&](bool ok, unsigned long long item) {
        const uint32_t key = (uint32_t)(item >> 32);
        if (ok) {
            mn = key < mn ? key : mn;
            mx = key > mx ? key : mx;
        }
    });
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    if (lane == 0) {
        s_w[wv] = (int)mn;
        s_w[4 + wv] = (int)mx;
    }
    __syncthreads();
    mn = min(min((uint32_t)s_w[0], (uint32_t)s_w[1]), min((uint32_t)s_w[2], (uint32_t)s_w[3]));
    mx = max(max((uint32_t)s_w[4], (uint32_t)s_w[5]), max((uint32_t)s_w[6], (uint32_t)s_w[7]));
    __syncthreads();
    // ---- cut to <= SFW_KEEP items: keep keys <= cutoff, cutoff from (nested) 256-bin histograms ----
    uint32_t cutoff = 0xffffffffu;
    if (c > SFW_KEEP) {
        uint32_t lo = mn;
        const uint32_t range = mx - mn;
        int sh = range >= 256u ? (32 - __clz((int)range)) - 8 : 0;
        int below = 0;   // items with key < lo (all kept)
        for (;;) {
            s_hist[tid] = 0;
            __syncthreads();
            for_each([&](bool ok, unsigned long long item) {
                const uint32_t key = (uint32_t)(item >> 32);
                if (ok && key >= lo && ((key - lo) >> sh) < 256u) atomicAdd(&s_hist[(key - lo) >> sh], 1);
            });
            __syncthreads();
            if (tid < 64) {   // wave 0: the bin in which the running count passes K
                int cc[4], c4 = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    cc[u] = s_hist[lane * 4 + u];
                    c4 += cc[u];
                }
                const int incl = wave_incl_scan(c4);
                int run = below + incl - c4;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (run < K && K <= run + cc[u]) {
                        s_w[0] = lane * 4 + u;
                        s_w[1] = run;
                        s_w[2] = run + cc[u];
                    }
                    run += cc[u];
                }
            }
            __syncthreads();
            const int b = s_w[0], before = s_w[1], kept = s_w[2];
            __syncthreads();
            if (kept <= SFW_KEEP || sh == 0) {
                if (kept > SFW_KEEP) {   // more than SFW_KEEP copies of one key around the K-th
                    give_up();
                    return;
                }
                const unsigned long long edge = (unsigned long long)lo + (((unsigned long long)b + 1ull) << sh) - 1ull;
                cutoff = edge > 0xffffffffull ? 0xffffffffu : (uint32_t)edge;
                break;
            }
            below = before;
            lo += (uint32_t)b << sh;
            sh = sh > 8 ? sh - 8 : 0;
        }
    }
    if (tid == 0) {
        flag[q] = 0;
        s_w[3] = 0;
    }
    __syncthreads();
    // ---- the kept items into LDS (any order: sorted next) ----
    for_each([&](bool ok, unsigned long long item) {
        const bool keep = ok && (uint32_t)(item >> 32) <= cutoff;
        const unsigned long long bal = __ballot(keep);
        if (bal) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_w[3], __popcll(bal));
            base = __shfl(base, 0, 64);
            if (keep) s_it[base + __popcll(bal & ((1ull << lane) - 1ull))] = item;
        }
    });
    __syncthreads();
    const int m = s_w[3];
    block_rank_sort<256, SFW_KEEP / 256>(s_it, m);   // (key, position) items are distinct
    if (cut_tie && tid == 0) {   // every candidate at the K-th key is among the kept items
        const bool tie = m > K && (uint32_t)(s_it[K] >> 32) == (uint32_t)(s_it[K - 1] >> 32);
        cut_tie[q] = tie ? 1 : 0;
        if (tie && tie_stats) atomicAdd(tie_stats + 1, 1ull);
    }
    const float sentinel = SMALLEST ? INFINITY : -INFINITY;
    const int nres = min(m, K);
    for (int r = tid; r < K; r += 256) {
        if (r < nres) {
            const unsigned long long it = s_it[r];
            const int ps = (int)(uint32_t)it;
            int lo2 = 0, hi2 = P - 1;
            while (lo2 < hi2) {   // last p with off[p] <= ps
                const int mid = (lo2 + hi2 + 1) >> 1;
                if (s_off[mid] <= ps) lo2 = mid; else hi2 = mid - 1;
            }
            const uint32_t key = (uint32_t)(it >> 32);
            out_vals[(int64_t)q * K + r] = key2f(SMALLEST ? key : ~key);
            out_pos[(int64_t)q * K + r] = ps;
            out_ids[(int64_t)q * K + r] = ids[s_base[lo2] + (ps - s_off[lo2])] & 0x7fffffffffffffffLL;
        } else {
            out_vals[(int64_t)q * K + r] = sentinel;
            out_pos[(int64_t)q * K + r] = -1;
            out_ids[(int64_t)q * K + r] = -1;
        }
    }
}

// ------------------------------------------------------------------------------------
// Merge of the per-shard candidate tables (list-sharded search, gamma_hip_ivfpq_merge_rerank):
// all_dis / all_ids [W][nq][R] -> the K = R best of each query's W*R candidates, ordered by
// (distance, shard, rank inside the shard) -- what selecting from the gathered [nq][W*R] row gives.
// One wave per query, the row's keys live in registers (NPL per lane):
//   1. K-th smallest key by bisection on the key value (count = compare + popcount, no memory)
//   2. keys below it, and the first few equal to it in index order, are compacted into LDS (<= 256)
//   3. four 64-item bitonic sorts in registers + rank merge (as k_select_final)
// Nothing is assumed about the order inside a shard's row.  Entries with id < 0 (padding: a shard
// with fewer than R candidates) or a sentinel distance are invalid and come out as (sentinel, -1).
// ------------------------------------------------------------------------------------
template <bool SMALLEST, int NPL>
__global__ __launch_bounds__(256) void k_merge_shards(const float* __restrict__ all_dis,
                                                      const int64_t* __restrict__ all_ids, int W, int nq,
                                                      int R, int q0, int nql, float* __restrict__ out_dis,
                                                      int64_t* __restrict__ out_ids) {
    __shared__ unsigned long long s_run[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nql) return;
    constexpr uint32_t KEYSENT = 0xff800000u;   // key of +inf (SMALLEST) / -inf (largest)
    const int n = W * R, K = R;
    const int64_t row = (int64_t)(q0 + q) * R, blk = (int64_t)nq * R;
    // element e = j * 64 + lane sits at shard e / R, rank e % R: walked incrementally
    const int q64 = 64 / R, m64 = 64 % R;
    uint32_t key[NPL];
    {
        float dv[NPL];
        int32_t iv[NPL];   // sign word of the id (little endian): all that is needed here
        const int32_t* idw = reinterpret_cast<const int32_t*>(all_ids);
        int sh = lane / R, r = lane % R;
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const bool in = j * 64 + lane < n;
            const int64_t at = in ? (int64_t)sh * blk + row + r : row;
            dv[j] = all_dis[at];
            iv[j] = idw[2 * at + 1];
            sh += q64;
            r += m64;
            if (r >= R) {
                r -= R;
                sh++;
            }
        }
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const bool in = j * 64 + lane < n;
            uint32_t kk = sel_key<SMALLEST>(dv[j]);
            if (iv[j] < 0 || kk > KEYSENT) kk = KEYSENT;
            key[j] = in ? kk : 0xffffffffu;
        }
    }
    const int Kq = min(K, n);
    // smallest v with #(key <= v) >= Kq
    uint32_t lo = 0u, hi = KEYSENT;   // every real key is <= KEYSENT and there are n >= Kq of them
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int j = 0; j < NPL; j++) c += key[j] <= mid ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if (c >= Kq) hi = mid;
        else lo = mid + 1u;
    }
    const uint32_t vstar = lo;
    int below = 0;
#pragma unroll
    for (int j = 0; j < NPL; j++) below += key[j] < vstar ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off, 64);
    const int need_eq = Kq - below;   // >= 1
    unsigned long long* runs = s_run[w];
    int m = 0, eq_seen = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const bool is_eq = key[j] == vstar;
        const unsigned long long eqb = __ballot(is_eq);
        const bool take = key[j] < vstar || (is_eq && eq_seen + __popcll(eqb & lt_mask) < need_eq);
        const unsigned long long tb = __ballot(take);
        if (take) runs[m + __popcll(tb & lt_mask)] = ((unsigned long long)key[j] << 32) | (unsigned)(j * 64 + lane);
        m += __popcll(tb);
        eq_seen += __popcll(eqb);
    }
    __builtin_amdgcn_wave_barrier();
    // m == Kq <= 256 items: four sorted runs of 64, then rank merge
    unsigned long long x[4];
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = (r * 64 + lane < m) ? runs[r * 64 + lane] : ~0ull;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = wave_sort64(x[r]);
#pragma unroll
    for (int r = 0; r < 4; r++) runs[r * 64 + lane] = x[r];
    __builtin_amdgcn_wave_barrier();
    int rk[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int rank = lane;
#pragma unroll
        for (int o = 0; o < 4; o++) {
            if (o == r) continue;
            const unsigned long long* ro = runs + o * 64;
            int lo2 = 0, n2 = 64;
#pragma unroll
            for (int st = 0; st < 7; st++) {
                if (n2 > 0) {
                    const int half = n2 >> 1;
                    if (ro[lo2 + half] < x[r]) {
                        lo2 += half + 1;
                        n2 -= half + 1;
                    } else {
                        n2 = half;
                    }
                }
            }
            rank += lo2;
        }
        rk[r] = rank;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (x[r] != ~0ull && rk[r] < 256) runs[rk[r]] = x[r];
    __builtin_amdgcn_wave_barrier();
    const float sentinel = SMALLEST ? INFINITY : -INFINITY;
    unsigned long long it[4];
    int64_t idv[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        it[r] = runs[min(lane + 64 * r, max(m - 1, 0))];
        const int e = (int)(uint32_t)it[r];
        const int sh = e / R;
        idv[r] = all_ids[(int64_t)sh * blk + row + (e - sh * R)];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int rank = lane + 64 * r;
        if (rank < m) {
            const uint32_t kk = (uint32_t)(it[r] >> 32);
            const bool valid = kk < KEYSENT && idv[r] >= 0;
            out_dis[(int64_t)q * K + rank] = valid ? key2f(SMALLEST ? kk : ~kk) : sentinel;
            out_ids[(int64_t)q * K + rank] = valid ? idv[r] : -1;
        }
    }
    for (int r = m + lane; r < K; r += 64) {
        out_dis[(int64_t)q * K + r] = sentinel;
        out_ids[(int64_t)q * K + r] = -1;
    }
}

bool launch_merge_shards(hipStream_t s, bool smallest, const float* all_dis, const int64_t* all_ids, int W,
                         int nq, int R, int q0, int nql, float* out_dis, int64_t* out_ids) {
    const int64_t n = (int64_t)W * R;
    if (R > 256 || n > 64 * 32 || nql <= 0) return false;
    const dim3 grid((nql + 3) / 4), block(256);
#define GH_MERGE(SM, NPL)                                                                                \
    hipLaunchKernelGGL((k_merge_shards<SM, NPL>), grid, block, 0, s, all_dis, all_ids, W, nq, R, q0, nql, \
                       out_dis, out_ids)
#define GH_MERGE_N(SM)                         \
    do {                                       \
        if (n <= 64 * 8) GH_MERGE(SM, 8);      \
        else if (n <= 64 * 16) GH_MERGE(SM, 16); \
        else GH_MERGE(SM, 32);                 \
    } while (0)
    if (smallest) GH_MERGE_N(true);
    else GH_MERGE_N(false);
#undef GH_MERGE_N
#undef GH_MERGE
    return true;
}

// ------------------------------------------------------------------------------------
// Flat search with a running bound (gamma_hip_search.cpp, flat_search_device_locked; the emitting
// distance kernel is k_pairwise_lds<.., EMIT> in kernels.hip).  A query's candidate list holds
// (key << 32 | row id) items: ascending 64-bit order = (distance, row id) order, the order the
// reference's scan + heap leaves behind up to the membership inside exact ties.
// ------------------------------------------------------------------------------------
constexpr uint32_t FLAT_ANY = 0xff7fffffu;   // bound that admits every non-sentinel key
constexpr int FLAT_CAP = 2048;               // items per query list (k_flat_compact holds them in registers)

__global__ __launch_bounds__(256) void k_flat_init(const float* __restrict__ vals, const int* __restrict__ pos,
                                                   int k, int64_t r0, bool smallest, FlatEmit em,
                                                   uint32_t* __restrict__ tau, int* __restrict__ kept) {
    // first chunk's top-k (sorted on (value, position), invalid entries behind with pos = -1)
    const int q = blockIdx.x;
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < k; i += 256) {
        const int ps = pos[(int64_t)q * k + i];
        if (ps >= 0) {
            const uint32_t kk = f2key(vals[(int64_t)q * k + i]);
            em.cand[(int64_t)q * em.cap + i] = ((unsigned long long)(smallest ? kk : ~kk) << 32) | (unsigned)(r0 + ps);
            mine++;
        }
    }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n = s_n;
        em.cnt[(int64_t)q * em.cstride] = n;
        if (kept) kept[q] = n;
        tau[q] = n == k ? (uint32_t)(em.cand[(int64_t)q * em.cap + k - 1] >> 32) : FLAT_ANY;
    }
}

// one wave per query: keep the k smallest items of the list, sorted; new bound = key of the k-th
__global__ __launch_bounds__(256) void k_flat_compact(int nq, int k, FlatEmit em, uint32_t* __restrict__ tau,
                                                      int* __restrict__ overflow, FlatLog lg) {
    constexpr int NPL = FLAT_CAP / 64;
    __shared__ unsigned long long s_run[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;
#define GH_CT(i) do { if (lg.dbg && q == 0 && lane == 0) lg.dbg[i] = wall_clock64(); } while (0)
    GH_CT(0);
    const int cnt = em.cnt[(int64_t)q * em.cstride];
    if (cnt > em.cap) {   // more survivors than the list holds: the caller redoes the call without a bound
        if (lane == 0) *overflow = 1;
        return;
    }
    if (lg.items && lane == 0 && cnt == 0) lg.cnt[(int64_t)q * lg.nsl + lg.pass] = 0;
    if (cnt == 0) return;
    unsigned long long* list = em.cand + (int64_t)q * em.cap;
    // (a list rarely holds more than a few hundred of its 2048 slots: every loop below stops at the last slot in use)
    const int nj = (cnt + 63) >> 6;   // uniform
    unsigned long long it[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) it[j] = ~0ull;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        if (j >= nj) continue;
        if (j * 64 + lane < cnt) it[j] = list[j * 64 + lane];
    }
    const int m = min(cnt, k);
    GH_CT(1);
    if (lg.items) {
        // exact ties: what this pass appended (entries behind the `kept` older ones) is the pass's part of the stream
        // the reference's heap saw -- a superset of what it took: the bound was its root or looser
        const int old = min(lg.kept[q], cnt);
        unsigned long long* dst = lg.items + ((int64_t)q * lg.nsl + lg.pass) * em.cap;
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            if (j >= nj) continue;
            const int idx = j * 64 + lane;
            if (idx >= old && idx < cnt) dst[idx - old] = it[j];
        }
        if (lane == 0) {
            lg.cnt[(int64_t)q * lg.nsl + lg.pass] = cnt - old;
            lg.kept[q] = m;
        }
    }
    // V with exactly m items <= V (items are distinct).  Bisection on the 32-bit KEYS between the smallest and the largest
    // one present (about twenty steps of 32-bit compares; the 64-step bisection on whole items was most of this kernel);
    // only when the m-th key is shared does a second bisection over the positions of the items that carry it decide.
    auto wave_sum = [&](int c) { return __reduce_add_sync(~0ull, c); };
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    GH_CT(2);
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        if (j >= nj) continue;
        const uint32_t kj = (uint32_t)(it[j] >> 32);   // (an empty slot carries 0xffffffff: never <= a bound below kmax)
        if (it[j] != ~0ull) {
            kmin = min(kmin, kj);
            kmax = max(kmax, kj);
        }
    }
    uint32_t klo = wave_min_u32(kmin), khi = wave_max_u32(kmax);
    // one step = a ballot and a scalar popcount per 64 slots in use: no cross-lane reduction, the count is wave-uniform
    // (mid < khi <= kmax < the empty slots' key, so those never count)
    auto bisect = [&](auto nj_c) {
        constexpr int NJ = decltype(nj_c)::value;
        while (klo < khi) {
            const uint32_t mid = klo + ((khi - klo) >> 1);
            int c = 0;
#pragma unroll
            for (int j = 0; j < NJ; j++) c += __popcll(__ballot((uint32_t)(it[j] >> 32) <= mid));
            if (c >= m) khi = mid;
            else klo = mid + 1u;
        }
    };
    if (nj <= 4) bisect(std::integral_constant<int, 4>{});
    else if (nj <= 8) bisect(std::integral_constant<int, 8>{});
    else bisect(std::integral_constant<int, NPL>{});
    const uint32_t K = klo;
    GH_CT(3);
    int c_le = 0, c_lt = 0;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        if (j >= nj) continue;
        const uint32_t kj = (uint32_t)(it[j] >> 32);
        c_le += (it[j] != ~0ull && kj <= K) ? 1 : 0;
        c_lt += (it[j] != ~0ull && kj < K) ? 1 : 0;
    }
    c_le = wave_sum(c_le);
    c_lt = wave_sum(c_lt);
    unsigned long long V = ((unsigned long long)K << 32) | 0xffffffffull;
    if (c_le > m) {   // (uniform) the cut goes through the items with key K: the (m - c_lt) lowest positions among them
        const int need = m - c_lt;
        uint32_t plo = 0u, phi = 0xffffffffu;
        while (plo < phi) {
            const uint32_t mid = plo + ((phi - plo) >> 1);
            int c = 0;
#pragma unroll
            for (int j = 0; j < NPL; j++) {
                if (j >= nj) continue;
                c += (it[j] != ~0ull && (uint32_t)(it[j] >> 32) == K && (uint32_t)it[j] <= mid) ? 1 : 0;
            }
            c = wave_sum(c);
            if (c >= need) phi = mid;
            else plo = mid + 1u;
        }
        V = ((unsigned long long)K << 32) | plo;
    }
    unsigned long long* runs = s_run[w];
    GH_CT(4);
    int at = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        if (j >= nj) continue;
        const bool take = it[j] <= V;
        const unsigned long long tb = __ballot(take);
        if (take) runs[at + __popcll(tb & lt_mask)] = it[j];
        at += __popcll(tb);
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long x[4];
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = (r * 64 + lane < m) ? runs[r * 64 + lane] : ~0ull;
    GH_CT(5);
    __builtin_amdgcn_wave_barrier();
    const int nr = (m + 63) >> 6;   // runs in use (uniform): k = 100 sorts and merges two, not four
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (r < nr) x[r] = wave_sort64(x[r]);
#pragma unroll
    for (int r = 0; r < 4; r++) runs[r * 64 + lane] = x[r];
    __builtin_amdgcn_wave_barrier();
    GH_CT(6);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r >= nr) continue;
        int rank = lane;
#pragma unroll
        for (int o = 0; o < 4; o++) {
            if (o == r || o >= nr) continue;
            const unsigned long long* ro = runs + o * 64;
            int lo2 = 0, n2 = 64;
#pragma unroll
            for (int st = 0; st < 7; st++) {
                if (n2 > 0) {
                    const int half = n2 >> 1;
                    if (ro[lo2 + half] < x[r]) {
                        lo2 += half + 1;
                        n2 -= half + 1;
                    } else {
                        n2 = half;
                    }
                }
            }
            rank += lo2;
        }
        if (x[r] != ~0ull) list[rank] = x[r];   // ranks are a permutation of 0..m-1
    }
    GH_CT(7);
#undef GH_CT
    if (lane == 0) {
        em.cnt[(int64_t)q * em.cstride] = m;
        tau[q] = m == k ? (uint32_t)(V >> 32) : FLAT_ANY;
    }
}

__global__ __launch_bounds__(256) void k_flat_final(int k, bool smallest, FlatEmit em, float neutral,
                                                    float* __restrict__ distances, int64_t* __restrict__ labels) {
    const int q = blockIdx.x;
    const int m = min(em.cnt[(int64_t)q * em.cstride], k);
    for (int i = threadIdx.x; i < k; i += 256) {
        float dv = neutral;
        int64_t id = -1;
        if (i < m) {
            const unsigned long long item = em.cand[(int64_t)q * em.cap + i];
            const uint32_t key = (uint32_t)(item >> 32);
            dv = key2f(smallest ? key : ~key);
            id = (int64_t)(uint32_t)item;
        }
        distances[(int64_t)q * k + i] = dv;
        labels[(int64_t)q * k + i] = id;
    }
}

int flat_list_cap() { return FLAT_CAP; }

void launch_flat_init(hipStream_t s, bool l2, const float* vals, const int* pos, int nq, int k, int64_t r0,
                      const FlatEmit& em, uint32_t* tau, int* kept) {
    if (nq > 0) hipLaunchKernelGGL(k_flat_init, dim3(nq), dim3(256), 0, s, vals, pos, k, r0, l2, em, tau, kept);
}
void launch_flat_compact(hipStream_t s, int nq, int k, const FlatEmit& em, uint32_t* tau, int* overflow, const FlatLog* log) {
    if (k > 256 || em.cap != FLAT_CAP) {   // callers gate on this
        launch_refused("launch_flat_compact: k > 256 or a candidate list of another capacity");
        return;
    }
    FlatLog lg = log ? *log : FlatLog{};
    static const bool want_dbg = getenv("GAMMA_HIP_COMPACT_DBG") != nullptr;
    static unsigned long long* dbg = nullptr;
    static int shown = 0;
    if (want_dbg) {
        if (!dbg) (void)hipMalloc((void**)&dbg, 64);
        if (shown++ % 16 == 15) {
            unsigned long long t[8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(t, dbg, sizeof(t), hipMemcpyDeviceToHost);
            fprintf(stderr, "flat compact (10 ns ticks, previous launch, query 0): load %llu log %llu minmax %llu bisect %llu tail-count %llu compact %llu sort %llu merge %llu\n",
                    t[1] - t[0], t[2] - t[1], 0ull, t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6]);
        }
        lg.dbg = dbg;
    }
    if (nq > 0) hipLaunchKernelGGL(k_flat_compact, dim3((nq + 3) / 4), dim3(256), 0, s, nq, k, em, tau, overflow, lg);
}
void launch_flat_final(hipStream_t s, bool l2, int nq, int k, const FlatEmit& em, float neutral, float* distances,
                       int64_t* labels) {
    if (nq > 0) hipLaunchKernelGGL(k_flat_final, dim3(nq), dim3(256), 0, s, k, l2, em, neutral, distances, labels);
}

// ------------------------------------------------------------------------------------
// Coarse quantizer, rows whose K-th distance is tied with an entry left outside (flag from
// k_select_wave): redo the row exactly as faiss's HeapResultHandler does (faiss:utils/Heap.h:103-131,
// faiss:impl/ResultHandler.h:112-117; knn_L2sqr uses it below 100 results) -- a max-heap of K
// (FLT_MAX, -1) entries, entries visited in index order, `if (top > dis) replace_top`, then
// heap_reorder -- so that the SAME tied lists are probed.  One wave per flagged row: the 64 lanes find
// the next entry that beats the heap's top, lane 0 sifts it in.  ~1e-5 of the rows on fp32 data.
// ------------------------------------------------------------------------------------
// The walk itself is heap_dev.h's HeapWalk: 64 candidates per ballot, accepted ones pipelined through an LDS heap.
constexpr int CH_MAXK = 1024;   // nprobe the coarse replay covers
constexpr int CH_HEAPK = 128;   // ... through the result heap (below RV_MIN_K = 100 probes; from there on the reservoir)

int coarse_heap_max_k() { return CH_MAXK; }

__global__ __launch_bounds__(256) void k_coarse_heap_fix(const float* __restrict__ mat, int64_t ld, int n, int K,
                                                         int nq, const uint8_t* __restrict__ flag,
                                                         float* __restrict__ out_vals, int* __restrict__ out_pos,
                                                         unsigned long long* __restrict__ tie_stats,
                                                         const int* __restrict__ rows) {
    // flag: row q of the matrix is query q, walked if flag[q]; rows: row i of the matrix is query rows[1 + i], rows[0] rows
    __shared__ __attribute__((aligned(16))) uint2 s_h[4][CH_HEAPK + 2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ri = blockIdx.x * 4 + w;
    if (ri >= nq) return;              // whole wave; no workgroup barrier below
    int q = ri;
    if (rows) {
        if (ri >= rows[0]) return;
        q = rows[1 + ri];
    } else if (!flag[q]) {
        return;
    }
    if (tie_stats && lane == 0) atomicAdd(tie_stats, 1ull);
    uint2* h = s_h[w];
    heap_fill(h, K, lane, 64);         // heap_heapify: (FLT_MAX, -1) everywhere
    __builtin_amdgcn_wave_barrier();
    HeapWalk hw;
    hw.begin(h, K);
    const float* v = mat + (int64_t)ri * ld;
    // sixteen 64-entry blocks of the row in flight ahead of the walk
    float t[16], tn[16];
#pragma unroll
    for (int u = 0; u < 16; u++) t[u] = v[min(u * 64 + lane, n - 1)];
    for (int i0 = 0; i0 < n; i0 += 1024) {
#pragma unroll
        for (int u = 0; u < 16; u++) tn[u] = v[min(i0 + 1024 + u * 64 + lane, n - 1)];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int j = i0 + u * 64 + lane;
            if (i0 + u * 64 < n) hw.accept(j < n, t[u], j);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) t[u] = tn[u];
    }
    hw.drain();
    // heap_reorder (faiss:impl/ResultHandler.h:112-117): K pops; with the heap in registers a level of a sift is a few
    // scalar instructions instead of an LDS round trip
    if (K <= 15) {
        RegHeap<1> rh;
        rh.load(h, K);
        const int real = rh.reorder_pops(K);
        rh.dump(h, K);
        heap_reorder_tail(h, K, real);
    } else {
        par_heap_reorder(h, K);
    }
    for (int r = lane; r < K; r += 64) {
        const uint2 e = h[1 + r];
        out_vals[(int64_t)q * K + r] = (int)e.y < 0 ? INFINITY : __uint_as_float(e.x);
        out_pos[(int64_t)q * K + r] = (int)e.y;
    }
}

// The same rows from 100 probes on: faiss collects those through ReservoirTopN (faiss:utils/distances.cpp:341-358),
// whose choice among tied centroids and their order is not the heap's -- reservoir_dev.h replays it.  One wave per row.
template <int MAXK, int WAVES>   // <256, 4>: four rows per workgroup; <1024, 1>: one (25 KB of LDS per row)
__global__ __launch_bounds__(64 * WAVES) void k_coarse_reservoir_fix(const float* __restrict__ mat, int64_t ld, int n, int K,
                                                                     int nq, const uint8_t* __restrict__ flag,
                                                                     float* __restrict__ out_vals, int* __restrict__ out_pos,
                                                                     unsigned long long* __restrict__ tie_stats,
                                                                     const int* __restrict__ rows) {
    __shared__ __attribute__((aligned(16))) uint2 s_h[WAVES][MAXK + 2];
    __shared__ float s_v[WAVES][reservoir_capacity(MAXK)];
    __shared__ int s_i[WAVES][reservoir_capacity(MAXK)];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ri = blockIdx.x * WAVES + w;
    if (ri >= nq) return;              // whole wave; no workgroup barrier below
    int q = ri;
    if (rows) {
        if (ri >= rows[0]) return;
        q = rows[1 + ri];
    } else if (!flag[q]) {
        return;
    }
    if (tie_stats && lane == 0) atomicAdd(tie_stats, 1ull);
    uint2* h = s_h[w];
    (void)reservoir_row(mat + (int64_t)ri * ld, n, K, s_v[w], s_i[w], h);
    for (int r = lane; r < K; r += 64) {
        const uint2 e = h[1 + r];
        out_vals[(int64_t)q * K + r] = (int)e.y < 0 ? INFINITY : __uint_as_float(e.x);
        out_pos[(int64_t)q * K + r] = (int)e.y;
    }
}
static void launch_coarse_tie_rows(hipStream_t s, const float* mat, int nlist, int K, int nq, const uint8_t* flag,
                                   float* out_vals, int* out_pos, unsigned long long* tie_stats, const int* rows) {
    if (K > 256)
        hipLaunchKernelGGL((k_coarse_reservoir_fix<CH_MAXK, 1>), dim3(nq), dim3(64), 0, s, mat, (int64_t)nlist, nlist, K, nq, flag,
                           out_vals, out_pos, tie_stats, rows);
    else if (K >= RV_MIN_K)
        hipLaunchKernelGGL((k_coarse_reservoir_fix<256, 4>), dim3((nq + 3) / 4), dim3(256), 0, s, mat, (int64_t)nlist, nlist, K, nq, flag,
                           out_vals, out_pos, tie_stats, rows);
    else
        hipLaunchKernelGGL(k_coarse_heap_fix, dim3((nq + 3) / 4), dim3(256), 0, s, mat, (int64_t)nlist, nlist, K, nq, flag,
                           out_vals, out_pos, tie_stats, rows);
}

// top-K nearest centroids of every row of the coarse distance matrix.  tie_flag != nullptr (nq bytes of
// scratch; exact ties): a row with two equal keys among its K + 1 smallest -- which of them is probed, or in which
// order their lists are scanned, is the doing of the reference's heap -- is redone by k_coarse_heap_fix
// (K <= 1024; from 100 probes on the walk is faiss's reservoir, k_coarse_reservoir_fix).
// side / fork / join: the walk of the flagged rows (one wave per row, all latency) may run on a side stream beside the
// caller's next kernels that do not read the assignment; the caller waits for `join` before the first one that does
bool launch_coarse_select(hipStream_t s, const float* mat, int nlist, int nq, int K, float* out_vals, int* out_pos,
                          uint8_t* tie_flag, unsigned long long* tie_stats, hipStream_t side, hipEvent_t fork,
                          hipEvent_t join) {
    static const bool off = getenv("GAMMA_HIP_NO_WAVE_SELECT") != nullptr;
    if (nq <= 0) return false;
    if (!tie_flag || K > CH_MAXK) {
        launch_select_topk(s, true, mat, nlist, nullptr, nlist, nlist, nq, K, out_vals, out_pos);
        return false;
    }
    if (!off && K <= 64 && nlist <= 64 * SW_NPL) {
        hipLaunchKernelGGL((k_select_wave<true, SW_NPL>), dim3((nq + 3) / 4), dim3(256), 0, s, mat, (int64_t)nlist, nullptr, nlist,
                           nq, K, out_vals, out_pos, tie_flag);
    } else {
        launch_select_topk(s, true, mat, nlist, nullptr, nlist, nlist, nq, K, out_vals, out_pos);
        (void)hipMemsetAsync(tie_flag, 0, (size_t)nq, s);
        launch_flag_cut_ties(s, mat, nlist, nullptr, nq, K, out_vals, out_pos, nullptr, tie_flag, nlist, 1);
    }
    hipStream_t rs = s;
    if (side && fork && join) {
        (void)hipEventRecord(fork, s);
        (void)hipStreamWaitEvent(side, fork, 0);
        rs = side;
    }
    launch_coarse_tie_rows(rs, mat, nlist, K, nq, tie_flag, out_vals, out_pos, tie_stats, nullptr);
    if (rs != s) (void)hipEventRecord(join, rs);
    return rs != s;
}

void launch_coarse_heap_rows(hipStream_t s, const float* mat, int nlist, int nq, int K, const int* rows, float* out_vals,
                             int* out_pos, unsigned long long* tie_stats) {
    if (nq <= 0 || K > CH_MAXK) return;
    launch_coarse_tie_rows(s, mat, nlist, K, nq, nullptr, out_vals, out_pos, tie_stats, rows);
}

int select_kpad(int K) {
    int p = 2;
    while (p < K) p <<= 1;
    return p;
}

template <bool SMALLEST>
static void launch_sel(hipStream_t s, const float* vals, int64_t seg_stride, const int* seg_len,
                       int fixed_len, int max_len, int nseg, int K, float* out_vals, int* out_pos,
                       const uint8_t* only) {
    const int Kpad = select_kpad(K);
    const size_t lds = (size_t)(Kpad + CAP) * sizeof(unsigned long long);
#define GH_SEL(NPT)                                                                              \
    hipLaunchKernelGGL((k_select2<SMALLEST, NPT>), dim3(nseg), dim3(256), lds, s, vals, seg_stride, \
                       seg_len, fixed_len, K, Kpad, out_vals, out_pos, only)
    static const bool no_wave = getenv("GAMMA_HIP_NO_WAVE_SELECT") != nullptr;
    if (K <= 64 && max_len <= 512 && !no_wave && !only)    // short rows (the coarse sample): 8 keys per lane
        hipLaunchKernelGGL((k_select_wave<SMALLEST, 8>), dim3((nseg + 3) / 4), dim3(256), 0, s, vals, seg_stride,
                           seg_len, fixed_len, nseg, K, out_vals, out_pos, nullptr);
    else if (K <= 64 && max_len <= 16384 && !no_wave && !only)   // longer rows: the streaming kernel wins
        hipLaunchKernelGGL((k_select_wave<SMALLEST, SW_NPL>), dim3((nseg + 3) / 4), dim3(256), 0, s, vals, seg_stride,
                           seg_len, fixed_len, nseg, K, out_vals, out_pos, nullptr);
    else if (max_len <= 256 * 4) GH_SEL(4);
    else if (max_len <= 256 * 16) GH_SEL(16);
    else if (K <= 1024 && (seg_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(vals) & 15) == 0 &&
             !getenv("GAMMA_HIP_NO_STREAM_SELECT"))
        hipLaunchKernelGGL((k_select_stream<SMALLEST>), dim3(nseg), dim3(256), 0, s, vals, seg_stride,
                           seg_len, fixed_len, K, Kpad, out_vals, out_pos, only);
    else if (K <= 2048 && (seg_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(vals) & 15) == 0 &&
             !getenv("GAMMA_HIP_NO_STREAM_SELECT") && !getenv("GAMMA_HIP_NO_STREAM_SELECT_2K"))
        hipLaunchKernelGGL((k_select_stream<SMALLEST, 4096>), dim3(nseg), dim3(256), 0, s, vals, seg_stride,
                           seg_len, fixed_len, K, Kpad, out_vals, out_pos, only);
    else GH_SEL(0);
#undef GH_SEL
}

void launch_select_topk(hipStream_t s, bool smallest, const float* vals, int64_t seg_stride,
                        const int* seg_len, int fixed_len, int max_len, int nseg, int K,
                        float* out_vals, int* out_pos, const uint8_t* only) {
    if (nseg <= 0 || K <= 0) return;
    if (smallest)
        launch_sel<true>(s, vals, seg_stride, seg_len, fixed_len, max_len, nseg, K, out_vals, out_pos, only);
    else
        launch_sel<false>(s, vals, seg_stride, seg_len, fixed_len, max_len, nseg, K, out_vals, out_pos, only);
}

void launch_select_final(hipStream_t s, bool smallest, const unsigned long long* surv, const int* gcnt,
                         int nslices, int slice_cap, const unsigned long long* ready, const int* pair_off, int P,
                         int nq, int K, const int64_t* pair_base, const int64_t* ids, uint8_t* flag, float* out_vals,
                         int* out_pos, int64_t* out_ids, uint8_t* cut_tie, unsigned long long* tie_stats,
                         int* rq_list, int* rq_count, unsigned long long* bound_stat) {
    if (nq <= 0) return;
    if (P > 128 || K > 1024 || (K <= 256 && nslices > 64)) {   // callers gate on this (gamma_hip_search.cpp, ivfpq_stage_a)
        launch_refused("launch_select_final: nprobe > 128, recall_num > 1024 or more than 64 survivor slices");
        return;
    }
    static const bool no_wave8 = getenv("GAMMA_HIP_NO_SELECT_WAVE8") != nullptr;
    if (K > 256 && K <= 512 && nslices <= 64 && !no_wave8) {   // a wave per query with eight sorted runs
#define GH_SF8(SM, PM)                                                                                           \
    hipLaunchKernelGGL((k_select_final8<SM, PM>), dim3((nq + 3) / 4), dim3(256), 0, s, surv, gcnt, nslices,   \
                       slice_cap, ready, pair_off, P, nq, K, pair_base, ids, flag,                               \
                       out_vals, out_pos, out_ids, cut_tie, tie_stats, rq_list, rq_count, bound_stat)
        if (smallest) {
            if (P <= 64) GH_SF8(true, 64);
            else GH_SF8(true, 128);
        } else {
            if (P <= 64) GH_SF8(false, 64);
            else GH_SF8(false, 128);
        }
#undef GH_SF8
        return;
    }
    if (K > 256) {          // one workgroup per query (recall_num up to 1024: the callers' gate)
        if (smallest)
            hipLaunchKernelGGL((k_select_final_wg<true>), dim3(nq), dim3(256), 0, s, surv, gcnt, nslices, slice_cap, ready, pair_off, P, nq,
                               K, pair_base, ids, flag, out_vals, out_pos, out_ids, cut_tie, tie_stats, rq_list, rq_count, bound_stat);
        else
            hipLaunchKernelGGL((k_select_final_wg<false>), dim3(nq), dim3(256), 0, s, surv, gcnt, nslices, slice_cap, ready, pair_off, P,
                               nq, K, pair_base, ids, flag, out_vals, out_pos, out_ids, cut_tie, tie_stats, rq_list, rq_count, bound_stat);
        return;
    }
#define GH_SF(SM, PM)                                                                                        \
    hipLaunchKernelGGL((k_select_final<SM, PM>), dim3((nq + 3) / 4), dim3(256), 0, s, surv, gcnt, nslices,   \
                       slice_cap, ready, pair_off, P, nq, K, pair_base, ids, flag,                           \
                       out_vals, out_pos, out_ids, cut_tie, tie_stats, rq_list, rq_count, bound_stat)
    if (smallest) {
        if (P <= 64) GH_SF(true, 64);
        else GH_SF(true, 128);
    } else {
        if (P <= 64) GH_SF(false, 64);
        else GH_SF(false, 128);
    }
#undef GH_SF
}

// ====================================================================================
// Small batches (nq <= 16): a search is a chain of dependent, nearly empty kernels, each of which costs ~4 us of
// launch + drain on this chip whatever it does, and a lone wave runs dependent instructions at ~10 cycles each.
// These two kernels fold seven launches into two, 16 waves per query:
//   k_small_coarse_select = top-nprobe of the coarse distances + the pairs' slab offsets (k_select_wave + k_pair_offsets)
//   k_small_tail          = top-recall_num of the ADC slab + ids + exact re-rank + top-k + output
//                           (k_select_stream + k_map_candidates + k_rerank_dist + k_select_wave + k_finalize_topk)
// One workgroup per query; same (key, position) orders, same arithmetic (rerank_dev.h) as the kernels they replace.
// ====================================================================================
}  // namespace gh
#include "rerank_dev.h"
namespace gh {
namespace {
constexpr int SM_NT = 1024, SM_NW = SM_NT / 64, SM_BINS = 2048;
constexpr uint32_t SM_KEYSENT = 0xff800000u;   // key of the filtered-entry marker: +inf (smallest first) / -inf (largest first)
constexpr int SM_SLICE = 16 * SM_NT;          // a row slice whose keys one workgroup keeps in registers

// exclusive scan of one int per thread over the 1024-thread block; s_w: SM_NW ints.  Two barriers.
__device__ __forceinline__ int block_excl_scan_sm(int v, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SM_NW; i++) {
        const int t = s_w[i];
        if (i < w) base += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return base + incl - v;
}

// Rank sort of a[0..n) (distinct 64-bit items) by the 16 waves of the block: item group g = 64 items, and the j range
// of the comparisons is cut into slices so that every wave has work (block_rank_sort leaves 12 of 16 waves idle at
// n = 200 and the other four walk all of a[]).  s_rank: n ints of scratch.
__device__ __forceinline__ void block_rank_sort_sm(unsigned long long* a, int n, int* s_rank) {
    if (n > SM_NT) {   // uniform
        block_rank_sort<SM_NT, 2>(a, n);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ngroups = (n + 63) >> 6;
    int lg = 0;
    while ((1 << lg) < ngroups) lg++;
    static_assert(SM_NW == 16, "the j range is cut into SM_NW >> lg = 2^(4 - lg) slices");
    for (int i = tid; i < n; i += SM_NT) s_rank[i] = 0;
    __syncthreads();
    const int g = w & ((1 << lg) - 1), sl = w >> lg;
    const int i = g * 64 + lane;
    const unsigned long long item = i < n ? a[i] : ~0ull;
    if (g < ngroups) {   // uniform per wave
        // (slices is the power of two SM_NW >> lg and sl * n < 2^14: shifts, not the 64-bit division the general form costs)
        const int j0 = (sl * n) >> (4 - lg), j1 = ((sl + 1) * n) >> (4 - lg);
        int rk = 0, j = j0;
        for (; j + 8 <= j1; j += 8) {
            unsigned long long x[8];
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = a[j + e];
#pragma unroll
            for (int e = 0; e < 8; e++) rk += x[e] < item ? 1 : 0;
        }
        for (; j < j1; j++) rk += a[j] < item ? 1 : 0;
        if (i < n && rk) atomicAdd(&s_rank[i], rk);
    }
    __syncthreads();
    if (sl == 0 && i < n) a[s_rank[i]] = item;
    __syncthreads();
}

// The K smallest (key, position) items of v[0..n) into s_it[0..min(n, K)), unsorted.  The K-th key is found exactly
// by histogram passes over [smallest key, largest key] cut into <= 2048 equal bins, then over the bin holding the
// K-th, and so on (2 passes for distances spread over ~2^22 key values); a radix pass on raw float bits would put a
// whole row into three or four bins and serialise its LDS atomics.  Items below the K-th key are all taken, items AT
// it in position order (every thread owns a contiguous range of the row).
template <bool SMALLEST>
__device__ __forceinline__ int block_select_items(const float* __restrict__ v, int n, int K, unsigned long long* s_it,
                                                  int* s_hist, int* s_w, uint32_t* s_pick) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (n <= K) {
        for (int i = tid; i < n; i += SM_NT) s_it[i] = ((unsigned long long)sel_key<SMALLEST>(v[i]) << 32) | (unsigned)i;
        return n;
    }
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    for (int i = tid; i < n; i += SM_NT) {
        const uint32_t key = sel_key<SMALLEST>(v[i]);
        kmin = key < kmin ? key : kmin;
        kmax = key > kmax ? key : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = (uint32_t)__shfl_xor((int)kmin, o, 64), b = (uint32_t)__shfl_xor((int)kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) {
        s_w[w] = (int)kmin;
        s_w[SM_NW + w] = (int)kmax;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SM_NW; i++) {
        const uint32_t a = (uint32_t)s_w[i], b = (uint32_t)s_w[SM_NW + i];
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    __syncthreads();
    uint32_t lo = kmin, range = kmax - kmin;   // keys of interest: lo .. lo + range
    int need = K;
    for (;;) {
        const int bits = 32 - __clz((int)range | 0) - (range == 0 ? 0 : 0);
        const int nbits = range == 0 ? 0 : bits;
        const int shift = nbits > 11 ? nbits - 11 : 0;
        for (int i = tid; i < SM_BINS; i += SM_NT) s_hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += SM_NT) {
            const uint32_t off = sel_key<SMALLEST>(v[i]) - lo;   // wraps above range for keys below lo
            if (off <= range) atomicAdd(&s_hist[off >> shift], 1);
        }
        __syncthreads();
        constexpr int PER = SM_BINS / SM_NT;
        int loc = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) loc += s_hist[tid * PER + u];
        int tot;
        const int ex = block_excl_scan_sm(loc, s_w, tot);
        if (ex < need && need <= ex + loc) {
            int run = ex;
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int hcur = s_hist[tid * PER + u];
                if (run < need && need <= run + hcur) {
                    s_pick[0] = (uint32_t)(tid * PER + u);
                    s_pick[1] = (uint32_t)(need - run);
                }
                run += hcur;
            }
        }
        __syncthreads();
        const uint32_t b = s_pick[0];
        need = (int)s_pick[1];
        const uint32_t sub = b << shift;
        lo += sub;
        const uint32_t rest = range - sub;
        range = shift == 0 ? 0u : (rest < ((1u << shift) - 1u) ? rest : ((1u << shift) - 1u));
        __syncthreads();
        if (shift == 0) break;
    }
    const uint32_t kth = lo;
    const int c = (n + SM_NT - 1) / SM_NT;
    const int i0 = min(n, tid * c), i1 = min(n, i0 + c);
    int nl = 0, ne = 0;
    for (int i = i0; i < i1; i++) {
        const uint32_t key = sel_key<SMALLEST>(v[i]);
        nl += key < kth ? 1 : 0;
        ne += key == kth ? 1 : 0;
    }
    int tl, te;
    int ol = block_excl_scan_sm(nl, s_w, tl);
    int oe = block_excl_scan_sm(ne, s_w, te);
    const int eq_base = K - need;   // == tl
    for (int i = i0; i < i1; i++) {
        const uint32_t key = sel_key<SMALLEST>(v[i]);
        const unsigned long long item = ((unsigned long long)key << 32) | (unsigned)i;
        if (key < kth) s_it[ol++] = item;
        else if (key == kth) {
            if (oe < need) s_it[eq_base + oe] = item;
            oe++;
        }
    }
    return K;
}

// The min(n, K) smallest (key, position) items of v[0..n), SORTED, in s_it (capacity MAXI * 1024 items).
// One pass: the row's keys stay in registers (NREG per thread when the row fits), one histogram over [min, max]
// finds the bin holding the K-th smallest; everything up to and including that bin -- K plus a handful -- is appended
// and rank-sorted.  Only when that bin is crowded (mass ties) does block_select_items refine it pass by pass.
template <bool SMALLEST, int NREG, int MAXI>
__device__ __forceinline__ int block_select_sorted(const float* __restrict__ v, int n, int K, unsigned long long* s_it,
                                                   int* s_hist, int* s_w, uint32_t* s_pick, int* n_sorted = nullptr) {
    // n_sorted: how many sorted items s_it holds -- K plus whatever else shared the K-th item's bin (so s_it[K], if
    // there, is the (K+1)-th smallest: exact ties look at it); -1: only K, and whether the cut went through a group of
    // equal keys is not known here (crowded bin)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (n <= K) {
        for (int i = tid; i < n; i += SM_NT) s_it[i] = ((unsigned long long)sel_key<SMALLEST>(v[i]) << 32) | (unsigned)i;
        block_rank_sort_sm(s_it, n, s_hist);
        if (n_sorted) *n_sorted = n;
        return n;
    }
    const bool inreg = n <= NREG * SM_NT;   // uniform
    uint32_t kreg[NREG];
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    if (inreg) {
#pragma unroll
        for (int u = 0; u < NREG; u++) {
            const int i = tid + SM_NT * u;
            kreg[u] = i < n ? sel_key<SMALLEST>(v[i]) : 0xffffffffu;
            if (i < n) kmin = kreg[u] < kmin ? kreg[u] : kmin;
            if (kreg[u] < SM_KEYSENT) kmax = kreg[u] > kmax ? kreg[u] : kmax;
        }
    } else {   // (batching these loads by hand, NREG in flight per thread, measured slower than the plain loops)
        for (int i = tid; i < n; i += SM_NT) {
            const uint32_t key = sel_key<SMALLEST>(v[i]);
            kmin = key < kmin ? key : kmin;
            if (key < SM_KEYSENT) kmax = key > kmax ? key : kmax;
        }
    }
    kmin = __reduce_min_sync(~0ull, kmin);   // DPP row operations, not LDS shuffles
    kmax = __reduce_max_sync(~0ull, kmax);
    for (int i = tid; i < SM_BINS; i += SM_NT) s_hist[i] = 0;
    if (lane == 0) {
        s_w[w] = (int)kmin;
        s_w[SM_NW + w] = (int)kmax;
    }
    if (tid == 0) s_pick[2] = 0;   // append counter
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SM_NW; i++) {
        const uint32_t a = (uint32_t)s_w[i], b = (uint32_t)s_w[SM_NW + i];
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    // the range is that of the VALID keys: the marker of filtered entries (+-inf, deleted or filtered documents) is
    // far from any distance and would leave the histogram a handful of useful bins; it lands in the last bin
    if (kmax < kmin) kmax = kmin;
    const uint32_t range = kmax - kmin;
    const int nbits = range == 0 ? 0 : 32 - __clz((int)range);
    const int shift = nbits > 11 ? nbits - 11 : 0;
    auto bin_of = [&](uint32_t key) -> uint32_t {
        const uint32_t b = (key - kmin) >> shift;
        return b < (uint32_t)(SM_BINS - 1) ? b : (uint32_t)(SM_BINS - 1);
    };
    if (inreg) {
#pragma unroll
        for (int u = 0; u < NREG; u++)
            if (tid + SM_NT * u < n) atomicAdd(&s_hist[bin_of(kreg[u])], 1);
    } else {
        for (int i = tid; i < n; i += SM_NT) atomicAdd(&s_hist[bin_of(sel_key<SMALLEST>(v[i]))], 1);
    }
    __syncthreads();
    constexpr int PER = SM_BINS / SM_NT;
    int loc = 0;
#pragma unroll
    for (int u = 0; u < PER; u++) loc += s_hist[tid * PER + u];
    int tot;
    const int ex = block_excl_scan_sm(loc, s_w, tot);
    if (ex < K && K <= ex + loc) {
        int run = ex;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int hcur = s_hist[tid * PER + u];
            if (run < K && K <= run + hcur) {
                s_pick[0] = (uint32_t)(tid * PER + u);
                s_pick[1] = (uint32_t)(run + hcur);   // items up to and including the bin
            }
            run += hcur;
        }
    }
    __syncthreads();
    const uint32_t b = s_pick[0];
    const int T = (int)s_pick[1];
    if (T > MAXI * SM_NT) {   // uniform: the boundary bin is crowded
        __syncthreads();
        block_select_items<SMALLEST>(v, n, K, s_it, s_hist, s_w, s_pick);
        block_rank_sort_sm(s_it, K, s_hist);
        if (n_sorted) *n_sorted = -1;
        return K;
    }
    if (n_sorted) *n_sorted = T;
    auto append = [&](uint32_t key, int i, bool live) {
        const bool take = live && bin_of(key) <= b;
        const unsigned long long mask = __ballot(take);
        if (mask) {   // uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_pick[2], (uint32_t)__popcll(mask));
            base = (uint32_t)__shfl((int)base, 0, 64);
            if (take) s_it[base + __popcll(mask & ((1ull << lane) - 1ull))] = ((unsigned long long)key << 32) | (unsigned)i;
        }
    };
    if (inreg) {
#pragma unroll
        for (int u = 0; u < NREG; u++) append(kreg[u], tid + SM_NT * u, tid + SM_NT * u < n);
    } else {
        for (int i0 = 0; i0 < n; i0 += SM_NT) {
            const int i = i0 + tid;
            append(i < n ? sel_key<SMALLEST>(v[i]) : 0u, i, i < n);
        }
    }
    block_rank_sort_sm(s_it, T, s_hist);   // the histogram is done with
    return K;
}
}  // namespace

struct SmallSelectArgs {
    const float* mat;
    int nlist, K;
    float* out_vals;
    int* out_pos;
    const int* list_len;
    const uint8_t* list_mask;
    const int64_t* list_off;
    int* pair_off;
    int* q_total;
    int64_t* pair_base;
    const float* x;    // inner-product metric: dis0 of the pairs
    const float* cc;
    int d;
    float* pair_ip;
    uint32_t* units;   // long lists: the scan's work list, (q << 20 | probe << 13 | chunk) per chunk_len codes of a pair
    int* unit_count;   // zeroed by the launch before
    int chunk_len;
    int exact_ties;    // rows with equal keys among the K + 1 smallest are redone through the reference's heap
    unsigned long long* tie_stats;
};
__device__ __forceinline__ void small_coarse_select_body(int q, const SmallSelectArgs& A) {
    const float* __restrict__ mat = A.mat;
    const int nlist = A.nlist, K = A.K;
    float* __restrict__ out_vals = A.out_vals;
    int* __restrict__ out_pos = A.out_pos;
    const int* __restrict__ list_len = A.list_len;
    const uint8_t* __restrict__ list_mask = A.list_mask;
    const int64_t* __restrict__ list_off = A.list_off;
    int* __restrict__ pair_off = A.pair_off;
    int* __restrict__ q_total = A.q_total;
    int64_t* __restrict__ pair_base = A.pair_base;
    __shared__ int s_hist[SM_BINS];
    __shared__ unsigned long long s_it[SM_NT];
    __shared__ int s_w[2 * SM_NW];
    __shared__ uint32_t s_pick[3];
    __shared__ int s_probe[128];
    const int tid = threadIdx.x, lane = tid & 63;
    const float* v = mat + (int64_t)q * nlist;
    int nsorted = 0;
    int cnt = block_select_sorted<true, 4, 1>(v, nlist, K, s_it, s_hist, s_w, s_pick, &nsorted);   // trailing barrier inside
    if (A.exact_ties) {   // uniform
        // two equal keys among the K + 1 smallest: which of them is probed, and in which order equal ones are scanned,
        // is the doing of faiss's HeapResultHandler -- the row is walked the way it walks it (k_coarse_heap_fix)
        __shared__ __attribute__((aligned(16))) uint2 s_heap[128 + 2];
        __shared__ int s_nreal;
        // (the selection leaves the (K+1)-th smallest item behind the K-th whenever it shares its histogram bin; from
        //  another bin it cannot carry the same key)
        bool eq = false;
        const int have = nsorted < 0 ? cnt : min(nsorted, K + 1);
        for (int r = tid; r + 1 < have; r += SM_NT) eq |= (uint32_t)(s_it[r] >> 32) == (uint32_t)(s_it[r + 1] >> 32);
        int tie = __syncthreads_or((eq || nsorted < 0) ? 1 : 0);
        if (tie && K >= RV_MIN_K) {   // from 100 probes on the reference collects through its reservoir (reservoir_dev.h)
            __shared__ float s_rv[reservoir_capacity(128)];
            __shared__ int s_ri[reservoir_capacity(128)];
            if (tid < 64) {
                if (A.tie_stats && tid == 0) atomicAdd(A.tie_stats, 1ull);
                const int nreal = reservoir_row(v, nlist, K, s_rv, s_ri, s_heap);
                for (int r = lane; r < K; r += 64) {
                    const uint2 e = s_heap[1 + r];
                    s_it[r] = (int)e.y < 0 ? ~0ull : (((unsigned long long)f2key(__uint_as_float(e.x)) << 32) | e.y);
                }
                if (lane == 0) s_nreal = nreal;
            }
            __syncthreads();
            cnt = s_nreal;
        } else if (tie) {
            heap_fill(s_heap, K, tid, SM_NT);
            __syncthreads();
            if (tid < 64) {
                if (A.tie_stats && tid == 0) atomicAdd(A.tie_stats, 1ull);
                HeapWalk hw;
                hw.begin(s_heap, K);
                float t[16], tn[16];
#pragma unroll
                for (int u = 0; u < 16; u++) t[u] = v[min(u * 64 + lane, nlist - 1)];
                for (int i0 = 0; i0 < nlist; i0 += 1024) {
#pragma unroll
                    for (int u = 0; u < 16; u++) tn[u] = v[min(i0 + 1024 + u * 64 + lane, nlist - 1)];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int j = i0 + u * 64 + lane;
                        if (i0 + u * 64 < nlist) hw.accept(j < nlist, t[u], j);
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) t[u] = tn[u];
                }
                hw.drain();
                const int nreal = par_heap_reorder(s_heap, K);
                for (int r = lane; r < K; r += 64) {
                    const uint2 e = s_heap[1 + r];
                    s_it[r] = (int)e.y < 0 ? ~0ull : (((unsigned long long)f2key(__uint_as_float(e.x)) << 32) | e.y);
                }
                if (lane == 0) s_nreal = nreal;
            }
            __syncthreads();
            cnt = s_nreal;
        }
    }
    if (tid < 64) {   // cnt <= K <= 128: the first wave, 64 probes per round
        int run_off = 0;   // slab offset of the round's first probe
        for (int b0 = 0; b0 < K; b0 += 64) {   // uniform
            const int i = b0 + lane;
            int pos = -1;
            float val = INFINITY;
            if (i < cnt) {
                pos = (int)(uint32_t)s_it[i];
                val = v[pos];
                if (val == INFINITY) pos = -1;
            }
            // k_pair_offsets for this query
            int len = 0;
            int64_t lbase = 0;
            if (i < K) {
                out_vals[(int64_t)q * K + i] = val;
                out_pos[(int64_t)q * K + i] = pos;
                s_probe[i] = pos;
                if (pos >= 0 && pos < nlist && (!list_mask || list_mask[pos])) {
                    len = list_len[pos];
                    lbase = list_off[pos];
                }
            }
            const int incl = wave_incl_scan(len);
            if (i < K) {
                pair_off[(int64_t)q * (K + 1) + i] = run_off + incl - len;
                if (pair_base) pair_base[(int64_t)q * K + i] = lbase;
            }
            if (A.units) {   // uniform
                const int nch = (len + A.chunk_len - 1) / A.chunk_len;
                const int uincl = wave_incl_scan(nch);
                const int utot = __shfl(uincl, 63, 64);
                int ub = 0;
                if (lane == 0 && utot) ub = atomicAdd(A.unit_count, utot);
                ub = __shfl(ub, 0, 64) + uincl - nch;
                for (int c = 0; c < nch; c++) A.units[ub + c] = ((uint32_t)q << 20) | ((uint32_t)i << 13) | (uint32_t)c;
            }
            run_off += __shfl(incl, 63, 64);
        }
        if (lane == 0) {
            pair_off[(int64_t)q * (K + 1) + K] = run_off;
            q_total[q] = run_off;
        }
    }
    if (A.pair_ip) {   // k_pair_ip for this query: eight threads per probe (rerank_dev.h: fvec_inner_product's lane order)
        __syncthreads();
        const int g = tid >> 3, l8 = tid & 7;
        for (int p0 = 0; p0 < K; p0 += SM_NT / 8) {
            const int p = p0 + g;
            const int l = p < K ? s_probe[p] : -1;
            const bool live = l >= 0 && l < nlist;
            const float ip = rerank_dist8<false>(A.x + (int64_t)q * A.d, A.cc + (int64_t)(live ? l : 0) * A.d, A.d, l8, live);
            if (l8 == 0 && p < K) A.pair_ip[(int64_t)q * K + p] = live ? ip : 0.f;
        }
    }
}

__global__ __launch_bounds__(SM_NT) void k_small_coarse_select(SmallSelectArgs A) { small_coarse_select_body(blockIdx.x, A); }

void launch_small_coarse_select(hipStream_t s, const float* mat, int nlist, int nq, int P, float* out_vals, int* out_pos,
                                const int* list_len, const uint8_t* list_mask, const int64_t* list_off, int* pair_off,
                                int* q_total, int64_t* pair_base, const float* x, const float* cc, int d, float* pair_ip,
                                uint32_t* units, int* unit_count, int chunk_len, int exact_ties,
                                unsigned long long* tie_stats) {
    if (nq <= 0) return;
    if (P > 128 || (units && (nq > 4096 || chunk_len < 1))) {   // callers gate on this
        launch_refused("launch_small_coarse_select: nprobe > 128, or a unit list for more than 4096 queries");
        return;
    }
    const SmallSelectArgs A{mat, nlist, P, out_vals, out_pos, list_len, list_mask, list_off, pair_off, q_total, pair_base,
                            x, cc, d, pair_ip, units, unit_count, chunk_len, exact_ties, tie_stats};
    hipLaunchKernelGGL(k_small_coarse_select, dim3(nq), dim3(SM_NT), 0, s, A);
}

// Long candidate rows (big indexes: nprobe x list length in the 10^5s) are too much for the one workgroup of
// k_small_tail: the row is cut into S = min(smax, ceil(n / SM_SLICE)) slices, workgroup (q, s) leaves the
// min(recall_num, slice) best of slice s -- value and slab position, best first -- and the tail selects among S x R.
// Equal keys keep their slab order through both levels (slices are in slab order, a slice's survivors in
// (key, position) order), so the result is the one-level selection's, ties included.
template <bool L2>
__global__ __launch_bounds__(SM_NT) void k_small_presel(const float* __restrict__ slab, int64_t q_stride,
                                                        const int* __restrict__ q_total, int R, int smax,
                                                        float* __restrict__ pre_val, int* __restrict__ pre_pos, int fixed_n,
                                                        int* __restrict__ pre_cut) {
    // pre_cut[q][slice] (exact ties; may be null): the slice's own top-R cut went through a group of equal values
    __shared__ int s_hist[SM_BINS];
    __shared__ unsigned long long s_it[2 * SM_NT];
    __shared__ int s_w[2 * SM_NW];
    __shared__ uint32_t s_pick[3];
    const int q = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
    const int n = q_total ? q_total[q] : fixed_n;   // (flat search: every query's row is the whole store)
    const int S = min(smax, (n + SM_SLICE - 1) / SM_SLICE);
    if (sl >= S) return;   // uniform
    const int per = (((n + S - 1) / S) + 3) & ~3;
    const int i0 = min(n, sl * per), m = min(n, i0 + per) - i0;
    const float* v = slab + (int64_t)q * q_stride + i0;
    int nsorted = 0;
    const int cnt = m > 0 ? block_select_sorted<L2, 16, 2>(v, m, R, s_it, s_hist, s_w, s_pick, &nsorted) : 0;   // barriers inside
    if (m <= 0) __syncthreads();
    if (pre_cut) {
        int cut = 0;
        if (cnt == R && m > R) {   // uniform
            const uint32_t vk = (uint32_t)(s_it[R - 1] >> 32);
            if (nsorted >= 0) {
                cut = nsorted > R && (uint32_t)(s_it[R] >> 32) == vk;
            } else {   // crowded bin: count the value over the slice
                int in_sel = 0, loc = 0, in_all;
                for (int r0 = 0; r0 < R; r0 += SM_NT)
                    in_sel += __syncthreads_count(r0 + tid < R && (uint32_t)(s_it[min(r0 + tid, R - 1)] >> 32) == vk);
                for (int i = tid; i < m; i += SM_NT) loc += sel_key<L2>(v[i]) == vk ? 1 : 0;
                (void)block_excl_scan_sm(min(loc, 2048), s_w, in_all);
                cut = in_all > in_sel;
            }
        }
        if (tid == 0) pre_cut[(int64_t)q * smax + sl] = cut;
    }
    const int64_t o = ((int64_t)q * smax + sl) * R;
    for (int r = tid; r < R; r += SM_NT) {
        float val = L2 ? INFINITY : -INFINITY;
        int pos = -1;
        if (r < cnt) {
            const unsigned long long it = s_it[r];
            const uint32_t key = (uint32_t)(it >> 32);
            val = key2f(L2 ? key : ~key);
            pos = i0 + (int)(uint32_t)it;
        }
        pre_val[o + r] = val;
        pre_pos[o + r] = pos;
    }
}

template <bool L2>
__global__ __launch_bounds__(SM_NT) void k_small_tail(const float* __restrict__ slab, int64_t q_stride,
                                                      const int* __restrict__ q_total, int R, int P,
                                                      const int* __restrict__ probe_list,
                                                      const int* __restrict__ pair_off,
                                                      const int64_t* __restrict__ list_off,
                                                      const int64_t* __restrict__ ids, float* __restrict__ cand_dis,
                                                      int* __restrict__ cand_pos, int64_t* __restrict__ cand_ids,
                                                      int has_rank, const float* __restrict__ x, int d,
                                                      const float* __restrict__ raw, int64_t nraw, int k,
                                                      float min_score, float max_score, float neutral,
                                                      float* __restrict__ distances, int64_t* __restrict__ labels,
                                                      const float* __restrict__ pre_val, const int* __restrict__ pre_pos,
                                                      int smax, int fixed_n, unsigned long long* __restrict__ dbg,
                                                      int exact_ties, TieReplayArgs tr, unsigned long long* tie_stats) {
    extern __shared__ __attribute__((aligned(16))) char s_tie_lds[];   // the replay's workspace (exact ties)
#define GH_T(i) do { if (dbg && threadIdx.x == 0 && blockIdx.x == 0) dbg[i] = wall_clock64(); } while (0)
    __shared__ int s_hist[SM_BINS];
    __shared__ unsigned long long s_it[2 * SM_NT];
    __shared__ int64_t s_id[1024];
    __shared__ float s_val[1024];
    __shared__ int s_w[2 * SM_NW];
    __shared__ uint32_t s_pick[3];
    __shared__ int s_off[129];
    __shared__ int64_t s_lbase[128];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* v = slab + (int64_t)q * q_stride;
    int n = q_total ? q_total[q] : fixed_n;   // flat search (no lists: pair_off == nullptr, a position IS the vector id)
    const int* ppos = nullptr;
    if (smax > 0) {   // long rows: k_small_presel has left min(R, slice) candidates of each of the row's slices
        v = pre_val + (int64_t)q * smax * R;
        ppos = pre_pos + (int64_t)q * smax * R;
        n = min(smax, (n + SM_SLICE - 1) / SM_SLICE) * R;
    }
    const float sentinel = L2 ? INFINITY : -INFINITY;
    // the pairs' slab offsets and list bases: loaded while the selection runs, read from LDS by the id look-up
    GH_T(0);
    if (pair_off && tid <= P) s_off[tid] = pair_off[(int64_t)q * (P + 1) + tid];
    if (pair_off && tid < P) {
        const int l = probe_list[(int64_t)q * P + tid];
        s_lbase[tid] = l >= 0 ? list_off[l] : 0;
    }
    int nsorted = 0;
    const int cnt = block_select_sorted<L2, 16, 2>(v, n, R, s_it, s_hist, s_w, s_pick, &nsorted);   // barriers inside
    GH_T(1);
    // exact ties (ties.hip): does the top-R cut go through a group of equal ADC distances?  Counted over the whole
    // slab row (a pre-selected slice may have dropped members of the group)
    int tie = 0;
    if (exact_ties && cnt == R) {   // uniform
        const uint32_t vk = (uint32_t)(s_it[R - 1] >> 32);
        if (vk != (L2 ? f2key(INFINITY) : ~f2key(-INFINITY))) {
            if (smax == 0 && nsorted >= 0) {
                // the (R+1)-th smallest item is behind the R-th whenever it shares its histogram bin; from another bin it
                // cannot carry the same key
                tie = nsorted > R && (uint32_t)(s_it[R] >> 32) == vk;
            } else {
                // crowded bin: count the key over the whole slab row
                int in_sel = 0;
                for (int r0 = 0; r0 < R; r0 += SM_NT)
                    in_sel += __syncthreads_count(r0 + tid < R && (uint32_t)(s_it[min(r0 + tid, R - 1)] >> 32) == vk);
                int loc = 0, in_all;
                bool slice_cut = false;
                if (smax > 0) {
                    // pre-selected row: the slices' tables hold every entry at the cut value unless a slice's own cut
                    // dropped some -- its table ends at that value and its own cut went through a tie (k_small_presel's
                    // flag).  n entries to look at instead of the row's hundreds of thousands.
                    const int* pcut = pre_pos + (int64_t)gridDim.x * smax * R + (int64_t)q * smax;
                    for (int i = tid; i < n; i += SM_NT) {
                        const bool eq = sel_key<L2>(v[i]) == vk;
                        loc += eq ? 1 : 0;
                        slice_cut |= eq && (i % R) == R - 1 && ppos[i] >= 0 && pcut[i / R] != 0;
                    }
                } else {
                    const float* v0 = slab + (int64_t)q * q_stride;
                    const int n0 = q_total ? q_total[q] : fixed_n;
                    for (int i = tid; i < n0; i += SM_NT) loc += sel_key<L2>(v0[i]) == vk ? 1 : 0;
                }
                (void)block_excl_scan_sm(min(loc, 2048), s_w, in_all);
                tie = in_all > in_sel || __syncthreads_or(slice_cut ? 1 : 0);
            }
        }
    }
    if (dbg && tid == 0 && q == 0) {
        dbg[6] = (unsigned long long)n;
        dbg[7] = (unsigned long long)s_pick[1];
    }
    // top-R table of the query: ADC distance, slab position, vector id (k_map_candidates)
    const int* off = s_off;
    for (int r = tid; r < R; r += SM_NT) {
        float val = sentinel;
        int pos = -1;
        int64_t id = -1;
        if (r < cnt) {
            pos = (int)(uint32_t)s_it[r];
            val = v[pos];
            if (val == sentinel) pos = -1;
            else if (ppos) pos = ppos[pos];
        }
        if (pos >= 0 && !pair_off) {
            id = pos;
        } else if (pos >= 0) {
            int lo = 0, hi = P - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (off[mid] <= pos) lo = mid; else hi = mid - 1;
            }
            id = ids[s_lbase[lo] + (pos - off[lo])] & 0x7fffffffffffffffLL;
        }
        cand_dis[(int64_t)q * R + r] = val;
        cand_pos[(int64_t)q * R + r] = pos;
        cand_ids[(int64_t)q * R + r] = id;
        s_id[r] = id;
        s_val[r] = val;
    }
    __syncthreads();
    GH_T(2);
    if (has_rank) {   // k_rerank_topk, 128 candidates in flight
        const int l = tid & 7, g = tid >> 3;
        const float* xq = x + (int64_t)q * d;
        for (int r0 = 0; r0 < R; r0 += SM_NT / 8) {
            const int r = r0 + g;
            int64_t id = -1;
            if (r < R) id = s_id[r];
            const bool live = id >= 0 && id < nraw;
            float dis = rerank_dist8<L2>(xq, raw + (live ? id : 0) * d, d, l, live);
            if (l == 0 && r < R) {
                if (!live || !(dis <= max_score && dis >= min_score)) dis = sentinel;
                const uint32_t key = L2 ? f2key(dis) : ~f2key(dis);
                s_it[r] = ((unsigned long long)key << 32) | (unsigned)r;
            }
        }
        GH_T(3);
        block_rank_sort_sm(s_it, R, s_hist);
        GH_T(4);
        if (exact_ties) {   // two of the first k + 1 exact distances equal (k_rerank_topk)
            bool eq = false;
            for (int i = tid; i < k && i + 1 < R; i += SM_NT) {
                const uint32_t ka = (uint32_t)(s_it[i] >> 32), kb = (uint32_t)(s_it[i + 1] >> 32);
                eq |= ka == kb && ka != (L2 ? f2key(sentinel) : ~f2key(sentinel));
            }
            tie |= __syncthreads_or(eq ? 1 : 0);
        }
        for (int i = tid; i < k; i += SM_NT) {
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
        GH_T(5);
    } else {          // k_finalize_norank
        int running = 0;
        for (int r0 = 0; r0 < R && running < k; r0 += SM_NT) {
            const int r = r0 + tid;
            float dis = 0.f;
            int64_t id = -1;
            if (r < R) {
                dis = s_val[r];
                id = s_id[r];
            }
            const int flag = (id != -1 && dis <= max_score && dis >= min_score) ? 1 : 0;
            int tot;
            const int ex = block_excl_scan_sm(flag, s_w, tot);
            const int slot = running + ex;
            bool eq = false;
            if (flag && slot < k) {
                distances[(int64_t)q * k + slot] = dis;
                labels[(int64_t)q * k + slot] = id;
                // an entry that is taken and its successor at the same ADC distance (k_finalize_norank)
                eq = r + 1 < R && s_id[r + 1] != -1 && s_val[r + 1] == dis;
            }
            if (exact_ties) tie |= __syncthreads_or(eq ? 1 : 0);
            running += tot;
        }
        for (int i = min(running, k) + tid; i < k; i += SM_NT) {
            distances[(int64_t)q * k + i] = neutral;
            labels[(int64_t)q * k + i] = -1;
        }
    }
    if (tie) {   // uniform: the query is redone the way the reference's heaps run it
        if (tie_stats && tid == 0) atomicAdd(tie_stats + 2, 1ull);
        tie_replay_query<L2, SM_NT, 2048>(tr, q, s_tie_lds, nullptr);
    }
}

void launch_small_tail(hipStream_t s, bool l2, const float* slab, int64_t q_stride, const int* q_total, int nq, int R, int P,
                       const int* probe_list, const int* pair_off, const int64_t* list_off, const int64_t* ids,
                       float* cand_dis, int* cand_pos, int64_t* cand_ids, int has_rank, const float* x, int d,
                       const float* raw, int64_t nraw, int k, float min_score, float max_score, float neutral,
                       float* distances, int64_t* labels, int smax, float* pre_val, int* pre_pos, int fixed_n,
                       const TieReplayArgs* tr, unsigned long long* tie_stats) {
    if (nq <= 0) return;
    if (R > 1024) {   // callers gate on this
        launch_refused("launch_small_tail: recall_num > 1024");
        return;
    }
    const int exact_ties = tr != nullptr ? 1 : 0;
    const TieReplayArgs tra = exact_ties ? *tr : TieReplayArgs{};
    const size_t lds = exact_ties ? tie_replay_lds_bytes_(R, k, P, 2048) : 0;
    if (exact_ties) {   // static + dynamic LDS go beyond the default 64 KB for large recall_num
        static std::atomic<uint64_t> done{0};
        if (first_call_on_device(done)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_tail<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_tail<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10);
        }
    }
    if (smax > 0) {
        if (l2)
            hipLaunchKernelGGL((k_small_presel<true>), dim3(nq, smax), dim3(SM_NT), 0, s, slab, q_stride, q_total, R, smax,
                               pre_val, pre_pos, fixed_n, exact_ties ? pre_pos + (int64_t)nq * smax * R : nullptr);
        else
            hipLaunchKernelGGL((k_small_presel<false>), dim3(nq, smax), dim3(SM_NT), 0, s, slab, q_stride, q_total, R, smax,
                               pre_val, pre_pos, fixed_n, exact_ties ? pre_pos + (int64_t)nq * smax * R : nullptr);
    }
    static unsigned long long* dbg = nullptr;
    static int shown = 0;
    if (getenv("GAMMA_HIP_SM_DBG")) {
        if (!dbg) (void)hipMalloc((void**)&dbg, 64);
        if (shown++ % 10 == 9) {
            unsigned long long h[8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "small tail (10 ns ticks): select %llu map %llu rerank %llu sort %llu out %llu (n %llu, taken %llu)\n",
                    h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6], h[7]);
        }
    }
    if (l2)
        hipLaunchKernelGGL((k_small_tail<true>), dim3(nq), dim3(SM_NT), lds, s, slab, q_stride, q_total, R, P, probe_list,
                           pair_off, list_off, ids, cand_dis, cand_pos, cand_ids, has_rank, x, d, raw, nraw, k, min_score,
                           max_score, neutral, distances, labels, pre_val, pre_pos, smax, fixed_n, dbg, exact_ties, tra,
                           tie_stats);
    else
        hipLaunchKernelGGL((k_small_tail<false>), dim3(nq), dim3(SM_NT), lds, s, slab, q_stride, q_total, R, P, probe_list,
                           pair_off, list_off, ids, cand_dis, cand_pos, cand_ids, has_rank, x, d, raw, nraw, k, min_score,
                           max_score, neutral, distances, labels, pre_val, pre_pos, smax, fixed_n, dbg, exact_ties, tra,
                           tie_stats);
}

}  // namespace gh
