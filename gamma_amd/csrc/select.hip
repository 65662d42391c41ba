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
#include <stdint.h>

#include "block_utils.h"
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
                                                 int* __restrict__ out_pos) {
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
                int cpad = 2;
                while (cpad < cnt) cpad <<= 1;
                bitonic_sort_lds(s_cand, cpad);
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
    bitonic_sort_lds(s_items, Kpad);
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

int select_kpad(int K) {
    int p = 2;
    while (p < K) p <<= 1;
    return p;
}

template <bool SMALLEST>
static void launch_sel(hipStream_t s, const float* vals, int64_t seg_stride, const int* seg_len,
                       int fixed_len, int max_len, int nseg, int K, float* out_vals, int* out_pos) {
    const int Kpad = select_kpad(K);
    const size_t lds = (size_t)(Kpad + CAP) * sizeof(unsigned long long);
#define GH_SEL(NPT)                                                                              \
    hipLaunchKernelGGL((k_select2<SMALLEST, NPT>), dim3(nseg), dim3(256), lds, s, vals, seg_stride, \
                       seg_len, fixed_len, K, Kpad, out_vals, out_pos)
    if (max_len <= 256 * 4) GH_SEL(4);
    else if (max_len <= 256 * 16) GH_SEL(16);
    else GH_SEL(0);
#undef GH_SEL
}

void launch_select_topk(hipStream_t s, bool smallest, const float* vals, int64_t seg_stride,
                        const int* seg_len, int fixed_len, int max_len, int nseg, int K,
                        float* out_vals, int* out_pos) {
    if (nseg <= 0 || K <= 0) return;
    if (smallest)
        launch_sel<true>(s, vals, seg_stride, seg_len, fixed_len, max_len, nseg, K, out_vals, out_pos);
    else
        launch_sel<false>(s, vals, seg_stride, seg_len, fixed_len, max_len, nseg, K, out_vals, out_pos);
}

}  // namespace gh
