// ivfflat.hip -- IVFFLAT list scan, list-major (GammaIVFFlatScanner1::scan_codes, index/impl/gamma_index_ivfflat.h:52-75).
//
// k_ivfflat_scan (kernels.hip) takes one (query, probe) pair per workgroup and gathers the list's rows for that one
// query: every row of a list is fetched once per query that probes the list (32 times at 4096 queries, nlist 4096,
// nprobe 32).  Here the work is turned around: one workgroup per LIST keeps 128 of its rows in registers (two
// threads per row, as k_pairwise_lds) and runs every query that probes the list past them, 32 queries per LDS tile.
// The (query, probe) pairs of a list come from an inverse index built per call (count / scan / fill: a counting
// sort of the coarse assignment by list).  Distances are the exact fvec_L2sqr / fvec_inner_product of the
// reference (eight AVX lane accumulators: four per thread of the pair, packed fp32), written to the same slab
// positions as the pair kernel, so everything downstream is unchanged.  d in {16, 32, 64, 96, 128}.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_math.h"
#include "filter_dev.h"
#include "kernels.h"

namespace gh {

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int FL_QT = 32;   // queries per LDS tile
constexpr int FL_UT = 1;    // tiles per work unit (4: no gain at 16384 queries, -12 % at 4096: fewer, longer units)
constexpr int FL_UQ = FL_QT * FL_UT;
}  // namespace

__global__ __launch_bounds__(256) void k_inv_count(const int* __restrict__ probe, int npairs, int nlist,
                                                   const int* __restrict__ list_len, int* __restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const int l = probe[i];
    if (l >= 0 && l < nlist && list_len[l] > 0) atomicAdd(&cnt[l], 1);
}
// start[l] = exclusive prefix of cnt; cur[l] = 0; ustart[l] = exclusive prefix of the list's work units
// (query tiles x 128-row chunks), ustart[nlist] = their total.  One block.
__global__ __launch_bounds__(1024) void k_inv_scan(const int* __restrict__ cnt, const int* __restrict__ list_len, int nlist,
                                                   int* __restrict__ start, int* __restrict__ cur,
                                                   int* __restrict__ ustart) {
    __shared__ int s_w[2][16];
    __shared__ int s_run[2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 2) s_run[tid] = 0;
    __syncthreads();
    for (int b = 0; b < nlist; b += 1024) {
        const int i = b + tid;
        const int v = i < nlist ? cnt[i] : 0;
        const int u = (i < nlist && v > 0) ? ((v + FL_UQ - 1) / FL_UQ) * ((list_len[i] + 127) / 128) : 0;
        int incl = v, uincl = u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64), tu = __shfl_up(uincl, o, 64);
            if (lane >= o) {
                incl += t;
                uincl += tu;
            }
        }
        if (lane == 63) {
            s_w[0][w] = incl;
            s_w[1][w] = uincl;
        }
        __syncthreads();
        int basew = s_run[0], baseu = s_run[1];
        for (int k = 0; k < w; k++) {
            basew += s_w[0][k];
            baseu += s_w[1][k];
        }
        if (i < nlist) {
            start[i] = basew + incl - v;
            cur[i] = 0;
            ustart[i] = baseu + uincl - u;
        }
        __syncthreads();
        if (tid == 1023) {
            s_run[0] = basew + incl;
            s_run[1] = baseu + uincl;
        }
        __syncthreads();
    }
    if (tid == 0) ustart[nlist] = s_run[1];
}
__global__ __launch_bounds__(256) void k_inv_fill(const int* __restrict__ probe, int npairs, int nlist,
                                                  const int* __restrict__ list_len, const int* __restrict__ start,
                                                  int* __restrict__ cur, int* __restrict__ inv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const int l = probe[i];
    if (l >= 0 && l < nlist && list_len[l] > 0) inv[start[l] + atomicAdd(&cur[l], 1)] = i;
}

template <bool L2, int D>
__global__ __launch_bounds__(256) void k_ivfflat_lm(const float* __restrict__ x, int P, const int* __restrict__ pair_off,
                                                    const int64_t* __restrict__ list_off,
                                                    const int* __restrict__ list_len, const int64_t* __restrict__ ids,
                                                    const float* __restrict__ raw, int64_t nraw,
                                                    const int* __restrict__ inv_start, const int* __restrict__ inv_cnt,
                                                    const int* __restrict__ inv, const int* __restrict__ ustart,
                                                    int nlist, int64_t q_stride,
                                                    float* __restrict__ out, const FilterDesc* __restrict__ ftab,
                                                    int need_filter, float min_score, float max_score) {
    __shared__ float4 s_x[FL_QT * D / 4];
    __shared__ int s_q[FL_QT];
    __shared__ int s_off[FL_QT];
    const int tid = threadIdx.x, half = tid & 1;
    const float sentinel = L2 ? INFINITY : -INFINITY;
    const int nunits = ustart[nlist];
    // a work unit = (list, 128-row chunk, FL_UT tiles of 32 of the queries probing the list): popular and long lists are cut
    // into many units, so the launch is balanced whatever the data looks like
    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
        int lo = 0, hi = nlist - 1;
        while (lo < hi) {   // last list with ustart[l] <= u
            const int mid = (lo + hi + 1) >> 1;
            if (ustart[mid] <= u) lo = mid; else hi = mid - 1;
        }
        const int l = lo;
        const int len = list_len[l], n = inv_cnt[l];
        const int chunks = (len + 127) / 128, rel = u - ustart[l];
        const int tq0 = (rel / chunks) * FL_UQ, tq1 = min(n, tq0 + FL_UQ), r0 = (rel - (rel / chunks) * chunks) * 128;
        const int64_t base = list_off[l];
        const int st = inv_start[l];
        {
        const int j = r0 + (tid >> 1);
        const bool inrow = j < len;
        int64_t id = -1;
        if (inrow) id = ids[base + j];
        const int64_t vid = id & 0x7fffffffffffffffLL;
        bool live = inrow && id >= 0 && vid < nraw;   // id < 0: bit 63, superseded by an Update
        if (need_filter && live) live = is_valid_doc(ftab[0], vid);
        f32x2 yr[D / 4];
        {
            const float4* yp = reinterpret_cast<const float4*>(raw + (live ? vid : 0) * D) + half;
#pragma unroll
            for (int i = 0; i < D / 8; i++) {
                const float4 v = yp[2 * i];
                yr[2 * i] = f32x2{v.x, v.y};
                yr[2 * i + 1] = f32x2{v.z, v.w};
            }
        }
        for (int t0 = tq0; t0 < tq1; t0 += FL_QT) {
            const int nqt = min(FL_QT, tq1 - t0);
            __syncthreads();   // the previous tile has been consumed
            if (tid < nqt) {
                const int pr = inv[st + t0 + tid];
                const int q = pr / P;
                s_q[tid] = q;
                s_off[tid] = pair_off[(int64_t)q * (P + 1) + (pr - q * P)];
            }
            __syncthreads();
            for (int e = tid; e < nqt * (D / 4); e += 256) {
                const int qi = e / (D / 4), c = e - qi * (D / 4);
                s_x[e] = reinterpret_cast<const float4*>(x + (int64_t)s_q[qi] * D)[c];
            }
            __syncthreads();
            for (int qi = 0; qi < nqt; qi++) {
                const float4* xq = s_x + qi * (D / 4) + half;
                f32x2 acc[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
                for (int i0 = 0; i0 < D / 8; i0 += 8) {
                    float4 xa[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) xa[u] = (i0 + u) < D / 8 ? xq[2 * (i0 + u)] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int i = i0 + u;
                        if (i < D / 8) {
                            const f32x2 xv[2] = {f32x2{xa[u].x, xa[u].y}, f32x2{xa[u].z, xa[u].w}};
#pragma unroll
                            for (int k = 0; k < 2; k++) {
                                if (L2) {
                                    const f32x2 t = xv[k] - yr[2 * i + k];
                                    acc[k] = __builtin_elementwise_fma(t, t, acc[k]);
                                } else {
                                    acc[k] = __builtin_elementwise_fma(xv[k], yr[2 * i + k], acc[k]);
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // even thread: AVX lanes 0..3, odd thread: lanes 4..7;  s[l] = acc[l+4] + acc[l]
                float s0, s1, s2, s3;
                add_xor1_x4(acc[0].x, acc[0].y, acc[1].x, acc[1].y, s0, s1, s2, s3);
                float dis = hsum4(s0, s1, s2, s3);
                if (!live || !(dis <= max_score && dis >= min_score)) dis = sentinel;
                if (inrow && half == 0) out[(int64_t)s_q[qi] * q_stride + s_off[qi] + j] = dis;
            }
        }
        }
    }
}

bool ivfflat_lm_supported(int d) { return d == 16 || d == 32 || d == 64 || d == 96 || d == 128; }

// scratch: inv [nq * P] ints, then cnt | start | cur [nlist] ints each
size_t ivfflat_lm_scratch_bytes(int nq, int P, int nlist) { return ((size_t)nq * P + 4 * (size_t)nlist + 1) * sizeof(int); }

void launch_ivfflat_lm(hipStream_t s, bool l2, const float* x, int nq, int d, int P, const int* probe, const int* pair_off,
                       const int64_t* list_off, const int* list_len, int nlist, const int64_t* ids, const float* raw,
                       int64_t nraw, int64_t q_stride, float* out, const FilterDesc* ftab, int need_filter, float min_score,
                       float max_score, void* scratch) {
    if (nq <= 0 || P <= 0) return;
    int* inv = static_cast<int*>(scratch);
    int* cnt = inv + (size_t)nq * P;
    int* start = cnt + nlist;
    int* cur = start + nlist;
    int* ustart = cur + nlist;
    const int npairs = nq * P;
    (void)hipMemsetAsync(cnt, 0, (size_t)nlist * sizeof(int), s);
    hipLaunchKernelGGL(k_inv_count, dim3((npairs + 255) / 256), dim3(256), 0, s, probe, npairs, nlist, list_len, cnt);
    hipLaunchKernelGGL(k_inv_scan, dim3(1), dim3(1024), 0, s, cnt, list_len, nlist, start, cur, ustart);
    hipLaunchKernelGGL(k_inv_fill, dim3((npairs + 255) / 256), dim3(256), 0, s, probe, npairs, nlist, list_len, start, cur, inv);
    const int grid = 256 * 8;   // persistent: every workgroup walks the unit list with stride gridDim.x
#define GH_FL(LL, DD)                                                                                              \
    hipLaunchKernelGGL((k_ivfflat_lm<LL, DD>), dim3(grid), dim3(256), 0, s, x, P, pair_off, list_off, list_len, ids, \
                       raw, nraw, start, cnt, inv, ustart, nlist, q_stride, out, ftab, need_filter, min_score, max_score)
#define GH_FLD(LL)                    \
    switch (d) {                      \
        case 128: GH_FL(LL, 128); break; \
        case 96: GH_FL(LL, 96); break;   \
        case 64: GH_FL(LL, 64); break;   \
        case 32: GH_FL(LL, 32); break;   \
        default: GH_FL(LL, 16); break;   \
    }
    if (l2) { GH_FLD(true) } else { GH_FLD(false) }
#undef GH_FLD
#undef GH_FL
}

}  // namespace gh
