// store_kernels.hip -- kernels of the writers and the trainer: shard gathers, bitmap bits, superseded marks, arena repack,
// per-code table sums of the filter pass, centroid sums of k-means, PQ encode.
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
// small utility kernels
// ------------------------------------------------------------------------------------
__global__ void k_pos_to_i32(const int* __restrict__ pos, int* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = pos[i];
}

// [shard][nq][R] -> [nq][shard*R] for the sharded merge
__global__ __launch_bounds__(256) void k_gather_shards(const float* __restrict__ all_dis,
                                                       const int64_t* __restrict__ all_ids,
                                                       int nshards, int nq, int R,
                                                       float* __restrict__ dis,
                                                       int64_t* __restrict__ ids, float sentinel) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)nshards * nq * R;
    if (i >= tot) return;
    const int r = (int)(i % R);
    const int64_t t = i / R;
    const int q = (int)(t % nq);
    const int sh = (int)(t / nq);
    const int64_t id = all_ids[i];
    const int64_t o = ((int64_t)q * nshards + sh) * R + r;
    dis[o] = id < 0 ? sentinel : all_dis[i];
    ids[o] = id;
}
void launch_gather_shards(hipStream_t s, const float* all_dis, const int64_t* all_ids, int nshards,
                          int nq, int R, float* dis, int64_t* ids, float sentinel) {
    int64_t tot = (int64_t)nshards * nq * R;
    if (tot <= 0) return;
    hipLaunchKernelGGL(k_gather_shards, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, all_dis,
                       all_ids, nshards, nq, R, dis, ids, sentinel);
}

// out_ids[q][r] = pos<0 ? -1 : src_ids[q][pos]
__global__ __launch_bounds__(256) void k_take_ids(const int* __restrict__ pos,
                                                  const int64_t* __restrict__ src_ids,
                                                  int64_t src_stride, int R, int64_t n,
                                                  int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t q = i / R;
    const int ps = pos[i];
    out[i] = ps < 0 ? -1 : src_ids[q * src_stride + ps];
}
void launch_take_ids(hipStream_t s, const int* pos, const int64_t* src_ids, int64_t src_stride,
                     int nq, int R, int64_t* out) {
    int64_t n = (int64_t)nq * R;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_take_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, src_ids,
                       src_stride, R, n, out);
}

// coarse result packing: selected positions are the centroid ids
__global__ void k_i32_copy_check(const int* __restrict__ in, int* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// set / clear bits of the delete bitmap
__global__ void k_bitmap_set(uint8_t* __restrict__ bm, const int64_t* __restrict__ docids, int64_t n,
                             int64_t nbits, int value) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t id = docids[i];
    if (id < 0 || id >= nbits) return;
    unsigned int* w = reinterpret_cast<unsigned int*>(bm) + (id >> 5);
    unsigned int m = 1u << (id & 31);  // little-endian: bit (id&7) of byte id>>3
    if (value) atomicOr(w, m); else atomicAnd(w, ~m);
}
void launch_bitmap_set(hipStream_t s, uint8_t* bm, const int64_t* docids, int64_t n, int64_t nbits,
                       int value) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_bitmap_set, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bm, docids, n,
                       nbits, value);
}

// mark an inverted-list entry as superseded (ids[pos] |= kDelIdxMask)
__global__ void k_mark_moved(int64_t* __restrict__ ids, int64_t pos) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ids[pos] |= (int64_t)(1ULL << 63);
}
void launch_mark_moved(hipStream_t s, int64_t* ids, int64_t pos) {
    hipLaunchKernelGGL(k_mark_moved, dim3(1), dim3(64), 0, s, ids, pos);
}

// Arena repack (gamma_hip_store.cpp, arena_repack): every list's live entries move from (old arrays, old offset)
// to (new arrays, new offset).  grid = (nlist, chunks); the code bytes move as dwords when M % 4 == 0.
__global__ __launch_bounds__(256) void k_repack_lists(const uint8_t* __restrict__ oc, const int64_t* __restrict__ oi,
                                                      uint8_t* __restrict__ nc, int64_t* __restrict__ ni,
                                                      const int64_t* __restrict__ old_off,
                                                      const int64_t* __restrict__ new_off,
                                                      const int* __restrict__ len, int M) {
    const int l = blockIdx.x;
    const int n = len[l];
    const int64_t a = old_off[l], b = new_off[l];
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) ni[b + i] = oi[a + i];
    if ((M & 3) == 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(oc + a * M);
        uint32_t* dst = reinterpret_cast<uint32_t*>(nc + b * M);
        const int64_t nw = (int64_t)n * (M >> 2);
        for (int64_t i = blockIdx.y * 256 + threadIdx.x; i < nw; i += gridDim.y * 256) dst[i] = src[i];
    } else {
        const int64_t nb = (int64_t)n * M;
        for (int64_t i = blockIdx.y * 256 + threadIdx.x; i < nb; i += gridDim.y * 256) nc[b * M + i] = oc[a * M + i];
    }
}
void launch_repack_lists(hipStream_t s, const uint8_t* oc, const int64_t* oi, uint8_t* nc, int64_t* ni,
                         const int64_t* old_off, const int64_t* new_off, const int* len, int nlist, int M,
                         int max_len) {
    if (nlist <= 0) return;
    const int chunks = std::max(1, std::min(64, (max_len + 1023) / 1024));
    hipLaunchKernelGGL(k_repack_lists, dim3(nlist, chunks), dim3(256), 0, s, oc, oi, nc, ni, old_off, new_off, len, M);
}

// Per-list checksum of (ids, codes) at given extents: what arena_repack compares between the source and -- read back
// through the NEW mapping, in a launch of its own behind a translation fence -- the target before the new version of the
// list tables is published (realtime/realtime_mem_data.cc:426-474 swaps a bucket's pointer only after the copy).
// sum over entries of mix(position in the list, word): a moved, dropped, zeroed or permuted entry changes it.
__device__ __forceinline__ unsigned long long chk_mix(unsigned long long x) {   // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_list_checksum(const uint8_t* __restrict__ codes, const int64_t* __restrict__ ids,
                                                       const int64_t* __restrict__ off, const int* __restrict__ len, int M,
                                                       unsigned long long* __restrict__ out) {
    const int l = blockIdx.x;
    const int n = len[l];
    const int64_t a = off[l];
    unsigned long long acc = 0;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256)
        acc += chk_mix(((unsigned long long)i << 1) ^ chk_mix((unsigned long long)ids[a + i]));
    const int64_t nb = (int64_t)n * M;
    const uint8_t* c = codes + a * M;
    for (int64_t i = blockIdx.y * 256 + threadIdx.x; i < nb; i += gridDim.y * 256)
        acc += chk_mix((((unsigned long long)i << 9) | 0x100ull | c[i]) * 0x2545f4914f6cdd1dull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out + l, acc);
}
void launch_list_checksum(hipStream_t s, const uint8_t* codes, const int64_t* ids, const int64_t* off, const int* len, int nlist,
                          int M, int max_len, unsigned long long* out) {
    if (nlist <= 0) return;
    (void)hipMemsetAsync(out, 0, (size_t)nlist * sizeof(unsigned long long), s);
    const int chunks = std::max(1, std::min(64, (max_len + 1023) / 1024));
    hipLaunchKernelGGL(k_list_checksum, dim3(nlist, chunks), dim3(256), 0, s, codes, ids, off, len, M, out);
}

// ------------------------------------------------------------------------------------
// Per-code table sums of the L2 scan's filter pass (k_ivfpq_scan_pair<.., CF>): sums[pos] = sum_m T2[list][m][code[m]]
// (sequential fp32 adds from 0).  The value is only ever used inside a bound with a margin that covers its rounding,
// so any fixed order would do.
//   ranges: range r = n[r] entries of list list_no[r] from arena position pos[r]  (grid = (ranges, chunks))
//   lists : every entry of every list at its current extent                      (grid = (nlist, chunks))
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void code_sums_span(const float* __restrict__ t2l, const uint8_t* __restrict__ codes, int M,
                                               int64_t pos, int n, float* __restrict__ sums) {
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
        const uint8_t* c = codes + (pos + i) * M;
        float acc = 0.f;
        for (int m = 0; m < M; m++) acc += t2l[m * 256 + c[m]];
        sums[pos + i] = acc;
    }
}
__global__ __launch_bounds__(256) void k_code_sums_ranges(const float* __restrict__ T2, const uint8_t* __restrict__ codes, int M,
                                                          const int* __restrict__ list_no, const int64_t* __restrict__ pos,
                                                          const int* __restrict__ n, float* __restrict__ sums) {
    const int r = blockIdx.x;
    code_sums_span(T2 + (int64_t)list_no[r] * M * 256, codes, M, pos[r], n[r], sums);
}
__global__ __launch_bounds__(256) void k_code_sums_lists(const float* __restrict__ T2, const uint8_t* __restrict__ codes, int M,
                                                         const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
                                                         float* __restrict__ sums) {
    const int l = blockIdx.x;
    code_sums_span(T2 + (int64_t)l * M * 256, codes, M, list_off[l], list_len[l], sums);
}
__global__ __launch_bounds__(256) void k_code_sums_one(const float* __restrict__ t2l, const uint8_t* __restrict__ codes, int M,
                                                       int64_t pos, int n, float* __restrict__ sums) {
    code_sums_span(t2l, codes, M, pos, n, sums);
}
void launch_code_sums_one(hipStream_t s, const float* T2, const uint8_t* codes, int M, int list_no, int64_t pos, int n,
                          float* sums) {
    if (n <= 0) return;
    const int chunks = std::max(1, std::min(64, (n + 1023) / 1024));
    hipLaunchKernelGGL(k_code_sums_one, dim3(1, chunks), dim3(256), 0, s, T2 + (int64_t)list_no * M * 256, codes, M, pos, n, sums);
}
void launch_code_sums_ranges(hipStream_t s, const float* T2, const uint8_t* codes, int M, const int* list_no, const int64_t* pos,
                             const int* n, int nranges, int max_n, float* sums) {
    if (nranges <= 0) return;
    const int chunks = std::max(1, std::min(64, (max_n + 1023) / 1024));
    hipLaunchKernelGGL(k_code_sums_ranges, dim3(nranges, chunks), dim3(256), 0, s, T2, codes, M, list_no, pos, n, sums);
}
void launch_code_sums_lists(hipStream_t s, const float* T2, const uint8_t* codes, int M, const int64_t* list_off,
                            const int* list_len, int nlist, int max_len, float* sums) {
    if (nlist <= 0) return;
    const int chunks = std::max(1, std::min(64, (max_len + 1023) / 1024));
    hipLaunchKernelGGL(k_code_sums_lists, dim3(nlist, chunks), dim3(256), 0, s, T2, codes, M, list_off, list_len, sums);
}
// t2max[l] = sum_m max_c |T2[l][m][c]|: bounds every partial sum of a code's table entries (the filter's margin)
__global__ __launch_bounds__(256) void k_t2_rowmax(const float* __restrict__ T2, int M, float* __restrict__ t2max) {
    __shared__ float s_w[4];
    const int l = blockIdx.x, tid = threadIdx.x;
    const float* t = T2 + (int64_t)l * M * 256;
    float tot = 0.f;
    for (int m = 0; m < M; m++) {
        float v = fabsf(t[m * 256 + tid]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
        __syncthreads();
        if ((tid & 63) == 0) s_w[tid >> 6] = v;
        __syncthreads();
        tot += fmaxf(fmaxf(s_w[0], s_w[1]), fmaxf(s_w[2], s_w[3]));
    }
    if (tid == 0) t2max[l] = tot;
}
void launch_t2_rowmax(hipStream_t s, const float* T2, int nlist, int M, float* t2max) {
    if (nlist > 0) hipLaunchKernelGGL(k_t2_rowmax, dim3(nlist), dim3(256), 0, s, T2, M, t2max);
}

// ------------------------------------------------------------------------------------
// k-means update (gamma_hip_train.cpp; compute_centroids, faiss:Clustering.cpp:138-208): cluster c = the points
// order[seg[c] .. seg[c + 1]) in ascending point order; one float accumulator per (cluster, dimension) adds them in that
// order, then c[j] *= 1 / count.  An empty cluster's centroid is zero (the host re-seeds it, split_clusters).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_centroid_update(const float* __restrict__ x, int d, const int* __restrict__ order,
                                                         const int* __restrict__ seg, float* __restrict__ centroids,
                                                         float* __restrict__ hassign) {
    const int c = blockIdx.x;
    const int b = seg[c], e = seg[c + 1];
    const float cnt = (float)(e - b);
    for (int j = threadIdx.x; j < d; j += 128) {
        float acc = 0.f;
        for (int m = b; m < e; m++) acc += x[(int64_t)order[m] * d + j];
        if (e > b) {
            const float norm = 1 / cnt;
            acc *= norm;
        }
        centroids[(int64_t)c * d + j] = acc;
    }
    if (threadIdx.x == 0) hassign[c] = cnt;
}
void launch_centroid_update(hipStream_t s, const float* x, int d, const int* order, const int* seg, int k, float* centroids,
                            float* hassign) {
    if (k > 0) hipLaunchKernelGGL(k_centroid_update, dim3(k), dim3(128), 0, s, x, d, order, seg, centroids, hassign);
}

// ------------------------------------------------------------------------------------
// a12 (Add path): residual + PQ encode.  assign comes from the coarse kernels + select.
//   code[m] = argmin_j fvec_L2sqr_ny(residual_m, c_mj)   (strict <, first minimum,
//   faiss:impl/ProductQuantizer.cpp:321-348).  grid = (M, n), block = 256 = ksub.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pq_encode(const float* __restrict__ x, int d, int M, int dsub,
                                                   const int* __restrict__ assign,
                                                   const float* __restrict__ cc,
                                                   const float* __restrict__ pqc,
                                                   uint8_t* __restrict__ codes) {
    __shared__ float s_res[64];
    __shared__ unsigned long long s_best[4];
    const int m = blockIdx.x, i = blockIdx.y, j = threadIdx.x;
    const int l = assign[i];
    if (j < dsub) {
        float xv = x[(int64_t)i * d + m * dsub + j];
        s_res[j] = l < 0 ? 0.f : xv - cc[(int64_t)l * d + m * dsub + j];
    }
    __syncthreads();
    const float* c = pqc + ((int64_t)m * 256 + j) * dsub;
    float dis = fvec_ny_row<true>(s_res, c, dsub);
    if (!(dis < 1e20f)) dis = INFINITY;  // reference never picks dis >= 1e20 (mindis init)
    // argmin with first-index tie rule: min over (key(dis), j)
    unsigned long long item = ((unsigned long long)f2key(dis) << 32) | (unsigned)j;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(item, off, 64);
        if (o < item) item = o;
    }
    if ((j & 63) == 0) s_best[j >> 6] = item;
    __syncthreads();
    if (j == 0) {
        unsigned long long b = s_best[0];
        for (int w = 1; w < 4; w++) if (s_best[w] < b) b = s_best[w];
        int best = (int)(uint32_t)b;
        if (key2f((uint32_t)(b >> 32)) == INFINITY) best = 0;  // idxm initial value
        codes[(int64_t)i * M + m] = (uint8_t)best;
    }
}
void launch_pq_encode(hipStream_t s, const float* x, int64_t n, int d, int M, const int* assign,
                      const float* cc, const float* pqc, uint8_t* codes) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pq_encode, dim3(M, (unsigned)n), dim3(256), 0, s, x, d, M, d / M, assign, cc,
                       pqc, codes);
}

}  // namespace gh
