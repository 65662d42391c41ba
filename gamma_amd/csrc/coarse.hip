// coarse.hip -- the coarse quantizer of a large batch without the [nq][nlist] distance matrix.
//
// Replaces, for nq >= 4096 and nlist >= 2048, the pair (k_l2_gemmform_strip -> matrix in HBM -> k_select_wave):
// faiss's knn_L2sqr in its BLAS form (faiss:utils/distances.cpp:215-296) followed by HeapResultHandler
// (faiss:impl/ResultHandler.h:112-117) keeps, per query, the nprobe smallest of nlist distances
//      dis(q, c) = max(0, (|x_q|^2 + |y_c|^2) - 2 <x_q, y_c>),   <.,.> one k-ascending fp32 fma chain.
// Writing all of them (268 MB per 16 384 queries at nlist 4096) and reading them back costs more than computing
// them.  Here:
//   A. the distances to a SAMPLE of the centroids (the first 512 columns; 1024 for nprobe > 32) are written out
//      (k_coarse_fused<.., STORE>); k_coarse_bound derives from them an upper bound tau_q of the final nprobe-th
//      smallest;
//   B. k_coarse_fused computes the other columns with the same MFMA chain and keeps only entries <= tau_q
//      (about nprobe * (nlist / sample) of them): per (query, column strip) list in HBM, slots handed out by
//      counters in registers (the 32 rows of a wave belong to it alone: ballot + popcount, no atomics);
//   C. k_coarse_final merges the sample entries <= tau_q and the survivors into the nprobe smallest (distance,
//      centroid);
//   D. a query whose list overflowed (tau loose: adversarial data) is redone from scratch by k_coarse_repair.
// Every distance is produced by the same instruction sequence as k_l2_gemmform_strip, so coarse_dis / coarse_idx
// are bit-identical to the unfused path (tests/test_gpu_more.py::test_fused_coarse_*).
//
// k_coarse_fused: a workgroup (4 waves) owns 128 queries x one strip of 64-centroid tiles.  The query fragments
// of v_mfma_f32_32x32x2_f32 live in REGISTERS for the whole strip (64 VGPRs at d = 128; wave w = rows 32w..32w+31);
// only centroid tiles go through LDS, double-buffered, one barrier per tile.  A wave alternates between the two
// 32-column blocks of a tile; the epilogue of one block is interleaved, instruction by instruction, with the
// MFMAs of the next, so the matrix pipe does not wait for the VALU work.  LDS: 2 x 64 x (d+1) floats = 66 KB at d = 128 -> two workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "device_math.h"
#include "heap_dev.h"
#include "kernels.h"

namespace gh {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int wave_incl_scan_i(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
}  // namespace

// STORE: no filter -- every distance of columns [col0, ny) goes to mat[row * ld + (col - col0)] (the sample columns
// the bound is derived from); tau / cand / cand_cnt unused, cap_stride = ld.
template <int NCH, bool STORE = false>   // d = 16 * NCH
__global__ __launch_bounds__(256, 2) void k_coarse_fused(const float* __restrict__ x, int nq,
                                                         const float* __restrict__ y, int ny, int col0,
                                                         const float* __restrict__ yn,
                                                         const float* __restrict__ tau,
                                                         int tiles_per_strip, int cap, int cap_stride,
                                                         unsigned long long* __restrict__ cand,
                                                         int* __restrict__ cand_cnt, int nseg,
                                                         const int* __restrict__ rowmap) {
    // rowmap (STORE only): rowmap[0] rows to compute, row i of the output = query rowmap[1 + i] (the rows k_coarse_final
    // flagged for the heap replay); a workgroup beyond the count has nothing to do
    constexpr int D = 16 * NCH, LD = D + 1, SEGS = D / 32;   // 32-float segments per row
    extern __shared__ float s_co[];                          // 2 x [64][LD]; first used as the query tile [128][LD]
    __shared__ float s_xn[128];
    __shared__ float2 s_xt[128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int seg = blockIdx.x, q_base = blockIdx.y * 128;
    int nrows = nq;
    if constexpr (STORE) {
        if (rowmap) {
            nrows = min(rowmap[0], nq);
            if (q_base >= nrows) return;   // uniform
        }
    }
    const int ntiles = (ny - col0 + 63) >> 6;
    const int t0 = seg * tiles_per_strip, t1 = min(ntiles, t0 + tiles_per_strip);
    // a wave instruction covers 8 rows x 32 floats (8 lanes per 128-byte row segment, coalesced); with the odd
    // row stride its four scalar LDS stores hit 32 distinct banks per half wave.  combo = w * NCH + it
    auto slot_r = [&](int it) { return (((w * NCH + it) / SEGS) << 3) + (lane >> 3); };
    auto slot_c = [&](int it) { return (((w * NCH + it) % SEGS) << 5) + ((lane & 7) << 2); };
    float4 vb[NCH];
    auto put = [&](float* base, int it) {
        float* p = base + slot_r(it) * LD + slot_c(it);
        p[0] = vb[it].x; p[1] = vb[it].y; p[2] = vb[it].z; p[3] = vb[it].w;
    };
    // ---- the 128 queries: LDS once, then norms and MFMA fragments into registers ----
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
        for (int it = 0; it < NCH; it++) {
            int row = min(q_base + half * 64 + slot_r(it), nrows - 1);
            if constexpr (STORE) {
                if (rowmap) row = rowmap[1 + row];
            }
            vb[it] = *reinterpret_cast<const float4*>(x + (int64_t)row * D + slot_c(it));
        }
#pragma unroll
        for (int it = 0; it < NCH; it++) put(s_co + half * 64 * LD, it);
    }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; half++) {   // fvec_norm_L2sqr order: 4 lanes per row, (a0 + a1) + (a2 + a3)
        const float* row = s_co + (half * 64 + (tid >> 2)) * LD;
        const int l4 = tid & 3;
        float nacc = 0.f;
        for (int i = 0; i < D; i += 4) {
            const float xv = row[i + l4];
            nacc = __builtin_fmaf(xv, xv, nacc);
        }
        const float t01 = nacc + __shfl_down(nacc, 1, 4);
        const float nn = t01 + __shfl_down(t01, 2, 4);
        if (l4 == 0) s_xn[half * 64 + (tid >> 2)] = nn;
    }
    float a[8 * NCH];
    {
        const float* fa = s_co + (w * 32 + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
        for (int u = 0; u < 8 * NCH; u++) a[u] = fa[2 * u];
    }
    __syncthreads();
    // (|x|^2, tau) of the 128 rows stay in LDS: the epilogue reads the pair of its row per element
    if (tid < 128) s_xt[tid] = make_float2(s_xn[tid], (!STORE && q_base + tid < nq) ? tau[q_base + tid] : -INFINITY);
    // ---- centroid tiles ----
    auto gload = [&](int t) {
#pragma unroll
        for (int it = 0; it < NCH; it++)
            vb[it] = *reinterpret_cast<const float4*>(y + (int64_t)min(col0 + t * 64 + slot_r(it), ny - 1) * D + slot_c(it));
    };
    if (t0 < t1) {
        gload(t0);
#pragma unroll
        for (int it = 0; it < NCH; it++) put(s_co, it);
    }
    __syncthreads();
    // slots of a row's list are handed out from a counter every lane of its half wave keeps in a register; the
    // stores go through a buffer descriptor so that a lane with nothing to store points out of range (dropped by
    // the hardware) instead of branching: the whole tile step stays one basic block and the compiler interleaves
    // the epilogue of tile t with the MFMAs of tile t+1
    int cnt[16];
#pragma unroll
    for (int r = 0; r < 16; r++) cnt[r] = 0;
    const uint32_t lt_mask = (1u << (lane & 31)) - 1u;
    // list entries (STORE: matrix elements) between consecutive queries, and the first row of this lane
    const unsigned row_stride = STORE ? (unsigned)cap_stride : (unsigned)nseg * (unsigned)cap_stride;
    const unsigned lane_row = STORE ? (unsigned)(q_base + w * 32 + 4 * (lane >> 5)) * (unsigned)cap_stride
                                    : ((unsigned)((q_base + w * 32 + 4 * (lane >> 5)) * nseg + seg)) * (unsigned)cap_stride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        cand, 0,
        (int)min(STORE ? (int64_t)nq * cap_stride * 4 : (int64_t)nq * nseg * cap_stride * 8, (int64_t)0x7fffffff),
        0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int sh = lane & 32;
    int rs[16];   // byte offset of row slot r's list from the lane's first row: uniform, rides in the store's soffset
#pragma unroll
    for (int r = 0; r < 16; r++) rs[r] = ((r & 3) + 8 * (r >> 2)) * (int)row_stride * (STORE ? 4 : 8);
    const float2* xt_row = s_xt + w * 32 + 4 * (lane >> 5);
    // One element of the epilogue: row slot r of a 32-column block, accumulators o.  f32 MFMAs run at the vector
    // FP32 rate and do not overlap with VALU work of the same SIMD (measured: the epilogue's cycles add to the
    // MFMAs'), so every instruction here counts: 13 VALU per element.
    //   (xn + yn) - 2 ip == fma(-2, ip, xn + yn): the product is exact.  A column beyond ny carries yn = +inf and
    //   fails the test by itself.  The clamp only matters for the stored key: max(bits, 0) | sign bit is the key
    //   of max(dis, +0) (dis is never -0: xn + yn >= +0).  A slot beyond the capacity collapses onto the last
    //   one of the row's list: that query's count says "overflowed" and its list is never read.
    auto epi = [&](const f32x16& o, int r, unsigned colv, float ync, int rsoff) {
        const float2 xt = xt_row[(r & 3) + 8 * (r >> 2)];
        if constexpr (STORE) {
            float dis = __builtin_fmaf(-2.f, o[r], xt.x + ync);
            if (dis < 0.f) dis = 0.f;
            // a row beyond nq falls outside the descriptor, a column beyond ny is steered there
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dis), rsrc,
                                                  (int)colv < ny ? (lane_row + (colv - (unsigned)col0)) * 4u : OOB, rsoff, 0);
            return;
        }
        const float dis = __builtin_fmaf(-2.f, o[r], xt.x + ync);
        const bool pass = dis <= xt.y;
        const unsigned long long mask = __ballot(pass);
        const uint32_t mh = (uint32_t)(mask >> sh);
        const unsigned slot = min((unsigned)cnt[r] + (unsigned)__popc(mh & lt_mask), (unsigned)cap);
        u32x2 item;
        item.x = colv;
        item.y = (uint32_t)max((int)__float_as_uint(dis), 0) | 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(item, rsrc, pass ? (lane_row + slot) * 8u : OOB, rsoff, 0);
        cnt[r] += __popc(mh);   // the same in the 32 lanes that share the row
    };
    // One unit of work = one 32-column block: its 8 * NCH MFMAs (a single dependent chain: issue interval and
    // dependent latency of v_mfma_f32_32x32x2_f32 are both 64 cycles) into n, with the 16 epilogue elements of the
    // PREVIOUS block (accumulators o, first column colo) spread between them.
    auto step = [&](auto do_mfma, auto do_epi, f32x16& n, const float* fb, const f32x16& o, int colo) {
        constexpr bool MF = decltype(do_mfma)::value, EP = decltype(do_epi)::value;
        float ync = 0.f;
        const unsigned colv = (unsigned)(colo + (lane & 31));
        if constexpr (EP) ync = (int)colv < ny ? yn[min((int)colv, ny - 1)] : INFINITY;
        if constexpr (MF) {
#pragma unroll
            for (int i = 0; i < 16; i++) n[i] = 0.f;
            float f[4];   // fragment ring: k-pair u + 3 is read while u feeds the matrix pipe
#pragma unroll
            for (int u = 0; u < 3; u++) f[u] = fb[2 * u];
            constexpr int KP = 8 * NCH, EVERY = KP / 16;   // k-pairs; one epilogue element per EVERY of them
#pragma unroll
            for (int u = 0; u < KP; u++) {
                if (u + 3 < KP) f[(u + 3) & 3] = fb[2 * (u + 3)];
                n = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], f[u & 3], n, 0, 0, 0);
                if constexpr (EP) {
                    if (u % EVERY == 0) epi(o, u / EVERY, colv, ync, rs[u / EVERY]);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the interleaving as written (and the live ranges short)
            }
        } else if constexpr (EP) {
#pragma unroll
            for (int r = 0; r < 16; r++) epi(o, r, colv, ync, rs[r]);
        }
    };
    const std::true_type yes;
    const std::false_type no;
    f32x16 acc0, acc1;   // column blocks 0 and 1 of the current tile
    auto stage = [&](int t) {   // registers -> the LDS buffer of tile t
#pragma unroll
        for (int it = 0; it < NCH; it++) put(s_co + ((t - t0) & 1) * 64 * LD, it);
    };
    for (int t = t0; t < t1; t++) {
        // here: tile t is complete in its LDS buffer; acc1 = block 1 of tile t-1 (if t > t0), epilogue pending
        const float* fb0 = s_co + ((t - t0) & 1) * 64 * LD + (lane & 31) * LD + (lane >> 5);
        const int colb = col0 + t * 64;
        if (t > t0) step(yes, yes, acc0, fb0, acc1, colb - 32);
        else step(yes, no, acc0, fb0, acc1, 0);
        const bool more = t + 1 < t1;   // uniform
        if (more) gload(t + 1);
        step(yes, yes, acc1, fb0 + 32 * LD, acc0, colb);
        if (more) stage(t + 1);   // the buffer of tile t-1: every wave passed the last barrier after reading it
        __syncthreads();
    }
    if (t0 < t1) step(no, yes, acc0, s_co, acc1, col0 + (t1 - 1) * 64 + 32);
    if (!STORE && (lane & 31) == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = q_base + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < nq) cand_cnt[row * nseg + seg] = cnt[r];
        }
    }
}


// K-th smallest (with multiplicity) of the 64 lane values: the value v with #(m < v) < K <= #(m <= v)
__device__ __forceinline__ uint32_t wave_kth_smallest(uint32_t m, int K) {
    int lt = 0, le = 0;
#pragma unroll
    for (int l = 0; l < 64; l++) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)m, l);
        lt += o < m ? 1 : 0;
        le += o <= m ? 1 : 0;
    }
    const unsigned long long who = __ballot(lt < K && K <= le);
    return (uint32_t)__shfl((int)m, (int)__ffsll((long long)who) - 1, 64);
}

// K-th smallest of the 128 values (m0, m1) of the 64 lanes, K <= 128
__device__ __forceinline__ uint32_t wave_kth_smallest2(uint32_t m0, uint32_t m1, int K) {
    int lt0 = 0, le0 = 0, lt1 = 0, le1 = 0;
#pragma unroll
    for (int l = 0; l < 64; l++) {
        const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)m0, l);
        const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)m1, l);
        lt0 += (o0 < m0 ? 1 : 0) + (o1 < m0 ? 1 : 0);
        le0 += (o0 <= m0 ? 1 : 0) + (o1 <= m0 ? 1 : 0);
        lt1 += (o0 < m1 ? 1 : 0) + (o1 < m1 ? 1 : 0);
        le1 += (o0 <= m1 ? 1 : 0) + (o1 <= m1 ? 1 : 0);
    }
    const unsigned long long who0 = __ballot(lt0 < K && K <= le0), who1 = __ballot(lt1 < K && K <= le1);
    return who0 ? (uint32_t)__shfl((int)m0, (int)__ffsll((long long)who0) - 1, 64)
                : (uint32_t)__shfl((int)m1, (int)__ffsll((long long)who1) - 1, 64);
}

// One wave per query: tau_q = the P-th smallest of the 64 lane minima of the query's sample row.  P lanes hold a
// value <= tau_q, so tau_q bounds the P-th smallest distance of the row from above, and with 8 values per lane it
// sits at the same quantile (about 1.3 P / sample) a full selection would reach -- at a tenth of its cost.
template <int NPL>   // sample = 64 * NPL columns
__global__ __launch_bounds__(256) void k_coarse_bound(const float* __restrict__ mat, int nq, int P, float* __restrict__ tau,
                                                      int* __restrict__ ovf) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ovf[0] = 0;   // the overflow list of this call starts empty (k_coarse_final appends)
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float* v = mat + (int64_t)q * (64 * NPL) + lane;
    // P > 32: two minima per lane, over the even and the odd columns of the lane -- the P-th smallest of the 128 sits
    // where the P/2-th of 64 would (the P-th of 64 lane minima is no bound up there: at P = 64 it is the largest)
    uint32_t m0 = 0xffffffffu, m1 = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const uint32_t k = f2key(v[64 * j]);
        if (j & 1) m1 = k < m1 ? k : m1;
        else m0 = k < m0 ? k : m0;
    }
    const uint32_t kb = P <= 32 ? wave_kth_smallest(m0 < m1 ? m0 : m1, P) : wave_kth_smallest2(m0, m1, P);
    if (lane == 0) tau[q] = key2f(kb);
}

// One wave per query: the P smallest (distance, centroid) of { sample entries <= tau } U { survivors of every strip }.
// Everything outside that union is > tau >= the P-th smallest, so the union holds the final top-P.  The bound is
// tightened once more the same way (P-th smallest lane minimum over ~340 entries leaves ~1.3 P); what passes is
// rank-sorted in LDS.  A query with an overflowed strip list, or too many entries at the bound, goes to
// k_coarse_repair.
constexpr int CF_BUF = 256;
template <int SNPL, int MAXSEG>
__global__ __launch_bounds__(256) void k_coarse_final(const float* __restrict__ mat, const float* __restrict__ tau,
                                                      const unsigned long long* __restrict__ cand,
                                                      const int* __restrict__ cand_cnt, int nseg, int cap,
                                                      int cap_stride, int nq, int P, float* __restrict__ out_dis,
                                                      int* __restrict__ out_idx, int* __restrict__ ovf, int flag_ties) {
    constexpr int NPL = SNPL + 2 * MAXSEG;
    __shared__ unsigned long long s_buf[4][CF_BUF];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;   // whole wave; no workgroup barrier below
    unsigned long long* buf = s_buf[w];
    const int cnt = lane < nseg ? cand_cnt[q * nseg + lane] : 0;
    if (__ballot(cnt > cap)) {
        if (lane == 0) ovf[1 + atomicAdd(ovf, 1)] = q;
        return;
    }
    unsigned long long it[NPL];
    {
        const float t = tau[q];
        const float* v = mat + (int64_t)q * (64 * SNPL) + lane;
#pragma unroll
        for (int j = 0; j < SNPL; j++) {
            const float x = v[64 * j];
            it[j] = x <= t ? (((unsigned long long)f2key(x) << 32) | (unsigned)(64 * j + lane)) : ~0ull;
        }
    }
#pragma unroll
    for (int sg = 0; sg < MAXSEG; sg++) {
        const int c = __shfl(cnt, sg, 64);   // 0 beyond nseg
        const unsigned long long* p = cand + ((int64_t)q * nseg + min(sg, nseg - 1)) * cap_stride;
        // entry e of strip sg goes to lane (e + 4 sg) mod 64: the ~40 entries of every strip would otherwise all sit in
        // the low lanes, the high lanes would hold nothing, and a bound taken from lane minima would be no bound
        const int e0 = (lane - 4 * sg) & 63;
        it[SNPL + 2 * sg] = e0 < c ? p[e0] : ~0ull;
        it[SNPL + 2 * sg + 1] = e0 + 64 < c ? p[e0 + 64] : ~0ull;
    }
    uint32_t m0 = 0xffffffffu, m1 = 0xffffffffu;   // empty slots carry the largest key
#pragma unroll
    for (int j = 0; j < NPL; j++) {   // two groups of slots: sample columns by parity, strips by parity
        const uint32_t k = (uint32_t)(it[j] >> 32);
        const bool g1 = j < SNPL ? (j & 1) != 0 : ((j >> 1) & 1) != 0;
        if (g1) m1 = k < m1 ? k : m1;
        else m0 = k < m0 ? k : m0;
    }
    // 0xffffffff when fewer than P lanes (P > 32: lane halves) hold entries: keep all
    const uint32_t kb = P <= 32 ? wave_kth_smallest(m0 < m1 ? m0 : m1, P) : wave_kth_smallest2(m0, m1, P);
    int c = 0;
#pragma unroll
    for (int j = 0; j < NPL; j++) c += ((uint32_t)(it[j] >> 32) <= kb && it[j] != ~0ull) ? 1 : 0;
    const int incl = wave_incl_scan_i(c);
    const int tot = __shfl(incl, 63, 64);
    if (tot > CF_BUF) {   // mass ties at the bound: not worth a second path here
        if (lane == 0) ovf[1 + atomicAdd(ovf, 1)] = q;
        return;
    }
    int off = incl - c;
#pragma unroll
    for (int j = 0; j < NPL; j++)
        if ((uint32_t)(it[j] >> 32) <= kb && it[j] != ~0ull) buf[off++] = it[j];
    __builtin_amdgcn_wave_barrier();
    // rank sort of the tot (<= 256) distinct items: lane holds items lane, lane + 64, ..
    unsigned long long mine[CF_BUF / 64];
    int rr[CF_BUF / 64];
#pragma unroll
    for (int u = 0; u < CF_BUF / 64; u++) {
        mine[u] = lane + 64 * u < tot ? buf[lane + 64 * u] : ~0ull;
        rr[u] = 0;
    }
    if (tot <= 64) {
        for (int j = 0; j < tot; j++) rr[0] += buf[j] < mine[0] ? 1 : 0;
    } else {
        for (int j = 0; j < tot; j++) {
            const unsigned long long xj = buf[j];
#pragma unroll
            for (int u = 0; u < CF_BUF / 64; u++) rr[u] += xj < mine[u] ? 1 : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < CF_BUF / 64; u++) {
        if (lane + 64 * u < tot && rr[u] < P) {
            out_dis[(int64_t)q * P + rr[u]] = key2f((uint32_t)(mine[u] >> 32));
            out_idx[(int64_t)q * P + rr[u]] = (int)(uint32_t)mine[u];
        }
    }
    for (int r = tot + lane; r < P; r += 64) {   // fewer than P centroids in all
        out_dis[(int64_t)q * P + r] = INFINITY;
        out_idx[(int64_t)q * P + r] = -1;
    }
    if (flag_ties) {
        // exact ties: two equal keys among the P + 1 smallest (every entry at the P-th key is <= kb, hence here) --
        // which of them is probed, and in which order equal ones are scanned, is the doing of the reference's heap:
        // the row goes to k_coarse_repair, which then replays that heap
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < CF_BUF / 64; u++)
            if (lane + 64 * u < tot) buf[rr[u]] = mine[u];
        __builtin_amdgcn_wave_barrier();
        const int have = min(tot, P + 1);
        bool eq = false;
        for (int i = lane; i + 1 < have; i += 64) eq |= (uint32_t)(buf[i] >> 32) == (uint32_t)(buf[i + 1] >> 32);
        if (__ballot(eq) && lane == 0) ovf[1 + atomicAdd(ovf, 1)] = q;
    }
}

// The queries k_coarse_final gave up on, one workgroup each: all ny distances with the chain of the MFMA kernels
// (k-ascending fma from 0, norms as fvec_norm_L2sqr), keys in a scratch row, then P rounds of block arg-min on
// (key, centroid).  Rare by construction; launched with a fixed small grid that finds an empty list.
__global__ __launch_bounds__(256) void k_coarse_repair(const float* __restrict__ x, int d, const float* __restrict__ y, int ny,
                                                       const float* __restrict__ yn, const int* __restrict__ ovf,
                                                       uint32_t* __restrict__ scratch, int P,
                                                       float* __restrict__ out_dis, int* __restrict__ out_idx) {
    extern __shared__ float s_x[];   // [d]
    __shared__ float s_n;
    __shared__ unsigned long long s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = ovf[0];
    uint32_t* keys = scratch + (int64_t)blockIdx.x * ny;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int q = ovf[1 + i];
        __syncthreads();
        for (int t = tid; t < d; t += 256) s_x[t] = x[(int64_t)q * d + t];
        __syncthreads();
        if (tid == 0) {
            float a4[4] = {0.f, 0.f, 0.f, 0.f};
            int t = 0;
            for (; t + 4 <= d; t += 4)
                for (int l = 0; l < 4; l++) a4[l] = __builtin_fmaf(s_x[t + l], s_x[t + l], a4[l]);
            for (int l = 0; t + l < d; l++) a4[l] = a4[l] + s_x[t + l] * s_x[t + l];   // masked tail block: mul + add
            s_n = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        }
        __syncthreads();
        const float xn = s_n;
        for (int c = tid; c < ny; c += 256) {
            const float* yr = y + (int64_t)c * d;
            float ip = 0.f;
            for (int t = 0; t < d; t++) ip = __builtin_fmaf(s_x[t], yr[t], ip);
            float dis = (xn + yn[c]) - 2.f * ip;
            if (dis < 0.f) dis = 0.f;
            keys[c] = f2key(dis);
        }
        __syncthreads();
        unsigned long long last = 0;
        bool first = true;
        for (int r = 0; r < P; r++) {
            unsigned long long best = ~0ull;
            for (int c = tid; c < ny; c += 256) {
                const unsigned long long item = ((unsigned long long)keys[c] << 32) | (unsigned)c;
                if ((first || item > last) && item < best) best = item;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long t = __shfl_xor(best, o, 64);
                best = t < best ? t : best;
            }
            if (lane == 0) s_red[w] = best;
            __syncthreads();
            best = s_red[0];
            for (int k = 1; k < 4; k++) best = s_red[k] < best ? s_red[k] : best;
            __syncthreads();
            if (tid == 0) {
                out_dis[(int64_t)q * P + r] = best == ~0ull ? INFINITY : key2f((uint32_t)(best >> 32));
                out_idx[(int64_t)q * P + r] = best == ~0ull ? -1 : (int)(uint32_t)best;
            }
            last = best;
            first = false;
            if (best == ~0ull) {
                for (int r2 = r + 1 + tid; r2 < P; r2 += 256) {
                    out_dis[(int64_t)q * P + r2] = INFINITY;
                    out_idx[(int64_t)q * P + r2] = -1;
                }
                break;
            }
        }
    }
}

// (sample, strips) for a shape, or false.  tau is the P-th smallest of 64 lane minima over NPL = sample / 64 values
// each (k_coarse_bound): it sits at the row quantile F = 1 - (1 - P/64)^(1/NPL) -- 0.69 / NPL at P = 32 -- and a strip
// then keeps about F (nlist - sample) / strips entries per query, sd ~17 %.  The strip lists hold 127: the plan keeps
// the mean at 48 or less (ten sigma of margin at C3's 37), first with 16 strips instead of 8, then with a larger
// sample.  P > 32: two minima per lane, the P-th of 128 (the P-th of 64 would be no bound: at P = 64 the LARGEST lane
// minimum, a quarter of the row), F = 1 - (1 - P/128)^(2/NPL).
static bool coarse_fused_shape(int nlist, int P, int* sample, int* nseg) {
    if (P < 1 || P > 64) return false;
    for (int sm = P <= 32 ? 512 : 1024; sm <= 2048; sm *= 2) {
        if (nlist < 4 * sm) break;
        const double F = P <= 32 ? 1.0 - pow(1.0 - P / 64.0, 64.0 / sm) : 1.0 - pow(1.0 - P / 128.0, 128.0 / sm);
        for (int sg = 8; sg <= 16; sg *= 2) {
            if (F * (nlist - sm) / sg <= 48.0) {
                *sample = sm;
                *nseg = sg;
                return true;
            }
        }
    }
    return false;
}

bool coarse_fused_supported(int nq, int d, int nlist, int P, bool exact_ties) {
    int sm, sg;
    if (!(nq >= 4096 && (d == 32 || d == 64 || d == 96 || d == 128) && coarse_fused_shape(nlist, P, &sm, &sg))) return false;
    // the strip lists and the sample matrix are addressed through 32-bit buffer offsets (k_coarse_fused); with exact
    // ties also the full rows of the flagged queries
    if (exact_ties && (int64_t)nq * nlist * 4 >= 0x7fffffffLL) return false;
    return (int64_t)nq * sg * kCoarseCap * 8 < 0x7fffffffLL && (int64_t)nq * sm * 4 < 0x7fffffffLL;
}

CoarseFusedPlan coarse_fused_plan(int nq, int nlist, int P, int cap, bool exact_ties) {
    CoarseFusedPlan pl;
    // C3 (nlist 4096, P 32): sample 512, 8 strips of 448 columns that keep 37 +- 9 entries per query; kCoarseCap = 128
    // is ten sigma away, the rest goes to k_coarse_repair
    if (!coarse_fused_shape(nlist, P, &pl.sample, &pl.nseg)) {   // callers ask coarse_fused_supported first
        launch_refused("coarse_fused_plan: shape outside the matrix-free coarse path (coarse_fused_supported)");
        pl.sample = 512;
        pl.nseg = 1;
    }
    const int ntiles = (nlist - pl.sample + 63) / 64;
    pl.tiles_per_strip = (ntiles + pl.nseg - 1) / pl.nseg;
    pl.nseg = (ntiles + pl.tiles_per_strip - 1) / pl.tiles_per_strip;
    pl.cap_stride = kCoarseCap;
    pl.cap = std::max(1, std::min(kCoarseCap - 1, cap));   // slot `cap` of a list is the dump slot of overflowing rows
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 255) & ~(size_t)255;
        return at;
    };
    pl.off_mat = take((size_t)nq * pl.sample * sizeof(float));
    pl.off_tau = take((size_t)nq * sizeof(float));
    pl.off_cand = take((size_t)nq * pl.nseg * pl.cap_stride * sizeof(unsigned long long));
    pl.off_cnt = take((size_t)nq * pl.nseg * sizeof(int));
    pl.off_ovf = take((size_t)(nq + 1) * sizeof(int));
    pl.off_scratch = take((size_t)kCoarseRepairGrid * nlist * sizeof(uint32_t));
    // exact ties: full distance rows of the flagged queries (row i = the i-th flagged query; sized for all of them)
    pl.off_full = exact_ties ? take((size_t)nq * nlist * sizeof(float)) : 0;
    pl.bytes = o;
    return pl;
}

void launch_coarse_fused(hipStream_t s, const CoarseFusedPlan& pl, void* ws, const float* x, int nq, int d,
                         const float* y, int nlist, const float* yn, int P, float* out_dis, int* out_idx, bool exact_ties,
                         unsigned long long* tie_stats, hipStream_t side, hipEvent_t fork, hipEvent_t join) {
    char* b = static_cast<char*>(ws);
    float* mat = reinterpret_cast<float*>(b + pl.off_mat);
    float* tau = reinterpret_cast<float*>(b + pl.off_tau);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(b + pl.off_cand);
    int* cnt = reinterpret_cast<int*>(b + pl.off_cnt);
    int* ovf = reinterpret_cast<int*>(b + pl.off_ovf);
    uint32_t* scratch = reinterpret_cast<uint32_t*>(b + pl.off_scratch);
    // A: the sample columns, every distance stored (two strips: the 128 queries' fragments are loaded once per strip)
    const size_t lds = (size_t)2 * 64 * (d + 1) * sizeof(float);
    {
        const int stiles = pl.sample / 64, sstrips = 4, tps = (stiles + sstrips - 1) / sstrips;
        dim3 sgrid((unsigned)sstrips, (unsigned)((nq + 127) / 128));
#define GH_CS(NCH)                                                                                                      \
    do {                                                                                                                \
        static std::atomic<uint64_t> attr{0};   /* per device */                                                        \
        if (first_call_on_device(attr)) {                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_coarse_fused<NCH, true>),                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (16 * NCH + 1) * 4);         \
        }                                                                                                               \
        hipLaunchKernelGGL((k_coarse_fused<NCH, true>), sgrid, dim3(256), lds, s, x, nq, y, pl.sample, 0, yn, nullptr, \
                           tps, 0, pl.sample, reinterpret_cast<unsigned long long*>(mat), nullptr, sstrips, nullptr);  \
    } while (0)
        switch (d) {
            case 32: GH_CS(2); break;
            case 64: GH_CS(4); break;
            case 96: GH_CS(6); break;
            default: GH_CS(8); break;
        }
#undef GH_CS
    }
    if (pl.sample == 512) hipLaunchKernelGGL(k_coarse_bound<8>, dim3((nq + 3) / 4), dim3(256), 0, s, mat, nq, P, tau, ovf);
    else if (pl.sample == 1024) hipLaunchKernelGGL(k_coarse_bound<16>, dim3((nq + 3) / 4), dim3(256), 0, s, mat, nq, P, tau, ovf);
    else hipLaunchKernelGGL(k_coarse_bound<32>, dim3((nq + 3) / 4), dim3(256), 0, s, mat, nq, P, tau, ovf);
    // B: the other columns, filtered
    dim3 grid((unsigned)pl.nseg, (unsigned)((nq + 127) / 128));
#define GH_CF(NCH)                                                                                                      \
    do {                                                                                                                \
        static std::atomic<uint64_t> attr{0};   /* per device */                                                        \
        if (first_call_on_device(attr)) {                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_coarse_fused<NCH>),                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (16 * NCH + 1) * 4);         \
        }                                                                                                               \
        hipLaunchKernelGGL((k_coarse_fused<NCH>), grid, dim3(256), lds, s, x, nq, y, nlist, pl.sample, yn, tau,         \
                           pl.tiles_per_strip, pl.cap, pl.cap_stride, cand, cnt, pl.nseg, nullptr);                          \
    } while (0)
    switch (d) {
        case 32: GH_CF(2); break;
        case 64: GH_CF(4); break;
        case 96: GH_CF(6); break;
        default: GH_CF(8); break;
    }
#undef GH_CF
    // C + D
#define GH_FIN(SN, MS)                                                                                                  \
    hipLaunchKernelGGL((k_coarse_final<SN, MS>), dim3((nq + 3) / 4), dim3(256), 0, s, mat, tau, cand, cnt, pl.nseg, pl.cap, \
                       pl.cap_stride, nq, P, out_dis, out_idx, ovf, exact_ties ? 1 : 0)
    if (pl.sample == 512) {
        if (pl.nseg <= 4) GH_FIN(8, 4);
        else if (pl.nseg <= 8) GH_FIN(8, 8);
        else GH_FIN(8, 16);
    } else if (pl.sample == 1024) {
        if (pl.nseg <= 8) GH_FIN(16, 8);
        else GH_FIN(16, 16);
    } else {
        if (pl.nseg <= 8) GH_FIN(32, 8);
        else GH_FIN(32, 16);
    }
#undef GH_FIN
    if (!exact_ties) {
        hipLaunchKernelGGL(k_coarse_repair, dim3(kCoarseRepairGrid), dim3(256), (size_t)d * sizeof(float), s, x, d, y, nlist, yn,
                           ovf, scratch, P, out_dis, out_idx);
        return;
    }
    // Exact ties: every row on the list (overflowed, or two equal keys among its P + 1 smallest -- a few in a thousand)
    // is recomputed in full by the same MFMA chain (store-all mode of the strip kernel, row i = the i-th listed
    // query) and walked the way faiss's HeapResultHandler walks it (k_coarse_heap_fix).  The two kernels may run on a
    // side stream beside the caller's next kernels that do not read the assignment; the caller waits for `join`
    // before the first one that does.
    // (the recompute stays on the caller's stream: beside the caller's next kernel -- the query tables, which fill the
    //  chip -- its few workgroups wait for slots and the launch takes 65 us instead of 25; the walk, one wave per row and
    //  all latency, is what runs beside it)
    static const bool store_side = getenv("GAMMA_HIP_COARSE_STORE_SIDE") != nullptr;
    hipStream_t rs = s;
    const bool forked = side && fork && join;
    if (forked && store_side) {
        (void)hipEventRecord(fork, s);
        (void)hipStreamWaitEvent(side, fork, 0);
        rs = side;
    }
    float* full = reinterpret_cast<float*>(b + pl.off_full);
    {
        // (64 strips: the few workgroups that have rows to do take one tile each at nlist 4096 -- the launch is the latency of one)
        const int ntiles = (nlist + 63) / 64, fstrips = std::min(ntiles, 64), tps = (ntiles + fstrips - 1) / fstrips;
        dim3 fgrid((unsigned)fstrips, (unsigned)((nq + 127) / 128));
#define GH_CR(NCH)                                                                                                      \
    hipLaunchKernelGGL((k_coarse_fused<NCH, true>), fgrid, dim3(256), lds, rs, x, nq, y, nlist, 0, yn, nullptr, tps, 0,  \
                       nlist, reinterpret_cast<unsigned long long*>(full), nullptr, fstrips, ovf)
        switch (d) {   // (the attribute of this instantiation was set by the sample launch above)
            case 32: GH_CR(2); break;
            case 64: GH_CR(4); break;
            case 96: GH_CR(6); break;
            default: GH_CR(8); break;
        }
#undef GH_CR
    }
    if (forked && !store_side) {
        (void)hipEventRecord(fork, s);
        (void)hipStreamWaitEvent(side, fork, 0);
        rs = side;
    }
    launch_coarse_heap_rows(rs, full, nlist, nq, P, ovf, out_dis, out_idx, tie_stats);
    if (rs != s) (void)hipEventRecord(join, rs);
}

}  // namespace gh
