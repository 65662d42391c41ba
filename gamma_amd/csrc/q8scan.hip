// q8scan.hip -- a6, list-major: the consumer side of the bounded L2 list scan re-mapped from "one query per workgroup"
// to "one inverted list x a tile of 8 of the queries that probe it" (VERDICT r4 #3).
//
// Why the mapping had to change, and why with BYTE tables.  The query-major filter pass (scan.hip, CF) issues one
// ds_read_b32 gather per (query, code, sub-quantizer): 64 random addresses in one 256-entry table row, ~3.5-way bank
// conflicts, 57 % of the LDS cycles (profiles/r04_pmc_scan_summary.json), and every one of the ~128 queries that probe a
// list in a 16384-query batch reads the list's codes again.  A gather can only serve several queries if their table
// entries for the same code byte sit side by side in LDS, i.e. if the workgroup's queries all want the SAME code: list-major.
// Round 2 tried that with the fp32 tables (scan_lm.hip: two queries per pass, 58 % of the vector instructions, but 1.07 ms
// against 0.70 ms -- a 16 KB table per (query, list) pair through the L2 costs more than the 4 KB of codes it is used on).
// The filter pass, however, does not need the table's exact values -- it only has to PROVE a code outside the bound, and
// the few codes it cannot prove outside are recomputed exactly afterwards (as in CF).  So the table travels as BYTES:
//     ip[q][m][c]  ~  lo[q][m] + delta[q] * u8[q][m][c],   |error| <= 0.5001 delta per entry,
// 4 KB per (query, list) pair instead of 16, eight queries' bytes in one 8-byte LDS entry:
//     LDS  lut[m][c + (c >> 3)] = { u8 of the tile's 8 queries }      (one pad entry per 8: conflict-free staging stores)
// and ONE ds_read_b64 per (code, m) serves eight queries; the per-query sums are integers (v_dot4_u32_u8 with a one-hot
// selector), exact.  With  U = sum_m u8[q][m][c_m]  the true value of the reference's ADC
//     v = dis0 + sum_m fma(-2, ip[q][m][c_m], T2[l][m][c_m])         (gamma_index_ivfpq.h:575-601, sequential fp32)
// satisfies  v >= (dis0 - 2 sum_m lo - 1.02 M delta) + s_j - 2 delta U - 50 * 2^-24 S   with the per-code table sum s_j and
// the magnitude bound S of the CF pass (scan.hip), so a code is kept as a CANDIDATE iff
//     (A_q + s_j) - 2 delta_q U  <=  tau_q + 2^-16 S ,   A_q = dis0 - cq_q,  cq_q = 2 sum_m lo + 1.02 M delta .
// Candidates (a few % of the codes; ~20 % more than the fp32 filter lets through) go to a per-query list; k_q8_exact -- one
// workgroup per query, the query's fp32 table in LDS -- gives them the reference's exact value in the reference's order of
// operations and appends those within the bound to the query's consumer slice: the same (key, position) items the CF pass
// and the regular loop would have put there, so k_select_final, the repair launch and the tie replay are unchanged.
//
// Launches (launch_q8_consumers): tables -> bytes | pairs counted per list | offsets, tiles | pairs filled | filter | exact.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "block_utils.h"
#include "device_math.h"
#include "filter_dev.h"
#include "kernels.h"
#include "scan_dev.h"

namespace gh {

namespace {
constexpr int Q8_T = 8;          // queries per tile
constexpr int Q8_DEMAND_MAX = 40;   // k_q8_exact: candidates of a query up to which their table entries are computed on demand
// LUT entries per sub-quantizer row: 256 + one pad per 8 at M = 16 (conflict-free staging stores; 36 KB, four workgroups
// per CU).  M = 32 goes without the pad (64 KB + the pool = two workgroups per CU; its lists are long, the staging amortised)
template <int MT> struct Q8Lut {
    static constexpr bool PAD = MT <= 16;
    static constexpr int ROW = PAD ? 288 : 256;
    __device__ static __forceinline__ uint32_t entry(uint32_t c) { return PAD ? c + (c >> 3) : c; }
    __device__ static __forceinline__ int block(int cb) { return PAD ? cb * 9 : cb * 8; }
};
// candidates staged per tile (all of its queries; more = the tile's queries take the unfiltered path): what the LDS beside
// the table leaves at four (M = 16) / two (M = 32: long lists, hundreds of candidates from the nearest lists) workgroups per CU
template <int MT> struct Q8Pool { static constexpr int N = MT <= 16 ? 640 : 3584; };
constexpr int Q8_POS_BITS = 25;  // candidate = position in the query's segment | probe << 25
}  // namespace

// candidates a query's list holds (beyond: the query takes the unfiltered path): GAMMA_HIP_Q8_CAND_MB (default 512) MB of
// workspace spread over the batch, 768 .. 32768 entries per query
int q8_cand_cap(int nq) {
    static const int64_t mb = getenv("GAMMA_HIP_Q8_CAND_MB") ? atoll(getenv("GAMMA_HIP_Q8_CAND_MB")) : 512;
    return (int)std::max<int64_t>(768, std::min<int64_t>(32768, ((mb << 20) / 4) / std::max(1, nq)));
}

// ------------------------------------------------------------------------------------
// u8 image of every query's inner-product table (k_pq_ip_table's st2): one workgroup per query, thread = code word c.
// meta[q] = { cq = 2 sum_m lo_m + 1.02 M delta,  -2 delta,  max |entry|,  0 }
// ------------------------------------------------------------------------------------
// fx != nullptr (two-phase list shards, round 6): the table is not in memory -- a shard sees W times the queries of a rank, and
// W x 32 KB per query written by k_pq_ip_table and read back here and by k_q8_exact was a tenth of its step -- the entries are
// computed here from the query and the PQ codebook (128 KB, L2-resident), the arithmetic of k_pq_ip_table: identical values.
template <int MT>
__global__ __launch_bounds__(256) void k_q8_quant(const float* __restrict__ st2, uint8_t* __restrict__ q8,
                                                  float4* __restrict__ meta, const float* __restrict__ fx,
                                                  const float* __restrict__ pqc, int d) {
    // wave w takes table rows w, w + 4, ..: a row is one 1 KB read of the wave (four code words per lane), its minimum and
    // maximum one wave reduction, its bytes one 256-byte store
    constexpr int NR = MT / 4;
    __shared__ float s_part[4][3];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float4 v[NR];
    float lo[NR], range = 0.f, L = 0.f, amax = 0.f;
    if (fx) {   // (uniform)
        const int dsub = d / MT;
        const float* xq = fx + (int64_t)q * d;
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int m = wv + 4 * k;
            const float* c = pqc + ((int64_t)m * 256 + 4 * lane) * dsub;
            v[k].x = fvec_ny_row<false>(xq + m * dsub, c, dsub);
            v[k].y = fvec_ny_row<false>(xq + m * dsub, c + dsub, dsub);
            v[k].z = fvec_ny_row<false>(xq + m * dsub, c + 2 * dsub, dsub);
            v[k].w = fvec_ny_row<false>(xq + m * dsub, c + 3 * dsub, dsub);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NR; k++) v[k] = *reinterpret_cast<const float4*>(st2 + ((int64_t)q * MT + wv + 4 * k) * 256 + 4 * lane);
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
        float mn = fminf(fminf(v[k].x, v[k].y), fminf(v[k].z, v[k].w)), mx = fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, o, 64));
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        }
        lo[k] = mn;
        range = fmaxf(range, mx - mn);
        L += mn;
        amax = fmaxf(amax, fmaxf(fabsf(mn), fabsf(mx)));
    }
    if (lane == 0) {
        s_part[wv][0] = range;
        s_part[wv][1] = L;
        s_part[wv][2] = amax;
    }
    __syncthreads();
    range = fmaxf(fmaxf(s_part[0][0], s_part[1][0]), fmaxf(s_part[2][0], s_part[3][0]));
    L = (s_part[0][1] + s_part[1][1]) + (s_part[2][1] + s_part[3][1]);
    amax = fmaxf(fmaxf(s_part[0][2], s_part[1][2]), fmaxf(s_part[2][2], s_part[3][2]));
    // delta a few ulps above range / 255: (hi - lo) / delta stays below 255.5 whatever the roundings
    const float delta = (range / 255.f) * 1.000001f;
    const float inv = delta > 0.f ? 1.f / delta : 0.f;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const float f[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            int u = (int)rintf((f[e] - lo[k]) * inv);
            u = min(255, max(0, u));
            w |= (uint32_t)u << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(q8 + ((int64_t)q * MT + wv + 4 * k) * 256 + 4 * lane) = w;
    }
    if (tid == 0) meta[q] = make_float4(2.f * L + 1.02f * (float)MT * delta, -2.f * delta, amax, 0.f);
}

// ------------------------------------------------------------------------------------
// The consumer pairs (query, probe >= G) of the queries WITH a bound, grouped by list -- a counting sort without global
// atomics (393 k device-scope atomics on 4096 counters cost 85 us a pass): Q8_NW workgroups each take a contiguous run of
// the pairs, count them per list in LDS (k_q8_hist -> hist[w][l]), k_q8_offsets turns the columns into exclusive offsets
// (hist[w][l] = first position of workgroup w's pairs of list l) and k_q8_fill places the pairs with LDS cursors.
// A query WITHOUT a bound goes to the repair list (its consumer groups are scored with stores by the repair launch:
// the unfiltered selection reads the slab).
// ------------------------------------------------------------------------------------
constexpr int Q8_NW = 64;
// what a tile needs to know about one of its (query, probe) pairs, computed once when the pair is placed (k_q8_hist<true>):
// the filter kernels read ONE 32-byte record per slot instead of walking pair -> pair_off / meta / coarse_dis / ready
struct Q8Rec {
    int q, pp, pb, pad;      // query, probe, position of the pair's list in the query's segment
    float A, nd, thr, pad2;  // dis0 - cq, -2 delta, tau + 2^-16 S
};
template <bool FILL>
__global__ __launch_bounds__(1024) void k_q8_hist(const int* __restrict__ probe_list, int nq, int P, int G,
                                                  const unsigned long long* __restrict__ ready, const int* __restrict__ list_len,
                                                  const uint8_t* __restrict__ list_mask, int nlist, int* __restrict__ hist,
                                                  Q8Rec* __restrict__ recs, int* __restrict__ rq_list, int* __restrict__ rq_count,
                                                  const float* __restrict__ coarse_dis, const float* __restrict__ t2max,
                                                  const float4* __restrict__ meta, const int* __restrict__ pair_off,
                                                  const int* __restrict__ off, const int* __restrict__ tile_first) {
    extern __shared__ int s_h[];   // [nlist]
    const int tid = threadIdx.x, w = blockIdx.x;
    const int per = P - G;
    const int64_t pairs = (int64_t)nq * per, chunk = (pairs + Q8_NW - 1) / Q8_NW;
    const int64_t a = (int64_t)w * chunk, b = min(pairs, a + chunk);
    for (int l = tid; l < nlist; l += 1024) s_h[l] = FILL ? hist[(int64_t)w * nlist + l] : 0;
    __syncthreads();
    for (int64_t idx = a + tid; idx < b; idx += 1024) {
        const int q = (int)(idx / per), p = G + (int)(idx - (int64_t)q * per);
        const unsigned long long word = ready[q];
        if ((word >> 32) != 1ull) {
            if (!FILL && p == G) rq_list[atomicAdd(rq_count, 1)] = q;
            continue;
        }
        const int pair = q * P + p;
        const int l = probe_list[pair];
        if (l < 0 || l >= nlist || (list_mask && !list_mask[l]) || list_len[l] <= 0) continue;
        const int at = atomicAdd(&s_h[l], 1);
        if (FILL) {
            const float4 mq = meta[q];
            const float dis0 = coarse_dis[pair];
            const float tau = key2f((uint32_t)word);
            const float S = fabsf(dis0) + t2max[l] + 32.f * mq.z;
            float thr = __builtin_fmaf(S, 1.f / 65536.f, tau);
            thr += fabsf(thr) * 2.4e-7f;   // the threshold's own roundings
            Q8Rec r;
            r.q = q; r.pp = p; r.pb = pair_off[(int64_t)q * (P + 1) + p]; r.pad = 0;
            r.A = dis0 - mq.x; r.nd = mq.y; r.thr = thr; r.pad2 = 0.f;
            recs[(int64_t)tile_first[l] * Q8_T + (at - off[l])] = r;   // slot (at - off) % 8 of tile tile_first + (at - off) / 8
        }
    }
    if (!FILL) {
        __syncthreads();
        for (int l = tid; l < nlist; l += 1024) hist[(int64_t)w * nlist + l] = s_h[l];
    }
}

// pairs per list = the sum of its column of the workgroups' histograms (thread = list: coalesced rows)
__global__ __launch_bounds__(256) void k_q8_colsum(const int* __restrict__ hist, int nlist, int* __restrict__ cnt) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nlist) return;
    int n = 0;
    for (int w = 0; w < Q8_NW; w++) n += hist[(int64_t)w * nlist + l];
    cnt[l] = n;
}
// the column becomes the workgroups' first positions inside the list's run
// ... and the list's tiles get their (list, length, arena offset) entries: a tile starts on its codes without walking the
// list tables
__global__ __launch_bounds__(256) void k_q8_colfix(int* __restrict__ hist, int nlist, const int* __restrict__ off,
                                                   const int* __restrict__ tile_first, int4* __restrict__ tile_list,
                                                   const int* __restrict__ list_len, const int64_t* __restrict__ list_off) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nlist) return;
    int run = off[l];
    for (int w = 0; w < Q8_NW; w++) {
        const int hv = hist[(int64_t)w * nlist + l];
        hist[(int64_t)w * nlist + l] = run;
        run += hv;
    }
    const int t0 = tile_first[l], t1 = tile_first[l + 1];
    if (t1 > t0) {
        const int64_t lo = list_off[l];
        const int4 ti = make_int4(l, list_len[l], (int)(uint32_t)lo, (int)(lo >> 32));
        for (int t = t0; t < t1; t++) tile_list[t] = ti;
    }
}
// offsets of the lists' pair runs, the tiles of 8 pairs, the tile -> list table; one workgroup
__global__ __launch_bounds__(1024) void k_q8_offsets(const int* __restrict__ cnt, int nlist, int* __restrict__ off,
                                                     int* __restrict__ tile_first, int* __restrict__ n_tiles) {
    __shared__ int s_c[1024], s_t[1024];
    const int tid = threadIdx.x;
    const int per = (nlist + 1023) / 1024, a = min(nlist, tid * per), b = min(nlist, a + per);
    int c = 0, t = 0;
    for (int l = a; l < b; l++) {
        c += cnt[l];
        t += (cnt[l] + Q8_T - 1) / Q8_T;
    }
    s_c[tid] = c;
    s_t[tid] = t;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // inclusive scans
        const int ac = tid >= o ? s_c[tid - o] : 0, at = tid >= o ? s_t[tid - o] : 0;
        __syncthreads();
        s_c[tid] += ac;
        s_t[tid] += at;
        __syncthreads();
    }
    int oc = s_c[tid] - c, ot = s_t[tid] - t;
    for (int l = a; l < b; l++) {
        const int n = cnt[l];
        off[l] = oc;
        tile_first[l] = ot;
        const int nt = (n + Q8_T - 1) / Q8_T;
        oc += n;
        ot += nt;
    }
    if (tid == 1023) {
        off[nlist] = s_c[1023];
        tile_first[nlist] = s_t[1023];
        *n_tiles = s_t[1023];
    }
}

// 4x4 byte transpose of the dwords a0..a3: t_j byte i = a_i byte j
__device__ __forceinline__ void tr4x4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t& t0, uint32_t& t1,
                                      uint32_t& t2, uint32_t& t3) {
    const uint32_t p0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), p1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
    const uint32_t p2 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), p3 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
    t0 = __builtin_amdgcn_perm(p2, p0, 0x05040100u);
    t1 = __builtin_amdgcn_perm(p2, p0, 0x07060302u);
    t2 = __builtin_amdgcn_perm(p3, p1, 0x05040100u);
    t3 = __builtin_amdgcn_perm(p3, p1, 0x07060302u);
}

// sum += byte k of w (exact integer)
template <int K>
__device__ __forceinline__ uint32_t add_byte(uint32_t acc, uint32_t w) {
    return __builtin_amdgcn_udot4(w, 1u << (8 * K), acc, false);
}

// LDS byte address of table entry (m, code byte k of w) in the UNPADDED layout (M = 32): (byte << 3) + 2048 m with the table
// at LDS address 0 (the kernels' only LDS is the dynamic block) -- one SDWA shift per look-up, the row offset in the ds_read's
// immediate (as lut_gather, scan_dev.h); the padded layout (M = 16) pays the extra c + (c >> 3)
template <int MT>
__device__ __forceinline__ uint2 q8_gather(const unsigned char* lut, uint32_t w, int k, int m) {
    if constexpr (!Q8Lut<MT>::PAD) {
        uint32_t a;
        switch (k) {   // constant after unrolling
            case 0: asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(w)); break;
            case 1: asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(w)); break;
            case 2: asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(w)); break;
            default: asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a) : "v"(w)); break;
        }
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>((uintptr_t)(a + 2048u * (uint32_t)m));
        return make_uint2(v.x, v.y);
    } else {
        const uint32_t c = (w >> (8 * k)) & 255u;
        return *reinterpret_cast<const uint2*>(lut + (size_t)m * Q8Lut<MT>::ROW * 8 + Q8Lut<MT>::entry(c) * 8);
    }
}

// ------------------------------------------------------------------------------------
// The filter: a persistent grid walks the tiles.  Q8_NT threads; a wave takes 64 codes of the list per step.
// (512 threads: the table limits a CU to two (M = 32) / four (M = 16) workgroups, and with four waves each the code loads
//  in flight -- one 2 KB step per wave -- left the pass waiting for memory: 5.4 ms per 3216 queries at full-size C4)
// ------------------------------------------------------------------------------------
constexpr int Q8_NT = 512;
// NID: the validity predicates are compiled in.  (Two kernels instead of a run-time `if (need_ids)`: with the predicates' loads --
// through generic pointers of the filter table -- in the loop body the compiler's wait-count pass puts `s_waitcnt vmcnt(0)` at
// the join behind them, in front of the gathers: every step then waited for the codes of the next steps it had just requested.)
template <int MT, bool NID>
__global__ __launch_bounds__(Q8_NT) void k_q8_filter(
        const int4* __restrict__ tile_list, const int* __restrict__ tile_first, const int* __restrict__ n_tiles,
        const int* __restrict__ pair_run, const Q8Rec* __restrict__ recs, const uint8_t* __restrict__ q8,
        const uint8_t* __restrict__ codes, const float* __restrict__ sums, const int64_t* __restrict__ ids,
        const FilterDesc* __restrict__ ftab, int need_ids, uint32_t* __restrict__ cand, int* __restrict__ ccnt, int cand_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_q8[];
    constexpr int ROW = Q8Lut<MT>::ROW;
    unsigned char* lut = s_q8;                                                              // [MT][ROW][8]
    float* s_A = reinterpret_cast<float*>(s_q8 + (size_t)MT * ROW * 8);                     // [8] dis0 - cq
    float* s_nd = s_A + 8;                                                                  // [8] -2 delta
    float* s_thr = s_nd + 8;                                                                // [8]
    int* s_q = reinterpret_cast<int*>(s_thr + 8);                                           // [8] query (-1: empty slot)
    int* s_pb = s_q + 8;                                                                    // [8] segment position of the pair's list
    int* s_pp = s_pb + 8;                                                                   // [8] probe
    int* s_n = s_pp + 8;                                                                    // [8] candidates
    int* s_g = s_n + 8;                                                                     // [8] base in the query's list
    int* s_cur = s_g + 8;                                                                   // [8] copied so far | [8]: pool counter | [9]: valid pool entries
    uint32_t* s_pool = reinterpret_cast<uint32_t*>(s_cur + 16);                             // [POOL] query slot << 28 | code
    constexpr int POOL = Q8Pool<MT>::N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntile = *n_tiles;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int4 ti = tile_list[tile];
        const int nit = Q8_T;   // (slots past the tile's pairs hold q = -1 records: the launcher's fill)
        if (tid < Q8_T) {
            Q8Rec r = recs[(int64_t)tile * Q8_T + tid];
            if (r.q < 0) { r.pp = 0; r.pb = 0; r.A = 0.f; r.nd = 0.f; r.thr = -INFINITY; }
            s_q[tid] = r.q;
            s_pb[tid] = r.pb;
            s_pp[tid] = r.pp;
            s_A[tid] = r.A;
            s_nd[tid] = r.nd;
            s_thr[tid] = r.thr;
            s_n[tid] = 0;
            s_cur[tid] = 0;
            if (tid == 0) s_cur[8] = s_cur[9] = 0;
        }
        __syncthreads();
        // ---- the tile's table: 8 queries' bytes side by side.  A thread takes (m, 8 consecutive code words) blocks ----
        for (int bi = tid; bi < MT * 32; bi += Q8_NT) {
            const int m = bi >> 5, cb = bi & 31;
            uint2 r[Q8_T];
#pragma unroll
            for (int i = 0; i < Q8_T; i++) {
                const int q = s_q[i];
                r[i] = q >= 0 ? *reinterpret_cast<const uint2*>(q8 + ((int64_t)q * MT + m) * 256 + cb * 8) : make_uint2(0u, 0u);
            }
            uint32_t xl[4], xh[4], yl[4], yh[4];
            tr4x4(r[0].x, r[1].x, r[2].x, r[3].x, xl[0], xl[1], xl[2], xl[3]);   // code words 0..3, queries 0..3
            tr4x4(r[4].x, r[5].x, r[6].x, r[7].x, xh[0], xh[1], xh[2], xh[3]);   //                 queries 4..7
            tr4x4(r[0].y, r[1].y, r[2].y, r[3].y, yl[0], yl[1], yl[2], yl[3]);   // code words 4..7
            tr4x4(r[4].y, r[5].y, r[6].y, r[7].y, yh[0], yh[1], yh[2], yh[3]);
            uint2* dst = reinterpret_cast<uint2*>(lut + ((size_t)m * ROW + Q8Lut<MT>::block(cb)) * 8);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                dst[j] = make_uint2(xl[j], xh[j]);
                dst[4 + j] = make_uint2(yl[j], yh[j]);
            }
        }
        __syncthreads();
        // ---- the list's codes ----
        const int len = ti.y;
        const int64_t off = (int64_t)(((uint64_t)(uint32_t)ti.w << 32) | (uint32_t)ti.z);
        const uint8_t* lc = codes + off * MT;
        const float* ls = sums + off;
        const int64_t* lid = ids + off;
        // the codes and sums of the next TWO steps are in flight during this step's gathers
        uint4 cn[2][MT / 16];
        float sn[2] = {0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; a++)
            if (wv * 64 + a * Q8_NT < len) {
                const int jc = min(wv * 64 + a * Q8_NT + lane, len - 1);
                const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)jc * MT);
#pragma unroll
                for (int u = 0; u < MT / 16; u++) cn[a][u] = cp[u];
                sn[a] = ls[jc];
            }
        for (int j0 = wv * 64; j0 < len; j0 += Q8_NT) {
            const int j = j0 + lane, jc = min(j, len - 1);
            uint32_t cw[MT / 4];
#pragma unroll
            for (int u = 0; u < MT / 16; u++) {
                cw[4 * u] = cn[0][u].x; cw[4 * u + 1] = cn[0][u].y; cw[4 * u + 2] = cn[0][u].z; cw[4 * u + 3] = cn[0][u].w;
                cn[0][u] = cn[1][u];
            }
            const float sj = sn[0];
            sn[0] = sn[1];
            if (j0 + 2 * Q8_NT < len) {   // (uniform)
                const int jn = min(j + 2 * Q8_NT, len - 1);
                const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)jn * MT);
#pragma unroll
                for (int u = 0; u < MT / 16; u++) cn[1][u] = cp[u];
                sn[1] = ls[jn];
            }
            bool ok = j < len;
            if constexpr (NID) {
                const int64_t id = lid[jc];
                ok = ok && id >= 0;
                if (ok) ok = is_valid_doc(ftab[0], id);
            }
            uint2 t[MT];
#pragma unroll
            for (int m = 0; m < MT; m++) t[m] = q8_gather<MT>(lut, cw[m >> 2], m & 3, m);
            __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the sums
            uint32_t acc[Q8_T] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < MT; m++) {
                acc[0] = add_byte<0>(acc[0], t[m].x);
                acc[1] = add_byte<1>(acc[1], t[m].x);
                acc[2] = add_byte<2>(acc[2], t[m].x);
                acc[3] = add_byte<3>(acc[3], t[m].x);
                acc[4] = add_byte<0>(acc[4], t[m].y);
                acc[5] = add_byte<1>(acc[5], t[m].y);
                acc[6] = add_byte<2>(acc[6], t[m].y);
                acc[7] = add_byte<3>(acc[7], t[m].y);
            }
#pragma unroll
            for (int i = 0; i < Q8_T; i++) {
                const float lhs = __builtin_fmaf(s_nd[i], (float)acc[i], s_A[i] + sj);
                const bool pass = ok && lhs <= s_thr[i];
                const unsigned long long bal = __ballot(pass);
                if (bal) {   // uniform per wave
                    const int nb = __popcll(bal);
                    int base = 0;
                    if (lane == 0) {
                        base = atomicAdd(&s_cur[8], nb);
                        if (base + nb <= POOL) {
                            atomicAdd(&s_n[i], nb);
                            atomicMax(&s_cur[9], base + nb);
                        }
                    }
                    base = __shfl(base, 0, 64);
                    if (base + nb <= POOL) {
                        if (pass) s_pool[base + __popcll(bal & ((1ull << lane) - 1ull))] = ((uint32_t)i << 28) | (uint32_t)j;
                    } else if (pass) {
                        // the pool is full (eight queries next to one long list): straight into the query's list, one
                        // returning atomic per candidate -- rare, and the query keeps its pre-filter
                        const int g = atomicAdd(&ccnt[s_q[i]], 1);
                        if (g < cand_cap) cand[(int64_t)s_q[i] * cand_cap + g] = (uint32_t)(s_pb[i] + j) | ((uint32_t)s_pp[i] << Q8_POS_BITS);
                    }
                }
            }
        }
        __syncthreads();
        // ---- the candidates of each of the tile's queries into its list (one returning atomic per pair) ----
        const int np = s_cur[9];   // pool entries [0, np) are valid (reservations past the end went straight to memory)
        if (tid < nit) {
            const int n = s_n[tid];
            s_g[tid] = (n > 0 && s_q[tid] >= 0) ? atomicAdd(&ccnt[s_q[tid]], n) : 0;
        }
        __syncthreads();
        for (int k = tid; k < np; k += Q8_NT) {
            const uint32_t e = s_pool[k];
            const int i = (int)(e >> 28);
            const int slot = s_g[i] + atomicAdd(&s_cur[i], 1);   // (the order inside a query's list is free)
            if (slot < cand_cap)
                cand[(int64_t)s_q[i] * cand_cap + slot] = (uint32_t)(s_pb[i] + (int)(e & 0x0fffffffu)) | ((uint32_t)s_pp[i] << Q8_POS_BITS);
        }
        __syncthreads();   // the tile's LDS is free
    }
}

// ------------------------------------------------------------------------------------
// The same filter for SHORT lists (a tile = a few hundred codes: one step of 64 per wave), software-pipelined across tiles:
// while tile t is scanned, the bytes of tile t + 1 are on their way into registers, the records of tile t + 2 and the first
// codes of tile t + 1 are requested, the atomic that reserves room for tile t's candidates is issued after the scan and
// its result is only used one tile later (copy-out of tile t - 1 beside the table stores of tile t + 1).  Two barriers per
// tile, no load waited for where it is issued.  Records 4-way, pools and counters 3-way rotated.
// ------------------------------------------------------------------------------------
constexpr int Q8_SL_POOL = 1024;
template <int MT, bool NID>
__global__ __launch_bounds__(Q8_NT) void k_q8_filter_sl(
        const int4* __restrict__ tile_list, const int* __restrict__ tile_first, const int* __restrict__ n_tiles,
        const int* __restrict__ pair_run, const Q8Rec* __restrict__ recs, const uint8_t* __restrict__ q8,
        const uint8_t* __restrict__ codes, const float* __restrict__ sums, const int64_t* __restrict__ ids,
        const FilterDesc* __restrict__ ftab, int need_ids, uint32_t* __restrict__ cand, int* __restrict__ ccnt, int cand_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_q8[];
    constexpr int ROW = Q8Lut<MT>::ROW;
    constexpr int NB = (MT * 32 + Q8_NT - 1) / Q8_NT;   // table blocks per thread
    unsigned char* lut = s_q8;                                                        // [MT][ROW][8]
    Q8Rec* s_m = reinterpret_cast<Q8Rec*>(s_q8 + (size_t)MT * ROW * 8);               // [4][8]
    int* s_n = reinterpret_cast<int*>(s_m + 4 * Q8_T);                                // [3][8] candidates per query
    int* s_g = s_n + 3 * Q8_T;                                                        // [3][8] base in the query's list
    int* s_cp = s_g + 3 * Q8_T;                                                       // [3][8] copied so far
    int* s_np = s_cp + 3 * Q8_T;                                                      // [3] pool counter | [3] valid entries
    uint32_t* s_pool = reinterpret_cast<uint32_t*>(s_np + 8);                         // [3][Q8_SL_POOL]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntile = *n_tiles, G = (int)gridDim.x;
    int tile = blockIdx.x;
    if (tile >= ntile) return;
    auto load_rec = [&](int t) -> Q8Rec {   // threads 0..7: ONE load (records are stored per tile, 8 slots, q = -1 in the unused ones)
        Q8Rec r;
        r.q = -1;
        if (t < ntile && tid < Q8_T) r = recs[(int64_t)t * Q8_T + tid];
        if (r.q < 0) { r.pp = 0; r.pb = 0; r.pad = 0; r.A = 0.f; r.nd = 0.f; r.thr = -INFINITY; r.pad2 = 0.f; }
        return r;
    };
    struct Tab { uint2 r[NB][Q8_T]; };
    auto issue_tables = [&](int ms, Tab& tb) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int bi = tid + b * Q8_NT;
            const int m = bi >> 5, cb = bi & 31;
#pragma unroll
            for (int i = 0; i < Q8_T; i++) {
                const int q = s_m[ms * Q8_T + i].q;
                tb.r[b][i] = (q >= 0 && bi < MT * 32) ? *reinterpret_cast<const uint2*>(q8 + ((int64_t)q * MT + m) * 256 + cb * 8) : make_uint2(0u, 0u);
            }
        }
    };
    auto write_lut = [&](const Tab& tb) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int bi = tid + b * Q8_NT;
            if (bi >= MT * 32) continue;
            const int m = bi >> 5, cb = bi & 31;
            uint32_t xl[4], xh[4], yl[4], yh[4];
            tr4x4(tb.r[b][0].x, tb.r[b][1].x, tb.r[b][2].x, tb.r[b][3].x, xl[0], xl[1], xl[2], xl[3]);
            tr4x4(tb.r[b][4].x, tb.r[b][5].x, tb.r[b][6].x, tb.r[b][7].x, xh[0], xh[1], xh[2], xh[3]);
            tr4x4(tb.r[b][0].y, tb.r[b][1].y, tb.r[b][2].y, tb.r[b][3].y, yl[0], yl[1], yl[2], yl[3]);
            tr4x4(tb.r[b][4].y, tb.r[b][5].y, tb.r[b][6].y, tb.r[b][7].y, yh[0], yh[1], yh[2], yh[3]);
            uint2* dst = reinterpret_cast<uint2*>(lut + ((size_t)m * ROW + Q8Lut<MT>::block(cb)) * 8);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                dst[j] = make_uint2(xl[j], xh[j]);
                dst[4 + j] = make_uint2(yl[j], yh[j]);
            }
        }
    };
    struct Codes { uint4 c[MT / 16]; float s; int len; int64_t off; };
    auto issue_codes = [&](int t, Codes& cd) {
        const int4 ti = tile_list[t];
        cd.len = ti.y;
        cd.off = (int64_t)(((uint64_t)(uint32_t)ti.w << 32) | (uint32_t)ti.z);
        cd.s = 0.f;
        if (wv * 64 < cd.len) {
            const int jc = min(wv * 64 + lane, cd.len - 1);
            const uint4* cp = reinterpret_cast<const uint4*>(codes + (cd.off + jc) * MT);
#pragma unroll
            for (int u = 0; u < MT / 16; u++) cd.c[u] = cp[u];
            cd.s = sums[cd.off + jc];
        }
    };
    auto copy_out = [&](int ps, int ms) {   // the pooled candidates of a finished tile into their queries' lists
        const int np = s_np[3 + ps];
        for (int k = tid; k < np; k += Q8_NT) {
            const uint32_t e = s_pool[ps * Q8_SL_POOL + k];
            const int i = (int)(e >> 28);
            const Q8Rec& r = s_m[ms * Q8_T + i];
            const int slot = s_g[ps * Q8_T + i] + atomicAdd(&s_cp[ps * Q8_T + i], 1);   // (the order inside a query's list is free)
            if (slot < cand_cap) cand[(int64_t)r.q * cand_cap + slot] = (uint32_t)(r.pb + (int)(e & 0x0fffffffu)) | ((uint32_t)r.pp << Q8_POS_BITS);
        }
    };
    // ---- prologue: tile 0's records, bytes and codes; tile 1's records ----
    if (tid < Q8_T) s_m[tid] = load_rec(tile);
    if (tid < 3 * Q8_T) s_n[tid] = s_cp[tid] = 0;
    if (tid < 8) s_np[tid] = 0;
    __syncthreads();
    Tab tb;
    Codes cd;
    {
        issue_tables(0, tb);
        Q8Rec r1 = load_rec(tile + G);
        issue_codes(tile, cd);
        write_lut(tb);
        if (tid < Q8_T) s_m[Q8_T + tid] = r1;
    }
    __syncthreads();
    int g_reg = 0;
    int it = 0;
    for (;; it++) {
        const int ms = it & 3, pc = it % 3, pprev = (it + 2) % 3, pres = (it + 1) % 3;
        const int nxt = tile + G;
        const bool has_n = nxt < ntile;
        Codes cn;
        Q8Rec r2;
        if (has_n) {
            issue_tables((it + 1) & 3, tb);
            issue_codes(nxt, cn);
        }
        if (tid < Q8_T) r2 = load_rec(nxt + G);
        // ---- scan tile `it` ----
        {
            const int len = cd.len;
            const uint8_t* lc = codes + cd.off * MT;
            const float* ls = sums + cd.off;
            const int64_t* lid = ids + cd.off;
            const Q8Rec* mr = s_m + ms * Q8_T;
            for (int j0 = wv * 64; j0 < len; j0 += Q8_NT) {
                const int j = j0 + lane, jc = min(j, len - 1);
                uint32_t cw[MT / 4];
#pragma unroll
                for (int u = 0; u < MT / 16; u++) {
                    cw[4 * u] = cd.c[u].x; cw[4 * u + 1] = cd.c[u].y; cw[4 * u + 2] = cd.c[u].z; cw[4 * u + 3] = cd.c[u].w;
                }
                const float sj = cd.s;
                if (j0 + Q8_NT < len) {   // (uniform) lists beyond 512 codes: the next step's codes
                    const int jn = min(j + Q8_NT, len - 1);
                    const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)jn * MT);
#pragma unroll
                    for (int u = 0; u < MT / 16; u++) cd.c[u] = cp[u];
                    cd.s = ls[jn];
                }
                bool ok = j < len;
                if constexpr (NID) {
                    const int64_t id = lid[jc];
                    ok = ok && id >= 0;
                    if (ok) ok = is_valid_doc(ftab[0], id);
                }
                uint2 t[MT];
#pragma unroll
                for (int m = 0; m < MT; m++) t[m] = q8_gather<MT>(lut, cw[m >> 2], m & 3, m);
                __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the sums
                uint32_t acc[Q8_T] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    acc[0] = add_byte<0>(acc[0], t[m].x);
                    acc[1] = add_byte<1>(acc[1], t[m].x);
                    acc[2] = add_byte<2>(acc[2], t[m].x);
                    acc[3] = add_byte<3>(acc[3], t[m].x);
                    acc[4] = add_byte<0>(acc[4], t[m].y);
                    acc[5] = add_byte<1>(acc[5], t[m].y);
                    acc[6] = add_byte<2>(acc[6], t[m].y);
                    acc[7] = add_byte<3>(acc[7], t[m].y);
                }
#pragma unroll
                for (int i = 0; i < Q8_T; i++) {
                    const float lhs = __builtin_fmaf(mr[i].nd, (float)acc[i], mr[i].A + sj);
                    const bool pass = ok && lhs <= mr[i].thr;
                    const unsigned long long bal = __ballot(pass);
                    if (bal) {   // uniform per wave
                        const int nb = __popcll(bal);
                        int base = 0;
                        if (lane == 0) {
                            base = atomicAdd(&s_np[pc], nb);
                            if (base + nb <= Q8_SL_POOL) {
                                atomicAdd(&s_n[pc * Q8_T + i], nb);
                                atomicMax(&s_np[3 + pc], base + nb);
                            }
                        }
                        base = __shfl(base, 0, 64);
                        if (base + nb <= Q8_SL_POOL) {
                            if (pass) s_pool[pc * Q8_SL_POOL + base + __popcll(bal & ((1ull << lane) - 1ull))] = ((uint32_t)i << 28) | (uint32_t)j;
                        } else if (pass) {   // the pool is full: straight into the query's list
                            const int g = atomicAdd(&ccnt[mr[i].q], 1);
                            if (g < cand_cap) cand[(int64_t)mr[i].q * cand_cap + g] = (uint32_t)(mr[i].pb + j) | ((uint32_t)mr[i].pp << Q8_POS_BITS);
                        }
                    }
                }
            }
        }
        if (it > 0 && tid < Q8_T) s_g[pprev * Q8_T + tid] = g_reg;   // the reservation issued one tile ago has returned long since
        __syncthreads();   // #1: the table and this tile's pool are complete
        if (tid < Q8_T) {
            const int n = s_n[pc * Q8_T + tid], q = s_m[ms * Q8_T + tid].q;
            g_reg = (n > 0 && q >= 0) ? atomicAdd(&ccnt[q], n) : 0;
        }
        if (tid >= 32 && tid < 32 + 3 * Q8_T) {   // the slot tile it + 1 will fill: its previous user (tile it - 2) was copied out
            const int k = tid - 32;               // before the last barrier
            if (k < Q8_T) s_n[pres * Q8_T + k] = 0;
            else if (k < 2 * Q8_T) s_cp[pres * Q8_T + k - Q8_T] = 0;
            else if (k == 2 * Q8_T) s_np[pres] = 0;
            else if (k == 2 * Q8_T + 1) s_np[3 + pres] = 0;
        }
        if (it > 0) copy_out(pprev, (it + 3) & 3);
        if (!has_n) break;
        write_lut(tb);
        if (tid < Q8_T) s_m[((it + 2) & 3) * Q8_T + tid] = r2;
        __syncthreads();   // #2: the next tile's table and records are in place
        tile = nxt;
        cd = cn;
    }
    // ---- epilogue: the last tile's candidates ----
    if (tid < Q8_T) s_g[(it % 3) * Q8_T + tid] = g_reg;
    __syncthreads();
    copy_out(it % 3, it & 3);
}

// ------------------------------------------------------------------------------------
// The candidates' exact values, one workgroup per query (the query's fp32 table in LDS): what the regular loop computes
// for these codes -- fma and adds in the reference's order -- and the query's consumer slice (slice 1 of 2) as the CF
// pass would have written it.
// ------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void k_q8_exact(const float* __restrict__ st2, const float* __restrict__ T2, int nq, int P, int G,
                                                  const int* __restrict__ probe_list, const float* __restrict__ coarse_dis,
                                                  const int64_t* __restrict__ list_off, const uint8_t* __restrict__ codes,
                                                  const int* __restrict__ pair_off, const unsigned long long* __restrict__ ready,
                                                  const uint32_t* __restrict__ cand, const int* __restrict__ ccnt, int cand_cap,
                                                  unsigned long long* __restrict__ surv, int* __restrict__ gcnt, int cnt_stride,
                                                  int slice_cap, const float* __restrict__ fx, const float* __restrict__ pqc, int d,
                                                  const float* __restrict__ xd) {
    __shared__ float s_lut[MT * 256];
    __shared__ int s_cnt[64];   // survivors per probe group (slice pg holds the positions of probe group pg: the order the tie
                                // replay walks the slices in, tie_dev.h)
    const int q = blockIdx.x, tid = threadIdx.x;
    const int ngroups = (P + G - 1) / G;   // == cnt_stride
    const unsigned long long word = ready[q];
    if ((word >> 32) != 1ull) {   // no bound: repair list (k_q8_pairs)
        for (int g = 1 + tid; g < ngroups; g += 256) gcnt[(int64_t)q * cnt_stride + g] = 0;
        return;
    }
    const int n = ccnt[q];
    if (n > cand_cap) {   // more candidates than the list holds: a slice reads as overflowed, k_select_final sends the query
        for (int g = 1 + tid; g < ngroups; g += 256) gcnt[(int64_t)q * cnt_stride + g] = g == 1 ? slice_cap + 1 : 0;   // to the repair launch
        return;
    }
    if (tid < 64) s_cnt[tid] = 0;
    // Few candidates (a list shard sees W times the queries with ~1 / W of their candidates each): the M table entries of a
    // candidate are computed ON DEMAND from the query and the codebook -- n x M x dsub floats from the L2-resident codebook
    // instead of the query's whole 4 M KB table from HBM (k_pq_ip_table's arithmetic: the identical value)
    const bool demand = xd != nullptr && n > 0 && n <= Q8_DEMAND_MAX;
    const bool fused = !demand && n > 0 && fx != nullptr;
    const int dsub = d > 0 ? d / MT : 1;
    if (fused) {   // (opt-in: the query's whole table computed here, k_q8_quant's note)
        const float* xq = fx + (int64_t)q * d;
#pragma unroll 4
        for (int i = 0; i < MT; i++) s_lut[i * 256 + tid] = fvec_ny_row<false>(xq + i * dsub, pqc + ((int64_t)i * 256 + tid) * dsub, dsub);
    } else if (n > 0 && !demand)
        for (int e = tid; e < MT * 256; e += 256) s_lut[e] = st2[(int64_t)q * MT * 256 + e];
    __syncthreads();
    const float* xq = (demand ? xd : fx) ? (demand ? xd : fx) + (int64_t)q * d : nullptr;
    const float tau_f = key2f((uint32_t)word);
    for (int c = tid; c < n; c += 256) {
        const uint32_t cd = cand[(int64_t)q * cand_cap + c];
        const int pos = (int)(cd & ((1u << Q8_POS_BITS) - 1u));
        const int p = (int)(cd >> Q8_POS_BITS), pair = q * P + p;
        const int l = probe_list[pair];
        const int j = pos - pair_off[(int64_t)q * (P + 1) + p];
        const uint8_t* cj = codes + (list_off[l] + j) * MT;
        const float* t2 = T2 + (int64_t)l * MT * 256;
        uint32_t cw[MT / 4];
#pragma unroll
        for (int u = 0; u < MT / 16; u++) {
            const uint4 cv = reinterpret_cast<const uint4*>(cj)[u];
            cw[4 * u] = cv.x; cw[4 * u + 1] = cv.y; cw[4 * u + 2] = cv.z; cw[4 * u + 3] = cv.w;
        }
        float dis = coarse_dis[pair];
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 8) {   // eight table entries in flight at a time
            float a[8];
#pragma unroll
            for (int m = 0; m < 8; m++) a[m] = t2[(m0 + m) * 256 + ((cw[(m0 + m) >> 2] >> (8 * (m & 3))) & 255u)];
#pragma unroll
            for (int m = 0; m < 8; m++) {   // the regular loop's table entry and its adds, in the reference's order
                const uint32_t cb = (cw[(m0 + m) >> 2] >> (8 * (m & 3))) & 255u;
                const float ip = demand ? fvec_ny_row<false>(xq + (m0 + m) * dsub, pqc + ((int64_t)(m0 + m) * 256 + cb) * dsub, dsub)
                                        : s_lut[(m0 + m) * 256 + cb];
                dis += __builtin_fmaf(-2.0f, ip, a[m]);
            }
        }
        if (dis <= tau_f) {
            const int g = p / G;
            const int at = atomicAdd(&s_cnt[g], 1);
            if (at < slice_cap)
                surv[((int64_t)q * cnt_stride + g) * slice_cap + at] = ((unsigned long long)dis_key<true>(dis) << 32) | (unsigned)pos;
        }
    }
    __syncthreads();
    for (int g = 1 + tid; g < ngroups; g += 256) gcnt[(int64_t)q * cnt_stride + g] = s_cnt[g];
}

bool q8_supported(int M, int P, int G, int64_t q_stride) { return (M == 16 || M == 32) && P <= 128 && G >= 1 && /* (the caller checks nlist <= 16384: the per-list counters live in LDS) */ (P + G - 1) / G <= 64 && q_stride < ((int64_t)1 << Q8_POS_BITS); }

// workspace: ccnt nq (zeroed here) | hist Q8_NW x nlist | cnt nlist | off nlist+1 | tile_first nlist+1 | n_tiles 1 | pad |
//            tile_list (int4) | recs (Q8Rec, 32 B)
size_t q8_int_words(int nq, int P, int G, int nlist) {
    const int64_t pairs = (int64_t)nq * (P - G);
    return (size_t)(nq + (int64_t)(Q8_NW + 1) * nlist + 2 * ((int64_t)nlist + 1) + 1 + 8 + 4 * (pairs / Q8_T + nlist + 1) + 8 * Q8_T * (pairs / Q8_T + nlist + 1) + 16);
}

void launch_q8_consumers(hipStream_t s, const Q8Args& a) {
    if (a.nq <= 0 || a.P <= a.G) return;
    const int nq = a.nq, P = a.P, G = a.G, nlist = a.nlist, M = a.M;
    int* ccnt = a.iwork;
    int* hist = ccnt + nq;
    int* cnt = hist + (int64_t)Q8_NW * nlist;
    int* off = cnt + nlist;
    int* tile_first = off + nlist + 1;
    int* n_tiles = tile_first + nlist + 1;
    const int64_t pairs = (int64_t)nq * (P - G);
    uintptr_t pa = (reinterpret_cast<uintptr_t>(n_tiles + 1) + 31) & ~(uintptr_t)31;
    int4* tile_list = reinterpret_cast<int4*>(pa);
    Q8Rec* recs = reinterpret_cast<Q8Rec*>(tile_list + (pairs / Q8_T + nlist + 1));
    (void)hipMemsetAsync(ccnt, 0, (size_t)nq * sizeof(int), s);
    // (records are stored per tile, 8 slots each: the slots past a tile's pairs read q = -1)
    (void)hipMemsetAsync(recs, 0xff, (size_t)(pairs / Q8_T + nlist + 1) * Q8_T * sizeof(Q8Rec), s);
    const int cap = q8_cand_cap(nq);
    if (M == 16) hipLaunchKernelGGL((k_q8_quant<16>), dim3(nq), dim3(256), 0, s, a.st2, a.q8, a.meta, a.fx, a.pqc, a.d);
    else hipLaunchKernelGGL((k_q8_quant<32>), dim3(nq), dim3(256), 0, s, a.st2, a.q8, a.meta, a.fx, a.pqc, a.d);
    const size_t hl = (size_t)nlist * sizeof(int);
    hipLaunchKernelGGL((k_q8_hist<false>), dim3(Q8_NW), dim3(1024), hl, s, a.probe_list, nq, P, G, a.ready, a.list_len, a.list_mask,
                       nlist, hist, (Q8Rec*)nullptr, a.rq_list, a.rq_count, a.coarse_dis, a.t2max, a.meta, a.pair_off, off, tile_first);
    hipLaunchKernelGGL(k_q8_colsum, dim3((nlist + 255) / 256), dim3(256), 0, s, hist, nlist, cnt);
    hipLaunchKernelGGL(k_q8_offsets, dim3(1), dim3(1024), 0, s, cnt, nlist, off, tile_first, n_tiles);
    hipLaunchKernelGGL(k_q8_colfix, dim3((nlist + 255) / 256), dim3(256), 0, s, hist, nlist, off, tile_first, tile_list, a.list_len, a.list_off);
    hipLaunchKernelGGL((k_q8_hist<true>), dim3(Q8_NW), dim3(1024), hl, s, a.probe_list, nq, P, G, a.ready, a.list_len, a.list_mask,
                       nlist, hist, recs, a.rq_list, a.rq_count, a.coarse_dis, a.t2max, a.meta, a.pair_off, off, tile_first);
    // short lists: the pipelined kernel (a tile is a step or two per wave: everything is latency); long lists: the plain loop
    static const double sl_len = getenv("GAMMA_HIP_Q8_SL_LEN") ? atof(getenv("GAMMA_HIP_Q8_SL_LEN")) : 1000.0;
    const bool sl = a.mean_len < sl_len;
    const size_t lut_bytes = (size_t)M * (M <= 16 ? 288 : 256) * 8;
    const size_t lds = sl ? lut_bytes + 4 * Q8_T * sizeof(Q8Rec) + (9 * Q8_T + 8) * sizeof(int) + 3 * (size_t)Q8_SL_POOL * sizeof(uint32_t)
                          : lut_bytes + 11 * 8 * sizeof(int) + (size_t)(M <= 16 ? Q8Pool<16>::N : Q8Pool<32>::N) * sizeof(uint32_t);
    const int per_cu = std::max(1, std::min(8, (int)((160 * 1024) / (lds + 512))));
    static const int grid_env = getenv("GAMMA_HIP_Q8_GRID") ? atoi(getenv("GAMMA_HIP_Q8_GRID")) : 0;
    const unsigned grid = grid_env > 0 ? (unsigned)grid_env : (unsigned)(256 * per_cu);
    static std::atomic<uint64_t> attr{0};   // the attribute is per DEVICE (an in-process group launches this on every member's device)
    if (first_call_on_device(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_q8_filter<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_q8_filter<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_q8_filter_sl<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_q8_filter_sl<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
    }
    (void)hipGetLastError();   // (a stale error of an earlier call is not this pass's)
#define GH_Q8F1(KERN, MM, NN)                                                                                                      \
    hipLaunchKernelGGL((KERN<MM, NN>), dim3(grid), dim3(Q8_NT), lds, s, tile_list, tile_first, n_tiles, off, recs, a.q8, a.codes, \
                       a.sums, a.ids, a.ftab, a.need_ids, a.cand, ccnt, cap)
#define GH_Q8F(KERN, MM)                    \
    do {                                    \
        if (a.need_ids) GH_Q8F1(KERN, MM, true); \
        else GH_Q8F1(KERN, MM, false);      \
    } while (0)
    if (M == 16) {
        if (sl) GH_Q8F(k_q8_filter_sl, 16);
        else GH_Q8F(k_q8_filter, 16);
        hipLaunchKernelGGL((k_q8_exact<16>), dim3(nq), dim3(256), 0, s, a.st2, a.T2, nq, P, G, a.probe_list, a.coarse_dis, a.list_off,
                           a.codes, a.pair_off, a.ready, a.cand, ccnt, cap, a.surv, a.gcnt, a.cnt_stride, a.slice_cap, a.fx, a.pqc, a.d, a.xd);
    } else {
        if (sl) GH_Q8F(k_q8_filter_sl, 32);
        else GH_Q8F(k_q8_filter, 32);
        hipLaunchKernelGGL((k_q8_exact<32>), dim3(nq), dim3(256), 0, s, a.st2, a.T2, nq, P, G, a.probe_list, a.coarse_dis, a.list_off,
                           a.codes, a.pair_off, a.ready, a.cand, ccnt, cap, a.surv, a.gcnt, a.cnt_stride, a.slice_cap, a.fx, a.pqc, a.d, a.xd);
    }
#undef GH_Q8F
#undef GH_Q8F1
    // a launch of this pass that did not start leaves ccnt at 0 and k_q8_exact would publish EMPTY consumer groups: never silent
    if (hipGetLastError() != hipSuccess) launch_refused("launch_q8_consumers: a kernel of the byte-table pass failed to launch");
}

}  // namespace gh
