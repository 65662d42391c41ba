// scan_lm.hip -- list-major scan of the CONSUMER probes (every probe behind a query's first group) for large
// batches: two queries that probe the same list are scored by ONE pass over that list.
//
// Why.  k_ivfpq_scan_pair (kernels.hip) scores one (query, list) pair at a time; per code and sub-quantizer it
// spends one address op, one LDS gather and one add, and at C3 both the vector ALU and the LDS sit at 75-80 % of
// what they can issue -- the kernel is bound by the NUMBER of instructions per (query, code), not by bytes.
// A batch of 16384 queries x 32 probes over 4096 lists sends ~100 queries to every list.  Here the look-up
// table holds TWO queries' entries side by side -- LUT2[m][j] = (lut_a, lut_b), 8 bytes -- so one SDWA address
// op, one ds_read_b64 and one v_pk_add_f32 (two independent IEEE adds: each query's chain stays sequential in
// m, the reference's order, index/impl/gamma_index_ivfpq.h:591-597) serve both queries: half the instructions
// per (query, code).  The list's T2 row is fetched once per two pairs, and because work is ordered by list
// inside a block of queries the rows hit in the XCD's L2 (the query-major kernel streams ~1.6 x the code bytes
// in T2 rows from the fabric).
//
// Shape of the stage (gamma_hip_search.cpp, ivfpq_stage_a):
//   1. k_ivfpq_scan_pair, producers only: every query's FIRST probe group, bound tau[q], its own survivors
//   2. k_lm_units: per block of LM_B queries (in the scan's spatial query order) the consumer pairs are
//      sorted by list and cut into units of two pairs (one when a list is probed an odd number of times);
//      a unit record carries everything the scan needs (list extent, both queries, their dis0, slab
//      offsets and bounds): one 64-byte load per unit
//   3. k_scan_lm: persistent workgroups, XCD x sweeps the blocks x, x + 8, ... in order; per unit: LUT2 built
//      with ds_write_addtid_b32 (lane parity = query), the list scanned once, candidates within a query's
//      bound appended to that PAIR's own mini-slice of the query's survivor area (plain stores, no atomics;
//      a pair with more than LM_PAIR_CAP survivors marks the query for the repair path)
//   4. k_select_final over slice 0 (the producer's) + the P - G pair slices.
// Results are those of the query-major path: same ADC values (same fma, same add order), same bounds.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "block_utils.h"
#include "device_math.h"
#include "filter_dev.h"
#include "kernels.h"
#include "scan_dev.h"

namespace gh {

namespace {
constexpr int LM_KEYS = 4096;     // consumer pairs of one query block (sorted in LDS)
constexpr int LM_REC = 16;        // ints per unit record (64 bytes)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
}  // namespace

int lm_pair_cap() { return LM_PAIR_CAP; }
int lm_block_queries(int P, int G) {
    int b = 128;
    while (b > 1 && b * (P - G) > LM_KEYS) b >>= 1;
    return b;
}
int lm_units_per_block() { return LM_KEYS; }

// ------------------------------------------------------------------------------------
// k_lm_units: one workgroup per query block.  keys = (list << 32 | pair index) of the block's consumer pairs,
// bitonic sort in LDS, runs of equal lists cut into units of two.
// record: [0] list [1] len [2,3] arena offset [4] qa [5] qb (-1: single) [6] pbase_a [7] pbase_b
//         [8] dis0_a [9] dis0_b [10] tau_a [11] tau_b (keys; valid iff flag) [12] flags: 1 bound_a, 2 bound_b
//         [13] slice index of a (probe - G) [14] slice index of b
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lm_units(const int* __restrict__ probe, const float* __restrict__ dis0,
                                                  const int* __restrict__ pair_off,
                                                  const int64_t* __restrict__ list_off,
                                                  const int* __restrict__ list_len,
                                                  const uint8_t* __restrict__ list_mask, int nlist,
                                                  const int* __restrict__ qperm,
                                                  const unsigned long long* __restrict__ ready, int nq, int P, int G,
                                                  int B, int* __restrict__ units, int* __restrict__ ucount) {
    __shared__ unsigned long long s_key[LM_KEYS];
    __shared__ int s_scan[256];
    __shared__ int s_tot;
    const int tid = threadIdx.x, blk = blockIdx.x;
    const int PC = P - G;
    const int nk = B * PC;   // <= LM_KEYS
    int npow = 2;
    while (npow < nk) npow <<= 1;
    for (int i = tid; i < npow; i += 256) {
        unsigned long long key = ~0ull;
        if (i < nk) {
            const int qi = blk * B + i / PC, p = G + i % PC;
            if (qi < nq) {
                const int q = qperm ? qperm[qi] : qi;
                const int l = probe[(int64_t)q * P + p];
                if (l >= 0 && l < nlist && (!list_mask || list_mask[l]) && list_len[l] > 0)
                    key = ((unsigned long long)(unsigned)l << 32) | (unsigned)(q * P + p);
            }
        }
        s_key[i] = key;
    }
    __syncthreads();
    // bitonic sort, ascending (invalid keys sink to the end)
    for (int size = 2; size <= npow; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (npow >> 1); i += 256) {
                const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1)), hi = lo | stride;
                const unsigned long long a = s_key[lo], b = s_key[hi];
                const bool up = (lo & size) == 0;
                if ((a > b) == up) {
                    s_key[lo] = b;
                    s_key[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    // unit leaders: even rank inside the run of its list.  Each thread owns npow / 256 consecutive keys.
    const int per = npow >> 8 > 0 ? npow >> 8 : 1;
    const int i0 = tid * per;
    // rank parity inside the run: a running parity that resets at run heads, carried across threads by a
    // (parity, closed) pair -- simpler: every thread walks back to its first key's run head (runs are short:
    // a list is probed by a handful of the block's queries)
    int par = 0;
    if (i0 < npow && s_key[i0] != ~0ull) {
        int j = i0;
        const uint32_t l0 = (uint32_t)(s_key[i0] >> 32);
        while (j > 0 && (uint32_t)(s_key[j - 1] >> 32) == l0) j--;
        par = (i0 - j) & 1;
    }
    int nlead = 0;
    {
        int pp = par;
        for (int i = i0; i < i0 + per && i < npow; i++) {
            const unsigned long long k = s_key[i];
            if (k == ~0ull) break;
            if (i > i0 && (uint32_t)(s_key[i - 1] >> 32) != (uint32_t)(k >> 32)) pp = 0;
            nlead += pp == 0 ? 1 : 0;
            pp ^= 1;
        }
    }
    // exclusive prefix of nlead over the 256 threads
    s_scan[tid] = nlead;
    __syncthreads();
    if (tid < 64) {
        int v[4], sum = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            v[u] = s_scan[tid * 4 + u];
            sum += v[u];
        }
        const int incl = wave_incl_scan(sum);
        int run = incl - sum;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s_scan[tid * 4 + u] = run;
            run += v[u];
        }
        if (tid == 63) s_tot = incl;
    }
    __syncthreads();
    int uo = s_scan[tid];
    if (tid == 0) ucount[blk] = s_tot;
    int* ub = units + (int64_t)blk * LM_KEYS * LM_REC;
    int pp = par;
    for (int i = i0; i < i0 + per && i < npow; i++) {
        const unsigned long long k = s_key[i];
        if (k == ~0ull) break;
        const uint32_t l = (uint32_t)(k >> 32);
        if (i > i0 && (uint32_t)(s_key[i - 1] >> 32) != l) pp = 0;
        if (pp == 0) {
            const int pa = (int)(uint32_t)k;
            int pb = -1;
            if (i + 1 < npow && s_key[i + 1] != ~0ull && (uint32_t)(s_key[i + 1] >> 32) == l) pb = (int)(uint32_t)s_key[i + 1];
            const int qa = pa / P, ia = pa % P;
            const int qb = pb >= 0 ? pb / P : -1, ib = pb >= 0 ? pb % P : 0;
            const unsigned long long wa = ready[qa], wb = pb >= 0 ? ready[qb] : 0ull;
            int* r = ub + (int64_t)uo * LM_REC;
            const int64_t off = list_off[l];
            r[0] = (int)l;
            r[1] = list_len[l];
            r[2] = (int)(uint32_t)off;
            r[3] = (int)(off >> 32);
            r[4] = qa;
            r[5] = qb;
            r[6] = pair_off[(int64_t)qa * (P + 1) + ia];
            r[7] = pb >= 0 ? pair_off[(int64_t)qb * (P + 1) + ib] : 0;
            r[8] = __float_as_int(dis0[pa]);
            r[9] = pb >= 0 ? __float_as_int(dis0[pb]) : 0;
            r[10] = (int)(uint32_t)wa;
            r[11] = (int)(uint32_t)wb;
            r[12] = ((wa >> 32) == 1ull ? 1 : 0) | ((pb >= 0 && (wb >> 32) == 1ull) ? 2 : 0);
            r[13] = ia - G;
            r[14] = ib - G;
            r[15] = 0;
            uo++;
        }
        pp ^= 1;
    }
}

void launch_lm_units(hipStream_t s, const int* probe, const float* dis0, const int* pair_off,
                     const int64_t* list_off, const int* list_len, const uint8_t* list_mask, int nlist,
                     const int* qperm, const unsigned long long* ready, int nq, int P, int G, int B, int* units,
                     int* ucount) {
    if (nq <= 0) return;
    const int nblk = (nq + B - 1) / B;
    hipLaunchKernelGGL(k_lm_units, dim3(nblk), dim3(256), 0, s, probe, dis0, pair_off, list_off, list_len, list_mask,
                       nlist, qperm, ready, nq, P, G, B, units, ucount);
}

// ------------------------------------------------------------------------------------
// k_scan_lm: see the header.  Dynamic LDS: LUT2 (MT * 256 float2) at address 0.
// ------------------------------------------------------------------------------------
// (byte << 3) + 2048 * m: entry (m, code byte) of LUT2
__device__ __forceinline__ f32x2 lut2_gather(uint32_t w, int k, int m) {
    uint32_t a;
    switch (k) {   // constant after unrolling
        case 0: asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(w)); break;
        case 1: asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(w)); break;
        case 2: asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(w)); break;
        default: asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a) : "v"(w)); break;
    }
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((uintptr_t)(a + 2048u * (uint32_t)m));
}

// ---- software pipeline --------------------------------------------------------------------------------------
// A unit is short (two ~16 KB table rows in, a few hundred codes scored): run naively it is a chain of dependent
// memory round trips -- record, then table rows and codes, then the stores -- of ~5 us for ~1 us of work, and the
// 32 KB LUT2 leaves only four workgroups per CU to hide it.  So every workgroup keeps ONE UNIT IN FLIGHT: the
// table entries (32 of the queries' st2 rows + 32 of the list's T2 row per thread) and the first codes of unit
// n + 1 are requested right after the LUT2 of unit n has been written, and arrive while unit n is scanned; the
// record of unit n + 1 is fetched one step earlier still.  The loads are asm statements (hipcc would sink an
// ordinary load to its first use, a whole scan later); their registers are "produced" by the wait statements
// of the next step, which is what keeps the compiler from touching them in between.
// 16-byte loads: a dword load moves 256 bytes per wave instruction through the texture addresser, and with 64 of
// them per thread and unit (the first version of this kernel) the addresser, not the LDS or the ALUs, was the
// bound (3200 cycles per unit and CU).  Thread t owns the four consecutive entries 4 (t + 256 c) .. + 3 of chunk
// c = 0..3: three 16-byte loads per chunk (query a's row, query b's row, the list's T2 row).
// The prefetch loads are VOLATILE: hipcc keeps a volatile access where it is written (an ordinary load is sunk to
// its first use, a whole scan later) and still tracks it as a pending load -- it waits before it reads or copies
// the destination registers.  (Loads issued from asm statements with the wait in a later asm statement are not
// safe here: the compiler may move their destination registers around in between, before the data has landed.)
// (global address space spelled out: a volatile load through a generic pointer becomes a FLAT load, which also
//  counts against lgkmcnt and would serialise the scan's LDS waits)
typedef const volatile __attribute__((address_space(1))) f32x4 gvol_f4;
typedef const volatile __attribute__((address_space(1))) u32x4 gvol_u4;
__device__ __forceinline__ f32x4 lm_ld16(const float* p) { return *reinterpret_cast<gvol_f4*>((uintptr_t)p); }

struct LmRec {   // one unit, all uniform
    int l, len;
    int64_t off;
    int qa, qb;          // qb == qa for a single
    bool has_b, bound_a, bound_b;
    int pbase_a, pbase_b, sl_a, sl_b;
    float dis0a, dis0b;
    uint32_t ka, kb;     // bound keys
};
__device__ __forceinline__ LmRec lm_load_rec(const int* __restrict__ r) {
    const int4* r4 = reinterpret_cast<const int4*>(r);
    const int4 w0 = r4[0], w1 = r4[1], w2 = r4[2], w3 = r4[3];
    LmRec c;
    c.l = w0.x;
    c.len = w0.y;
    c.off = (int64_t)(uint32_t)w0.z | ((int64_t)w0.w << 32);
    c.qa = w1.x;
    c.has_b = w1.y >= 0;
    c.qb = c.has_b ? w1.y : w1.x;
    c.pbase_a = w1.z;
    c.pbase_b = w1.w;
    c.dis0a = __int_as_float(w2.x);
    c.dis0b = __int_as_float(w2.y);
    c.ka = (uint32_t)w2.z;
    c.kb = (uint32_t)w2.w;
    c.bound_a = (w3.x & 1) != 0;
    c.bound_b = (w3.x & 2) != 0;
    c.sl_a = w3.y;
    c.sl_b = w3.z;
    return c;
}

template <bool L2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void k_scan_lm(LmScanArgs a) {
    constexpr int MT = 16, msz = MT * 256;
    extern __shared__ float s_lut2[];   // [MT][256][2] at LDS address 0, then the two survivor counters
    int* s_cnt = reinterpret_cast<int*>(s_lut2 + 2 * msz);
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, wid = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const float sentinel = L2 ? INFINITY : -INFINITY;
    const FilterDesc& filt = a.ftab[0];
    const int nblk = (a.nq + a.B - 1) / a.B;
    // this workgroup's units: XCD x sweeps the blocks x, x + 8, ...; its workgroups deal the units round robin,
    // carrying the stride over block ends
    int blk = xcd, u = wid, nu = blk < nblk ? a.ucount[blk] : 0;
    auto settle = [&]() {   // -> is (blk, u) a unit?
        while (blk < nblk && u >= nu) {
            u -= nu;
            blk += 8;
            nu = blk < nblk ? a.ucount[blk] : 0;
        }
        return blk < nblk;
    };
    if (!settle()) return;
    LmRec cur = lm_load_rec(a.units + ((int64_t)blk * LM_KEYS + u) * LM_REC);
    f32x4 ra[4], rb[4], rt[4];   // st2[qa], st2[qb], T2[l]: entries 4 (tid + 256 c) .. + 3
    u32x4 cn;
    auto issue = [&](const LmRec& c) {
        const float* sa = a.st2 + (int64_t)c.qa * msz + 4 * tid;
        const float* sb2 = a.st2 + (int64_t)c.qb * msz + 4 * tid;
        const float* tb = a.T2 + (int64_t)c.l * msz + 4 * tid;
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
            ra[ch] = lm_ld16(sa + 1024 * ch);
            rb[ch] = lm_ld16(sb2 + 1024 * ch);
            if (L2) rt[ch] = lm_ld16(tb + 1024 * ch);
        }
        cn = *reinterpret_cast<gvol_u4*>((uintptr_t)(a.codes + c.off * MT + (int64_t)min(tid, c.len - 1) * MT));
    };
    issue(cur);
    bool have = true;
    while (have) {
        // the record after this one (consumed a whole step later)
        u += wpx;
        const bool have_next = settle();
        LmRec nxt = cur;
        if (have_next) nxt = lm_load_rec(a.units + ((int64_t)blk * LM_KEYS + u) * LM_REC);
        __syncthreads();   // the previous unit's gathers and survivor counts are done with
        if (tid < 2) s_cnt[tid] = 0;
        // ---- LUT2 of the current unit from the registers requested a step ago ----
        const u32x4 ccur = cn;
        {
            // LUT2[e] = (lut_a, lut_b): two 16-byte stores per chunk (entries e0, e0 + 1 | e0 + 2, e0 + 3)
            typedef __attribute__((address_space(3))) f32x4 lds_f4;
            lds_f4* dst = reinterpret_cast<lds_f4*>((uintptr_t)((uint32_t)tid * 32u));
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                f32x4 la, lb;
                if (L2) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        la[k] = __builtin_fmaf(-2.0f, ra[ch][k], rt[ch][k]);
                        lb[k] = __builtin_fmaf(-2.0f, rb[ch][k], rt[ch][k]);
                    }
                } else {
                    la = ra[ch];
                    lb = rb[ch];
                }
                dst[512 * ch] = f32x4{la[0], lb[0], la[1], lb[1]};       // 8192 bytes per chunk = 512 float4
                dst[512 * ch + 1] = f32x4{la[2], lb[2], la[3], lb[3]};
            }
        }
        __syncthreads();
        if (have_next) issue(nxt);   // in flight during the scan below
        // ---- scan: one code per thread and step, both queries per look-up ----
        {
            const LmRec& c = cur;
            const float tau_a = key2f(L2 ? c.ka : ~c.ka), tau_b = key2f(L2 ? c.kb : ~c.kb);
            // distances are stored where something reads them: a query without a bound (unfiltered selection),
            // or everything in exact-ties mode (the replay walks the slab)
            const bool store_a = !c.bound_a || a.store_all, store_b = c.has_b && (!c.bound_b || a.store_all);
            const bool keep_b = c.has_b && c.bound_b;
            const uint8_t* lc = a.codes + c.off * MT;
            const int64_t* lid = a.ids + c.off;
            float* oa = a.out + (int64_t)c.qa * a.q_stride + c.pbase_a;
            float* ob = a.out + (int64_t)c.qb * a.q_stride + c.pbase_b;
            unsigned long long* sva = a.surv + ((int64_t)c.qa * a.nslc + c.sl_a) * LM_PAIR_CAP;
            unsigned long long* svb = a.surv + ((int64_t)c.qb * a.nslc + c.sl_b) * LM_PAIR_CAP;
            for (int j0 = 0; j0 < c.len; j0 += 256) {
                const int j = j0 + tid;
                if (j < c.len) {
                    uint32_t cw[4];
                    if (j0 == 0) {
#pragma unroll
                        for (int i = 0; i < 4; i++) cw[i] = ccur[i];
                    } else {
                        const u32x4 cv = *reinterpret_cast<const u32x4*>(lc + (int64_t)j * MT);
#pragma unroll
                        for (int i = 0; i < 4; i++) cw[i] = cv[i];
                    }
                    bool ok = true;
                    if (a.need_ids) {
                        const int64_t id = lid[j];
                        ok = id >= 0;
                        if (ok) ok = is_valid_doc(filt, id);
                    }
                    f32x2 acc = f32x2{c.dis0a, c.dis0b};
#pragma unroll
                    for (int m0 = 0; m0 < MT; m0 += 8) {
                        f32x2 t[8];
#pragma unroll
                        for (int m = 0; m < 8; m++) t[m] = lut2_gather(cw[(m0 + m) >> 2], (m0 + m) & 3, m0 + m);
                        __builtin_amdgcn_sched_barrier(0);   // the gathers in flight before the add chain
#pragma unroll
                        for (int m = 0; m < 8; m++) acc = acc + t[m];   // per component sequential: the reference's order
                    }
                    const float va = ok ? acc.x : sentinel, vb = ok ? acc.y : sentinel;
                    if (store_a) oa[j] = va;
                    if (store_b) ob[j] = vb;
                    // survivors: slot from an LDS counter per query (a few per unit), plain stores
                    if (c.bound_a && (L2 ? va <= tau_a : va >= tau_a)) {
                        const int slot = atomicAdd(&s_cnt[0], 1);
                        if (slot < LM_PAIR_CAP) sva[slot] = ((unsigned long long)dis_key<L2>(va) << 32) | (unsigned)(c.pbase_a + j);
                    }
                    if (keep_b && (L2 ? vb <= tau_b : vb >= tau_b)) {
                        const int slot = atomicAdd(&s_cnt[1], 1);
                        if (slot < LM_PAIR_CAP) svb[slot] = ((unsigned long long)dis_key<L2>(vb) << 32) | (unsigned)(c.pbase_b + j);
                    }
                }
            }
            __syncthreads();
            if (tid == 0) a.cnt[(int64_t)c.qa * a.cnt_stride + 1 + c.sl_a] = s_cnt[0];
            if (tid == 1 && c.has_b) a.cnt[(int64_t)c.qb * a.cnt_stride + 1 + c.sl_b] = s_cnt[1];
        }
        cur = nxt;
        have = have_next;
    }
}

void launch_scan_lm(hipStream_t s, bool l2, int M, const LmScanArgs& a) {
    if (a.nq <= 0) return;
    if (M != 16) abort();   // callers gate on this
    const size_t lds = (size_t)M * 256 * 2 * sizeof(float) + 16;
    // persistent workgroups: as many as are resident (LDS: four per CU), a multiple of the 8 XCDs
    const int grid = 256 * 4;
#define GH_LM(LL)                                                                                     \
    do {                                                                                              \
        static bool attr = false;                                                                     \
        if (!attr) {                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_lm<LL>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
            attr = true;                                                                              \
        }                                                                                             \
        hipLaunchKernelGGL((k_scan_lm<LL>), dim3(grid), dim3(256), lds, s, a);                        \
    } while (0)
    if (l2) GH_LM(true);
    else GH_LM(false);
#undef GH_LM
}

}  // namespace gh
