// scan.hip -- a4 + a5 + a6 + a8: the IVFPQ inverted-list scan (the roofline kernel), its threshold pre-filter and the
// filter pass of the L2 consumers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <utility>

#include "block_utils.h"
#include "device_math.h"
#include "filter_dev.h"
#include "kernels.h"
#include "rerank_dev.h"
#include "scan_dev.h"

namespace gh {

// phase clocks of the bounded-scan kernel (build with -DGH_SCAN_TIMING; the launcher prints the sums every 16th launch)
#ifdef GH_SCAN_TIMING
__device__ unsigned long long g_scan_t[32];
#define GH_ST(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define GH_ST_ADD(i, a, b) do { if (threadIdx.x == 0 && (blockIdx.x & 127) < 8) atomicAdd(&g_scan_t[i], (b) - (a)); } while (0)
#define GH_ST_CNT(i) do { if (threadIdx.x == 0 && (blockIdx.x & 127) < 8) atomicAdd(&g_scan_t[i], 1ull); } while (0)
#else
#define GH_ST(var)
#define GH_ST_ADD(i, a, b)
#define GH_ST_CNT(i)
#endif

// ------------------------------------------------------------------------------------
// a4+a5+a6+a8: IVFPQ list scan, one workgroup per (query, probe) pair.
//   LUT (M x 256 fp32) built in LDS:  L2: lut = T2[list] + (-2) * st2[q]  (fvec_madd)
//                                     IP: lut = st2[q]
//   dis0: L2 = coarse distance; IP = <x_q, centroid> in fvec_inner_product order.
//   per code j:  skip if ids[j] bit 63 / !IsValid;  dis = dis0; for m: dis += lut[m][code[m]]
//   (sequential fp32 adds, gamma_index_ivfpq.h:591-597).  Distances go to the pair's slot
//   range in out; filtered entries get the sentinel.
// Codes are AoS [len][M] exactly as the reference stores them; a 16-byte code is one
// dwordx4 load per lane, so a wave reads 1 KiB contiguous.
// ------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int SCAN_STAGE = 256;                  // survivors staged in LDS per workgroup
constexpr int SCAN_SLICE = 1024;                 // candidate slice of one workgroup (global) up to recall_num 256; a producer needs recall_num +
                                                 // one histogram bin: 2048 beyond (ScanBound::slice_cap, scan_slice_cap)
constexpr int SCAN_BATCH = 64;                   // queries per XCD by which producers run ahead

// amdgpu_num_sgpr(96): 8 waves per SIMD need <= 96 SGPRs each (800 per SIMD); the FILT variant
// would otherwise take 100 and lose one of the eight resident workgroups per CU
// IPF (sharded search with every probe of a query in ONE workgroup): the query's table <x_q,m , c_mj> is
// computed here from the PQ codebook (128 KB, L2 resident; `st2` then points at it) instead of being
// written to HBM by k_pq_ip_table and read back -- with W shards that table is W x 16 KB per query of
// traffic that does not shrink with the shard, and each of its entries would be read exactly once.
// CF (L2, FILT, MT 16 / 32, large batches): the consumer groups of a query with a bound run a FILTER pass without the
// per-list table -- see "filter pass" in the body.
constexpr int SCAN_CF_CAP = 768;   // filter-pass candidates staged per workgroup (8 bytes each)
constexpr int C8_CAND = 1536;      // byte-table pass: its candidates sit in the 12 KB of the fp32 table's place that the bytes leave free (M = 16)
// PCF (with CF): the producer runs on the filter pass's arithmetic too (ScanBound::prod_cf) -- a variant of its own: the extra
// path costs the plain filter-pass kernel 16 VGPRs (73 -> 89: six -> five waves per SIMD) even when it is not taken
// C8 (with CF): the filter pass gathers BYTES (ScanBound::c8; "byte table" in the body).
// RES (L2 table mode 0, round 6): the index has NO precomputed table -- it would exceed faiss's precomputed_table_max_bytes
// (faiss:IndexIVFPQ.cpp:441-449) -- and the reference scores every (query, list) pair with the distance table of the RESIDUAL:
// r = x_q - centroid_l, lut[m][j] = fvec_L2sqr_ny(r_m, c_mj), dis0 = 0 (index/impl/gamma_index_ivfpq.h:239-245).  Here `st2`
// points at the PQ codebook and `T2` is null; the residual sits in LDS behind the staging words.
// PC8 (with C8, round 6): the producer scores its group on the byte image as well (ScanBound::prod_c8), see "producer on the byte image"
constexpr int PC8_MAXN = 3072;     // codes of a producer's group whose lower estimates fit the 12 KB behind the byte image (M = 16)
constexpr int PC8_MAXG = 8;        // lists of the estimate group
// LDS byte offset of the residual image (M = 16): behind the table's place, the survivor stage, the 16 words, the list counter's 16 bytes
// and the estimate group's lists -- the dynamic buffer is all of the kernel's LDS and starts at address 0 (see lut_gather)
constexpr int PC8_IMG2_OFF = 16 * 1024 + SCAN_STAGE * 8 + 64 + 16 + PC8_MAXG * 24;
template <bool L2, int MT, bool FILT, bool IPF, bool UNITS, bool CF, bool PCF, bool C8 = false, bool RES = false, bool PC8 = false>
__device__ __forceinline__ void scan_pair_body(

        const float* __restrict__ x, int nq, int d, int M, int P, int G, const int* __restrict__ probe_list,
        const float* __restrict__ coarse_dis, const float* __restrict__ cc,
        const float* __restrict__ st2, const float* __restrict__ T2,
        const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
        const uint8_t* __restrict__ list_mask, int nlist, const uint8_t* __restrict__ codes,
        const int64_t* __restrict__ ids, const int* __restrict__ pair_off, int64_t q_stride,
        float* __restrict__ out, const FilterDesc* __restrict__ ftab, const int* __restrict__ qfil, int need_ids,
        float sentinel, const int* __restrict__ qperm, int pg_lo, int pg_cnt, int sparse, ScanBound sb,
        const int* __restrict__ rq_list, const int* __restrict__ rq_count, int chunk_len) {
    // UNITS (small batches over long lists, G = 1, no bound; a variant of its own so that the bulk kernel carries
    // none of its state): rq_list is a work list of
    // (query << 20 | probe << 13 | chunk) units written by k_small_coarse_select, one per chunk_len codes of a
    // (query, probe) pair, walked by a fixed grid -- one query's 64 long lists are then a few hundred pieces of
    // even size instead of 64 workgroups that run for as long as the longest list takes.
    // This launch covers probe groups [pg_lo, pg_lo + pg_cnt) of every query.
    // FILT (pg_lo = 0, pg_cnt >= 2): threshold pre-filter.  The workgroup of a query's FIRST probe
    // group (its nearest lists) ends by bounding the query's recall_num-th best distance from above
    // with the candidates it has just scored, and publishes that bound; the workgroups of the other
    // groups append every candidate within the bound, as a (key, position) item, to the query's
    // short survivor list.  The exact top-recall_num is then selected from the first group's
    // candidates within the bound plus a few hundred survivors, instead of all ~10^4 candidates
    // (select.hip, k_select_final).
    // One workgroup scans G consecutive probes of one query: the query's 16 KB table st2 is
    // read ONCE into registers (MT per thread) and reused for the G list-specific LUTs, so
    // the per-pair table traffic drops from 2 x M KB to (1 + 1/G) x M KB.
    extern __shared__ float s_lut[];  // M*256
    // XCD-aware placement (speed only): block b runs on XCD b % 8 with its own L2, so all PGN
    // workgroups of one query are given block ids with the same residue -- the query's table
    // st2[q] is then fetched from HBM/MALL once per XCD and served from that L2 afterwards.
    // With qperm (queries sorted by the spatial rank of their nearest list, k_query_order) XCD x
    // takes the x-th contiguous eighth of that order, in order: concurrently running queries
    // probe overlapping lists, so the 16 KB T2 rows they stream are mostly L2 hits as well.
    const int xcd = blockIdx.x & 7;
    int slot = blockIdx.x >> 3, pg, qslot;
    if (FILT) {
        // Block order inside an XCD (slot = XCD-local index): producers (group 0, they publish
        // the bounds) run one batch of SCAN_BATCH queries AHEAD of the consumers (other groups):
        //   P(0) | P(1) C(0) | P(2) C(1) | ...
        // Workgroups are dispatched in id order, so a consumer only ever waits for a producer that
        // is already resident or finished -- and it starts >= SCAN_BATCH * pg_cnt dispatches after
        // its producer, by when the bound is normally there.  The query's table st2[q] is still in
        // this XCD's L2 when its consumers arrive.
        const int nq8 = (nq + 7) >> 3;
        const int SB = sb.batch > 0 ? sb.batch : SCAN_BATCH;
        if (slot < SB) {
            pg = 0;
            qslot = slot;
        } else {
            const int s2 = slot - SB, period = SB * pg_cnt;
            const int t = s2 / period, r = s2 % period;
            if (r < SB) {
                pg = 0;
                qslot = (t + 1) * SB + r;
            } else {
                const int i = r - SB;
                pg = 1 + i % (pg_cnt - 1);
                qslot = t * SB + i / (pg_cnt - 1);
            }
        }
        if (qslot >= nq8) return;
    } else {
        pg = pg_lo + slot % pg_cnt;
        qslot = slot / pg_cnt;
    }
#ifdef GH_SCAN_TIMING
    if constexpr (CF) {
        if (sb.dbg_part && (sb.dbg_part == 1) == (pg > 0)) return;   // timing experiments (ScanBound::dbg_part)
    }
#endif
    if constexpr (FILT) {
        if (sb.part && (sb.part == 1) == (pg > 0)) return;   // two-phase shard search (ScanBound::part)
    }
    const bool repair = !FILT && rq_list != nullptr;
    int q = 0;
    if (repair) {
        // repair launch (launch_ivfpq_scan_repair): a fixed grid walks the (query, probe group) items of the
        // queries k_select_final could not finish from their survivor slices -- consumer groups with a bound
        // do not store distances (finish() below) -- and scores those groups again, storing everything
    } else if (qperm) {
        const int qi = xcd * ((nq + 7) >> 3) + qslot;
        if (qi >= nq) return;
        q = qperm[qi];
    } else {
        q = qslot * 8 + xcd;
        if (q >= nq) return;
    }
    // ALL of the kernel's LDS is the dynamic buffer, the LUT first: its LDS address is then the constant 0 and a
    // gather address is just (code byte << 2) + an immediate offset (one VALU op per look-up instead of two)
    unsigned long long* s_stage = reinterpret_cast<unsigned long long*>(s_lut + M * 256);   // [SCAN_STAGE]
    int& s_nstage = *reinterpret_cast<int*>(s_stage + SCAN_STAGE);
    uint32_t& s_tau = *(reinterpret_cast<uint32_t*>(s_stage + SCAN_STAGE) + 1);
    uint32_t* s_red = reinterpret_cast<uint32_t*>(s_stage + SCAN_STAGE) + 2;                  // [12]
    int& s_ncand = *(reinterpret_cast<int*>(s_stage + SCAN_STAGE) + 14);                       // CF: staged candidates
    uint2* s_cand = reinterpret_cast<uint2*>(reinterpret_cast<int*>(s_stage + SCAN_STAGE) + 16);  // CF: [SCAN_CF_CAP]
    float* s_res = reinterpret_cast<float*>(reinterpret_cast<int*>(s_stage + SCAN_STAGE) + 16);   // RES (never with CF): [d]
    static_assert(!RES || (L2 && !CF && !IPF && !PCF && !C8), "residual tables: the plain L2 loop only");
    static_assert(!PC8 || (C8 && MT == 16), "producer on the byte image: the M = 16 byte-image kernel");
    int cbase = 0;     // unit mode: first code of the unit within its list
    int lut_q = -1;    // unit mode, inner product: the query whose table is in LDS
    int lut_pair = -1; // unit mode, L2: the (query, probe) pair whose table is in LDS
    auto body = [&](const int q, const int pg) {
    GH_ST(t_start);
    // validity predicates of THIS query: entry qfil[q] of the call's filter table (one entry unless the
    // call is a combined batch of requests with their own filters); only read when need_ids
    const FilterDesc& filt = ftab[(need_ids && qfil) ? qfil[q] : 0];
    const int lane = threadIdx.x & 63;
    // Survivors are staged in LDS (one LDS atomic per wave and iteration) and flushed to the
    // query's list with ONE global atomic per workgroup; a returning global atomic per wave
    // iteration would put ~1 us of latency into the scan loop.  All lanes of a wave call append().
    uint32_t tauq = 0xffffffffu;
    float tau_f = sentinel;   // the bound as a distance: a candidate survives iff it is not worse than tau_f
    bool bound_on = false;
    // producer (pg == 0): range and count of its valid distances.  Kept as floats (one min, one max per code
    // instead of a key conversion and two compare-selects), turned into keys once at the end.
    float g_fmn = INFINITY, g_fmx = -INFINITY;
    int g_nv = 0;
    auto within = [&](float val) -> bool { return L2 ? val <= tau_f : val >= tau_f; };
    auto append = [&](bool keep, float val, int pos) {
        const unsigned long long bal = __ballot(keep);
        if (bal) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_nstage, __popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);   // (lane 0 holds it: no LDS permute)
            if (keep) {
                const int at = base + __popcll(bal & ((1ull << lane) - 1ull));
                const unsigned long long item = ((unsigned long long)dis_key<L2>(val) << 32) | (unsigned)pos;
                if (at < SCAN_STAGE) s_stage[at] = item;
                else if (at < sb.slice_cap)   // staging full (rare): the slot number is already unique
                    sb.surv[((int64_t)q * sb.cnt_stride + pg) * sb.slice_cap + at] = item;
            }
        }
    };
    // every consumer workgroup owns one fixed slice of its query's survivor list: no global
    // atomics, the count (> SCAN_SLICE = overflowed) is a plain store
    auto flush = [&]() {   // whole workgroup
        __syncthreads();
        const int n = s_nstage;
        const int64_t slice = (int64_t)q * sb.cnt_stride + pg;   // (cnt_stride = slices per query: the groups of this launch, or more)
        if (threadIdx.x == 0) sb.gcnt[slice] = n;
        for (int i = threadIdx.x; i < min(n, SCAN_STAGE); i += 256) sb.surv[slice * sb.slice_cap + i] = s_stage[i];
    };
    // (CF: the LAST group takes every probe behind the ones before it -- its table is the query's, not a list's, so
    //  one workgroup per query serves all consumer probes: one table write, one slice)
    // (long lists: several consumer groups of sb.cf_span probes each, so that no group's candidates outgrow its stage)
    const int cfs = CF ? sb.cf_span : 0;
    const int p_begin = (CF && cfs > 0 && pg > 0) ? G + (pg - 1) * cfs : pg * G;
    const int p_end = CF ? (pg == 0 ? min(P, G) : (cfs > 0 ? min(P, p_begin + cfs) : P)) : min(P, p_begin + G);
    const int tid = threadIdx.x;
    const int msz = M * 256;
    // LDS byte address of this wave's 256-byte segment of a LUT row (lut_store)
    const uint32_t lut_m0 = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)s_lut + 256u * (uint32_t)(tid >> 6));
    // nothing to scan in this group (a shard owns ~1/W of the probed lists): leave before the
    // 16 KB query table is fetched (sparse = sharded search only: the check costs two dependent
    // scalar loads per probe).  Producers always go on: they must publish.
    if (sparse && (!FILT || pg > 0)) {
        bool any = false;
        for (int p = p_begin; p < p_end; p++) {
            const int l = probe_list[q * P + p];
            if (l >= 0 && l < nlist && (!list_mask || list_mask[l]) && list_len[l] > 0) any = true;
        }
        if (!any) {   // uniform
            if (FILT && threadIdx.x == 0) sb.gcnt[(int64_t)q * sb.cnt_stride + pg] = 0;
            return;
        }
    }
    const float* st2q = st2 + (int64_t)q * msz;
    float s2r[MT > 0 ? MT : 1];
    // (uniform) this workgroup is a producer that takes the byte image: its group fits the estimates' place in LDS and holds
    // at least recall_num codes
    bool pc8 = false;
    int g_est = 0;   // lists of the estimate group: the longest prefix of the first probe group whose codes fit the estimates' place
    if constexpr (PC8) {
        if (pg == 0 && sb.prod_c8) {
            const int* po = pair_off + (int64_t)q * (P + 1);
            pc8 = true;
            for (int g = 1; g <= min(min(G, P), PC8_MAXG); g++)
                if (po[g] <= PC8_MAXN) g_est = g;
        }
    }
    if (IPF && MT > 0) {
        // same arithmetic as k_pq_ip_table: one fvec_inner_products_ny row per (m, code word)
        const int dsub = d / M;
        const float* xq = x + (int64_t)q * d;
#pragma unroll
        for (int i = 0; i < MT; i++) s2r[i] = fvec_ny_row<false>(xq + i * dsub, st2 + ((int64_t)i * 256 + tid) * dsub, dsub);
    } else if (C8 && (pg > 0 || pc8)) {
        // byte table: wave w reads table rows w, w + 4, .. (four code words per lane), see below; s2r is loaded later
    } else if (RES) {
        // no query table: the per-list table comes from the residual and the codebook
    } else if (MT > 0 && (!UNITS || (L2 ? q * P + pg != lut_pair : q != lut_q))) {
#pragma unroll
        for (int i = 0; i < MT; i++) s2r[i] = st2q[tid + 256 * i];
    }
    float4 v8[C8 ? MT / 4 : 1];
    if constexpr (C8) {
        if (pg > 0 || pc8) {
#pragma unroll
            for (int k = 0; k < MT / 4; k++)
                v8[k] = *reinterpret_cast<const float4*>(st2q + ((tid >> 6) + 4 * k) * 256 + 4 * lane);
        }
    }
    if (!L2 && (!UNITS || q != lut_q)) {   // inner product: the LUT is the query table itself, list independent
        if (UNITS) lut_q = q;
        if (MT > 0) {
            lut_store_begin(lut_m0);
            lut_store_rows<MT>([&](int i) { return s2r[i]; }, std::make_integer_sequence<int, (MT > 0 ? MT : 1)>{});
            lut_store_done();
        } else {
            for (int e = tid; e < msz; e += 256) s_lut[e] = st2q[e];
        }
    }
    if constexpr (CF && !C8) {
        if (pg > 0 || PCF) {   // (uniform) largest |entry| of the query's table: one word per wave, read behind the barrier below
            float mxv = 0.f;
#pragma unroll
            for (int i = 0; i < MT; i++) mxv = fmaxf(mxv, fabsf(s2r[i]));
            const uint32_t wmx = __reduce_max_sync(~0ull, __float_as_uint(mxv));   // non-negative floats order as integers
            if (lane == 0) s_red[tid >> 6] = wmx;
        }
    }
    if (FILT) {   // placed after the table loads were issued: their latency and this one overlap
        if (threadIdx.x == 0) {
            s_nstage = 0;
            if (CF) s_ncand = 0;
            if (pg > 0) {   // wait for this query's bound (published by its group-0 workgroup)
                // ONE relaxed 64-bit word carries (state << 32 | bound): no acquire/release fence is
                // needed (nothing else the producer wrote is read here), and agent-scope fences
                // would write back / invalidate the L2 this kernel lives on
                int spins = 0;
                unsigned long long word;
                // (bounded: dispatch order is not a contract -- if the producer has not published within ~2e6
                //  cycles the group goes on without a bound and the query takes the unfiltered selection)
                const int spin_max = sb.spins > 0 ? sb.spins : (1 << 11);
                while ((word = __hip_atomic_load(&sb.ready[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull &&
                       ++spins < spin_max)
                    __builtin_amdgcn_s_sleep(16);
                if (word != 0ull) {
                    s_tau = (word >> 32) == 1ull ? (uint32_t)word : 0xffffffffu;
                } else {   // give the query to the unfiltered selection instead of hanging
                    s_tau = 0xffffffffu;
                    s_nstage = sb.slice_cap + 1;
                    if (sb.timeouts) atomicAdd(sb.timeouts, 1ull);
                }
            }
        }
        __syncthreads();
        if (pg > 0) {
            tauq = s_tau;
            bound_on = tauq < KEY_SENTINEL;   // otherwise the query takes the unfiltered selection
            tau_f = key2f(L2 ? tauq : ~tauq);
        }
    }
    // the byte image of the query's table ("byte table" below), made by the whole workgroup from v8 where the fp32 table would be;
    // contains one barrier, none behind the image's stores
    auto c8_image = [&](float& qmax, float& c8_cq, float& c8_nd, auto two_c) {
        constexpr bool TWO = decltype(two_c)::value;   // + the image of the residuals (one workgroup per query, below)
        if constexpr (C8) {
        float lo[MT / 4], range = 0.f, Lsum = 0.f, amax = 0.f;
#pragma unroll
        for (int k = 0; k < MT / 4; k++) {
            float mn = fminf(fminf(v8[k].x, v8[k].y), fminf(v8[k].z, v8[k].w));
            float mx = fmaxf(fmaxf(v8[k].x, v8[k].y), fmaxf(v8[k].z, v8[k].w));
            // (wave reductions on the order-preserving keys: DPP, not 12 LDS permutes per row)
            mn = key2f(__reduce_min_sync(~0ull, f2key(mn)));
            mx = key2f(__reduce_max_sync(~0ull, f2key(mx)));
            lo[k] = mn;
            range = fmaxf(range, mx - mn);
            Lsum += mn;
            amax = fmaxf(amax, fmaxf(fabsf(mn), fabsf(mx)));
        }
        float* s_part = reinterpret_cast<float*>(s_red);   // [4 waves][3]
        if (lane == 0) {
            s_part[3 * (tid >> 6)] = range;
            s_part[3 * (tid >> 6) + 1] = Lsum;
            s_part[3 * (tid >> 6) + 2] = amax;
        }
        __syncthreads();
        range = fmaxf(fmaxf(s_part[0], s_part[3]), fmaxf(s_part[6], s_part[9]));
        Lsum = (s_part[1] + s_part[4]) + (s_part[7] + s_part[10]);
        amax = fmaxf(fmaxf(s_part[2], s_part[5]), fmaxf(s_part[8], s_part[11]));
        const float delta = (range / 255.f) * 1.000001f;   // (hi - lo) / delta stays below 255.5 whatever the roundings
        const float inv = delta > 0.f ? 1.f / delta : 0.f;
        uint32_t* s_b8 = reinterpret_cast<uint32_t*>(s_lut);
        const float delta2 = (delta / 255.f) * 1.000001f, inv2 = delta2 > 0.f ? 1.f / delta2 : 0.f;
#pragma unroll
        for (int k = 0; k < MT / 4; k++) {
            const float f[4] = {v8[k].x, v8[k].y, v8[k].z, v8[k].w};
            uint32_t w = 0, w2 = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int u = (int)rintf((f[e] - lo[k]) * inv);
                u = min(255, max(0, u));
                w |= (uint32_t)u << (8 * e);
                if constexpr (TWO) {
                    // ip = (lo - delta / 2) + delta u + delta2 u2 + e2,  |e2| <= 0.55 delta2 (the residual (ip - lo) - delta u lies in
                    // [-delta / 2, delta / 2] up to roundings of 3 * 2^-24 of the row's range = 0.004 delta2)
                    const float r = (f[e] - lo[k]) - delta * (float)u;
                    int u2 = (int)rintf(__builtin_fmaf(0.5f, delta, r) * inv2);
                    u2 = min(255, max(0, u2));
                    w2 |= (uint32_t)u2 << (8 * e);
                }
            }
            s_b8[((tid >> 6) + 4 * k) * 64 + lane] = w;
            if constexpr (TWO) s_b8[PC8_IMG2_OFF / 4 + ((tid >> 6) + 4 * k) * 64 + lane] = w2;
        }
        qmax = amax;
        c8_cq = 2.f * Lsum + 1.02f * (float)MT * delta;
        c8_nd = -2.f * delta;
        }
    };
    // ---- one workgroup per query (ScanBound::prod_c8, round 6): the bound from the BYTE IMAGE, then the filter pass over ALL probes ----
    // The regular producer builds a 16 KB table per list (the T2 row through the L2, 4096 fma + LDS stores, two barriers) to score
    // a few hundred codes: 43 % of the launch's workgroup time for 11 % of its codes, and a second workgroup per query that
    // waits for it, reads the query's table again and makes the byte image again.  All the bound needs is an UPPER bound of the
    // recall_num-th best exact value.  The image gives every code j of the first group the consumers' test value
    //       f_j = (dis0 - 2 sum_m lo - 1.02 M delta) + s_j - 2 delta U_j,        f_j - eps <= v_j <= f_j + 2.0202 M delta + eps
    // (the image's error bound |ip - lo - delta u8| <= 0.5001 delta per entry; eps = 50 * 2^-24 S, the roundings, as in the filter
    // pass), so the recall_num-th smallest f plus W = 2.03 M delta + 2^-16 S_max bounds the recall_num-th smallest exact value of the
    // group, hence the query's: estimates of the group in LDS (units of 64 codes dealt round-robin to the waves across all of its
    // lists), 256-bin histogram, tau1 = edge + W published -- and the SAME workgroup goes on with the filter pass over every probe
    // of the query, the first group included, against tau1: candidates f <= tau1 + margin get the reference's arithmetic, those with
    // v <= tau1 are the query's survivors (slice 0; slice 1 stays empty).  No slab segment is written: the callers score group 0
    // (repair launch) for every query whose slab is read (unfiltered selection, tie replay).
    // A group that does not fit (more than PC8_MAXN codes, fewer than recall_num valid ones, more than PC8_MAXG lists) takes the
    // regular producer, and the workgroup then runs the consumers' pass for the other probes as a second stage (slice 1).
    bool fused = false;
    float f_qmax = 0.f, f_cq = 0.f, f_nd = 0.f;
    if constexpr (PC8) {
        if (pc8) {   // (uniform)
            float* s_f = s_lut + MT * 64;   // [n0 <= PC8_MAXN]: the estimates (the filter pass's candidates take the place afterwards)
            uint32_t& s_smax = *(reinterpret_cast<uint32_t*>(s_cand) + 1);
            // the estimate group's lists: (first code | codes | dis0 | position of the first code in the query's row), read once
            int64_t* s_moff = reinterpret_cast<int64_t*>(s_cand + 2);             // [PC8_MAXG]
            int* s_mlen = reinterpret_cast<int*>(s_moff + PC8_MAXG);               // [PC8_MAXG]
            float* s_mdis = reinterpret_cast<float*>(s_mlen + PC8_MAXG);           // [PC8_MAXG]
            int* s_mpos = reinterpret_cast<int*>(s_mdis + PC8_MAXG);               // [PC8_MAXG]
            const int* poff = pair_off + (int64_t)q * (P + 1);
            const int ng = g_est, n0 = poff[g_est];
            if (tid < 64) {   // wave 0 (ng <= PC8_MAXG lanes of it)
                float sl = 0.f;
                if (tid < ng) {
                    const int pair = q * P + tid;
                    const int l = probe_list[pair];
                    int len = 0;
                    int64_t off = 0;
                    float t2m = 0.f;
                    if (l >= 0 && l < nlist && (!list_mask || list_mask[l])) {
                        len = max(0, list_len[l]);
                        off = list_off[l];
                        t2m = sb.t2max[l];
                    }
                    const float dis0 = coarse_dis[pair];
                    s_moff[tid] = off;
                    s_mlen[tid] = len;
                    s_mdis[tid] = dis0;
                    s_mpos[tid] = poff[tid];
                    if (len > 0) sl = fabsf(dis0) + t2m;
                }
                const uint32_t smx = wave_max_u32(__float_as_uint(sl));   // non-negative floats order as integers
                if (tid == 0) s_smax = smx;
            }
            float qmax = 0.f, c8_cq = 0.f, c8_nd = 0.f;
            c8_image(qmax, c8_cq, c8_nd, std::true_type{});
            __syncthreads();   // the images and the lists' data are in place
            GH_ST(t_pq);
            GH_ST_CNT(0);
            GH_ST_ADD(1, t_start, t_pq);
            // ---- the estimates: units of 64 codes dealt round-robin to the four waves across ALL lists of the group, the next
            // unit's codes and sums requested before the current unit's gathers.  Both images:
            //     f2_j = dis0 + s_j - 2 (sum_m (lo_m - delta / 2) + delta U_j + delta2 U2_j) - 1.1 M delta2,
            //     f2_j - eps <= v_j <= f2_j + 2.2 M delta2 + eps      (|e2| <= 0.55 delta2 per entry, eps: the roundings, <= 2^-16 S)
            const float delta = -0.5f * c8_nd, delta2 = (delta / 255.f) * 1.000001f;
            // c8_cq = 2 sum lo + 1.02 M delta  ->  2 sum (lo - delta / 2) + 1.1 M delta2
            const float cq2 = (c8_cq - 2.02f * (float)MT * delta) + 1.1f * (float)MT * delta2;
            const float nd2 = -2.f * delta2;
            {
                const int wv = tid >> 6;
                int r = 0, ub = 0;   // cursor of the unit whose codes are being requested: list r, whose first unit is ub
                int len = ng > 0 ? s_mlen[0] : 0, ns = (len + 63) >> 6;
                uint4 cn[MT / 16];
                float sn = 0.f;
                // state of the requested unit
                int n_len = 0, n_j0 = 0, n_pos = 0;
                float n_dis = 0.f;
                int64_t n_off = 0;
                bool n_have = false;
                auto request = [&](int u) {   // (uniform)
                    while (r < ng && u >= ub + ns) {
                        ub += ns;
                        r++;
                        len = r < ng ? s_mlen[r] : 0;
                        ns = (len + 63) >> 6;
                    }
                    n_have = r < ng;
                    if (n_have) {
                        n_len = len;
                        n_j0 = (u - ub) << 6;
                        n_pos = s_mpos[r];
                        n_dis = s_mdis[r];
                        n_off = s_moff[r];
                        const int jc = min(n_j0 + lane, n_len - 1);
                        const uint4* cp = reinterpret_cast<const uint4*>(codes + (n_off + jc) * MT);
#pragma unroll
                        for (int u2 = 0; u2 < MT / 16; u2++) cn[u2] = cp[u2];
                        sn = sb.sums[n_off + jc];
                    }
                };
                int u = wv;
                request(u);
                while (n_have) {
                    const int c_len = n_len, c_j0 = n_j0, c_pos = n_pos;
                    const float c_dis = n_dis;
                    uint32_t cw[MT / 4];
#pragma unroll
                    for (int u2 = 0; u2 < MT / 16; u2++) {
                        cw[4 * u2] = cn[u2].x; cw[4 * u2 + 1] = cn[u2].y; cw[4 * u2 + 2] = cn[u2].z; cw[4 * u2 + 3] = cn[u2].w;
                    }
                    const float sj = sn;
                    u += 4;
                    request(u);
                    const int j = c_j0 + lane;
                    uint32_t t[MT], t2[MT];
#pragma unroll
                    for (int m = 0; m < MT; m++) t[m] = lut_gather_u8(cw[m >> 2], m & 3, m);
#pragma unroll
                    for (int m = 0; m < MT; m++) t2[m] = lut_gather_u8_at<PC8_IMG2_OFF>(cw[m >> 2], m & 3, m);
                    __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the adds
                    uint32_t u4[4] = {t[0], t[1], t[2], t[3]}, v4[4] = {t2[0], t2[1], t2[2], t2[3]};
#pragma unroll
                    for (int m = 4; m < MT; m++) {
                        u4[m & 3] += t[m];
                        v4[m & 3] += t2[m];
                    }
                    const uint32_t U = (u4[0] + u4[1]) + (u4[2] + u4[3]), U2 = (v4[0] + v4[1]) + (v4[2] + v4[3]);
                    const float f = __builtin_fmaf(nd2, (float)U2, __builtin_fmaf(c8_nd, (float)U, (c_dis - cq2) + sj));
                    if (j < c_len) {   // (no validity predicates on this path: every code counts)
                        s_f[c_pos + j] = f;
                        g_fmn = fminf(g_fmn, f);
                        g_fmx = fmaxf(g_fmx, f);
                    }
                }
            }
            GH_ST(t_pl);
            GH_ST_ADD(3, t_pq, t_pl);
            // range of the estimates, then the upper edge of the 256-bin histogram's bin that holds the K-th smallest (the regular
            // producer's procedure, on LDS)
            int* hist = reinterpret_cast<int*>(s_stage);   // staging has not been used yet
            uint32_t rmn = 0, rmx = 0;
            {
                const float fmn = g_fmn == 0.f ? -0.f : g_fmn, fmx = g_fmx == 0.f ? 0.f : g_fmx;
                uint32_t mn = g_fmn <= g_fmx ? dis_key<true>(fmn) : 0xffffffffu;
                uint32_t mx = g_fmn <= g_fmx ? dis_key<true>(fmx) : 0u;
                mn = wave_min_u32(mn);
                mx = wave_max_u32(mx);
                if (lane == 0) {   // (s_red: last read in front of the image's barrier)
                    s_red[tid >> 6] = mn;
                    s_red[4 + (tid >> 6)] = mx;
                }
                hist[tid] = 0;
                __syncthreads();   // ... and every wave's estimates are in LDS
                rmn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
                rmx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
            }
            GH_ST(t_p1);
            GH_ST_ADD(16, t_pl, t_p1);
            uint32_t tk = KEY_SENTINEL - 1u;
            if (n0 >= sb.K) {   // (uniform)
                const uint32_t range = rmx - rmn;
                const int sh = range >= 256u ? (32 - __clz((int)range)) - 8 : 0;   // (range >> sh) < 256
                for (int i = tid; i < n0; i += 256) atomicAdd(&hist[(dis_key<true>(s_f[i]) - rmn) >> sh], 1);
                __syncthreads();
                if (tid < 64) {   // wave 0: scan of the 256 bins, 4 per lane
                    int c[4], c4 = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        c[u] = hist[lane * 4 + u];
                        c4 += c[u];
                    }
                    const int incl = wave_incl_scan(c4);
                    int run = incl - c4;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (run < sb.K && sb.K <= run + c[u]) {
                            unsigned long long edge = (unsigned long long)rmn + (((unsigned long long)(lane * 4 + u) + 1ull) << sh) - 1ull;
                            if (edge > (unsigned long long)rmx) edge = rmx;
                            s_tau = (uint32_t)edge;
                        }
                        run += c[u];
                    }
                }
                __syncthreads();
                const float sm = (__uint_as_float(s_smax) + 32.f * qmax) * (1.f / 65536.f);
                float tau1 = key2f(s_tau) + (2.21f * (float)MT * delta2 + sm);
                tau1 += fabsf(tau1) * 2.4e-7f;   // the sums' own roundings
                tk = min(dis_key<true>(tau1), KEY_SENTINEL - 1u);
                fused = true;
            }
            GH_ST(t_pp);
            GH_ST_ADD(4, t_pl, t_pp);
            if (tid == 0) {
                __hip_atomic_store(&sb.ready[q], (1ull << 32) | tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sb.gcnt[(int64_t)q * sb.cnt_stride + 1] = 0;   // (slice 1: unused)
                // a query whose nearest list alone is longer than the estimates' place, or whose first group holds fewer than
                // recall_num codes: to the unfiltered selection, via an "overflowed" slice (the repair launch scores all its groups)
                if (!fused) s_nstage = sb.slice_cap + 1;
            }
            if (!fused) {
                flush();
                return;
            }
            tauq = tk;
            tau_f = key2f(tk);
            bound_on = true;
            f_qmax = qmax;
            f_cq = c8_cq;
            f_nd = c8_nd;
        }
    }
    GH_ST(t_bound);
    if constexpr (CF) {
        if ((pg > 0 && bound_on) || fused) {   // (uniform)
            GH_ST_CNT(8);
            GH_ST_ADD(9, t_start, t_bound);
            // ---- filter pass (L2 consumers with a bound) ----------------------------------------------------------
            // Half of the regular loop's instructions build the per-list table T2[l] - 2 ip[q] (4096 entries for
            // lists of a few hundred codes).  Here the LUT is the QUERY's table ip[q] alone, written once per
            // workgroup -- no per-list build, no barriers in the probe loop -- and a code is tested on
            //     f = (dis0 + s_j) - 2 sum_m ip[q][m][c_m],     s_j = sum_m T2[l][m][c_m]  (kept beside the code, 4 bytes),
            // which differs from the reference's value  dis0 + sum_m fma(-2, ip, T2)  (sequential) only by rounding:
            // every one of the < 50 roundings of either evaluation is at most 2^-24 times a partial sum, and every
            // partial sum is bounded by S = |dis0| + sum_m max_c |T2[l][m][c]| + 2 sum_m max_c |ip[q][m][c]|, so
            // |f - exact| <= 50 * 2^-24 * S.  A code passes when f <= tau + 2^-17 S (a margin 2.5 times that, the
            // second term of S taken as 32 times the largest |entry| of the query's table).  The few that pass
            // (about as many as end up in the slice) get the EXACT value afterwards -- table entries fetched from
            // the L2-resident T2 row, fma and adds in the reference's order -- and the slice receives what the
            // regular loop would have put there: same keys, same positions.
            float qmax = f_qmax, c8_cq = f_cq, c8_nd = f_nd;
            // (one workgroup per query: the image is in place, the pass covers every probe)
            const int pb = fused ? 0 : p_begin, pe = fused ? P : p_end;
            if (fused) {
            } else if constexpr (C8) {
                // ---- byte table (round 5) --------------------------------------------------------------------------
                // 64 lanes gathering random fp32 entries of one 256-entry row hit the 32 banks ~3.5 deep (a half wave's 32
                // requests over 32 banks, 8 entries per bank): 57 % of this kernel's LDS cycles were conflicts.  A row of
                // BYTES is 64 words, two per bank: a gather is at most 2 deep.  The pass only has to prove a code OUTSIDE
                // the bound, so it may run on the u8 image of the query's table that the list-major pass uses
                // (q8scan.hip: ip ~ lo_m + delta * u8, |error| <= 0.5001 delta, one delta per query), same proof:
                //     v >= (dis0 - 2 sum_m lo - 1.02 M delta) + s_j - 2 delta U - 50 * 2^-24 S,   U = sum_m u8[m][c_m] (exact),
                // a code is a candidate iff (A + s_j) - 2 delta U <= tau + 2^-16 S.  The image is made HERE (the arithmetic
                // of k_q8_quant: wave w owns rows w, w + 4, ..; minimum and maximum of a row are one wave reduction) and
                // lives where the fp32 table will be written for the exact recompute of the candidates.
                c8_image(qmax, c8_cq, c8_nd, std::false_type{});
            } else {
                lut_store_begin(lut_m0);
                lut_store_rows<MT>([&](int i) { return s2r[i]; }, std::make_integer_sequence<int, (MT > 0 ? MT : 1)>{});
                lut_store_done();
                qmax = __uint_as_float(max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
            }
            // The table does not depend on the list, so nothing in this loop needs the workgroup in step: every WAVE
            // takes whole lists of the group (next one from a counter in LDS), 64 codes per step, the next step's codes
            // and sums requested before the current step's gathers -- four independent latency chains per workgroup
            // instead of one, and no barrier until the candidates are complete.
            int& s_next = *reinterpret_cast<int*>(s_cand + (C8 ? 0 : SCAN_CF_CAP));   // (the byte pass has no stage of its own)
            if (tid == 0) s_next = 0;
            __syncthreads();   // the LUT and the list counter are in place
            GH_ST(t_quant);
            GH_ST_ADD(10, t_bound, t_quant);
            const int ng = pe - pb;
            // (two copies of the loop, with and without the validity predicates: with their loads -- through generic pointers of
            //  the filter table -- anywhere in the loop body, the compiler's wait-count pass puts `s_waitcnt vmcnt(0)` in front of the
            //  gathers at the join behind them, i.e. every step waited for the NEXT step's codes it had just requested)
            auto filter_lists = [&](auto nid_c) {
            constexpr bool NID = decltype(nid_c)::value;
            // The wave's lists are pipelined.  What the loop needs of a list -- its codes' place in the arena, its length, dis0, the
            // position of its first code in the query's row -- is indexed by (query, probe) (pair_off, pair_base, coarse_dis), no
            // look-up through the list id: every lane holds one list of the group (loaded before the loop), a list's data are four
            // v_readlane, and the first 64 codes of the wave's NEXT list are requested during the current list's last step.  (Before:
            // probe -> list id -> length / offset -> codes, three dependent loads a wave waited for at the head of each of its ~7
            // lists.)  The margin's S takes the largest per-list bound of the index instead of the list's own.
            constexpr int NSET = C8 ? 1 : 2;   // (the byte-image pass: at most 64 lists per group)
            int mv_pos[NSET], mv_len[NSET], mv_olo[NSET], mv_ohi[NSET];
            float mv_dis[NSET];
#pragma unroll
            for (int t = 0; t < NSET; t++) {
                const int rr = t * 64 + lane, p = pb + min(rr, ng - 1);
                const int* po = pair_off + (int64_t)q * (P + 1) + p;
                const int o0 = po[0], o1 = po[1];
                const int64_t bo = sb.pair_base[(int64_t)q * P + p];
                mv_pos[t] = o0;
                mv_len[t] = rr < ng ? o1 - o0 : 0;
                mv_olo[t] = (int)(uint32_t)bo;
                mv_ohi[t] = (int)(uint32_t)((uint64_t)bo >> 32);
                mv_dis[t] = coarse_dis[(int64_t)q * P + p];
            }
            auto rl = [&](const int (&v)[NSET], int r) -> int {   // (r uniform)
                if constexpr (NSET == 1) return __builtin_amdgcn_readlane(v[0], r);
                else return r < 64 ? __builtin_amdgcn_readlane(v[0], r) : __builtin_amdgcn_readlane(v[1], r - 64);
            };
            auto grab = [&]() -> int {   // the wave's next list with codes (ng: none)
                for (;;) {
                    int r = 0;
                    if (lane == 0) r = atomicAdd(&s_next, 1);
                    r = __builtin_amdgcn_readfirstlane(r);
                    if (r >= ng) return ng;
                    if (rl(mv_len, r) > 0) return r;
                }
            };
            uint4 cn[MT / 16];
            float sn = 0.f;
            auto request_first = [&](int r) {   // the first 64 codes and sums of list r
                const int len = rl(mv_len, r);
                const int64_t off = (int64_t)(((uint64_t)(uint32_t)rl(mv_ohi, r) << 32) | (uint32_t)rl(mv_olo, r));
                const int jc = min(lane, len - 1);
                const uint4* cp = reinterpret_cast<const uint4*>(codes + (off + jc) * MT);
#pragma unroll
                for (int u = 0; u < MT / 16; u++) cn[u] = cp[u];
                sn = sb.sums[off + jc];
            };
            int r_cur = grab();
            if (r_cur < ng) request_first(r_cur);
            const float S_q = sb.t2max_all + 32.f * qmax;
            while (r_cur < ng) {
                const int p = pb + r_cur, len = rl(mv_len, r_cur), pbase = rl(mv_pos, r_cur);
                const int64_t off = (int64_t)(((uint64_t)(uint32_t)rl(mv_ohi, r_cur) << 32) | (uint32_t)rl(mv_olo, r_cur));
                float dis0;
                if constexpr (NSET == 1) dis0 = __builtin_amdgcn_readlane(mv_dis[0], r_cur);
                else dis0 = r_cur < 64 ? __builtin_amdgcn_readlane(mv_dis[0], r_cur) : __builtin_amdgcn_readlane(mv_dis[1], r_cur - 64);
                const uint8_t* lc = codes + off * MT;
                const float* ls = sb.sums + off;
                const int64_t* lid = ids + off;
                const float S = fabsf(dis0) + S_q;
                float thr = __builtin_fmaf(S, C8 ? 1.f / 65536.f : 1.f / 131072.f, tau_f);
                thr += fabsf(thr) * (C8 ? 2.4e-7f : 1.2e-7f);   // the threshold's own rounding(s)
                const float c8_A = dis0 - c8_cq;
                const int r_nxt = grab();
                for (int j0 = 0; j0 < len; j0 += 64) {
                    const int j = j0 + lane;
                    uint32_t cw[MT / 4];
#pragma unroll
                    for (int u = 0; u < MT / 16; u++) {
                        cw[4 * u] = cn[u].x; cw[4 * u + 1] = cn[u].y; cw[4 * u + 2] = cn[u].z; cw[4 * u + 3] = cn[u].w;
                    }
                    const float sj = sn;
                    if (j0 + 64 < len) {   // (uniform) the next step's codes and sums, in flight during this step's gathers
                        const int jc = min(j + 64, len - 1);
                        const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)jc * MT);
#pragma unroll
                        for (int u = 0; u < MT / 16; u++) cn[u] = cp[u];
                        sn = ls[jc];
                    } else if (r_nxt < ng) {
                        request_first(r_nxt);   // ... or the first of the wave's next list
                    }
                    bool ok = j < len;
                    if constexpr (NID) {
                        const int64_t id = lid[min(j, len - 1)];
                        ok = ok && id >= 0;
                        if (ok) ok = is_valid_doc(filt, id);
                    }
                    float f;
                    if constexpr (C8) {
                        uint32_t t[MT];
#pragma unroll
                        for (int m = 0; m < MT; m++) t[m] = lut_gather_u8(cw[m >> 2], m & 3, m);
                        __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the adds
                        uint32_t u4[4] = {t[0], t[1], t[2], t[3]};
#pragma unroll
                        for (int m = 4; m < MT; m++) u4[m & 3] += t[m];
                        const uint32_t U = (u4[0] + u4[1]) + (u4[2] + u4[3]);
                        f = __builtin_fmaf(c8_nd, (float)U, c8_A + sj);
                    } else {
                        float t[MT];
#pragma unroll
                        for (int m = 0; m < MT; m++) t[m] = lut_gather(cw[m >> 2], m & 3, m);
                        __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the adds
                        float g4[4] = {t[0], t[1], t[2], t[3]};   // four independent chains: the order is free here
#pragma unroll
                        for (int m = 4; m < MT; m++) g4[m & 3] += t[m];
                        const float g = (g4[0] + g4[1]) + (g4[2] + g4[3]);
                        f = __builtin_fmaf(-2.f, g, dis0 + sj);
                    }
                    const bool cand = ok && f <= thr;
                    const unsigned long long bal = __ballot(cand);
                    if (bal) {   // uniform per wave
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_ncand, __popcll(bal));
                        base = __builtin_amdgcn_readfirstlane(base);   // (lane 0 holds it: no LDS permute)
                        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                        if constexpr (C8) {   // the candidates sit in the part of the fp32 table's place the bytes leave free
                            if (cand && slot < C8_CAND)
                                reinterpret_cast<uint2*>(s_lut + MT * 64)[slot] = make_uint2((uint32_t)(pbase + j), (uint32_t)p);
                        } else {
                            if (cand && slot < SCAN_CF_CAP) s_cand[slot] = make_uint2((uint32_t)(pbase + j), (uint32_t)p);
                        }
                    }
                }
                r_cur = r_nxt;
            }
            };
            if (need_ids) filter_lists(std::true_type{});
            else filter_lists(std::false_type{});
            __syncthreads();
            GH_ST(t_loop);
            GH_ST_ADD(11, t_quant, t_loop);
            const int nc = s_ncand;
            if (nc > (C8 ? C8_CAND : SCAN_CF_CAP)) {   // (uniform) more candidates than the stage holds: the query takes the unfiltered path
                if (tid == 0) s_nstage = sb.slice_cap + 1;
            } else {
                // the exact value of a candidate: the regular loop's table entries and its adds, in the reference's order (the
                // list's T2 row from the L2, the query's table entries from LDS)
                auto exact_one = [&](bool have, const uint2 cd, auto glob) {   // whole workgroup: append() ballots
                    constexpr bool GLOB = decltype(glob)::value;   // the query's table entries from memory (L2), not from LDS
                    bool keep = false;
                    float dis = 0.f;
                    int pos = 0;
                    if (have) {
                        pos = (int)cd.x;
                        const int p = (int)cd.y, pair = q * P + p;
                        const int l = probe_list[pair];
                        const int j = pos - pair_off[(int64_t)q * (P + 1) + p];
                        const uint8_t* cj = codes + (list_off[l] + j) * MT;
                        const float* t2 = T2 + (int64_t)l * msz;
                        uint32_t cw[MT / 4];
#pragma unroll
                        for (int u = 0; u < MT / 16; u++) {
                            const uint4 cv = reinterpret_cast<const uint4*>(cj)[u];
                            cw[4 * u] = cv.x; cw[4 * u + 1] = cv.y; cw[4 * u + 2] = cv.z; cw[4 * u + 3] = cv.w;
                        }
                        dis = coarse_dis[pair];
#pragma unroll
                        for (int m0 = 0; m0 < MT; m0 += 8) {   // eight table entries in flight at a time
                            float a[8];
#pragma unroll
                            for (int m = 0; m < 8; m++) a[m] = t2[(m0 + m) * 256 + ((cw[(m0 + m) >> 2] >> (8 * (m & 3))) & 255u)];
                            float b[8];
#pragma unroll
                            for (int m = 0; m < 8; m++) {
                                const int e = (m0 + m) * 256 + ((cw[(m0 + m) >> 2] >> (8 * (m & 3))) & 255u);
                                b[m] = GLOB ? st2q[e] : s_lut[e];
                            }
#pragma unroll
                            for (int m = 0; m < 8; m++) dis += __builtin_fmaf(-2.0f, b[m], a[m]);
                        }
                        keep = dis <= tau_f;
                    }
                    append(keep, dis, pos);
                };
                if constexpr (C8) {
                    // The ~260 candidates' table entries come from the query's fp32 table in memory (16 KB, L2-hot: this workgroup
                    // read it a moment ago) -- 16 more gathers per candidate on the otherwise idle vector-memory path instead of
                    // writing the table into LDS (64 stores, two barriers) and gathering there: the LDS array is what this kernel
                    // is short of (profiles/r05_scan_parts.txt).  The candidates stay where the filter loop put them.
                    for (int c0 = 0; c0 < nc; c0 += 256) {   // uniform trip count
                        const int c = c0 + tid;
                        exact_one(c < nc, c < nc ? reinterpret_cast<const uint2*>(s_lut + MT * 64)[c] : make_uint2(0u, 0u), std::true_type{});
                    }
                } else {
                    for (int c0 = 0; c0 < nc; c0 += 256) {   // uniform trip count
                        const int c = c0 + tid;
                        exact_one(c < nc, c < nc ? s_cand[c] : make_uint2(0u, 0u), std::false_type{});
                    }
                }
            }
            flush();
            GH_ST(t_cend);
            GH_ST_ADD(12, t_loop, t_cend);
            return;
        }
    }
    if constexpr (C8) {
        if (pg > 0 || pc8) {   // (uniform) a consumer without a bound / a producer that left the byte image: the regular loop, its table
                               // entries the regular way
#pragma unroll
            for (int i = 0; i < MT; i++) s2r[i] = st2q[tid + 256 * i];
        }
    }
    // ---- producer with the filter pass's arithmetic (sb.prod_cf, round 5) ---------------------------------------------------
    // The producer was the expensive quarter of the launch: a per-list table (16 KB of T2 through the L2, 4096 fma + LDS stores,
    // two barriers) for lists of a few hundred codes -- 1635 cycles per (query, list) pair against 675 for a consumer pair.
    // It needs exact values for nothing but its own ~K + one-bin candidates: the bound only has to be an UPPER bound of the
    // K-th best.  With f_j = (dis0 + s_j) - 2 sum_m ip[c_m] (the consumers' test value, |f_j - v_j| <= 50 * 2^-24 S) the K-th
    // smallest f plus the margin 2^-17 S_max bounds the K-th smallest exact value, so: score the group with the query's table
    // alone, histogram the f's, publish tau' = edge + margin, recompute exactly the codes with f <= tau' + margin and keep
    // those with v <= tau'.  The slab segment of group 0 holds the f's, NOT the reference's values (see ScanBound::prod_cf).
    float prod_smax = 0.f;
    bool prod_done = false;
    if constexpr (CF && PCF) {
        if (pg == 0) {   // (uniform)
            prod_done = true;
            lut_store_begin(lut_m0);
            lut_store_rows<MT>([&](int i) { return s2r[i]; }, std::make_integer_sequence<int, (MT > 0 ? MT : 1)>{});
            lut_store_done();
            int& s_next = *reinterpret_cast<int*>(s_cand + SCAN_CF_CAP);
            uint32_t& s_smax = *(reinterpret_cast<uint32_t*>(s_cand + SCAN_CF_CAP) + 1);
            if (tid == 0) {
                s_next = 0;
                s_smax = 0u;
            }
            __syncthreads();   // the LUT, the list counter and the per-wave maxima are in place
            const float qmax = __uint_as_float(max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
            const int ng = p_end - p_begin;
            for (;;) {
                int r = 0;
                if (lane == 0) r = atomicAdd(&s_next, 1);
                r = __builtin_amdgcn_readfirstlane(r);
                if (r >= ng) break;
                const int p = p_begin + r, pair = q * P + p;
                const int l = probe_list[pair];
                if (l < 0 || l >= nlist) continue;            // uniform per wave
                if (list_mask && !list_mask[l]) continue;
                const int len = list_len[l];
                if (len <= 0) continue;
                const int64_t off = list_off[l];
                const uint8_t* lc = codes + off * MT;
                const float* ls = sb.sums + off;
                const int64_t* lid = ids + off;
                const float dis0 = coarse_dis[pair];
                float* o = out + (int64_t)q * q_stride + pair_off[(int64_t)q * (P + 1) + p];
                prod_smax = fmaxf(prod_smax, fabsf(dis0) + sb.t2max[l] + 32.f * qmax);
                uint4 cn[MT / 16];
                float sn;
                {
                    const int jc = min(lane, len - 1);
                    const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)jc * MT);
#pragma unroll
                    for (int u = 0; u < MT / 16; u++) cn[u] = cp[u];
                    sn = ls[jc];
                }
                for (int j0 = 0; j0 < len; j0 += 64) {
                    const int j = j0 + lane;
                    uint32_t cw[MT / 4];
#pragma unroll
                    for (int u = 0; u < MT / 16; u++) {
                        cw[4 * u] = cn[u].x; cw[4 * u + 1] = cn[u].y; cw[4 * u + 2] = cn[u].z; cw[4 * u + 3] = cn[u].w;
                    }
                    const float sj = sn;
                    if (j0 + 64 < len) {   // (uniform)
                        const int jc = min(j + 64, len - 1);
                        const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)jc * MT);
#pragma unroll
                        for (int u = 0; u < MT / 16; u++) cn[u] = cp[u];
                        sn = ls[jc];
                    }
                    bool ok = j < len;
                    if (need_ids && ok) {
                        const int64_t id = lid[j];
                        ok = id >= 0;
                        if (ok) ok = is_valid_doc(filt, id);
                    }
                    float t[MT];
#pragma unroll
                    for (int m = 0; m < MT; m++) t[m] = lut_gather(cw[m >> 2], m & 3, m);
                    __builtin_amdgcn_sched_barrier(0);
                    float g4[4] = {t[0], t[1], t[2], t[3]};
#pragma unroll
                    for (int m = 4; m < MT; m++) g4[m & 3] += t[m];
                    const float g = (g4[0] + g4[1]) + (g4[2] + g4[3]);
                    const float f = __builtin_fmaf(-2.f, g, dis0 + sj);
                    if (j < len) {
                        const float val = ok ? f : sentinel;
                        o[j] = val;
                        g_fmn = fminf(g_fmn, ok ? val : INFINITY);
                        g_fmx = fmaxf(g_fmx, ok ? val : -INFINITY);
                        g_nv += ok ? 1 : 0;
                    }
                }
            }
            if (lane == 0) atomicMax(&s_smax, __float_as_uint(prod_smax));   // non-negative floats order as integers
        }
    }
    if (!L2) __syncthreads();   // the LUT (written once per query) is complete; L2 rebuilds it per list
    GH_ST(t_lists);
    if (FILT && pg == 0) { GH_ST_CNT(0); GH_ST_ADD(1, t_start, t_lists); }
    for (int p = p_begin; p < (prod_done ? p_begin : p_end); p++) {
        GH_ST(t_l0);
        const int pair = q * P + p;
        const int l = probe_list[pair];
        if (l < 0 || l >= nlist) continue;            // uniform
        if (list_mask && !list_mask[l]) continue;
        int len = list_len[l];
        int64_t off = list_off[l];
        if (UNITS) {
            len = min(len - cbase, chunk_len);
            off += cbase;
        }
        if (len <= 0) continue;
        const uint8_t* lc = codes + off * M;
        // the first 256 codes are requested BEFORE the T2 row: both latencies overlap, and lists of
        // up to 256 codes (most of them) never wait for their codes after the LUT is ready
        // compiled code widths: any multiple of 8 bytes up to 64; a code is NLD loads of LW dwords
        constexpr bool PRE = MT > 0 && MT % 8 == 0 && MT <= 64;
        constexpr int LW = (MT % 16 == 0) ? 4 : 2, NLD = PRE ? MT / (4 * LW) : 1;
        typedef uint32_t cvec_t __attribute__((ext_vector_type(LW)));
        // (issued as inline asm: hipcc sinks an ordinary load down to its first use, behind both
        // barriers; the matching s_waitcnt is placed by hand where the codes are consumed)
        cvec_t cfirst[NLD];
        if (PRE) {
            const uint8_t* cp0 = lc + (int64_t)min(tid, len - 1) * (PRE ? MT : 16);
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                const uint8_t* a = cp0 + 4 * LW * u;
                if constexpr (LW == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cfirst[u]) : "v"(a) : "memory");
                else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(cfirst[u]) : "v"(a) : "memory");
            }
        }
        if (L2 && (!UNITS || pair != lut_pair)) {   // (uniform)
            if (UNITS) lut_pair = pair;
            __syncthreads();   // the previous list's gathers are finished
            const float* t2 = T2 + (int64_t)l * msz;
            if constexpr (RES) {
                // compute_residual (faiss:Index.cpp:95-102: x - reconstruct(key)) + ProductQuantizer::compute_distance_table
                // (faiss:impl/ProductQuantizer.cpp:505-516: one fvec_L2sqr_ny row per sub-quantizer)
                const int dsub = d / M;
                const float* xq = x + (int64_t)q * d;
                const float* cl = cc + (int64_t)l * d;
                for (int t = tid; t < d; t += 256) s_res[t] = xq[t] - cl[t];
                __syncthreads();
                if (MT > 0) {
                    lut_store_begin(lut_m0);
                    lut_store_rows<MT>([&](int i) { return fvec_ny_row<true>(s_res + i * dsub, st2 + ((int64_t)i * 256 + tid) * dsub, dsub); },
                                       std::make_integer_sequence<int, (MT > 0 ? MT : 1)>{});
                    lut_store_done();
                } else {
                    for (int e = tid; e < msz; e += 256) s_lut[e] = fvec_ny_row<true>(s_res + (e >> 8) * dsub, st2 + (int64_t)e * dsub, dsub);
                }
            } else if (MT > 0) {
                float tv[MT > 0 ? MT : 1];
#pragma unroll
                for (int i = 0; i < MT; i++) tv[i] = t2[tid + 256 * i];   // MT loads in flight
                lut_store_begin(lut_m0);
                lut_store_rows<MT>([&](int i) { return __builtin_fmaf(-2.0f, s2r[i], tv[i]); },
                                   std::make_integer_sequence<int, (MT > 0 ? MT : 1)>{});
                lut_store_done();
            } else {
                for (int e = tid; e < msz; e += 256) s_lut[e] = __builtin_fmaf(-2.0f, st2q[e], t2[e]);
            }
            __syncthreads();
        }
        GH_ST(t_l1);
        if (FILT && pg == 0) GH_ST_ADD(2, t_l0, t_l1);
        // L2: the coarse distance; IP: <x_q, centroid_l>, computed per pair by k_pair_ip (a chain of d/8
        // dependent fmas per AVX lane has no place inside this loop)
        const float dis0 = RES ? 0.f : coarse_dis[pair];   // (table mode 0: `float dis0 = 0`, gamma_index_ivfpq.h:236)
        const int64_t* lid = ids + off;
        const int pbase = pair_off[(int64_t)q * (P + 1) + p] + (UNITS ? cbase : 0);
        float* o = out + (int64_t)q * q_stride + pbase;
        // store + what the pre-filter tracks about a scored code
        // Distances are stored where something reads them: the first group's (its producer's histogram, the
        // unfiltered selection, the tie replay) and those of a group without a bound.  A consumer with a bound
        // keeps only its survivors; if k_select_final cannot finish the query from the slices (a slice
        // overflowed, > 256 equal keys at the cut) the repair launch scores the group again with stores.
        const bool store = !FILT || pg == 0 || !bound_on || sb.store_all;
        auto finish = [&](int j, bool ok, float dis) -> float {
            const float val = ok ? dis : sentinel;
            if (store) o[j] = val;
            if (FILT && pg == 0) {
                const bool valid = val != sentinel;
                g_fmn = fminf(g_fmn, valid ? val : INFINITY);
                g_fmx = fmaxf(g_fmx, valid ? val : -INFINITY);
                g_nv += valid ? 1 : 0;
            }
            return val;
        };
        // one code: validity, ADC (gathers issued together, adds in reference order), store
        auto do_code = [&](int j, const uint32_t* cw) -> float {
            // ids are read only when something can reject an entry (delete bit, range filter,
            // superseded slot); otherwise 8 of the 28 bytes per candidate stay in HBM
            bool ok = true;
            if (need_ids) {
                const int64_t id = lid[j];
                ok = id >= 0;  // bit 63 = kDelIdxMask (realtime_mem_data.h:26)
                if (ok) ok = is_valid_doc(filt, id);
            }
            float dis = dis0;
            if (PRE) {
                float t[PRE ? MT : 1];
#pragma unroll
                for (int m = 0; m < (PRE ? MT : 1); m++) t[m] = lut_gather(cw[m >> 2], m & 3, m);
                __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the add chain
#pragma unroll
                for (int m = 0; m < (PRE ? MT : 1); m++) dis += t[m];   // sequential, reference order
            } else {
                const uint8_t* cj = lc + (int64_t)j * M;
                for (int m = 0; m < M; m++) dis += s_lut[m * 256 + cj[m]];
            }
            return finish(j, ok, dis);
        };
        // uniform trip counts: append() ballots.  First 256 codes: already in registers.
        if (PRE) {
#pragma unroll
            for (int u = 0; u < NLD; u++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cfirst[u]) : : "memory");
        }
        {
            float val = sentinel;
            if (tid < len) {
                uint32_t cw[PRE ? MT / 4 : 1];
                if (PRE) {
#pragma unroll
                    for (int u = 0; u < NLD; u++)
#pragma unroll
                        for (int i = 0; i < LW; i++) cw[LW * u + i] = cfirst[u][i];
                }
                val = do_code(tid, cw);
            }
            if (FILT && bound_on) append(within(val), val, pbase + tid);
        }
        if constexpr (MT == 64) {
            // 64-byte codes: the 64 KB LUT leaves two workgroups per CU (2 waves per SIMD), so each
            // thread scores TWO codes per iteration -- 128 LDS gathers in flight, two independent
            // add chains -- instead of relying on other waves to cover its latency
            int j0 = 256;
            for (; j0 + 256 < len; j0 += 512) {   // both halves hold codes (uniform)
                const int ja = j0 + tid, jb = ja + 256;
                const bool ina = true, inb = jb < len;
                uint32_t cwa[16], cwb[16];
                {
                    const uint4* pa = reinterpret_cast<const uint4*>(lc + (int64_t)min(ja, len - 1) * 64);
                    const uint4* pb = reinterpret_cast<const uint4*>(lc + (int64_t)min(jb, len - 1) * 64);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint4 a = pa[u], b = pb[u];
                        cwa[4 * u] = a.x; cwa[4 * u + 1] = a.y; cwa[4 * u + 2] = a.z; cwa[4 * u + 3] = a.w;
                        cwb[4 * u] = b.x; cwb[4 * u + 1] = b.y; cwb[4 * u + 2] = b.z; cwb[4 * u + 3] = b.w;
                    }
                }
                bool oka = true, okb = true;
                if (need_ids) {
                    const int64_t ida = lid[min(ja, len - 1)], idb = lid[min(jb, len - 1)];
                    oka = ida >= 0;
                    if (oka) oka = is_valid_doc(filt, ida);
                    okb = idb >= 0;
                    if (okb) okb = is_valid_doc(filt, idb);
                }
                float ta[64], tb[64];
#pragma unroll
                for (int m = 0; m < 64; m++) ta[m] = lut_gather(cwa[m >> 2], m & 3, m);
#pragma unroll
                for (int m = 0; m < 64; m++) tb[m] = lut_gather(cwb[m >> 2], m & 3, m);
                __builtin_amdgcn_sched_barrier(0);   // all gathers in flight before the add chains
                float da = dis0, db = dis0;
#pragma unroll
                for (int m = 0; m < 64; m++) {       // each chain sequential, reference order
                    da += ta[m];
                    db += tb[m];
                }
                const float vala = ina ? finish(ja, oka, da) : sentinel;
                const float valb = inb ? finish(jb, okb, db) : sentinel;
                if (FILT && bound_on) {
                    append(within(vala), vala, pbase + ja);
                    append(within(valb), valb, pbase + jb);
                }
            }
            if (j0 < len) {   // at most 256 codes left: one per thread
                const int j = j0 + tid;
                float val = sentinel;
                if (j < len) {
                    uint32_t cw[16];
                    const uint4* cp = reinterpret_cast<const uint4*>(lc + (int64_t)j * 64);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint4 cv = cp[u];
                        cw[4 * u] = cv.x; cw[4 * u + 1] = cv.y; cw[4 * u + 2] = cv.z; cw[4 * u + 3] = cv.w;
                    }
                    val = do_code(j, cw);
                }
                if (FILT && bound_on) append(within(val), val, pbase + j);
            }
        } else {
            for (int j0 = 256; j0 < len; j0 += 256) {
                const int j = j0 + tid;
                float val = sentinel;
                if (j < len) {
                    uint32_t cw[PRE ? MT / 4 : 1];
                    if (PRE) {
                        const cvec_t* cp = reinterpret_cast<const cvec_t*>(lc + (int64_t)j * (PRE ? MT : 16));
#pragma unroll
                        for (int u = 0; u < NLD; u++) {
                            const cvec_t cv = cp[u];
#pragma unroll
                            for (int i = 0; i < LW; i++) cw[LW * u + i] = cv[i];
                        }
                    }
                    val = do_code(j, cw);
                }
                if (FILT && bound_on) append(within(val), val, pbase + j);
            }
        }
    }
    if (FILT && pg > 0) flush();   // also without a bound: the slice count must be written (0)
    GH_ST(t_lend);
    if (FILT && pg == 0) GH_ST_ADD(3, t_lists, t_lend);
    if (FILT && pg == 0) {
        // ---- producer: bound of this query's K-th best from its first probe group ----
        // 256-bin histogram of the group's valid keys over [min, max]; tau = upper edge of the bin
        // holding the K-th smallest.  At least K candidates are <= tau, hence the whole final top-K.
        __syncthreads();   // this workgroup's distance stores are visible to all its threads
        int* hist = reinterpret_cast<int*>(s_stage);   // staging has not been used yet
        const int n0 = pair_off[(int64_t)q * (P + 1) + min(G, P)];
        const float* o0 = out + (int64_t)q * q_stride;
        // float range -> key range (a zero may carry either sign: take the widest pair of keys)
        const float fmn = g_fmn == 0.f ? -0.f : g_fmn, fmx = g_fmx == 0.f ? 0.f : g_fmx;
        uint32_t mn = g_nv ? (L2 ? dis_key<L2>(fmn) : dis_key<L2>(fmx)) : 0xffffffffu;
        uint32_t mx = g_nv ? (L2 ? dis_key<L2>(fmx) : dis_key<L2>(fmn)) : 0u;
        int nv = g_nv;
        mn = wave_min_u32(mn);
        mx = wave_max_u32(mx);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nv += __shfl_xor(nv, off, 64);
        if (lane == 0) {
            s_red[threadIdx.x >> 6] = mn;
            s_red[4 + (threadIdx.x >> 6)] = mx;
            s_red[8 + (threadIdx.x >> 6)] = (uint32_t)nv;
        }
        hist[threadIdx.x] = 0;
        __syncthreads();
        mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
        nv = (int)(s_red[8] + s_red[9] + s_red[10] + s_red[11]);
        uint32_t tau = 0xffffffffu;
        if (nv >= sb.K) {   // uniform
            const uint32_t range = mx - mn;
            const int sh = range >= 256u ? (32 - __clz((int)range)) - 8 : 0;   // (range >> sh) < 256
            for (int i0 = 0; i0 < n0; i0 += 256 * 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = o0[min(i0 + u * 256 + (int)threadIdx.x, n0 - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t key = dis_key<L2>(t[u]);
                    if (i0 + u * 256 + (int)threadIdx.x < n0 && key < KEY_SENTINEL)
                        atomicAdd(&hist[(key - mn) >> sh], 1);
                }
            }
            __syncthreads();
            if (threadIdx.x < 64) {   // wave 0: scan of the 256 bins, 4 per lane
                int c[4], c4 = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    c[u] = hist[lane * 4 + u];
                    c4 += c[u];
                }
                const int incl = wave_incl_scan(c4);
                int run = incl - c4;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (run < sb.K && sb.K <= run + c[u]) {
                        unsigned long long edge = (unsigned long long)mn +
                                                  (((unsigned long long)(lane * 4 + u) + 1ull) << sh) - 1ull;
                        if (edge > (unsigned long long)mx) edge = mx;
                        s_tau = (uint32_t)edge;
                    }
                    run += c[u];
                }
            }
            __syncthreads();
            tau = s_tau;
        } else {
            // fewer than K candidates in the first group (a shard that holds one short list of the query, the tail of
            // a filter): everything valid is a survivor -- the widest bound there is.  The group's own < K candidates
            // fit its slice; consumers append all of theirs (a slice that overflows sends the query to the unfiltered
            // selection, as ever).  Without this such a query had no bound and went there always: a handful per chunk of a
            // list shard, each a latency-bound pass over its slab behind the chunk's scan.
            tau = KEY_SENTINEL - 1u;
        }
        float prod_m = 0.f;
        if constexpr (CF && PCF) {
            if (prod_done && tau < KEY_SENTINEL) {   // (uniform) the bound of the EXACT values: the f's edge + their error margin
                const uint32_t smax = *(reinterpret_cast<const uint32_t*>(s_cand + SCAN_CF_CAP) + 1);
                prod_m = __uint_as_float(smax) * (1.f / 131072.f);
                if (tau < KEY_SENTINEL - 1u) {
                    float tp = key2f(tau) + prod_m;
                    tp += fabsf(tp) * 1.2e-7f;   // the sum's own rounding
                    tau = min(dis_key<L2>(tp), KEY_SENTINEL - 1u);
                }
            }
        }
        if (threadIdx.x == 0)
            __hip_atomic_store(&sb.ready[q], tau < KEY_SENTINEL ? ((1ull << 32) | tau) : (2ull << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        GH_ST(t_pub);
        GH_ST_ADD(4, t_lend, t_pub);
        // the group's own candidates within the bound become its survivor slice (slice 0), like a
        // consumer's: k_select_final then reads a few hundred items per query and never the distance
        // buffer (one wave walking a long first group -- 24 k candidates at C4 -- was the slow part)
        bool prod_exact = false;
        if constexpr (CF && PCF) prod_exact = prod_done;
        if (tau < KEY_SENTINEL && prod_exact) {   // uniform
            if constexpr (CF && PCF) {
                // candidates: f <= tau' + margin (every code whose exact value is within tau' is among them), then the exact
                // value in the reference's order of operations -- the list's T2 row from the L2, the query's table in LDS
                __syncthreads();        // the histogram (aliasing the staging area) has been read
                const float taup = key2f(tau);
                float thr = taup + prod_m;
                thr += fabsf(thr) * 1.2e-7f;
                for (int i0 = 0; i0 < n0; i0 += 256) {
                    const int idx = i0 + (int)threadIdx.x;
                    const float fv = o0[min(idx, n0 - 1)];
                    const bool cand = idx < n0 && fv <= thr;   // (the sentinel is +inf)
                    const unsigned long long bal = __ballot(cand);
                    if (bal) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_ncand, __popcll(bal));
                        base = __builtin_amdgcn_readfirstlane(base);   // (lane 0 holds it: no LDS permute)
                        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                        if (cand && slot < SCAN_CF_CAP) s_cand[slot] = make_uint2((uint32_t)idx, 0u);
                    }
                }
                __syncthreads();
                const int nc = s_ncand;
                if (nc > SCAN_CF_CAP) {   // (uniform) more candidates than the stage holds: the query takes the unfiltered path
                    if (tid == 0) s_nstage = sb.slice_cap + 1;
                } else {
                    const int* poff = pair_off + (int64_t)q * (P + 1);
                    for (int c0 = 0; c0 < nc; c0 += 256) {   // uniform trip count: append() ballots
                        const int c = c0 + tid;
                        bool keep = false;
                        float dis = 0.f;
                        int pos = 0;
                        if (c < nc) {
                            pos = (int)s_cand[c].x;
                            int p = p_begin;
                            for (int pp = p_begin + 1; pp < p_end; pp++) p = poff[pp] <= pos ? pp : p;   // last probe with off <= pos
                            const int pair = q * P + p;
                            const int l = probe_list[pair];
                            const int j = pos - poff[p];
                            const uint8_t* cj = codes + (list_off[l] + j) * MT;
                            const float* t2 = T2 + (int64_t)l * msz;
                            uint32_t cw[MT / 4];
#pragma unroll
                            for (int u = 0; u < MT / 16; u++) {
                                const uint4 cv = reinterpret_cast<const uint4*>(cj)[u];
                                cw[4 * u] = cv.x; cw[4 * u + 1] = cv.y; cw[4 * u + 2] = cv.z; cw[4 * u + 3] = cv.w;
                            }
                            dis = coarse_dis[pair];
#pragma unroll
                            for (int m0 = 0; m0 < MT; m0 += 8) {
                                float a[8];
#pragma unroll
                                for (int m = 0; m < 8; m++) a[m] = t2[(m0 + m) * 256 + ((cw[(m0 + m) >> 2] >> (8 * (m & 3))) & 255u)];
#pragma unroll
                                for (int m = 0; m < 8; m++)
                                    dis += __builtin_fmaf(-2.0f, s_lut[(m0 + m) * 256 + ((cw[(m0 + m) >> 2] >> (8 * (m & 3))) & 255u)], a[m]);
                            }
                            keep = dis <= taup;
                        }
                        append(keep, dis, pos);
                    }
                }
            }
        } else if (tau < KEY_SENTINEL) {   // uniform
            __syncthreads();        // the histogram (aliasing the staging area) has been read
            for (int i0 = 0; i0 < n0; i0 += 256 * 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = o0[min(i0 + u * 256 + (int)threadIdx.x, n0 - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int idx = i0 + u * 256 + (int)threadIdx.x;
                    append(idx < n0 && dis_key<L2>(t[u]) <= tau, t[u], idx);
                }
            }
        }
        flush();   // without a bound: count 0
        GH_ST(t_pend);
        GH_ST_ADD(5, t_pub, t_pend);
    }
    };   // body
    if (!repair) {
        body(q, pg);
    } else if (UNITS) {
        // a workgroup takes a contiguous run of units: consecutive chunks of one list share its LUT
        const int nu = *rq_count, per = (nu + (int)gridDim.x - 1) / (int)gridDim.x;
        const int w0 = (int)blockIdx.x * per, w1 = min(nu, w0 + per);
        for (int w = w0; w < w1; w++) {
            const uint32_t u = (uint32_t)rq_list[w];
            cbase = (int)(u & 8191u) * chunk_len;
            body((int)(u >> 20), (int)((u >> 13) & 127u));
            __syncthreads();   // the LUT of this unit has been consumed
        }
    } else {
        const int nrq = *rq_count;
        for (int w = blockIdx.x; w / pg_cnt < nrq; w += gridDim.x) {
            body(rq_list[w / pg_cnt], pg_lo + w % pg_cnt);
            __syncthreads();   // the LUT of this item has been consumed
        }
    }
}

// the kernels: the body above under its register budgets
template <bool L2, int MT, bool FILT, bool IPF = false, bool UNITS = false, bool CF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void k_ivfpq_scan_pair(

        const float* __restrict__ x, int nq, int d, int M, int P, int G, const int* __restrict__ probe_list,
        const float* __restrict__ coarse_dis, const float* __restrict__ cc,
        const float* __restrict__ st2, const float* __restrict__ T2,
        const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
        const uint8_t* __restrict__ list_mask, int nlist, const uint8_t* __restrict__ codes,
        const int64_t* __restrict__ ids, const int* __restrict__ pair_off, int64_t q_stride,
        float* __restrict__ out, const FilterDesc* __restrict__ ftab, const int* __restrict__ qfil, int need_ids,
        float sentinel, const int* __restrict__ qperm, int pg_lo, int pg_cnt, int sparse, ScanBound sb,
        const int* __restrict__ rq_list, const int* __restrict__ rq_count, int chunk_len) {
    scan_pair_body<L2, MT, FILT, IPF, UNITS, CF, false>(x, nq, d, M, P, G, probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off, q_stride, out, ftab, qfil, need_ids, sentinel, qperm, pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len);
}
// L2 table mode 0: per-list tables from the residual (RES in the body)
template <int MT, bool FILT, bool UNITS>
__global__ __launch_bounds__(256) void k_ivfpq_scan_pair_res(

        const float* __restrict__ x, int nq, int d, int M, int P, int G, const int* __restrict__ probe_list,
        const float* __restrict__ coarse_dis, const float* __restrict__ cc,
        const float* __restrict__ pqc, const float* __restrict__ T2,
        const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
        const uint8_t* __restrict__ list_mask, int nlist, const uint8_t* __restrict__ codes,
        const int64_t* __restrict__ ids, const int* __restrict__ pair_off, int64_t q_stride,
        float* __restrict__ out, const FilterDesc* __restrict__ ftab, const int* __restrict__ qfil, int need_ids,
        float sentinel, const int* __restrict__ qperm, int pg_lo, int pg_cnt, int sparse, ScanBound sb,
        const int* __restrict__ rq_list, const int* __restrict__ rq_count, int chunk_len) {
    scan_pair_body<true, MT, FILT, false, UNITS, false, false, false, true>(x, nq, d, M, P, G, probe_list, coarse_dis, cc, pqc, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off, q_stride, out, ftab, qfil, need_ids, sentinel, qperm, pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len);
}

// filter pass + the producer on its arithmetic (ScanBound::prod_cf): held to six waves per SIMD like the plain filter-pass kernel
template <int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(6, 8))) void k_ivfpq_scan_pair_pcf(

        const float* __restrict__ x, int nq, int d, int M, int P, int G, const int* __restrict__ probe_list,
        const float* __restrict__ coarse_dis, const float* __restrict__ cc,
        const float* __restrict__ st2, const float* __restrict__ T2,
        const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
        const uint8_t* __restrict__ list_mask, int nlist, const uint8_t* __restrict__ codes,
        const int64_t* __restrict__ ids, const int* __restrict__ pair_off, int64_t q_stride,
        float* __restrict__ out, const FilterDesc* __restrict__ ftab, const int* __restrict__ qfil, int need_ids,
        float sentinel, const int* __restrict__ qperm, int pg_lo, int pg_cnt, int sparse, ScanBound sb,
        const int* __restrict__ rq_list, const int* __restrict__ rq_count, int chunk_len) {
    scan_pair_body<true, MT, true, false, false, true, true>(x, nq, d, M, P, G, probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off, q_stride, out, ftab, qfil, need_ids, sentinel, qperm, pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len);
}

// filter pass on the byte table (ScanBound::c8)
template <int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(112), amdgpu_waves_per_eu(7, 8))) void k_ivfpq_scan_pair_c8(

        const float* __restrict__ x, int nq, int d, int M, int P, int G, const int* __restrict__ probe_list,
        const float* __restrict__ coarse_dis, const float* __restrict__ cc,
        const float* __restrict__ st2, const float* __restrict__ T2,
        const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
        const uint8_t* __restrict__ list_mask, int nlist, const uint8_t* __restrict__ codes,
        const int64_t* __restrict__ ids, const int* __restrict__ pair_off, int64_t q_stride,
        float* __restrict__ out, const FilterDesc* __restrict__ ftab, const int* __restrict__ qfil, int need_ids,
        float sentinel, const int* __restrict__ qperm, int pg_lo, int pg_cnt, int sparse, ScanBound sb,
        const int* __restrict__ rq_list, const int* __restrict__ rq_count, int chunk_len) {
    scan_pair_body<true, MT, true, false, false, true, false, true>(x, nq, d, M, P, G, probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off, q_stride, out, ftab, qfil, need_ids, sentinel, qperm, pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len);
}

// ... with the producer on the byte image as well (ScanBound::prod_c8)
template <int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(112), amdgpu_waves_per_eu(7, 8))) void k_ivfpq_scan_pair_pc8(

        const float* __restrict__ x, int nq, int d, int M, int P, int G, const int* __restrict__ probe_list,
        const float* __restrict__ coarse_dis, const float* __restrict__ cc,
        const float* __restrict__ st2, const float* __restrict__ T2,
        const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
        const uint8_t* __restrict__ list_mask, int nlist, const uint8_t* __restrict__ codes,
        const int64_t* __restrict__ ids, const int* __restrict__ pair_off, int64_t q_stride,
        float* __restrict__ out, const FilterDesc* __restrict__ ftab, const int* __restrict__ qfil, int need_ids,
        float sentinel, const int* __restrict__ qperm, int pg_lo, int pg_cnt, int sparse, ScanBound sb,
        const int* __restrict__ rq_list, const int* __restrict__ rq_count, int chunk_len) {
    scan_pair_body<true, MT, true, false, false, true, false, true, false, true>(x, nq, d, M, P, G, probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off, q_stride, out, ftab, qfil, need_ids, sentinel, qperm, pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len);
}

// (M = 32: the 32 KB table leaves four workgroups per CU whatever the registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void k_ivfpq_scan_pair_c8m32(

        const float* __restrict__ x, int nq, int d, int M, int P, int G, const int* __restrict__ probe_list,
        const float* __restrict__ coarse_dis, const float* __restrict__ cc,
        const float* __restrict__ st2, const float* __restrict__ T2,
        const int64_t* __restrict__ list_off, const int* __restrict__ list_len,
        const uint8_t* __restrict__ list_mask, int nlist, const uint8_t* __restrict__ codes,
        const int64_t* __restrict__ ids, const int* __restrict__ pair_off, int64_t q_stride,
        float* __restrict__ out, const FilterDesc* __restrict__ ftab, const int* __restrict__ qfil, int need_ids,
        float sentinel, const int* __restrict__ qperm, int pg_lo, int pg_cnt, int sparse, ScanBound sb,
        const int* __restrict__ rq_list, const int* __restrict__ rq_count, int chunk_len) {
    scan_pair_body<true, 32, true, false, false, true, false, true>(x, nq, d, M, P, G, probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off, q_stride, out, ftab, qfil, need_ids, sentinel, qperm, pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len);
}

// queries WITHOUT a bound join the repair list (ScanBound::prod_cf launches: their first group's slab segment holds the
// producer's approximate values, and the unfiltered selection reads the slab)
__global__ __launch_bounds__(256) void k_rq_nobound(const unsigned long long* __restrict__ ready, int nq, int* __restrict__ rq_list,
                                                    int* __restrict__ rq_count) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < nq && (ready[q] >> 32) != 1ull) rq_list[atomicAdd(rq_count, 1)] = q;
}
void launch_rq_nobound(hipStream_t s, const unsigned long long* ready, int nq, int* rq_list, int* rq_count) {
    if (nq > 0) hipLaunchKernelGGL(k_rq_nobound, dim3((nq + 255) / 256), dim3(256), 0, s, ready, nq, rq_list, rq_count);
}

int scan_slice_cap(int K) { return K <= 256 ? SCAN_SLICE : 2 * SCAN_SLICE; }
bool scan_cf_applies(bool l2, int M, int P, int G, bool have_sums, bool store_all) {
    return l2 && have_sums && !store_all && (M == 16 || M == 32) && P > G;
}

int scan_group_size(int nq, int P, int G0) {
    // probes per workgroup: amortise the query table, but keep >= ~4096 workgroups in flight
    const char* ge = getenv("GAMMA_HIP_SCAN_G");   // (read per call: tools sweep it inside one process)
    const int g_env = ge ? atoi(ge) : 0;
    int G = g_env > 0 ? g_env : G0;
    // (a start that is not a power of two -- 5 with the byte-image pass -- steps to the power of two below it: 5 -> 4 -> 2, not
    //  5 -> 2, which skipped the G >= 4 the bounded scan needs for batches of 512 .. 585 queries at 32 probes)
    while (G > 1 && (int64_t)nq * ((P + G - 1) / G) < 4096) G = (G & (G - 1)) ? (1 << (31 - __builtin_clz((unsigned)G))) : G >> 1;
    return std::max(1, std::min(G, P));
}

void launch_ivfpq_scan_pair(hipStream_t s, bool l2, const float* x, int nq, int d, int M, int P,
                            const int* probe_list, const float* coarse_dis, const float* cc,
                            const float* st2, const float* T2, const int64_t* list_off,
                            const int* list_len, const uint8_t* list_mask, int nlist,
                            const uint8_t* codes, const int64_t* ids, const int* pair_off,
                            int64_t q_stride, float* out, const FilterDesc* ftab, const int* qfil, int need_ids,
                            const int* qperm, int G, int pg_lo, int pg_cnt, int sparse, const ScanBound* bound,
                            const float* pqc_fused, const int* rq_list, const int* rq_count, int chunk_len, int max_units) {
    if (nq <= 0 || pg_cnt <= 0) return;
#ifdef GH_SCAN_TIMING
    {
        static int calls = 0;
        if (bound && (++calls & 15) == 0) {
            unsigned long long t[32];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_scan_t), sizeof(t));
            const double np = (double)std::max<unsigned long long>(1, t[0]), nc = (double)std::max<unsigned long long>(1, t[8]);
            fprintf(stderr, "pc8: range %.0f, hist1 %.0f, exact loop %.0f, candidates %.1f of %.1f\n", t[16] / np, t[17] / np, t[18] / np, t[19] / np, t[20] / np);
            fprintf(stderr, "scan phases, shader cycles per workgroup -- producers (%llu): start %.0f, lists total %.0f (of which table builds incl. "
                    "their waits %.0f), bound %.0f, own slice %.0f; consumers (%llu): to the bound %.0f, byte image %.0f, filter loop %.0f, exact + flush %.0f\n",
                    t[0], t[1] / np, t[3] / np, t[2] / np, t[4] / np, t[5] / np, t[8], t[9] / nc, t[10] / nc, t[11] / nc, t[12] / nc);
            unsigned long long z[32] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_scan_t), z, sizeof(z));
        }
    }
#endif
    if (chunk_len > 0 && (bound || pqc_fused || !rq_list || G != 1 || pg_lo != 0 || pg_cnt != P || max_units < 1)) {
        launch_refused("launch_ivfpq_scan_pair: unit mode with a bound, a fused table, grouped probes or no work list");
        return;
    }
    // L2 table mode 0 (no precomputed table on the handle): `st2` is the PQ codebook, the per-list tables come from the residual
    const bool res = l2 && T2 == nullptr;
    if (res && (pqc_fused || (bound && bound->sums))) {
        launch_refused("launch_ivfpq_scan_pair: table mode 0 takes neither fused query tables nor the filter pass");
        return;
    }
    if (pqc_fused) {   // the table is computed inside the kernel (IPF): one workgroup per query, M 16 / 32
        if (!bound || pg_cnt != 1 || (M != 16 && M != 32)) {
            launch_refused("launch_ivfpq_scan_pair: fused query tables need a bound, one group per query and M 16 / 32");
            return;
        }
        st2 = pqc_fused;
    }
    // LUT | survivor staging | a few words (see the kernel)
    size_t lds = (size_t)M * 256 * sizeof(float) + SCAN_STAGE * sizeof(unsigned long long) + 16 * sizeof(int);
    if (res) lds += (size_t)d * sizeof(float);   // the residual
    dim3 grid((unsigned)(8 * (int64_t)((nq + 7) / 8) * pg_cnt));
    if (bound) {   // P(0) | P(t+1) C(t) ...: whole batches, see the kernel
        const int SB = bound->batch > 0 ? bound->batch : SCAN_BATCH;
        const int64_t nq8 = (nq + 7) / 8, nb = (nq8 + SB - 1) / SB;
        grid.x = (unsigned)(8 * (SB + nb * SB * pg_cnt));
    }
    ScanBound sb = {};
    sb.slice_cap = SCAN_SLICE;
    if (bound) sb = *bound;
    // filter pass for the consumers of a bounded L2 scan: needs the per-code sums (sb.sums) and survivor-only consumers
    const bool cf = bound && l2 && !pqc_fused && (pg_cnt > 1 || (sb.prod_c8 && pg_cnt == 1 && P > G)) && sb.sums && sb.t2max &&
                    !sb.store_all && (M == 16 || M == 32);
    if (rq_list) {   // repair launch: a fixed grid loops over the flagged (query, group) items
        if (bound || pqc_fused) {
            launch_refused("launch_ivfpq_scan_pair: a repair launch takes neither a bound nor fused tables");
            return;
        }
        grid.x = (unsigned)std::min<int64_t>((int64_t)nq * pg_cnt, 2048);
    }
    if (chunk_len > 0) {   // as many workgroups as are resident at once (LDS: the LUT), no more than there can be units
        const int per_cu = std::max(1, std::min(8, (int)(160 * 1024 / (lds + 1024))));
        grid.x = (unsigned)std::min<int64_t>(max_units, 256 * per_cu);
    }
    // the filter pass holds its group's lists one per lane (two registers for the fp32 pass, one for the byte image)
    if (cf && (!sb.pair_base || (sb.cf_span > 0 ? sb.cf_span : (sb.prod_c8 ? P : P - G)) > (sb.c8 ? 64 : 128))) {
        launch_refused("launch_ivfpq_scan_pair: the filter pass needs pair_base and at most 64 (byte image) / 128 lists per consumer group");
        return;
    }
    if (sb.prod_c8 && !(cf && sb.c8 && M == 16 && pg_cnt == 1 && sb.cnt_stride >= 2 && sb.cf_span == 0)) {
        launch_refused("launch_ivfpq_scan_pair: one workgroup per query (prod_c8) needs the M = 16 byte-image pass, one group per launch and two slices");
        return;
    }
    if (cf) lds += (sb.c8 ? 0 : SCAN_CF_CAP * sizeof(uint2)) + 16 + (sb.prod_c8 ? PC8_MAXG * 24 + 4096 : 0);
#define GH_SCAN(LL, MT, FF)                                                                       \
    GH_SCAN4(LL, MT, FF, false)
#define GH_SCAN4(LL, MT, FF, II) GH_SCAN5(LL, MT, FF, II, false)
#define GH_SCAN5(LL, MT, FF, II, UU)                                                                       \
    hipLaunchKernelGGL((k_ivfpq_scan_pair<LL, MT, FF, II, UU>), grid, dim3(256), lds, s, x, nq, d, M, P, G,     \
                       probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, \
                       ids, pair_off, q_stride, out, ftab, qfil, need_ids, LL ? INFINITY : -INFINITY, qperm,   \
                       pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len)
#define GH_SCAN_M(LL, FF)                       \
    do {                                        \
        if (M == 16) GH_SCAN(LL, 16, FF);       \
        else if (M == 32) GH_SCAN(LL, 32, FF);  \
        else if (M == 64) GH_SCAN(LL, 64, FF);  \
        else if (M == 8) GH_SCAN(LL, 8, FF);    \
        else if (M == 24) GH_SCAN(LL, 24, FF);  \
        else if (M == 48) GH_SCAN(LL, 48, FF);  \
        else GH_SCAN(LL, 0, FF);                \
    } while (0)
#define GH_SCAN_RES(FF, UU)                                                                                                  \
    do {                                                                                                                    \
        if (M == 16) GH_SCAN_RES1(16, FF, UU);                                                                              \
        else if (M == 32) GH_SCAN_RES1(32, FF, UU);                                                                         \
        else if (M == 64) GH_SCAN_RES1(64, FF, UU);                                                                         \
        else if (M == 8) GH_SCAN_RES1(8, FF, UU);                                                                           \
        else GH_SCAN_RES1(0, FF, UU);                                                                                       \
    } while (0)
#define GH_SCAN_RES1(MT, FF, UU)                                                                                            \
    hipLaunchKernelGGL((k_ivfpq_scan_pair_res<MT, FF, UU>), grid, dim3(256), lds, s, x, nq, d, M, P, G, probe_list, coarse_dis, \
                       cc, st2, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off, q_stride, out, ftab, qfil,      \
                       need_ids, INFINITY, qperm, pg_lo, pg_cnt, sparse, sb, rq_list, rq_count, chunk_len)
    if (res) {
        if (lds > (64u << 10)) {   // M = 64: 64 KB of table + the staging words
            static std::atomic<uint64_t> attr{0};
            if (first_call_on_device(attr)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ivfpq_scan_pair_res<64, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ivfpq_scan_pair_res<64, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ivfpq_scan_pair_res<64, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ivfpq_scan_pair_res<0, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ivfpq_scan_pair_res<0, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ivfpq_scan_pair_res<0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
            }
        }
        if (chunk_len > 0) GH_SCAN_RES(false, true);
        else if (bound) GH_SCAN_RES(true, false);
        else GH_SCAN_RES(false, false);
    } else if (chunk_len > 0) {
#define GH_SCAN_U(LL)                                           \
    do {                                                        \
        if (M == 16) GH_SCAN5(LL, 16, false, false, true);      \
        else if (M == 32) GH_SCAN5(LL, 32, false, false, true); \
        else if (M == 64) GH_SCAN5(LL, 64, false, false, true); \
        else if (M == 8) GH_SCAN5(LL, 8, false, false, true);   \
        else GH_SCAN5(LL, 0, false, false, true);               \
    } while (0)
        if (l2) GH_SCAN_U(true);
        else GH_SCAN_U(false);
#undef GH_SCAN_U
    } else if (pqc_fused) {
        if (l2 && M == 16) GH_SCAN4(true, 16, true, true);
        else if (l2) GH_SCAN4(true, 32, true, true);
        else if (M == 16) GH_SCAN4(false, 16, true, true);
        else GH_SCAN4(false, 32, true, true);
    } else if (cf) {
#define GH_SCAN_CF(KERN)                                                                                                \
    hipLaunchKernelGGL((KERN), grid, dim3(256), lds, s, x, nq, d, M, P, G,                                              \
                       probe_list, coarse_dis, cc, st2, T2, list_off, list_len, list_mask, nlist, codes, ids, pair_off,  \
                       q_stride, out, ftab, qfil, need_ids, INFINITY, qperm, pg_lo, pg_cnt, sparse, sb, rq_list,         \
                       rq_count, chunk_len)
        if (sb.prod_cf) {
            if (M == 16) GH_SCAN_CF(k_ivfpq_scan_pair_pcf<16>);
            else GH_SCAN_CF(k_ivfpq_scan_pair_pcf<32>);
        } else if (sb.prod_c8) {
            GH_SCAN_CF(k_ivfpq_scan_pair_pc8<16>);
        } else if (sb.c8 && M == 16) {
            GH_SCAN_CF(k_ivfpq_scan_pair_c8<16>);
        } else if (sb.c8) {
            GH_SCAN_CF(k_ivfpq_scan_pair_c8m32);
        } else {
            if (M == 16) GH_SCAN_CF((k_ivfpq_scan_pair<true, 16, true, false, false, true>));
            else GH_SCAN_CF((k_ivfpq_scan_pair<true, 32, true, false, false, true>));
        }
#undef GH_SCAN_CF
    } else if (bound) {
        if (l2) GH_SCAN_M(true, true);
        else GH_SCAN_M(false, true);
    } else {
        if (l2) GH_SCAN_M(true, false);
        else GH_SCAN_M(false, false);
    }
#undef GH_SCAN_M
#undef GH_SCAN_RES
#undef GH_SCAN_RES1
#undef GH_SCAN
#undef GH_SCAN4
#undef GH_SCAN5
}

}  // namespace gh
