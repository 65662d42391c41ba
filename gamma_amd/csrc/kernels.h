// kernels.h -- launch wrappers of the gfx950 kernels in kernels.hip (internal to
// libgamma_hip.so; the public surface is include/gamma_hip.h).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gh {

// A launcher that is handed a shape its callers are supposed to rule out does not abort the process (a database server's): it
// records what it refused, launches nothing, and the C-ABI entry point under way fails with GAMMA_HIP_EDEVICE and that text at
// its next GH_CHECK (gamma_hip_internal.h).  Thread-local: a call enqueues from one thread.
void launch_refused(const char* what);
const char* launch_refused_take();   // what was refused since the last take (nullptr: nothing), cleared


// once-per-device initialisation of a kernel attribute (hipFuncSetAttribute is per device; a process may hold handles
// on several): true the first time the calling thread's current device comes by
inline bool first_call_on_device(std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    return (done.fetch_or(bit) & bit) == 0;
}

constexpr int kMaxRange = 8;

// RangeQueryResult on device (table/range_query_result.h:53-67)
struct RangeDesc {
    const uint8_t* bitmap;
    int32_t min_doc, max_doc, min_aligned, b_not_in;
};

// numeric column predicate (IsInRange<T>, index/impl/gpu/gamma_index_ivfpq_gpu.cc:685-727)
constexpr int kMaxField = 8;
struct FieldDesc {
    const void* col;
    int64_t n;
    int32_t dtype, incl;   // incl: bit 0 include_lower, bit 1 include_upper
    int64_t lo_i, hi_i;
    double lo_f, hi_f;
};

// term filter on a STRING field (FilteredByTermFilter, index/impl/gpu/gamma_index_ivfpq_gpu.cc:727-762): the
// field's items (split at \001) are dictionary-encoded on the host; doc i holds tok[off[i] .. off[i + 1])
constexpr int kMaxTerm = 8, kMaxTermItems = 16;
struct TermDesc {
    const int64_t* off;
    const int32_t* tok;
    int64_t n;              // docs in the column
    int32_t op, n_items;    // FilterOperator (table/field_range_index.h:23): 0 And, 1 Or, 2 Not
    int32_t items[kMaxTermItems];   // -1: an item no doc has
};

// everything GammaSearchCondition::IsValid reads (common/gamma_common_data.h:99-108)
struct FilterDesc {
    const uint8_t* del_bitmap;
    int64_t del_bits;
    const int* vid2doc;      // VIDMgr::VID2DocID of multi-vector documents; nullptr: docid == vid
    int64_t n_vid2doc;
    int32_t has_range, n_range;
    RangeDesc range[kMaxRange];
    int32_t n_field, n_term;
    FieldDesc field[kMaxField];
    TermDesc term[kMaxTerm];
};

void launch_pairwise(hipStream_t s, bool l2, const float* x, int nq, int d, const float* y,
                     int64_t ny, float* out, int64_t ld_out);
// field / term clauses of f -> one bit per document id in [0, nbits) (out: ceil(nbits / 64) * 8 bytes); f must carry no
// range bitmaps, no delete bitmap and no vid -> doc map (it is evaluated on document ids)
void launch_filter_bitmap(hipStream_t s, const FilterDesc& f, int64_t nbits, uint8_t* out);
// every list cut down to the entries that pass ftab[0] (delete bitmap, superseded slots, every clause), in order,
// at the same offsets of (out_codes, out_ids); out_len: the new lengths
void launch_compact_lists(hipStream_t s, const int64_t* list_off, const int* list_len, int nlist, const uint8_t* codes,
                          const int64_t* ids, int code_size, const FilterDesc* ftab /* device */, uint8_t* out_codes,
                          int64_t* out_ids, int* out_len, const float* sums = nullptr, float* out_sums = nullptr);
void launch_pairwise_filtered(hipStream_t s, bool l2, const float* x, int nq, int d,
                              const float* y, int64_t ny, float* out, int64_t ld_out,
                              const FilterDesc& filt, float min_score, float max_score,
                              int64_t row_base);
// Flat search with a running bound: candidate lists of the queries of one call
struct FlatEmit {
    const uint32_t* tau;        // [nq] key bound per query (0xff7fffff = anything valid)
    unsigned long long* cand;   // [nq][cap] (key << 32 | row id)
    int* cnt;                   // [nq * cstride] items appended (may exceed cap: overflow); query q's counter at q * cstride
    int cap;
    int cstride = 1;            // counters one 128-byte line apart (32): appends of different queries do not serialise on a line
};
bool pairwise_can_emit(int nq, int d, int64_t ny);
int flat_list_cap();
void launch_pairwise_emit(hipStream_t s, bool l2, const float* x, int nq, int d, const float* y, int64_t ny,
                          const FilterDesc& filt, float min_score, float max_score, int64_t row_base,
                          const FlatEmit& em);
// candidate lists from the first chunk's top-k (vals / positions inside the chunk starting at row r0)
void launch_flat_init(hipStream_t s, bool l2, const float* vals, const int* pos, int nq, int k, int64_t r0,
                      const FlatEmit& em, uint32_t* tau, int* kept = nullptr);   // kept: FlatLog::kept
// per query: keep the k best of its list (sorted), set the new bound; lists that overflowed set *overflow
// log (exact ties): what pass `pass` (1..) appended to a query's list, kept for the replay -- items [nq][nsl][cap] in the
// layout of the scan's survivor slices, counts [nq][nsl]; kept[nq]: list entries that are older than the pass
struct FlatLog {
    unsigned long long* items = nullptr;
    int* cnt = nullptr;
    int* kept = nullptr;
    int nsl = 0, pass = 0;
    unsigned long long* dbg = nullptr;   // GAMMA_HIP_COMPACT_DBG: phase clocks of query 0
};
void launch_flat_compact(hipStream_t s, int nq, int k, const FlatEmit& em, uint32_t* tau, int* overflow,
                         const FlatLog* log = nullptr);
// sorted lists -> distances / labels (neutral / -1 padded)
void launch_flat_final(hipStream_t s, bool l2, int nq, int k, const FlatEmit& em, float neutral, float* distances,
                       int64_t* labels);
void launch_row_norms(hipStream_t s, const float* y, int64_t n, int d, float* out);
// Flat search on the matrix pipe (flat_mfma.hip): a bf16 hi / lo filter with a proven margin + the survivors' exact
// distances -- appends to the FlatEmit lists exactly the items launch_pairwise_emit would append.
bool flat_filter_supported(int nq, int d, int64_t ny);
int64_t flat_filter_pair_cap(int nq);
int flat_filter_counter_bytes();   // the pair list's counters (one per segment, a cache line apart): zeroed before a pass
size_t flat_filter_query_image_bytes(int nq, int d);
void launch_flat_prep_queries(hipStream_t s, const float* x, int nq, int d, void* image);
size_t flat_filter_bounds_bytes(int nq);
// survivors of the pass as (query, store row) pairs in any order: pairs [cap] of 8 bytes, *npairs appended (zeroed by the caller)
void launch_flat_filter(hipStream_t s, bool l2, int d, const void* qimage, const float* xn, const uint32_t* tau, float* bounds,
                        int nq, const float* y, int64_t ny, int64_t row_base, void* pairs, int* npairs, int64_t cap);
void launch_flat_exact(hipStream_t s, bool l2, const void* pairs, const int* npairs, int64_t cap, const float* x, int nq, int d,
                       const float* store, const FilterDesc& filt, float min_score, float max_score, const FlatEmit& em,
                       int* overflow);
// How the compiled reference's sgemm_ (MKL, the BLAS faiss links in the reference's build) sums the K dimension of
// exhaustive_L2sqr_blas's x . y^T (faiss:utils/distances.cpp:215-296), measured against the compiled library
// (oracle/gamma_oracle.c go_gemm_k_split, tests/test_oracle_vs_ref.py): K <= 384 -- one k-ascending fma chain per
// element, which is what v_mfma_f32_32x32x2_f32 computes; 384 < K <= 768 with K % 8 == 0 -- two chains, [0, K/2) and
// [K/2, K), each from zero, added once; beyond that MKL's blocking is not restated (one chain).  Returns the split
// point, 0 = none.  With it the GEMM-form coarse distances are the library's, bit for bit.
inline int gemm_k_split(int d) { return (d > 384 && d <= 768 && d % 8 == 0) ? d / 2 : 0; }
void launch_l2_gemmform(hipStream_t s, const float* x, int nq, int d, const float* y, int64_t ny,
                        const float* xn, const float* yn, float* out, int64_t ld_out,
                        bool use_mfma);
void launch_pq_ip_table(hipStream_t s, const float* x, int nq, int d, int M, const float* pqc,
                        float* out);
void launch_precompute_table(hipStream_t s, const float* cc, int nlist, int d, int M,
                             const float* pqc, float* out);
// per-call state cleared by the pair-offset kernel (one byte and one 8-byte word per query, two counters); null = skip
struct PairZero {
    uint8_t* bytes = nullptr;
    unsigned long long* words = nullptr;
    int* count_a = nullptr;
    int* count_b = nullptr;
    // histogram pass of the grid-wide query order (launch_query_order with hist_done): list_rank, qkey[nq], bins
    const int* qo_rank = nullptr;
    int* qo_key = nullptr;
    int* qo_bins = nullptr;
};
void launch_pair_offsets(hipStream_t s, const int* probe_list, int nq, int P, const int* list_len,
                         const uint8_t* list_mask, int nlist, int* pair_off, int* q_total,
                         unsigned long long* scan_codes, const int64_t* list_off = nullptr,
                         int64_t* pair_base = nullptr, const PairZero* zero = nullptr);
void launch_compact_probes(hipStream_t s, const int* probe_in, const float* cdis_in, int nq, int P,
                           const int* list_len, const uint8_t* list_mask, int nlist, int* probe_out,
                           float* cdis_out, int P_out = 0);   // P_out: row length of the output (0: P)
// two-phase shard search: the producers' bounds as floats (+-inf: none) for the reduction across shards, and back
void launch_bound_export(hipStream_t s, bool l2, const unsigned long long* ready, int nq, float* out);
void launch_bound_import(hipStream_t s, bool l2, const float* in, int nq, unsigned long long* ready);
void launch_fill_f32(hipStream_t s, float* p, int n, float v);
void launch_bound_combine(hipStream_t s, float* acc, const float* in, int n, int take_max);
void launch_pq_ip_table_rows(hipStream_t s, const float* x, int d, int M, const float* pqc, float* out, const int* rq_list,
                             const int* rq_count);
// inner-product scan: dis0[q][p] = <x_q, centroid of probe p> (fvec_inner_product order); the scan takes it
// through its coarse_dis argument
void launch_pair_ip(hipStream_t s, const float* x, const float* cc, const int* probe_list, int nq, int P, int d,
                    int nlist, float* out);
// threshold pre-filter of the scan (kernels.hip, k_ivfpq_scan_pair<.., FILT>)
struct ScanBound {
    unsigned long long* ready;  // [nq] 0 = not yet published; (1 << 32 | key bound) = bound valid;
                                // (2 << 32) = no bound (unfiltered selection); zeroed per launch
    unsigned long long* surv;   // [nq][groups][scan_slice_cap()] candidates within the bound, per workgroup,
                                // as (key << 32 | position in the query's segment)
    int* gcnt;                  // [nq][groups] items per slice; > scan_slice_cap() = overflowed
    int K;                      // recall_num
    int cnt_stride;             // gcnt[q * cnt_stride + group]  (groups of this launch; more with pair slices behind)
    int store_all;              // != 0: consumer groups store their distances even with a bound (exact-ties replay)
    int* rq_list;               // queries k_select_final could not finish from the slices (launch_ivfpq_scan_repair)
    int* rq_count;
    const float* sums;          // filter pass of the L2 consumers (k_ivfpq_scan_pair<.., CF>): per arena entry
                                // sum_m T2[list][m][code[m]]; per list sum_m max_c |T2[l][m][c]|.  nullptr: regular loop
    const float* t2max;
    int cf_span;                // filter pass: probes per consumer group behind the producer's G (0: one consumer takes them all)
    int spins;                  // sleeps a consumer waits for its producer's bound before it goes on without one (0: 2048)
    unsigned long long* timeouts;   // consumers that gave up waiting (diagnostics; may be nullptr)
    int slice_cap;              // items a slice holds: scan_slice_cap(K)
    int prod_cf;                // != 0 (filter-pass launches only): the PRODUCER scores its probes with the query's table + the per-code
                                // sums too -- no per-list table -- bounds from those values plus their error margin, and gives only its
                                // candidates the exact arithmetic.  Its slab segment then holds APPROXIMATE values: the callers re-score
                                // group 0 (repair launch) for every query whose slab is read (unfiltered selection, tie replay)
    int batch;                  // queries per XCD by which the producers run ahead of the consumers (0: 64)
    int part;                   // two-phase shard search: 1 = only the producers of this launch work (the consumers leave at once),
                                // 2 = only the consumers (the bounds are in `ready` already: imported after the reduction across shards)
    int dbg_part;               // timing experiments only (GAMMA_HIP_SCAN_PART): 1 = consumers leave at once, 2 = producers do
    int c8;                     // != 0 (filter-pass launches only): the consumers' filter pass gathers from a BYTE image of the query's
                                // table made in the workgroup (2-way bank conflicts at most instead of ~3.5); candidates as ever
    const int64_t* pair_base;   // filter pass: [nq][P] arena offset of the pair's list (k_pair_offsets) -- with pair_off the pass needs no
                                // look-up through the list id in front of a list
    float t2max_all;            // filter pass: max over t2max (the margin's bound S without a look-up per list)
    int prod_c8;                // != 0 (with c8, M = 16, one group per launch, two slices per query): ONE workgroup per query -- the bound from
                                // byte-image estimates of the first probe group (their recall_num-th smallest + the image's proven error
                                // width), then the filter pass over ALL probes in the same workgroup (scan.hip, "one workgroup per query").
                                // The first group's slab segment is NOT written: the callers score group 0 (repair launch) for every
                                // query whose slab is read (unfiltered selection, tie replay), as with prod_cf; slice 0 holds the
                                // survivors of every probe (TieReplayArgs::slice0_all)
};
// ---- q8scan.hip: the consumer probes of a bounded L2 scan, list-major over byte tables (one list x 8 queries per tile) ----
struct Q8Args {
    int nq, P, G, M, nlist;
    double mean_len;                  // codes per list (short lists take the pipelined filter kernel)
    const int* probe_list;            // [nq][P]
    const float* coarse_dis;          // [nq][P] dis0
    const float* st2;                 // [nq][M][256] fp32 inner-product tables (k_pq_ip_table)
    const float* fx = nullptr;        // != nullptr: no st2 -- the tables are computed from the queries [nq][d] and the codebook
    const float* pqc = nullptr;
    int d = 0;
    const float* xd = nullptr;        // != nullptr (with pqc, d): k_q8_exact computes the table entries of a query's candidates on
                                      // demand when they are few -- the queries [nq][d]
    const float* T2;                  // [nlist][M][256]
    const float* t2max;               // [nlist]
    const float* sums;                // per arena entry: sum_m T2[list][m][code[m]]
    const uint8_t* codes;
    const int64_t* ids;
    const int64_t* list_off;
    const int* list_len;
    const uint8_t* list_mask;
    const int* pair_off;              // [nq][P + 1]
    const unsigned long long* ready;  // [nq] the producers' bounds
    const FilterDesc* ftab;           // entry 0: the call's validity predicate (need_ids)
    int need_ids;
    unsigned long long* surv;         // survivor slices [nq][cnt_stride][slice_cap]; slices 1 .. of the consumer probe groups are written here
    int* gcnt;
    int cnt_stride, slice_cap;
    int* rq_list;                     // queries without a bound are appended (their consumer groups are scored by the repair launch)
    int* rq_count;
    uint8_t* q8;                      // workspace [nq][M][256]
    float4* meta;                     // workspace [nq]
    uint32_t* cand;                   // workspace [nq][q8_cand_cap(nq)]
    int* iwork;                       // workspace q8_int_words(..) ints
};
bool q8_supported(int M, int P, int G, int64_t q_stride);
int q8_cand_cap(int nq);
size_t q8_int_words(int nq, int P, int G, int nlist);
void launch_q8_consumers(hipStream_t s, const Q8Args& a);
int scan_slice_cap(int K);   // 1024 up to recall_num 256, 2048 up to 1024
void launch_rq_nobound(hipStream_t s, const unsigned long long* ready, int nq, int* rq_list, int* rq_count);
// true when launch_ivfpq_scan_pair would run the filter pass (CF) for a bounded scan with these arguments; the caller
// then launches TWO groups per query -- the producer's G probes and one consumer group with all the others
bool scan_cf_applies(bool l2, int M, int P, int G, bool have_sums, bool store_all);
int scan_group_size(int nq, int P, int G0 = 8);   // G0: probes per workgroup to start from; shrinks through the powers of two below it
void launch_ivfpq_scan_pair(hipStream_t s, bool l2, const float* x, int nq, int d, int M, int P,
                            const int* probe_list, const float* coarse_dis, const float* cc,
                            const float* st2, const float* T2, const int64_t* list_off,
                            const int* list_len, const uint8_t* list_mask, int nlist,
                            const uint8_t* codes, const int64_t* ids, const int* pair_off,
                            int64_t q_stride, float* out, const FilterDesc* ftab /* device */, const int* qfil /* device, may be null */,
                            int need_ids,
                            const int* qperm, int G, int pg_lo, int pg_cnt, int sparse, const ScanBound* bound,
                            const float* pqc_fused = nullptr,   // != nullptr: query table computed in the kernel
                            const int* rq_list = nullptr, const int* rq_count = nullptr,   // repair launch (kernels.hip)
                            int chunk_len = 0, int max_units = 0);   // > 0: rq_list is a work list of list chunks (kernels.hip)
int query_order_bins();
bool query_order_grid(int nq);   // the batch is sorted over the whole grid (needs bins)
// bins: 2 * query_order_bins() ints (histogram | cursors) for the grid-wide sort of large batches; the histogram half is
// zero on entry and left zero.  hist_done: qkey and the histogram were filled by the pair-offset kernel (PairZero::qo_*)
void launch_query_order(hipStream_t s, const int* probe_list, int nq, int P, const int* list_rank,
                        int nlist, int* qkey, int* qperm, int* bins = nullptr, bool hist_done = false);
void launch_sum_totals(hipStream_t s, const int* q_total, int nq, unsigned long long* acc);
// list shard over a supplied assignment: out_max[0] = the longest candidate row of the batch over the lists scanned here,
// out_max[1] = the most owned, non-empty probes of any query (TWO words)
void launch_max_local_total(hipStream_t s, const int* probe_list, int nq, int P, const int* list_len,
                            const uint8_t* list_mask, int nlist, int* out_max);
int select_kpad(int K);
// coarse quantizer selection (ties at the K-th distance resolved like the reference's heap); tie_flag: nq bytes
// coarse.hip: coarse quantizer of a large batch without the distance matrix (sample -> bound -> filtered GEMM
// epilogue -> merge; see the file header).  ws: pl.bytes of scratch.
constexpr int kCoarseCap = 128, kCoarseRepairGrid = 256;
struct CoarseFusedPlan {
    int sample, nseg, tiles_per_strip, cap, cap_stride;
    size_t off_mat, off_tau, off_cand, off_cnt, off_ovf, off_scratch, off_full, bytes;
};
bool coarse_fused_supported(int nq, int d, int nlist, int P, bool exact_ties = false);
CoarseFusedPlan coarse_fused_plan(int nq, int nlist, int P, int cap, bool exact_ties = false);
// rows[0] distance rows mat[i][0..nlist) of the queries rows[1 + i], each walked through faiss's result heap
// (select.hip k_coarse_heap_fix): the K nearest in the reference's order, ties included
void launch_coarse_heap_rows(hipStream_t s, const float* mat, int nlist, int nq, int K, const int* rows, float* out_vals,
                             int* out_pos, unsigned long long* tie_stats);
void launch_coarse_fused(hipStream_t s, const CoarseFusedPlan& pl, void* ws, const float* x, int nq, int d,
                         const float* y, int nlist, const float* yn, int P, float* out_dis, int* out_idx,
                         bool exact_ties = false, unsigned long long* tie_stats = nullptr,
                         hipStream_t side = nullptr, hipEvent_t fork = nullptr, hipEvent_t join = nullptr);
// small batches: exact coarse distances [nq][nlist] + inner-product tables [nq][M][256] in one launch; false = shape
// not covered (nq > 16), nothing launched
bool launch_small_coarse_ip(hipStream_t s, const float* x, int nq, int d, const float* cc, int nlist, float* mat, int M,
                            const float* pqc, float* st2, int* zero_me = nullptr);
// small batches (select.hip): coarse top-nprobe + slab offsets in one kernel; ADC top-recall_num + ids + exact
// re-rank + top-k + output in one kernel.  P <= 64, R <= 1024.
// pair_ip != nullptr (inner-product metric): also dis0 = <x_q, centroid> of every probe in fvec_inner_product order
void launch_small_coarse_select(hipStream_t s, const float* mat, int nlist, int nq, int P, float* out_vals, int* out_pos,
                                const int* list_len, const uint8_t* list_mask, const int64_t* list_off, int* pair_off,
                                int* q_total, int64_t* pair_base, const float* x = nullptr, const float* cc = nullptr,
                                int d = 0, float* pair_ip = nullptr, uint32_t* units = nullptr, int* unit_count = nullptr,
                                int chunk_len = 0, int exact_ties = 0, unsigned long long* tie_stats = nullptr);
struct TieReplayArgs;   // below
void launch_small_tail(hipStream_t s, bool l2, const float* slab, int64_t q_stride, const int* q_total, int nq, int R, int P,
                       const int* probe_list, const int* pair_off, const int64_t* list_off, const int64_t* ids,
                       float* cand_dis, int* cand_pos, int64_t* cand_ids, int has_rank, const float* x, int d,
                       const float* raw, int64_t nraw, int k, float min_score, float max_score, float neutral,
                       float* distances, int64_t* labels, int smax = 0, float* pre_val = nullptr, int* pre_pos = nullptr,
                       int fixed_n = 0,    // q_total == nullptr (flat search): every row has fixed_n entries, position = vector id
                       const TieReplayArgs* tr = nullptr,   // exact ties: a query with a tie at a cut is replayed in the kernel
                       unsigned long long* tie_stats = nullptr);
// IVFFLAT: exact distances of every entry of the probed lists (rows from the raw store) into the query's slab
void launch_ivfflat_scan(hipStream_t s, bool l2, const float* x, int nq, int d, int P, const int* pair_off,
                         const int64_t* pair_base, const int64_t* ids, const float* raw, int64_t nraw, int64_t q_stride,
                         float* out, const FilterDesc* ftab, int need_filter, float min_score, float max_score);
// IVFFLAT, list-major (ivfflat.hip): one workgroup per list, every query probing it run past the list's rows
bool ivfflat_lm_supported(int d);
size_t ivfflat_lm_scratch_bytes(int nq, int P, int nlist);
void launch_ivfflat_lm(hipStream_t s, bool l2, const float* x, int nq, int d, int P, const int* probe, const int* pair_off,
                       const int64_t* list_off, const int* list_len, int nlist, const int64_t* ids, const float* raw,
                       int64_t nraw, int64_t q_stride, float* out, const FilterDesc* ftab, int need_filter, float min_score,
                       float max_score, void* scratch);
// returns true when the walk of the tied rows went to `side` (the caller then waits for `join` before reading the result)
bool launch_coarse_select(hipStream_t s, const float* mat, int nlist, int nq, int K, float* out_vals, int* out_pos,
                          uint8_t* tie_flag, unsigned long long* tie_stats = nullptr, hipStream_t side = nullptr,
                          hipEvent_t fork = nullptr, hipEvent_t join = nullptr);
void launch_select_topk(hipStream_t s, bool smallest, const float* vals, int64_t seg_stride,
                        const int* seg_len, int fixed_len, int max_len, int nseg, int K,
                        float* out_vals, int* out_pos, const uint8_t* only = nullptr);
// threshold pre-filter of the scan: exact top-K from the survivor lists (select.hip); rows that end
// with flag != 0 are left to launch_select_topk(..., only = flag)
void launch_select_final(hipStream_t s, bool smallest, const unsigned long long* surv, const int* gcnt,
                         int nslices, int slice_cap, const unsigned long long* ready, const int* pair_off, int P,
                         int nq, int K, const int64_t* pair_base, const int64_t* ids, uint8_t* flag, float* out_vals,
                         int* out_pos, int64_t* out_ids, uint8_t* cut_tie = nullptr,
                         unsigned long long* tie_stats = nullptr, int* rq_list = nullptr, int* rq_count = nullptr, unsigned long long* bound_stat = nullptr);   // slices 1.. in their own
                                                                                        // array [nq][nslices - 1][cap_c]   // cut_tie[q] = 1: the K-th and (K+1)-th keys are equal
void launch_map_candidates(hipStream_t s, const int* pos, int nq, int R, int P,
                           const int* probe_list, const int* pair_off, const int64_t* list_off,
                           const int64_t* ids, int64_t* cand_ids, const uint8_t* only = nullptr);
void launch_rerank_dist(hipStream_t s, bool l2, const float* x, int nq, int d, const float* raw,
                        int64_t nraw, const int64_t* cand_ids, int R, float min_score,
                        float max_score, float* out, const int32_t* slot = nullptr, int64_t nslot = 0);
// raw vectors sharded with their lists (round 6).  slot [nslot]: vector id -> row of `raw` on THIS shard, -1 = held elsewhere.
// _export_exact: ex[f][j] = exact distance (no score window) of entry j of exported stream row f when its ADC value is
// within bound[f] and its vector is held here, NaN otherwise.  _lookup_exact: cand_exact[q][r] = the exact distance that
// travelled with candidate cand_ids[q][r] in one of the W shard tables (all_ids / all_exact [W][nq][R]), sentinel if none.
void launch_export_exact(hipStream_t s, bool l2, const float* xf, int nf, int d, const float* raw, const int32_t* slot, int64_t nslot,
                         const float* vals, const int64_t* ids, int64_t stride, const int32_t* off, int P, const float* bound, float* ex);
void launch_lookup_exact(hipStream_t s, bool l2, const float* all_dis, const int64_t* all_ids, const float* all_exact, int W, int nq, int R,
                         int q0, int nql, const float* cand_dis, const int64_t* cand_ids, float* cand_exact);
// Exact-tie bookkeeping (ties.hip, gamma_hip_set_exact_ties).  cut[q] != 0: the top-recall_num cut of query q
// went through a group of equal ADC distances (set by the selection kernels).  The final-stage kernels add
// their own condition -- two of the first k+1 final distances equal -- and append every query either one
// holds for to list[*count]; launch_tie_replay then redoes those queries the way the reference's heaps do.
// stats: device counters {coarse rows redone, cut ties, queries replayed}, accumulated over calls.
struct TieFlags {
    const uint8_t* cut = nullptr;
    int* list = nullptr;     // nullptr: exact ties off
    int* count = nullptr;
    unsigned long long* stats = nullptr;
};
void launch_rerank_topk(hipStream_t s, bool l2, const float* x, int nq, int d, const float* raw,
                        int64_t nraw, const int64_t* cand_ids, int R, int k, float min_score,
                        float max_score, float neutral, float* distances, int64_t* labels,
                        const int* qperm = nullptr,   // qperm: run the queries in this order (speed only)
                        const TieFlags* ties = nullptr);
// what k_tie_replay needs to redo one query (ties.hip)
struct TieReplayArgs {
    const int* list;              // flagged queries, *count of them
    const int* count;
    int nq;
    const float* slab;            // ADC distances [nq][q_stride] in scan order, +-inf = filtered entry
    int64_t q_stride;
    const int* pair_off;          // [nq][P + 1]
    const int64_t* pair_base;     // [nq][P] arena offset of each probed list
    const int64_t* ids;           // list arena
    int P, G;                     // probes per query; probes of the first group (bounded scan)
    const unsigned long long* ready;   // bounded scan (may be null): bound word, survivor slices, counts
    const unsigned long long* surv;
    const int* gcnt;
    int nsl, slice_cap;
    const float* x;               // queries [nq][d]
    int d;
    const float* raw;             // raw vectors (has_rank)
    int64_t nraw;
    const float* ex_slab = nullptr;   // raw vectors sharded with their lists (round 6): the exact distance of stream entry ps
                                      // travelled with the shards' exports -- [rows][q_stride] like the slab, NaN where a shard
                                      // did not compute it; read instead of the raw row
    int* ex_missing = nullptr;        // counts R-heap members whose exact distance was NOT exported (must stay 0)
    int R, k, has_rank;
    float min_score, max_score, neutral;
    float* cand_dis;              // [nq][R] recall-stage table, rewritten for the replayed queries
    int64_t* cand_ids;
    float* distances;             // [nq][k]
    int64_t* labels;
    unsigned long long* dbg = nullptr;   // phase clocks of the first replayed query (GAMMA_HIP_TIE_DBG)
    int pop_push = 0;             // the heap takes a candidate with heap_pop + heap_push (IVFFLAT / flat scanners) instead
                                  // of heap_replace_top (the IVFPQ scanner)
    int fixed_n = 0;              // pair_off == nullptr (flat): every row has fixed_n entries, a position IS the vector id
    int compact_rows = 0;         // slab row i belongs to the i-th flagged query (list[i]) instead of query i
    int slice0_all = 0;           // slice 0 holds the survivors of EVERY probe (ScanBound::prod_c8): the slices are walked from slice 0 on and
                                  // entries of the first group (in the slab part already) are skipped
    int always_sliced = 0;        // ready == nullptr: every query is first G slab entries + slices 1.. (flat search with the
                                  // running bound: first row chunk + the candidates each later pass emitted)
};
int tie_replay_max_k();
int tie_small_max_k();
void launch_tie_list(hipStream_t s, const uint8_t* cut, const uint8_t* extra, int nq, int* list, int* count,
                     unsigned long long* tie_stats);
// test hook: one stream through one heap with each form of the sifts (ties.hip)
void launch_debug_heap_stream(hipStream_t s, int op, int k, int n, const float* vals, uint2* out_arr, uint2* out_sorted);
int tie_replay_max_probes();
void launch_tie_replay(hipStream_t s, bool l2, const TieReplayArgs& a);
void launch_flag_cut_ties(hipStream_t s, const float* slab, int64_t q_stride, const int* q_total, int nq, int K,
                          const float* sel_vals, const int* sel_pos, const uint8_t* only, uint8_t* tflag,
                          int fixed_n = 0, int inside = 0);   // q_total == nullptr: rows of fixed_n entries; inside: also
                                                              // equal values among the selected K
int coarse_heap_max_k();
size_t tie_replay_lds_bytes(int R, int k, int P);
// flat search under exact ties: the chunked paths run for k + 1 results; this copies the first k to the caller's rows and
// lists the queries with two equal distances among the k + 1 (their order, or which of them stays, is the heap's)
void launch_flat_take_flag(hipStream_t s, const float* D1, const int64_t* I1, int nq, int k, float* distances, int64_t* labels,
                           int* list, int* count, unsigned long long* tie_stats);
// exact ties across list shards (ties.hip): cut-tie flags of the merged tables, a shard's export of the flagged queries'
// candidate streams, the owner's assembly of the exports
void launch_flag_merge_cut(hipStream_t s, const float* all_dis, int W, int nq, int R, int q0, int nql, const float* merged,
                           const int64_t* merged_ids, uint8_t* tcut, const uint8_t* shard_flags = nullptr);
void launch_shard_export(hipStream_t s, const int32_t* probe, int nf, int P, const int* list_len, const int64_t* list_off,
                         const uint8_t* list_mask, int nlist, const int64_t* ids, const float* slab, int64_t q_stride,
                         int64_t stride, float* vals, int64_t* out_ids, int32_t* off);
void launch_shard_export_rows(hipStream_t s, const int32_t* probe, int nf, int P, const int* list_len, const uint8_t* list_mask,
                              int nlist, int* max_entries);
void launch_merge_streams(hipStream_t s, int W, int nf, int P, int64_t stride, int64_t mstride, const float* vals, const int64_t* ids,
                          const int32_t* off, float* m_vals, int64_t* m_ids, int32_t* m_off, int64_t* m_base, float sentinel);
void launch_gather_words(hipStream_t s, const void* src, const int* list, int n, int words, void* out);
// out[i] = x[list[i]] for i < n (rows of d floats)
void launch_gather_rows(hipStream_t s, const float* x, const int* list, int n, int d, float* out);
void launch_finalize_topk(hipStream_t s, const float* sel_vals, const int* sel_pos, int nq, int k,
                          const int64_t* src_ids, int64_t src_stride, int64_t id_base,
                          float neutral, float* distances, int64_t* labels);
void launch_finalize_norank(hipStream_t s, const float* cand_dis, const int64_t* cand_ids, int nq,
                            int R, int k, float min_score, float max_score, float neutral,
                            float* distances, int64_t* labels, const TieFlags* ties = nullptr);
void launch_gather_shards(hipStream_t s, const float* all_dis, const int64_t* all_ids, int nshards,
                          int nq, int R, float* dis, int64_t* ids, float sentinel);
// one-kernel merge of the shard tables [W][nq][R] for queries [q0, q0 + nql) -> [nql][R]; false = shape
// not covered (R > 256 or W * R > 2048), use launch_gather_shards + launch_select_topk + launch_take_ids
bool launch_merge_shards(hipStream_t s, bool smallest, const float* all_dis, const int64_t* all_ids, int W,
                         int nq, int R, int q0, int nql, float* out_dis, int64_t* out_ids);
void launch_take_ids(hipStream_t s, const int* pos, const int64_t* src_ids, int64_t src_stride,
                     int nq, int R, int64_t* out);
void launch_bitmap_set(hipStream_t s, uint8_t* bm, const int64_t* docids, int64_t n, int64_t nbits,
                       int value);
void launch_mark_moved(hipStream_t s, int64_t* ids, int64_t pos);
void launch_list_checksum(hipStream_t s, const uint8_t* codes, const int64_t* ids, const int64_t* off, const int* len, int nlist,
                          int M, int max_len, unsigned long long* out);
void launch_repack_lists(hipStream_t s, const uint8_t* oc, const int64_t* oi, uint8_t* nc, int64_t* ni,
                         const int64_t* old_off, const int64_t* new_off, const int* len, int nlist, int M,
                         int max_len);
// per-code table sums of the L2 scan's filter pass and the per-list bound of their magnitude (kernels.hip)
void launch_code_sums_ranges(hipStream_t s, const float* T2, const uint8_t* codes, int M, const int* list_no, const int64_t* pos,
                             const int* n, int nranges, int max_n, float* sums);
void launch_code_sums_one(hipStream_t s, const float* T2, const uint8_t* codes, int M, int list_no, int64_t pos, int n,
                          float* sums);
void launch_code_sums_lists(hipStream_t s, const float* T2, const uint8_t* codes, int M, const int64_t* list_off,
                            const int* list_len, int nlist, int max_len, float* sums);
void launch_t2_rowmax(hipStream_t s, const float* T2, int nlist, int M, float* t2max);
void launch_centroid_update(hipStream_t s, const float* x, int d, const int* order, const int* seg, int k, float* centroids,
                            float* hassign);
void launch_pq_encode(hipStream_t s, const float* x, int64_t n, int d, int M, const int* assign,
                      const float* cc, const float* pqc, uint8_t* codes);

}  // namespace gh
