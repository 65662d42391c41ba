// gamma_hip_internal.h -- the handle of libgamma_hip.so (HBM-resident state, workspaces, locks) and the helpers its
// translation units share: gamma_hip.cpp (life cycle, switches, accounting), gamma_hip_store.cpp (realtime lists,
// raw store, columns, bitmap: the writers), gamma_hip_search.cpp (the search pipelines and their entry points).
#pragma once
#include "../../include/gamma_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"

namespace ghi {

constexpr double kPI = 3.14159265;  // realtime/realtime_mem_data.h:24
constexpr int64_t kDelMask = (int64_t)(1ULL << 63);

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            if (e != hipSuccess) return e;
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

struct StageEvent {
    int stage;
    hipEvent_t a, b;
    bool count;
};

}  // namespace ghi
using ghi::DevBuf;
using ghi::StageEvent;

struct WriteLock;

// A device array that grows in place: an address range reserved once (the device's whole memory -- addresses are free),
// physical chunks mapped behind what is in use.  Nothing moves, so readers of the mapped part go on while it grows.
struct VmRange {
    char* base = nullptr;
    size_t va_bytes = 0, mapped = 0, gran = 0;
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    std::vector<size_t> chunk_bytes;
    bool tainted = false;   // unmapped without the fence (unmap_all): must not be mapped again
    bool on() const { return base != nullptr; }
    // false: the runtime does not offer it (nothing is left behind)
    bool reserve(int device_, size_t chunk_min);
    // mapped >= bytes afterwards; grows by at least 1/8 of what is mapped, in whole chunks.  On failure *what names the
    // step and nothing of the failed step stays mapped.
    hipError_t map_to(size_t bytes, const char** what);
    void release();
    // the physical memory goes back, the address range stays reserved (mapped = 0)
    void unmap_all();
};

// GEMM-form distance calls (coarse quantizer, assign, k-means: faiss's exhaustive_L2sqr_blas, faiss:utils/distances.cpp:215-296)
// whose sgemm_ kernel is NOT restated (oracle/gamma_oracle.c, go_gemm_k_split): K beyond 768 or an odd split, K = 384 with
// a database remainder block of 9..512 rows, remainder blocks of a few rows (nx mod 4096 / ny mod 1024 in 1..7: MKL takes
// another kernel there -- measured: the rows of a 1..3-row query remainder differ in ~10 % of their entries by an ulp).
// Such calls are COUNTED (gamma_hip_blas_form_not_restated), never silent.
inline bool blas_form_not_restated(int64_t nx, int64_t ny, int d) {
    if (d > 768 || (d > 384 && (d & 7))) return true;
    const int64_t rx = nx % 4096, ry = ny % 1024;
    if ((rx >= 1 && rx <= 7) || (ry >= 1 && ry <= 7)) return true;
    return d == 384 && ry >= 9 && ry <= 512;
}

struct gamma_hip_index {
    int device = 0;
    // Concurrency (SURVEY 8b "Threading": Search from any number of client threads while ONE indexing thread adds
    // and API threads delete; the reference's lists are lock-free for readers, realtime_mem_data.cc:279-300):
    //   stream / search_mu : searches.  One at a time (they share the workspaces); search_mu is held for a whole
    //                        call, mu only while the call reads the handle's state and enqueues its kernels.
    //   wstream / writer_mu: writers (list appends, encode, raw / bitmap / column updates) on their own stream, so
    //                        they neither wait for the searches in flight nor hold them up; mu while they work.
    //   list meta versions : a search's kernels read the (offset, length) table of the VERSION that was current
    //                        when it was enqueued; a writer publishes a new version after its copies (the
    //                        reference publishes retrieve_idx_pos_ after the copy, realtime_mem_data.cc:299-300).
    //                        Old extents stay intact inside the arena, so a search in flight keeps reading a
    //                        consistent prefix of the insert log.
    //   reallocation       : growing the arena / raw store / bitmap frees memory a search in flight may read --
    //                        the writer then takes search_mu too and drains both streams first (WriteLock::exclusive).
    // Lock order: writer_mu -> search_mu -> mu.
    hipStream_t stream = nullptr, wstream = nullptr;
    // side stream of the searches: kernels of a call that may overlap its main chain (the heap replay of the coarse
    // rows with a tie, beside the query tables); forked from and joined into `stream` by events, never used alone
    hipStream_t side = nullptr;
    hipStream_t side2 = nullptr;   // the deferred tie replay (its own stream: the coarse heap fix must not queue behind it)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool coarse_join_pending = false;
    // deferred tie replay (gamma_hip_set_deferred_replay): the replay of a device-pointer call's flagged queries runs on
    // the side stream and the search stream waits for it only before the next kernel that touches what it reads (the
    // next call's or chunk's pair offsets) -- its latency hides behind the next coarse quantizer and query tables
    hipEvent_t ev_rfork = nullptr, ev_rdone = nullptr;
    bool defer_replay = false, defer_now = false, replay_pending = false;
    // gamma_hip_ivfpq_search_device_wait: completion events of the calls in flight (rotating; a waiter that finds its slot
    // re-recorded by a later call waits for that one -- later implies earlier on in-order streams)
    hipEvent_t ev_call[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned call_seq = 0;
    // large host-buffer IVFPQ calls from several client threads (ivfpq_search_host_overlap): a call's queries go up on
    // `up_stream` into its own staging slot BEFORE it takes the search lock, its results come down behind its tie replay
    // on the side stream into pinned memory, and the caller waits for its own event with the handle already free
    struct HostSlot {
        DevBuf x, D, I;
        void* pin = nullptr;
        size_t pin_bytes = 0;
        hipEvent_t ev_up = nullptr, ev_done = nullptr;
        bool busy = false;
    };
    HostSlot hslot[2];
    std::mutex hs_mu;
    std::condition_variable hs_cv;
    hipStream_t up_stream = nullptr;
    // flat search with overlapping callers (gamma_hip_flat_search_device_wait): what a flat call's tie replay reads -- the first
    // row chunk's slab, the survivor log, the flag list, the candidate tables -- exists TWICE; a call that finds a flat replay
    // pending switches to the other bank instead of waiting, and waits (in stream order) only for the replay that last read
    // the bank it is about to overwrite
    DevBuf fbank[5];
    int flat_bank = 0;
    hipEvent_t ev_bank[2] = {nullptr, nullptr};
    bool bank_used[2] = {false, false};
    bool replay_is_flat = false;
    // (the flat call's "did a survivor list overflow" word is read by the host; a _wait call reads it AFTER its completion
    //  event, with the handle free, from one of four pinned words, and redoes the call without a bound in the rare yes)
    int* pin_flat_over = nullptr;
    int* flat_over_dst = nullptr;    // set by the _wait entry for the call under way
    bool flat_no_bound = false;
    std::atomic<int> big_calls_in_flight{0};
    std::atomic<int64_t> big_calls_overlap_seen_ns{INT64_MIN / 2};
    // the shadow lists of compact_lists_for_call stay valid while nothing was written and the call has no clauses of its own
    // (standing deletes): write_gen counts writer calls (WriteLock), cmp_gen = the count the shadow lists were built at
    uint64_t write_gen = 1, cmp_gen = 0;
    bool cmp_has_sums = false, cmp_sums_built = false;
    bool cmp_by_clause = false;   // the shadow lists were cut under a request's own clauses (as short as those make them)
    bool merge_flags = false;   // the last gamma_hip_ivfpq_merge_rerank left tie flags of its slice in w_tlist
    int shard_cut_nq = 0;       // the last shard search left the cut-tie flags of its nq queries in w_tcut
    bool shard_cut_chunked = false;   // ... of a call of several chunks: gathered in w_shard_cut
    const uint8_t* merge_shard_flags = nullptr;   // [nshards][nq] for the next merge (gamma_hip_ivfpq_merge_set_shard_flags)
    int merge_nql = 0;
    std::mutex mu, search_mu, writer_mu;
    WriteLock* wl = nullptr;   // the writer holding mu (for exclusive() deep inside the arena code)
    static constexpr int NVER = 4;
    int64_t* d_ver_off[NVER] = {nullptr, nullptr, nullptr, nullptr};
    int* d_ver_len[NVER] = {nullptr, nullptr, nullptr, nullptr};
    void* pin_ver[NVER] = {nullptr, nullptr, nullptr, nullptr};   // pinned staging of a version's tables
    hipEvent_t ver_ev[NVER] = {nullptr, nullptr, nullptr, nullptr};   // wstream: the version's tables are in place
    hipEvent_t rd_ev[NVER] = {nullptr, nullptr, nullptr, nullptr};    // stream: the last search reading it is past its list kernels
    bool rd_set[NVER] = {false, false, false, false};
    int cur_ver = 0;
    std::string err;

    // raw vector store
    int raw_d = 0;
    float* d_raw = nullptr;
    int64_t nraw = 0, raw_cap = 0;
    // raw vectors SHARDED with their lists (gamma_hip_raw_put, round 6): the store holds the rows of the vectors in this shard's
    // lists only, in arrival order; raw_slot[vid] = row (-1: held by another shard).  Such a handle re-ranks nothing by itself --
    // has_rank searches, flat search and raw_gets refuse -- it serves _shard_exact / _shard_export_exact.
    bool raw_sparse = false;
    std::vector<int32_t> h_raw_slot;
    int32_t* d_raw_slot = nullptr;
    int64_t raw_slot_cap = 0;
    // The store grows IN PLACE where the runtime offers virtual memory management: one address range reserved up
    // front, physical chunks mapped behind the rows as they come -- no copy, no second allocation, no wait for the
    // searches in flight (the reference keeps 500 000-vector segments for the same reason, vector/memory_raw_vector.cc:
    // 90-142; a segment table would cost every gather an indirection, a mapped range costs nothing).  Fallback: a
    // geometric reallocation under the exclusive lock.
    bool raw_vmm = false;
    VmRange raw_vm;
    int64_t raw_regrows = 0;   // reallocations that moved the store (0 with virtual memory management)

    // numeric scalar columns (on-device range filters)
    struct Column {
        int dtype = 0;
        uint8_t* d = nullptr;
        int64_t n = 0, cap = 0;
    };
    std::map<int, Column> fields;
    // STRING columns as dictionary-encoded item lists (on-device term filters): doc i = tok[start .. start + len),
    // row word off[i] = start << 16 | len (gamma_hip_term_update rewrites a row)
    struct TermColumn {
        int64_t* d_off = nullptr;
        int32_t* d_tok = nullptr;
        int64_t ndocs = 0, cap_docs = 0, ntok = 0, cap_tok = 0;
        std::vector<int64_t> h_rows;   // host copy of the row words
    };
    std::map<int, TermColumn> terms;

    // delete bitmap
    uint8_t* d_bitmap = nullptr;
    int64_t bitmap_bits = 0;
    size_t bitmap_cap_bytes = 0;
    std::vector<uint8_t> h_bitmap;
    bool bitmap_any = false;   // any delete bit set
    int64_t n_moved = 0;       // inverted-list slots marked superseded (bit 63) since creation

    // IVFPQ model
    bool ivf_init = false, trained = false;
    int d = 0, nlist = 0, M = 0, ksub = 256, dsub = 0, code_size = 0, metric = GAMMA_HIP_METRIC_L2;
    int bucket_init = 1000, bucket_max = 1280000;
    float *d_cc = nullptr, *d_cc_norms = nullptr, *d_pqc = nullptr, *d_T2 = nullptr;
    int* d_list_rank = nullptr;   // spatial order of the coarse centroids (scan locality only)
    bool sort_queries = getenv("GAMMA_HIP_NO_QUERY_SORT") == nullptr;
    bool scan_bound = getenv("GAMMA_HIP_NO_SCAN_BOUND") == nullptr;
    // Feedback for the bounded scan (results are the same either way): the pre-filter pays when the first probe group
    // bounds well.  On data where it does not (full-size C5: noise-dominated inner-product vectors, 2460 survivors per
    // query instead of ~100, every query sent to the unfiltered selection AFTER the filtered scan) the path costs more
    // than it saves.  k_select_final counts the queries it gives up on; the totals come back through a pinned word every
    // few calls, and while more than half of the recent queries fell through the handle scans unbounded, re-probing now
    // and then.
    unsigned long long* d_bound_stat = nullptr;    // 2 slots x {queries given to the unfiltered selection, queries}, then: consumer groups that gave up waiting
    unsigned long long* pin_bound_stat = nullptr;  // the two slots as of some recent call (pinned host memory)
    unsigned long long bound_seen[2] = {0, 0};     // what the last decision had read
    hipEvent_t bound_copy_ev = nullptr;            // behind the last copy into pin_bound_stat
    bool bound_copy_pending = false;
    uint64_t bound_sig = 0;                        // (nprobe, recall_num, metric, filter, shard) of the calls the counts are of
    int bound_epoch = 0;                           // kinds of call seen; its low bit selects the counter slot
    bool bound_feedback_off = false;               // gamma_hip_set_scan_bound_feedback(h, 0): the pre-filter whenever it applies
    int bound_calls = 0, bound_off_calls = 0;      // calls since the last copy; unbounded calls left before the re-probe
    int64_t bound_backoffs = 0;                    // times the handle turned the pre-filter off

    // inverted-list arena
    uint8_t* d_codes = nullptr;
    int64_t* d_ids = nullptr;
    // L2 scan filter pass (k_ivfpq_scan_pair<.., CF>): per arena entry sum_m T2[list][m][code[m]], kept beside the codes
    // by every writer; per list sum_m max_c |T2[l][m][c]| (the margin of the filter).  Null for IVFFLAT / inner-product
    // indexes (the inner-product scan has no per-list table to avoid).
    float* d_sums = nullptr;
    float* d_t2max = nullptr;
    float t2max_all = 0.f;   // max over d_t2max (set with it)
    // L2 table mode (faiss::IndexIVFPQ::use_precomputed_table after train / Load): 1 = the precomputed table T2 is resident,
    // 0 = it would exceed precomputed_table_max_bytes (faiss:IndexIVFPQ.cpp:441-449) -- none is built, L2 searches score with
    // per-(query, list) residual tables (gamma_index_ivfpq.h:239-245).  Decided by Init from nlist * M and the limit.
    int table_mode = 1;
    // what the scan launcher takes as (st2, T2): mode 0 -> (the PQ codebook, nullptr), see scan.hip RES
    const float* scan_st2(bool l2) const { return (l2 && table_mode == 0) ? d_pqc : w_st2.as<float>(); }
    bool keep_sums = false;   // set by Init: IVFPQ handles (GAMMA_HIP_NO_CODE_SUMS=1 turns the filter pass off)
    // list shard over a supplied assignment (gamma_hip_ivfpq_search_shard_preassigned): the longest candidate row of the
    // call, measured on the device -- the slab stride of its chunks (0: nprobe x the longest list)
    int64_t q_stride_cap = 0;
    int64_t arena_cap = 0, arena_used = 0, arena_waste = 0;   // entries; waste = abandoned extents inside used
    // the three arena arrays (codes, ids, code sums) are mapped ranges like the raw store where the runtime allows:
    // growth maps chunks behind them -- no copy, no second arena, no exclusive lock (the reference grows per bucket for
    // the same reason, realtime/realtime_mem_data.cc:152-188,426-474).  Fallback: reallocation under the exclusive lock.
    bool arena_vmm = false;
    VmRange vm_codes, vm_ids, vm_sums;
    VmRange alt_codes, alt_ids, alt_sums;   // the repack's target set (the two sets swap roles; see arena_repack)
    std::vector<VmRange> vm_retired;        // ranges that may not be mapped again (address space only; freed with the handle)
    int64_t arena_regrows = 0;   // growths that moved the arena (0 with virtual memory management)
    int64_t repack_min_entries = 1 << 16;                     // no repack for less waste than this
    int64_t n_repacks = 0;
    int64_t repack_verified = 0, repack_verify_failures = 0;   // read-backs of a repack target / those that differed from the source
    std::vector<int64_t> h_list_off;
    std::vector<int> h_list_len, h_list_cap, h_deleted;
    std::vector<uint8_t> h_extend_time;
    int64_t* d_list_off = nullptr;
    int* d_list_len = nullptr;
    uint8_t* d_list_mask = nullptr;
    std::vector<uint8_t> h_list_mask;
    std::vector<int64_t> vid_pos;
    int max_list_len = 0;
    int64_t ntotal = 0;

    // workspace
    DevBuf w_mat, w_coarse_dis, w_probe, w_xn, w_st2, w_pair_off, w_qtotal, w_dist, w_cand_dis,
            w_cand_pos, w_cand_ids, w_exact, w_selv, w_selp, w_x, w_outd, w_outl, w_stage, w_shard_cut, w_filter,
            w_m_dis, w_m_ids, w_part_v, w_part_i, w_assign, w_codes_tmp, w_qperm, w_qbins, w_scnt, w_sflag, w_surv, w_pair_base, w_q8, w_q8meta, w_q8cand, w_q8int,
            w_pair_ip, w_flat_cand, w_flat_meta, w_full_cdis, w_full_probe, w_ftab, w_qfil, w_tieflag, w_tcut, w_tlist, w_textra, w_fq, w_fraw, w_frcnt, w_lm_units, w_lm_cnt, w_fbits, w_cmp_codes, w_cmp_ids, w_cmp_len, w_cmp_sums, w_fD, w_fI, w_fx, w_fslab, w_flog, w_mr_vals, w_mr_ids, w_mr_meta,
            we_mat, we_cdis, we_x, we_assign, we_codes, we_stage, we_chk;   // writer side (encode, bitmap_set): never shared with a search
    unsigned long long* d_scan_codes = nullptr;
    size_t dist_budget_bytes = (size_t)8 << 30;   // per-chunk ADC distance buffer (288 GB of HBM per GPU)

    // device copy of the filter table of the running call (entry 0 = the call's own descriptor) and the
    // host image of what entry 0 holds, so an unchanged descriptor is not uploaded again
    gh::FilterDesc ftab_shadow;
    bool ftab_valid = false;

    bool exact_ties = true;    // gamma_hip_set_exact_ties
    std::atomic<int64_t> ties_unhonoured{0};   // gamma_hip_ties_not_honoured
    int scan_dbg_now = 0;   // GAMMA_HIP_SCAN_PART (timing experiments)
    std::atomic<int64_t> blas_unrestated{0};   // gamma_hip_blas_form_not_restated
    bool coarse_fused = true;  // gamma_hip_set_coarse_fused
    bool small_path = true;    // gamma_hip_set_small_path
    int small_presel = 0;      // 0: pre-selection by estimate, > 0: always, that many slices (tests)
    // multi-vector documents (VIDMgr::VID2DocID, vector/raw_vector_common.h:90-95): docid of every vid, host + device;
    // empty = single-vector documents, docid == vid.  Every delete-bitmap / filter test goes through it.
    std::vector<int32_t> h_v2d;
    int32_t* d_v2d = nullptr;
    int64_t v2d_cap = 0;
    int64_t doc_of(int64_t v) const { return (v >= 0 && (size_t)v < h_v2d.size()) ? (int64_t)h_v2d[v] : v; }
    bool doc_deleted(int64_t v) const {
        const int64_t dd = doc_of(v);
        return dd >= 0 && dd < bitmap_bits && !h_bitmap.empty() && ((h_bitmap[dd >> 3] >> (dd & 7)) & 1);
    }
    bool ivfflat = false;      // gamma_hip_ivfflat_init: lists of vector ids (1 dummy code byte), rows from the raw store
    int coarse_cap = gh::kCoarseCap;
    unsigned long long* d_tie_stats = nullptr;   // {coarse rows redone, top-R cuts through a tie, queries replayed}
    // what stage A leaves for the tie replay of stage B (ties.hip)
    struct TieCtx {
        bool on = false, bounded = false;
        int G = 0, nsl = 0, cap = 0;
        int64_t q_stride = 0;
        // ScanBound::prod_cf: the slab segment of a query's first probe group holds approximate values; a query whose slab is
        // read (tie replay) gets it re-scored first -- what that launch needs of stage A's arguments
        bool prod_cf = false;
        bool slice0_all = false;   // ScanBound::prod_c8: slice 0 holds the survivors of every probe (TieReplayArgs::slice0_all)
        int need_ids = 0;
        const void* d_ftab = nullptr;
        const int* d_qf = nullptr;
        const float* dis0 = nullptr;
    } tie;

    // last-search stage info
    int last_nq = 0, last_P = 0, last_R = 0;
    const int* last_qperm = nullptr;   // query order of the last stage A (null: arrival order)

    // request combining of small concurrent host-buffer searches (gamma_hip_ivfpq_search)
    struct Waiter {
        const gamma_hip_search_params* p;
        int nq, k, mode;   // mode: coarse path resolved from THIS request's size
        int kind = 0;      // 0: IVFPQ search, 1: flat search
        const float* x;
        float* D;
        int64_t* I;
        int rc = 0;
        // delivery (gamma_hip_search.cpp, combine_worker): every waiter sleeps on its OWN mutex + condition variable; the
        // worker's helper copies a batch's rows into the callers' buffers, links the waiters into a forest, wakes the
        // first FAN, and every woken waiter wakes FAN more on its way out -- a batch of 128 is awake after three hops
        // instead of 128 wake-ups issued by one thread through one mutex
        static constexpr int FAN = 4;
        bool done = false;
        std::mutex wm;
        std::condition_variable cv;
        Waiter* child[FAN] = {nullptr, nullptr, nullptr, nullptr};
    };
    std::mutex comb_mu;
    std::condition_variable comb_wcv;   // the worker waits here
    std::deque<Waiter*> comb_q;
    bool prefiltered = false;           // this call runs over lists compacted under its filter: nothing is left to reject
    bool comb_busy = false;             // a batch (or a direct call) is in flight
    bool comb_stop = false;
    std::thread comb_thread;
    bool combine = getenv("GAMMA_HIP_NO_COMBINE") == nullptr;
    // pinned staging of the combined batches: one set per batch from formation until its last caller has copied its
    // rows out (only the worker allocates them); four, so that forming a batch never waits for a delivery
    static constexpr int NSET = 4;
    void* comb_pin[NSET] = {nullptr, nullptr, nullptr, nullptr};
    void* comb_pin_dev[NSET] = {nullptr, nullptr, nullptr, nullptr};   // their addresses on the device (results are stored in place)
    size_t comb_pin_bytes[NSET] = {0, 0, 0, 0};
    // pinned staging of small direct calls (host_search; the search lock serialises its users)
    void* dir_pin = nullptr;
    void* dir_pin_dev = nullptr;   // its address on the device (hipHostGetDevicePointer), nullptr: not mapped
    size_t dir_pin_bytes = 0;

    // profiling
    int profile = 0;   // 0 off, 1 every stage + the scanned-code counter, 2 the scan stage alone (gamma_hip_profile_enable)
    std::vector<StageEvent> events;
    std::vector<hipEvent_t> event_pool;
    double stage_ms[GAMMA_HIP_NUM_STAGES] = {0};
    int64_t stage_n[GAMMA_HIP_NUM_STAGES] = {0};
    int64_t scan_pairs = 0;
};

// a writer call: writer_mu (one writer at a time) + mu; exclusive() before memory that a search in flight may be
// reading is freed or moved: search_mu as well (no search can start), both streams drained
struct WriteLock {
    gamma_hip_index* h;
    std::unique_lock<std::mutex> w, s, m;
    explicit WriteLock(gamma_hip_index* h_) : h(h_), w(h_->writer_mu), s(h_->search_mu, std::defer_lock), m(h_->mu) {
        h->wl = this;
        h->write_gen++;   // (under mu: searches read it under mu)
    }
    ~WriteLock() { h->wl = nullptr; }
    hipError_t exclusive() {
        if (!s.owns_lock()) {
            m.unlock();
            s.lock();
            m.lock();
        }
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->wstream);
        if (e == hipSuccess && h->side) e = hipStreamSynchronize(h->side);
        if (e == hipSuccess && h->side2) e = hipStreamSynchronize(h->side2);   // a deferred tie replay reads lists and rows
        return e;
    }
    // searches may start again (the writer keeps writer_mu + mu): after a step that only needed the device idle
    void shared() {
        if (s.owns_lock()) s.unlock();
    }
};

namespace ghi {

using H = gamma_hip_index;

// a search-type call: search_mu for the whole call, mu while it reads the handle and enqueues
struct SearchLock {
    std::unique_lock<std::mutex> s, m;
    explicit SearchLock(H* h) : s(h->search_mu), m(h->mu) {}
    void enqueued() {   // everything is on the stream: writers may go on while the call waits for the GPU
        if (m.owns_lock()) m.unlock();
    }
};

#define GH_CHECK(h, expr)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return e_ == hipErrorOutOfMemory ? GAMMA_HIP_ENOMEM : GAMMA_HIP_EDEVICE;       \
        }                                                                                  \
        if (const char* r_ = gh::launch_refused_take()) {   /* kernels.h: a launcher was handed a shape its callers rule out */ \
            (h)->err = std::string("internal: ") + r_;                                     \
            return GAMMA_HIP_EDEVICE;                                                      \
        }                                                                                  \
    } while (0)

#define GH_TRY(expr)                   \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != GAMMA_HIP_OK) return rc_; \
    } while (0)

inline int fail(H* h, int code, const char* msg) {
    h->err = msg;
    return code;
}

struct StageScope {
    H* h;
    int stage;
    bool count;   // false: add the time to the stage but do not count a new invocation
    hipEvent_t a = nullptr, b = nullptr;
    StageScope(H* h_, int st, bool count_ = true) : h(h_), stage(st), count(count_) {
        if (h->profile == 1 || (h->profile == 2 && st == GAMMA_HIP_STAGE_SCAN && count_)) {
            // events are recycled: creating / destroying two per stage and step costs the host
            // more than the stages' launches
            auto take = [&]() -> hipEvent_t {
                if (!h->event_pool.empty()) {
                    hipEvent_t e = h->event_pool.back();
                    h->event_pool.pop_back();
                    return e;
                }
                hipEvent_t e = nullptr;
                return hipEventCreate(&e) == hipSuccess ? e : nullptr;
            };
            a = take();
            b = take();
            if (!a || !b) {
                a = b = nullptr;
                return;
            }
            (void)hipEventRecord(a, h->stream);
        }
    }
    ~StageScope() {
        if (a && b) {
            (void)hipEventRecord(b, h->stream);
            h->events.push_back({stage, a, b, count});
        }
    }
};

inline int drain_events(H* h) {
    if (h->events.empty()) return GAMMA_HIP_OK;
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    for (auto& e : h->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            h->stage_ms[e.stage] += ms;
            h->stage_n[e.stage] += e.count ? 1 : 0;
        }
        h->event_pool.push_back(e.a);
        h->event_pool.push_back(e.b);
    }
    h->events.clear();
    return GAMMA_HIP_OK;
}

inline double extend_coefficient(uint8_t t) { return 1.1 + kPI / 2 - atan((double)t); }

// A new version of the lists' (offset, length) tables: what the host mirror holds, copied through the
// version's pinned staging on the writer stream -- behind the data copies of the writer that calls this, so a
// search that uses the version finds the entries in place (publish after write, realtime_mem_data.cc:299-300).
// The slot that is overwritten was current NVER - 1 versions ago; the last search that read it is awaited first.
inline int publish_meta(H* h) {
    const int v = (h->cur_ver + 1) % H::NVER;
    if (h->rd_set[v]) GH_CHECK(h, hipStreamWaitEvent(h->wstream, h->rd_ev[v], 0));
    // the staging itself: free once the copies of the version's previous use are done (the writer stream is
    // drained at the end of every writer call, so they are)
    int64_t* po = reinterpret_cast<int64_t*>(h->pin_ver[v]);
    int* pl = reinterpret_cast<int*>(po + h->nlist);
    memcpy(po, h->h_list_off.data(), (size_t)h->nlist * sizeof(int64_t));
    memcpy(pl, h->h_list_len.data(), (size_t)h->nlist * sizeof(int));
    GH_CHECK(h, hipMemcpyAsync(h->d_ver_off[v], po, (size_t)h->nlist * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipMemcpyAsync(h->d_ver_len[v], pl, (size_t)h->nlist * sizeof(int), hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipEventRecord(h->ver_ev[v], h->wstream));
    h->cur_ver = v;
    h->d_list_off = h->d_ver_off[v];
    h->d_list_len = h->d_ver_len[v];
    int mx = 0;
    for (int l = 0; l < h->nlist; l++) mx = std::max(mx, h->h_list_len[l]);
    h->max_list_len = mx;
    return GAMMA_HIP_OK;
}

inline size_t field_elem_size(int dtype) {
    return dtype == GAMMA_HIP_FIELD_INT || dtype == GAMMA_HIP_FIELD_FLOAT ? 4 : 8;
}

}  // namespace ghi
