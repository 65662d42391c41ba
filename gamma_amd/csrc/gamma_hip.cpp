// gamma_hip.cpp -- host side of libgamma_hip.so: the handle, HBM-resident state
// (coarse centroids, PQ codebooks, precomputed table, realtime inverted-list arena, raw
// vectors, delete bitmap), workspace management and the search pipelines, behind the
// C ABI declared in include/gamma_hip.h.  No CPU fallback: every entry point either runs
// the HIP kernels of kernels.hip or returns an error.
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
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"

namespace {

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

}  // namespace

struct WriteLock;

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

    // numeric scalar columns (on-device range filters)
    struct Column {
        int dtype = 0;
        uint8_t* d = nullptr;
        int64_t n = 0, cap = 0;
    };
    std::map<int, Column> fields;
    // STRING columns as dictionary-encoded item lists (on-device term filters): doc i = tok[off[i] .. off[i + 1])
    struct TermColumn {
        int64_t* d_off = nullptr;
        int32_t* d_tok = nullptr;
        int64_t ndocs = 0, cap_docs = 0, ntok = 0, cap_tok = 0;
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

    // inverted-list arena
    uint8_t* d_codes = nullptr;
    int64_t* d_ids = nullptr;
    int64_t arena_cap = 0, arena_used = 0, arena_waste = 0;   // entries; waste = abandoned extents inside used
    int64_t repack_min_entries = 1 << 16;                     // no repack for less waste than this
    int64_t n_repacks = 0;
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
            w_cand_pos, w_cand_ids, w_exact, w_selv, w_selp, w_x, w_outd, w_outl, w_stage, w_filter,
            w_m_dis, w_m_ids, w_part_v, w_part_i, w_assign, w_codes_tmp, w_qperm, w_scnt, w_sflag, w_surv, w_pair_base,
            w_pair_ip, w_flat_cand, w_flat_meta, w_full_cdis, w_full_probe, w_ftab, w_qfil, w_tieflag, w_tcut, w_tlist, w_survc, w_lm_units, w_lm_cnt,
            we_mat, we_cdis, we_x, we_assign, we_codes, we_stage;   // writer side (encode, bitmap_set): never shared with a search
    unsigned long long* d_scan_codes = nullptr;
    size_t dist_budget_bytes = (size_t)8 << 30;   // per-chunk ADC distance buffer (288 GB of HBM per GPU)

    // device copy of the filter table of the running call (entry 0 = the call's own descriptor) and the
    // host image of what entry 0 holds, so an unchanged descriptor is not uploaded again
    gh::FilterDesc ftab_shadow;
    bool ftab_valid = false;

    bool exact_ties = false;   // gamma_hip_set_exact_ties
    bool list_major = false;   // gamma_hip_set_list_major
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
        bool done = false;
        std::condition_variable cv;   // woken when done
    };
    std::mutex comb_mu;
    std::condition_variable comb_wcv;   // the worker waits here
    std::deque<Waiter*> comb_q;
    bool comb_busy = false;             // a batch (or a direct call) is in flight
    bool comb_stop = false;
    std::thread comb_thread;
    bool combine = getenv("GAMMA_HIP_NO_COMBINE") == nullptr;
    // pinned staging of the combined batches, two sets (only the worker touches them)
    void* comb_pin[2] = {nullptr, nullptr};
    size_t comb_pin_bytes[2] = {0, 0};
    // pinned staging of small direct calls (host_search; the search lock serialises its users)
    void* dir_pin = nullptr;
    size_t dir_pin_bytes = 0;

    // profiling
    bool profile = false;
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
    explicit WriteLock(gamma_hip_index* h_) : h(h_), w(h_->writer_mu), s(h_->search_mu, std::defer_lock), m(h_->mu) { h->wl = this; }
    ~WriteLock() { h->wl = nullptr; }
    hipError_t exclusive() {
        if (!s.owns_lock()) {
            m.unlock();
            s.lock();
            m.lock();
        }
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->wstream);
        return e;
    }
};

namespace {

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
    } while (0)

#define GH_TRY(expr)                   \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != GAMMA_HIP_OK) return rc_; \
    } while (0)

int fail(H* h, int code, const char* msg) {
    h->err = msg;
    return code;
}

struct StageScope {
    H* h;
    int stage;
    bool count;   // false: add the time to the stage but do not count a new invocation
    hipEvent_t a = nullptr, b = nullptr;
    StageScope(H* h_, int st, bool count_ = true) : h(h_), stage(st), count(count_) {
        if (h->profile) {
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

int drain_events(H* h) {
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

double extend_coefficient(uint8_t t) { return 1.1 + kPI / 2 - atan((double)t); }

// A new version of the lists' (offset, length) tables: what the host mirror holds, copied through the
// version's pinned staging on the writer stream -- behind the data copies of the writer that calls this, so a
// search that uses the version finds the entries in place (publish after write, realtime_mem_data.cc:299-300).
// The slot that is overwritten was current NVER - 1 versions ago; the last search that read it is awaited first.
int publish_meta(H* h) {
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

// ---- arena ---------------------------------------------------------------------------
int arena_reserve(H* h, int64_t need_entries) {
    if (h->arena_used + need_entries <= h->arena_cap) return GAMMA_HIP_OK;
    int64_t ncap = std::max<int64_t>(h->arena_cap * 2, h->arena_used + need_entries);
    ncap += ncap / 8;
    uint8_t* nc = nullptr;
    int64_t* ni = nullptr;
    if (h->wl) GH_CHECK(h, h->wl->exclusive());   // the old arrays are freed below: no search may be reading them
    GH_CHECK(h, hipMalloc((void**)&nc, (size_t)ncap * h->code_size));
    GH_CHECK(h, hipMalloc((void**)&ni, (size_t)ncap * sizeof(int64_t)));
    if (h->arena_used > 0) {
        GH_CHECK(h, hipMemcpyAsync(nc, h->d_codes, (size_t)h->arena_used * h->code_size,
                                   hipMemcpyDeviceToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(ni, h->d_ids, (size_t)h->arena_used * sizeof(int64_t),
                                   hipMemcpyDeviceToDevice, h->wstream));
    }
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    if (h->d_codes) GH_CHECK(h, hipFree(h->d_codes));
    if (h->d_ids) GH_CHECK(h, hipFree(h->d_ids));
    h->d_codes = nc;
    h->d_ids = ni;
    h->arena_cap = ncap;
    return GAMMA_HIP_OK;
}

// The reference frees a bucket's old memory after a grow / compact swap (delayed by 1 s,
// realtime_mem_data.cc:457-466).  Here grown and compacted extents are abandoned inside the arena
// (arena_waste); once they are more than half of what is in use -- and worth at least a megabyte of codes --
// every list moves into a fresh, tight arena: one kernel, offsets re-published in stream order.
int arena_repack(H* h) {
    int64_t total = 0;
    std::vector<int64_t> noff(h->nlist);
    for (int l = 0; l < h->nlist; l++) {
        noff[l] = total;
        total += h->h_list_cap[l];
    }
    const int64_t ncap = total + total / 8 + 1024;
    uint8_t* nc = nullptr;
    int64_t* ni = nullptr;
    if (h->wl) GH_CHECK(h, h->wl->exclusive());   // every list moves and the old arrays are freed
    GH_TRY(publish_meta(h));                      // the device tables the kernel below reads = the host mirror
    GH_CHECK(h, hipMalloc((void**)&nc, (size_t)ncap * h->code_size));
    if (hipMalloc((void**)&ni, (size_t)ncap * sizeof(int64_t)) != hipSuccess) {
        (void)hipFree(nc);
        return fail(h, GAMMA_HIP_ENOMEM, "arena repack: out of memory");
    }
    GH_CHECK(h, h->we_stage.ensure((size_t)h->nlist * sizeof(int64_t)));
    GH_CHECK(h, hipMemcpyAsync(h->we_stage.p, noff.data(), (size_t)h->nlist * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
    gh::launch_repack_lists(h->wstream, h->d_codes, h->d_ids, nc, ni, h->d_list_off, h->we_stage.as<int64_t>(),
                            h->d_list_len, h->nlist, h->code_size, h->max_list_len);
    GH_CHECK(h, hipStreamSynchronize(h->wstream));   // noff is a local; the old arrays are free to go
    GH_CHECK(h, hipFree(h->d_codes));
    GH_CHECK(h, hipFree(h->d_ids));
    h->d_codes = nc;
    h->d_ids = ni;
    h->arena_cap = ncap;
    h->arena_used = total;
    h->arena_waste = 0;
    h->h_list_off = noff;
    h->n_repacks++;
    return publish_meta(h);
}
int arena_repack_if_need(H* h) {
    const int64_t min_waste = std::max<int64_t>(h->repack_min_entries, 1);
    if (h->arena_waste < min_waste || h->arena_waste * 2 < h->arena_used) return GAMMA_HIP_OK;
    return arena_repack(h);
}

// RealTimeMemData::ExtendBucketIfNeed + RTInvertBucketData::ExtendBucketMem
// (realtime_mem_data.cc:383-421,152-188): same growth law, region moved inside the arena.
int list_ensure(H* h, int l, int add) {
    const int len = h->h_list_len[l], cap = h->h_list_cap[l];
    if ((int64_t)len + add <= cap) return GAMMA_HIP_OK;
    if ((int64_t)cap * 2 >= h->bucket_max) return fail(h, GAMMA_HIP_EFULL, "exceed the max bucket keys");
    const int least = len + add;
    double coef = extend_coefficient(++h->h_extend_time[l]);
    int ext = (int)(cap * coef);
    while (ext < least) {
        coef = extend_coefficient(++h->h_extend_time[l]);
        ext = (int)(ext * coef);
    }
    GH_TRY(arena_reserve(h, ext));
    const int64_t noff = h->arena_used;
    if (len > 0) {
        GH_CHECK(h, hipMemcpyAsync(h->d_codes + noff * h->code_size,
                                   h->d_codes + h->h_list_off[l] * h->code_size,
                                   (size_t)len * h->code_size, hipMemcpyDeviceToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(h->d_ids + noff, h->d_ids + h->h_list_off[l],
                                   (size_t)len * sizeof(int64_t), hipMemcpyDeviceToDevice, h->wstream));
    }
    h->arena_waste += cap;
    h->arena_used += ext;
    h->h_list_off[l] = noff;   // the old extent stays intact: searches in flight read it through their version
    h->h_list_cap[l] = ext;
    return GAMMA_HIP_OK;
}

int add_keys_locked(H* h, int l, int n, const int64_t* vids, const uint8_t* codes) {
    if (l < 0 || l >= h->nlist || n < 0) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
    if (n == 0) return GAMMA_HIP_OK;
    GH_TRY(list_ensure(h, l, n));
    const int64_t pos = h->h_list_off[l] + h->h_list_len[l];
    GH_CHECK(h, hipMemcpyAsync(h->d_ids + pos, vids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice,
                               h->wstream));
    GH_CHECK(h, hipMemcpyAsync(h->d_codes + pos * h->code_size, codes, (size_t)n * h->code_size,
                               hipMemcpyHostToDevice, h->wstream));
    for (int i = 0; i < n; i++) {
        const int64_t v = vids[i];
        if (v < 0) {   // superseded slot restored from a dump (ReadInvertedLists, gamma_index_io.cc:186-189)
            h->h_deleted[l]++;
            h->n_moved++;
            continue;
        }
        if ((size_t)v >= h->vid_pos.size()) h->vid_pos.resize(std::max<size_t>(h->vid_pos.size() * 2, v + 1), -1);
        h->vid_pos[v] = ((int64_t)l << 32) | (int64_t)(h->h_list_len[l] + i);
        if (h->doc_deleted(v)) h->h_deleted[l]++;  // realtime_mem_data.cc:293-296
    }
    h->h_list_len[l] += n;  // publish after the copies (realtime_mem_data.cc:299-300)
    h->ntotal += n;
    GH_TRY(publish_meta(h));
    // the host buffers may be reused by the caller as soon as we return
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return arena_repack_if_need(h);
}

// off != nullptr: the range bitmaps go to w_filter at *off (advanced; the caller has sized w_filter for all
// the requests of a combined batch); nullptr: a call of its own, bitmaps from offset 0
int build_filter(H* h, const gamma_hip_search_params* p, gh::FilterDesc* f, size_t* off_io = nullptr) {
    memset(f, 0, sizeof(*f));
    f->del_bitmap = h->d_bitmap;
    f->del_bits = h->d_bitmap ? h->bitmap_bits : 0;
    f->vid2doc = h->h_v2d.empty() ? nullptr : h->d_v2d;
    f->n_vid2doc = (int64_t)h->h_v2d.size();
    f->has_range = p->has_range ? 1 : 0;
    f->n_range = p->has_range ? p->n_range : 0;
    if (f->n_range > gh::kMaxRange) return fail(h, GAMMA_HIP_EINVAL, "too many range filters");
    if (f->n_range > 0) {
        size_t tot = 0;
        for (int i = 0; i < f->n_range; i++) tot += ((size_t)p->range[i].bitmap_bytes + 15) & ~(size_t)15;
        if (!off_io) GH_CHECK(h, h->w_filter.ensure(tot));
        size_t off = off_io ? *off_io : 0;
        for (int i = 0; i < f->n_range; i++) {
            const gamma_hip_range_filter& r = p->range[i];
            uint8_t* dst = h->w_filter.as<uint8_t>() + off;
            GH_CHECK(h, hipMemcpyAsync(dst, r.bitmap, (size_t)r.bitmap_bytes, hipMemcpyHostToDevice,
                                       h->stream));
            f->range[i].bitmap = dst;
            f->range[i].min_doc = r.min_doc;
            f->range[i].max_doc = r.max_doc;
            f->range[i].min_aligned = r.min_aligned;
            f->range[i].b_not_in = r.b_not_in;
            off += ((size_t)r.bitmap_bytes + 15) & ~(size_t)15;
        }
        if (off_io) *off_io = off;
    }
    f->n_field = p->n_field;
    if (p->n_field < 0 || p->n_field > gh::kMaxField || (p->n_field > 0 && !p->field))
        return fail(h, GAMMA_HIP_EINVAL, "bad field filters");
    for (int i = 0; i < p->n_field; i++) {
        const gamma_hip_field_filter& ff = p->field[i];
        auto it = h->fields.find(ff.field_id);
        if (it == h->fields.end()) return fail(h, GAMMA_HIP_EINVAL, "field filter on an unknown column");
        gh::FieldDesc& fd = f->field[i];
        fd.col = it->second.d;
        fd.n = it->second.n;
        fd.dtype = it->second.dtype;
        fd.incl = (ff.include_lower ? 1 : 0) | (ff.include_upper ? 2 : 0);
        fd.lo_i = ff.lower_i;
        fd.hi_i = ff.upper_i;
        fd.lo_f = ff.lower_f;
        fd.hi_f = ff.upper_f;
    }
    f->n_term = p->n_term;
    if (p->n_term < 0 || p->n_term > gh::kMaxTerm || (p->n_term > 0 && !p->term))
        return fail(h, GAMMA_HIP_EINVAL, "bad term filters");
    for (int i = 0; i < p->n_term; i++) {
        const gamma_hip_term_filter& tf = p->term[i];
        auto it = h->terms.find(tf.field_id);
        if (it == h->terms.end()) return fail(h, GAMMA_HIP_EINVAL, "term filter on an unknown column");
        if (tf.n_items < 0 || tf.n_items > gh::kMaxTermItems || tf.op < 0 || tf.op > 2)
            return fail(h, GAMMA_HIP_EINVAL, "bad term filter");
        gh::TermDesc& td = f->term[i];
        td.off = it->second.d_off;
        td.tok = it->second.d_tok;
        td.n = it->second.ndocs;
        td.op = tf.op;
        td.n_items = tf.n_items;
        for (int k = 0; k < tf.n_items; k++) td.items[k] = tf.items[k];
    }
    return GAMMA_HIP_OK;
}

// What the scan needs to know about the validity predicates of a call: the device filter table, the
// optional query -> entry map (combined batches of requests with their own filters), and whether
// anything but the delete bitmap can reject an entry.
struct FiltCtx {
    const gh::FilterDesc* d_tab = nullptr;
    const int* d_qf = nullptr;
    bool any_clause = false;
    FiltCtx at(int q0) const {   // the same context for the queries from q0 on
        FiltCtx c = *this;
        if (c.d_qf) c.d_qf += q0;
        return c;
    }
};

int filt_ctx_single(H* h, const gh::FilterDesc& f, FiltCtx* c) {
    GH_CHECK(h, h->w_ftab.ensure(sizeof(gh::FilterDesc)));
    if (!h->ftab_valid || memcmp(&h->ftab_shadow, &f, sizeof(f)) != 0) {
        // the stream may still be reading the previous image: the copy is ordered behind it
        h->ftab_shadow = f;
        h->ftab_valid = true;
        GH_CHECK(h, hipMemcpyAsync(h->w_ftab.p, &h->ftab_shadow, sizeof(f), hipMemcpyHostToDevice, h->stream));
    }
    c->d_tab = h->w_ftab.as<gh::FilterDesc>();
    c->d_qf = nullptr;
    c->any_clause = f.has_range || f.n_field > 0 || f.n_term > 0;
    return GAMMA_HIP_OK;
}

// Spatial order of the coarse centroids by recursive principal-axis bisection: split the set at
// the median of its projection on the dominant direction (a few power iterations), recurse.
// Neighbouring ranks = neighbouring centroids.  Used only to order queries for cache locality.
void centroid_order_rec(const float* cc, int d, std::vector<int>& idx, int lo, int hi, std::vector<float>& proj,
                        std::vector<double>& mean, std::vector<double>& dir, std::vector<double>& tmp) {
    const int n = hi - lo;
    if (n <= 2) return;
    for (int t = 0; t < d; t++) mean[t] = 0;
    for (int i = lo; i < hi; i++) {
        const float* c = cc + (size_t)idx[i] * d;
        for (int t = 0; t < d; t++) mean[t] += c[t];
    }
    for (int t = 0; t < d; t++) mean[t] /= n;
    // start from the direction to the point farthest from the mean (never orthogonal to the data)
    double best = -1;
    int far = lo;
    for (int i = lo; i < hi; i++) {
        const float* c = cc + (size_t)idx[i] * d;
        double s = 0;
        for (int t = 0; t < d; t++) s += (c[t] - mean[t]) * (c[t] - mean[t]);
        if (s > best) { best = s; far = i; }
    }
    for (int t = 0; t < d; t++) dir[t] = cc[(size_t)idx[far] * d + t] - mean[t];
    for (int it = 0; it < 6; it++) {
        for (int t = 0; t < d; t++) tmp[t] = 0;
        for (int i = lo; i < hi; i++) {
            const float* c = cc + (size_t)idx[i] * d;
            double pr = 0;
            for (int t = 0; t < d; t++) pr += (c[t] - mean[t]) * dir[t];
            for (int t = 0; t < d; t++) tmp[t] += pr * (c[t] - mean[t]);
        }
        double nrm = 0;
        for (int t = 0; t < d; t++) nrm += tmp[t] * tmp[t];
        if (nrm <= 0) break;
        nrm = std::sqrt(nrm);
        for (int t = 0; t < d; t++) dir[t] = tmp[t] / nrm;
    }
    for (int i = lo; i < hi; i++) {
        const float* c = cc + (size_t)idx[i] * d;
        double pr = 0;
        for (int t = 0; t < d; t++) pr += (c[t] - mean[t]) * dir[t];
        proj[idx[i]] = (float)pr;
    }
    const int mid = lo + n / 2;
    std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                     [&](int a, int b) { return proj[a] < proj[b] || (proj[a] == proj[b] && a < b); });
    centroid_order_rec(cc, d, idx, lo, mid, proj, mean, dir, tmp);
    centroid_order_rec(cc, d, idx, mid, hi, proj, mean, dir, tmp);
}

std::vector<int> centroid_rank(const float* cc, int nlist, int d) {
    std::vector<int> idx(nlist), rank(nlist);
    for (int i = 0; i < nlist; i++) idx[i] = i;
    std::vector<float> proj(nlist);
    std::vector<double> mean(d), dir(d), tmp(d);
    centroid_order_rec(cc, d, idx, 0, nlist, proj, mean, dir, tmp);
    for (int i = 0; i < nlist; i++) rank[idx[i]] = i;
    return rank;
}

int check_params(H* h, const gamma_hip_search_params* p, int nq, int k) {
    if (!p) return fail(h, GAMMA_HIP_EINVAL, "null params");
    if (nq < 0) return fail(h, GAMMA_HIP_EINVAL, "nq < 0");
    if (p->metric != GAMMA_HIP_METRIC_IP && p->metric != GAMMA_HIP_METRIC_L2)
        return fail(h, GAMMA_HIP_EINVAL, "bad metric");
    if (k > 4096) return fail(h, GAMMA_HIP_EINVAL, "k > 4096 unsupported");
    return GAMMA_HIP_OK;
}

// ---- IVFPQ stage A: coarse + tables + scan + top-R + ids ------------------------------
// results: w_cand_dis [nq*R] (ADC distance, best first, sentinel pad), w_cand_ids [nq*R]
int ivfpq_coarse(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, float* out_dis = nullptr,
                 int* out_probe = nullptr) {
    const int P = p->nprobe, d = h->d, nlist = h->nlist;
    hipStream_t s = h->stream;
    int mode = p->coarse_mode;
    if (mode < 0) mode = nq < 20 ? 0 : 1;  // faiss:utils/distances.cpp:303,346
    // large batches: no distance matrix (coarse.hip); exact ties replay rows of the matrix, so they keep it
    const bool fused = mode == 1 && h->coarse_fused && !h->exact_ties && gh::coarse_fused_supported(nq, d, nlist, P);
    gh::CoarseFusedPlan plan;
    if (fused) {
        plan = gh::coarse_fused_plan(nq, nlist, P, h->coarse_cap);
        GH_CHECK(h, h->w_mat.ensure(plan.bytes));
    } else {
        GH_CHECK(h, h->w_mat.ensure((size_t)nq * nlist * sizeof(float)));
    }
    if (!out_dis || !out_probe) {   // the workspace the scan reads
        GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nq * P * sizeof(float)));
        GH_CHECK(h, h->w_probe.ensure((size_t)nq * P * sizeof(int)));
        out_dis = h->w_coarse_dis.as<float>();
        out_probe = h->w_probe.as<int>();
    }
    StageScope t(h, GAMMA_HIP_STAGE_COARSE);
    if (fused) {
        gh::launch_coarse_fused(s, plan, h->w_mat.p, d_x, nq, d, h->d_cc, nlist, h->d_cc_norms, P, out_dis, out_probe);
        static const bool dbg = getenv("GAMMA_HIP_COARSE_DBG") != nullptr;
        if (dbg) {   // how many queries the strip lists could not hold (they went through the repair kernel)
            int n_ovf = 0;
            GH_CHECK(h, hipStreamSynchronize(s));
            GH_CHECK(h, hipMemcpy(&n_ovf, static_cast<char*>(h->w_mat.p) + plan.off_ovf, sizeof(int), hipMemcpyDeviceToHost));
            fprintf(stderr, "coarse fused: nlist %d nprobe %d sample %d strips %d: %d of %d queries repaired\n", nlist, P,
                    plan.sample, plan.nseg, n_ovf, nq);
        }
        return GAMMA_HIP_OK;
    }
    if (mode == 0) {
        gh::launch_pairwise(s, true, d_x, nq, d, h->d_cc, nlist, h->w_mat.as<float>(), nlist);
    } else {
        // query norms: fused into the MFMA kernel (xn = nullptr) when a tile holds whole rows (d <= 128); longer rows
        // get them from their own pass -- inside the K-slab loop they cost a fifth of the kernel (d = 768: 3.46 -> 2.72 ms
        // per 8192 x 16384 with the conflict-free staging)
        const float* xn = nullptr;
        if (d > 128) {
            GH_CHECK(h, h->w_xn.ensure((size_t)nq * sizeof(float)));
            gh::launch_row_norms(s, d_x, nq, d, h->w_xn.as<float>());
            xn = h->w_xn.as<float>();
        }
        gh::launch_l2_gemmform(s, d_x, nq, d, h->d_cc, nlist, xn, h->d_cc_norms,
                               h->w_mat.as<float>(), nlist, true);
    }
    if (h->exact_ties) GH_CHECK(h, h->w_tieflag.ensure((size_t)nq));
    gh::launch_coarse_select(s, h->w_mat.as<float>(), nlist, nq, P, out_dis, out_probe,
                             h->exact_ties ? h->w_tieflag.as<uint8_t>() : nullptr, h->d_tie_stats);
    return GAMMA_HIP_OK;
}

// pre_dis / pre_probe: coarse assignment computed elsewhere (sharded search: the rank owning the
// query slice), device pointers [nq*nprobe]; nullptr = run the coarse quantizer here
int ivfpq_stage_a(H* h, const gamma_hip_search_params* p, const FiltCtx& fc, int nq,
                  const float* d_x, int R, const float* pre_dis = nullptr, const int* pre_probe = nullptr,
                  bool shard = false, float* out_dis = nullptr, int64_t* out_ids = nullptr) {
    const int P = p->nprobe, d = h->d, M = h->M, nlist = h->nlist;
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    hipStream_t s = h->stream;
    // this call reads the lists through the version of their (offset, length) tables that is current now:
    // behind the writer's copies (ver_ev), and the version is not reused before the kernels below are done (rd_ev)
    const int ver = h->cur_ver;
    GH_CHECK(h, hipStreamWaitEvent(s, h->ver_ev[ver], 0));
    GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nq * P * sizeof(float)));
    GH_CHECK(h, h->w_probe.ensure((size_t)nq * P * sizeof(int)));
    GH_CHECK(h, h->w_pair_off.ensure((size_t)nq * (P + 1) * sizeof(int)));
    GH_CHECK(h, h->w_pair_base.ensure((size_t)nq * P * sizeof(int64_t)));
    GH_CHECK(h, h->w_qtotal.ensure((size_t)nq * sizeof(int)));
    GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq * R * sizeof(int)));
    // the top-R table goes to the workspace (stage B reads it there) or straight into the caller's
    // buffers (sharded search: 12 B x R per query would otherwise be copied once more)
    if (!out_dis) {
        GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq * R * sizeof(float)));
        out_dis = h->w_cand_dis.as<float>();
    }
    if (!out_ids) {
        GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq * R * sizeof(int64_t)));
        out_ids = h->w_cand_ids.as<int64_t>();
    }
    if (pre_dis && pre_probe) {
        if (shard) {
            // dense probe groups for the owned lists (kernels.hip, k_compact_probes)
            gh::launch_compact_probes(s, pre_probe, pre_dis, nq, P, h->d_list_len, h->d_list_mask, nlist,
                                      h->w_probe.as<int>(), h->w_coarse_dis.as<float>());
        } else {
            GH_CHECK(h, hipMemcpyAsync(h->w_coarse_dis.p, pre_dis, (size_t)nq * P * sizeof(float),
                                       hipMemcpyDeviceToDevice, s));
            GH_CHECK(h, hipMemcpyAsync(h->w_probe.p, pre_probe, (size_t)nq * P * sizeof(int),
                                       hipMemcpyDeviceToDevice, s));
        }
    } else {
        GH_TRY(ivfpq_coarse(h, p, nq, d_x));
    }
    h->scan_pairs += (int64_t)nq * P;
    // ids are only read during the scan when something can reject an entry: a delete bit,
    // a range filter, or a superseded (bit 63) slot left behind by Update
    const int need_ids =
            (fc.any_clause || (h->d_bitmap && h->bitmap_any) || h->n_moved > 0) ? 1 : 0;
    const int* qperm = nullptr;
    // Probes per workgroup.  Sharded search with a compacted assignment: a query keeps ~P/W probes on
    // this shard, all in its first group(s) -- the other P/G - 1 workgroups of the query would start only
    // to find nothing to do (at W = 8 that was half of the scan time).  When the expected candidate
    // count per query is small, ONE workgroup takes all of a query's probes (G = P): it bounds the
    // R-th best itself (producer path of the pre-filter, no consumers), and computes the query's PQ
    // table on the fly instead of reading it back from HBM (IPF, kernels.hip).
    // 8 probes per workgroup pay off with short lists (half the query-table re-reads, a tighter bound from a
    // first group of 8 lists); long lists or many probes balance better with 4 (tools/shape_sweep.py)
    int G0 = ((double)h->ntotal / std::max(1, nlist) <= 700.0 && P <= 64) ? 8 : 4;
    int64_t t2_bytes = (int64_t)nlist * M * 256 * sizeof(float);
    const bool compacted = shard && pre_dis && pre_probe;
    if (compacted && h->scan_bound && R <= 256) {
        int64_t owned = 0;
        for (int l = 0; l < nlist; l++)
            owned += h->h_list_len[l] > 0 && (h->h_list_mask.empty() || h->h_list_mask[l]);
        t2_bytes = owned * M * 256 * (int64_t)sizeof(float);
        const double exp_probes = (double)P * (double)owned / std::max(1, nlist);
        const double exp_cand = exp_probes * (owned ? (double)h->ntotal / (double)owned : 0.0);
        if (exp_cand <= 16384.0) {
            G0 = 1;
            while (G0 < P) G0 <<= 1;
        }
    }
    const int G = gh::scan_group_size(nq, P, G0), PGN = (P + G - 1) / G;
    // exact ties (ties.hip): queries whose top-R cut goes through a group of equal ADC distances are marked here
    // and redone by the replay at the end of stage B
    h->tie = H::TieCtx();
    h->tie.on = h->exact_ties && !shard && R <= gh::tie_replay_max_k() && P <= gh::tie_replay_max_probes();
    if (h->tie.on) {
        GH_CHECK(h, h->w_tcut.ensure((size_t)nq));
        GH_CHECK(h, h->w_tlist.ensure(((size_t)nq + 1) * sizeof(int)));   // count | list[nq]
        GH_CHECK(h, hipMemsetAsync(h->w_tcut.p, 0, (size_t)nq, s));
        GH_CHECK(h, hipMemsetAsync(h->w_tlist.p, 0, sizeof(int), s));
    }
    // Threshold pre-filter: scan the nearest probe group first, bound each query's R-th best
    // distance from it, and let the scan of the remaining groups keep a short survivor list per
    // query; the exact top-R then comes from a few hundred survivors instead of ~10^4 candidates
    // (select.hip).  Queries without a usable bound fall back to the unfiltered selection.
    // (sharded without a supplied assignment: probe groups are sparse, nothing to bound from)
    // small batches: one probe per workgroup, a single list rarely holds R candidates, and the
    // unfiltered selection is latency-bound anyway
    // (one group per query -- few probes, or a shard -- is fine: the producer bounds and compacts its own candidates)
    const bool bounded = (!shard || compacted) && h->scan_bound && R <= 256 && P <= 128 && G >= 4;
    const bool fuse_ip = bounded && PGN == 1 && (M == 16 || M == 32) && !getenv("GAMMA_HIP_NO_FUSED_IP");
    {
        StageScope t(h, GAMMA_HIP_STAGE_TABLES);
        if (!fuse_ip) {
            GH_CHECK(h, h->w_st2.ensure((size_t)nq * M * 256 * sizeof(float)));
            gh::launch_pq_ip_table(s, d_x, nq, d, M, h->d_pqc, h->w_st2.as<float>());
        }
        gh::launch_pair_offsets(s, h->w_probe.as<int>(), nq, P, h->d_list_len, h->d_list_mask, nlist,
                                h->w_pair_off.as<int>(), h->w_qtotal.as<int>(),
                                h->profile ? h->d_scan_codes : nullptr, h->d_list_off,
                                h->w_pair_base.as<int64_t>());
        // enough queries that L2 capacity matters: run them in spatial order (kernels.hip)
        // (not when the T2 rows of the lists scanned here fit the L2s anyway: a shard of a small index)
        if (h->sort_queries && h->d_list_rank && nq >= 256 && t2_bytes > ((int64_t)8 << 20)) {
            GH_CHECK(h, h->w_qperm.ensure(((size_t)2 * nq + gh::query_order_bins()) * sizeof(int)));   // qperm | qkey | bins
            gh::launch_query_order(s, h->w_probe.as<int>(), nq, P, h->d_list_rank, nlist,
                                   h->w_qperm.as<int>() + nq, h->w_qperm.as<int>(), h->w_qperm.as<int>() + 2 * (size_t)nq);
            qperm = h->w_qperm.as<int>();
        }
        h->last_qperm = qperm;   // stage B runs the re-rank in the same order
    }
    // per-query slab of the distance buffer; multiple of 4 floats so rows are 16-byte aligned
    const int64_t q_stride = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
    GH_CHECK(h, h->w_dist.ensure((size_t)nq * q_stride * sizeof(float)));
    // dis0 of every (query, probe) pair: the coarse distance (L2) or <x_q, centroid> (inner product)
    const float* dis0 = h->w_coarse_dis.as<float>();
    if (!l2) {
        StageScope t(h, GAMMA_HIP_STAGE_TABLES, false);
        GH_CHECK(h, h->w_pair_ip.ensure((size_t)nq * P * sizeof(float)));
        gh::launch_pair_ip(s, d_x, h->d_cc, h->w_probe.as<int>(), nq, P, d, nlist, h->w_pair_ip.as<float>());
        dis0 = h->w_pair_ip.as<float>();
    }
    auto scan = [&](int gsz, int pg_lo, int pg_cnt, const gh::ScanBound* bound, bool count) {
        StageScope t(h, GAMMA_HIP_STAGE_SCAN, count);
        gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, d, M, P, h->w_probe.as<int>(),
                                   dis0, h->d_cc, h->w_st2.as<float>(), h->d_T2,
                                   h->d_list_off, h->d_list_len, h->d_list_mask, nlist, h->d_codes,
                                   h->d_ids, h->w_pair_off.as<int>(), q_stride, h->w_dist.as<float>(),
                                   fc.d_tab, fc.d_qf, need_ids, qperm, gsz, pg_lo, pg_cnt, shard ? 1 : 0, bound,
                                   fuse_ip ? h->d_pqc : nullptr);
    };
    if (!bounded) {
        scan(G, 0, PGN, nullptr, true);
        StageScope t(h, GAMMA_HIP_STAGE_SELECT);
        gh::launch_select_topk(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), 0,
                               (int)std::min<int64_t>(q_stride, 1 << 30), nq, R,
                               out_dis, h->w_cand_pos.as<int>());
        gh::launch_map_candidates(s, h->w_cand_pos.as<int>(), nq, R, P, h->w_probe.as<int>(),
                                  h->w_pair_off.as<int>(), h->d_list_off, h->d_ids,
                                  out_ids);
        if (h->tie.on)
            gh::launch_flag_cut_ties(s, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nq, R, out_dis,
                                     h->w_cand_pos.as<int>(), nullptr, h->w_tcut.as<uint8_t>());
    } else {
        // List-major consumer scan (scan_lm.hip) for large batches: the producers run alone, then every other probe
        // is scored two queries per list pass.  Gated on what k_scan_lm covers; results are the same either way.
        // Measured at C3 (16384 queries): 233 k units for 393 k consumer pairs, 58 % of the query-major kernel's
        // vector instructions and 84 % of its LDS cycles -- but 1069 us against ~700 us for the same pairs: two
        // 16 KB query-table rows per unit instead of one per eight pairs (11 GB through the L2 per launch), four
        // workgroups per CU behind a 32 KB LUT2, and ds_read_b64 gathers that conflict more than ds_read_b32.
        // Lists of a few hundred codes are too short to pay for it; it stays OFF unless asked for
        // (gamma_hip_set_list_major, GAMMA_HIP_LM=1), kept for long-list shapes and covered by a parity test.
        static const bool env_lm = getenv("GAMMA_HIP_LM") != nullptr;
        const bool lm = (env_lm || h->list_major) && !shard && PGN > 1 && M == 16 && nq >= 2048 && 1 + (P - G) <= 64 &&
                        !fc.d_qf && h->d_list_mask == nullptr;
        const int cap = gh::scan_slice_cap();
        // one survivor slice per probe group (slice 0: the producer's own) -- or, list-major, per consumer PAIR
        const int nsl = lm ? 1 + (P - G) : PGN;
        // rq | ready[nq] | gcnt[nq][nsl]   (rq: count + list of the queries that need the repair launch, 8-byte aligned)
        const size_t rq_bytes = (((size_t)nq + 1) * sizeof(int) + 7) & ~(size_t)7;
        GH_CHECK(h, h->w_scnt.ensure(rq_bytes + (size_t)nq * (sizeof(unsigned long long) + (size_t)nsl * sizeof(int))));
        GH_CHECK(h, h->w_sflag.ensure((size_t)nq));
        GH_CHECK(h, h->w_surv.ensure((size_t)nq * (lm ? 1 : nsl) * cap * sizeof(unsigned long long)));
        unsigned long long* ready = reinterpret_cast<unsigned long long*>(h->w_scnt.as<char>() + rq_bytes);
        GH_CHECK(h, hipMemsetAsync(h->w_scnt.p, 0, sizeof(int), s));
        GH_CHECK(h, hipMemsetAsync(ready, 0, (size_t)nq * sizeof(unsigned long long), s));
        gh::ScanBound sb;
        sb.ready = ready;
        sb.surv = h->w_surv.as<unsigned long long>();
        sb.gcnt = reinterpret_cast<int*>(ready + nq);
        sb.K = R;
        sb.cnt_stride = nsl;
        // exact ties: the replay reads a bounded query's candidates from slab group 0 + the survivor slices, and an
        // unbounded one's from the slab the repair launch below fills -- nothing else needs the consumers' distances
        // (the list-major variant has no repair launch: it stores everything)
        sb.store_all = (h->tie.on && lm) ? 1 : 0;
        sb.rq_count = h->w_scnt.as<int>();
        sb.rq_list = h->w_scnt.as<int>() + 1;
        const unsigned long long* surv_c = nullptr;
        if (!lm) {
            scan(G, 0, PGN, &sb, true);
        } else {
            const int PC = P - G, B = gh::lm_block_queries(P, G), nblk = (nq + B - 1) / B;
            GH_CHECK(h, h->w_survc.ensure((size_t)nq * PC * gh::lm_pair_cap() * sizeof(unsigned long long)));
            GH_CHECK(h, h->w_lm_units.ensure((size_t)nblk * gh::lm_units_per_block() * 16 * sizeof(int)));
            GH_CHECK(h, h->w_lm_cnt.ensure((size_t)nblk * sizeof(int)));
            GH_CHECK(h, hipMemsetAsync(sb.gcnt, 0, (size_t)nq * nsl * sizeof(int), s));   // pairs never scored: 0 survivors
            scan(G, 0, 1, &sb, true);   // producers: first probe group, bound, own survivors (slice 0)
            StageScope t2(h, GAMMA_HIP_STAGE_SCAN, false);
            gh::launch_lm_units(s, h->w_probe.as<int>(), dis0, h->w_pair_off.as<int>(), h->d_list_off, h->d_list_len,
                                h->d_list_mask, nlist, qperm, ready, nq, P, G, B, h->w_lm_units.as<int>(),
                                h->w_lm_cnt.as<int>());
            gh::LmScanArgs la;
            la.units = h->w_lm_units.as<int>();
            la.ucount = h->w_lm_cnt.as<int>();
            la.nq = nq;
            la.B = B;
            la.st2 = h->w_st2.as<float>();
            la.T2 = h->d_T2;
            la.codes = h->d_codes;
            la.ids = h->d_ids;
            la.out = h->w_dist.as<float>();
            la.q_stride = q_stride;
            la.surv = h->w_survc.as<unsigned long long>();
            la.cnt = sb.gcnt;
            la.nslc = PC;
            la.cnt_stride = nsl;
            la.store_all = sb.store_all;
            la.need_ids = need_ids;
            la.ftab = fc.d_tab;
            gh::launch_scan_lm(s, l2, M, la);
            surv_c = h->w_survc.as<unsigned long long>();
            static const bool lm_dbg = getenv("GAMMA_HIP_LM_DBG") != nullptr;
            static int lm_shown = 0;
            if (lm_dbg && lm_shown++ < 2) {
                std::vector<int> uc(nblk);
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(uc.data(), h->w_lm_cnt.p, (size_t)nblk * sizeof(int), hipMemcpyDeviceToHost);
                int64_t tot = 0;
                for (int v : uc) tot += v;
                fprintf(stderr, "list-major scan: %d blocks of %d queries, %lld units for %lld consumer pairs\n", nblk, B,
                        (long long)tot, (long long)nq * PC);
            }
        }
        static const bool dbg = getenv("GAMMA_HIP_BOUND_DBG") != nullptr;
        static int shown = 0;
        StageScope t(h, GAMMA_HIP_STAGE_SELECT);
        gh::launch_select_final(s, l2, sb.surv, sb.gcnt, nsl, cap, sb.ready, h->w_pair_off.as<int>(), P, nq, R,
                                h->w_pair_base.as<int64_t>(), h->d_ids,
                                h->w_sflag.as<uint8_t>(), out_dis,
                                h->w_cand_pos.as<int>(), out_ids, h->tie.on ? h->w_tcut.as<uint8_t>() : nullptr,
                                h->d_tie_stats, sb.rq_list, sb.rq_count, surv_c, gh::lm_pair_cap());
        if (PGN > 1 && !sb.store_all) {
            // queries the slices could not answer: their consumer groups are scored again, distances stored
            StageScope t2(h, GAMMA_HIP_STAGE_SCAN, false);
            gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, d, M, P, h->w_probe.as<int>(), dis0, h->d_cc,
                                       h->w_st2.as<float>(), h->d_T2, h->d_list_off, h->d_list_len, h->d_list_mask,
                                       nlist, h->d_codes, h->d_ids, h->w_pair_off.as<int>(), q_stride,
                                       h->w_dist.as<float>(), fc.d_tab, fc.d_qf, need_ids, nullptr, G, 1, PGN - 1,
                                       shard ? 1 : 0, nullptr, nullptr, sb.rq_list, sb.rq_count);
        }
        gh::launch_select_topk(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), 0,
                               (int)std::min<int64_t>(q_stride, 1 << 30), nq, R,
                               out_dis, h->w_cand_pos.as<int>(), h->w_sflag.as<uint8_t>());
        if (h->tie.on) {
            gh::launch_flag_cut_ties(s, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nq, R, out_dis,
                                     h->w_cand_pos.as<int>(), h->w_sflag.as<uint8_t>(), h->w_tcut.as<uint8_t>());
            h->tie.bounded = !lm;   // list-major: the replay walks the whole slab (everything is stored with exact ties on)
            h->tie.nsl = nsl;
            h->tie.cap = cap;
        }
        if (dbg && shown++ >= 8 && shown <= 13) {
            std::vector<uint8_t> hf(nq);
            std::vector<int> hc((size_t)nq * nsl);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(hf.data(), h->w_sflag.p, nq, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hc.data(), sb.gcnt, hc.size() * sizeof(int), hipMemcpyDeviceToHost);
            std::vector<unsigned long long> hr(nq);
            (void)hipMemcpy(hr.data(), sb.ready, (size_t)nq * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            int64_t nf = 0, tot = 0, mx = 0, nobound = 0;
            for (int i = 0; i < nq; i++) {
                nf += hf[i];
                nobound += (hr[i] >> 32) != 1ull;
            }
            for (size_t i = 0; i < hc.size(); i++) {
                tot += hc[i];
                mx = std::max<int64_t>(mx, hc[i]);
            }
            fprintf(stderr, "scan bound: %lld of %d queries unfiltered (%lld without a bound), survivors per query mean %.1f, "
                    "per slice max %lld\n",
                    (long long)nf, nq, (long long)nobound, (double)tot / nq, (long long)mx);
        }
        gh::launch_map_candidates(s, h->w_cand_pos.as<int>(), nq, R, P, h->w_probe.as<int>(),
                                  h->w_pair_off.as<int>(), h->d_list_off, h->d_ids,
                                  out_ids, h->w_sflag.as<uint8_t>());
    }
    h->tie.G = G;
    h->tie.q_stride = q_stride;
    GH_CHECK(h, hipEventRecord(h->rd_ev[ver], s));
    h->rd_set[ver] = true;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

// ---- stage B: compute_dis (gamma_index_ivfpq.cc:642-697) ------------------------------
int ivfpq_stage_b(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int R, int k,
                  const float* cand_dis, const int64_t* cand_ids, float* d_distances,
                  int64_t* d_labels, const int* qperm = nullptr, bool tie_replay = false) {
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    hipStream_t s = h->stream;
    StageScope t(h, GAMMA_HIP_STAGE_RERANK);
    // exact ties: the final-stage kernel lists the queries with a tie among their first k+1 distances (or with a
    // tied top-R cut, stage A) and k_tie_replay redoes those the way the reference's heaps do (ties.hip)
    const bool ties = tie_replay && h->tie.on;
    gh::TieFlags tf;
    if (ties) {
        tf.cut = h->w_tcut.as<uint8_t>();
        tf.count = h->w_tlist.as<int>();
        tf.list = h->w_tlist.as<int>() + 1;
        tf.stats = h->d_tie_stats;
    }
    auto replay = [&]() {
        gh::TieReplayArgs a;
        a.list = tf.list;
        a.count = tf.count;
        a.nq = nq;
        a.slab = h->w_dist.as<float>();
        a.q_stride = h->tie.q_stride;
        a.pair_off = h->w_pair_off.as<int>();
        a.pair_base = h->w_pair_base.as<int64_t>();
        a.ids = h->d_ids;
        a.P = p->nprobe;
        a.G = h->tie.G;
        const size_t rq_bytes = (((size_t)nq + 1) * sizeof(int) + 7) & ~(size_t)7;
        unsigned long long* ready = reinterpret_cast<unsigned long long*>(h->w_scnt.as<char>() + rq_bytes);
        a.ready = h->tie.bounded ? ready : nullptr;
        a.surv = h->w_surv.as<unsigned long long>();
        a.gcnt = reinterpret_cast<int*>(ready + nq);
        a.nsl = h->tie.nsl;
        a.slice_cap = h->tie.cap;
        a.x = d_x;
        a.d = h->d;
        a.raw = h->d_raw;
        a.nraw = h->nraw;
        a.R = R;
        a.k = k;
        a.has_rank = p->has_rank ? 1 : 0;
        a.min_score = p->min_score;
        a.max_score = p->max_score;
        a.neutral = neutral;
        a.cand_dis = const_cast<float*>(cand_dis);
        a.cand_ids = const_cast<int64_t*>(cand_ids);
        a.distances = d_distances;
        a.labels = d_labels;
        gh::launch_tie_replay(s, l2, a);
    };
    if (p->has_rank) {
        if (!h->d_raw || h->raw_d != h->d) return fail(h, GAMMA_HIP_EINVAL, "has_rank needs the raw store");
        if (R <= 1024 && (nq >= 256 || ties)) {
            // one fused kernel: exact distances + top-k + output
            gh::launch_rerank_topk(s, l2, d_x, nq, h->d, h->d_raw, h->nraw, cand_ids, R, k, p->min_score,
                                   p->max_score, neutral, d_distances, d_labels, qperm, ties ? &tf : nullptr);
            if (ties) replay();
            GH_CHECK(h, hipGetLastError());
            return GAMMA_HIP_OK;
        }
        GH_CHECK(h, h->w_exact.ensure((size_t)nq * R * sizeof(float)));
        GH_CHECK(h, h->w_selv.ensure((size_t)nq * k * sizeof(float)));
        GH_CHECK(h, h->w_selp.ensure((size_t)nq * k * sizeof(int)));
        gh::launch_rerank_dist(s, l2, d_x, nq, h->d, h->d_raw, h->nraw, cand_ids, R, p->min_score,
                               p->max_score, h->w_exact.as<float>());
        gh::launch_select_topk(s, l2, h->w_exact.as<float>(), R, nullptr, R, R, nq, k,
                               h->w_selv.as<float>(), h->w_selp.as<int>());
        gh::launch_finalize_topk(s, h->w_selv.as<float>(), h->w_selp.as<int>(), nq, k, cand_ids, R, 0,
                                 neutral, d_distances, d_labels);
    } else {
        gh::launch_finalize_norank(s, cand_dis, cand_ids, nq, R, k, p->min_score, p->max_score, neutral,
                                   d_distances, d_labels, ties ? &tf : nullptr);
        if (ties) replay();
    }
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

// ---- small batches (nq <= 512; measured cross-over with the regular chain ~1000): four or five launches instead of eleven ---------------------------------
// exact coarse distances + query tables | top-nprobe + slab offsets | scan | top-recall_num + ids + re-rank + top-k
// (kernels.hip k_small_coarse_ip, select.hip k_small_coarse_select / k_small_tail).  Each launch of the regular
// chain costs ~4 us of launch + drain at this size, whatever it computes.
bool ivfpq_small_ok(H* h, const gamma_hip_search_params* p, const FiltCtx& fc, int nq, int R) {
    static const bool off = getenv("GAMMA_HIP_NO_SMALL_PATH") != nullptr;
    static const int max_nq = getenv("GAMMA_HIP_SMALL_MAX") ? atoi(getenv("GAMMA_HIP_SMALL_MAX")) : 512;
    // exact coarse distances (faiss below 20 queries) come from the fused first kernel, which covers 16 queries; the
    // GEMM form (20 queries and more) from the regular matrix kernel
    return !off && h->small_path && nq >= 1 && nq <= max_nq && (p->coarse_mode == 1 || nq <= 16) &&
           p->nprobe <= 64 && R <= 1024 && !h->exact_ties && !h->profile && !fc.d_qf && !h->d_list_mask &&
           h->nlist <= 16384 &&
           (int64_t)p->nprobe * std::max(1, h->max_list_len) <= (1 << 22) &&
           // long lists: beyond ~5e7 codes per call the regular chain's bound filter wins (full-size C4, 390 k codes per
           // query: 64 queries 0.46 ms against 1.04, 256 queries 1.63 against 1.33)
           (int64_t)nq * p->nprobe * (h->ntotal / std::max(1, h->nlist)) <= 48000000LL && (!p->has_rank || (h->d_raw && h->raw_d == h->d));
}

int ivfpq_small(H* h, const gamma_hip_search_params* p, const FiltCtx& fc, int nq, const float* d_x, int R, int k,
                float* d_distances, int64_t* d_labels) {
    const int P = p->nprobe, d = h->d, M = h->M, nlist = h->nlist;
    hipStream_t s = h->stream;
    const int ver = h->cur_ver;
    GH_CHECK(h, hipStreamWaitEvent(s, h->ver_ev[ver], 0));
    GH_CHECK(h, h->w_mat.ensure((size_t)nq * nlist * sizeof(float)));
    GH_CHECK(h, h->w_st2.ensure((size_t)nq * M * 256 * sizeof(float)));
    GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nq * P * sizeof(float)));
    GH_CHECK(h, h->w_probe.ensure((size_t)nq * P * sizeof(int)));
    GH_CHECK(h, h->w_pair_off.ensure((size_t)nq * (P + 1) * sizeof(int)));
    GH_CHECK(h, h->w_pair_base.ensure((size_t)nq * P * sizeof(int64_t)));
    GH_CHECK(h, h->w_qtotal.ensure((size_t)nq * sizeof(int)));
    GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq * R * sizeof(int)));
    GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq * R * sizeof(float)));
    GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq * R * sizeof(int64_t)));
    const int64_t q_stride = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
    GH_CHECK(h, h->w_dist.ensure((size_t)nq * q_stride * sizeof(float)));
    // (folding the selection into the first launch -- last workgroup done selects -- was tried: the device-scope
    // release / acquire it needs costs more than the launch it saves, 24 us against 4 + 8: the XCDs' L2s are
    // written back and invalidated either way)
    // long lists: the scan walks a work list of (query, probe, chunk of the list) units written by the selection kernel,
    // pieces of even size for a grid that fills the chip, instead of one workgroup per pair that runs for as long as its
    // list is (small_presel: tests force the path on short lists)
    static const int chunk_env = getenv("GAMMA_HIP_SMALL_CHUNK") ? atoi(getenv("GAMMA_HIP_SMALL_CHUNK")) : 512;
    int chunk_len = 0, max_units = 0;
    uint32_t* d_units = nullptr;
    int* d_nunits = nullptr;
    if (h->ntotal / std::max(1, nlist) > 1024 || h->max_list_len > 8192 || h->small_presel > 0) {
        chunk_len = h->small_presel > 0 ? 512 : std::max(512, (chunk_env + 511) & ~511);
        const int64_t mu = (int64_t)nq * P * (1 + (int64_t)h->max_list_len / chunk_len);
        max_units = (int)std::min<int64_t>(mu, INT32_MAX);
        GH_CHECK(h, h->w_lm_units.ensure((size_t)mu * sizeof(uint32_t)));
        GH_CHECK(h, h->w_lm_cnt.ensure(64));
        d_units = h->w_lm_units.as<uint32_t>();
        d_nunits = h->w_lm_cnt.as<int>();
    }
    if (p->coarse_mode == 1) {
        if (d_nunits) GH_CHECK(h, hipMemsetAsync(d_nunits, 0, sizeof(int), s));
        gh::launch_l2_gemmform(s, d_x, nq, d, h->d_cc, nlist, nullptr, h->d_cc_norms, h->w_mat.as<float>(), nlist, true);
        gh::launch_pq_ip_table(s, d_x, nq, d, M, h->d_pqc, h->w_st2.as<float>());
    } else if (!gh::launch_small_coarse_ip(s, d_x, nq, d, h->d_cc, nlist, h->w_mat.as<float>(), M, h->d_pqc,
                                           h->w_st2.as<float>(), d_nunits)) {
        return fail(h, GAMMA_HIP_EINVAL, "small path: shape not covered");   // ivfpq_small_ok gates on the same shapes
    }
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    if (!l2) GH_CHECK(h, h->w_pair_ip.ensure((size_t)nq * P * sizeof(float)));
    gh::launch_small_coarse_select(s, h->w_mat.as<float>(), nlist, nq, P, h->w_coarse_dis.as<float>(), h->w_probe.as<int>(),
                                   h->d_list_len, h->d_list_mask, h->d_list_off, h->w_pair_off.as<int>(),
                                   h->w_qtotal.as<int>(), h->w_pair_base.as<int64_t>(), d_x, h->d_cc, d,
                                   l2 ? nullptr : h->w_pair_ip.as<float>(), d_units, d_nunits, chunk_len);
    h->scan_pairs += (int64_t)nq * P;
    const int need_ids = (fc.any_clause || (h->d_bitmap && h->bitmap_any) || h->n_moved > 0) ? 1 : 0;
    gh::launch_ivfpq_scan_pair(s, l2, d_x, nq, d, M, P, h->w_probe.as<int>(),
                               l2 ? h->w_coarse_dis.as<float>() : h->w_pair_ip.as<float>(), h->d_cc,
                               h->w_st2.as<float>(), h->d_T2, h->d_list_off, h->d_list_len, h->d_list_mask, nlist,
                               h->d_codes, h->d_ids, h->w_pair_off.as<int>(), q_stride, h->w_dist.as<float>(), fc.d_tab,
                               fc.d_qf, need_ids, nullptr, 1, 0, P, 0, nullptr, nullptr, reinterpret_cast<const int*>(d_units), d_nunits,
                               chunk_len, max_units);
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    // long candidate rows (expected nprobe x 1.5 mean list lengths beyond what one workgroup keeps in registers): a first
    // selection over slices of the row by several workgroups per query, then the tail among their survivors
    int smax = 0;
    {
        const int64_t slice = 16384;   // select.hip SM_SLICE
        const int64_t est = (int64_t)P * (h->ntotal / std::max(1, nlist)) * 3 / 2;
        const int64_t bound = (int64_t)P * std::max(1, h->max_list_len);
        if (h->small_presel > 0) smax = h->small_presel;
        else if (est > slice) smax = (int)std::min<int64_t>(std::min<int64_t>(64, (bound + slice - 1) / slice), std::max(2, 4096 / nq));
        if (smax > 0) {
            GH_CHECK(h, h->w_selv.ensure((size_t)nq * smax * R * sizeof(float)));
            GH_CHECK(h, h->w_selp.ensure((size_t)nq * smax * R * sizeof(int)));
        }
    }
    gh::launch_small_tail(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nq, R, P, h->w_probe.as<int>(),
                          h->w_pair_off.as<int>(), h->d_list_off, h->d_ids, h->w_cand_dis.as<float>(),
                          h->w_cand_pos.as<int>(), h->w_cand_ids.as<int64_t>(), p->has_rank ? 1 : 0, d_x, d, h->d_raw,
                          h->nraw, k, p->min_score, p->max_score, neutral, d_distances, d_labels, smax,
                          smax ? h->w_selv.as<float>() : nullptr, smax ? h->w_selp.as<int>() : nullptr);
    h->tie = H::TieCtx();
    h->tie.G = 1;
    h->tie.q_stride = q_stride;
    h->last_qperm = nullptr;
    GH_CHECK(h, hipEventRecord(h->rd_ev[ver], s));
    h->rd_set[ver] = true;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int ivfpq_check(H* h, const gamma_hip_search_params* p, int nq, int k) {
    GH_TRY(check_params(h, p, nq, k));
    if (!h->ivf_init || h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "ivfpq not trained");
    if (p->nprobe <= 0 || p->nprobe > h->nlist) return fail(h, GAMMA_HIP_EINVAL, "nprobe out of range");
    if (std::max(p->recall_num, k) > 4096) return fail(h, GAMMA_HIP_EINVAL, "recall_num > 4096 unsupported");
    return GAMMA_HIP_OK;
}

// queries per internal chunk: the coarse distance matrix (nlist floats per query) and the ADC distance
// slab (nprobe x longest list floats per query) each stay inside the workspace budget
int coarse_chunk(H* h, int nq) {
    const int64_t by_mat = (int64_t)(h->dist_budget_bytes / ((size_t)h->nlist * sizeof(float)));
    return (int)std::max<int64_t>(1, std::min<int64_t>(by_mat, nq));
}
int scan_chunk(H* h, int nq, int P) {
    const int64_t q_stride = std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len));
    const int64_t by_dist = (int64_t)(h->dist_budget_bytes / (q_stride * sizeof(float)));
    return (int)std::max<int64_t>(1, std::min<int64_t>(by_dist, nq));
}
int query_chunk(H* h, int nq, int P) { return std::min(coarse_chunk(h, nq), scan_chunk(h, nq, P)); }

// given != nullptr: the filter context of a combined batch (p's own filter clauses are ignored)
int ivfpq_search_device_locked(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int k,
                               float* d_distances, int64_t* d_labels, const FiltCtx* given = nullptr) {
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;  // gamma_index_ivfpq.cc:753-756
    GH_CHECK(h, hipSetDevice(h->device));
    const int R = std::max(p->recall_num, k);
    FiltCtx fc;
    if (given) {
        fc = *given;
    } else {
        gh::FilterDesc filt;
        GH_TRY(build_filter(h, p, &filt));
        GH_TRY(filt_ctx_single(h, filt, &fc));
    }
    // faiss picks the coarse path from the size of the WHOLE call (faiss:utils/distances.cpp:346);
    // the internal chunks must not re-decide it
    gamma_hip_search_params pp = *p;
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;
    p = &pp;
    if (ivfpq_small_ok(h, p, fc, nq, R)) {
        GH_TRY(ivfpq_small(h, p, fc, nq, d_x, R, k, d_distances, d_labels));
        h->last_nq = nq;
        h->last_P = p->nprobe;
        h->last_R = R;
        return GAMMA_HIP_OK;
    }
    const int chunk = scan_chunk(h, nq, p->nprobe), P = p->nprobe;
    // long lists (C4: 64 probes x lists of tens of thousands) make the ADC slab the limit: the coarse
    // quantizer then still runs over the whole call (one GEMM instead of one per slab chunk)
    const bool coarse_first = chunk < nq;
    if (coarse_first) {
        GH_CHECK(h, h->w_full_cdis.ensure((size_t)nq * P * sizeof(float)));
        GH_CHECK(h, h->w_full_probe.ensure((size_t)nq * P * sizeof(int)));
        const int cc = coarse_chunk(h, nq);
        for (int q0 = 0; q0 < nq; q0 += cc)
            GH_TRY(ivfpq_coarse(h, p, std::min(cc, nq - q0), d_x + (size_t)q0 * h->d,
                                h->w_full_cdis.as<float>() + (size_t)q0 * P, h->w_full_probe.as<int>() + (size_t)q0 * P));
    }
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nc = std::min(chunk, nq - q0);
        if (coarse_first)
            GH_TRY(ivfpq_stage_a(h, p, fc.at(q0), nc, d_x + (size_t)q0 * h->d, R,
                                 h->w_full_cdis.as<float>() + (size_t)q0 * P, h->w_full_probe.as<int>() + (size_t)q0 * P));
        else
            GH_TRY(ivfpq_stage_a(h, p, fc.at(q0), nc, d_x + (size_t)q0 * h->d, R));
        GH_TRY(ivfpq_stage_b(h, p, nc, d_x + (size_t)q0 * h->d, R, k, h->w_cand_dis.as<float>(),
                             h->w_cand_ids.as<int64_t>(), d_distances + (size_t)q0 * k,
                             d_labels + (size_t)q0 * k, getenv("GAMMA_HIP_NO_RERANK_ORDER") ? nullptr : h->last_qperm,
                             /*tie_replay=*/true));
        h->last_nq = nc;
    }
    h->last_P = p->nprobe;
    h->last_R = R;
    return GAMMA_HIP_OK;
}

// ---- IVFFLAT (index/impl/gamma_index_ivfflat.cc:392-567) ------------------------------------------------------
// coarse quantizer (the IVFPQ one) -> slab offsets -> exact distance of every entry of the probed lists
// (k_ivfflat_scan) -> top-k of the slab in (distance, scan position) order -> ids.  The reference's k-heap keeps
// the same k entries (up to its order inside exact ties).
// IVFFLAT, small batches: the chain of ivfpq_small without tables and re-rank -- exact coarse distances | top-nprobe +
// slab offsets | exact distances of the probed lists' rows, one workgroup per pair | top-k + ids + score window
int ivfflat_small(H* h, const gamma_hip_search_params* p, const FiltCtx& fc, int nq, const float* d_x, int k,
                  float* d_distances, int64_t* d_labels) {
    const int P = p->nprobe, d = h->d, nlist = h->nlist;
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    hipStream_t s = h->stream;
    const int ver = h->cur_ver;
    GH_CHECK(h, hipStreamWaitEvent(s, h->ver_ev[ver], 0));
    GH_CHECK(h, h->w_mat.ensure((size_t)nq * nlist * sizeof(float)));
    GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nq * P * sizeof(float)));
    GH_CHECK(h, h->w_probe.ensure((size_t)nq * P * sizeof(int)));
    GH_CHECK(h, h->w_pair_off.ensure((size_t)nq * (P + 1) * sizeof(int)));
    GH_CHECK(h, h->w_pair_base.ensure((size_t)nq * P * sizeof(int64_t)));
    GH_CHECK(h, h->w_qtotal.ensure((size_t)nq * sizeof(int)));
    GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq * k * sizeof(int)));
    GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq * k * sizeof(float)));
    GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq * k * sizeof(int64_t)));
    const int64_t q_stride = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
    GH_CHECK(h, h->w_dist.ensure((size_t)nq * q_stride * sizeof(float)));
    if (p->coarse_mode == 1) {
        gh::launch_l2_gemmform(s, d_x, nq, d, h->d_cc, nlist, nullptr, h->d_cc_norms, h->w_mat.as<float>(), nlist, true);
    } else if (!gh::launch_small_coarse_ip(s, d_x, nq, d, h->d_cc, nlist, h->w_mat.as<float>(), 0, nullptr, nullptr)) {
        return fail(h, GAMMA_HIP_EINVAL, "small path: shape not covered");
    }
    gh::launch_small_coarse_select(s, h->w_mat.as<float>(), nlist, nq, P, h->w_coarse_dis.as<float>(), h->w_probe.as<int>(),
                                   h->d_list_len, h->d_list_mask, h->d_list_off, h->w_pair_off.as<int>(),
                                   h->w_qtotal.as<int>(), h->w_pair_base.as<int64_t>());
    const int need_filter = (fc.any_clause || (h->d_bitmap && h->bitmap_any)) ? 1 : 0;
    gh::launch_ivfflat_scan(s, l2, d_x, nq, d, P, h->w_pair_off.as<int>(), h->w_pair_base.as<int64_t>(), h->d_ids, h->d_raw,
                            h->nraw, q_stride, h->w_dist.as<float>(), fc.d_tab, need_filter, p->min_score, p->max_score);
    int smax = 0;   // long candidate rows: two-level selection (ivfpq_small)
    {
        const int64_t slice = 16384;
        const int64_t est = (int64_t)P * (h->ntotal / std::max(1, nlist)) * 3 / 2;
        const int64_t bound = (int64_t)P * std::max(1, h->max_list_len);
        if (h->small_presel > 0) smax = h->small_presel;
        else if (est > slice) smax = (int)std::min<int64_t>(std::min<int64_t>(64, (bound + slice - 1) / slice), std::max(2, 4096 / nq));
        if (smax > 0) {
            GH_CHECK(h, h->w_selv.ensure((size_t)nq * smax * k * sizeof(float)));
            GH_CHECK(h, h->w_selp.ensure((size_t)nq * smax * k * sizeof(int)));
        }
    }
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    gh::launch_small_tail(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), nq, k, P, h->w_probe.as<int>(),
                          h->w_pair_off.as<int>(), h->d_list_off, h->d_ids, h->w_cand_dis.as<float>(),
                          h->w_cand_pos.as<int>(), h->w_cand_ids.as<int64_t>(), 0, d_x, d, h->d_raw, h->nraw, k, p->min_score,
                          p->max_score, neutral, d_distances, d_labels, smax, smax ? h->w_selv.as<float>() : nullptr,
                          smax ? h->w_selp.as<int>() : nullptr);
    GH_CHECK(h, hipEventRecord(h->rd_ev[ver], s));
    h->rd_set[ver] = true;
    h->last_nq = nq;
    h->last_P = P;
    h->last_R = k;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int ivfflat_search_device_locked(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int k,
                                 float* d_distances, int64_t* d_labels) {
    GH_TRY(check_params(h, p, nq, k));
    if (!h->ivf_init || !h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfflat not initialised");
    if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "ivfflat not trained");
    if (p->nprobe <= 0 || p->nprobe > h->nlist) return fail(h, GAMMA_HIP_EINVAL, "nprobe out of range");
    if (!h->d_raw || h->raw_d != h->d) return fail(h, GAMMA_HIP_EINVAL, "ivfflat needs the raw store");
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt));
    FiltCtx fc;
    GH_TRY(filt_ctx_single(h, filt, &fc));
    gamma_hip_search_params pp = *p;
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;   // faiss:utils/distances.cpp:303,346, whole call
    const int P = pp.nprobe, nlist = h->nlist;
    hipStream_t s = h->stream;
    {   // small batches, as long as the pair-per-workgroup scan is the one that would run anyway (below 2 nlist pairs)
        static const bool off = getenv("GAMMA_HIP_NO_SMALL_PATH") != nullptr;
        if (!off && h->small_path && nq <= 512 && (pp.coarse_mode == 1 || nq <= 16) && P <= 64 && k <= 1024 && !h->profile &&
            !fc.d_qf && !h->d_list_mask && nlist <= 16384 && (int64_t)nq * P < 2 * (int64_t)nlist &&
            (int64_t)P * std::max(1, h->max_list_len) <= (1 << 22))
            return ivfflat_small(h, &pp, fc, nq, d_x, k, d_distances, d_labels);
    }
    const int chunk = scan_chunk(h, nq, P);
    const int need_filter = (fc.any_clause || (h->d_bitmap && h->bitmap_any)) ? 1 : 0;
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nc = std::min(chunk, nq - q0);
        const float* xq = d_x + (size_t)q0 * h->d;
        const int ver = h->cur_ver;
        GH_CHECK(h, hipStreamWaitEvent(s, h->ver_ev[ver], 0));
        GH_CHECK(h, h->w_pair_off.ensure((size_t)nc * (P + 1) * sizeof(int)));
        GH_CHECK(h, h->w_pair_base.ensure((size_t)nc * P * sizeof(int64_t)));
        GH_CHECK(h, h->w_qtotal.ensure((size_t)nc * sizeof(int)));
        GH_CHECK(h, h->w_cand_pos.ensure((size_t)nc * k * sizeof(int)));
        GH_CHECK(h, h->w_cand_dis.ensure((size_t)nc * k * sizeof(float)));
        GH_CHECK(h, h->w_cand_ids.ensure((size_t)nc * k * sizeof(int64_t)));
        GH_TRY(ivfpq_coarse(h, &pp, nc, xq));
        gh::launch_pair_offsets(s, h->w_probe.as<int>(), nc, P, h->d_list_len, h->d_list_mask, nlist,
                                h->w_pair_off.as<int>(), h->w_qtotal.as<int>(), nullptr, h->d_list_off,
                                h->w_pair_base.as<int64_t>());
        const int64_t q_stride = (std::max<int64_t>(1, (int64_t)P * std::max(1, h->max_list_len)) + 3) & ~(int64_t)3;
        GH_CHECK(h, h->w_dist.ensure((size_t)nc * q_stride * sizeof(float)));
        {
            StageScope t(h, GAMMA_HIP_STAGE_SCAN);
            // enough (query, probe) pairs that lists are shared: list-major (ivfflat.hip), a list's rows are read once
            // for all the queries probing it; else one workgroup per pair
            static const bool no_lm = getenv("GAMMA_HIP_NO_IVFFLAT_LM") != nullptr;
            if (!no_lm && gh::ivfflat_lm_supported(h->d) && (int64_t)nc * P >= 2 * (int64_t)nlist && !h->d_list_mask) {
                GH_CHECK(h, h->w_lm_units.ensure(gh::ivfflat_lm_scratch_bytes(nc, P, nlist)));
                gh::launch_ivfflat_lm(s, l2, xq, nc, h->d, P, h->w_probe.as<int>(), h->w_pair_off.as<int>(), h->d_list_off,
                                      h->d_list_len, nlist, h->d_ids, h->d_raw, h->nraw, q_stride, h->w_dist.as<float>(),
                                      fc.d_tab, need_filter, p->min_score, p->max_score, h->w_lm_units.p);
            } else {
                gh::launch_ivfflat_scan(s, l2, xq, nc, h->d, P, h->w_pair_off.as<int>(), h->w_pair_base.as<int64_t>(),
                                        h->d_ids, h->d_raw, h->nraw, q_stride, h->w_dist.as<float>(), fc.d_tab, need_filter,
                                        p->min_score, p->max_score);
            }
        }
        {
            StageScope t(h, GAMMA_HIP_STAGE_SELECT);
            gh::launch_select_topk(s, l2, h->w_dist.as<float>(), q_stride, h->w_qtotal.as<int>(), 0,
                                   (int)std::min<int64_t>(q_stride, 1 << 30), nc, k, h->w_cand_dis.as<float>(),
                                   h->w_cand_pos.as<int>());
            gh::launch_map_candidates(s, h->w_cand_pos.as<int>(), nc, k, P, h->w_probe.as<int>(), h->w_pair_off.as<int>(),
                                      h->d_list_off, h->d_ids, h->w_cand_ids.as<int64_t>());
            gh::launch_finalize_norank(s, h->w_cand_dis.as<float>(), h->w_cand_ids.as<int64_t>(), nc, k, k, p->min_score,
                                       p->max_score, neutral, d_distances + (size_t)q0 * k, d_labels + (size_t)q0 * k,
                                       nullptr);
        }
        GH_CHECK(h, hipEventRecord(h->rd_ev[ver], s));
        h->rd_set[ver] = true;
        h->last_nq = nc;
    }
    h->last_P = P;
    h->last_R = k;
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

// ---- flat ------------------------------------------------------------------------------
int flat_search_device_locked(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int k,
                              float* d_distances, int64_t* d_labels) {
    GH_TRY(check_params(h, p, nq, k));
    if (!h->d_raw && h->nraw > 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const float neutral = l2 ? 3.402823466e+38f : -3.402823466e+38f;
    const float sentinel = l2 ? INFINITY : -INFINITY;
    const int d = h->raw_d;
    const int64_t N = h->nraw;
    hipStream_t s = h->stream;
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt));
    // query chunks x row chunks so the distance slab stays inside the budget
    int64_t rows_chunk = std::max<int64_t>(256, std::min<int64_t>(N, (int64_t)1 << 16));
    rows_chunk = (rows_chunk + 255) / 256 * 256;
    int qc = (int)std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)(h->dist_budget_bytes / (rows_chunk * sizeof(float)))));
    const int nchunks = (int)std::max<int64_t>(1, (N + rows_chunk - 1) / rows_chunk);
    GH_CHECK(h, h->w_dist.ensure((size_t)qc * rows_chunk * sizeof(float)));
    GH_CHECK(h, h->w_part_v.ensure((size_t)qc * nchunks * k * sizeof(float)));
    GH_CHECK(h, h->w_part_i.ensure((size_t)qc * nchunks * k * sizeof(int64_t)));
    GH_CHECK(h, h->w_selv.ensure((size_t)qc * k * sizeof(float)));
    GH_CHECK(h, h->w_selp.ensure((size_t)qc * k * sizeof(int)));
    GH_CHECK(h, h->w_cand_pos.ensure((size_t)qc * k * sizeof(int)));
    GH_CHECK(h, h->w_cand_dis.ensure((size_t)qc * k * sizeof(float)));
    StageScope t(h, GAMMA_HIP_STAGE_FLAT);
    // the reference's loop: every row, one query at a time, a k-heap (gamma_index_flat.cc:118-300).
    // Here: distance slab of one row chunk -> per-chunk top-k -> merge of the chunks' tables.
    auto unbounded = [&](int q0, int nc) -> int {
        const float* xq = d_x + (size_t)q0 * d;
        for (int c = 0; c < nchunks; c++) {
            const int64_t r0 = (int64_t)c * rows_chunk;
            const int64_t nr = std::min<int64_t>(rows_chunk, N - r0);
            gh::launch_pairwise_filtered(s, l2, xq, nc, d, h->d_raw + r0 * d, nr, h->w_dist.as<float>(),
                                         rows_chunk, filt, p->min_score, p->max_score, r0);
            // per-chunk top-k: values + positions relative to the chunk
            gh::launch_select_topk(s, l2, h->w_dist.as<float>(), rows_chunk, nullptr, (int)nr, (int)nr, nc, k,
                                   h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>());
            // scatter into the partial table [q][chunk][k] with global ids
            // (reuse finalize_topk: labels = pos, then offset by r0 on the fly below)
            gh::launch_finalize_topk(s, h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>(), nc, k, nullptr,
                                     0, r0, sentinel, h->w_selv.as<float>(),
                                     h->w_part_i.as<int64_t>() + (size_t)c * nc * k);
            GH_CHECK(h, hipMemcpyAsync(h->w_part_v.as<float>() + (size_t)c * nc * k, h->w_selv.p,
                                       (size_t)nc * k * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        // merge: layout [chunk][q][k] == the sharded layout [shard][nq][R]
        GH_CHECK(h, h->w_m_dis.ensure((size_t)nc * nchunks * k * sizeof(float)));
        GH_CHECK(h, h->w_m_ids.ensure((size_t)nc * nchunks * k * sizeof(int64_t)));
        gh::launch_gather_shards(s, h->w_part_v.as<float>(), h->w_part_i.as<int64_t>(), nchunks, nc, k,
                                 h->w_m_dis.as<float>(), h->w_m_ids.as<int64_t>(), sentinel);
        gh::launch_select_topk(s, l2, h->w_m_dis.as<float>(), (int64_t)nchunks * k, nullptr, nchunks * k,
                               nchunks * k, nc, k, h->w_selv.as<float>(), h->w_selp.as<int>());
        gh::launch_finalize_topk(s, h->w_selv.as<float>(), h->w_selp.as<int>(), nc, k,
                                 h->w_m_ids.as<int64_t>(), (int64_t)nchunks * k, 0, neutral,
                                 d_distances + (size_t)q0 * k, d_labels + (size_t)q0 * k);
        return GAMMA_HIP_OK;
    };
    // Running bound: only the first chunk goes through a distance slab.  Its k-th best bounds the
    // answer; the remaining rows are scored in passes that double the rows seen so far, each pass
    // appending only the distances within the current bound to the query's candidate list (about k
    // per pass and query) and ending with a compaction that tightens the bound.  The k smallest
    // (distance, row id) items are the same either way.  A list that overflows (rows arriving in
    // improving order) is detected and the call redone without a bound.
    const int cap = gh::flat_list_cap();
    auto bounded = [&](int q0, int nc, bool* redo) -> int {
        const float* xq = d_x + (size_t)q0 * d;
        GH_CHECK(h, h->w_flat_cand.ensure((size_t)nc * cap * sizeof(unsigned long long)));
        GH_CHECK(h, h->w_flat_meta.ensure((size_t)(2 * nc + 1) * sizeof(int)));   // tau[nc] | cnt[nc] | overflow
        uint32_t* tau = h->w_flat_meta.as<uint32_t>();
        int* cnt = h->w_flat_meta.as<int>() + nc;
        int* over = cnt + nc;
        gh::FlatEmit em{tau, h->w_flat_cand.as<unsigned long long>(), cnt, cap};
        GH_CHECK(h, hipMemsetAsync(over, 0, sizeof(int), s));
        gh::launch_pairwise_filtered(s, l2, xq, nc, d, h->d_raw, rows_chunk, h->w_dist.as<float>(), rows_chunk, filt,
                                     p->min_score, p->max_score, 0);
        gh::launch_select_topk(s, l2, h->w_dist.as<float>(), rows_chunk, nullptr, (int)rows_chunk, (int)rows_chunk,
                               nc, k, h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>());
        gh::launch_flat_init(s, l2, h->w_cand_dis.as<float>(), h->w_cand_pos.as<int>(), nc, k, 0, em, tau);
        for (int64_t r = rows_chunk; r < N;) {
            const int64_t nr = std::min<int64_t>(r, N - r);
            gh::launch_pairwise_emit(s, l2, xq, nc, d, h->d_raw + r * d, nr, filt, p->min_score, p->max_score, r, em);
            gh::launch_flat_compact(s, nc, k, em, tau, over);
            r += nr;
        }
        gh::launch_flat_final(s, l2, nc, k, em, neutral, d_distances + (size_t)q0 * k, d_labels + (size_t)q0 * k);
        GH_CHECK(h, hipGetLastError());
        int h_over = 0;
        GH_CHECK(h, hipMemcpyAsync(&h_over, over, sizeof(int), hipMemcpyDeviceToHost, s));
        GH_CHECK(h, hipStreamSynchronize(s));
        *redo = h_over != 0;
        return GAMMA_HIP_OK;
    };
    for (int q0 = 0; q0 < nq; q0 += qc) {
        const int nc = std::min(qc, nq - q0);
        if (N == 0) {
            // nothing to scan: all-empty result
            GH_CHECK(h, hipMemsetAsync(h->w_selp.p, 0xff, (size_t)nc * k * sizeof(int), s));
            gh::launch_finalize_topk(s, h->w_selv.as<float>(), h->w_selp.as<int>(), nc, k, nullptr, 0, 0,
                                     neutral, d_distances + (size_t)q0 * k, d_labels + (size_t)q0 * k);
            continue;
        }
        bool redo = true;
        if (k <= 256 && N > rows_chunk && N < ((int64_t)1 << 32) &&
            gh::pairwise_can_emit(nc, d, N - rows_chunk))
            GH_TRY(bounded(q0, nc, &redo));
        if (redo) GH_TRY(unbounded(q0, nc));
    }
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

// host-pointer wrapper shared by ivfpq / flat
// sync = false: everything is only enqueued (pinned host buffers); the caller synchronises the stream
// lk != nullptr: the caller's SearchLock; its mu is released once everything is enqueued, so writers go on while
// this call waits for the GPU (search_mu stays: the workspaces are in use)
template <typename F>
int host_search(H* h, int nq, int d, const float* x, int k, float* distances, int64_t* labels, F&& f,
                bool sync = true, SearchLock* lk = nullptr) {
    if (nq <= 0 || k <= 0) return f(nullptr, nullptr, nullptr);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, h->w_x.ensure((size_t)nq * d * sizeof(float)));
    GH_CHECK(h, h->w_outd.ensure((size_t)nq * k * sizeof(float)));
    GH_CHECK(h, h->w_outl.ensure((size_t)nq * k * sizeof(int64_t)));
    // Small synchronous calls (a client thread's single query): the caller's buffers are pageable, and a pageable
    // copy is a blocking staged transfer -- three of them cost more than the search chain.  Queries and results go
    // through a pinned staging area instead: the copies are true asynchronous transfers in stream order, the thread
    // blocks once, and the results are copied out by the CPU.
    const size_t bx = (size_t)nq * d * sizeof(float), bd = (size_t)nq * k * sizeof(float), bi = (size_t)nq * k * sizeof(int64_t);
    const size_t off_i = (bx + 63) & ~(size_t)63, off_d = off_i + ((bi + 63) & ~(size_t)63), need = off_d + bd;
    static const bool no_pin = getenv("GAMMA_HIP_NO_PINNED_CALLS") != nullptr;
    if (sync && !no_pin && need <= ((size_t)1 << 20)) {
        if (need > h->dir_pin_bytes) {
            if (h->dir_pin) (void)hipHostFree(h->dir_pin);
            h->dir_pin = nullptr;
            h->dir_pin_bytes = 0;
            GH_CHECK(h, hipHostMalloc(&h->dir_pin, std::max<size_t>(need * 2, 65536), hipHostMallocDefault));
            h->dir_pin_bytes = std::max<size_t>(need * 2, 65536);
        }
        char* base = static_cast<char*>(h->dir_pin);
        std::memcpy(base, x, bx);
        GH_CHECK(h, hipMemcpyAsync(h->w_x.p, base, bx, hipMemcpyHostToDevice, h->stream));
        // results: the last kernel of the chain stores them straight into the staging area (pinned host memory is
        // mapped into the device's address space; a few KB of posted writes) -- no copy back at all
        static const bool no_map = getenv("GAMMA_HIP_NO_MAPPED_RESULTS") != nullptr;
        void* dbase = nullptr;
        if (!no_map && hipHostGetDevicePointer(&dbase, base, 0) == hipSuccess && dbase) {
            char* db = static_cast<char*>(dbase);
            GH_TRY(f(h->w_x.as<float>(), reinterpret_cast<float*>(db + off_d), reinterpret_cast<int64_t*>(db + off_i)));
        } else {
            GH_TRY(f(h->w_x.as<float>(), h->w_outd.as<float>(), h->w_outl.as<int64_t>()));
            GH_CHECK(h, hipMemcpyAsync(base + off_d, h->w_outd.p, bd, hipMemcpyDeviceToHost, h->stream));
            GH_CHECK(h, hipMemcpyAsync(base + off_i, h->w_outl.p, bi, hipMemcpyDeviceToHost, h->stream));
        }
        if (lk) lk->enqueued();
        GH_CHECK(h, hipStreamSynchronize(h->stream));
        std::memcpy(distances, base + off_d, bd);
        std::memcpy(labels, base + off_i, bi);
        return GAMMA_HIP_OK;
    }
    GH_CHECK(h, hipMemcpyAsync(h->w_x.p, x, (size_t)nq * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    GH_TRY(f(h->w_x.as<float>(), h->w_outd.as<float>(), h->w_outl.as<int64_t>()));
    GH_CHECK(h, hipMemcpyAsync(distances, h->w_outd.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    GH_CHECK(h, hipMemcpyAsync(labels, h->w_outl.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    if (lk) lk->enqueued();
    if (sync) GH_CHECK(h, hipStreamSynchronize(h->stream));
    return GAMMA_HIP_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* gamma_hip_strerror(int code) {
    switch (code) {
        case GAMMA_HIP_OK: return "ok";
        case GAMMA_HIP_EINVAL: return "invalid argument";
        case GAMMA_HIP_ENOTTRAINED: return "index not trained";
        case GAMMA_HIP_EDEVICE: return "HIP runtime error";
        case GAMMA_HIP_ENOMEM: return "out of memory";
        case GAMMA_HIP_EFULL: return "inverted list full";
        default: return "unknown error";
    }
}

int gamma_hip_create(int device, gamma_hip_index** out) {
    if (!out) return GAMMA_HIP_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GAMMA_HIP_EDEVICE;
    if (device < 0 || device >= ndev) return GAMMA_HIP_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return GAMMA_HIP_EDEVICE;
    H* h = new (std::nothrow) H();
    if (!h) return GAMMA_HIP_ENOMEM;
    h->device = device;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->wstream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return GAMMA_HIP_EDEVICE;
    }
    for (int v = 0; v < H::NVER; v++)
        if (hipEventCreateWithFlags(&h->ver_ev[v], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->rd_ev[v], hipEventDisableTiming) != hipSuccess) {
            delete h;
            return GAMMA_HIP_EDEVICE;
        }
    if (hipMalloc((void**)&h->d_scan_codes, sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(h->d_scan_codes, 0, sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&h->d_tie_stats, 3 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(h->d_tie_stats, 0, 3 * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipStreamDestroy(h->stream);
        delete h;
        return GAMMA_HIP_EDEVICE;
    }
    // workspace budget of the chunked buffers: an eighth of the device memory, 1..32 GiB (36 GB -> 32 GiB
    // on a 288 GB MI355X); gamma_hip_set_workspace_budget overrides
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0)
        h->dist_budget_bytes = std::min<size_t>((size_t)32 << 30, std::max<size_t>((size_t)1 << 30, total_b / 8));
    *out = h;
    return GAMMA_HIP_OK;
}

int gamma_hip_destroy(gamma_hip_index* h) {
    if (!h) return GAMMA_HIP_OK;
    {
        std::unique_lock<std::mutex> lk(h->comb_mu);
        h->comb_stop = true;
        h->comb_wcv.notify_all();
    }
    if (h->comb_thread.joinable()) h->comb_thread.join();
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamSynchronize(h->wstream);
    for (int v = 0; v < H::NVER; v++) {
        if (h->ver_ev[v]) (void)hipEventDestroy(h->ver_ev[v]);
        if (h->rd_ev[v]) (void)hipEventDestroy(h->rd_ev[v]);
        if (h->d_ver_off[v]) (void)hipFree(h->d_ver_off[v]);
        if (h->d_ver_len[v]) (void)hipFree(h->d_ver_len[v]);
        if (h->pin_ver[v]) (void)hipHostFree(h->pin_ver[v]);
    }
    for (auto& e : h->events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    void* ptrs[] = {h->d_list_rank, h->d_raw, h->d_bitmap, h->d_cc, h->d_cc_norms, h->d_pqc, h->d_T2, h->d_codes,
                    h->d_ids, h->d_list_mask, h->d_scan_codes, h->d_tie_stats, h->d_v2d};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& kv : h->fields)
        if (kv.second.d) (void)hipFree(kv.second.d);
    for (auto& kv : h->terms) {
        if (kv.second.d_off) (void)hipFree(kv.second.d_off);
        if (kv.second.d_tok) (void)hipFree(kv.second.d_tok);
    }
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    for (void* pp : h->comb_pin)
        if (pp) (void)hipHostFree(pp);
    if (h->dir_pin) (void)hipHostFree(h->dir_pin);
    DevBuf* bufs[] = {&h->w_mat, &h->w_coarse_dis, &h->w_probe, &h->w_xn, &h->w_st2, &h->w_pair_off,
                      &h->w_qtotal, &h->w_dist, &h->w_cand_dis, &h->w_cand_pos, &h->w_cand_ids,
                      &h->w_exact, &h->w_selv, &h->w_selp, &h->w_x, &h->w_outd, &h->w_outl, &h->w_stage,
                      &h->w_filter, &h->w_m_dis, &h->w_m_ids, &h->w_part_v, &h->w_part_i, &h->w_assign,
                      &h->w_codes_tmp, &h->w_qperm, &h->w_scnt, &h->w_sflag, &h->w_surv, &h->w_pair_base,
                      &h->w_pair_ip, &h->w_flat_cand, &h->w_flat_meta, &h->w_full_cdis,
                      &h->w_full_probe, &h->w_ftab, &h->w_qfil, &h->w_tieflag, &h->w_tcut, &h->w_tlist, &h->w_survc, &h->w_lm_units, &h->w_lm_cnt,
                      &h->we_mat, &h->we_cdis, &h->we_x, &h->we_assign, &h->we_codes, &h->we_stage};
    for (DevBuf* b : bufs) b->release();
    (void)hipStreamDestroy(h->stream);
    (void)hipStreamDestroy(h->wstream);
    delete h;
    return GAMMA_HIP_OK;
}

const char* gamma_hip_last_error(gamma_hip_index* h) { return h ? h->err.c_str() : "null handle"; }
void* gamma_hip_stream(gamma_hip_index* h) { return h ? (void*)h->stream : nullptr; }

int gamma_hip_synchronize(gamma_hip_index* h) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

/* ---- raw store ---------------------------------------------------------------------- */
namespace {
size_t field_elem_size(int dtype) {
    return dtype == GAMMA_HIP_FIELD_INT || dtype == GAMMA_HIP_FIELD_FLOAT ? 4 : 8;
}
}  // namespace

int gamma_hip_set_exact_ties(gamma_hip_index* h, int on) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->exact_ties = on != 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_set_coarse_fused(gamma_hip_index* h, int on, int list_cap) {
    if (!h || list_cap < 1 || list_cap > gh::kCoarseCap) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->coarse_fused = on != 0;
    h->coarse_cap = list_cap;
    return GAMMA_HIP_OK;
}

int gamma_hip_set_small_path(gamma_hip_index* h, int on) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->small_path = on != 0;
    h->small_presel = on >= 2 ? on : 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_set_list_major(gamma_hip_index* h, int on) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->list_major = on != 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_tie_stats(gamma_hip_index* h, int64_t* out3, int reset) {
    if (!h || !out3) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    unsigned long long v[3];
    GH_CHECK(h, hipMemcpy(v, h->d_tie_stats, sizeof(v), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) out3[i] = (int64_t)v[i];
    if (reset) GH_CHECK(h, hipMemset(h->d_tie_stats, 0, sizeof(v)));
    return GAMMA_HIP_OK;
}

int gamma_hip_set_workspace_budget(gamma_hip_index* h, int64_t bytes) {
    if (!h || bytes <= 0) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->dist_budget_bytes = (size_t)bytes;
    return GAMMA_HIP_OK;
}

int gamma_hip_field_append(gamma_hip_index* h, int field_id, int dtype, int64_t n, const void* values) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (dtype < GAMMA_HIP_FIELD_INT || dtype > GAMMA_HIP_FIELD_DOUBLE || n < 0 || (n > 0 && !values))
        return fail(h, GAMMA_HIP_EINVAL, "bad column append");
    GH_CHECK(h, hipSetDevice(h->device));
    auto& c = h->fields[field_id];
    if (c.n == 0 && c.cap == 0) c.dtype = dtype;
    if (c.dtype != dtype) return fail(h, GAMMA_HIP_EINVAL, "column dtype mismatch");
    const size_t es = field_elem_size(dtype);
    if (c.n + n > c.cap) {
        const int64_t ncap = std::max<int64_t>(c.n + n, std::max<int64_t>(1 << 16, c.cap * 2));
        uint8_t* nd = nullptr;
        GH_CHECK(h, lk.exclusive());   // the old column is freed below
        GH_CHECK(h, hipMalloc((void**)&nd, (size_t)ncap * es));
        if (c.n) GH_CHECK(h, hipMemcpyAsync(nd, c.d, (size_t)c.n * es, hipMemcpyDeviceToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        if (c.d) (void)hipFree(c.d);
        c.d = nd;
        c.cap = ncap;
    }
    if (n) {
        GH_CHECK(h, hipMemcpyAsync(c.d + (size_t)c.n * es, values, (size_t)n * es, hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
    }
    c.n += n;
    return GAMMA_HIP_OK;
}

int gamma_hip_field_update(gamma_hip_index* h, int field_id, int64_t docid, const void* value) {
    if (!h || !value) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    auto it = h->fields.find(field_id);
    if (it == h->fields.end() || docid < 0 || docid >= it->second.n) return fail(h, GAMMA_HIP_EINVAL, "bad column update");
    GH_CHECK(h, hipSetDevice(h->device));
    const size_t es = field_elem_size(it->second.dtype);
    GH_CHECK(h, hipMemcpyAsync(it->second.d + (size_t)docid * es, value, es, hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

int gamma_hip_vid2docid_append(gamma_hip_index* h, int64_t n, const int32_t* docids) {
    if (!h || n < 0 || (n > 0 && !docids)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    const int64_t have = (int64_t)h->h_v2d.size();
    if (have + n > h->v2d_cap) {
        const int64_t ncap = std::max<int64_t>(have + n, std::max<int64_t>(1 << 16, h->v2d_cap * 2));
        int32_t* nd = nullptr;
        GH_CHECK(h, lk.exclusive());   // the old array is freed below
        GH_CHECK(h, hipMalloc((void**)&nd, (size_t)ncap * sizeof(int32_t)));
        if (have) GH_CHECK(h, hipMemcpyAsync(nd, h->d_v2d, (size_t)have * sizeof(int32_t), hipMemcpyDeviceToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        if (h->d_v2d) (void)hipFree(h->d_v2d);
        h->d_v2d = nd;
        h->v2d_cap = ncap;
    }
    if (n) {
        GH_CHECK(h, hipMemcpyAsync(h->d_v2d + have, docids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        h->h_v2d.insert(h->h_v2d.end(), docids, docids + n);   // published last: searches enqueued before see the shorter map
    }
    return GAMMA_HIP_OK;
}

int64_t gamma_hip_vid2docid_count(gamma_hip_index* h) {
    if (!h) return -1;
    std::lock_guard<std::mutex> g(h->mu);
    return (int64_t)h->h_v2d.size();
}

int64_t gamma_hip_field_count(gamma_hip_index* h, int field_id) {
    if (!h) return -1;
    std::lock_guard<std::mutex> g(h->mu);
    auto it = h->fields.find(field_id);
    return it == h->fields.end() ? 0 : it->second.n;
}

int gamma_hip_term_append(gamma_hip_index* h, int field_id, int64_t n_docs, const int32_t* counts,
                          const int32_t* items) {
    if (!h || n_docs < 0 || (n_docs > 0 && !counts)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    auto& c = h->terms[field_id];
    int64_t add_tok = 0;
    for (int64_t i = 0; i < n_docs; i++) {
        if (counts[i] < 0) return fail(h, GAMMA_HIP_EINVAL, "negative item count");
        add_tok += counts[i];
    }
    if (add_tok > 0 && !items) return fail(h, GAMMA_HIP_EINVAL, "null items");
    if (c.ndocs + n_docs + 1 > c.cap_docs || c.ntok + add_tok > c.cap_tok || !c.d_off) {
        // growth frees the old arrays: no search may be reading them
        GH_CHECK(h, lk.exclusive());
        const int64_t nd = std::max<int64_t>(c.ndocs + n_docs + 1, std::max<int64_t>(1 << 16, c.cap_docs * 2));
        const int64_t nt = std::max<int64_t>(c.ntok + add_tok, std::max<int64_t>(1 << 16, c.cap_tok * 2));
        int64_t* no = nullptr;
        int32_t* ntk = nullptr;
        GH_CHECK(h, hipMalloc((void**)&no, (size_t)nd * sizeof(int64_t)));
        GH_CHECK(h, hipMalloc((void**)&ntk, (size_t)nt * sizeof(int32_t)));
        if (c.d_off) {
            GH_CHECK(h, hipMemcpyAsync(no, c.d_off, (size_t)(c.ndocs + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, h->wstream));
            if (c.ntok) GH_CHECK(h, hipMemcpyAsync(ntk, c.d_tok, (size_t)c.ntok * sizeof(int32_t), hipMemcpyDeviceToDevice, h->wstream));
        } else {
            GH_CHECK(h, hipMemsetAsync(no, 0, sizeof(int64_t), h->wstream));   // off[0] = 0
        }
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        if (c.d_off) (void)hipFree(c.d_off);
        if (c.d_tok) (void)hipFree(c.d_tok);
        c.d_off = no;
        c.d_tok = ntk;
        c.cap_docs = nd;
        c.cap_tok = nt;
    }
    if (n_docs > 0) {
        std::vector<int64_t> off(n_docs);
        int64_t run = c.ntok;
        for (int64_t i = 0; i < n_docs; i++) {
            run += counts[i];
            off[i] = run;
        }
        if (add_tok) GH_CHECK(h, hipMemcpyAsync(c.d_tok + c.ntok, items, (size_t)add_tok * sizeof(int32_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(c.d_off + c.ndocs + 1, off.data(), (size_t)n_docs * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        c.ntok += add_tok;
        c.ndocs += n_docs;   // published last: a search enqueued before sees the shorter column
    }
    return GAMMA_HIP_OK;
}

int64_t gamma_hip_term_count(gamma_hip_index* h, int field_id) {
    if (!h) return -1;
    std::lock_guard<std::mutex> g(h->mu);
    auto it = h->terms.find(field_id);
    return it == h->terms.end() ? 0 : it->second.ndocs;
}

int gamma_hip_raw_init(gamma_hip_index* h, int d) {
    if (!h || d <= 0) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->raw_d != 0 && h->raw_d != d) return fail(h, GAMMA_HIP_EINVAL, "raw store dimension mismatch");
    h->raw_d = d;
    return GAMMA_HIP_OK;
}

static int raw_reserve(H* h, int64_t need) {
    if (need <= h->raw_cap) return GAMMA_HIP_OK;
    int64_t ncap = std::max<int64_t>(need, h->raw_cap + h->raw_cap / 2);
    ncap = std::max<int64_t>(ncap, 1024);
    float* np = nullptr;
    if (h->wl) GH_CHECK(h, h->wl->exclusive());   // the old store is freed below
    GH_CHECK(h, hipMalloc((void**)&np, (size_t)ncap * h->raw_d * sizeof(float)));
    if (h->nraw > 0)
        GH_CHECK(h, hipMemcpyAsync(np, h->d_raw, (size_t)h->nraw * h->raw_d * sizeof(float),
                                   hipMemcpyDeviceToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    if (h->d_raw) GH_CHECK(h, hipFree(h->d_raw));
    h->d_raw = np;
    h->raw_cap = ncap;
    return GAMMA_HIP_OK;
}

int gamma_hip_raw_append(gamma_hip_index* h, int64_t n, const float* vecs) {
    if (!h || n < 0 || (n > 0 && !vecs)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (n == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(raw_reserve(h, h->nraw + n));
    GH_CHECK(h, hipMemcpyAsync(h->d_raw + h->nraw * h->raw_d, vecs, (size_t)n * h->raw_d * sizeof(float),
                               hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->nraw += n;
    return GAMMA_HIP_OK;
}

int gamma_hip_raw_write(gamma_hip_index* h, int64_t first_vid, int64_t n, const float* vecs) {
    if (!h || n < 0 || first_vid < 0 || (n > 0 && !vecs)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (first_vid > h->nraw) return fail(h, GAMMA_HIP_EINVAL, "raw write would leave a gap");
    if (n == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(raw_reserve(h, first_vid + n));
    GH_CHECK(h, hipMemcpyAsync(h->d_raw + first_vid * h->raw_d, vecs, (size_t)n * h->raw_d * sizeof(float),
                               hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->nraw = std::max(h->nraw, first_vid + n);
    return GAMMA_HIP_OK;
}

int gamma_hip_raw_update(gamma_hip_index* h, int64_t vid, const float* vec) {
    if (!h || !vec) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (vid < 0 || vid >= h->nraw) return fail(h, GAMMA_HIP_EINVAL, "vid out of range");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipMemcpyAsync(h->d_raw + vid * h->raw_d, vec, (size_t)h->raw_d * sizeof(float),
                               hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

int64_t gamma_hip_raw_count(gamma_hip_index* h) { return h ? h->nraw : -1; }

/* ---- delete bitmap ------------------------------------------------------------------- */
static int bitmap_reserve(H* h, int64_t nbits) {
    size_t bytes = (((size_t)nbits >> 3) + 1 + 3) & ~(size_t)3;
    if (bytes <= h->bitmap_cap_bytes) return GAMMA_HIP_OK;
    size_t ncap = std::max(bytes, h->bitmap_cap_bytes * 2);
    uint8_t* np = nullptr;
    if (h->wl) GH_CHECK(h, h->wl->exclusive());   // the old bitmap is freed below
    GH_CHECK(h, hipMalloc((void**)&np, ncap));
    GH_CHECK(h, hipMemsetAsync(np, 0, ncap, h->wstream));
    if (h->d_bitmap)
        GH_CHECK(h, hipMemcpyAsync(np, h->d_bitmap, h->bitmap_cap_bytes, hipMemcpyDeviceToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    if (h->d_bitmap) GH_CHECK(h, hipFree(h->d_bitmap));
    h->d_bitmap = np;
    h->bitmap_cap_bytes = ncap;
    h->h_bitmap.resize(ncap, 0);
    return GAMMA_HIP_OK;
}

int gamma_hip_bitmap_upload(gamma_hip_index* h, const uint8_t* bitmap, int64_t nbits) {
    if (!h || nbits < 0 || (nbits > 0 && !bitmap)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(bitmap_reserve(h, nbits));
    size_t bytes = ((size_t)nbits >> 3) + 1;  // bitmap::create, util/bitmap.cc:15-23
    GH_CHECK(h, hipMemsetAsync(h->d_bitmap, 0, h->bitmap_cap_bytes, h->wstream));
    GH_CHECK(h, hipMemcpyAsync(h->d_bitmap, bitmap, bytes, hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    std::fill(h->h_bitmap.begin(), h->h_bitmap.end(), 0);
    memcpy(h->h_bitmap.data(), bitmap, bytes);
    h->bitmap_bits = nbits;
    h->bitmap_any = false;
    for (size_t i = 0; i < bytes && !h->bitmap_any; i++) h->bitmap_any = bitmap[i] != 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_bitmap_set(gamma_hip_index* h, const int64_t* docids, int64_t n, int value) {
    if (!h || n < 0 || (n > 0 && !docids)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (n == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    int64_t mx = 0;
    for (int64_t i = 0; i < n; i++) mx = std::max(mx, docids[i]);
    if (mx >= h->bitmap_bits) {
        GH_TRY(bitmap_reserve(h, mx + 1));
        h->bitmap_bits = std::max<int64_t>(h->bitmap_bits, mx + 1);
    }
    GH_CHECK(h, h->we_stage.ensure((size_t)n * sizeof(int64_t)));
    GH_CHECK(h, hipMemcpyAsync(h->we_stage.p, docids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
    gh::launch_bitmap_set(h->wstream, h->d_bitmap, h->we_stage.as<int64_t>(), n, h->bitmap_bits, value);
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    for (int64_t i = 0; i < n; i++) {
        int64_t id = docids[i];
        if (id < 0) continue;
        if (value) h->bitmap_any = true;
        if (value) h->h_bitmap[id >> 3] |= (uint8_t)(1u << (id & 7));
        else h->h_bitmap[id >> 3] &= (uint8_t)~(1u << (id & 7));
    }
    return GAMMA_HIP_OK;
}

/* ---- IVFPQ / IVFFLAT models ----------------------------------------------------------- */
static int ivf_init_locked(gamma_hip_index* h, int d, int nlist, int M, int metric, int bucket_init_size,
                           int bucket_max_size, bool flat);
int gamma_hip_ivfpq_init(gamma_hip_index* h, int d, int nlist, int M, int nbits, int metric,
                         int bucket_init_size, int bucket_max_size) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "already initialised");
    if (d <= 0 || nlist <= 0 || M <= 0) return fail(h, GAMMA_HIP_EINVAL, "bad d/nlist/M");
    if (nbits != 8) return fail(h, GAMMA_HIP_EINVAL, "only nbits_per_idx == 8 is supported on device");
    if (d % M != 0) return fail(h, GAMMA_HIP_EINVAL, "d must be divisible by nsubvector");
    if (d / M > 64) return fail(h, GAMMA_HIP_EINVAL, "dsub > 64 unsupported");
    if (M > 64) return fail(h, GAMMA_HIP_EINVAL, "nsubvector > 64 unsupported (LUT must fit 64 KiB LDS)");
    return ivf_init_locked(h, d, nlist, M, metric, bucket_init_size, bucket_max_size, false);
}

static int ivf_init_locked(gamma_hip_index* h, int d, int nlist, int M, int metric, int bucket_init_size,
                           int bucket_max_size, bool flat) {
    if (metric != GAMMA_HIP_METRIC_IP && metric != GAMMA_HIP_METRIC_L2) return fail(h, GAMMA_HIP_EINVAL, "bad metric");
    GH_CHECK(h, hipSetDevice(h->device));
    h->ivfflat = flat;
    h->d = d;
    h->nlist = nlist;
    h->M = M;
    h->dsub = d / M;
    h->code_size = M;   // IVFFLAT: M = 1, one dummy byte per entry (the arena code keeps its shape)
    h->metric = metric;
    h->bucket_init = bucket_init_size > 0 ? bucket_init_size : 1000;
    h->bucket_max = bucket_max_size > 0 ? bucket_max_size : 1280000;
    GH_CHECK(h, hipMalloc((void**)&h->d_cc, (size_t)nlist * d * sizeof(float)));
    GH_CHECK(h, hipMalloc((void**)&h->d_cc_norms, (size_t)nlist * sizeof(float)));
    GH_CHECK(h, hipMalloc((void**)&h->d_pqc, flat ? 256 : (size_t)M * 256 * h->dsub * sizeof(float)));
    GH_CHECK(h, hipMalloc((void**)&h->d_T2, flat ? 256 : (size_t)nlist * M * 256 * sizeof(float)));
    for (int v = 0; v < H::NVER; v++) {
        GH_CHECK(h, hipMalloc((void**)&h->d_ver_off[v], (size_t)nlist * sizeof(int64_t)));
        GH_CHECK(h, hipMalloc((void**)&h->d_ver_len[v], (size_t)nlist * sizeof(int)));
        GH_CHECK(h, hipHostMalloc(&h->pin_ver[v], (size_t)nlist * (sizeof(int64_t) + sizeof(int)), hipHostMallocDefault));
    }
    h->d_list_off = h->d_ver_off[0];
    h->d_list_len = h->d_ver_len[0];
    // RTInvertBucketData::Init (realtime_mem_data.cc:57-96): bucket_init entries per list
    h->h_list_off.resize(nlist);
    h->h_list_len.assign(nlist, 0);
    h->h_list_cap.assign(nlist, h->bucket_init);
    h->h_deleted.assign(nlist, 0);
    h->h_extend_time.assign(nlist, 0);
    for (int l = 0; l < nlist; l++) h->h_list_off[l] = (int64_t)l * h->bucket_init;
    h->arena_used = 0;
    GH_TRY(arena_reserve(h, (int64_t)nlist * h->bucket_init));
    h->arena_used = (int64_t)nlist * h->bucket_init;
    GH_TRY(publish_meta(h));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->vid_pos.assign((size_t)nlist * h->bucket_init, -1);
    h->ivf_init = true;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfflat_init(gamma_hip_index* h, int d, int nlist, int metric, int bucket_init_size, int bucket_max_size) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "already initialised");
    if (d <= 0 || nlist <= 0) return fail(h, GAMMA_HIP_EINVAL, "bad d/nlist");
    return ivf_init_locked(h, d, nlist, 1, metric, bucket_init_size, bucket_max_size, true);
}

int gamma_hip_ivfflat_set_trained(gamma_hip_index* h, const float* cc) {
    if (!h || !cc) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init || !h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfflat not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipMemcpyAsync(h->d_cc, cc, (size_t)h->nlist * h->d * sizeof(float), hipMemcpyHostToDevice, h->wstream));
    gh::launch_row_norms(h->wstream, h->d_cc, h->nlist, h->d, h->d_cc_norms);
    GH_CHECK(h, hipGetLastError());
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->trained = true;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_set_trained(gamma_hip_index* h, const float* cc, const float* pqc, const float* table) {
    if (!h || !cc || !pqc) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init || h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    const size_t ncc = (size_t)h->nlist * h->d, npq = (size_t)h->M * 256 * h->dsub;
    const size_t nt = (size_t)h->nlist * h->M * 256;
    GH_CHECK(h, hipMemcpyAsync(h->d_cc, cc, ncc * sizeof(float), hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipMemcpyAsync(h->d_pqc, pqc, npq * sizeof(float), hipMemcpyHostToDevice, h->wstream));
    gh::launch_row_norms(h->wstream, h->d_cc, h->nlist, h->d, h->d_cc_norms);
    if (table)
        GH_CHECK(h, hipMemcpyAsync(h->d_T2, table, nt * sizeof(float), hipMemcpyHostToDevice, h->wstream));
    else
        gh::launch_precompute_table(h->wstream, h->d_cc, h->nlist, h->d, h->M, h->d_pqc, h->d_T2);
    {
        std::vector<int> rank = centroid_rank(cc, h->nlist, h->d);
        if (!h->d_list_rank) GH_CHECK(h, hipMalloc((void**)&h->d_list_rank, (size_t)h->nlist * sizeof(int)));
        GH_CHECK(h, hipMemcpyAsync(h->d_list_rank, rank.data(), (size_t)h->nlist * sizeof(int),
                                   hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));   // rank is a local
    }
    GH_CHECK(h, hipGetLastError());
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->trained = true;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_get_precomputed_table(gamma_hip_index* h, float* out) {
    if (!h || !out) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "not trained");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipMemcpyAsync(out, h->d_T2, (size_t)h->nlist * h->M * 256 * sizeof(float),
                               hipMemcpyDeviceToHost, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

/* ---- realtime lists ------------------------------------------------------------------- */
int gamma_hip_ivfpq_add_keys(gamma_hip_index* h, int list_no, int n, const int64_t* vids, const uint8_t* codes) {
    if (!h || (n > 0 && (!vids || !codes))) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    return add_keys_locked(h, list_no, n, vids, codes);
}

int gamma_hip_ivfpq_add_keys_batch(gamma_hip_index* h, int nlists, const int32_t* list_nos,
                                   const int32_t* counts, const int64_t* vids, const uint8_t* codes) {
    if (!h || nlists < 0 || (nlists > 0 && (!list_nos || !counts || !vids || !codes))) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    int64_t off = 0;
    // grow first so that no copy below races with an arena move; a list named twice reserves for the sum
    {
        std::map<int, int64_t> per_list;
        for (int i = 0; i < nlists; i++) {
            if (list_nos[i] < 0 || list_nos[i] >= h->nlist || counts[i] < 0) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
            per_list[list_nos[i]] += counts[i];
        }
        for (auto& kv : per_list) {
            if (kv.second > h->bucket_max) return fail(h, GAMMA_HIP_EFULL, "exceed the max bucket keys");
            GH_TRY(list_ensure(h, kv.first, (int)kv.second));
        }
    }
    for (int i = 0; i < nlists; i++) {
        const int l = list_nos[i], n = counts[i];
        if (n == 0) continue;
        const int64_t pos = h->h_list_off[l] + h->h_list_len[l];
        GH_CHECK(h, hipMemcpyAsync(h->d_ids + pos, vids + off, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(h->d_codes + pos * h->code_size, codes + off * h->code_size,
                                   (size_t)n * h->code_size, hipMemcpyHostToDevice, h->wstream));
        for (int j = 0; j < n; j++) {
            const int64_t v = vids[off + j];
            if (v < 0) {   // superseded slot restored from a dump: same accounting as add_keys_locked, so that
                h->h_deleted[l]++;   // the scan reads the ids (n_moved) and never returns the slot
                h->n_moved++;
                continue;
            }
            if ((size_t)v >= h->vid_pos.size()) h->vid_pos.resize(std::max<size_t>(h->vid_pos.size() * 2, v + 1), -1);
            h->vid_pos[v] = ((int64_t)l << 32) | (int64_t)(h->h_list_len[l] + j);
            if (h->doc_deleted(v)) h->h_deleted[l]++;
        }
        h->h_list_len[l] += n;
        h->ntotal += n;
        if (h->h_list_len[l] > h->max_list_len) h->max_list_len = h->h_list_len[l];
        off += n;
    }
    // publish the new lengths (and moved extents) after the copies, in stream order
    GH_TRY(publish_meta(h));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return arena_repack_if_need(h);
}

int gamma_hip_ivfpq_update(gamma_hip_index* h, int list_no, int64_t vid, const uint8_t* code) {
    if (!h || !code) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init || list_no < 0 || list_no >= h->nlist) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
    if (vid < 0 || (size_t)vid >= h->vid_pos.size()) return GAMMA_HIP_OK;  // realtime_mem_data.cc:307
    const int64_t bp = h->vid_pos[vid];
    if (bp == -1) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const int ob = (int)(bp >> 32), op = (int)(bp & 0xffffffff);
    if (ob == list_no) {
        GH_CHECK(h, hipMemcpyAsync(h->d_codes + (h->h_list_off[ob] + op) * h->code_size, code, h->code_size,
                                   hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        return GAMMA_HIP_OK;
    }
    gh::launch_mark_moved(h->wstream, h->d_ids, h->h_list_off[ob] + op);
    h->h_deleted[ob]++;
    h->n_moved++;
    h->ntotal -= 1;  // add_keys_locked re-counts it
    return add_keys_locked(h, list_no, 1, &vid, code);
}

int gamma_hip_ivfpq_has_vid(gamma_hip_index* h, const int64_t* vids, int n, uint8_t* out) {
    if (!h || n < 0 || (n > 0 && (!vids || !out))) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    for (int i = 0; i < n; i++)
        out[i] = vids[i] >= 0 && (size_t)vids[i] < h->vid_pos.size() && h->vid_pos[vids[i]] != -1;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_remove(gamma_hip_index* h, int64_t vid) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    if (vid < 0 || (size_t)vid >= h->vid_pos.size()) return GAMMA_HIP_OK;
    const int64_t bp = h->vid_pos[vid];
    if (bp == -1) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const int ob = (int)(bp >> 32), op = (int)(bp & 0xffffffff);
    gh::launch_mark_moved(h->wstream, h->d_ids, h->h_list_off[ob] + op);
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->h_deleted[ob]++;
    h->n_moved++;
    h->ntotal -= 1;
    h->vid_pos[vid] = -1;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_delete(gamma_hip_index* h, const int64_t* vids, int n) {
    if (!h || (n > 0 && !vids)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    for (int i = 0; i < n; i++) {
        if (vids[i] < 0 || (size_t)vids[i] >= h->vid_pos.size()) continue;
        const int64_t bp = h->vid_pos[vids[i]];
        if (bp == -1) continue;
        h->h_deleted[bp >> 32]++;
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_compact_if_need(gamma_hip_index* h) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    std::vector<int64_t> ids;
    std::vector<uint8_t> codes;
    bool changed = false;
    for (int l = 0; l < h->nlist; l++) {
        const int len = h->h_list_len[l];
        if (!((float)h->h_deleted[l] / len >= 0.3f)) continue;  // Compactable, :373-377
        ids.resize(len);
        codes.resize((size_t)len * h->code_size);
        GH_CHECK(h, hipMemcpyAsync(ids.data(), h->d_ids + h->h_list_off[l], (size_t)len * sizeof(int64_t), hipMemcpyDeviceToHost, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(codes.data(), h->d_codes + h->h_list_off[l] * h->code_size, (size_t)len * h->code_size, hipMemcpyDeviceToHost, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        int pos = 0;
        for (int i = 0; i < len; i++) {  // CompactOne, :98-112
            const int64_t id = ids[i];
            const int64_t v = id & ~kDelMask;
            const bool deleted = h->doc_deleted(v);
            if (!(id & kDelMask) && !deleted) {
                ids[pos] = id;
                memmove(codes.data() + (size_t)pos * h->code_size, codes.data() + (size_t)i * h->code_size, h->code_size);
                h->vid_pos[id] = ((int64_t)l << 32) | pos;
                pos++;
            }
        }
        // new region of the same capacity (copy-on-write swap, :426-474)
        GH_TRY(arena_reserve(h, h->h_list_cap[l]));
        const int64_t noff = h->arena_used;
        h->arena_used += h->h_list_cap[l];
        h->arena_waste += h->h_list_cap[l];
        if (pos > 0) {
            GH_CHECK(h, hipMemcpyAsync(h->d_ids + noff, ids.data(), (size_t)pos * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
            GH_CHECK(h, hipMemcpyAsync(h->d_codes + noff * h->code_size, codes.data(), (size_t)pos * h->code_size, hipMemcpyHostToDevice, h->wstream));
        }
        h->h_list_off[l] = noff;
        h->ntotal -= (len - pos);
        h->h_list_len[l] = pos;
        h->h_deleted[l] = 0;
        GH_CHECK(h, hipStreamSynchronize(h->wstream));   // ids / codes are locals of this loop
        changed = true;
    }
    if (changed) {
        GH_TRY(publish_meta(h));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        GH_TRY(arena_repack_if_need(h));
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_arena_stats(gamma_hip_index* h, int64_t* out4) {
    if (!h || !out4) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    out4[0] = h->arena_cap;
    out4[1] = h->arena_used;
    out4[2] = h->arena_waste;
    out4[3] = h->n_repacks;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_set_repack_threshold(gamma_hip_index* h, int64_t min_waste_entries) {
    if (!h || min_waste_entries < 0) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    h->repack_min_entries = min_waste_entries;
    return arena_repack_if_need(h);
}

int64_t gamma_hip_ivfpq_list_size(gamma_hip_index* h, int l) {
    if (!h || !h->ivf_init || l < 0 || l >= h->nlist) return -1;
    return h->h_list_len[l];
}
int64_t gamma_hip_ivfpq_list_capacity(gamma_hip_index* h, int l) {
    if (!h || !h->ivf_init || l < 0 || l >= h->nlist) return -1;
    return h->h_list_cap[l];
}

int gamma_hip_ivfpq_get_list(gamma_hip_index* h, int l, int64_t* vids, uint8_t* codes) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init || l < 0 || l >= h->nlist) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
    const int len = h->h_list_len[l];
    if (len == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    if (vids) GH_CHECK(h, hipMemcpyAsync(vids, h->d_ids + h->h_list_off[l], (size_t)len * sizeof(int64_t), hipMemcpyDeviceToHost, h->wstream));
    if (codes) GH_CHECK(h, hipMemcpyAsync(codes, h->d_codes + h->h_list_off[l] * h->code_size, (size_t)len * h->code_size, hipMemcpyDeviceToHost, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_set_list_mask(gamma_hip_index* h, const uint8_t* owned) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, lk.exclusive());
    if (!owned) {
        if (h->d_list_mask) GH_CHECK(h, hipFree(h->d_list_mask));
        h->d_list_mask = nullptr;
        h->h_list_mask.clear();
        return GAMMA_HIP_OK;
    }
    if (!h->d_list_mask) GH_CHECK(h, hipMalloc((void**)&h->d_list_mask, (size_t)h->nlist));
    h->h_list_mask.assign(owned, owned + h->nlist);
    GH_CHECK(h, hipMemcpyAsync(h->d_list_mask, owned, (size_t)h->nlist, hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

/* ---- device-side encode / add ----------------------------------------------------------- */
// exact: the arithmetic form faiss picks from the size of the WHOLE assign() call (n < 20), not of a chunk
static int encode_locked(H* h, int64_t n, const float* d_vecs, int* d_assign, uint8_t* d_codes_out, bool exact) {
    // quantizer->assign == search with k = 1 (faiss rule for the arithmetic form)
    hipStream_t s = h->wstream;   // own stream and own workspace: runs beside the searches
    const int d = h->d, nlist = h->nlist;
    GH_CHECK(h, h->we_mat.ensure((size_t)n * nlist * sizeof(float)));
    GH_CHECK(h, h->we_cdis.ensure((size_t)n * sizeof(float)));
    if (exact) {
        gh::launch_pairwise(s, true, d_vecs, (int)n, d, h->d_cc, nlist, h->we_mat.as<float>(), nlist);
    } else {
        gh::launch_l2_gemmform(s, d_vecs, (int)n, d, h->d_cc, nlist, nullptr, h->d_cc_norms,
                               h->we_mat.as<float>(), nlist, true);
    }
    gh::launch_select_topk(s, true, h->we_mat.as<float>(), nlist, nullptr, nlist, nlist, (int)n, 1,
                           h->we_cdis.as<float>(), d_assign);
    if (h->ivfflat) GH_CHECK(h, hipMemsetAsync(d_codes_out, 0, (size_t)n, s));   // the dummy byte of every entry
    else gh::launch_pq_encode(s, d_vecs, n, d, h->M, d_assign, h->d_cc, h->d_pqc, d_codes_out);
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_encode(gamma_hip_index* h, int64_t n, const float* vecs, int64_t* list_nos, uint8_t* codes) {
    if (!h || n < 0 || (n > 0 && (!vecs || !list_nos || !codes))) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "not trained");
    if (n == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n, 65536), (int64_t)(h->dist_budget_bytes / ((size_t)h->nlist * sizeof(float)))));
    std::vector<int> assign(chunk);
    for (int64_t i0 = 0; i0 < n; i0 += chunk) {
        const int64_t nc = std::min(chunk, n - i0);
        GH_CHECK(h, h->we_x.ensure((size_t)nc * h->d * sizeof(float)));
        GH_CHECK(h, h->we_assign.ensure((size_t)nc * sizeof(int)));
        GH_CHECK(h, h->we_codes.ensure((size_t)nc * h->code_size));
        GH_CHECK(h, hipMemcpyAsync(h->we_x.p, vecs + i0 * h->d, (size_t)nc * h->d * sizeof(float), hipMemcpyHostToDevice, h->wstream));
        GH_TRY(encode_locked(h, nc, h->we_x.as<float>(), h->we_assign.as<int>(), h->we_codes.as<uint8_t>(), n < 20));
        GH_CHECK(h, hipMemcpyAsync(assign.data(), h->we_assign.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(codes + i0 * h->code_size, h->we_codes.p, (size_t)nc * h->code_size, hipMemcpyDeviceToHost, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        for (int64_t i = 0; i < nc; i++) list_nos[i0 + i] = assign[i];
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_assign(gamma_hip_index* h, int d, int64_t n, const float* x, int k, const float* centroids,
                     int32_t* assign, float* dis) {
    if (!h || d <= 0 || n < 0 || k <= 0 || (n > 0 && (!x || !centroids || !assign))) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    if (n == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // centroids + their norms live in the (otherwise unused here) partial-result buffers
    GH_CHECK(h, h->w_part_v.ensure((size_t)k * d * sizeof(float)));
    GH_CHECK(h, h->w_xn.ensure((size_t)k * sizeof(float)));
    GH_CHECK(h, hipMemcpyAsync(h->w_part_v.p, centroids, (size_t)k * d * sizeof(float), hipMemcpyHostToDevice, s));
    gh::launch_row_norms(s, h->w_part_v.as<float>(), k, d, h->w_xn.as<float>());
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(h->dist_budget_bytes / ((size_t)k * sizeof(float)))));
    for (int64_t i0 = 0; i0 < n; i0 += chunk) {
        const int64_t nc = std::min(chunk, n - i0);
        GH_CHECK(h, h->w_x.ensure((size_t)nc * d * sizeof(float)));
        GH_CHECK(h, h->w_mat.ensure((size_t)nc * k * sizeof(float)));
        GH_CHECK(h, h->w_assign.ensure((size_t)nc * sizeof(int)));
        GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nc * sizeof(float)));
        GH_CHECK(h, hipMemcpyAsync(h->w_x.p, x + i0 * d, (size_t)nc * d * sizeof(float), hipMemcpyHostToDevice, s));
        gh::launch_l2_gemmform(s, h->w_x.as<float>(), (int)nc, d, h->w_part_v.as<float>(), k, nullptr,
                               h->w_xn.as<float>(), h->w_mat.as<float>(), k, true);
        gh::launch_select_topk(s, true, h->w_mat.as<float>(), k, nullptr, k, k, (int)nc, 1,
                               h->w_coarse_dis.as<float>(), h->w_assign.as<int>());
        GH_CHECK(h, hipGetLastError());
        GH_CHECK(h, hipMemcpyAsync(assign + i0, h->w_assign.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, s));
        if (dis) GH_CHECK(h, hipMemcpyAsync(dis + i0, h->w_coarse_dis.p, (size_t)nc * sizeof(float), hipMemcpyDeviceToHost, s));
        GH_CHECK(h, hipStreamSynchronize(s));
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_add(gamma_hip_index* h, int64_t n, const float* vecs, int64_t first_vid) {
    if (!h || n < 0 || (n > 0 && !vecs)) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    std::vector<int64_t> lno(n);
    std::vector<uint8_t> codes;
    {
        std::lock_guard<std::mutex> g(h->mu);
        if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "not trained");
        codes.resize((size_t)n * h->code_size);
    }
    GH_TRY(gamma_hip_ivfpq_encode(h, n, vecs, lno.data(), codes.data()));
    // group by list in ascending list order (std::map in gamma_index_ivfpq.cc:428-494)
    const int cs = h->code_size;
    std::vector<int64_t> order(n);
    for (int64_t i = 0; i < n; i++) {
        if (lno[i] < 0) lno[i] = (first_vid + i) % h->nlist;
        order[i] = i;
    }
    // list-sharded index: every shard is handed the same batch and keeps the vectors whose list it owns
    // (realtime inserts route to the owner of the assigned list) -- no exchange needed
    {
        std::lock_guard<std::mutex> g(h->mu);
        if (!h->h_list_mask.empty()) {
            int64_t m = 0;
            for (int64_t i = 0; i < n; i++)
                if (h->h_list_mask[lno[i]]) order[m++] = i;
            order.resize(m);
        }
    }
    const int64_t nkeep = (int64_t)order.size();
    if (nkeep == 0) return GAMMA_HIP_OK;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return lno[a] < lno[b]; });
    std::vector<int32_t> lists, counts;
    std::vector<int64_t> vids(nkeep);
    std::vector<uint8_t> gcodes((size_t)nkeep * cs);
    for (int64_t i = 0; i < nkeep; i++) {
        const int64_t src = order[i];
        vids[i] = first_vid + src;
        memcpy(gcodes.data() + (size_t)i * cs, codes.data() + (size_t)src * cs, cs);
        if (lists.empty() || lists.back() != (int32_t)lno[src]) {
            lists.push_back((int32_t)lno[src]);
            counts.push_back(0);
        }
        counts.back()++;
    }
    return gamma_hip_ivfpq_add_keys_batch(h, (int)lists.size(), lists.data(), counts.data(), vids.data(), gcodes.data());
}

/* ---- search ----------------------------------------------------------------------------- */
int gamma_hip_ivfpq_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                  const float* d_x, int k, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    return ivfpq_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
}

int gamma_hip_ivfflat_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* d_x,
                                    int k, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    return ivfflat_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
}

int gamma_hip_ivfflat_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                             float* distances, int64_t* labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(check_params(h, p, nq, k));
    if (!h->ivf_init || !h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfflat not initialised");
    if (nq > 0 && k > 0 && (!x || !distances || !labels)) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    return host_search(h, nq, h->d, x, k, distances, labels, [&](const float* dx, float* dd, int64_t* dl) {
        return ivfflat_search_device_locked(h, p, nq, dx, k, dd, dl);
    }, true, &lk);
}

static int flat_search_host_locked(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                                  float* distances, int64_t* labels) {
    SearchLock lk(h);
    GH_TRY(check_params(h, p, nq, k));
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (nq > 0 && k > 0 && (!x || !distances || !labels)) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    return host_search(h, nq, h->raw_d, x, k, distances, labels, [&](const float* dx, float* dd, int64_t* dl) {
        return flat_search_device_locked(h, p, nq, dx, k, dd, dl);
    }, true, &lk);
}

static int ivfpq_search_host_locked(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x,
                                   int k, float* distances, int64_t* labels) {
    SearchLock lk(h);
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (nq > 0 && k > 0 && (!x || !distances || !labels)) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    return host_search(h, nq, h->d, x, k, distances, labels, [&](const float* dx, float* dd, int64_t* dl) {
        return ivfpq_search_device_locked(h, p, nq, dx, k, dd, dl);
    }, true, &lk);
}

// Search is re-entrant in the reference and is called from many client threads at once, typically with
// one query each (tools/perf.cc).  One GPU stream serves one call at a time, so small calls
// that arrive while another is in flight are COMBINED: they queue, and a worker thread of the handle
// (the reference's GPU model funnels its searches through one thread as well) takes every queued
// request with the same parameters, runs them as one batch and hands the results back.  A call that
// finds the handle idle runs directly on the caller's thread.
// Results are those of the separate calls: rows are independent, and the coarse path (exact below 20
// queries, GEMM form from 20 on, faiss:utils/distances.cpp:346) is the one each request's OWN size
// selects -- requests only share a batch with requests that resolve to the same path.
constexpr int COMB_MAX_NQ = 256, COMB_MAX_TOTAL = 4096;

// filter table of a combined batch (h->mu held): entry i = request i's clauses + the delete bitmap
static int build_group_filters(gamma_hip_index* h, const std::vector<gamma_hip_index::Waiter*>& grp, int total,
                               std::vector<gh::FilterDesc>& tab, std::vector<int>& qf, FiltCtx* fc) {
    GH_CHECK(h, hipSetDevice(h->device));
    size_t tot = 0;
    for (auto* g : grp)
        if (g->p->has_range)
            for (int i = 0; i < g->p->n_range; i++) tot += ((size_t)g->p->range[i].bitmap_bytes + 15) & ~(size_t)15;
    GH_CHECK(h, h->w_filter.ensure(std::max<size_t>(tot, 16)));
    tab.resize(grp.size());
    qf.resize(total);
    size_t off = 0;
    int at = 0;
    for (size_t i = 0; i < grp.size(); i++) {
        GH_TRY(build_filter(h, grp[i]->p, &tab[i], &off));
        for (int j = 0; j < grp[i]->nq; j++) qf[at++] = (int)i;
    }
    GH_CHECK(h, h->w_ftab.ensure(tab.size() * sizeof(gh::FilterDesc)));
    GH_CHECK(h, h->w_qfil.ensure(qf.size() * sizeof(int)));
    h->ftab_valid = false;   // entry 0 no longer holds a single call's descriptor
    GH_CHECK(h, hipMemcpyAsync(h->w_ftab.p, tab.data(), tab.size() * sizeof(gh::FilterDesc), hipMemcpyHostToDevice, h->stream));
    GH_CHECK(h, hipMemcpyAsync(h->w_qfil.p, qf.data(), qf.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    fc->d_tab = h->w_ftab.as<gh::FilterDesc>();
    fc->d_qf = h->w_qfil.as<int>();
    fc->any_clause = true;
    return GAMMA_HIP_OK;
}

static void combine_worker(gamma_hip_index* h) {
    using W = gamma_hip_index::Waiter;
    auto same = [](const W* a, const W* b) {
        return a->kind == b->kind && a->k == b->k && a->mode == b->mode && a->p->metric == b->p->metric &&
               a->p->nprobe == b->p->nprobe &&
               a->p->recall_num == b->p->recall_num && a->p->has_rank == b->p->has_rank &&
               a->p->min_score == b->p->min_score && a->p->max_score == b->p->max_score;
    };
    // a batch in flight: its requests, where its results land, whether the stream still has to be awaited
    struct Batch {
        std::vector<W*> grp;
        int rc = GAMMA_HIP_OK, total = 0, kk = 0;
        float* sd = nullptr;
        int64_t* si = nullptr;
        bool enqueued = false;
        int set = -1;                       // pinned staging set holding its inputs / results
        std::vector<gh::FilterDesc> ftab;   // host images of the uploads, alive until the batch is awaited
        std::vector<int> qf;
        std::vector<int> rcs;               // per-request codes when the batch had to be redone one by one
    };
    // Results -> callers: a helper thread copies them out of the pinned staging set and wakes the callers
    // (one futex wake per request costs the worker more than launching the next batch), so the worker only
    // forms, launches and awaits batches.  A staging set is reused once its batch has been delivered.
    std::mutex n_mu;
    std::condition_variable n_cv;
    std::deque<Batch> n_q;
    bool n_stop = false;
    std::atomic<bool> set_busy[2];
    set_busy[0] = false;
    set_busy[1] = false;
    std::thread notifier([&]() {
        std::unique_lock<std::mutex> nl(n_mu);
        for (;;) {
            n_cv.wait(nl, [&] { return n_stop || !n_q.empty(); });
            if (n_q.empty()) break;   // stop requested and nothing left
            Batch b = std::move(n_q.front());
            n_q.pop_front();
            nl.unlock();
            if (b.rc == GAMMA_HIP_OK && b.sd) {   // no lock needed for the copies: the callers are blocked
                size_t at = 0;
                for (W* g : b.grp) {
                    std::memcpy(g->D, b.sd + at * b.kk, (size_t)g->nq * b.kk * sizeof(float));
                    std::memcpy(g->I, b.si + at * b.kk, (size_t)g->nq * b.kk * sizeof(int64_t));
                    at += g->nq;
                }
            }
            if (b.set >= 0) set_busy[b.set].store(false, std::memory_order_release);
            {
                std::lock_guard<std::mutex> cl(h->comb_mu);
                for (size_t i = 0; i < b.grp.size(); i++) {
                    W* g = b.grp[i];
                    g->rc = b.rcs.empty() ? b.rc : b.rcs[i];
                    g->done = true;
                    g->cv.notify_one();
                }
            }
            nl.lock();
        }
    });
    auto post = [&](Batch&& b) {
        if (b.grp.empty()) return;
        {
            std::lock_guard<std::mutex> nl(n_mu);
            n_q.push_back(std::move(b));
        }
        n_cv.notify_one();
    };
    Batch cur;
    int set = 0;
    static const bool dbg = getenv("GAMMA_HIP_COMB_DBG") != nullptr;   // phase times of the worker, printed at exit
    double us_stage = 0, us_deliver = 0, us_sync = 0;
    long n_batches = 0, n_reqs = 0;
    std::unique_lock<std::mutex> lk(h->comb_mu);
    for (;;) {
        h->comb_wcv.wait(lk, [&] { return h->comb_stop || (!h->comb_busy && !h->comb_q.empty()); });
        if (h->comb_stop) break;
        h->comb_busy = true;
        // the handle stays busy until the queue is drained; delivery of batch N overlaps with forming and
        // launching batch N+1
        for (;;) {
            cur = Batch();
            const auto t_a = std::chrono::steady_clock::now();
            if (!h->comb_q.empty()) {   // one group: the oldest request and everything compatible with it
                W* first = h->comb_q.front();
                for (auto it = h->comb_q.begin(); it != h->comb_q.end();) {
                    if (same(first, *it) && (cur.grp.empty() || cur.total + (*it)->nq <= COMB_MAX_TOTAL)) {
                        cur.total += (*it)->nq;
                        cur.grp.push_back(*it);
                        it = h->comb_q.erase(it);
                    } else {
                        ++it;
                    }
                }
            }
            lk.unlock();
            if (!cur.grp.empty()) {
                W* first = cur.grp.front();
                gamma_hip_search_params pp = *first->p;
                pp.coarse_mode = first->mode;
                const bool flat = first->kind == 1;
                const int d = flat ? h->raw_d : h->d, kk = first->k, total = cur.total;
                cur.kk = kk;
                const size_t bx = (size_t)total * d * sizeof(float), bd = (size_t)total * kk * sizeof(float),
                             bi = (size_t)total * kk * sizeof(int64_t);
                const size_t off_i = (bx + 15) & ~(size_t)15, off_d = off_i + ((bi + 15) & ~(size_t)15),
                             need = off_d + bd;
                while (set_busy[set].load(std::memory_order_acquire)) std::this_thread::yield();   // its last batch is being delivered
                if (need > h->comb_pin_bytes[set]) {
                    if (h->comb_pin[set]) (void)hipHostFree(h->comb_pin[set]);
                    h->comb_pin[set] = nullptr;
                    h->comb_pin_bytes[set] = 0;
                    if (hipSetDevice(h->device) == hipSuccess &&
                        hipHostMalloc(&h->comb_pin[set], need * 2, hipHostMallocDefault) == hipSuccess)
                        h->comb_pin_bytes[set] = need * 2;
                    else
                        cur.rc = GAMMA_HIP_ENOMEM;
                }
                if (cur.rc == GAMMA_HIP_OK) {
                    char* base = static_cast<char*>(h->comb_pin[set]);
                    float* sx = reinterpret_cast<float*>(base);
                    cur.si = reinterpret_cast<int64_t*>(base + off_i);
                    cur.sd = reinterpret_cast<float*>(base + off_d);
                    size_t at = 0;
                    for (W* g : cur.grp) {
                        std::memcpy(sx + at * d, g->x, (size_t)g->nq * d * sizeof(float));
                        at += g->nq;
                    }
                    cur.set = set;
                    set_busy[set].store(true, std::memory_order_release);
                    h->search_mu.lock();   // held until the batch has been awaited (below)
                    h->mu.lock();          // while the batch reads the handle and is enqueued
                    cur.rc = flat ? check_params(h, &pp, total, kk) : ivfpq_check(h, &pp, total, kk);
                    // requests with their own filter clauses: one table entry per request, a query -> entry map
                    // (IVFPQ only: filtered flat requests are not combined)
                    FiltCtx fc;
                    bool any_filter = false;
                    for (W* g : cur.grp) any_filter |= g->p->has_range || g->p->n_field > 0 || g->p->n_term > 0;
                    const bool multi = !flat && any_filter && cur.grp.size() > 1;
                    if (cur.rc == GAMMA_HIP_OK && multi) cur.rc = build_group_filters(h, cur.grp, total, cur.ftab, cur.qf, &fc);
                    if (cur.rc == GAMMA_HIP_OK)
                        cur.rc = host_search(h, total, d, sx, kk, cur.sd, cur.si,
                                             [&](const float* dx, float* dd, int64_t* dl) {
                                                 if (flat) return flat_search_device_locked(h, &pp, total, dx, kk, dd, dl);
                                                 return ivfpq_search_device_locked(h, &pp, total, dx, kk, dd, dl,
                                                                                   multi ? &fc : nullptr);
                                             },
                                             /*sync=*/false);
                    h->mu.unlock();
                    cur.enqueued = true;
                }
                set ^= 1;
            }
            const auto t_b = std::chrono::steady_clock::now();
            const auto t_c = t_b;
            if (cur.enqueued) {
                if (hipStreamSynchronize(h->stream) != hipSuccess && cur.rc == GAMMA_HIP_OK) cur.rc = GAMMA_HIP_EDEVICE;
                h->search_mu.unlock();
                cur.enqueued = false;
                if (cur.rc != GAMMA_HIP_OK && cur.grp.size() > 1) {
                    // one request's parameters may be at fault (a filter on an unknown column, ...): every
                    // request gets the outcome of its own call
                    for (W* g : cur.grp) {
                        gamma_hip_search_params pg = *g->p;
                        pg.coarse_mode = g->mode;
                        cur.rcs.push_back(g->kind == 1 ? flat_search_host_locked(h, &pg, g->nq, g->x, g->k, g->D, g->I)
                                                       : ivfpq_search_host_locked(h, &pg, g->nq, g->x, g->k, g->D, g->I));
                    }
                    cur.sd = nullptr;   // results are already in the callers' buffers
                    cur.rc = GAMMA_HIP_OK;
                }
            }
            if (dbg) {
                const auto t_d = std::chrono::steady_clock::now();
                us_stage += std::chrono::duration<double, std::micro>(t_b - t_a).count();
                us_deliver += std::chrono::duration<double, std::micro>(t_c - t_b).count();
                us_sync += std::chrono::duration<double, std::micro>(t_d - t_c).count();
                n_batches++;
                n_reqs += (long)cur.grp.size();
            }
            post(std::move(cur));
            lk.lock();
            if (h->comb_q.empty()) break;
        }
        h->comb_busy = false;
    }
    lk.unlock();
    {
        std::lock_guard<std::mutex> nl(n_mu);
        n_stop = true;
    }
    n_cv.notify_one();
    notifier.join();
    if (dbg && n_batches)
        fprintf(stderr, "combine worker: %ld batches, %.1f requests each; per batch: group+stage+enqueue %.1f us, deliver previous %.1f us, "
                "wait for the GPU %.1f us\n", n_batches, (double)n_reqs / n_batches, us_stage / n_batches, us_deliver / n_batches,
                us_sync / n_batches);
}

static int combined_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                           float* distances, int64_t* labels, int kind = 0) {
    gamma_hip_index::Waiter w;
    w.p = p; w.nq = nq; w.k = k; w.x = x; w.D = distances; w.I = labels;
    w.kind = kind;
    w.mode = kind == 1 ? 0 : (p->coarse_mode < 0 ? (nq < 20 ? 0 : 1) : p->coarse_mode);
    std::unique_lock<std::mutex> lk(h->comb_mu);
    if (!h->comb_busy && h->comb_q.empty()) {   // idle handle: run on this thread, no hop
        h->comb_busy = true;
        lk.unlock();
        gamma_hip_search_params pp = *p;
        pp.coarse_mode = w.mode;
        const int rc = kind == 1 ? flat_search_host_locked(h, &pp, nq, x, k, distances, labels)
                                 : ivfpq_search_host_locked(h, &pp, nq, x, k, distances, labels);
        lk.lock();
        h->comb_busy = false;
        if (!h->comb_q.empty()) h->comb_wcv.notify_one();
        return rc;
    }
    if (!h->comb_thread.joinable()) h->comb_thread = std::thread(combine_worker, h);
    h->comb_q.push_back(&w);
    h->comb_wcv.notify_one();
    w.cv.wait(lk, [&] { return w.done; });
    return w.rc;
}

int gamma_hip_ivfpq_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x,
                           int k, float* distances, int64_t* labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    if (h->combine && p && nq > 0 && nq <= COMB_MAX_NQ && k > 0 && x && distances && labels && h->ivf_init && h->d > 0 &&
        (!p->has_range || (p->n_range >= 0 && p->n_range <= gh::kMaxRange && (p->n_range == 0 || p->range))) &&
        p->n_field >= 0 && p->n_field <= gh::kMaxField && (p->n_field == 0 || p->field) &&
        p->n_term >= 0 && p->n_term <= gh::kMaxTerm && (p->n_term == 0 || p->term))
        return combined_search(h, p, nq, x, k, distances, labels);
    return ivfpq_search_host_locked(h, p, nq, x, k, distances, labels);
}

int gamma_hip_ivfpq_last_stages(gamma_hip_index* h, float* coarse_dis, int64_t* coarse_idx,
                                float* recall_dis, int64_t* recall_ids) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    const int nq = h->last_nq, P = h->last_P, R = h->last_R;
    if (nq <= 0) return fail(h, GAMMA_HIP_EINVAL, "no previous search");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    if (coarse_dis) GH_CHECK(h, hipMemcpy(coarse_dis, h->w_coarse_dis.p, (size_t)nq * P * sizeof(float), hipMemcpyDeviceToHost));
    if (coarse_idx) {
        std::vector<int> tmp((size_t)nq * P);
        GH_CHECK(h, hipMemcpy(tmp.data(), h->w_probe.p, tmp.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); i++) coarse_idx[i] = tmp[i];
    }
    if (recall_dis) GH_CHECK(h, hipMemcpy(recall_dis, h->w_cand_dis.p, (size_t)nq * R * sizeof(float), hipMemcpyDeviceToHost));
    if (recall_ids) GH_CHECK(h, hipMemcpy(recall_ids, h->w_cand_ids.p, (size_t)nq * R * sizeof(int64_t), hipMemcpyDeviceToHost));
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_search_shard(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                 const float* d_x, int k, float* d_recall_dis, int64_t* d_recall_ids) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;
    if (!d_x || !d_recall_dis || !d_recall_ids) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    GH_CHECK(h, hipSetDevice(h->device));
    const int R = std::max(p->recall_num, k);
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt));
    FiltCtx fc;
    GH_TRY(filt_ctx_single(h, filt, &fc));
    gamma_hip_search_params pp = *p;
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;   // decided on the whole call, not per chunk
    p = &pp;
    const int chunk = query_chunk(h, nq, p->nprobe);
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nc = std::min(chunk, nq - q0);
        GH_TRY(ivfpq_stage_a(h, p, fc.at(q0), nc, d_x + (size_t)q0 * h->d, R, nullptr, nullptr, /*shard=*/true,
                             d_recall_dis + (size_t)q0 * R, d_recall_ids + (size_t)q0 * R));
        h->last_nq = nc;
    }
    h->last_P = p->nprobe;
    h->last_R = R;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_coarse_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                  const float* d_x, float* d_coarse_dis, int32_t* d_probe) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(ivfpq_check(h, p, nq, 1));
    if (nq == 0) return GAMMA_HIP_OK;
    if (!d_x || !d_coarse_dis || !d_probe) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    GH_CHECK(h, hipSetDevice(h->device));
    const int P = p->nprobe;
    gamma_hip_search_params pp = *p;   // the caller resolves -1 on the size of the whole batch; a slice
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;   // that arrives unresolved decides by itself
    p = &pp;
    const int chunk = coarse_chunk(h, nq);
    for (int q0 = 0; q0 < nq; q0 += chunk)
        GH_TRY(ivfpq_coarse(h, p, std::min(chunk, nq - q0), d_x + (size_t)q0 * h->d, d_coarse_dis + (size_t)q0 * P,
                            d_probe + (size_t)q0 * P));
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_search_shard_preassigned(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                             const float* d_x, const float* d_coarse_dis,
                                             const int32_t* d_probe, int k, float* d_recall_dis,
                                             int64_t* d_recall_ids) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;
    if (!d_coarse_dis || !d_probe) return fail(h, GAMMA_HIP_EINVAL, "null coarse assignment");
    if (!d_x || !d_recall_dis || !d_recall_ids) return fail(h, GAMMA_HIP_EINVAL, "null buffer");
    GH_CHECK(h, hipSetDevice(h->device));
    const int R = std::max(p->recall_num, k), P = p->nprobe;
    gh::FilterDesc filt;
    GH_TRY(build_filter(h, p, &filt));
    FiltCtx fc;
    GH_TRY(filt_ctx_single(h, filt, &fc));
    const int chunk = query_chunk(h, nq, P);
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nc = std::min(chunk, nq - q0);
        GH_TRY(ivfpq_stage_a(h, p, fc.at(q0), nc, d_x + (size_t)q0 * h->d, R, d_coarse_dis + (size_t)q0 * P,
                             d_probe + (size_t)q0 * P, /*shard=*/true, d_recall_dis + (size_t)q0 * R,
                             d_recall_ids + (size_t)q0 * R));
        h->last_nq = nc;
    }
    h->last_P = P;
    h->last_R = R;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_merge_rerank(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nq,
                                 const float* d_x, int k, const float* d_all_dis, const int64_t* d_all_ids,
                                 int q0, int nq_local, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_TRY(ivfpq_check(h, p, nq, k));
    if (nshards <= 0 || q0 < 0 || nq_local < 0 || q0 + nq_local > nq) return fail(h, GAMMA_HIP_EINVAL, "bad shard/query range");
    if (k <= 0 || nq_local == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const bool l2 = p->metric == GAMMA_HIP_METRIC_L2;
    const int R = std::max(p->recall_num, k);
    if ((int64_t)nshards * R > (int64_t)1 << 24) return fail(h, GAMMA_HIP_EINVAL, "too many candidates");
    hipStream_t s = h->stream;
    GH_CHECK(h, h->w_cand_dis.ensure((size_t)nq_local * R * sizeof(float)));
    GH_CHECK(h, h->w_cand_ids.ensure((size_t)nq_local * R * sizeof(int64_t)));
    {
        StageScope t(h, GAMMA_HIP_STAGE_SELECT);
        static const bool no_merge_kernel = getenv("GAMMA_HIP_NO_MERGE_KERNEL") != nullptr;
        if (no_merge_kernel ||
            !gh::launch_merge_shards(s, l2, d_all_dis, d_all_ids, nshards, nq, R, q0, nq_local,
                                     h->w_cand_dis.as<float>(), h->w_cand_ids.as<int64_t>())) {
            // general shapes: transpose to [nq][W * R], select, translate positions to ids
            GH_CHECK(h, h->w_m_dis.ensure((size_t)nq * nshards * R * sizeof(float)));
            GH_CHECK(h, h->w_m_ids.ensure((size_t)nq * nshards * R * sizeof(int64_t)));
            GH_CHECK(h, h->w_cand_pos.ensure((size_t)nq_local * R * sizeof(int)));
            gh::launch_gather_shards(s, d_all_dis, d_all_ids, nshards, nq, R, h->w_m_dis.as<float>(),
                                     h->w_m_ids.as<int64_t>(), l2 ? INFINITY : -INFINITY);
            const int64_t stride = (int64_t)nshards * R;
            gh::launch_select_topk(s, l2, h->w_m_dis.as<float>() + (size_t)q0 * stride, stride, nullptr,
                                   (int)stride, (int)stride, nq_local, R, h->w_cand_dis.as<float>(),
                                   h->w_cand_pos.as<int>());
            gh::launch_take_ids(s, h->w_cand_pos.as<int>(), h->w_m_ids.as<int64_t>() + (size_t)q0 * stride, stride,
                                nq_local, R, h->w_cand_ids.as<int64_t>());
        }
        GH_CHECK(h, hipGetLastError());
    }
    return ivfpq_stage_b(h, p, nq_local, d_x + (size_t)q0 * h->d, R, k, h->w_cand_dis.as<float>(),
                         h->w_cand_ids.as<int64_t>(), d_distances, d_labels);
}

int gamma_hip_flat_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                 const float* d_x, int k, float* d_distances, int64_t* d_labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    return flat_search_device_locked(h, p, nq, d_x, k, d_distances, d_labels);
}

int gamma_hip_flat_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x,
                          int k, float* distances, int64_t* labels) {
    if (!h) return GAMMA_HIP_EINVAL;
    // small unfiltered calls from concurrent client threads share device batches (see gamma_hip_ivfpq_search)
    if (h->combine && p && nq > 0 && nq <= COMB_MAX_NQ && k > 0 && x && distances && labels && h->raw_d > 0 &&
        !p->has_range && p->n_range == 0 && p->n_field == 0 && p->n_term == 0)
        return combined_search(h, p, nq, x, k, distances, labels, /*kind=*/1);
    return flat_search_host_locked(h, p, nq, x, k, distances, labels);
}

/* ---- accounting ----------------------------------------------------------------------------- */
int64_t gamma_hip_total_mem_bytes(gamma_hip_index* h) {
    if (!h) return -1;
    std::lock_guard<std::mutex> g(h->mu);
    int64_t b = 0;
    b += h->raw_cap * h->raw_d * (int64_t)sizeof(float);
    b += (int64_t)h->bitmap_cap_bytes;
    for (auto& kv : h->fields) b += kv.second.cap * (int64_t)field_elem_size(kv.second.dtype);
    for (auto& kv : h->terms) b += kv.second.cap_docs * 8 + kv.second.cap_tok * 4;
    if (h->ivf_init) {
        b += (int64_t)h->nlist * h->d * 4 + (int64_t)h->nlist * 4 + (int64_t)h->M * 256 * h->dsub * 4;
        b += (int64_t)h->nlist * h->M * 256 * 4;
        b += h->arena_cap * (h->code_size + (int64_t)sizeof(int64_t));
        b += (int64_t)h->nlist * 12;
    }
    return b;
}

int gamma_hip_profile_enable(gamma_hip_index* h, int on) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->profile = on != 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_profile_reset(gamma_hip_index* h) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(drain_events(h));
    for (int i = 0; i < GAMMA_HIP_NUM_STAGES; i++) {
        h->stage_ms[i] = 0;
        h->stage_n[i] = 0;
    }
    h->scan_pairs = 0;
    GH_CHECK(h, hipMemsetAsync(h->d_scan_codes, 0, sizeof(unsigned long long), h->stream));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    return GAMMA_HIP_OK;
}

int gamma_hip_profile_get(gamma_hip_index* h, int stage, double* total_ms, int64_t* launches) {
    if (!h || stage < 0 || stage >= GAMMA_HIP_NUM_STAGES) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(drain_events(h));
    if (total_ms) *total_ms = h->stage_ms[stage];
    if (launches) *launches = h->stage_n[stage];
    return GAMMA_HIP_OK;
}

int gamma_hip_profile_scan_bytes(gamma_hip_index* h, int64_t* bytes, int64_t* pairs) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    unsigned long long codes = 0;
    GH_CHECK(h, hipMemcpy(&codes, h->d_scan_codes, sizeof(codes), hipMemcpyDeviceToHost));
    if (bytes) *bytes = (int64_t)codes * h->code_size;
    if (pairs) *pairs = h->scan_pairs;
    return GAMMA_HIP_OK;
}

}  // extern "C"
