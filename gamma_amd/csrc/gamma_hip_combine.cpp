// gamma_hip_combine.cpp -- the combining queue of libgamma_hip.so: small host-buffer Search calls from concurrent client
// threads share device batches (include/gamma_hip.h, gamma_hip_ivfpq_search / gamma_hip_flat_search).  The pipelines the
// batches run through are gamma_hip_search.cpp's.
#include "gamma_hip_search.h"

namespace ghi {

// Search is re-entrant in the reference and is called from many client threads at once, typically with
// one query each (tools/perf.cc).  One GPU stream serves one call at a time, so small calls
// that arrive while another is in flight are COMBINED: they queue, and a worker thread of the handle
// (the reference's GPU model funnels its searches through one thread as well) takes every queued
// request with the same parameters, runs them as one batch and hands the results back.  A call that
// finds the handle idle runs directly on the caller's thread.
// Results are those of the separate calls: rows are independent, and the coarse path (exact below 20
// queries, GEMM form from 20 on, faiss:utils/distances.cpp:346) is the one each request's OWN size
// selects -- requests only share a batch with requests that resolve to the same path.

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

// wake one waiter of the combining queue.  Notified under ITS mutex: it cannot leave combined_search (and destroy the
// condition variable, which lives on its stack) before this thread is done with it.
static void comb_wake(gamma_hip_index::Waiter* w) {
    std::lock_guard<std::mutex> l(w->wm);
    w->done = true;
    w->cv.notify_one();
}

static void combine_worker(gamma_hip_index* h) {
    using W = gamma_hip_index::Waiter;
    auto same = [](const W* a, const W* b) {
        return a->kind == b->kind && a->k == b->k && a->mode == b->mode && a->p->metric == b->p->metric &&
               a->p->nprobe == b->p->nprobe &&
               a->p->recall_num == b->p->recall_num && a->p->has_rank == b->p->has_rank &&
               a->p->min_score == b->p->min_score && a->p->max_score == b->p->max_score &&
               a->p->exact_ties == b->p->exact_ties;
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
    constexpr int NSET = gamma_hip_index::NSET;
    std::atomic<bool> set_busy[NSET];
    for (int i = 0; i < NSET; i++) set_busy[i] = false;
    std::thread notifier([&]() {
        std::unique_lock<std::mutex> nl(n_mu);
        for (;;) {
            n_cv.wait(nl, [&] { return n_stop || !n_q.empty(); });
            if (n_q.empty()) break;   // stop requested and nothing left
            Batch b = std::move(n_q.front());
            n_q.pop_front();
            nl.unlock();
            // rows -> the callers' buffers (they are blocked; a batch of 128 is 15 KB), the staging set is free again --
            // NOT left to the callers: one of them descheduled for a time slice would hold its set, and with more
            // client threads than cores every set was soon held by a straggler -- then the waiters are linked and the
            // roots woken (comb_wake); the forest unfolds on the callers' own threads
            const size_t n = b.grp.size();
            constexpr size_t ROOTS = 16;   // woken by this thread (batches up to 16: all of them); 16 + 64 + 256 in two hops
            if (b.rc == GAMMA_HIP_OK && b.sd) {
                size_t at = 0;
                for (W* g : b.grp) {
                    std::memcpy(g->D, b.sd + at * b.kk, (size_t)g->nq * b.kk * sizeof(float));
                    std::memcpy(g->I, b.si + at * b.kk, (size_t)g->nq * b.kk * sizeof(int64_t));
                    at += g->nq;
                }
            }
            if (b.set >= 0) set_busy[b.set].store(false, std::memory_order_release);
            for (size_t i = 0; i < n; i++) {
                W* g = b.grp[i];
                g->rc = b.rcs.empty() ? b.rc : b.rcs[i];
                for (size_t j = 0; j < (size_t)W::FAN; j++) {   // waiter i wakes ROOTS + FAN i .. ROOTS + FAN i + FAN - 1
                    const size_t c = ROOTS + W::FAN * i + j;
                    g->child[j] = c < n ? b.grp[c] : nullptr;
                }
            }
            for (size_t i = 0; i < std::min<size_t>(n, ROOTS); i++) comb_wake(b.grp[i]);   // the roots, from here
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
    bool holding = false;          // this thread holds h->search_mu
    hipEvent_t done_ev[NSET] = {nullptr, nullptr, nullptr, nullptr};   // end of the batch staged in set i
    (void)hipSetDevice(h->device);
    for (auto& e : done_ev)
        // (blocking: the worker sleeps while the GPU runs its batch instead of spinning on a core for the whole busy
        //  period -- with 128 clients under the test box's 16-core quota 275 k -> 328 k queries/s sustained, the
        //  median latency no worse; GAMMA_HIP_COMB_SPIN=1 spins)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming | (getenv("GAMMA_HIP_COMB_SPIN") ? 0 : hipEventBlockingSync)) !=
            hipSuccess)
            e = nullptr;
    static const bool dbg = getenv("GAMMA_HIP_COMB_DBG") != nullptr;   // phase times of the worker, printed at exit
    double us_stage = 0, us_deliver = 0, us_sync = 0;
    long n_batches = 0, n_reqs = 0;
    std::unique_lock<std::mutex> lk(h->comb_mu);
    for (;;) {
        h->comb_wcv.wait(lk, [&] { return h->comb_stop || (!h->comb_busy && !h->comb_q.empty()); });
        if (h->comb_stop) break;
        h->comb_busy = true;
        // The handle stays busy until the queue is drained.  One batch at a time: formed, staged, enqueued, awaited
        // through its event, handed to the notifier.  GAMMA_HIP_COMB_PIPELINE=1 keeps TWO in flight (batch N+1 is
        // formed and enqueued while the GPU runs batch N; the stream orders them, so workspaces are reused safely and
        // results land in different staging sets; search_mu is then held across batches and given up at least every
        // 32).  Measured with closed-loop single-query clients (tools/plugin_clients.py): no gain -- 8 threads 67 k
        // against 67 k queries/s, 32 threads 190 k against 196 k, 128 threads 440 k against 470 k with a worse p99: the
        // batches get smaller by what the overlap saves, each still pays its fixed 20 us of enqueue and ~45 us of GPU.
        Batch prev;
        bool have_prev = false;
        int streak = 0;
        // await and deliver a batch; its per-request redo when the batch failed as a whole
        auto finish = [&](Batch& b) {
            if (b.enqueued) {
                if (b.rc == GAMMA_HIP_OK) {
                    if (hipEventSynchronize(done_ev[b.set]) != hipSuccess) b.rc = GAMMA_HIP_EDEVICE;
                } else if (hipStreamSynchronize(h->stream) != hipSuccess) {
                    b.rc = GAMMA_HIP_EDEVICE;
                }
                b.enqueued = false;
                if (b.rc != GAMMA_HIP_OK && b.grp.size() > 1) {
                    // one request's parameters may be at fault (a filter on an unknown column, ...): every request gets
                    // the outcome of its own call -- with the stream drained and the handle released
                    (void)hipStreamSynchronize(h->stream);
                    if (holding) {
                        h->search_mu.unlock();
                        holding = false;
                    }
                    for (W* g : b.grp) {
                        gamma_hip_search_params pg = *g->p;
                        pg.coarse_mode = g->mode;
                        b.rcs.push_back(g->kind == 1 ? flat_search_host_locked(h, &pg, g->nq, g->x, g->k, g->D, g->I)
                                                     : ivfpq_search_host_locked(h, &pg, g->nq, g->x, g->k, g->D, g->I));
                    }
                    b.sd = nullptr;   // results are already in the callers' buffers
                    b.rc = GAMMA_HIP_OK;
                }
            }
            if (dbg) {
                n_batches++;
                n_reqs += (long)b.grp.size();
            }
            post(std::move(b));
        };
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
                    h->comb_pin_dev[set] = nullptr;
                    if (hipSetDevice(h->device) == hipSuccess &&
                        hipHostMalloc(&h->comb_pin[set], need * 2, hipHostMallocDefault) == hipSuccess) {
                        h->comb_pin_bytes[set] = need * 2;
                        if (hipHostGetDevicePointer(&h->comb_pin_dev[set], h->comb_pin[set], 0) != hipSuccess)
                            h->comb_pin_dev[set] = nullptr;
                    } else {
                        cur.rc = GAMMA_HIP_ENOMEM;
                    }
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
                    static const bool no_map = getenv("GAMMA_HIP_NO_MAPPED_RESULTS") != nullptr;
                    const bool map_ok = !no_map && h->comb_pin_dev[set] != nullptr;
                    set_busy[set].store(true, std::memory_order_release);
                    if (!holding) {        // held while batches are in flight (the workspaces are in use), see below
                        h->search_mu.lock();
                        holding = true;
                    }
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
                                             /*sync=*/false, nullptr,
                                             map_ok ? reinterpret_cast<float*>(static_cast<char*>(h->comb_pin_dev[set]) + off_d) : nullptr,
                                             map_ok ? reinterpret_cast<int64_t*>(static_cast<char*>(h->comb_pin_dev[set]) + off_i) : nullptr);
                    if (cur.rc == GAMMA_HIP_OK && hipEventRecord(done_ev[set], h->stream) != hipSuccess) cur.rc = GAMMA_HIP_EDEVICE;
                    h->mu.unlock();
                    cur.enqueued = true;
                }
                set = (set + 1) % NSET;
            }
            const auto t_b = std::chrono::steady_clock::now();
            // the batch before this one: await, deliver.  A failed batch is finished before anything else goes on.
            if (have_prev) {
                finish(prev);
                have_prev = false;
            }
            const auto t_c = std::chrono::steady_clock::now();
            static const bool pipeline = getenv("GAMMA_HIP_COMB_PIPELINE") != nullptr;   // off: measured, see above
            if (pipeline && cur.enqueued && cur.rc == GAMMA_HIP_OK && ++streak < 32) {
                prev = std::move(cur);
                have_prev = true;
            } else if (!cur.grp.empty()) {
                finish(cur);
                streak = 32;
            }
            if (streak >= 32 && !have_prev) {   // nothing in flight: let others at the handle
                if (holding) {
                    h->search_mu.unlock();
                    holding = false;
                }
                streak = 0;
            }
            if (dbg) {
                us_stage += std::chrono::duration<double, std::micro>(t_b - t_a).count();
                us_sync += std::chrono::duration<double, std::micro>(t_c - t_b).count();
            }
            lk.lock();
            if (h->comb_q.empty()) {
                if (have_prev) {   // drain the pipeline; requests may arrive meanwhile
                    lk.unlock();
                    finish(prev);
                    have_prev = false;
                    lk.lock();
                }
                if (h->comb_q.empty()) break;
            }
        }
        if (holding) {
            h->search_mu.unlock();
            holding = false;
        }
        streak = 0;
        h->comb_busy = false;
    }
    lk.unlock();
    {
        std::lock_guard<std::mutex> nl(n_mu);
        n_stop = true;
    }
    n_cv.notify_one();
    notifier.join();
    for (auto& e : done_ev)
        if (e) (void)hipEventDestroy(e);
    if (dbg && n_batches)
        fprintf(stderr, "combine worker: %ld batches, %.1f requests each; per batch: group+stage+enqueue %.1f us (the batch before it on the GPU meanwhile), "
                "(unused %.1f) then waiting for that batch %.1f us\n", n_batches, (double)n_reqs / n_batches, us_stage / n_batches, us_deliver / n_batches,
                us_sync / n_batches);
}

int combined_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                    float* distances, int64_t* labels, int kind) {
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
    lk.unlock();
    {
        std::unique_lock<std::mutex> wl(w.wm);
        w.cv.wait(wl, [&] { return w.done; });
    }
    // (every field of w was written before done; the children are still blocked, their Waiters alive)
    for (gamma_hip_index::Waiter* c : w.child)
        if (c) comb_wake(c);
    return w.rc;
}

}  // namespace ghi
