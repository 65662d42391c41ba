// gamma_hip_search.h -- what the search translation units of libgamma_hip.so share: gamma_hip_search.cpp (the search
// pipelines and their entry points) and gamma_hip_combine.cpp (the combining queue of small concurrent calls).
#pragma once
#include "gamma_hip_internal.h"

namespace ghi {

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

int build_filter(H* h, const gamma_hip_search_params* p, gh::FilterDesc* f, size_t* off_io = nullptr, int64_t est_codes = 0);
int filt_ctx_single(H* h, const gh::FilterDesc& f, FiltCtx* c);
int check_params(H* h, const gamma_hip_search_params* p, int nq, int k);
int ivfpq_check(H* h, const gamma_hip_search_params* p, int nq, int k);
// given != nullptr: the filter context of a combined batch (p's own filter clauses are ignored)
int ivfpq_search_device_locked(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int k, float* d_distances,
                               int64_t* d_labels, const FiltCtx* given = nullptr);
int flat_search_device_locked(H* h, const gamma_hip_search_params* p, int nq, const float* d_x, int k, float* d_distances,
                              int64_t* d_labels);
// a whole host-buffer call on the caller's thread (search lock, staging, wait)
int flat_search_host_locked(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                            float* distances, int64_t* labels);
int ivfpq_search_host_locked(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                             float* distances, int64_t* labels);
// small host-buffer calls that find the handle busy share device batches (gamma_hip_combine.cpp); kind 0 IVFPQ, 1 flat
constexpr int COMB_MAX_NQ = 256, COMB_MAX_TOTAL = 4096;
int combined_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k, float* distances,
                    int64_t* labels, int kind = 0);

// host-pointer wrapper shared by ivfpq / flat
// sync = false: everything is only enqueued (pinned host buffers); the caller synchronises the stream
// lk != nullptr: the caller's SearchLock; its mu is released once everything is enqueued, so writers go on while
// this call waits for the GPU (search_mu stays: the workspaces are in use)
// a deferred tie replay (gamma_hip_set_deferred_replay) is complete on the search stream
inline int replay_join(H* h) {
    if (h->replay_pending) {
        GH_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_rdone, 0));
        h->replay_pending = false;
    }
    return GAMMA_HIP_OK;
}

template <typename F>
int host_search(H* h, int nq, int d, const float* x, int k, float* distances, int64_t* labels, F&& f,
                bool sync = true, SearchLock* lk = nullptr, float* mapped_d = nullptr, int64_t* mapped_i = nullptr) {
    if (nq <= 0 || k <= 0) return f(nullptr, nullptr, nullptr);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(replay_join(h));
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
            h->dir_pin_dev = nullptr;
            if (hipHostGetDevicePointer(&h->dir_pin_dev, h->dir_pin, 0) != hipSuccess) h->dir_pin_dev = nullptr;
        }
        char* base = static_cast<char*>(h->dir_pin);
        std::memcpy(base, x, bx);
        GH_CHECK(h, hipMemcpyAsync(h->w_x.p, base, bx, hipMemcpyHostToDevice, h->stream));
        // results: the last kernel of the chain stores them straight into the staging area (pinned host memory is
        // mapped into the device's address space; a few KB of posted writes) -- no copy back at all
        static const bool no_map = getenv("GAMMA_HIP_NO_MAPPED_RESULTS") != nullptr;
        if (!no_map && h->dir_pin_dev) {
            char* db = static_cast<char*>(h->dir_pin_dev);
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
    if (mapped_d && mapped_i) {   // distances / labels are pinned and mapped (the combining queue's staging set): stored in place
        GH_TRY(f(h->w_x.as<float>(), mapped_d, mapped_i));
        if (lk) lk->enqueued();
        if (sync) GH_CHECK(h, hipStreamSynchronize(h->stream));
        return GAMMA_HIP_OK;
    }
    GH_TRY(f(h->w_x.as<float>(), h->w_outd.as<float>(), h->w_outl.as<int64_t>()));
    GH_CHECK(h, hipMemcpyAsync(distances, h->w_outd.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    GH_CHECK(h, hipMemcpyAsync(labels, h->w_outl.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    if (lk) lk->enqueued();
    if (sync) GH_CHECK(h, hipStreamSynchronize(h->stream));
    return GAMMA_HIP_OK;
}
}  // namespace ghi
