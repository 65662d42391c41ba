// gamma_hip.cpp -- libgamma_hip.so, C ABI of include/gamma_hip.h: the handle's life cycle, the per-handle switches,
// memory accounting and stage profiling.  The writers (realtime lists, raw store, columns, bitmap) are in
// gamma_hip_store.cpp, the search pipelines and their entry points in gamma_hip_search.cpp, the handle itself in
// gamma_hip_internal.h.  No CPU fallback anywhere: an entry point runs the HIP kernels or returns an error.
#include "gamma_hip_internal.h"

using namespace ghi;

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
        case GAMMA_HIP_EUNSUPPORTED: return "exact ties requested beyond the replay's range";
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
        hipStreamCreateWithFlags(&h->wstream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_rfork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_rdone, hipEventDisableTiming) != hipSuccess) {
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
        hipMemset(h->d_tie_stats, 0, 3 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&h->d_bound_stat, 5 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(h->d_bound_stat, 0, 5 * sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc((void**)&h->pin_bound_stat, 4 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(h->stream);
        delete h;
        return GAMMA_HIP_EDEVICE;
    }
    memset(h->pin_bound_stat, 0, 4 * sizeof(unsigned long long));
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
    if (h->side) (void)hipStreamSynchronize(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (hipEvent_t& e : h->ev_call)
        if (e) (void)hipEventDestroy(e);
    for (auto& sl : h->hslot) {
        sl.x.release();
        sl.D.release();
        sl.I.release();
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (h->up_stream) (void)hipStreamDestroy(h->up_stream);
    for (auto& b : h->fbank) b.release();
    for (hipEvent_t& e : h->ev_bank)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_rfork) (void)hipEventDestroy(h->ev_rfork);
    if (h->ev_rdone) (void)hipEventDestroy(h->ev_rdone);
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
    if (h->raw_vmm) {   // the mapped store: unmap the chunks, release them, free the range
        h->raw_vm.release();
        h->d_raw = nullptr;
    }
    // (release() of a range that was never reserved does nothing; an arena that left virtual memory management after a
    //  failed repack read-back still owns its reserved ranges)
    h->vm_codes.release();
    h->vm_ids.release();
    h->vm_sums.release();
    h->alt_codes.release();
    h->alt_ids.release();
    h->alt_sums.release();
    for (auto& r : h->vm_retired) r.release();
    if (h->arena_vmm) {
        h->d_codes = nullptr;
        h->d_ids = nullptr;
        h->d_sums = nullptr;
    }
    if (h->d_raw_slot) (void)hipFree(h->d_raw_slot);
    void* ptrs[] = {h->d_list_rank, h->d_raw, h->d_bitmap, h->d_cc, h->d_cc_norms, h->d_pqc, h->d_T2, h->d_codes,
                    h->d_ids, h->d_list_mask, h->d_scan_codes, h->d_tie_stats, h->d_v2d, h->d_sums, h->d_t2max, h->d_bound_stat};
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
    if (h->bound_copy_ev) (void)hipEventDestroy(h->bound_copy_ev);
    if (h->pin_bound_stat) (void)hipHostFree(h->pin_bound_stat);
    if (h->pin_flat_over) (void)hipHostFree(h->pin_flat_over);
    DevBuf* bufs[] = {&h->w_mat, &h->w_coarse_dis, &h->w_probe, &h->w_xn, &h->w_st2, &h->w_pair_off,
                      &h->w_qtotal, &h->w_dist, &h->w_cand_dis, &h->w_cand_pos, &h->w_cand_ids,
                      &h->w_exact, &h->w_selv, &h->w_selp, &h->w_x, &h->w_outd, &h->w_outl, &h->w_stage, &h->w_shard_cut,
                      &h->w_filter, &h->w_m_dis, &h->w_m_ids, &h->w_part_v, &h->w_part_i, &h->w_assign,
                      &h->w_codes_tmp, &h->w_qperm, &h->w_qbins, &h->w_scnt, &h->w_sflag, &h->w_surv, &h->w_pair_base,
                      &h->w_pair_ip, &h->w_flat_cand, &h->w_flat_meta, &h->w_full_cdis,
                      &h->w_full_probe, &h->w_ftab, &h->w_qfil, &h->w_tieflag, &h->w_tcut, &h->w_tlist, &h->w_lm_units, &h->w_lm_cnt, &h->w_fbits, &h->w_cmp_codes, &h->w_cmp_ids, &h->w_cmp_len, &h->w_cmp_sums, &h->w_fD, &h->w_fI, &h->w_fx, &h->w_fslab, &h->w_flog, &h->w_mr_vals, &h->w_mr_ids, &h->w_mr_meta,
                      &h->we_mat, &h->we_cdis, &h->we_x, &h->we_assign, &h->we_codes, &h->we_stage};
    for (DevBuf* b : bufs) b->release();
    (void)hipStreamDestroy(h->stream);
    (void)hipStreamDestroy(h->wstream);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->side2) (void)hipStreamDestroy(h->side2);
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
    if (h->side) GH_CHECK(h, hipStreamSynchronize(h->side));
    if (h->side2) GH_CHECK(h, hipStreamSynchronize(h->side2));
    h->replay_pending = false;
    return GAMMA_HIP_OK;
}

int gamma_hip_set_deferred_replay(gamma_hip_index* h, int on) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    if (!on && h->replay_pending) {
        GH_CHECK(h, hipSetDevice(h->device));
        GH_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_rdone, 0));
        h->replay_pending = false;
    }
    h->defer_replay = on != 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_join(gamma_hip_index* h) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    if (h->replay_pending) {
        GH_CHECK(h, hipSetDevice(h->device));
        GH_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_rdone, 0));
        h->replay_pending = false;
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_dim(gamma_hip_index* h) { return (h && h->ivf_init) ? h->d : 0; }
int gamma_hip_ivfpq_nlist(gamma_hip_index* h) { return (h && h->ivf_init) ? h->nlist : 0; }
int gamma_hip_ivfpq_code_size(gamma_hip_index* h) { return (h && h->ivf_init) ? h->code_size : 0; }

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

int gamma_hip_tie_stats(gamma_hip_index* h, int64_t* out3, int reset) {
    if (!h || !out3) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    // (everything on the handle's own streams: a null-stream hipMemset is asynchronous and NOT ordered against the
    //  non-blocking search stream -- a reset could land after the next search's first increments)
    if (h->side) GH_CHECK(h, hipStreamSynchronize(h->side));
    if (h->side2) GH_CHECK(h, hipStreamSynchronize(h->side2));
    unsigned long long v[3];
    GH_CHECK(h, hipMemcpyAsync(v, h->d_tie_stats, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    if (reset) GH_CHECK(h, hipMemsetAsync(h->d_tie_stats, 0, sizeof(v), h->stream));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 3; i++) out3[i] = (int64_t)v[i];
    return GAMMA_HIP_OK;
}

int gamma_hip_ties_not_honoured(gamma_hip_index* h, int64_t* out_calls, int reset) {
    if (!h || !out_calls) return GAMMA_HIP_EINVAL;
    *out_calls = reset ? h->ties_unhonoured.exchange(0) : h->ties_unhonoured.load();
    return GAMMA_HIP_OK;
}

int gamma_hip_blas_form_not_restated(gamma_hip_index* h, int64_t* out_calls, int reset) {
    if (!h || !out_calls) return GAMMA_HIP_EINVAL;
    *out_calls = reset ? h->blas_unrestated.exchange(0) : h->blas_unrestated.load();
    return GAMMA_HIP_OK;
}

int gamma_hip_set_scan_bound_feedback(gamma_hip_index* h, int on) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->bound_feedback_off = on == 0;
    h->bound_off_calls = 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_scan_bound_stats(gamma_hip_index* h, int64_t* out4) {
    if (!h || !out4) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipStreamSynchronize(h->stream));
    unsigned long long v[5] = {0, 0, 0, 0, 0};
    GH_CHECK(h, hipMemcpy(v, h->d_bound_stat, sizeof(v), hipMemcpyDeviceToHost));
    const int slot = h->bound_epoch & 1;   // the counts of the current kind of call
    out4[0] = (int64_t)v[2 * slot];
    out4[1] = (int64_t)v[2 * slot + 1];
    out4[2] = h->bound_backoffs;
    out4[3] = (int64_t)v[4];
    return GAMMA_HIP_OK;
}

int gamma_hip_set_workspace_budget(gamma_hip_index* h, int64_t bytes) {
    if (!h || bytes <= 0) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->dist_budget_bytes = (size_t)bytes;
    return GAMMA_HIP_OK;
}

/* ---- accounting ----------------------------------------------------------------------------- */
int64_t gamma_hip_total_mem_bytes(gamma_hip_index* h) {
    if (!h) return -1;
    std::lock_guard<std::mutex> g(h->mu);
    int64_t b = 0;
    b += h->raw_cap * h->raw_d * (int64_t)sizeof(float) + h->raw_slot_cap * (int64_t)sizeof(int32_t);
    b += (int64_t)h->bitmap_cap_bytes;
    for (auto& kv : h->fields) b += kv.second.cap * (int64_t)field_elem_size(kv.second.dtype);
    for (auto& kv : h->terms) b += kv.second.cap_docs * 8 + kv.second.cap_tok * 4;
    if (h->ivf_init) {
        b += (int64_t)h->nlist * h->d * 4 + (int64_t)h->nlist * 4 + (int64_t)h->M * 256 * h->dsub * 4;
        if (h->d_T2) b += h->ivfflat ? 256 : (int64_t)h->nlist * h->M * 256 * 4;
        b += h->arena_cap * (h->code_size + (int64_t)sizeof(int64_t));
        if (h->d_sums) b += h->arena_cap * (int64_t)sizeof(float) + (int64_t)h->nlist * 4;
        b += (int64_t)h->nlist * 12;
        // the shadow arena of calls that run over lists compacted under their filter (kept between calls)
        b += (int64_t)(h->w_cmp_codes.cap + h->w_cmp_ids.cap + h->w_cmp_len.cap);
    }
    return b;
}

int gamma_hip_profile_enable(gamma_hip_index* h, int on) {
    if (!h) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    h->profile = on == 2 ? 2 : (on != 0 ? 1 : 0);
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