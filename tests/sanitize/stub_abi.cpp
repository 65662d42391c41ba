// stub_abi.cpp -- TEST INFRASTRUCTURE (never shipped, never linked by the product): the subset of the C ABI of
// include/gamma_hip.h that the RetrievalModel plugins (gamma_amd/host) call, implemented on the CPU oracle
// (oracle/gamma_oracle.h), so that the threaded HOST code -- plugin mirror of the raw store, Add / Update / Delete beside
// concurrent Search, Dump -- can run under -fsanitize=thread and -fsanitize=address,undefined in this container, where
// no GPU exists.  One mutex per handle stands in for the library's own reader / writer discipline: what
// is under test here is the plugins' locking, not the library's.  Entry points the stress script does not reach return
// GAMMA_HIP_EUNSUPPORTED.
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gamma_hip.h"
#include "../../oracle/gamma_oracle.h"

struct gamma_hip_index {
    std::mutex mu;
    go_ivfpq* ix = nullptr;
    int d = 0, raw_d = 0, nlist = 0, M = 0, metric = 1;
    std::vector<float> raw;
    int64_t nraw = 0, nadded = 0;
    std::vector<uint8_t> bitmap;
    int64_t nbits = 0;
    bool trained = false;
    std::string err;
};
#ifndef GAMMA_STUB_REAL_GROUP
struct gamma_hip_group { int dummy; };
#endif

extern "C" {
int gamma_hip_create(int, gamma_hip_index** out) { *out = new gamma_hip_index(); return GAMMA_HIP_OK; }
int gamma_hip_destroy(gamma_hip_index* h) {
    if (!h) return GAMMA_HIP_EINVAL;
    if (h->ix) go_ivfpq_free(h->ix);
    delete h;
    return GAMMA_HIP_OK;
}
const char* gamma_hip_strerror(int code) { return code == 0 ? "ok" : "stub error"; }
const char* gamma_hip_last_error(gamma_hip_index* h) { return h ? h->err.c_str() : ""; }
int gamma_hip_set_exact_ties(gamma_hip_index*, int) { return GAMMA_HIP_OK; }
int gamma_hip_raw_init(gamma_hip_index* h, int d) {
    std::lock_guard<std::mutex> g(h->mu);
    h->raw_d = d;
    h->raw.reserve((size_t)d * 400000);   // the oracle borrows the pointer: no reallocation in the script's range
    return GAMMA_HIP_OK;
}
int gamma_hip_raw_write(gamma_hip_index* h, int64_t first, int64_t n, const float* v) {
    std::lock_guard<std::mutex> g(h->mu);
    if (first > h->nraw) return GAMMA_HIP_EINVAL;
    if ((size_t)(first + n) * h->raw_d > h->raw.capacity()) return GAMMA_HIP_ENOMEM;
    if ((size_t)(first + n) * h->raw_d > h->raw.size()) h->raw.resize((size_t)(first + n) * h->raw_d);
    memcpy(h->raw.data() + first * h->raw_d, v, sizeof(float) * (size_t)n * h->raw_d);
    if (first + n > h->nraw) h->nraw = first + n;
    if (h->ix) go_ivfpq_set_raw(h->ix, h->raw.data(), h->nraw);
    return GAMMA_HIP_OK;
}
int gamma_hip_raw_update(gamma_hip_index* h, int64_t vid, const float* v) {
    std::lock_guard<std::mutex> g(h->mu);
    if (vid < 0 || vid >= h->nraw) return GAMMA_HIP_EINVAL;
    memcpy(h->raw.data() + vid * h->raw_d, v, sizeof(float) * h->raw_d);
    return GAMMA_HIP_OK;
}
int gamma_hip_raw_update_batch(gamma_hip_index* h, int64_t n, const int64_t* vids, const float* v) {
    std::lock_guard<std::mutex> g(h->mu);
    for (int64_t i = 0; i < n; i++)
        if (vids[i] >= 0 && vids[i] < h->nraw) memcpy(h->raw.data() + vids[i] * h->raw_d, v + i * h->raw_d, sizeof(float) * h->raw_d);
    return GAMMA_HIP_OK;
}
int gamma_hip_bitmap_upload(gamma_hip_index* h, const uint8_t* bm, int64_t nbits) {
    std::lock_guard<std::mutex> g(h->mu);
    h->bitmap.assign(bm, bm + (nbits >> 3) + 1);
    h->bitmap.resize(std::max<size_t>(h->bitmap.size(), (size_t)1 << 20));
    h->nbits = std::max<int64_t>(nbits, h->nbits);
    return GAMMA_HIP_OK;
}
int gamma_hip_bitmap_set(gamma_hip_index* h, const int64_t* docids, int64_t n, int value) {
    std::lock_guard<std::mutex> g(h->mu);
    if (h->bitmap.empty()) h->bitmap.assign((size_t)1 << 20, 0);
    for (int64_t i = 0; i < n; i++) {
        if (docids[i] < 0 || (size_t)(docids[i] >> 3) >= h->bitmap.size()) return GAMMA_HIP_EINVAL;
        if (value) h->bitmap[docids[i] >> 3] |= (uint8_t)(1 << (docids[i] & 7));
        else h->bitmap[docids[i] >> 3] &= (uint8_t)~(1 << (docids[i] & 7));
        h->nbits = std::max<int64_t>(h->nbits, docids[i] + 1);
    }
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_init(gamma_hip_index* h, int d, int nlist, int M, int nbits, int metric, int bis, int bms) {
    std::lock_guard<std::mutex> g(h->mu);
    h->ix = go_ivfpq_new(d, nlist, M, nbits, metric, bis, bms);
    h->d = d; h->nlist = nlist; h->M = M; h->metric = metric;
    return h->ix ? GAMMA_HIP_OK : GAMMA_HIP_ENOMEM;
}
int gamma_hip_ivfpq_set_trained(gamma_hip_index* h, const float* cc, const float* pq, const float* t) {
    std::lock_guard<std::mutex> g(h->mu);
    go_ivfpq_set_trained(h->ix, cc, pq, t);
    h->trained = true;
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_use_precomputed_table(gamma_hip_index* h) {
    std::lock_guard<std::mutex> g(h->mu);
    // (the product decides at Init from nlist * M and the limit; so does this)
    return h->ix ? ((size_t)h->nlist * h->M * 1024 > go_get_precomputed_table_max_bytes() ? 0 : 1) : GAMMA_HIP_EINVAL;
}
int64_t gamma_hip_get_precomputed_table_max_bytes(void) { return (int64_t)go_get_precomputed_table_max_bytes(); }
int gamma_hip_set_precomputed_table_max_bytes(int64_t b) {
    if (b < 0) return GAMMA_HIP_EINVAL;
    go_set_precomputed_table_max_bytes((size_t)b);
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_train(gamma_hip_index*, int d, int64_t n, const float* x, int nlist, int M, float* cc, float* pq) {
    go_ivfpq_train(d, nlist, M, n, x, cc, pq);
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_add(gamma_hip_index* h, int64_t n, const float* v, int64_t first_vid) {
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->trained || first_vid != h->nadded) return GAMMA_HIP_EINVAL;   // the oracle numbers vids consecutively
    go_ivfpq_set_raw(h->ix, h->raw.data(), h->nraw);
    if (!go_ivfpq_add(h->ix, n, v)) return GAMMA_HIP_EFULL;
    h->nadded += n;
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_add_keys(gamma_hip_index* h, int l, int n, const int64_t* vids, const uint8_t* codes) {
    std::lock_guard<std::mutex> g(h->mu);
    return go_ivfpq_add_keys(h->ix, l, n, vids, codes) ? GAMMA_HIP_OK : GAMMA_HIP_EFULL;
}
int gamma_hip_ivfpq_update_batch(gamma_hip_index* h, int n, const int64_t* vids, const float* v) {
    std::lock_guard<std::mutex> g(h->mu);
    for (int i = 0; i < n; i++) go_ivfpq_update(h->ix, vids[i], v + (size_t)i * h->d);
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_delete(gamma_hip_index* h, const int64_t* vids, int n) {
    std::lock_guard<std::mutex> g(h->mu);
    go_ivfpq_delete(h->ix, vids, n, h->bitmap.empty() ? nullptr : h->bitmap.data());
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_compact_if_need(gamma_hip_index* h) {
    std::lock_guard<std::mutex> g(h->mu);
    go_ivfpq_compact_if_need(h->ix, h->bitmap.empty() ? nullptr : h->bitmap.data());
    return GAMMA_HIP_OK;
}
int64_t gamma_hip_ivfpq_list_size(gamma_hip_index* h, int l) {
    std::lock_guard<std::mutex> g(h->mu);
    return h->ix ? go_ivfpq_list_size(h->ix, l) : -1;
}
int gamma_hip_ivfpq_get_list(gamma_hip_index* h, int l, int64_t* vids, uint8_t* codes) {
    std::lock_guard<std::mutex> g(h->mu);
    go_ivfpq_get_list(h->ix, l, vids, codes);
    return GAMMA_HIP_OK;
}
static void make_ctx(gamma_hip_index* h, const gamma_hip_search_params* p, go_search_ctx* c, std::vector<go_range_filter>* rf) {
    memset(c, 0, sizeof(*c));
    c->docids_bitmap = h->bitmap.empty() ? nullptr : h->bitmap.data();
    c->docids_bitmap_bits = h->nbits;
    c->min_score = p->min_score;
    c->max_score = p->max_score;
    c->has_range = p->has_range;
    c->n_range = p->n_range;
    rf->resize(p->n_range > 0 ? p->n_range : 0);
    for (int i = 0; i < p->n_range; i++) {
        (*rf)[i].bitmap = p->range[i].bitmap;
        (*rf)[i].min_doc = p->range[i].min_doc;
        (*rf)[i].max_doc = p->range[i].max_doc;
        (*rf)[i].min_aligned = p->range[i].min_aligned;
        (*rf)[i].b_not_in = p->range[i].b_not_in;
    }
    c->range = rf->data();
}
int gamma_hip_ivfpq_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k, float* D, int64_t* I) {
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->trained) return GAMMA_HIP_ENOTTRAINED;
    if (p->n_field || p->n_term) return GAMMA_HIP_EUNSUPPORTED;
    go_search_ctx c;
    std::vector<go_range_filter> rf;
    make_ctx(h, p, &c, &rf);
    go_ivfpq_set_raw(h->ix, h->raw.data(), h->nraw);
    return go_ivfpq_search(h->ix, &c, p->metric, p->nprobe, p->recall_num, p->has_rank, p->coarse_mode, nq, x, k, D, I, nullptr, nullptr,
                           nullptr, nullptr)
                   ? GAMMA_HIP_EINVAL
                   : GAMMA_HIP_OK;
}
int gamma_hip_flat_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k, float* D, int64_t* I) {
    std::lock_guard<std::mutex> g(h->mu);
    if (p->n_field || p->n_term) return GAMMA_HIP_EUNSUPPORTED;
    go_search_ctx c;
    std::vector<go_range_filter> rf;
    make_ctx(h, p, &c, &rf);
    return go_flat_search(h->raw.data(), h->nraw, h->raw_d, &c, p->metric, nq, x, k, D, I) ? GAMMA_HIP_EINVAL : GAMMA_HIP_OK;
}
int64_t gamma_hip_total_mem_bytes(gamma_hip_index* h) {
    std::lock_guard<std::mutex> g(h->mu);
    return (int64_t)h->raw.size() * 4 + (int64_t)h->bitmap.size();
}
int gamma_hip_ties_not_honoured(gamma_hip_index*, int64_t* out, int) { *out = 0; return GAMMA_HIP_OK; }
int gamma_hip_blas_form_not_restated(gamma_hip_index*, int64_t* out, int) { *out = 0; return GAMMA_HIP_OK; }
int gamma_hip_ivfpq_repack_verify_stats(gamma_hip_index*, int64_t* out2) { out2[0] = out2[1] = 0; return GAMMA_HIP_OK; }
int gamma_hip_profile_enable(gamma_hip_index*, int) { return GAMMA_HIP_OK; }
int gamma_hip_profile_get(gamma_hip_index*, int, double* ms, int64_t* n) { *ms = 0; *n = 0; return GAMMA_HIP_OK; }
int64_t gamma_hip_vid2docid_count(gamma_hip_index*) { return 0; }
/* ---- not reached by the stress script ---- */
int gamma_hip_vid2docid_append(gamma_hip_index*, int64_t, const int32_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfflat_init(gamma_hip_index*, int, int, int, int, int) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfflat_set_trained(gamma_hip_index*, const float*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfflat_search(gamma_hip_index*, const gamma_hip_search_params*, int, const float*, int, float*, int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_term_update(gamma_hip_index*, int, int64_t, int32_t, const int32_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_term_append(gamma_hip_index*, int, int64_t, const int32_t*, const int32_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_field_update(gamma_hip_index*, int, int64_t, const void*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_field_append(gamma_hip_index*, int, int, int64_t, const void*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_kmeans(gamma_hip_index*, int, int64_t, const float*, int, int, int64_t, int, float*, float*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_assign(gamma_hip_index*, int d, int64_t n, const float* x, int k, const float* centroids, int32_t* assign, float* dis) {
    std::vector<float> D((size_t)n);
    std::vector<int64_t> I((size_t)n);
    go_knn_L2sqr(n < 20 ? 0 : 1, x, centroids, (size_t)d, (size_t)n, (size_t)k, 1, D.data(), I.data());
    for (int64_t i = 0; i < n; i++) {
        assign[i] = (int32_t)I[i];
        if (dis) dis[i] = D[i];
    }
    return GAMMA_HIP_OK;
}
#ifndef GAMMA_STUB_REAL_GROUP   /* the group build links the REAL gamma_hip_group.cpp instead (Makefile: stress_group_tsan) */
int gamma_hip_group_create(const int*, int, gamma_hip_group**) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_destroy(gamma_hip_group*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_size(const gamma_hip_group*) { return 0; }
gamma_hip_index* gamma_hip_group_member(gamma_hip_group*, int) { return nullptr; }
const char* gamma_hip_group_last_error(gamma_hip_group*) { return ""; }
int gamma_hip_group_set_owners(gamma_hip_group*, const int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_set_placement(gamma_hip_group*, int) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_owner(const gamma_hip_group*, int) { return -1; }
int64_t gamma_hip_group_total_mem_bytes(gamma_hip_group*) { return 0; }
int gamma_hip_group_ivfpq_update(gamma_hip_group*, int, const int64_t*, const float*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_ivfpq_search(gamma_hip_group*, const gamma_hip_search_params*, int, const float*, int, float*, int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int64_t gamma_hip_group_ivfpq_list_size(gamma_hip_group*, int) { return -1; }
int gamma_hip_group_ivfpq_get_list(gamma_hip_group*, int, int64_t*, uint8_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_ivfpq_delete(gamma_hip_group*, const int64_t*, int) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_ivfpq_compact_if_need(gamma_hip_group*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_ivfpq_add_keys(gamma_hip_group*, int, int, const int64_t*, const uint8_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_group_ivfpq_add(gamma_hip_group*, int64_t, const float*, int64_t) { return GAMMA_HIP_EUNSUPPORTED; }
#endif

/* ---- what gamma_hip_group.cpp calls on its members (replicate placement: every member holds every list) ---- */
void* gamma_hip_stream(gamma_hip_index*) { return nullptr; }
int gamma_hip_synchronize(gamma_hip_index*) { return GAMMA_HIP_OK; }
int gamma_hip_ivfpq_dim(gamma_hip_index* h) { return h ? h->d : -1; }
int gamma_hip_ivfpq_nlist(gamma_hip_index* h) { return h ? h->nlist : -1; }
int gamma_hip_ivfpq_code_size(gamma_hip_index* h) { return h ? h->M : -1; }
int gamma_hip_ivfpq_set_list_mask(gamma_hip_index*, const uint8_t*) { return GAMMA_HIP_OK; }
int gamma_hip_ivfpq_encode(gamma_hip_index* h, int64_t n, const float* v, int64_t* lno, uint8_t* codes) {
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->trained) return GAMMA_HIP_ENOTTRAINED;
    go_ivfpq_encode(h->ix, n, v, lno, codes);
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_encode_each(gamma_hip_index* h, int64_t n, const float* v, int64_t* lno, uint8_t* codes) {
    return gamma_hip_ivfpq_encode(h, n, v, lno, codes);
}
int gamma_hip_ivfpq_add_keys_batch(gamma_hip_index* h, int nl, const int32_t* lists, const int32_t* counts, const int64_t* vids,
                                   const uint8_t* codes) {
    std::lock_guard<std::mutex> g(h->mu);
    size_t at = 0;
    for (int i = 0; i < nl; i++) {
        if (!go_ivfpq_add_keys(h->ix, lists[i], counts[i], vids + at, codes + at * h->M)) return GAMMA_HIP_EFULL;
        at += counts[i];
    }
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_has_vid(gamma_hip_index* h, const int64_t* vids, int n, uint8_t* out) {
    std::lock_guard<std::mutex> g(h->mu);
    for (int i = 0; i < n; i++) out[i] = go_ivfpq_has_vid(h->ix, vids[i]) ? 1 : 0;
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_apply_updates(gamma_hip_index* h, int n, const int32_t* lists, const int64_t* vids, const uint8_t* codes,
                                  const uint8_t* ops) {
    std::lock_guard<std::mutex> g(h->mu);
    for (int i = 0; i < n; i++) {
        const int op = ops ? ops[i] : 0;
        if (op == 2) go_ivfpq_remove(h->ix, vids[i]);
        else if (op == 1) go_ivfpq_add_keys(h->ix, lists[i], 1, vids + i, codes + (size_t)i * h->M);
        else go_ivfpq_update_code(h->ix, lists[i], vids[i], codes + (size_t)i * h->M);
    }
    return GAMMA_HIP_OK;
}
int gamma_hip_ivfpq_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k, float* D, int64_t* I) {
    return gamma_hip_ivfpq_search(h, p, nq, x, k, D, I);   // "device" memory is host memory here (fakehip)
}
int gamma_hip_ivfpq_search_device_wait(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k, float* D, int64_t* I) {
    return gamma_hip_ivfpq_search(h, p, nq, x, k, D, I);
}
int gamma_hip_flat_search_device_wait(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k, float* D, int64_t* I) {
    return gamma_hip_flat_search(h, p, nq, x, k, D, I);
}
/* list-shard entry points: not reached in replicate placement */
int gamma_hip_bound_combine(void*, float*, const float*, int, int) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_raw_put(gamma_hip_index*, int64_t, const int64_t*, const float*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_shard_exact(gamma_hip_index*, const gamma_hip_search_params*, int, const float*, const int64_t*, int, float*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_merge_rerank_exact(gamma_hip_index*, const gamma_hip_search_params*, int, int, const float*, int, const float*, const int64_t*,
                                       const float*, int, int, float*, int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_shard_export_exact(gamma_hip_index*, const gamma_hip_search_params*, int, const float*, const float*, const int64_t*,
                                       const int32_t*, int64_t, const float*, float*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_merge_replay_exact(gamma_hip_index*, const gamma_hip_search_params*, int, int, const float*, int64_t, const float*,
                                       const int64_t*, const int32_t*, const float*, int, const int32_t*, float*, int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_search_shard_bounded(gamma_hip_index*, const gamma_hip_search_params*, int, const float*, const float*, const int32_t*,
                                         int, float*, int64_t*, float*, gamma_hip_bound_reduce_fn, void*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_gather_rows(gamma_hip_index*, const void*, int, const int32_t*, int, void*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_coarse_device(gamma_hip_index*, const gamma_hip_search_params*, int, const float*, float*, int32_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_search_shard_preassigned(gamma_hip_index*, const gamma_hip_search_params*, int, const float*, const float*, const int32_t*,
                                             int, float*, int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_merge_rerank(gamma_hip_index*, const gamma_hip_search_params*, int, int, const float*, int, const float*, const int64_t*,
                                 int, int, float*, int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_shard_cut_flags(gamma_hip_index*, int, uint8_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_merge_set_shard_flags(gamma_hip_index*, const uint8_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_merge_flagged(gamma_hip_index*, int*, const int32_t**) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_shard_export_rows(gamma_hip_index*, const gamma_hip_search_params*, int, const int32_t*, int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_shard_export(gamma_hip_index*, const gamma_hip_search_params*, int, const float*, const float*, const int32_t*, int64_t,
                                 float*, int64_t*, int32_t*) { return GAMMA_HIP_EUNSUPPORTED; }
int gamma_hip_ivfpq_merge_replay(gamma_hip_index*, const gamma_hip_search_params*, int, int, const float*, int64_t, const float*, const int64_t*,
                                 const int32_t*, int, const int32_t*, float*, int64_t*) { return GAMMA_HIP_EUNSUPPORTED; }
}
