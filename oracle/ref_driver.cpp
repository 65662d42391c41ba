// Thin extern "C" driver around the REAL faiss 1.7.1 (built from the reference's
// vendored tarball by oracle/Makefile.ref).  This file is our own code: it contains no
// reference source, only calls into the library the reference's hot path calls into
// (index/impl/gamma_index_ivfpq.cc:560,701-890 -> faiss::IndexFlatL2::search,
// faiss::ProductQuantizer, faiss heap, fvec_*).  It is TEST INFRASTRUCTURE: it exists so
// that oracle/gamma_oracle.c (the CPU restatement) can be pinned bit-for-bit against the
// real arithmetic, and so tests/gen_golden.py can emit golden vectors.  Nothing in the
// product path may link or load it.
#include <faiss/Clustering.h>
#include <faiss/IndexFlat.h>
#include <faiss/IndexIVFPQ.h>
#include <faiss/impl/ProductQuantizer.h>
#include <faiss/index_io.h>
#include <faiss/invlists/InvertedLists.h>
#include <faiss/utils/Heap.h>
#include <faiss/utils/distances.h>
#include <faiss/utils/random.h>
#include <faiss/utils/utils.h>

#include <cstdint>
#include <cstring>
#include <vector>

using idx_t = faiss::Index::idx_t;

namespace {
struct RefIVFPQ {
    faiss::IndexFlatL2* quantizer;  // Gamma always uses IndexFlatL2 (gamma_index_ivfpq.cc:147)
    faiss::IndexIVFPQ* index;
};

template <class C>
void heap_stream(size_t k, size_t n, const float* vals, const int64_t* ids,
                 float* heap_vals, int64_t* heap_ids, float* sorted_vals,
                 int64_t* sorted_ids) {
    // KnnSearchResults::add semantics (gamma_index_ivfpq.h:363-369)
    faiss::heap_heapify<C>(k, heap_vals, heap_ids);
    for (size_t i = 0; i < n; i++) {
        if (C::cmp(heap_vals[0], vals[i])) {
            faiss::heap_replace_top<C>(k, heap_vals, heap_ids, vals[i], ids[i]);
        }
    }
    memcpy(sorted_vals, heap_vals, k * sizeof(float));
    memcpy(sorted_ids, heap_ids, k * sizeof(int64_t));
    faiss::heap_reorder<C>(k, sorted_vals, sorted_ids);
}

template <class C>
void heap_pop_push_stream(size_t k, size_t n, const float* vals,
                          const int64_t* ids, float* sorted_vals,
                          int64_t* sorted_ids) {
    // compute_dis / GammaFLATIndex::Search semantics
    // (gamma_index_ivfpq.cc:664-680, gamma_index_flat.cc:203-206)
    faiss::heap_heapify<C>(k, sorted_vals, sorted_ids);
    for (size_t i = 0; i < n; i++) {
        if (C::cmp(sorted_vals[0], vals[i])) {
            faiss::heap_pop<C>(k, sorted_vals, sorted_ids);
            faiss::heap_push<C>(k, sorted_vals, sorted_ids, vals[i], ids[i]);
        }
    }
    faiss::heap_reorder<C>(k, sorted_vals, sorted_ids);
}
}  // namespace

extern "C" {

/* ---------------- scalar primitives (faiss/utils/distances_simd.cpp) ------------- */
float ref_fvec_L2sqr(const float* x, const float* y, size_t d) {
    return faiss::fvec_L2sqr(x, y, d);
}
float ref_fvec_inner_product(const float* x, const float* y, size_t d) {
    return faiss::fvec_inner_product(x, y, d);
}
float ref_fvec_norm_L2sqr(const float* x, size_t d) {
    return faiss::fvec_norm_L2sqr(x, d);
}
void ref_fvec_inner_products_ny(float* dis, const float* x, const float* y,
                                size_t d, size_t ny) {
    faiss::fvec_inner_products_ny(dis, x, y, d, ny);
}
void ref_fvec_L2sqr_ny(float* dis, const float* x, const float* y, size_t d,
                       size_t ny) {
    faiss::fvec_L2sqr_ny(dis, x, y, d, ny);
}
void ref_fvec_madd(size_t n, const float* a, float bf, const float* b, float* c) {
    faiss::fvec_madd(n, a, bf, b, c);
}

void ref_set_blas_threshold(int t) { faiss::distance_compute_blas_threshold = t; }

/* ---------------- training: the real rand_perm and the real Clustering ---------------- */
void ref_rand_perm(int* perm, size_t n, int64_t seed) { faiss::rand_perm(perm, n, seed); }
/* faiss::Clustering over an IndexFlatL2, the way IndexIVF::train_q1 / ProductQuantizer::train run it; returns the
 * objective of the last iteration */
float ref_kmeans(int d, int64_t n, const float* x, int k, int niter, int64_t seed, float* centroids) {
    faiss::ClusteringParameters cp;
    cp.niter = niter;
    cp.seed = (int)seed;
    faiss::Clustering clus(d, k, cp);
    faiss::IndexFlatL2 index(d);
    clus.train(n, x, index);
    memcpy(centroids, clus.centroids.data(), sizeof(float) * (size_t)d * k);
    return clus.iteration_stats.empty() ? 0.f : clus.iteration_stats.back().obj;
}
int ref_get_blas_threshold() { return faiss::distance_compute_blas_threshold; }

/* ---------------- brute-force kNN as the coarse quantizer runs it ---------------- */
void ref_flat_l2_search(size_t d, size_t ny, const float* y, size_t nx,
                        const float* x, size_t k, float* D, int64_t* I) {
    faiss::IndexFlatL2 index(d);
    index.add(ny, y);
    index.search(nx, x, k, D, (idx_t*)I);
}
void ref_flat_ip_search(size_t d, size_t ny, const float* y, size_t nx,
                        const float* x, size_t k, float* D, int64_t* I) {
    faiss::IndexFlatIP index(d);
    index.add(ny, y);
    index.search(nx, x, k, D, (idx_t*)I);
}

/* ---------------- heap mechanics (faiss/utils/Heap.h) ---------------------------- */
void ref_heap_stream(int keep_smallest, size_t k, size_t n, const float* vals,
                     const int64_t* ids, float* heap_vals, int64_t* heap_ids,
                     float* sorted_vals, int64_t* sorted_ids) {
    if (keep_smallest)
        heap_stream<faiss::CMax<float, int64_t>>(k, n, vals, ids, heap_vals,
                                                 heap_ids, sorted_vals, sorted_ids);
    else
        heap_stream<faiss::CMin<float, int64_t>>(k, n, vals, ids, heap_vals,
                                                 heap_ids, sorted_vals, sorted_ids);
}
void ref_heap_pop_push_stream(int keep_smallest, size_t k, size_t n,
                              const float* vals, const int64_t* ids,
                              float* sorted_vals, int64_t* sorted_ids) {
    if (keep_smallest)
        heap_pop_push_stream<faiss::CMax<float, int64_t>>(k, n, vals, ids,
                                                          sorted_vals, sorted_ids);
    else
        heap_pop_push_stream<faiss::CMin<float, int64_t>>(k, n, vals, ids,
                                                          sorted_vals, sorted_ids);
}

/* ---------------- IVFPQ exactly as Gamma configures it --------------------------- */
void* ref_ivfpq_new(int d, int nlist, int M, int nbits, int metric_ip, int niter) {
    RefIVFPQ* r = new RefIVFPQ;
    r->quantizer = new faiss::IndexFlatL2(d);
    r->index = new faiss::IndexIVFPQ(
            r->quantizer, d, nlist, M, nbits,
            metric_ip ? faiss::METRIC_INNER_PRODUCT : faiss::METRIC_L2);
    // gamma_index_ivfpq.cc:172-185
    r->index->own_fields = false;
    r->index->quantizer_trains_alone = 0;
    r->index->cp.niter = niter;
    r->index->by_residual = true;
    r->index->use_precomputed_table = 0;
    r->index->scan_table_threshold = 0;
    r->index->do_polysemous_training = false;
    r->index->polysemous_ht = 0;
    return r;
}
void ref_ivfpq_free(void* h) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    delete r->index;
    delete r->quantizer;
    delete r;
}
void ref_ivfpq_train(void* h, int64_t n, const float* x) {
    ((RefIVFPQ*)h)->index->train(n, x);
}
void ref_ivfpq_add(void* h, int64_t n, const float* x) {
    ((RefIVFPQ*)h)->index->add(n, x);
}
// the library's process-wide limit above which no precomputed table is built and the index stays in table mode 0
// (faiss:IndexIVFPQ.cpp:379,441-449); lowered by the tests to reach that branch at a small shape
void ref_set_precomputed_table_max_bytes(size_t bytes) { faiss::precomputed_table_max_bytes = bytes; }
size_t ref_get_precomputed_table_max_bytes() { return faiss::precomputed_table_max_bytes; }
int ref_ivfpq_use_precomputed_table(void* h) {
    return ((RefIVFPQ*)h)->index->use_precomputed_table;
}
void ref_ivfpq_set_metric(void* h, int metric_ip) {
    ((RefIVFPQ*)h)->index->metric_type =
            metric_ip ? faiss::METRIC_INNER_PRODUCT : faiss::METRIC_L2;
}
// faiss's own "IwPQ" writer (faiss:impl/index_write.cpp): the byte layout Gamma's Dump mirrors
// (index/gamma_index_io.cc:16-192, index/impl/gamma_index_ivfpq.cc:958-992)
void ref_ivfpq_write_index(void* h, const char* path) {
    faiss::write_index(((RefIVFPQ*)h)->index, path);
}
void ref_ivfpq_set_nprobe(void* h, int nprobe) {
    ((RefIVFPQ*)h)->index->nprobe = nprobe;
}
void ref_ivfpq_get_coarse_centroids(void* h, float* out) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    memcpy(out, r->quantizer->xb.data(), r->quantizer->xb.size() * sizeof(float));
}
void ref_ivfpq_get_pq_centroids(void* h, float* out) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    memcpy(out, r->index->pq.centroids.data(),
           r->index->pq.centroids.size() * sizeof(float));
}
int64_t ref_ivfpq_precomputed_table_size(void* h) {
    return (int64_t)((RefIVFPQ*)h)->index->precomputed_table.size();
}
void ref_ivfpq_get_precomputed_table(void* h, float* out) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    memcpy(out, r->index->precomputed_table.data(),
           r->index->precomputed_table.size() * sizeof(float));
}
int64_t ref_ivfpq_list_size(void* h, int64_t l) {
    return (int64_t)((RefIVFPQ*)h)->index->invlists->list_size(l);
}
void ref_ivfpq_get_list(void* h, int64_t l, int64_t* ids, uint8_t* codes) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    size_t n = r->index->invlists->list_size(l);
    faiss::InvertedLists::ScopedIds sids(r->index->invlists, l);
    faiss::InvertedLists::ScopedCodes scodes(r->index->invlists, l);
    memcpy(ids, sids.get(), n * sizeof(int64_t));
    memcpy(codes, scodes.get(), n * r->index->code_size);
}
void ref_ivfpq_search(void* h, int64_t nq, const float* x, int64_t k, int nprobe,
                      float* D, int64_t* I) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    r->index->nprobe = nprobe;
    r->index->search(nq, x, k, D, (idx_t*)I);
}
// coarse stage alone, exactly the call Gamma makes (gamma_index_ivfpq.cc:560)
void ref_ivfpq_coarse(void* h, int64_t nq, const float* x, int nprobe,
                      float* coarse_dis, int64_t* idx) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    r->quantizer->search(nq, x, nprobe, coarse_dis, (idx_t*)idx);
}
void ref_ivfpq_search_preassigned(void* h, int64_t nq, const float* x, int64_t k,
                                  int nprobe, const int64_t* keys,
                                  const float* coarse_dis, float* D, int64_t* I) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    r->index->nprobe = nprobe;
    r->index->search_preassigned(nq, x, k, (const idx_t*)keys, coarse_dis, D,
                                 (idx_t*)I, false);
}
// Gamma's Add path arithmetic (gamma_index_ivfpq.cc:455-472): assign, residual, encode
void ref_ivfpq_encode(void* h, int64_t n, const float* x, int64_t* list_nos,
                      uint8_t* codes) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    int d = r->index->d;
    r->quantizer->assign(n, x, (idx_t*)list_nos);
    std::vector<float> residuals((size_t)n * d);
    for (int64_t i = 0; i < n; i++) {
        if (list_nos[i] < 0)
            memset(residuals.data() + i * d, 0, sizeof(float) * d);
        else
            r->quantizer->compute_residual(x + i * d, residuals.data() + i * d,
                                           list_nos[i]);
    }
    r->index->pq.compute_codes(residuals.data(), codes, n);
}
// The whole search the way GammaIVFPQIndex::Search runs it (index/impl/gamma_index_ivfpq.cc:514-566,
// 701-890), on the real library: coarse quantizer + list scan with recall_num results (faiss's own
// IndexIVFPQ::search: the same quantizer->search + search_preassigned pair, OpenMP over queries), then
// compute_dis with has_rank (:646-680): exact fvec_L2sqr / fvec_inner_product against the raw vectors and a
// k-heap fed with heap_pop + heap_push, heap_reorder.  Used as the timed CPU baseline of bench.py
// (cpu_baseline.kind = "reference"); the recall-stage candidates arrive sorted here instead of in heap-array
// order, which changes nothing about the work done.
void ref_ivfpq_search_rerank(void* h, int64_t nq, const float* x, int k, int recall_num, int nprobe,
                             const float* raw, float* D, int64_t* I) {
    RefIVFPQ* r = (RefIVFPQ*)h;
    const int d = r->index->d;
    const bool ip = r->index->metric_type == faiss::METRIC_INNER_PRODUCT;
    r->index->nprobe = nprobe;
    std::vector<float> rd((size_t)nq * recall_num);
    std::vector<idx_t> ri((size_t)nq * recall_num);
    r->index->search(nq, x, recall_num, rd.data(), ri.data());
#pragma omp parallel for schedule(dynamic)
    for (int64_t q = 0; q < nq; q++) {
        float* simi = D + q * k;
        idx_t* idxi = (idx_t*)I + q * k;
        const float* xi = x + q * d;
        if (ip) faiss::heap_heapify<faiss::CMin<float, idx_t>>(k, simi, idxi);
        else faiss::heap_heapify<faiss::CMax<float, idx_t>>(k, simi, idxi);
        for (int j = 0; j < recall_num; j++) {
            const idx_t id = ri[q * recall_num + j];
            if (id < 0) continue;
            if (ip) {
                const float dis = faiss::fvec_inner_product(xi, raw + id * d, d);
                if (faiss::CMin<float, idx_t>::cmp(simi[0], dis)) {
                    faiss::heap_pop<faiss::CMin<float, idx_t>>(k, simi, idxi);
                    faiss::heap_push<faiss::CMin<float, idx_t>>(k, simi, idxi, dis, id);
                }
            } else {
                const float dis = faiss::fvec_L2sqr(xi, raw + id * d, d);
                if (faiss::CMax<float, idx_t>::cmp(simi[0], dis)) {
                    faiss::heap_pop<faiss::CMax<float, idx_t>>(k, simi, idxi);
                    faiss::heap_push<faiss::CMax<float, idx_t>>(k, simi, idxi, dis, id);
                }
            }
        }
        if (ip) faiss::heap_reorder<faiss::CMin<float, idx_t>>(k, simi, idxi);
        else faiss::heap_reorder<faiss::CMax<float, idx_t>>(k, simi, idxi);
    }
}
void ref_ivfpq_inner_prod_table(void* h, const float* x, float* table) {
    ((RefIVFPQ*)h)->index->pq.compute_inner_prod_table(x, table);
}

}  // extern "C"
