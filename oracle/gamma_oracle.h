/*
 * gamma_oracle.h -- CPU restatement ("oracle") of the vearch/gamma retrieval hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call anything declared here; the product path
 * (gamma_amd/, libgamma_hip.so) never does and fails loudly without its HIP library.
 *
 * What it restates (paths relative to /root/reference; "faiss:" = inside
 * third_party/faiss-1.7.1.tar.gz, faiss-1.7.1/faiss/):
 *   - fvec_* SIMD distance primitives incl. their FMA-contracted summation order
 *     (faiss:utils/distances_simd.cpp:159-437,709-750, built -mavx2 -mfma)
 *   - binary heap top-k (faiss:utils/Heap.h:46-131,195-217,246-330,
 *     faiss:utils/ordered_key_value.h:42-75)
 *   - coarse quantizer IndexFlatL2::search -> knn_L2sqr, sequential and "BLAS" forms
 *     (faiss:utils/distances.cpp:130-155,215-296,334-360)
 *   - PQ tables / precomputed table / encoding
 *     (faiss:impl/ProductQuantizer.cpp:321-348,472-531, faiss:IndexIVFPQ.cpp:412-511)
 *   - GammaIVFPQIndex::Search / search_preassigned / scan / compute_dis
 *     (index/impl/gamma_index_ivfpq.cc:514-566,597-697,701-890,
 *      index/impl/gamma_index_ivfpq.h:148-168,184-257,363-369,575-601)
 *   - GammaFLATIndex::Search (index/impl/gamma_index_flat.cc:118-300)
 *   - realtime inverted lists (realtime/realtime_mem_data.cc:57-518)
 *   - validity predicates (common/gamma_common_data.h:95-108,
 *     table/range_query_result.h:53-67,169-179, util/bitmap.cc:25-31)
 *
 * Pinning: every primitive and the IVFPQ end-to-end search are checked bit-for-bit
 * against the real faiss 1.7.1 built from the reference's tarball (oracle/_ref, see
 * oracle/Makefile.ref and tests/test_oracle_vs_ref.py) and against golden vectors
 * generated from it (tests/golden/).  Gamma's own glue on top of faiss (filters,
 * re-rank, -1 padding) cannot be executed here (its sources need TBB and
 * flatbuffers-generated headers the image lacks) and is restated from source only.
 */
#ifndef GAMMA_ORACLE_H_
#define GAMMA_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO_METRIC_IP 0 /* DistanceComputeType::INNER_PRODUCT (retrieval_model.h:20) */
#define GO_METRIC_L2 1

/* ---- primitives ---------------------------------------------------------------- */
float go_fvec_L2sqr(const float* x, const float* y, size_t d);
float go_fvec_inner_product(const float* x, const float* y, size_t d);
float go_fvec_norm_L2sqr(const float* x, size_t d);
void go_fvec_inner_products_ny(float* dis, const float* x, const float* y, size_t d, size_t ny);
void go_fvec_L2sqr_ny(float* dis, const float* x, const float* y, size_t d, size_t ny);
void go_fvec_madd(size_t n, const float* a, float bf, const float* b, float* c);

/* ---- heap ---------------------------------------------------------------------- */
/* keep_smallest=1 -> CMax heap (L2), 0 -> CMin heap (IP) */
void go_heap_heapify(int keep_smallest, size_t k, float* val, int64_t* ids);
void go_heap_replace_top(int keep_smallest, size_t k, float* val, int64_t* ids, float v, int64_t id);
void go_heap_pop(int keep_smallest, size_t k, float* val, int64_t* ids);
void go_heap_push(int keep_smallest, size_t k, float* val, int64_t* ids, float v, int64_t id);
size_t go_heap_reorder(int keep_smallest, size_t k, float* val, int64_t* ids);
/* stream helpers used by the pinning tests (mirror oracle/ref_driver.cpp) */
void go_heap_stream(int keep_smallest, size_t k, size_t n, const float* vals, const int64_t* ids,
                    float* heap_vals, int64_t* heap_ids, float* sorted_vals, int64_t* sorted_ids);
void go_heap_pop_push_stream(int keep_smallest, size_t k, size_t n, const float* vals,
                             const int64_t* ids, float* sorted_vals, int64_t* sorted_ids);

/* ReservoirTopN (faiss:impl/ResultHandler.h:131-187; shrink = partition_fuzzy_median3, faiss:utils/partitioning.cpp:
 * 119-215): what knn_L2sqr / knn_inner_product collect through from k = 100 on.  A stream of (value, id; ids == NULL:
 * the position) through a reservoir of capacity go_reservoir_capacity(k), then to_result: k sorted entries. */
size_t go_reservoir_capacity(size_t k);
void go_reservoir_stream(int keep_smallest, size_t k, size_t n, const float* vals, const int64_t* ids,
                         float* sorted_vals, int64_t* sorted_ids);

/* ---- brute-force kNN (coarse quantizer / flat) ---------------------------------- */
/* mode 0: sequential fvec_L2sqr per pair (faiss nx<20 path); mode 1: norms + k-ordered
 * fmaf inner product (the "BLAS" form, with the GEMM restated as a k-sequential fma chain,
 * which is exactly what the MI355X fp32 MFMA computes) */
void go_knn_L2sqr(int mode, const float* x, const float* y, size_t d, size_t nx, size_t ny,
                  size_t k, float* D, int64_t* I);
void go_knn_inner_product(const float* x, const float* y, size_t d, size_t nx, size_t ny,
                          size_t k, float* D, int64_t* I);

/* ---- product quantizer ---------------------------------------------------------- */
void go_pq_inner_prod_table(const float* pq_centroids, int M, int ksub, int dsub,
                            const float* x, float* table);
void go_pq_compute_codes(const float* pq_centroids, int M, int ksub, int dsub,
                         const float* x, uint8_t* codes, size_t n);
void go_ivfpq_precompute_table(const float* coarse_centroids, int nlist, int d,
                               const float* pq_centroids, int M, int ksub, float* table);

/* ---- filter description (what IsValid() needs) ---------------------------------- */
typedef struct {
    const uint8_t* bitmap; /* bit i <-> byte i>>3, mask 1<<(i&7); relative to min_aligned */
    int32_t min_doc, max_doc, min_aligned;
    int32_t b_not_in;
} go_range_filter;

typedef struct {
    const uint8_t* docids_bitmap; /* delete bitmap, may be NULL */
    int64_t docids_bitmap_bits;
    int32_t has_range;          /* range_query_result != nullptr */
    int32_t n_range;            /* number of RangeQueryResult (0 => Has() is false) */
    const go_range_filter* range;
    float min_score, max_score;
    /* VIDMgr::VID2DocID (vector/raw_vector_common.h:90-95): NULL = single-vector documents, docid == vid */
    const int32_t* vid2docid;
    int64_t n_vid2docid;
} go_search_ctx;

/* ---- index objects --------------------------------------------------------------- */
typedef struct go_ivfpq go_ivfpq;

go_ivfpq* go_ivfpq_new(int d, int nlist, int M, int nbits, int metric,
                       int bucket_init_size, int bucket_max_size);
void go_ivfpq_free(go_ivfpq* ix);
/* trained state: copies the arrays; computes the precomputed table when `table` is NULL */
void go_ivfpq_set_trained(go_ivfpq* ix, const float* coarse_centroids, const float* pq_centroids,
                          const float* table);
const float* go_ivfpq_table(go_ivfpq* ix); /* NULL in table mode 0 */
/* faiss::precomputed_table_max_bytes (faiss:IndexIVFPQ.cpp:379): above it set_trained builds no table and the
 * search scores with per-(query, list) residual tables (index/impl/gamma_index_ivfpq.h:239-245) */
void go_set_precomputed_table_max_bytes(size_t bytes);
size_t go_get_precomputed_table_max_bytes(void);
int go_ivfpq_use_precomputed_table(const go_ivfpq* ix);
/* delete bitmap the list writer consults (AddKeys, CompactBucket); borrowed pointer */
void go_ivfpq_set_docids_bitmap(go_ivfpq* ix, const uint8_t* bm, int64_t nbits);
/* vid -> docid of multi-vector documents for the list writer's delete tests (realtime_mem_data.cc:102,294); borrowed */
void go_ivfpq_set_vid2docid(go_ivfpq* ix, const int32_t* map, int64_t n);
/* raw vector store for re-rank (VectorReader::Gets): pointer is borrowed */
void go_ivfpq_set_raw(go_ivfpq* ix, const float* raw, int64_t n);
/* Add path (gamma_index_ivfpq.cc:424-512): assign + residual + encode + AddKeys; vids are
 * consecutive from indexed_vec_count_.  Returns 1 on success (bool Add). */
int go_ivfpq_add(go_ivfpq* ix, int64_t n, const float* x);
/* assign mode used by Add/encode: -1 faiss rule (n<20 sequential, else BLAS form), 0, 1 */
void go_set_assign_mode(int mode);
/* encode only: list numbers + codes */
void go_ivfpq_encode(go_ivfpq* ix, int64_t n, const float* x, int64_t* list_nos, uint8_t* codes);
/* realtime list writer side */
int go_ivfpq_add_keys(go_ivfpq* ix, int list_no, int n, const int64_t* keys, const uint8_t* codes);
int go_ivfpq_update(go_ivfpq* ix, int64_t vid, const float* x);
int go_ivfpq_update_code(go_ivfpq* ix, int list_no, int64_t vid, const uint8_t* code);
int go_ivfpq_has_vid(go_ivfpq* ix, int64_t vid);
int go_ivfpq_remove(go_ivfpq* ix, int64_t vid);
int go_ivfpq_delete(go_ivfpq* ix, const int64_t* vids, int n, const uint8_t* docids_bitmap);
int go_ivfpq_compact_if_need(go_ivfpq* ix, const uint8_t* docids_bitmap);
int64_t go_ivfpq_list_size(go_ivfpq* ix, int list_no);
int64_t go_ivfpq_list_capacity(go_ivfpq* ix, int list_no);
void go_ivfpq_get_list(go_ivfpq* ix, int list_no, int64_t* ids, uint8_t* codes);
int64_t go_ivfpq_vid_pos(go_ivfpq* ix, int64_t vid);

/* GammaIVFPQIndex::Search.  coarse_mode: -1 = faiss rule (nq<20 seq else BLAS form),
 * 0 = seq, 1 = BLAS form.  Optional outputs (may be NULL): coarse_dis/idx [nq*nprobe],
 * recall_dis/ids [nq*recall_num] (R-heap after heap_reorder, i.e. sorted). */
int go_ivfpq_search(go_ivfpq* ix, const go_search_ctx* ctx, int metric, int nprobe,
                    int recall_num, int has_rank, int coarse_mode, int nq, const float* x,
                    int k, float* distances, int64_t* labels, float* coarse_dis_out,
                    int64_t* coarse_idx_out, float* recall_dis_out, int64_t* recall_ids_out);

/* GammaFLATIndex::Search over a raw vector store [n][d] */
int go_ivfflat_search(go_ivfpq* ix, const go_search_ctx* ctx, int metric, int nprobe, int coarse_mode,
                      int nq, const float* x, int k, float* distances, int64_t* labels,
                      float* coarse_dis_out, int64_t* coarse_idx_out);
void go_ivfflat_assign(go_ivfpq* ix, int64_t n, const float* x, int64_t* list_nos);
int go_flat_search(const float* raw, int64_t n, int d, const go_search_ctx* ctx, int metric,
                   int nq, const float* x, int k, float* distances, int64_t* labels);

/* training: faiss::Clustering::train and IndexIVFPQ::train restated (gamma_oracle.c, "Training") */
void go_rand_perm(int* perm, size_t n, int64_t seed);
void go_set_kmeans_assign_mode(int mode);
void go_kmeans_update(int d, int k, int64_t n, const float* x, const int64_t* assign, float* hassign, float* centroids);
int go_kmeans_split(int d, int k, int64_t n, float* hassign, float* centroids);
float go_kmeans(int d, int64_t n, const float* x, int k, int niter, int64_t seed, int max_points_per_centroid,
                float* centroids);
void go_ivfpq_train(int d, int nlist, int M, int64_t n, const float* x, float* cc, float* pq);
int go_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* GAMMA_ORACLE_H_ */
