/*
 * gamma_hip.h -- C ABI of libgamma_hip.so, the MI355X (gfx950) device shim behind Gamma's
 * RetrievalModel plugin boundary.
 *
 * The only callers are the plugin classes in gamma_amd/host/ (GammaIVFPQHIPIndex, GammaIVFFlatHIPIndex,
 * GammaFLATHIPIndex -- REGISTER_MODEL(HIPIVFPQ / HIPIVFFLAT / HIPFLAT)), i.e. what a Gamma maintainer
 * compiles into libgamma next to index/impl/gamma_index_ivfpq.cc, and
 * the ctypes binding used by this repo's tests and bench.  Everything is extern "C", plain
 * pointers and sizes, caller-owned buffers, int return codes (0 = ok, <0 = error, see
 * gamma_hip_strerror); no C++/torch types, no exceptions cross this line.
 *
 * Each entry point names the reference interface it stands in for (paths relative to the
 * reference tree; "faiss:" = third_party/faiss-1.7.1.tar.gz, faiss-1.7.1/faiss/).
 *
 * Threading (the retrieval contract of SURVEY.md 8b: Search from any number of client threads while ONE indexing thread
 * adds / updates and API threads delete).  A handle owns one device, a search stream and a writer stream.
 *   - Searches may be called from any number of threads.  They take turns on the search stream (they share its
 *     workspaces); small host-buffer calls that find the handle busy are queued and run as ONE device batch.
 *   - Writers (Add, Update, Delete, raw rows, bitmap, columns) serialise among themselves and run on the writer stream
 *     beside the searches: a search reads the inverted lists through the VERSION of their (offset, length) tables that was
 *     current when it was enqueued, a writer publishes a new version after its copies (realtime_mem_data.cc:299-300).
 *   - The list arena and the raw store grow IN PLACE (mapped address ranges, gamma_hip_ivfpq_arena_growth /
 *     gamma_hip_raw_stats): growth waits for nobody.  Only operations that free or move memory a search may be reading
 *     (the arena repack after compactions, column growth, the reallocating fallbacks) wait for the searches in flight.
 * One handle per GPU; several GPUs behind one index object: gamma_hip_group_* below.
 */
#ifndef GAMMA_HIP_H_
#define GAMMA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAMMA_HIP_METRIC_IP 0 /* DistanceComputeType::INNER_PRODUCT, index/retrieval_model.h:20 */
#define GAMMA_HIP_METRIC_L2 1 /* DistanceComputeType::L2 */

/* error codes */
#define GAMMA_HIP_OK 0
#define GAMMA_HIP_EINVAL (-1)     /* bad argument / unsupported parameter */
#define GAMMA_HIP_ENOTTRAINED (-2)/* search before centroids/codebooks were set */
#define GAMMA_HIP_EDEVICE (-3)    /* a HIP runtime call failed (hipGetLastError text kept) */
#define GAMMA_HIP_ENOMEM (-4)     /* device or host allocation failed */
#define GAMMA_HIP_EFULL (-5)      /* list would exceed bucket_max_size (AddKeys returns false) */
#define GAMMA_HIP_EUNSUPPORTED (-6) /* the request asked for exact_ties = 1 on a shape the replay does not cover */

typedef struct gamma_hip_index gamma_hip_index;

/* RangeQueryResult as the scan sees it (table/range_query_result.h:53-67,96-109): a bitmap
 * relative to min_aligned plus [min,max] and the NOT flag.  Host memory; copied per call. */
typedef struct {
    const uint8_t* bitmap;
    int64_t bitmap_bytes;
    int32_t min_doc, max_doc, min_aligned;
    int32_t b_not_in;
} gamma_hip_range_filter;

/* Scalar range filter evaluated ON the device against a numeric column that lives in HBM
 * (gamma_hip_field_append).  Stands in for the per-request docid bitmap that
 * MultiFieldsRangeIndex::Search produces on the host (table/field_range_index.cc:1015-1200)
 * and for the per-doc check the reference's own GPU model makes through Table
 * (IsInRange<T>, index/impl/gpu/gamma_index_ivfpq_gpu.cc:685-727): same comparison, same
 * include_lower / include_upper rules; several filters are AND-ed; a doc beyond the end of
 * the column does not match.  INT / LONG columns compare lower_i / upper_i, FLOAT / DOUBLE
 * columns lower_f / upper_f (FLOAT in float). */
#define GAMMA_HIP_FIELD_INT 0
#define GAMMA_HIP_FIELD_LONG 1
#define GAMMA_HIP_FIELD_FLOAT 2
#define GAMMA_HIP_FIELD_DOUBLE 3
#define GAMMA_HIP_MAX_FIELD_FILTERS 8
typedef struct {
    int32_t field_id;
    int32_t include_lower, include_upper;
    int32_t reserved;
    int64_t lower_i, upper_i;
    double lower_f, upper_f;
} gamma_hip_field_filter;

/* Term filter on a STRING field evaluated on the device (GammaSearchCondition::term_filters).  The field value and the
 * filter value are lists of items separated by \001; is_union Or = any filter item among the doc's items, And = all of
 * them, as in FilteredByTermFilter (index/impl/gpu/gamma_index_ivfpq_gpu.cc:727-762).  Not (FilterOperator::Not,
 * table/field_range_index.h:23) = NONE of the filter items, which is what the engine's CPU bitmap path computes
 * (the reference's GPU-side FilteredByTermFilter treats Not like And; the device filter follows the bitmap path so that
 * "device_filters": 1 and 0 return the same documents).  Items are dictionary-encoded by the
 * caller (the plugin keeps the dictionary, gamma_amd/host/filter_bridge.h); an item the dictionary does not
 * know is passed as -1.  The doc's items live in HBM (gamma_hip_term_append). */
#define GAMMA_HIP_MAX_TERM_FILTERS 8
#define GAMMA_HIP_MAX_TERM_ITEMS 16
#define GAMMA_HIP_TERM_AND 0
#define GAMMA_HIP_TERM_OR 1
#define GAMMA_HIP_TERM_NOT 2
typedef struct {
    int32_t field_id;
    int32_t op;        /* TermFilter::is_union as the reference passes it: 0 And, 1 Or, 2 Not */
    int32_t n_items;
    int32_t items[GAMMA_HIP_MAX_TERM_ITEMS];
} gamma_hip_term_filter;

/* What GammaSearchCondition + IVFPQRetrievalParameters carry into Search()
 * (common/gamma_common_data.h:39-124, index/impl/gamma_index_ivfpq.h:629-673). */
typedef struct {
    int32_t metric;       /* per-request DistanceComputeType (gamma_index_ivfpq.cc:744-750) */
    int32_t nprobe;       /* resolved by the caller as gamma_index_ivfpq.cc:539-545 does */
    int32_t recall_num;   /* raised to k inside (gamma_index_ivfpq.cc:762-765) */
    int32_t has_rank;     /* GammaSearchCondition::has_rank: exact re-rank of recall_num */
    float min_score;      /* IsSimilarScoreValid window (gamma_common_data.h:95-97) */
    float max_score;
    int32_t coarse_mode;  /* -1: faiss rule nq<20 -> exact per-pair, else GEMM form
                             (faiss:utils/distances.cpp:303,346); 0 exact; 1 GEMM (MFMA) */
    int32_t has_range;    /* range_query_result != nullptr */
    int32_t n_range;      /* number of RangeQueryResult; 0 with has_range => nothing valid */
    const gamma_hip_range_filter* range;
    int32_t n_field;      /* GammaSearchCondition::range_filters evaluated on device columns */
    int32_t n_term;       /* GammaSearchCondition::term_filters evaluated on device columns */
    const gamma_hip_field_filter* field;
    const gamma_hip_term_filter* term;
    int32_t exact_ties;   /* the reference's heap order inside exact distance ties (gamma_hip_set_exact_ties) for this
                             request: 0 = the handle's setting (default on), 1 = on, -1 = off */
} gamma_hip_search_params;

/* ---- lifecycle ------------------------------------------------------------------- */
/* replaces: new GammaIVFPQIndex / GammaFLATIndex + Init (gamma_index_ivfpq.cc:119-214,
 * gamma_index_flat.cc:62-76).  One handle = one device. */
int gamma_hip_create(int device, gamma_hip_index** out);
int gamma_hip_destroy(gamma_hip_index* h);
const char* gamma_hip_strerror(int code);
const char* gamma_hip_last_error(gamma_hip_index* h);
/* hipStream_t the handle launches on (for event timing / overlap by the caller) */
void* gamma_hip_stream(gamma_hip_index* h);
int gamma_hip_synchronize(gamma_hip_index* h);

/* The IVFPQ scan of large batches bounds each query's recall_num-th best distance from its nearest probe group and keeps
 * only the candidates within the bound of the other groups (csrc/scan.hip); queries whose bound turns out too loose (a
 * survivor slice overflows) are scored again and selected from the full candidate row.  Results do not depend on it.  On
 * data where the nearest lists do not hold the best candidates (full-size C5: noise-dominated inner-product vectors) nearly
 * every query falls through and the pre-filter costs more than it saves (41 ms per 4096 queries against 24 without it):
 * the handle counts the queries that fall through and, while they are more than half of the recent ones, scans unbounded,
 * on = 0: no feedback, the pre-filter whenever the shape allows it.
 * re-probing every 256 calls; what it learnt holds for one kind of call (nprobe, recall_num, metric, filter or not).
 * stats: out4 = {queries that fell through, queries that went through the bounded scan -- both of the current kind of call --,
 * times the handle backed off, consumer workgroups that ever gave up waiting for their bound}. */
int gamma_hip_set_scan_bound_feedback(gamma_hip_index* h, int on);
int gamma_hip_scan_bound_stats(gamma_hip_index* h, int64_t* out4);
/* Upper bound in bytes of the per-chunk workspaces (coarse distance matrix, ADC distance buffer);
 * larger calls are processed in chunks of queries.  Default: an eighth of the device memory, 1 to 32 GiB. */
int gamma_hip_set_workspace_budget(gamma_hip_index* h, int64_t bytes);
/* Exact distance ties.  The reference keeps candidates in faiss binary heaps (faiss:utils/Heap.h:46-131): WHICH of
 * several candidates at exactly the same fp32 distance survive a cut, and the order equal distances come out in,
 * is what the heap's sift order leaves -- at the nprobe cut of the coarse quantizer (HeapResultHandler,
 * faiss:impl/ResultHandler.h:112-117), at the recall_num cut of the list scan (KnnSearchResults::add,
 * index/impl/gamma_index_ivfpq.h:363-369) and in compute_dis, which feeds the k-heap in the ARRAY order of the
 * unsorted recall heap (index/impl/gamma_index_ivfpq.cc:646-680).  on != 0 (the DEFAULT): rows / queries in which
 * such a tie can change the result are detected by the regular kernels and redone by replaying the reference's heaps
 * (csrc/ties.hip, csrc/heap_dev.h), so labels and ranks are the reference's, ties included.  on == 0: a top-k is "the
 * k smallest (distance, scan position) pairs" -- the same distances at every rank, the same ids up to the order /
 * membership inside groups of exactly equal distances, and no replay cost (a few queries in a thousand on integer
 * data, none to speak of on real-valued data).  Per request: gamma_hip_search_params.exact_ties.
 * Covers IVFPQ / IVFFLAT / flat search, every batch size, every recall_num and k the ABI accepts (<= 4096), the
 * list-sharded merge and the group.  From 100 probes on the coarse assignment is replayed through faiss's ReservoirTopN
 * (what knn_L2sqr collects through there, faiss:utils/distances.cpp:341-358; csrc/reservoir_dev.h), below through its
 * result heap.  NOT covered: nprobe > 1024, a flat search for k = 4096 or over 2^31 rows.  Such a call never
 * degrades silently: with exact_ties = 1 in the request it fails with GAMMA_HIP_EUNSUPPORTED; when it merely inherits the
 * handle's default it runs with the (distance, position) order inside ties and is counted
 * (gamma_hip_ties_not_honoured). */
int gamma_hip_set_exact_ties(gamma_hip_index* h, int on);
/* Coarse quantizer of large batches (>= 4096 queries, >= 2048 lists, d in {32, 64, 96, 128}, nprobe <= 64) without
 * the [nq][nlist] distance matrix (csrc/coarse.hip).  on: 1 = automatic (default), 0 = always the matrix path.
 * list_cap: capacity of a query's per-strip survivor list, 1..128 (default 128); a query that overflows it is redone
 * by the repair kernel -- tests shrink it to force that path.  Results are identical either way. */
int gamma_hip_set_coarse_fused(gamma_hip_index* h, int on, int list_cap);
/* Calls of up to 512 queries (nprobe <= 128, recall_num <= 1024) run as a chain of
 * four or five fused kernels instead of eleven (DESIGN.md "small batches").  on: 1 = automatic (default), 0 = always
 * the regular chain, n >= 2 = automatic with the two-level selection of long candidate rows forced on with n slices
 * (tests; by default it is used when nprobe x 1.5 mean list lengths exceed 16384).  Results are identical either way. */
int gamma_hip_set_small_path(gamma_hip_index* h, int on);
/* out3 = {coarse rows redone, queries whose recall_num cut went through a tie, queries replayed} since creation
 * or the last reset; meaningful with exact ties on */
int gamma_hip_tie_stats(gamma_hip_index* h, int64_t* out3, int reset);
/* search calls since creation (or the last reset) that ran WITHOUT the exact-ties mode although the handle's default asked
 * for it, because their shape is beyond the replay's range (see gamma_hip_set_exact_ties); the reference has no such
 * limit (faiss:utils/Heap.h:103-131, index/impl/gamma_index_ivfpq.cc:762-770) */
int gamma_hip_ties_not_honoured(gamma_hip_index* h, int64_t* out_calls, int reset);
/* Calls (searches, Add / assign, k-means iterations' assignments) whose GEMM-form coarse distances fell into a shape for
 * which the compiled library's sgemm_ kernel is NOT restated (faiss:utils/distances.cpp:215-296 through MKL: K > 768 or an
 * odd split, K = 384 with a database remainder block of 9..512 rows, remainder blocks of 1..7 rows -- nq mod 4096 or
 * nlist mod 1024).  There the coarse distances of a few entries can differ from the library's by an ulp (measured:
 * tests/test_oracle_vs_ref.py::test_unrestated_blas_corners_differ_by_ulps); every other shape is the library's bit for
 * bit.  Counted, never silent; the plugins log a new count. */
int gamma_hip_blas_form_not_restated(gamma_hip_index* h, int64_t* out_calls, int reset);

/* ---- numeric scalar columns for on-device range filters (docid = row).  The engine side
 *      appends a doc's value when the doc is added (Table::Add, table/table.cc) ----------- */
int gamma_hip_field_append(gamma_hip_index* h, int field_id, int dtype, int64_t n, const void* values);
int gamma_hip_field_update(gamma_hip_index* h, int field_id, int64_t docid, const void* value);
int64_t gamma_hip_field_count(gamma_hip_index* h, int field_id);
/* STRING columns for on-device term filters: n more docs (docid = row), doc i with counts[i] dictionary-encoded
 * items, all items concatenated in `items` (Table::Add -> the plugin's dictionary, filter_bridge.h) */
int gamma_hip_term_append(gamma_hip_index* h, int field_id, int64_t n_docs, const int32_t* counts,
                          const int32_t* items);
int64_t gamma_hip_term_count(gamma_hip_index* h, int field_id);
/* a doc's items rewritten after its STRING field was updated in the table (Table::Update, table/table.cc:420-470):
 * in place when they fit, else at the end of the item array; the doc's row switches over with one 8-byte store */
int gamma_hip_term_update(gamma_hip_index* h, int field_id, int64_t docid, int32_t count, const int32_t* items);

/* ---- raw vector store (VectorReader::Gets / MemoryRawVector, vector/raw_vector.cc:99-109,
 *      vector/memory_raw_vector.cc:90-142): device mirror, vid = row ------------------- */
int gamma_hip_raw_init(gamma_hip_index* h, int d);
int gamma_hip_raw_append(gamma_hip_index* h, int64_t n, const float* vecs);
int gamma_hip_raw_update(gamma_hip_index* h, int64_t vid, const float* vec);
/* n rows rewritten with one wait for the device (vids the mirror has not reached yet are skipped) */
int gamma_hip_raw_update_batch(gamma_hip_index* h, int64_t n, const int64_t* vids, const float* vecs);
/* rows [first_vid, first_vid + n) written at their own position: idempotent (a repeated or overlapping
 * call rewrites the same rows with the same bytes), the row count only ever grows to first_vid + n.
 * first_vid may not leave a gap (first_vid <= gamma_hip_raw_count).  This is what a mirror of the engine's
 * vector store needs when Search (brute force before training) and the indexing thread's Add race. */
int gamma_hip_raw_write(gamma_hip_index* h, int64_t first_vid, int64_t n, const float* vecs);
/* VectorReader::Gets (vector/raw_vector.cc:99-109, vector/memory_raw_vector.cc:136-142): rows vids[0..n) of the device
 * store copied to out [n][d] (the rows compute_dis reads); GAMMA_HIP_EINVAL for a vid outside [0, gamma_hip_raw_count) */
int gamma_hip_raw_gets(gamma_hip_index* h, int64_t n, const int64_t* vids, float* out);
/* Raw vectors SHARDED WITH THEIR LISTS (round 6; SURVEY 8(e): "shard by the same owner"): a list shard of a multi-GPU job keeps
 * the rows of the vectors in ITS lists only -- rows vecs[i] of vector ids vids[i], in any order, appended to the store with a
 * vid -> row table (4 bytes per vector id) beside it; C4 on 8 GPUs: 6.4 GB of rows per GPU instead of 51.2.  The first call
 * turns the (empty) store into that form for good.  Such a handle serves gamma_hip_ivfpq_shard_exact / _shard_export_exact; what
 * needs every row -- has_rank inside a search on this handle, flat / IVFFLAT search, raw_gets / raw_write / raw_update --
 * answers GAMMA_HIP_EUNSUPPORTED.  Replaces MemoryRawVector::GetVector for the rows a shard owns
 * (vector/memory_raw_vector.cc:136-142). */
int gamma_hip_raw_put(gamma_hip_index* h, int64_t n, const int64_t* vids, const float* vecs);
int64_t gamma_hip_raw_count(gamma_hip_index* h);
/* out4 = {rows, rows the mapped / allocated memory holds, reallocations that MOVED the store so far, 1 when the store
 * grows in place (virtual memory management: physical chunks mapped behind the rows, nothing ever moves or waits for
 * the searches in flight -- what the reference gets from its 500 000-vector segments, vector/memory_raw_vector.cc:90-142)} */
int gamma_hip_raw_stats(gamma_hip_index* h, int64_t* out4);

/* ---- delete bitmap (bitmap::BitmapManager, util/bitmap_manager.cc:171-192): bit = docid,
 *      byte docid>>3, mask 1<<(docid&7) --------------------------------------------------- */
int gamma_hip_bitmap_upload(gamma_hip_index* h, const uint8_t* bitmap, int64_t nbits);
int gamma_hip_bitmap_set(gamma_hip_index* h, const int64_t* docids, int64_t n, int value);

/* ---- IVFPQ model state (faiss::IndexIVFPQ members the scanner reads) --------------- */
/* replaces GammaIVFPQIndex::Init's parameter checks (gamma_index_ivfpq.cc:119-214);
 * nbits must be 8, d % M == 0 (OPQ / HNSW quantizer / padding are rejected: EINVAL) */
int gamma_hip_ivfpq_init(gamma_hip_index* h, int d, int nlist, int M, int nbits, int metric,
                         int bucket_init_size, int bucket_max_size);
/* quantizer->xb (nlist*d), pq.centroids (M*ksub*dsub), and the precomputed table
 * (faiss:IndexIVFPQ.cpp:412-479; NULL => computed on device with identical arithmetic) */
int gamma_hip_ivfpq_set_trained(gamma_hip_index* h, const float* coarse_centroids,
                                const float* pq_centroids, const float* precomputed_table);
int gamma_hip_ivfpq_get_precomputed_table(gamma_hip_index* h, float* out);
/* L2 table mode (faiss::IndexIVFPQ::use_precomputed_table as train / Load leave it, faiss:IndexIVFPQ.cpp:132-135,
 * index/impl/gamma_index_ivfpq.cc:1033-1034).  Replaces the extern `faiss::precomputed_table_max_bytes`
 * (faiss:IndexIVFPQ.cpp:379, 2 GiB; process-wide here too): an index whose table nlist * nsubvector * 1 KiB would be
 * LARGER keeps mode 0 (faiss:IndexIVFPQ.cpp:441-449) -- no table is built or held, gamma_hip_ivfpq_get_precomputed_table
 * returns GAMMA_HIP_EUNSUPPORTED, and L2 searches score every (query, list) pair with the distance table of the residual
 * x - centroid (index/impl/gamma_index_ivfpq.h:239-245: compute_residual + pq.compute_distance_table, dis0 = 0), whose
 * fp32 values differ from mode 1's by rounding.  The rule is applied by gamma_hip_ivfpq_init (it depends on nlist, M and
 * the limit only): set the limit before Init.  _use_precomputed_table: 0 / 1, GAMMA_HIP_EINVAL before Init. */
int gamma_hip_set_precomputed_table_max_bytes(int64_t bytes);
int64_t gamma_hip_get_precomputed_table_max_bytes(void);
int gamma_hip_ivfpq_use_precomputed_table(gamma_hip_index* h);
/* shape of an initialised model: dimension, number of lists, bytes per code (nsubvector; 1 for IVFFLAT); 0 before Init */
int gamma_hip_ivfpq_dim(gamma_hip_index* h);
int gamma_hip_ivfpq_nlist(gamma_hip_index* h);
int gamma_hip_ivfpq_code_size(gamma_hip_index* h);

/* ---- realtime inverted lists (realtime::RTInvertIndex, realtime/realtime_invert_index.h)
 *      resident in HBM ----------------------------------------------------------------- */
/* RTInvertIndex::AddKeys for one bucket (realtime_mem_data.cc:264-303) */
int gamma_hip_ivfpq_add_keys(gamma_hip_index* h, int list_no, int n, const int64_t* vids,
                             const uint8_t* codes);
/* many buckets at once: vids/codes grouped by list, list_nos[i] has counts[i] entries */
int gamma_hip_ivfpq_add_keys_batch(gamma_hip_index* h, int nlists, const int32_t* list_nos,
                                   const int32_t* counts, const int64_t* vids,
                                   const uint8_t* codes);
/* RTInvertIndex::Update (realtime_mem_data.cc:305-327) */
int gamma_hip_ivfpq_update(gamma_hip_index* h, int list_no, int64_t vid, const uint8_t* code);
/* The two halves of Update for a list-sharded index, where the list a vector leaves and the list it joins may
 * live on different shards (gamma_amd/dist.py sharded_update): _has_vid: out[i] = 1 when this handle holds a live
 * entry of vids[i]; _remove: the first half of RealTimeMemData::Update (:318-321) alone -- the entry is flagged
 * as moved away and counted as deleted; unknown vids are ignored like Update ignores them (:307-311). */
int gamma_hip_ivfpq_has_vid(gamma_hip_index* h, const int64_t* vids, int n, uint8_t* out);
/* GammaIVFPQIndex::Update for a batch of n vectors (gamma_index_ivfpq.cc:375-422; the engine drains up to 20 000
 * updated vids per pass, vector/vector_manager.cc:355-380): every vector is assigned and encoded the way a call of
 * its own assigns it (quantizer->assign(1, ..), the exact form), in ONE device pass; then RTInvertIndex::Update for
 * each (vid, list, code) in order, one publish of the lists' tables, one wait for the device.  vecs: n*d fp32. */
int gamma_hip_ivfpq_update_batch(gamma_hip_index* h, int n, const int64_t* vids, const float* vecs);
/* the encode half alone (list_nos[n], codes[n*code_size] to host), and the list half alone with the codes supplied:
 * ops[i] = 0 Update (a vid this handle does not hold is ignored, realtime_mem_data.cc:307-311), 1 = the vid is held by
 * ANOTHER shard of a list-sharded index and joins list_nos[i] here (AddKeys), 2 = the vid leaves this shard (the first
 * half of Update: flagged as moved, counted as deleted); ops == NULL: all 0.  (gamma_hip_group_ivfpq_update) */
int gamma_hip_ivfpq_encode_each(gamma_hip_index* h, int64_t n, const float* vecs, int64_t* list_nos, uint8_t* codes);
int gamma_hip_ivfpq_apply_updates(gamma_hip_index* h, int n, const int32_t* list_nos, const int64_t* vids,
                                  const uint8_t* codes, const uint8_t* ops);
int gamma_hip_ivfpq_remove(gamma_hip_index* h, int64_t vid);
/* RTInvertIndex::Delete (realtime_mem_data.cc:329-335,190-199): counter only */
int gamma_hip_ivfpq_delete(gamma_hip_index* h, const int64_t* vids, int n);
/* RTInvertIndex::CompactIfNeed (realtime_mem_data.cc:354-381,119-150); needs the delete
 * bitmap the handle mirrors */
int gamma_hip_ivfpq_compact_if_need(gamma_hip_index* h);
/* Extents abandoned by list growth and compaction (the reference frees them 1 s after the swap,
 * realtime_mem_data.cc:457-466) stay inside the arena until they exceed half of what is in use and
 * `min_waste_entries` (default 65536): then every list moves into a fresh tight arena.  out4 = {capacity, in use,
 * abandoned, repacks so far}, in list entries (code_size + 8 bytes each).  Setting the threshold re-checks at once. */
int gamma_hip_ivfpq_arena_stats(gamma_hip_index* h, int64_t* out4);
int gamma_hip_ivfpq_set_repack_threshold(gamma_hip_index* h, int64_t min_waste_entries);
/* out2 = {repack targets read back before their version of the list tables was published, read-backs that DIFFERED from
 * the source}.  A repack moves every list (realtime/realtime_mem_data.cc:426-474 swaps a bucket's pointer only after the
 * copy); the new version is published only after per-list checksums of ids + codes, recomputed through the new mapping in
 * a launch of their own, equal those of the source.  A difference keeps the old version, retires the target's address
 * ranges and repeats the move into ordinary allocations; the plugins log a non-zero second value. */
int gamma_hip_ivfpq_repack_verify_stats(gamma_hip_index* h, int64_t* out2);
/* How the arena grows: out2 = {growths that MOVED the arena (reallocation + copy under the exclusive lock), 1 when the
 * arena's arrays are mapped address ranges that grow in place (chunks mapped behind them: no copy, no second arena, no
 * wait for the searches in flight -- the reference grows bucket by bucket for the same reason,
 * realtime/realtime_mem_data.cc:152-188,426-474)}.  GAMMA_HIP_NO_ARENA_VMM=1 in the environment forces the fallback. */
int gamma_hip_ivfpq_arena_growth(gamma_hip_index* h, int64_t* out2);
int64_t gamma_hip_ivfpq_list_size(gamma_hip_index* h, int list_no);
int64_t gamma_hip_ivfpq_list_capacity(gamma_hip_index* h, int list_no);
/* RealTimeMemData::RetrieveCodes ("for unit test", realtime_mem_data.h:95-96) */
int gamma_hip_ivfpq_get_list(gamma_hip_index* h, int list_no, int64_t* vids, uint8_t* codes);
/* restrict the scan to the lists a shard owns (multi-GPU list sharding): owner[l] != 0.  With a mask
 * set, gamma_hip_ivfpq_add keeps only the vectors assigned to owned lists: every shard is handed the same
 * batch, which routes realtime inserts to the owner of the list without any exchange. */
int gamma_hip_ivfpq_set_list_mask(gamma_hip_index* h, const uint8_t* owned);

/* device-side Add path (gamma_index_ivfpq.cc:424-512): assign + residual + PQ encode +
 * AddKeys for n vectors with consecutive vids starting at first_vid. */
int gamma_hip_ivfpq_add(gamma_hip_index* h, int64_t n, const float* vecs, int64_t first_vid);
int gamma_hip_ivfpq_encode(gamma_hip_index* h, int64_t n, const float* vecs, int64_t* list_nos,
                           uint8_t* codes);

/* Training helper (GammaIVFPQIndex::Indexing -> faiss::Clustering's assignment step,
 * gamma_index_ivfpq.cc:272-354): nearest of k centroids (squared L2, GEMM form on MFMA) for n
 * host vectors of dimension d; the Lloyd update stays on the host.  dis may be NULL. */
int gamma_hip_assign(gamma_hip_index* h, int d, int64_t n, const float* x, int k,
                     const float* centroids, int32_t* assign, float* dis);

/* Training on the device: faiss::Clustering::train (faiss:Clustering.cpp:255-560; nredo 1, L2, no weights) as
 * GammaIVFPQIndex::Indexing runs it through IndexIVFPQ::train (gamma_index_ivfpq.cc:272-354) -- subsample_training_set
 * to k * max_points_per_centroid with rand_perm(seed), centroids initialised from rand_perm(seed + 1), niter times
 * { assign (the coarse quantizer's kernels), compute_centroids (float sums in point order), split_clusters }.  The
 * training set is uploaded once; x: n*d fp32 host, centroids: k*d fp32 host out, objective (may be NULL): sum of the
 * distances of the last assignment.  Bit-identical to the oracle's go_kmeans, which is pinned against compiled faiss. */
int gamma_hip_kmeans(gamma_hip_index* h, int d, int64_t n, const float* x, int k, int niter, int64_t seed,
                     int max_points_per_centroid, float* centroids, float* objective);
/* IndexIVFPQ::train as GammaIVFPQIndex::Indexing runs it (index/impl/gamma_index_ivfpq.cc:272-354, faiss:IndexIVFPQ.cpp:
 * 67-131): coarse k-means (niter 10), residuals of at most 65536 subsampled points, one 256-centroid k-means (niter 25)
 * per sub-quantizer -- every k-means the gamma_hip_kmeans above.  x: n*d fp32 host; coarse: nlist*d, pq: M*256*(d/M) fp32
 * host out.  Bit-identical to the oracle's go_ivfpq_train, which is bit-identical to the compiled library's
 * IndexIVFPQ::train at its default BLAS threshold (tests/test_training_cpu.py). */
int gamma_hip_ivfpq_train(gamma_hip_index* h, int d, int64_t n, const float* x, int nlist, int M, float* coarse, float* pq);
/* faiss::rand_perm (faiss:utils/random.cpp:136-146; std::mt19937): the permutation IndexIVFPQ::train_residual_o
 * subsamples its training set with (host only) */
void gamma_hip_rand_perm(int32_t* perm, int64_t n, int64_t seed);

/* ---- multi-vector documents (VIDMgr, vector/raw_vector_common.h:36-110) -------------------------
 * A table whose documents carry several vectors per field has vid != docid; every validity test of the path --
 * the delete bitmap, the per-request range bitmaps, the device columns (GammaSearchCondition::IsValid,
 * common/gamma_common_data.h:99-108) and the realtime lists' delete accounting (realtime_mem_data.cc:102,294) --
 * is on the DOC id.  Append the docid of vids [count, count + n) in vid order; never called = single-vector
 * documents (docid == vid).  Labels returned by the searches stay vector ids, as in the reference. */
int gamma_hip_vid2docid_append(gamma_hip_index* h, int64_t n, const int32_t* docids);
int64_t gamma_hip_vid2docid_count(gamma_hip_index* h);

/* ---- IVFFLAT (index/impl/gamma_index_ivfflat.{h,cc}) ----------------------------------------
 * The reference's IVFFLAT lists carry the vectors themselves (code_size = 4 d, gamma_index_ivfflat.cc:155); here a
 * list carries vector ids (plus one dummy code byte so that the realtime-list code is shared) and the rows come
 * from the raw store (gamma_hip_raw_*), which must hold every indexed vector.  After _init the list entry points
 * above serve the handle unchanged: gamma_hip_ivfpq_add (= GammaIndexIVFFlat::Add :305-355: assign + AddKeys),
 * _encode (list_nos = quantizer->assign; codes = dummy bytes), _add_keys / _update / _delete /
 * _compact_if_need / _get_list / _list_size.  nprobe, metric, filters and the score window come from
 * gamma_hip_search_params (recall_num / has_rank are ignored: every scanned entry gets its exact distance). */
int gamma_hip_ivfflat_init(gamma_hip_index* h, int d, int nlist, int metric, int bucket_init_size,
                           int bucket_max_size);
/* quantizer->xb (nlist*d): IndexIVFFlat::train == the coarse k-means only */
int gamma_hip_ivfflat_set_trained(gamma_hip_index* h, const float* coarse_centroids);
/* replaces GammaIndexIVFFlat::Search (gamma_index_ivfflat.cc:392-421 + search_preassigned :423-612, pmode 0):
 * x nq*d fp32 host; distances/labels nq*k host, best first, -1 / heap neutral padded */
int gamma_hip_ivfflat_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* x, int k,
                             float* distances, int64_t* labels);
int gamma_hip_ivfflat_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* d_x,
                                    int k, float* d_distances, int64_t* d_labels);

/* ---- search ------------------------------------------------------------------------ */
/* replaces GammaIVFPQIndex::Search (gamma_index_ivfpq.cc:514-566 + search_preassigned
 * :701-890).  x: nq*d fp32 host; distances/labels: nq*k host, best first, unused slots
 * label -1 / distance = heap neutral (FLT_MAX for L2, -FLT_MAX for IP).
 * Re-entrant like the reference's Search: any number of threads may call it on one handle.  Calls of up to 256 queries
 * that find the handle busy are queued and run as ONE device batch per set of equal parameters (each caller gets the
 * result of its own call, bit for bit); a call that finds it idle runs on the caller's thread, its queries and results
 * staged in pinned memory (no pageable copies).  The buffers need not be pinned. */
int gamma_hip_ivfpq_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                           const float* x, int k, float* distances, int64_t* labels);
/* same, all pointers in device memory, enqueued on the handle's stream, no sync */
int gamma_hip_ivfpq_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                  const float* d_x, int k, float* d_distances, int64_t* d_labels);
/* Streaming callers of gamma_hip_ivfpq_search_device (exact ties on): with deferred replay the queries flagged for the
 * heap replay (a fraction of a percent; one query's replay is a ~0.2 ms chain of dependent sifts, §4 of DESIGN.md) are
 * redone on a side stream, and the handle's stream waits for them only before the next kernel that touches what the
 * replay reads -- in the NEXT call (or the next chunk of a large call), after its coarse quantizer and query tables.
 * The contract while it is on: d_distances / d_labels of a call are complete -- and d_x may be reused -- after the
 * next search call on the handle, gamma_hip_join (the handle's stream waits for the pending replay; no host wait) or
 * gamma_hip_synchronize.  Host-buffer calls are not affected.  Default: off. */
int gamma_hip_set_deferred_replay(gamma_hip_index* h, int on);
/* The reference's Search is re-entrant and returns a COMPLETE result to each of its client threads (tests/test.h:1033-1062,
 * tools/perf.cc:364-395).  _search_device_wait is that contract for device buffers: it returns when THIS call's
 * d_distances / d_labels are complete (it waits on the host for the call's own completion event) -- and while it waits
 * the handle is free: the tie replay of this call (serial by nature, ~200 waves on the whole device) runs on the side
 * stream and the NEXT caller's coarse quantizer, query tables and scan run beside it, exactly as under
 * gamma_hip_set_deferred_replay, with no contract on the callers but "wait for your own call".  Two client threads
 * alternating on one handle thus hide each other's replay tail; one thread alone gets what _search_device +
 * gamma_hip_synchronize gives.  d_x must stay untouched until the call returns. */
int gamma_hip_ivfpq_search_device_wait(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                       const float* d_x, int k, float* d_distances, int64_t* d_labels);
/* the same contract for the flat search (GammaFLATIndex::Search, gamma_index_flat.cc:118-300): complete on return, and the call's
 * heap replay -- a third of a 1024-query call over 1 M rows -- runs on the side stream beside the next caller's filter passes.
 * What the replay reads (the first row chunk's slab, the survivor log, the flag list, the candidate tables) exists twice: a call
 * that finds a flat replay pending works in the other bank. */
int gamma_hip_flat_search_device_wait(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                      const float* d_x, int k, float* d_distances, int64_t* d_labels);
int gamma_hip_join(gamma_hip_index* h);
/* stage outputs of the LAST search for parity tests / sharded merge (device->host):
 * coarse_dis/idx [nq*nprobe], recall_dis/ids [nq*recall_num] (sorted best first, -1 pad) */
int gamma_hip_ivfpq_last_stages(gamma_hip_index* h, float* coarse_dis, int64_t* coarse_idx,
                                float* recall_dis, int64_t* recall_ids);
/* sharded search, stage 1: coarse + owned-list scan + local top-recall_num, results left in
 * device buffers d_recall_dis/d_recall_ids [nq*recall_num] for the all-gather */
int gamma_hip_ivfpq_search_shard(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                 const float* d_x, int k, float* d_recall_dis,
                                 int64_t* d_recall_ids);
/* sharded search with the query batch split over the ranks: stage 0 = the coarse quantizer alone
 * (IndexFlatL2::search, faiss:IndexFlat.cpp:35-55) for this rank's slice of the queries; outputs
 * [nq*nprobe] in device memory, exchanged with an all-gather */
int gamma_hip_ivfpq_coarse_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                  const float* d_x, float* d_coarse_dis, int32_t* d_probe);
/* stage 1 with the coarse assignment supplied = search_preassigned restricted to the owned lists
 * (gamma_index_ivfpq.cc:701-890): tables + scan + local top-recall_num.  Enqueued like every device-pointer call, except
 * when the batch does not fit the workspace budget at the general slab stride (nprobe x the longest list; a W-rank job hands
 * every rank W x the queries but ~nprobe / W of their probes): then the longest candidate row of THIS batch over THIS shard's
 * lists is measured on the device and read back (one word; the call waits for the work enqueued before it) and stride and
 * chunks are sized by it -- full-size C4, 8 shards: 1-2 chunks instead of 20. */
int gamma_hip_ivfpq_search_shard_preassigned(gamma_hip_index* h, const gamma_hip_search_params* p,
                                             int nq, const float* d_x, const float* d_coarse_dis,
                                             const int32_t* d_probe, int k, float* d_recall_dis,
                                             int64_t* d_recall_ids);
/* Stage 1 in TWO PHASES with a bound of the GLOBAL recall_num-th best (round 6).  A list shard on its own can only
 * bound its LOCAL recall_num-th best -- with W shards about W times looser than what one GPU holding every list gets from
 * the query's nearest probes -- so all its probes take the expensive exact path and it exports recall_num candidates per
 * query of which the merge keeps ~1 / W.  Here the shard first scores only the query's nearest `GAMMA_HIP_SHARD_G1`
 * (default 2) probes it OWNS, which bound its local recall_num-th best exactly as the single-GPU producers do
 * (gamma_index_ivfpq.h:351-370: anything worse than the recall_num-th best seen can never enter the R-heap), writes that
 * bound per query to d_bound [nq] (+inf / -inf: none), and calls `reduce` ONCE: the caller's collective replaces d_bound[q]
 * by the minimum (L2; take_max = 0) or maximum (inner product; take_max = 1) over all shards -- every shard's value is an
 * upper bound of the global recall_num-th best, so the tightest one is too -- enqueued on `stream` (the handle's).
 * The shard's other probes then run as consumers of that bound (byte-table filter pass, list-major for long lists) and
 * the shard's table holds its candidates within the bound, best first, padded with (sentinel, -1): a SUPERSET of the
 * shard's part of the global top-recall_num, so the merge (gamma_hip_ivfpq_merge_rerank) and the tie machinery are
 * unchanged.  Results are identical to _search_shard_preassigned's after the merge.  `reduce` is called exactly once per
 * call -- with neutral values when the call cannot run in two phases (several chunks) -- so the shards' collectives
 * always pair up; NULL reduce = a single shard (the bounds stay local).  d_bound holds the reduced bounds on return
 * (stream order).  Replaces the per-shard scan of faiss:IndexShards.cpp:283-345 /
 * index/impl/gpu/gamma_gpu_cloner.cpp:200-253, where every shard scans with its own heap threshold. */
typedef int (*gamma_hip_bound_reduce_fn)(void* user, float* d_bound, int nq, int take_max, void* stream);
int gamma_hip_ivfpq_search_shard_bounded(gamma_hip_index* h, const gamma_hip_search_params* p,
                                         int nq, const float* d_x, const float* d_coarse_dis,
                                         const int32_t* d_probe, int k, float* d_recall_dis,
                                         int64_t* d_recall_ids, float* d_bound,
                                         gamma_hip_bound_reduce_fn reduce, void* user);
/* building block of a reduce callback without a collective library (the in-process group's peer-copy transport):
 * d_acc[i] = min (take_max = 0) / max (1) of d_acc[i] and d_in[i], i < n, enqueued on `stream` of the calling thread's
 * current device.  Takes no handle and no lock: it is called from inside _search_shard_bounded's callback. */
int gamma_hip_bound_combine(void* stream, float* d_acc, const float* d_in, int n, int take_max);
/* compute_dis's exact distances (gamma_index_ivfpq.cc:646-680: fvec_L2sqr / fvec_inner_product on the raw query, then
 * IsSimilarScoreValid) where the ROWS are: d_exact[q][r] for candidate d_ids[q][r] (device, [nq][R]; -1 = none) of query
 * d_x[q] -- the sentinel (+inf L2 / -inf inner product) for an empty slot, a score outside the request's window, or a vector
 * this handle's sharded store does not hold.  With raw vectors sharded by list owner every shard calls this on the candidates
 * it is about to send (after the two-phase scan: ~recall_num / W per query) and the distances travel with them. */
int gamma_hip_ivfpq_shard_exact(gamma_hip_index* h, const gamma_hip_search_params* p, int nq, const float* d_x,
                                const int64_t* d_ids, int R, float* d_exact);
/* gamma_hip_ivfpq_merge_rerank for candidates that arrive WITH their exact distances (d_all_exact laid out like d_all_dis):
 * the global top-recall_num by ADC value as ever, then compute_dis's k-heap fed from the travelled distances -- the owner of
 * the query slice needs no raw row.  Same results, flags and tie list as _merge_rerank on a handle that holds every row. */
int gamma_hip_ivfpq_merge_rerank_exact(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nq,
                                       const float* d_x, int k, const float* d_all_dis, const int64_t* d_all_ids,
                                       const float* d_all_exact, int q0, int nq_local, float* d_distances, int64_t* d_labels);
/* the tie phase with sharded raw vectors: after gamma_hip_ivfpq_shard_export the shard computes the exact distance (no score
 * window) of every exported entry whose ADC value lies within the query's bound d_bound_f[f] (NaN: of every entry) and whose
 * vector it holds -- the only entries that can end up in the recall_num-heap -- into d_ex [nf][stride], NaN elsewhere;
 * gamma_hip_ivfpq_merge_replay_exact assembles those streams beside the ADC values and the replay reads a member's distance
 * there instead of its row.  A member without a travelled distance fails the call (never silent). */
int gamma_hip_ivfpq_shard_export_exact(gamma_hip_index* h, const gamma_hip_search_params* p, int nf, const float* d_xf,
                                       const float* d_vals, const int64_t* d_ids, const int32_t* d_off, int64_t stride,
                                       const float* d_bound_f, float* d_ex);
int gamma_hip_ivfpq_merge_replay_exact(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nf,
                                       const float* d_x_slice, int64_t stride, const float* d_vals_all, const int64_t* d_ids_all,
                                       const int32_t* d_off_all, const float* d_ex_all, int k, const int32_t* d_list,
                                       float* d_distances, int64_t* d_labels);
/* sharded search, stage 2: merge nshards*recall_num candidates per query (layout
 * [shard][nq][recall_num]) into the global top-recall_num, then compute_dis (re-rank or
 * truncate, gamma_index_ivfpq.cc:642-697) for queries [q0, q0+nq_local) */
int gamma_hip_ivfpq_merge_rerank(gamma_hip_index* h, const gamma_hip_search_params* p,
                                 int nshards, int nq, const float* d_x, int k,
                                 const float* d_all_dis, const int64_t* d_all_ids, int q0,
                                 int nq_local, float* d_distances, int64_t* d_labels);

/* replaces GammaFLATIndex::Search (gamma_index_flat.cc:118-300) over the raw store */
int gamma_hip_flat_search(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                          const float* x, int k, float* distances, int64_t* labels);
int gamma_hip_flat_search_device(gamma_hip_index* h, const gamma_hip_search_params* p, int nq,
                                 const float* d_x, int k, float* d_distances, int64_t* d_labels);

/* Test hook: n keys (smaller is better; payload = position) through ONE heap of k <= 1024 entries with one form of the
 * device's sifts -- op 0: heap_replace_top stream through the pipelined walk (csrc/heap_dev.h HeapWalk), 1: heap_pop +
 * heap_push through ParHeap (all lanes per sift), 2: the same through the sequential forms, 3: heap_replace_top through
 * ParHeap's sift, 4: faiss's ReservoirTopN (what knn_L2sqr collects through from 100 results on; arr = sorted = its
 * to_result output, n >= 1).  arr_*: the heap ARRAY when the stream is through ((FLT_MAX, -1) = empty); sorted_*: after heap_reorder.
 * tests/test_gpu_heaps.py compares them with the oracle's heaps (= the compiled library's, faiss:utils/Heap.h). */
int gamma_hip_debug_heap_stream(gamma_hip_index* h, int op, int k, int n, const float* vals, float* arr_vals, int32_t* arr_ids,
                                float* sorted_vals, int32_t* sorted_ids);
/* Exact ties across list shards.  gamma_hip_ivfpq_merge_rerank (exact ties on) leaves a list of the slice's queries whose
 * result a tie can change: two equal exact distances among the first k + 1 (or equal ADC distances among the taken ones
 * without re-rank), or the cut of the merged top-recall_num going through a group of equal distances (also when a shard's
 * own cut may have dropped members of the group).  Those are replayed through the reference's heaps over the stream its
 * scanner saw (faiss:utils/Heap.h:103-131, gamma_index_ivfpq.h:363-369, gamma_index_ivfpq.cc:664-696) -- which is
 * spread over the shards:
 *   shard_cut_flags after a shard search of nq queries: d_flags[q] != 0 = the shard's own top-recall_num cut went through
 *                   a group of equal distances (it may have dropped members of the group);
 *   merge_set_shard_flags  before merge_rerank: those flags of every shard for the merge's queries, [nshards][nq] (the
 *                   layout of d_all_dis without the recall_num axis); without them a shard table that ends at the merged
 *                   cut value is assumed to have cut a tie (more queries replayed, same results);
 *   merge_flagged   n_flagged (host; waits for the handle's stream) and the device list of flagged slice-local query
 *                   indices (valid until the next search-type call on the handle);
 *   gather_rows     d_dst[i] = row d_list[i] of d_src (rows of row_words 32-bit words): the flagged queries' vectors and
 *                   assignment rows for the shards;
 *   shard_export_rows  on every shard: the longest export row of the nf flagged queries (entries of the probed lists this
 *                   shard owns; host value, waits for the stream): the callers take `stride` = the maximum over the shards;
 *   shard_export    on EVERY shard: for the nf flagged queries (vectors d_xf, assignment d_cdis_f / d_probe_f [nf][nprobe])
 *                   the ADC distances (+-inf = filtered entry) and vector ids of the probed lists this shard owns, in list
 *                   order: rows of `stride` entries, d_off[f][p .. p + 1) = the entries of probe p (empty for a list of
 *                   another shard);
 *   merge_replay    on the slice's owner: the exports of all shards ([nshards][nf][stride], [nshards][nf][nprobe + 1])
 *                   are assembled probe by probe into the streams and replayed; rows d_list[f] of d_distances /
 *                   d_labels (the slice's result rows) are rewritten.  d_x_slice: the slice's query vectors. */
int gamma_hip_ivfpq_shard_cut_flags(gamma_hip_index* h, int nq, uint8_t* d_flags);
int gamma_hip_ivfpq_merge_set_shard_flags(gamma_hip_index* h, const uint8_t* d_flags);
int gamma_hip_ivfpq_merge_flagged(gamma_hip_index* h, int* n_flagged, const int32_t** d_list);
int gamma_hip_gather_rows(gamma_hip_index* h, const void* d_src, int row_words, const int32_t* d_list, int n, void* d_dst);
int gamma_hip_ivfpq_max_list_len(gamma_hip_index* h);
int gamma_hip_ivfpq_shard_export_rows(gamma_hip_index* h, const gamma_hip_search_params* p, int nf, const int32_t* d_probe_f,
                                      int64_t* max_entries);
int gamma_hip_ivfpq_shard_export(gamma_hip_index* h, const gamma_hip_search_params* p, int nf, const float* d_xf,
                                 const float* d_cdis_f, const int32_t* d_probe_f, int64_t stride, float* d_vals,
                                 int64_t* d_ids, int32_t* d_off);
int gamma_hip_ivfpq_merge_replay(gamma_hip_index* h, const gamma_hip_search_params* p, int nshards, int nf,
                                 const float* d_x_slice, int64_t stride, const float* d_vals_all, const int64_t* d_ids_all,
                                 const int32_t* d_off_all, int k, const int32_t* d_list, float* d_distances,
                                 int64_t* d_labels);

/* ---- several GPUs in ONE process: a group of handles, the index sharded by IVF list ----------------------------
 * What the reference's GPU model does with faiss's IndexShards / GpuClonerOptions (index/impl/gpu/gamma_gpu_cloner.cpp:200-269,
 * index/impl/gpu/gamma_index_ivfpq_gpu.cc:356-436, faiss:IndexShards.cpp:283-345): one index object, a host thread per
 * GPU, every search fanned out and merged.  Here member i is an ordinary handle on devices[i] (several members may
 * share a device: tests) that owns the lists owner(l) == i; centroids, codebooks, T2, the delete bitmap and the raw
 * vectors are replicated (broadcast the replicated state yourself through gamma_hip_group_member, as the plugins do).
 * A search: every member runs the coarse quantizer for its slice of the queries, the assignment is exchanged
 * (device-to-device copies), every member scans the probed lists it owns for the whole batch and keeps a local
 * top-recall_num, each member pulls the candidates of ITS query slice from all members, merges them (k_merge_shards) and
 * runs compute_dis, then the tie phase above (flagged queries' streams exported by every member, replayed at the owner).
 * Results equal those of one handle holding every list.  The multi-process form of the same steps over RCCL is gamma_amd/dist.py.
 * Threading: group-level calls (search, add, update, delete, compaction) may come from any thread but run ONE AT A TIME
 * (a call occupies every member's worker thread and their barriers): concurrent client threads queue behind each other
 * and behind the indexing thread, and small calls are not combined -- batch queries above the group, as the reference's
 * GPU model does (its GPUItem queue, index/impl/gpu/gamma_index_ivfpq_gpu.cc:356-436).  A member that fails in any phase
 * ends the call with that member's error on every thread (the go / no-go of each phase is one snapshot all members share). */
typedef struct gamma_hip_group gamma_hip_group;
int gamma_hip_group_create(const int* devices, int n, gamma_hip_group** out);
int gamma_hip_group_destroy(gamma_hip_group* g);
int gamma_hip_group_size(const gamma_hip_group* g);
gamma_hip_index* gamma_hip_group_member(gamma_hip_group* g, int i);
const char* gamma_hip_group_last_error(gamma_hip_group* g);
/* Transport of the two exchanges of a sharded search (the assignment, the per-shard top-recall_num tables of every owner's
 * slice).  0 (default): device-to-device copies ordered by events and host barriers.  1: RCCL over xGMI -- ONE in-place
 * ncclAllGather of the assignment and one grouped ncclSend / ncclRecv exchange of the tables, one communicator per member
 * (ncclCommInitAll), resolved from librccl.so at run time; formed at the next sharded search, and only when every member
 * has a device of its own (otherwise, or when the library cannot be loaded, the group stays on copies and says why in
 * gamma_hip_group_transport_note).  The environment variable GAMMA_HIP_GROUP_RCCL=1 selects 1 for every new group.
 * gamma_hip_group_transport: out2 = {1 when a communicator is in use, searches whose exchanges went through RCCL}.
 * The rare tie-phase exports stay on copies.  Results do not depend on the transport. */
int gamma_hip_group_set_transport(gamma_hip_group* g, int rccl);
int gamma_hip_group_transport(gamma_hip_group* g, int64_t* out2);
const char* gamma_hip_group_transport_note(gamma_hip_group* g);
/* list -> member.  weights[nlist] (expected or actual list sizes: the training set's assignment counts, the sizes in a
 * dump) are balanced greedily, heaviest list first -- probe popularity follows list size, so this balances scan
 * bytes; NULL: l mod n.  Call once, after gamma_hip_ivfpq_init on every member and before the first Add. */
int gamma_hip_group_set_owners(gamma_hip_group* g, const int64_t* weights);
/* Placement, before gamma_hip_group_set_owners.  0 (default): sharded by list as above.  1: REPLICATED -- every member
 * holds every list (Add / AddKeys / Update / Delete go to all members, one encode), a search splits the QUERIES: member
 * i answers its slice with the ordinary single-handle search, so results are those of one handle bit for bit (exact
 * ties included) and nothing is exchanged but the slice and its k results.  The choice for an index that is small next
 * to one GPU's memory: list sharding leaves the per-query fixed work (table, bound, selection) on every member for
 * every query, replicas divide it (multi-GPU).  The per-list getters read member 0. */
int gamma_hip_group_set_placement(gamma_hip_group* g, int replicate);
int gamma_hip_group_placement(const gamma_hip_group* g);
int gamma_hip_group_owner(const gamma_hip_group* g, int list_no);
/* GammaIVFPQIndex::Add: ONE encode (members take turns), then AddKeys at the owner of every assigned list */
int gamma_hip_group_ivfpq_add(gamma_hip_group* g, int64_t n, const float* vecs, int64_t first_vid);
/* AddKeys / list read-back at the owner (Load / Dump) */
int gamma_hip_group_ivfpq_add_keys(gamma_hip_group* g, int list_no, int n, const int64_t* vids, const uint8_t* codes);
int64_t gamma_hip_group_ivfpq_list_size(gamma_hip_group* g, int list_no);
int gamma_hip_group_ivfpq_get_list(gamma_hip_group* g, int list_no, int64_t* vids, uint8_t* codes);
/* GammaIVFPQIndex::Update: one encode; the member holding the old entry flags it, the owner of the new list appends */
int gamma_hip_group_ivfpq_update(gamma_hip_group* g, int n, const int64_t* vids, const float* vecs);
int gamma_hip_group_ivfpq_delete(gamma_hip_group* g, const int64_t* vids, int n);
int gamma_hip_group_ivfpq_compact_if_need(gamma_hip_group* g);
/* GammaIVFPQIndex::Search over all members; host buffers as gamma_hip_ivfpq_search */
int gamma_hip_group_ivfpq_search(gamma_hip_group* g, const gamma_hip_search_params* p, int nq, const float* x, int k,
                                 float* distances, int64_t* labels);
/* the same with the queries and the result buffers in the memory of member 0's device; returns when the results are there */
int gamma_hip_group_ivfpq_search_device(gamma_hip_group* g, const gamma_hip_search_params* p, int nq, const float* d_x,
                                        int k, float* d_distances, int64_t* d_labels);
int64_t gamma_hip_group_total_mem_bytes(gamma_hip_group* g);

/* ---- accounting (GetTotalMemBytes, index/retrieval_model.h:287; PerfTool :23-50) ------ */
int64_t gamma_hip_total_mem_bytes(gamma_hip_index* h);
/* per-stage HIP-event timing of the kernels launched by search calls */
#define GAMMA_HIP_STAGE_COARSE 0
#define GAMMA_HIP_STAGE_TABLES 1
#define GAMMA_HIP_STAGE_SCAN 2
#define GAMMA_HIP_STAGE_SELECT 3
#define GAMMA_HIP_STAGE_RERANK 4
#define GAMMA_HIP_STAGE_FLAT 5
#define GAMMA_HIP_NUM_STAGES 6
/* on: 0 off; 1 every stage, plus the counter of scanned codes behind gamma_hip_profile_scan_bytes; 2 the scan stage alone --
 * two events around the scan launch and nothing else.  (Under rocprofv3's kernel trace an event pair between two kernels
 * shows as a 10-16 us gap; without the tracer the call rate is the same with 0, 1 and 2 -- bench.py --timed-events, 60
 * steps of 16384 queries each: 1.635 / 1.629 / 1.642 ms.) */
int gamma_hip_profile_enable(gamma_hip_index* h, int on);
int gamma_hip_profile_reset(gamma_hip_index* h);
/* total milliseconds and number of timed launches of a stage since the last reset;
 * scan_bytes = algorithmic list-scan bytes (sum over probed pairs of len*code_size) */
int gamma_hip_profile_get(gamma_hip_index* h, int stage, double* total_ms, int64_t* launches);
int gamma_hip_profile_scan_bytes(gamma_hip_index* h, int64_t* bytes, int64_t* pairs);

#ifdef __cplusplus
}
#endif
#endif /* GAMMA_HIP_H_ */
