// GammaIVFPQHIPIndex -- see gamma_index_ivfpq_hip.h.  Host side only: parameter handling,
// training driver, bookkeeping; every distance / scan / selection runs in libgamma_hip.so.
#include "gamma_index_ivfpq_hip.h"

#include "iwpq_io.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/stat.h>

#include <algorithm>
#include <random>

namespace tig_gamma {

REGISTER_MODEL(HIPIVFPQ, GammaIVFPQHIPIndex);

#define HLOG(...)                      \
  do {                                 \
    fprintf(stderr, "[HIPIVFPQ] ");    \
    fprintf(stderr, __VA_ARGS__);      \
    fprintf(stderr, "\n");             \
  } while (0)

int HIPIVFPQModelParams::Parse(const char *str) {
  utils::JsonParser jp;
  if (jp.Parse(str)) {
    HLOG("parse IVFPQ retrieval parameters error: %s", str);
    return -1;
  }
  int v;
  if (!jp.GetInt("ncentroids", v)) {
    if (v < -1) return -1;
    if (v > 0) ncentroids = v;
  } else {
    HLOG("cannot get ncentroids for ivfpq, set it when create space");
    return -1;
  }
  if (!jp.GetInt("nsubvector", v)) {
    if (v < -1) return -1;
    if (v > 0) nsubvector = v;
  } else {
    HLOG("cannot get nsubvector for ivfpq, set it when create space");
    return -1;
  }
  if (!jp.GetInt("nbits_per_idx", v)) {
    if (v < -1) return -1;
    if (v > 0) nbits_per_idx = v;
  }
  if (!jp.GetInt("nprobe", v)) {
    if (v < -1) return -1;
    if (v > 0) nprobe = v;
    if (nprobe > ncentroids) {
      HLOG("nprobe should less than ncentroids");
      return -1;
    }
  }
  if (!jp.GetInt("support_indivisible_nsubvector", v)) support_indivisible_nsubvector = v != 0;
  if (!jp.GetInt("device_filters", v)) device_filters = v != 0;
  if (!jp.GetInt("exact_ties", v)) exact_ties = v != 0;
  if (!jp.GetInt("perf_stages", v)) perf_stages = v != 0;
  std::string devs;
  if (!jp.GetString("devices", devs)) {   // "0,1,2,3" (the engine's JsonParser has no arrays)
    devices.clear();
    size_t at = 0;
    while (at < devs.size()) {
      size_t end = devs.find(',', at);
      if (end == std::string::npos) end = devs.size();
      const std::string tok = devs.substr(at, end - at);
      char *rest = nullptr;
      const long dv = strtol(tok.c_str(), &rest, 10);
      if (tok.empty() || (rest && *rest != '\0' && *rest != ' ') || dv < 0) {
        HLOG("invalid devices = %s", devs.c_str());
        return -1;
      }
      devices.push_back((int)dv);
      at = end + 1;
    }
  } else if (!jp.GetInt("devices", v)) {   // a single ordinal
    if (v < 0) return -1;
    devices.assign(1, v);
  }
  std::string plc;
  if (!jp.GetString("placement", plc)) {   // several devices: "shard" (by IVF list, the default) | "replicate"
    if (strcasecmp("shard", plc.c_str()) && strcasecmp("replicate", plc.c_str())) {
      HLOG("invalid placement = %s", plc.c_str());
      return -1;
    }
    replicate = !strcasecmp("replicate", plc.c_str());
  }
  if (!jp.GetInt("bucket_init_size", v)) {
    if (v < -1) return -1;
    if (v > 0) bucket_init_size = v;
  }
  if (!jp.GetInt("bucket_max_size", v)) {
    if (v < -1) return -1;
    if (v > 0) bucket_max_size = v;
  }
  std::string mt;
  if (!jp.GetString("metric_type", mt)) {
    if (strcasecmp("L2", mt.c_str()) && strcasecmp("InnerProduct", mt.c_str())) {
      HLOG("invalid metric_type = %s", mt.c_str());
      return -1;
    }
    metric_type = !strcasecmp("L2", mt.c_str()) ? DistanceComputeType::L2 : DistanceComputeType::INNER_PRODUCT;
  }
  utils::JsonParser sub;
  has_hnsw = !jp.GetObject("hnsw", sub);
  has_opq = !jp.GetObject("opq", sub);
  if (ncentroids <= 0 || nsubvector <= 0 || nbits_per_idx <= 0) return -1;
  return 0;
}

GammaIVFPQHIPIndex::GammaIVFPQHIPIndex() {}

GammaIVFPQHIPIndex::~GammaIVFPQHIPIndex() {
  if (grp_) gamma_hip_group_destroy(grp_);   // owns its members, h_ among them
  else if (h_) gamma_hip_destroy(h_);
  delete model_param_;
}

// one handle, or a group of handles with the lists sharded by owner (gamma_hip_group_*; what the reference's GPU model
// does with IndexShards, index/impl/gpu/gamma_gpu_cloner.cpp:200-269)
int GammaIVFPQHIPIndex::OpenDevices(const std::vector<int> &devices, bool replicate) {
  if (devices.size() > 1) {
    int rc = gamma_hip_group_create(devices.data(), (int)devices.size(), &grp_);
    if (!rc) rc = gamma_hip_group_set_placement(grp_, replicate ? 1 : 0);
    if (rc) {
      HLOG("gamma_hip_group_create failed: %s", gamma_hip_strerror(rc));
      return -1;
    }
    for (int i = 0; i < gamma_hip_group_size(grp_); i++) members_.push_back(gamma_hip_group_member(grp_, i));
    h_ = members_[0];
    return 0;
  }
  const char *dev = getenv("GAMMA_HIP_DEVICE");
  const int ordinal = devices.size() == 1 ? devices[0] : (dev ? atoi(dev) : 0);
  int rc = gamma_hip_create(ordinal, &h_);
  if (rc) {
    HLOG("gamma_hip_create failed: %s", gamma_hip_strerror(rc));
    return -1;
  }
  members_.assign(1, h_);
  return 0;
}

int GammaIVFPQHIPIndex::Init(const std::string &model_parameters, int indexing_size) {
  indexing_size_ = indexing_size;
  model_param_ = new HIPIVFPQModelParams();
  HIPIVFPQModelParams &pa = *model_param_;
  if (model_parameters != "" && pa.Parse(model_parameters.c_str())) return -1;
  if (!vector_) {
    HLOG("vector_ must be set before Init");
    return -1;
  }
  d_ = vector_->MetaInfo()->Dimension();
  if (d_ % pa.nsubvector != 0) {
    HLOG("Dimension [%d] cannot divide by nsubvector [%d] (support_indivisible_nsubvector is not "
         "available on the HIP path)", d_, pa.nsubvector);
    return -2;
  }
  if (pa.has_hnsw || pa.has_opq || pa.support_indivisible_nsubvector || pa.nbits_per_idx != 8) {
    HLOG("hnsw / opq / padded dimensions / nbits_per_idx != 8 are not supported by HIPIVFPQ");
    return -2;
  }
  nlist_ = pa.ncentroids;
  M_ = pa.nsubvector;
  metric_type_ = pa.metric_type;
  nprobe_ = pa.nprobe;
  if (pa.devices.size() > 1 && pa.device_filters) {
    HLOG("device_filters with several devices is not supported (the columns live on one handle)");
    return -2;
  }
  if (OpenDevices(pa.devices, pa.replicate)) return -1;
  int rc = ForAll([&](gamma_hip_index *m) {
    int r = gamma_hip_ivfpq_init(m, d_, nlist_, M_, 8,
                                 metric_type_ == DistanceComputeType::L2 ? GAMMA_HIP_METRIC_L2 : GAMMA_HIP_METRIC_IP,
                                 pa.bucket_init_size, pa.bucket_max_size);
    if (!r) r = gamma_hip_raw_init(m, d_);
    if (!r) r = gamma_hip_set_exact_ties(m, pa.exact_ties ? 1 : 0);
    if (!r && pa.perf_stages) r = gamma_hip_profile_enable(m, 1);
    return r;
  });
  if (rc) {
    HLOG("device init failed: %s (%s)", gamma_hip_strerror(rc), gamma_hip_last_error(h_));
    return -1;
  }
  // what IndexIVFPQ::precompute_table prints in verbose mode (faiss:IndexIVFPQ.cpp:443-448): never silent
  if (gamma_hip_ivfpq_use_precomputed_table(h_) == 0)
    HLOG("not precomputing table, it would be too big: %lld bytes (max %lld) -- L2 searches score with residual tables "
         "(use_precomputed_table = 0)", (long long)nlist_ * M_ * 1024, (long long)gamma_hip_get_precomputed_table_max_bytes());
  return 0;
}

RetrievalParameters *GammaIVFPQHIPIndex::Parse(const std::string &parameters) {
  if (parameters == "") return new HIPIVFPQRetrievalParameters(metric_type_);
  utils::JsonParser jp;
  if (jp.Parse(parameters.c_str())) {
    HLOG("parse retrieval parameters error: %s", parameters.c_str());
    return nullptr;
  }
  HIPIVFPQRetrievalParameters *rp = new HIPIVFPQRetrievalParameters();
  std::string mt;
  if (!jp.GetString("metric_type", mt)) {
    rp->SetDistanceComputeType(!strcasecmp("L2", mt.c_str()) ? DistanceComputeType::L2
                                                              : DistanceComputeType::INNER_PRODUCT);
  } else {
    rp->SetDistanceComputeType(metric_type_);
  }
  int v;
  if (!jp.GetInt("recall_num", v) && v > 0) rp->SetRecallNum(v);
  if (!jp.GetInt("nprobe", v) && v > 0) rp->SetNprobe(v);
  if (!jp.GetInt("parallel_on_queries", v)) rp->SetParallelOnQueries(v != 0);
  if (!jp.GetInt("exact_ties", v)) rp->SetExactTies(v != 0 ? 1 : -1);   // HIP only: this request's choice
  return rp;
}

// Training = faiss's, on the device (gamma_hip_kmeans: faiss::Clustering::train -- subsampling, rand_perm seeds,
// centroid sums, empty-cluster splits -- with the assignment and the sums on the GPU, the training set resident).
int GammaIVFPQHIPIndex::TrainCoarse(size_t num, const float *xt) {
  // IndexIVF::train_q1: Clustering(d, nlist, cp) with cp.niter = 10 (gamma_index_ivfpq.cc:175), seed 1234,
  // max_points_per_centroid 256
  coarse_centroids_.resize((size_t)nlist_ * d_);
  return gamma_hip_kmeans(h_, d_, (int64_t)num, xt, nlist_, 10, 1234, 256, coarse_centroids_.data(), nullptr);
}

int GammaIVFPQHIPIndex::TrainOnHost(size_t num, const float *xt) {
  // IndexIVFPQ::train (train_q1 + train_residual_o + ProductQuantizer::train) on the device: gamma_hip_ivfpq_train
  coarse_centroids_.resize((size_t)nlist_ * d_);
  pq_centroids_.resize((size_t)M_ * 256 * (d_ / M_));
  return gamma_hip_ivfpq_train(h_, d_, (int64_t)num, xt, nlist_, M_, coarse_centroids_.data(), pq_centroids_.data());
}

int GammaIVFPQHIPIndex::Indexing() {
  if (is_trained_) {
    HLOG("already trained, skip indexing");
    return 0;
  }
  std::vector<float> xt;
  size_t num = 0;
  if (TrainingSet(xt, num)) return -1;
  int rc = TrainOnHost(num, xt.data());
  if (!rc) rc = ForAll([&](gamma_hip_index *m) {
    return gamma_hip_ivfpq_set_trained(m, coarse_centroids_.data(), pq_centroids_.data(), nullptr);
  });
  if (!rc && grp_) {
    // list -> GPU by the training set's list sizes, balanced greedily: the lists never move afterwards
    std::vector<int32_t> assign(num);
    rc = gamma_hip_assign(h_, d_, (int64_t)num, xt.data(), nlist_, coarse_centroids_.data(), assign.data(), nullptr);
    std::vector<int64_t> weight(nlist_, 0);
    for (size_t i = 0; i < num && !rc; i++)
      if (assign[i] >= 0 && assign[i] < nlist_) weight[assign[i]]++;
    if (!rc) rc = gamma_hip_group_set_owners(grp_, weight.data());
  }
  if (rc) {
    HLOG("training failed: %s (%s)", gamma_hip_strerror(rc), gamma_hip_last_error(h_));
    return -1;
  }
  is_trained_ = true;
  return 0;
}

// the first `num` vectors of the store, num by the rule of gamma_index_ivfpq.cc:280-301 (= gamma_index_ivfflat.cc:252-276)
int GammaIVFPQHIPIndex::TrainingSet(std::vector<float> &xt, size_t &num) {
  const size_t vectors_count = vector_->MetaInfo()->Size();
  if ((size_t)indexing_size_ < (size_t)nlist_) num = (size_t)nlist_ * 39;
  else if ((size_t)indexing_size_ <= (size_t)nlist_ * 256) num = (size_t)indexing_size_;
  else num = (size_t)nlist_ * 256;
  if (num > vectors_count) {
    HLOG("vector total count [%zu] less then index_size[%zu], failed!", vectors_count, num);
    return -1;
  }
  std::vector<int64_t> vids(num);
  for (size_t i = 0; i < num; i++) vids[i] = (int64_t)i;
  ScopeVectors sv;
  if (vector_->Gets(vids, sv)) return -1;
  xt.resize(num * d_);
  for (size_t i = 0; i < num; i++) memcpy(&xt[i * d_], sv.Get((int)i), sizeof(float) * d_);
  return 0;
}

bool GammaIVFPQHIPIndex::Add(int n, const uint8_t *vec) {
  // vids are consecutive from indexed_vec_count_ (gamma_index_ivfpq.cc:475-489); the raw
  // vectors are mirrored to HBM for the exact re-rank (VectorReader::Gets on the CPU path).
  // The mirror is written at explicit rows (gamma_hip_raw_write): a brute-force Search that mirrors the
  // same rows at the same time (EnsureRaw) rewrites identical bytes instead of appending them twice.
  const float *v = reinterpret_cast<const float *>(vec);
  const int64_t end = (int64_t)indexed_vec_count_ + n;
  if (EnsureRaw(indexed_vec_count_)) return false;
  {
    std::lock_guard<std::mutex> g(raw_mu_);
    if (ForAll([&](gamma_hip_index *m) { return gamma_hip_raw_write(m, indexed_vec_count_, n, v); })) return false;
    raw_uploaded_ = std::max(raw_uploaded_, end);
  }
  if (SyncVid2DocID(end)) return false;   // before AddKeys: it counts vectors of deleted DOCS (realtime_mem_data.cc:294)
  int rc = grp_ ? gamma_hip_group_ivfpq_add(grp_, n, v, indexed_vec_count_) : gamma_hip_ivfpq_add(h_, n, v, indexed_vec_count_);
  if (rc) {
    HLOG("add failed: %s (%s)", gamma_hip_strerror(rc), grp_ ? gamma_hip_group_last_error(grp_) : gamma_hip_last_error(h_));
    return false;
  }
  indexed_vec_count_ += n;
  return true;
}

int GammaIVFPQHIPIndex::Update(const std::vector<int64_t> &ids, const std::vector<const uint8_t *> &vecs) {
  // the engine drains up to 20 000 updated vids per pass (vector/vector_manager.cc:355-380): ONE encode of the batch
  // (each vector assigned as quantizer->assign(1, ..) assigns it, gamma_index_ivfpq.cc:398), the list updates in
  // order, one publish -- instead of two device round trips per vid
  const size_t n = ids.size();
  if (n == 0) return 0;
  if (vecs.size() != n) return -1;
  std::vector<float> x(n * (size_t)d_);
  for (size_t i = 0; i < n; i++) memcpy(&x[i * d_], vecs[i], sizeof(float) * d_);
  int rc = grp_ ? gamma_hip_group_ivfpq_update(grp_, (int)n, ids.data(), x.data())
                : gamma_hip_ivfpq_update_batch(h_, (int)n, ids.data(), x.data());
  if (rc) {
    HLOG("update failed: %s (%s)", gamma_hip_strerror(rc), grp_ ? gamma_hip_group_last_error(grp_) : gamma_hip_last_error(h_));
    return -1;
  }
  {
    std::lock_guard<std::mutex> g(raw_mu_);   // rows the mirror has not reached yet are skipped: EnsureRaw brings them
    if (ForAll([&](gamma_hip_index *m) { return gamma_hip_raw_update_batch(m, (int64_t)n, ids.data(), x.data()); })) return -1;
  }
  if (grp_) gamma_hip_group_ivfpq_compact_if_need(grp_);
  else gamma_hip_ivfpq_compact_if_need(h_);   // gamma_index_ivfpq.cc:420
  if (model_param_ && model_param_->device_filters) {
    // the engine updated these docs (table first, then this model): the device mirror of their scalar fields follows
    std::vector<int64_t> docs(ids);
    RawVector *rv = dynamic_cast<RawVector *>(vector_);
    if (rv && rv->VidMgr() && rv->VidMgr()->MultiVids())
      for (size_t i = 0; i < docs.size(); i++) docs[i] = rv->VidMgr()->VID2DocID((int)ids[i]);
    if (columns_.Refresh(h_, docs)) return -1;
  }
  return 0;
}

int GammaIVFPQHIPIndex::Delete(const std::vector<int64_t> &ids) {
  // the engine sets the doc bit in its BitmapManager (search/gamma_engine.cc:810-812); the
  // device keeps a mirror of that bitmap, fed from here (vid == docid for single-vector docs)
  if (ids.empty()) return 0;
  // the bitmap is on DOC ids (VIDMgr::VID2DocID; the identity for single-vector documents)
  std::vector<int64_t> docs(ids);
  RawVector *rv = dynamic_cast<RawVector *>(vector_);
  if (rv && rv->VidMgr() && rv->VidMgr()->MultiVids())
    for (size_t i = 0; i < docs.size(); i++) docs[i] = rv->VidMgr()->VID2DocID((int)ids[i]);
  if (ForAll([&](gamma_hip_index *m) { return gamma_hip_bitmap_set(m, docs.data(), (int64_t)docs.size(), 1); })) return -1;
  if (grp_) return gamma_hip_group_ivfpq_delete(grp_, ids.data(), (int)ids.size()) ? -1 : 0;
  return gamma_hip_ivfpq_delete(h_, ids.data(), (int)ids.size()) ? -1 : 0;
}

int GammaIVFPQHIPIndex::Search(RetrievalContext *retrieval_context, int n, const uint8_t *x, int k,
                               float *distances, int64_t *ids) {
  HIPIVFPQRetrievalParameters *rp = dynamic_cast<HIPIVFPQRetrievalParameters *>(retrieval_context->RetrievalParams());
  HIPIVFPQRetrievalParameters defaults;
  if (rp == nullptr) rp = &defaults;
  GammaSearchCondition *cond = dynamic_cast<GammaSearchCondition *>(retrieval_context);
  gamma_hip_search_params p;
  memset(&p, 0, sizeof(p));
  p.metric = rp->GetDistanceComputeType() == DistanceComputeType::INNER_PRODUCT ? GAMMA_HIP_METRIC_IP
                                                                                : GAMMA_HIP_METRIC_L2;
  p.recall_num = rp->RecallNum();
  p.has_rank = cond ? (cond->has_rank ? 1 : 0) : 1;
  p.min_score = cond ? cond->min_score : std::numeric_limits<float>::min();
  p.max_score = cond ? cond->max_score : std::numeric_limits<float>::max();
  p.coarse_mode = -1;
  p.exact_ties = rp->ExactTies();
  std::vector<gamma_hip_range_filter> rf;
  std::vector<gamma_hip_field_filter> ff;
  std::vector<gamma_hip_term_filter> tf;
  // scalar filters: on the device against mirrored columns when asked for and possible, else the request's
  // flattened docid bitmaps as the CPU models take them
  if (!(model_param_ && model_param_->device_filters &&
        columns_.Prepare(h_, cond, DocCountOf(this, (int64_t)vector_->MetaInfo()->Size()), p, ff, tf)))
    FillRangeFilters(cond, p, rf);
  const float *xq = reinterpret_cast<const float *>(x);
  int rc;
  if ((cond && cond->brute_force_search) || !is_trained_) {
    if (EnsureRaw((int64_t)vector_->MetaInfo()->Size())) return -1;
    rc = gamma_hip_flat_search(h_, &p, n, xq, k, distances, ids);   // gamma_index_ivfpq.cc:529-537
  } else {
    // nprobe rule of gamma_index_ivfpq.cc:539-545
    p.nprobe = (rp->Nprobe() > 0 && rp->Nprobe() <= nlist_) ? rp->Nprobe() : nprobe_;
    rc = grp_ ? gamma_hip_group_ivfpq_search(grp_, &p, n, xq, k, distances, ids)
              : gamma_hip_ivfpq_search(h_, &p, n, xq, k, distances, ids);
  }
  PerfLabels(cond);
  if (rc) {
    HLOG("search failed: %s (%s)", gamma_hip_strerror(rc), grp_ ? gamma_hip_group_last_error(grp_) : gamma_hip_last_error(h_));
    return rc;
  }
  WarnTiesNotHonoured(h_, grp_, &ties_said_);
  WarnBlasCorners(h_, grp_, &blas_said_);
  return 0;
}

// A search beyond the exact-ties mode's range (nprobe > 1024, a flat search for k = 4096) that merely inherited the model's
// default ran with the (distance, position) order inside ties: said once per new occurrence count of THIS model (the
// high-water mark is a member of the index object; a group's count is the sum over its members), never silently
// (gamma_hip_ties_not_honoured; a request that sets "exact_ties": 1 itself fails instead).  The same line reports repack
// read-backs that differed from their source (gamma_hip_ivfpq_repack_verify_stats).
void WarnTiesNotHonoured(gamma_hip_index *h, gamma_hip_group *grp, std::atomic<int64_t> *said) {
  int64_t n = 0, bad_repacks = 0;
  const int members = grp ? gamma_hip_group_size(grp) : 1;
  for (int i = 0; i < members; i++) {
    gamma_hip_index *m = grp ? gamma_hip_group_member(grp, i) : h;
    int64_t c = 0, rv[2] = {0, 0};
    if (m && !gamma_hip_ties_not_honoured(m, &c, 0)) n += c;
    if (m && !gamma_hip_ivfpq_repack_verify_stats(m, rv)) bad_repacks += rv[1];
  }
  const int64_t mark = n + (bad_repacks << 40);
  if (mark <= 0 || !said) return;
  int64_t prev = said->load();
  if (mark > prev && said->compare_exchange_strong(prev, mark)) {
    if (n > 0)
      HLOG("%lld search call(s) ran without the reference's heap order inside exact ties: shape beyond the mode's range "
           "(nprobe > 1024 or flat k = 4096)", (long long)n);
    if (bad_repacks > 0)
      HLOG("%lld list-arena repack(s) did not read back as written: the previous arena was kept and the move repeated into "
           "ordinary allocations", (long long)bad_repacks);
  }
}

// Calls whose GEMM-form coarse distances fell into a shape for which the library's sgemm_ kernel is not restated
// (gamma_hip_blas_form_not_restated: ulp-level differences in a few coarse distances): logged when the count first
// becomes non-zero and at every doubling -- an Add stream of odd batch sizes would otherwise fill the log.
void WarnBlasCorners(gamma_hip_index *h, gamma_hip_group *grp, std::atomic<int64_t> *said) {
  int64_t n = 0;
  const int members = grp ? gamma_hip_group_size(grp) : 1;
  for (int i = 0; i < members; i++) {
    gamma_hip_index *m = grp ? gamma_hip_group_member(grp, i) : h;
    int64_t c = 0;
    if (m && !gamma_hip_blas_form_not_restated(m, &c, 0)) n += c;
  }
  if (n <= 0 || !said) return;
  int64_t prev = said->load();
  if (n >= 2 * prev && n > prev && said->compare_exchange_strong(prev, n))
    HLOG("%lld call(s) computed coarse distances in a GEMM shape whose MKL kernel is not restated (K > 768, K = 384 with a "
         "9..512-row remainder, remainder blocks of 1..7 rows): a few coarse distances may differ from faiss's by an ulp",
         (long long)n);
}

// PerfTool (index/retrieval_model.h:23-50; printed by the engine at online_log_level=debug): one label for the device
// call, and with "perf_stages": 1 the device time of every stage of the handle's searches since the last label
// (HIP events, gamma_hip_profile_get; concurrent searches share the counters)
void GammaIVFPQHIPIndex::PerfLabels(GammaSearchCondition *cond) {
  if (!cond || !cond->perf_tool_) return;
  cond->GetPerfTool().Perf("hip search");
  if (!(model_param_ && model_param_->perf_stages) || grp_) return;
  static const char *names[] = {"coarse", "tables", "scan", "select", "rerank", "flat"};
  std::lock_guard<std::mutex> g(perf_mu_);
  for (int st = 0; st < GAMMA_HIP_NUM_STAGES; st++) {
    double ms = 0;
    int64_t launches = 0;
    if (gamma_hip_profile_get(h_, st, &ms, &launches)) continue;
    if (ms > perf_ms_[st]) cond->GetPerfTool().perf_ss << "hip " << names[st] << " [" << ms - perf_ms_[st] << "]ms ";
    perf_ms_[st] = ms;
  }
}

// mirror vids [raw_uploaded_, upto) of the engine's vector store into HBM.  Called from Search (any number of
// client threads: brute force before training), from Add (the indexing thread) and from Load: raw_mu_ makes
// "read the watermark, copy, advance it" one step, and the rows go to their own positions.
int GammaIVFPQHIPIndex::EnsureRaw(int64_t upto) {
  std::lock_guard<std::mutex> g(raw_mu_);
  const int64_t step = 65536;
  for (int64_t i0 = raw_uploaded_; i0 < upto; i0 += step) {
    const int64_t nb = std::min(step, upto - i0);
    std::vector<int64_t> vids(nb);
    for (int64_t i = 0; i < nb; i++) vids[i] = i0 + i;
    ScopeVectors sv;
    if (vector_->Gets(vids, sv)) return -1;
    std::vector<float> buf((size_t)nb * d_);
    for (int64_t i = 0; i < nb; i++) memcpy(&buf[(size_t)i * d_], sv.Get((int)i), sizeof(float) * d_);
    if (ForAll([&](gamma_hip_index *m) { return gamma_hip_raw_write(m, i0, nb, buf.data()); })) return -1;
    raw_uploaded_ = i0 + nb;
  }
  return 0;
}

// the engine's delete bitmap -> the device mirror (after a restart the engine has loaded its bitmap file
// before it calls Load on the models, util/bitmap_manager.cc:96-161; vector_ is a RawVector, raw_vector.h:171)
// multi-vector documents: the device tests the delete bitmap and every filter on the DOC id of a scanned vector
// (include/gamma_hip.h gamma_hip_vid2docid_append); the mapping is the engine's VIDMgr
int GammaIVFPQHIPIndex::SyncVid2DocID(int64_t upto) {
  RawVector *rv = dynamic_cast<RawVector *>(vector_);
  if (!rv || !rv->VidMgr() || !rv->VidMgr()->MultiVids()) return 0;
  const int64_t have = gamma_hip_vid2docid_count(h_);
  if (have < 0) return -1;
  if (upto <= have) return 0;
  std::vector<int32_t> m((size_t)(upto - have));
  for (int64_t v = have; v < upto; v++) m[(size_t)(v - have)] = rv->VidMgr()->VID2DocID((int)v);
  return ForAll([&](gamma_hip_index *mm) { return gamma_hip_vid2docid_append(mm, (int64_t)m.size(), m.data()); });
}

int GammaIVFPQHIPIndex::UploadEngineBitmap() {
  RawVector *rv = dynamic_cast<RawVector *>(vector_);
  if (!rv || !rv->Bitmap() || rv->Bitmap()->BitSize() == 0) return 0;
  return ForAll([&](gamma_hip_index *m) {
    return gamma_hip_bitmap_upload(m, reinterpret_cast<const uint8_t *>(rv->Bitmap()->Bitmap()), (int64_t)rv->Bitmap()->BitSize());
  });
}

int GammaIVFPQHIPIndex::SetTrained(const float *coarse, const float *pq) {
  coarse_centroids_.assign(coarse, coarse + (size_t)nlist_ * d_);
  pq_centroids_.assign(pq, pq + (size_t)M_ * 256 * (d_ / M_));
  if (ForAll([&](gamma_hip_index *m) {
        return gamma_hip_ivfpq_set_trained(m, coarse_centroids_.data(), pq_centroids_.data(), nullptr);
      }))
    return -1;
  // several GPUs: no list sizes to balance yet -- lists are dealt round robin (Indexing balances by the training
  // set's assignment, Load by the dumped sizes)
  if (grp_ && gamma_hip_group_owner(grp_, 0) < 0 && gamma_hip_group_set_owners(grp_, nullptr)) return -1;
  is_trained_ = true;
  return 0;
}

long GammaIVFPQHIPIndex::GetTotalMemBytes() {
  if (grp_) return (long)gamma_hip_group_total_mem_bytes(grp_);
  return h_ ? (long)gamma_hip_total_mem_bytes(h_) : 0;
}

// Dump / Load in the reference's own file format (gamma_index_ivfpq.cc:958-1048): an index dumped by
// the CPU "IVFPQ" model loads here and the other way round (iwpq_io.h for the record layout).
int GammaIVFPQHIPIndex::Dump(const std::string &dir) {
  if (!is_trained_) {
    HLOG("index is not trained, skip dumping");
    return 0;
  }
  const std::string index_dir = dir + "/" + vector_->MetaInfo()->AbsoluteName();
  if (mkdir(index_dir.c_str(), 0755) && errno != EEXIST) {
    HLOG("mkdir error, index dir=%s", index_dir.c_str());
    return -1;
  }
  IwPQFile f;
  f.d = d_;
  f.ntotal = 0;   // GammaIVFPQIndex never advances faiss's ntotal; its dumps carry 0
  f.metric = metric_type_ == DistanceComputeType::INNER_PRODUCT ? 0 : 1;
  f.nlist = (size_t)nlist_;
  f.nprobe = (size_t)nprobe_;
  f.coarse = coarse_centroids_;
  f.by_residual = true;
  f.code_size = (size_t)M_;
  f.M = (size_t)M_;
  f.nbits = 8;
  f.pq = pq_centroids_;
  f.sizes.resize(nlist_);
  f.codes.resize(nlist_);
  f.ids.resize(nlist_);
  for (int l = 0; l < nlist_; l++) {
    const int64_t len = grp_ ? gamma_hip_group_ivfpq_list_size(grp_, l) : gamma_hip_ivfpq_list_size(h_, l);
    if (len < 0) return -1;
    f.sizes[l] = (size_t)len;
    if (len == 0) continue;
    f.ids[l].resize(len);
    f.codes[l].resize((size_t)len * M_);
    if (grp_ ? gamma_hip_group_ivfpq_get_list(grp_, l, f.ids[l].data(), f.codes[l].data())
             : gamma_hip_ivfpq_get_list(h_, l, f.ids[l].data(), f.codes[l].data()))
      return -1;
  }
  if (WriteIwPQ(index_dir + "/ivfpq.index", f)) {
    HLOG("write error, index dir=%s", index_dir.c_str());
    return -1;
  }
  return 0;
}

int GammaIVFPQHIPIndex::Load(const std::string &dir) {
  const std::string path = dir + "/" + vector_->MetaInfo()->AbsoluteName() + "/ivfpq.index";
  FILE *probe = fopen(path.c_str(), "rb");
  if (!probe) {
    HLOG("%s isn't existed, skip loading", path.c_str());
    return 0;   // it should train again after load
  }
  fclose(probe);
  IwPQFile f;
  const int rc = ReadIwPQ(path, &f);
  if (rc) {
    HLOG("cannot read %s (%d)", path.c_str(), rc);
    return -1;
  }
  if (f.d != d_ || (int)f.nlist != nlist_ || (int)f.M != M_ || f.nbits != 8 || (int)f.code_size != M_ ||
      !f.by_residual || f.pq.size() != (size_t)M_ * 256 * (d_ / M_)) {
    HLOG("index file does not match the table's retrieval_param");
    return -1;
  }
  if (grp_) {   // several GPUs: list -> GPU balanced by the dumped list sizes, before the lists come back
    std::vector<int64_t> weight(nlist_);
    for (int l = 0; l < nlist_; l++) weight[l] = (int64_t)f.sizes[l];
    if (gamma_hip_group_set_owners(grp_, weight.data())) return -1;
  }
  if (SetTrained(f.coarse.data(), f.pq.data())) return -1;   // T2 is recomputed, as in the reference
  // deletes that happened before the restart: the bitmap must be in place BEFORE the lists come back, so that
  // AddKeys counts the deleted entries per list as the reference does (realtime_mem_data.cc:293-296)
  if (UploadEngineBitmap()) return -1;
  if (SyncVid2DocID((int64_t)vector_->MetaInfo()->Size())) return -1;
  metric_type_ = f.metric == 0 ? DistanceComputeType::INNER_PRODUCT : DistanceComputeType::L2;
  int64_t count = 0;
  for (int l = 0; l < nlist_; l++) {
    const size_t n = f.sizes[l];
    if (n == 0) continue;
    if (grp_ ? gamma_hip_group_ivfpq_add_keys(grp_, l, (int)n, f.ids[l].data(), f.codes[l].data())
             : gamma_hip_ivfpq_add_keys(h_, l, (int)n, f.ids[l].data(), f.codes[l].data()))
      return -1;
    for (size_t i = 0; i < n; i++)
      if (f.ids[l][i] >= 0) count++;   // bit 63 = superseded by an Update (gamma_index_io.cc:186-189)
  }
  indexed_vec_count_ = (int)count;
  // raw vectors for the re-rank come back from the engine's vector store
  if (EnsureRaw(std::min<int64_t>(indexed_vec_count_, (int64_t)vector_->MetaInfo()->Size()))) return -1;
  return indexed_vec_count_;
}


// ------------------------------------------------------------------------------------------------------------
// HIPIVFFLAT (gamma_index_ivfflat.{h,cc})
// ------------------------------------------------------------------------------------------------------------
REGISTER_MODEL(HIPIVFFLAT, GammaIVFFlatHIPIndex);

int GammaIVFFlatHIPIndex::Init(const std::string &model_parameters, int indexing_size) {
  indexing_size_ = indexing_size;
  model_param_ = new HIPIVFPQModelParams();   // ncentroids / nprobe / metric_type / bucket sizes / device_filters
  HIPIVFPQModelParams &pa = *model_param_;
  // IVFFlatModelParams::Parse (gamma_index_ivfflat.cc:41-101): ncentroids default 2048, nprobe 80, metric InnerProduct
  utils::JsonParser jp;
  if (model_parameters != "" && jp.Parse(model_parameters.c_str())) return -1;
  int v = 0;
  if (jp.Contains("ncentroids")) {
    if (jp.GetInt("ncentroids", v)) return -1;
    if (v > 0) pa.ncentroids = v;
    else if (v != -1) return -1;
  }
  if (!jp.GetInt("nprobe", v)) {
    if (v < -1) return -1;
    if (v > 0) pa.nprobe = v;
    if (pa.nprobe > pa.ncentroids) return -1;
  }
  std::string mt;
  if (!jp.GetString("metric_type", mt)) {
    if (strcasecmp("L2", mt.c_str()) && strcasecmp("InnerProduct", mt.c_str())) return -1;
    pa.metric_type = !strcasecmp("L2", mt.c_str()) ? DistanceComputeType::L2 : DistanceComputeType::INNER_PRODUCT;
  }
  if (!jp.GetInt("device_filters", v)) pa.device_filters = v != 0;
  if (!jp.GetInt("exact_ties", v)) pa.exact_ties = v != 0;
  if (!jp.GetInt("bucket_init_size", v) && v > 0) pa.bucket_init_size = v;
  if (!jp.GetInt("bucket_max_size", v) && v > 0) pa.bucket_max_size = v;
  if (!vector_) {
    HLOG("vector_ must be set before Init");
    return -1;
  }
  d_ = vector_->MetaInfo()->Dimension();
  nlist_ = pa.ncentroids;
  M_ = 1;   // one dummy code byte per list entry (include/gamma_hip.h, IVFFLAT)
  metric_type_ = pa.metric_type;
  nprobe_ = pa.nprobe;
  const char *dev = getenv("GAMMA_HIP_DEVICE");
  int rc = gamma_hip_create(dev ? atoi(dev) : 0, &h_);
  if (rc) return -1;
  members_.assign(1, h_);   // one GPU (the sharded group serves HIPIVFPQ)
  rc = gamma_hip_ivfflat_init(h_, d_, nlist_, metric_type_ == DistanceComputeType::L2 ? GAMMA_HIP_METRIC_L2 : GAMMA_HIP_METRIC_IP,
                              pa.bucket_init_size, pa.bucket_max_size);
  if (!rc) rc = gamma_hip_raw_init(h_, d_);
  if (!rc) rc = gamma_hip_set_exact_ties(h_, pa.exact_ties ? 1 : 0);
  if (rc) {
    HLOG("device init failed: %s (%s)", gamma_hip_strerror(rc), gamma_hip_last_error(h_));
    return -1;
  }
  return 0;
}

RetrievalParameters *GammaIVFFlatHIPIndex::Parse(const std::string &parameters) {   // gamma_index_ivfflat.cc:188-242
  if (parameters == "") return new HIPIVFFlatRetrievalParameters(metric_type_);
  utils::JsonParser jp;
  if (jp.Parse(parameters.c_str())) {
    HLOG("parse retrieval parameters error: %s", parameters.c_str());
    return nullptr;
  }
  HIPIVFFlatRetrievalParameters *rp = new HIPIVFFlatRetrievalParameters();
  std::string mt;
  if (!jp.GetString("metric_type", mt)) {
    if (!strcasecmp("L2", mt.c_str())) rp->SetDistanceComputeType(DistanceComputeType::L2);
    else if (!strcasecmp("InnerProduct", mt.c_str())) rp->SetDistanceComputeType(DistanceComputeType::INNER_PRODUCT);
    else rp->SetDistanceComputeType(metric_type_);
  } else {
    rp->SetDistanceComputeType(metric_type_);
  }
  int v;
  if (!jp.GetInt("nprobe", v) && v > 0) rp->SetNprobe(v);
  if (!jp.GetInt("parallel_on_queries", v)) rp->SetParallelOnQueries(v != 0);
  if (!jp.GetInt("exact_ties", v)) rp->SetExactTies(v != 0 ? 1 : -1);
  return rp;
}

int GammaIVFFlatHIPIndex::SetTrainedCoarse(const float *coarse) {
  coarse_centroids_.assign(coarse, coarse + (size_t)nlist_ * d_);
  if (gamma_hip_ivfflat_set_trained(h_, coarse_centroids_.data())) return -1;
  is_trained_ = true;
  return 0;
}

int GammaIVFFlatHIPIndex::Indexing() {   // IndexIVFFlat::train == the coarse k-means (gamma_index_ivfflat.cc:244-303)
  if (is_trained_) return 0;
  std::vector<float> xt;
  size_t num = 0;
  if (TrainingSet(xt, num)) return -1;
  if (TrainCoarse(num, xt.data())) return -1;
  return gamma_hip_ivfflat_set_trained(h_, coarse_centroids_.data()) ? -1 : (is_trained_ = true, 0);
}

int GammaIVFFlatHIPIndex::Search(RetrievalContext *retrieval_context, int n, const uint8_t *x, int k, float *distances,
                                 int64_t *ids) {
  HIPIVFFlatRetrievalParameters *rp = dynamic_cast<HIPIVFFlatRetrievalParameters *>(retrieval_context->RetrievalParams());
  HIPIVFFlatRetrievalParameters defaults(metric_type_);
  if (rp == nullptr) rp = &defaults;
  GammaSearchCondition *cond = dynamic_cast<GammaSearchCondition *>(retrieval_context);
  gamma_hip_search_params p;
  memset(&p, 0, sizeof(p));
  p.metric = rp->GetDistanceComputeType() == DistanceComputeType::INNER_PRODUCT ? GAMMA_HIP_METRIC_IP : GAMMA_HIP_METRIC_L2;
  p.min_score = cond ? cond->min_score : std::numeric_limits<float>::min();
  p.max_score = cond ? cond->max_score : std::numeric_limits<float>::max();
  p.coarse_mode = -1;
  p.exact_ties = rp->ExactTies();
  std::vector<gamma_hip_range_filter> rf;
  std::vector<gamma_hip_field_filter> ff;
  std::vector<gamma_hip_term_filter> tf;
  if (!(model_param_ && model_param_->device_filters &&
        columns_.Prepare(h_, cond, DocCountOf(this, (int64_t)vector_->MetaInfo()->Size()), p, ff, tf)))
    FillRangeFilters(cond, p, rf);
  const float *xq = reinterpret_cast<const float *>(x);
  int rc;
  if ((cond && cond->brute_force_search) || !is_trained_) {
    // (the reference's IVFFLAT has no brute-force branch; an untrained model would crash there.  Same service as HIPIVFPQ.)
    if (EnsureRaw((int64_t)vector_->MetaInfo()->Size())) return -1;
    rc = gamma_hip_flat_search(h_, &p, n, xq, k, distances, ids);
  } else {
    // the reference takes retrieval_params->Nprobe() as it is (-1 when the request did not set it: a negative
    // allocation there, gamma_index_ivfflat.cc:405-410); the model's nprobe is the sane reading
    p.nprobe = (rp->Nprobe() > 0 && rp->Nprobe() <= nlist_) ? rp->Nprobe() : nprobe_;
    rc = gamma_hip_ivfflat_search(h_, &p, n, xq, k, distances, ids);
  }
  if (rc) {
    HLOG("search failed: %s (%s)", gamma_hip_strerror(rc), gamma_hip_last_error(h_));
    return rc;
  }
  WarnTiesNotHonoured(h_, nullptr, &ties_said_);
  WarnBlasCorners(h_, nullptr, &blas_said_);
  return 0;
}

// "ivfflat.index" in the reference's layout (gamma_index_ivfflat.cc:620-690; iwpq_io.h): a list's codes are its vectors
int GammaIVFFlatHIPIndex::Dump(const std::string &dir) {
  if (!is_trained_) return 0;
  const std::string index_dir = dir + "/" + vector_->MetaInfo()->AbsoluteName();
  if (mkdir(index_dir.c_str(), 0755) && errno != EEXIST) return -1;
  IwPQFile f;
  f.d = d_;
  f.ntotal = 0;
  f.metric = metric_type_ == DistanceComputeType::INNER_PRODUCT ? 0 : 1;
  f.nlist = (size_t)nlist_;
  f.nprobe = (size_t)nprobe_;
  f.coarse = coarse_centroids_;
  f.code_size = sizeof(float) * (size_t)d_;
  f.sizes.resize(nlist_);
  f.codes.resize(nlist_);
  f.ids.resize(nlist_);
  std::vector<uint8_t> dummy;
  for (int l = 0; l < nlist_; l++) {
    const int64_t len = gamma_hip_ivfpq_list_size(h_, l);
    if (len < 0) return -1;
    f.sizes[l] = (size_t)len;
    if (len == 0) continue;
    f.ids[l].resize(len);
    dummy.resize(len);
    if (gamma_hip_ivfpq_get_list(h_, l, f.ids[l].data(), dummy.data())) return -1;
    // the vectors of the list, from the engine's store (what the reference's lists hold)
    std::vector<int64_t> vids(len);
    for (int64_t i = 0; i < len; i++) vids[i] = f.ids[l][i] & 0x7fffffffffffffffLL;
    ScopeVectors sv;
    if (vector_->Gets(vids, sv)) return -1;
    f.codes[l].resize((size_t)len * f.code_size);
    for (int64_t i = 0; i < len; i++) memcpy(&f.codes[l][(size_t)i * f.code_size], sv.Get((int)i), f.code_size);
  }
  return WriteIvFl(index_dir + "/ivfflat.index", f, indexed_vec_count_) ? -1 : 0;
}

int GammaIVFFlatHIPIndex::Load(const std::string &dir) {
  const std::string path = dir + "/" + vector_->MetaInfo()->AbsoluteName() + "/ivfflat.index";
  FILE *probe = fopen(path.c_str(), "rb");
  if (!probe) return 0;   // it should train again after load
  fclose(probe);
  IwPQFile f;
  int indexed = 0;
  if (ReadIvFl(path, &f, &indexed)) return -1;
  if (f.d != d_ || (int)f.nlist != nlist_ || indexed < 0 || indexed > (int)vector_->MetaInfo()->Size()) return -1;
  if (SetTrainedCoarse(f.coarse.data())) return -1;
  if (UploadEngineBitmap()) return -1;
  if (SyncVid2DocID((int64_t)vector_->MetaInfo()->Size())) return -1;
  metric_type_ = f.metric == 0 ? DistanceComputeType::INNER_PRODUCT : DistanceComputeType::L2;
  std::vector<uint8_t> dummy;
  for (int l = 0; l < nlist_; l++) {
    const size_t n = f.sizes[l];
    if (n == 0) continue;
    dummy.assign(n, 0);
    if (gamma_hip_ivfpq_add_keys(h_, l, (int)n, f.ids[l].data(), dummy.data())) return -1;
  }
  indexed_vec_count_ = indexed;
  if (EnsureRaw(std::min<int64_t>(indexed_vec_count_, (int64_t)vector_->MetaInfo()->Size()))) return -1;
  return indexed_vec_count_;
}

}  // namespace tig_gamma
