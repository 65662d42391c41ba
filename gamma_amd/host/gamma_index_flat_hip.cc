// GammaFLATHIPIndex -- see gamma_index_flat_hip.h
#include "gamma_index_flat_hip.h"
#include "gamma_index_ivfpq_hip.h"   // WarnTiesNotHonoured

#include "filter_bridge.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <algorithm>
#include <limits>

namespace tig_gamma {

REGISTER_MODEL(HIPFLAT, GammaFLATHIPIndex);

GammaFLATHIPIndex::~GammaFLATHIPIndex() {
  if (h_) gamma_hip_destroy(h_);
}

int GammaFLATHIPIndex::Init(const std::string &model_parameters, int indexing_size) {
  indexing_size_ = indexing_size;
  if (!vector_) return -1;
  if (model_parameters != "") {   // FLATModelParams::Parse, gamma_index_flat.cc:34-55
    utils::JsonParser jp;
    if (jp.Parse(model_parameters.c_str())) return -1;
    std::string mt;
    if (!jp.GetString("metric_type", mt)) {
      if (strcasecmp("L2", mt.c_str()) && strcasecmp("InnerProduct", mt.c_str())) return -1;
      metric_type_ = !strcasecmp("L2", mt.c_str()) ? DistanceComputeType::L2 : DistanceComputeType::INNER_PRODUCT;
    }
    int v = 0;
    if (!jp.GetInt("device_filters", v)) device_filters_ = v != 0;   // HIP only, see filter_bridge.h
    if (!jp.GetInt("exact_ties", v)) exact_ties_ = v != 0;           // HIP only: the reference's heap order inside ties (default on)
  }
  d_ = vector_->MetaInfo()->Dimension();
  const char *dev = getenv("GAMMA_HIP_DEVICE");
  if (gamma_hip_create(dev ? atoi(dev) : 0, &h_)) return -1;
  if (gamma_hip_set_exact_ties(h_, exact_ties_ ? 1 : 0)) return -1;
  return gamma_hip_raw_init(h_, d_) ? -1 : 0;
}

RetrievalParameters *GammaFLATHIPIndex::Parse(const std::string &parameters) {
  if (parameters == "") return new HIPFlatRetrievalParameters(metric_type_);
  utils::JsonParser jp;
  if (jp.Parse(parameters.c_str())) return nullptr;
  DistanceComputeType type = metric_type_;
  std::string mt;
  if (!jp.GetString("metric_type", mt))
    type = !strcasecmp("L2", mt.c_str()) ? DistanceComputeType::L2 : DistanceComputeType::INNER_PRODUCT;
  int poq = 1;
  jp.GetInt("parallel_on_queries", poq);
  HIPFlatRetrievalParameters *rp = new HIPFlatRetrievalParameters(poq != 0, type);
  int et = 0;
  if (!jp.GetInt("exact_ties", et)) rp->SetExactTies(et != 0 ? 1 : -1);
  return rp;
}

bool GammaFLATHIPIndex::Add(int n, const uint8_t *vec) {
  // the CPU model reads vector_ at search time; the device keeps a mirror, fed in vid order at explicit rows
  std::lock_guard<std::mutex> g(raw_mu_);
  if (gamma_hip_raw_write(h_, uploaded_, n, reinterpret_cast<const float *>(vec))) return false;
  uploaded_ += n;
  return SyncVid2DocID(uploaded_) == 0;
}

// multi-vector documents: docids of the mirrored vids to the device (filters and the delete bitmap are on DOC ids)
int GammaFLATHIPIndex::SyncVid2DocID(int64_t upto) {
  RawVector *rv = dynamic_cast<RawVector *>(vector_);
  if (!rv || !rv->VidMgr() || !rv->VidMgr()->MultiVids()) return 0;
  const int64_t have = gamma_hip_vid2docid_count(h_);
  if (have < 0) return -1;
  if (upto <= have) return 0;
  std::vector<int32_t> m((size_t)(upto - have));
  for (int64_t v = have; v < upto; v++) m[(size_t)(v - have)] = rv->VidMgr()->VID2DocID((int)v);
  return gamma_hip_vid2docid_append(h_, (int64_t)m.size(), m.data());
}

int GammaFLATHIPIndex::Update(const std::vector<int64_t> &ids, const std::vector<const uint8_t *> &vecs) {
  std::lock_guard<std::mutex> g(raw_mu_);
  for (size_t i = 0; i < ids.size(); i++)
    if (ids[i] < uploaded_ && gamma_hip_raw_update(h_, ids[i], reinterpret_cast<const float *>(vecs[i]))) return -1;
  return 0;
}

int GammaFLATHIPIndex::Delete(const std::vector<int64_t> &ids) {
  if (ids.empty()) return 0;
  std::vector<int64_t> docs(ids);   // the bitmap is on DOC ids (VIDMgr::VID2DocID; identity for single-vector documents)
  RawVector *rv = dynamic_cast<RawVector *>(vector_);
  if (rv && rv->VidMgr() && rv->VidMgr()->MultiVids())
    for (size_t i = 0; i < docs.size(); i++) docs[i] = rv->VidMgr()->VID2DocID((int)ids[i]);
  return gamma_hip_bitmap_set(h_, docs.data(), (int64_t)docs.size(), 1) ? -1 : 0;
}

int GammaFLATHIPIndex::Load(const std::string &dir) {
  // deletes made before the restart: the engine has loaded its bitmap file already (vector_ is a RawVector)
  if (RawVector *rv = dynamic_cast<RawVector *>(vector_)) {
    if (rv->Bitmap() && rv->Bitmap()->BitSize() > 0 &&
        gamma_hip_bitmap_upload(h_, reinterpret_cast<const uint8_t *>(rv->Bitmap()->Bitmap()),
                                (int64_t)rv->Bitmap()->BitSize()))
      return -1;
  }
  std::lock_guard<std::mutex> g(raw_mu_);
  const int64_t nvec = (int64_t)vector_->MetaInfo()->Size();
  for (int64_t i0 = uploaded_; i0 < nvec; i0 += 65536) {
    const int64_t nb = std::min<int64_t>(65536, nvec - i0);
    std::vector<int64_t> vids(nb);
    for (int64_t i = 0; i < nb; i++) vids[i] = i0 + i;
    ScopeVectors sv;
    if (vector_->Gets(vids, sv)) return -1;
    std::vector<float> buf((size_t)nb * d_);
    for (int64_t i = 0; i < nb; i++) memcpy(&buf[(size_t)i * d_], sv.Get((int)i), sizeof(float) * d_);
    if (gamma_hip_raw_write(h_, i0, nb, buf.data())) return -1;
    uploaded_ = i0 + nb;
  }
  if (SyncVid2DocID(uploaded_)) return -1;
  return (int)uploaded_;
}

int GammaFLATHIPIndex::Search(RetrievalContext *retrieval_context, int n, const uint8_t *x, int k,
                              float *distances, int64_t *ids) {
  HIPFlatRetrievalParameters *rp = dynamic_cast<HIPFlatRetrievalParameters *>(retrieval_context->RetrievalParams());
  HIPFlatRetrievalParameters defaults(true, DistanceComputeType::L2);   // gamma_index_flat.cc:125-128
  if (rp == nullptr) rp = &defaults;
  if (x == nullptr) return -1;
  GammaSearchCondition *cond = dynamic_cast<GammaSearchCondition *>(retrieval_context);
  gamma_hip_search_params p;
  memset(&p, 0, sizeof(p));
  p.metric = rp->GetDistanceComputeType() == DistanceComputeType::INNER_PRODUCT ? GAMMA_HIP_METRIC_IP
                                                                                : GAMMA_HIP_METRIC_L2;
  p.min_score = cond ? cond->min_score : std::numeric_limits<float>::min();
  p.max_score = cond ? cond->max_score : std::numeric_limits<float>::max();
  p.exact_ties = rp->ExactTies();
  std::vector<gamma_hip_range_filter> rf;
  std::vector<gamma_hip_field_filter> ff;
  std::vector<gamma_hip_term_filter> tf;
  if (!(device_filters_ && columns_.Prepare(h_, cond, DocCountOf(this, (int64_t)vector_->MetaInfo()->Size()), p, ff, tf)))
    FillRangeFilters(cond, p, rf);
  const int rc = gamma_hip_flat_search(h_, &p, n, reinterpret_cast<const float *>(x), k, distances, ids);
  if (!rc) WarnTiesNotHonoured(h_, nullptr, &ties_said_);
  return rc;
}

}  // namespace tig_gamma
