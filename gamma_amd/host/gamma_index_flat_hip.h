// GammaFLATHIPIndex -- RetrievalModel plugin "HIPFLAT": Gamma's brute-force model
// (reference index/impl/gamma_index_flat.{h,cc}) on an MI355X.  Same JSON keys
// (metric_type, parallel_on_queries), same Search contract.
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gamma_hip.h"
#include "plugin_includes.h"
#include "filter_bridge.h"

namespace tig_gamma {

class HIPFlatRetrievalParameters : public RetrievalParameters {
 public:
  HIPFlatRetrievalParameters() : RetrievalParameters(), parallel_on_queries_(true), exact_ties_(0) {}
  HIPFlatRetrievalParameters(bool parallel_on_queries, enum DistanceComputeType type)
      : RetrievalParameters(type), parallel_on_queries_(parallel_on_queries), exact_ties_(0) {}
  HIPFlatRetrievalParameters(enum DistanceComputeType type) : RetrievalParameters(type), parallel_on_queries_(true), exact_ties_(0) {}
  bool ParallelOnQueries() { return parallel_on_queries_; }
  // HIP only ("exact_ties" in the request's retrieval parameters): 0 = the model's setting, 1 = on, -1 = off
  int ExactTies() { return exact_ties_; }
  void SetExactTies(int v) { exact_ties_ = v; }

 private:
  bool parallel_on_queries_;   // accepted for compatibility; the device path is always batched
  int exact_ties_;
};

class GammaFLATHIPIndex : public RetrievalModel {
 public:
  GammaFLATHIPIndex() {}
  ~GammaFLATHIPIndex() override;
  int Init(const std::string &model_parameters, int indexing_size) override;
  RetrievalParameters *Parse(const std::string &parameters) override;
  int Indexing() override { return 0; }
  bool Add(int n, const uint8_t *vec) override;
  int Update(const std::vector<int64_t> &ids, const std::vector<const uint8_t *> &vecs) override;
  int Delete(const std::vector<int64_t> &ids) override;
  int Search(RetrievalContext *retrieval_context, int n, const uint8_t *x, int k, float *distances,
             int64_t *ids) override;
  long GetTotalMemBytes() override { return h_ ? (long)gamma_hip_total_mem_bytes(h_) : 0; }
  int Dump(const std::string &dir) override { return 0; }
  int Load(const std::string &dir) override;
  DistanceComputeType metric_type_ = DistanceComputeType::INNER_PRODUCT;

 private:
  gamma_hip_index *h_ = nullptr;
  std::atomic<int64_t> ties_said_{0};   // WarnTiesNotHonoured: what this model has reported so far
  int d_ = 0;
  int64_t uploaded_ = 0;
  int SyncVid2DocID(int64_t upto);
  std::mutex raw_mu_;   // uploaded_ + the mirror writes
  bool device_filters_ = false;
  bool exact_ties_ = true;      // "exact_ties" of the model parameters (HIP only, default on)
  DeviceColumns columns_;
};

}  // namespace tig_gamma
