// GammaIVFPQHIPIndex -- RetrievalModel plugin "HIPIVFPQ": Gamma's IVFPQ model with the whole
// search path (coarse quantizer, LUT, list scan, filters, top-k, re-rank) and the Add-path
// encoding on an MI355X through the C ABI of include/gamma_hip.h.
//
// Mirrors GammaIVFPQIndex (reference index/impl/gamma_index_ivfpq.{h,cc}): same JSON keys and
// defaults (the reference's IVFPQModelParams :675-887 and IVFPQRetrievalParameters :629-673), same
// return codes, same Search contract.  The parameter classes carry a HIP prefix: the plugin is compiled
// INTO libgamma next to the reference's own IVFPQ model, where a second
// tig_gamma::IVFPQModelParams with another layout would be an ODR violation.  Unsupported on device and rejected in Init like any bad parameter:
// hnsw quantizer, opq, support_indivisible_nsubvector, nbits_per_idx != 8.
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gamma_hip.h"
#include "plugin_includes.h"
#include "filter_bridge.h"

namespace tig_gamma {

void WarnTiesNotHonoured(gamma_hip_index *h, gamma_hip_group *grp, std::atomic<int64_t> *said);
void WarnBlasCorners(gamma_hip_index *h, gamma_hip_group *grp, std::atomic<int64_t> *said);   // gamma_index_ivfpq_hip.cc

class HIPIVFPQRetrievalParameters : public RetrievalParameters {
 public:
  HIPIVFPQRetrievalParameters() : RetrievalParameters(), parallel_on_queries_(true), recall_num_(100), nprobe_(-1), exact_ties_(0) {}
  HIPIVFPQRetrievalParameters(enum DistanceComputeType type)
      : RetrievalParameters(type), parallel_on_queries_(true), recall_num_(100), nprobe_(-1), exact_ties_(0) {}
  int RecallNum() { return recall_num_; }
  void SetRecallNum(int recall_num) { recall_num_ = recall_num; }
  int Nprobe() { return nprobe_; }
  void SetNprobe(int nprobe) { nprobe_ = nprobe; }
  bool ParallelOnQueries() { return parallel_on_queries_; }
  void SetParallelOnQueries(bool p) { parallel_on_queries_ = p; }
  // HIP only ("exact_ties" in the request's retrieval parameters): 0 = the model's setting, 1 = on, -1 = off
  int ExactTies() { return exact_ties_; }
  void SetExactTies(int v) { exact_ties_ = v; }

 protected:
  bool parallel_on_queries_;   // accepted for compatibility; the device path is always batched
  int recall_num_;
  int nprobe_;
  int exact_ties_;
};

struct HIPIVFPQModelParams {
  int ncentroids = 2048;
  int nsubvector = 64;
  bool support_indivisible_nsubvector = false;
  int nbits_per_idx = 8;
  int nprobe = 80;
  DistanceComputeType metric_type = DistanceComputeType::INNER_PRODUCT;
  bool has_hnsw = false;
  bool has_opq = false;
  int bucket_init_size = 1000;
  int bucket_max_size = 1280000;
  bool device_filters = false;   // HIP only: evaluate range / term filters on device-resident columns (filter_bridge.h)
  bool exact_ties = true;        // HIP only: the reference's heap order inside exact distance ties (gamma_hip_set_exact_ties)
  bool perf_stages = false;      // HIP only: per-stage device times in the request's PerfTool (stage events on every search)
  std::vector<int> devices;      // HIP only: "devices": "0,1,2,3" -- the index sharded by IVF list over these GPUs in this
                                 // process (gamma_hip_group_*); empty: one GPU, GAMMA_HIP_DEVICE or 0
  bool replicate = false;        // HIP only: "placement": "replicate" -- every device holds every list, a search splits
                                 // the queries (results of one GPU bit for bit); "shard" (default): by IVF list
  int Parse(const char *str);   // 0 ok, -1 bad (same rules as gamma_index_ivfpq.h:708-851)
};

class GammaIVFPQHIPIndex : public RetrievalModel {
 public:
  GammaIVFPQHIPIndex();
  ~GammaIVFPQHIPIndex() override;
  int Init(const std::string &model_parameters, int indexing_size) override;
  RetrievalParameters *Parse(const std::string &parameters) override;
  int Indexing() override;
  bool Add(int n, const uint8_t *vec) override;
  int Update(const std::vector<int64_t> &ids, const std::vector<const uint8_t *> &vecs) override;
  int Delete(const std::vector<int64_t> &ids) override;
  int Search(RetrievalContext *retrieval_context, int n, const uint8_t *x, int k, float *distances,
             int64_t *ids) override;
  long GetTotalMemBytes() override;
  int Dump(const std::string &dir) override;
  int Load(const std::string &dir) override;

  // install an externally trained quantizer (tests: same centroids as the oracle)
  int SetTrained(const float *coarse_centroids, const float *pq_centroids);

  // exposed for the harness / tests
  bool is_trained_ = false;
  int d_ = 0, nlist_ = 0, M_ = 0, nprobe_ = 80;
  DistanceComputeType metric_type_ = DistanceComputeType::INNER_PRODUCT;
  int indexed_vec_count_ = 0;
  std::vector<float> coarse_centroids_, pq_centroids_;

 protected:
  int TrainOnHost(size_t num, const float *xt);
  int TrainCoarse(size_t num, const float *xt);
  int TrainingSet(std::vector<float> &xt, size_t &num);
  int EnsureRaw(int64_t upto);
  int UploadEngineBitmap();
  int SyncVid2DocID(int64_t upto);   // multi-vector documents: docids of vids [0, upto) to the device (VIDMgr)
  std::mutex raw_mu_;   // raw_uploaded_ + the mirror writes (Search threads, the indexing thread, Load)
  DeviceColumns columns_;
  // one GPU: h_ alone.  "devices" with several entries: a group of handles, lists sharded by owner; h_ is member 0 and
  // serves what needs no lists (training's assignment step, brute-force search over the replicated raw vectors)
  gamma_hip_group *grp_ = nullptr;
  std::vector<gamma_hip_index *> members_;   // where replicated state goes: {h_} or every member of the group
  template <typename F>
  int ForAll(F f) {
    for (gamma_hip_index *m : members_) {
      const int rc = f(m);
      if (rc) return rc;
    }
    return 0;
  }
  int OpenDevices(const std::vector<int> &devices, bool replicate = false);
  void PerfLabels(GammaSearchCondition *cond);
  std::mutex perf_mu_;
  double perf_ms_[GAMMA_HIP_NUM_STAGES] = {0};
  gamma_hip_index *h_ = nullptr;
  std::atomic<int64_t> blas_said_{0};   // WarnBlasCorners
  std::atomic<int64_t> ties_said_{0};   // WarnTiesNotHonoured: what this model has reported so far
  HIPIVFPQModelParams *model_param_ = nullptr;
  int64_t raw_uploaded_ = 0;
};

// "HIPIVFFLAT": the reference's IVFFLAT model (index/impl/gamma_index_ivfflat.{h,cc}) on the device.  Same JSON
// keys (ncentroids, nprobe, metric_type; retrieval: metric_type, nprobe, parallel_on_queries), same Search
// contract.  The reference keeps the vectors inside the inverted lists; here the lists hold vector ids and the rows
// come from the HBM mirror of the vector store the IVFPQ plugin already keeps for its re-rank, so Add / Update /
// Delete and the raw mirror are inherited; Init, Indexing (coarse k-means only), Search, Dump / Load ("IvFl" file,
// iwpq_io.h) differ.
class HIPIVFFlatRetrievalParameters : public RetrievalParameters {
 public:
  HIPIVFFlatRetrievalParameters() : RetrievalParameters(), parallel_on_queries_(true), nprobe_(-1), exact_ties_(0) {}
  HIPIVFFlatRetrievalParameters(enum DistanceComputeType type)
      : RetrievalParameters(type), parallel_on_queries_(true), nprobe_(-1), exact_ties_(0) {}
  int Nprobe() { return nprobe_; }
  // HIP only ("exact_ties" in the request's retrieval parameters): 0 = the model's setting, 1 = on, -1 = off
  int ExactTies() { return exact_ties_; }
  void SetExactTies(int v) { exact_ties_ = v; }
  void SetNprobe(int nprobe) { nprobe_ = nprobe; }
  bool ParallelOnQueries() { return parallel_on_queries_; }
  void SetParallelOnQueries(bool p) { parallel_on_queries_ = p; }

 protected:
  bool parallel_on_queries_;   // accepted for compatibility; the device path is always batched
  int nprobe_;
  int exact_ties_;
};

class GammaIVFFlatHIPIndex : public GammaIVFPQHIPIndex {
 public:
  int Init(const std::string &model_parameters, int indexing_size) override;
  RetrievalParameters *Parse(const std::string &parameters) override;
  int Indexing() override;
  int Search(RetrievalContext *retrieval_context, int n, const uint8_t *x, int k, float *distances,
             int64_t *ids) override;
  int Dump(const std::string &dir) override;
  int Load(const std::string &dir) override;
  int SetTrainedCoarse(const float *coarse_centroids);
};

}  // namespace tig_gamma
