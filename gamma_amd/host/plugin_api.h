// plugin_api.h -- this repository's own restatement of Gamma's plugin interface
// (reference: index/retrieval_model.h:18-310, index/reflector.h:15-80, the parts of
// common/gamma_common_data.h:39-124 and table/range_query_result.h:25-200 a plugin reads).
//
// Same class names, method signatures, argument meaning and return conventions, so the plugin
// sources in this directory compile UNCHANGED inside the Gamma tree against the real headers
// (INTEGRATION.md).  Two deliberate differences, both confined to this standalone header:
//   * RetrievalModel::updated_vids_ is a tbb::concurrent_bounded_queue<int> in the reference
//     (retrieval_model.h:306); TBB headers are not in this image, so the standalone build uses
//     the small mutex-protected queue below with the same push/try_pop/size/empty calls.
//   * GammaSearchCondition / MultiRangeQueryResults are reduced to the members the plugin
//     boundary touches.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <iostream>
#include <limits>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

enum class VectorValueType : std::uint8_t { FLOAT = 0, BINARY = 1, INT8 = 2 };
enum class DistanceComputeType : std::uint8_t { INNER_PRODUCT = 0, L2, Cosine };

// ---- reflector (index/reflector.h) ------------------------------------------------------
class RetrievalModel;

class ModelFactory {
 public:
  virtual ~ModelFactory() {}
  virtual RetrievalModel *NewModel() = 0;
};

class Reflector {
 public:
  ~Reflector() {
    for (auto &f : model_factories_) delete f.second;
  }
  void RegisterFactory(const std::string &model_name, ModelFactory *model_factory) {
    std::lock_guard<std::mutex> lock(mutex_);
    if (model_factories_.count(model_name)) {
      std::cout << "Duplicated model [" << model_name << "]";
      delete model_factory;
    } else {
      model_factories_[model_name] = model_factory;
    }
  }
  RetrievalModel *GetNewModel(const std::string &model_name) {
    auto it = model_factories_.find(model_name);
    return it == model_factories_.end() ? nullptr : it->second->NewModel();
  }

 private:
  std::map<std::string, ModelFactory *> model_factories_;
  std::mutex mutex_;
};

Reflector &reflector();

#define REGISTER_MODEL(model_name, class_name)                                              \
  class ModelFactory_##class_name : public ModelFactory {                                   \
   public:                                                                                  \
    RetrievalModel *NewModel() { return new class_name(); }                                 \
  };                                                                                        \
  class Register_##class_name {                                                             \
   public:                                                                                  \
    Register_##class_name() {                                                               \
      reflector().RegisterFactory(#model_name, new ModelFactory_##class_name());            \
    }                                                                                       \
  };                                                                                        \
  Register_##class_name register_##class_name;

// ---- perf tool (retrieval_model.h:23-50) --------------------------------------------------
class PerfTool {
 public:
  std::stringstream perf_ss;
  void Perf(const std::string &msg) { perf_ss << msg << " "; }
  void Perf(const char *msg) { perf_ss << msg << " "; }
  const std::stringstream &OutputPerf() { return perf_ss; }
};

class RetrievalParameters {
 public:
  RetrievalParameters() : distance_compute_type_(DistanceComputeType::L2) {}
  RetrievalParameters(const DistanceComputeType &type) : distance_compute_type_(type) {}
  virtual ~RetrievalParameters() {}
  DistanceComputeType GetDistanceComputeType() { return distance_compute_type_; }
  void SetDistanceComputeType(DistanceComputeType type) { distance_compute_type_ = type; }

 protected:
  enum DistanceComputeType distance_compute_type_;
};

class RetrievalContext {
 public:
  RetrievalContext() : retrieval_params_(nullptr), perf_tool_(nullptr) {}
  virtual ~RetrievalContext() {
    delete retrieval_params_;
    retrieval_params_ = nullptr;
  }
  RetrievalParameters *RetrievalParams() { return retrieval_params_; }
  virtual bool IsValid(int id) const = 0;
  virtual bool IsSimilarScoreValid(float score) const = 0;
  PerfTool &GetPerfTool() { return *perf_tool_; }
  RetrievalParameters *retrieval_params_;
  PerfTool *perf_tool_;
};

class VectorMetaInfo {
 public:
  VectorMetaInfo(const std::string &name, int dimension, const VectorValueType &type, int version = 0)
      : name_(name), dimension_(dimension), data_type_(type), size_(0), mem_bytes_(0), version_(version) {
    data_size_ = data_type_ == VectorValueType::FLOAT ? sizeof(float) : sizeof(uint8_t);
  }
  std::string &Name() { return name_; }
  int Dimension() { return dimension_; }
  VectorValueType DataType() { return data_type_; }
  size_t Size() { return size_; }
  long MemBytes() { return mem_bytes_; }
  int DataSize() { return data_size_; }
  std::string AbsoluteName() {
    char v[4];
    snprintf(v, sizeof(v), "%03d", version_);
    return name_ + "." + v;
  }
  std::string name_;
  int dimension_;
  VectorValueType data_type_;
  // (the engine publishes the store size with a plain store and the models read it unlocked, vector/raw_vector_common.h;
  //  the stand-in makes that hand-over explicit so that a ThreadSanitizer run of the plugins sees the protocol, not a race)
  std::atomic<size_t> size_;
  long mem_bytes_;
  int data_size_;
  int version_;
};

class ScopeVectors {
 public:
  ~ScopeVectors() {
    for (size_t i = 0; i < deletable_.size(); i++)
      if (deletable_[i] && ptr_[i]) delete[] ptr_[i];
  }
  void Add(const uint8_t *ptr_in, bool deletable = true) {
    ptr_.push_back(ptr_in);
    deletable_.push_back(deletable);
  }
  const std::vector<const uint8_t *> &Get() { return ptr_; }
  const uint8_t *Get(int idx) { return ptr_[idx]; }
  size_t Size() { return ptr_.size(); }
  std::vector<const uint8_t *> ptr_;
  std::vector<bool> deletable_;
};

class VectorReader {
 public:
  VectorReader(VectorMetaInfo *meta_info) : meta_info_(meta_info) {}
  virtual ~VectorReader() {
    delete meta_info_;
    meta_info_ = nullptr;
  }
  virtual int Gets(const std::vector<int64_t> &vids, ScopeVectors &vecs) const = 0;
  VectorMetaInfo *MetaInfo() { return meta_info_; }

 protected:
  VectorMetaInfo *meta_info_;
};

// bitmap::BitmapManager (util/bitmap_manager.h:13-52), in-memory part: bit i <-> byte i>>3, mask 1<<(i&7)
namespace bitmap {
class BitmapManager {
 public:
  BitmapManager() : size_(0) {}
  int Init(uint32_t bit_size) {
    size_ = bit_size;
    bytes_.assign((bit_size >> 3) + 1, 0);
    return 0;
  }
  int Set(uint32_t bit_id) {
    if (bit_id >= size_) return -1;
    bytes_[bit_id >> 3] |= (char)(1 << (bit_id & 7));
    return 0;
  }
  int Unset(uint32_t bit_id) {
    if (bit_id >= size_) return -1;
    bytes_[bit_id >> 3] &= (char)~(1 << (bit_id & 7));
    return 0;
  }
  bool Test(uint32_t bit_id) { return bit_id < size_ && ((bytes_[bit_id >> 3] >> (bit_id & 7)) & 1); }
  uint32_t BitSize() { return size_; }
  char *Bitmap() { return bytes_.data(); }
  uint32_t BytesSize() { return (size_ >> 3) + 1; }

 private:
  std::vector<char> bytes_;
  uint32_t size_;
};
}  // namespace bitmap

// RawVector (vector/raw_vector.h:70-216), the one member a model reads besides VectorReader's: the engine's
// delete bitmap (:171)
// VIDMgr (vector/raw_vector_common.h:36-110): vid -> docid of tables whose documents carry several vectors
class VIDMgr {
 public:
  explicit VIDMgr(bool multi_vids) : multi_vids_(multi_vids) {}
  bool MultiVids() { return multi_vids_; }
  int VID2DocID(int vid) {
    if (!multi_vids_) return vid;
    return (size_t)vid < vid2docid_.size() ? vid2docid_[vid] : -1;
  }
  void Add(int vid, int docid) {   // standalone stand-in for VIDMgr::Add(vid, docid)
    if ((size_t)vid >= vid2docid_.size()) vid2docid_.resize(vid + 1, -1);
    vid2docid_[vid] = docid;
  }

 private:
  bool multi_vids_;
  std::vector<int> vid2docid_;
};

class RawVector : public VectorReader {
 public:
  RawVector(VectorMetaInfo *meta_info, bitmap::BitmapManager *docids_bitmap)
      : VectorReader(meta_info), docids_bitmap_(docids_bitmap), vid_mgr_(nullptr) {}
  bitmap::BitmapManager *Bitmap() { return docids_bitmap_; }
  VIDMgr *VidMgr() const { return vid_mgr_; }   // raw_vector.h:169

 protected:
  bitmap::BitmapManager *docids_bitmap_;
  VIDMgr *vid_mgr_;
};

// stand-in for tbb::concurrent_bounded_queue<int> (see header comment)
class UpdatedVidQueue {
 public:
  void push(int v) {
    std::lock_guard<std::mutex> l(m_);
    q_.push_back(v);
  }
  bool try_pop(int &v) {
    std::lock_guard<std::mutex> l(m_);
    if (q_.empty()) return false;
    v = q_.front();
    q_.pop_front();
    return true;
  }
  size_t size() {
    std::lock_guard<std::mutex> l(m_);
    return q_.size();
  }
  bool empty() { return size() == 0; }

 private:
  std::mutex m_;
  std::deque<int> q_;
};

class RetrievalModel {
 public:
  RetrievalModel() : vector_(nullptr), indexed_count_(0), indexing_size_(0) {}
  virtual ~RetrievalModel() {}
  virtual int Init(const std::string &model_parameters, int indexing_size) = 0;
  virtual RetrievalParameters *Parse(const std::string &parameters) = 0;
  virtual int Indexing() = 0;
  virtual bool Add(int n, const uint8_t *vec) = 0;
  virtual int Update(const std::vector<int64_t> &ids, const std::vector<const uint8_t *> &vecs) = 0;
  virtual int Delete(const std::vector<int64_t> &ids) = 0;
  virtual int Search(RetrievalContext *retrieval_context, int n, const uint8_t *x, int k,
                     float *distances, int64_t *ids) = 0;
  virtual long GetTotalMemBytes() = 0;
  virtual int Dump(const std::string &dir) = 0;
  virtual int Load(const std::string &dir) = 0;
  virtual void train(int64_t n, const float *x) {}

  VectorReader *vector_;
  UpdatedVidQueue updated_vids_;
  int indexed_count_;  // only used by the framework
  int indexing_size_;
};

// ---- the slice of table/range_query_result.h the scan consults ----------------------------
namespace tig_gamma {

class RangeQueryResult {
 public:
  RangeQueryResult() : min_(std::numeric_limits<int>::max()), max_(0), min_aligned_(0), max_aligned_(0),
                       bitmap_(nullptr), b_not_in_(false) {}
  RangeQueryResult(RangeQueryResult &&o) { bitmap_ = nullptr; *this = std::move(o); }
  RangeQueryResult &operator=(RangeQueryResult &&o) {
    min_ = o.min_; max_ = o.max_; min_aligned_ = o.min_aligned_; max_aligned_ = o.max_aligned_;
    free(bitmap_);
    bitmap_ = o.bitmap_; o.bitmap_ = nullptr; b_not_in_ = o.b_not_in_;
    return *this;
  }
  ~RangeQueryResult() { free(bitmap_); }
  bool Has(int doc) const {
    if (b_not_in_) {
      if (doc < min_ || doc > max_) return true;
      doc -= min_aligned_;
      return !((bitmap_[doc >> 3] >> (doc & 7)) & 1);
    }
    if (doc < min_ || doc > max_) return false;
    doc -= min_aligned_;
    return (bitmap_[doc >> 3] >> (doc & 7)) & 1;
  }
  void SetRange(int x, int y) {
    min_ = std::min(min_, x);
    max_ = std::max(max_, y);
    min_aligned_ = (min_ / 8) * 8;
    max_aligned_ = (max_ / 8 + 1) * 8 - 1;
  }
  void Resize() {   // bitmap::create: (n >> 3) + 1 zeroed bytes (util/bitmap.cc:15-23)
    int n = max_aligned_ - min_aligned_ + 1;
    free(bitmap_);
    bitmap_ = (char *)calloc((n >> 3) + 1, 1);
  }
  void Set(int pos) { bitmap_[pos >> 3] |= (char)(1 << (pos & 7)); }
  int Min() const { return min_; }
  int Max() const { return max_; }
  int MinAligned() { return min_aligned_; }
  int MaxAligned() { return max_aligned_; }
  char *&Ref() { return bitmap_; }
  void SetNotIn(bool b) { b_not_in_ = b; }
  bool NotIn() { return b_not_in_; }

 private:
  int min_, max_, min_aligned_, max_aligned_;
  char *bitmap_;
  bool b_not_in_;
};

class MultiRangeQueryResults {
 public:
  bool Has(int doc) const {
    if (all_results_.size() == 0) return false;
    for (auto &r : all_results_)
      if (!r.Has(doc)) return false;
    return true;
  }
  size_t Size() { return all_results_.size(); }
  void Add(RangeQueryResult &&result) { all_results_.emplace_back(std::move(result)); }
  const RangeQueryResult *GetAllResult() const { return &all_results_[0]; }

 private:
  std::vector<RangeQueryResult> all_results_;
};

}  // namespace tig_gamma

// ---- scalar side of a request: the filters as the client sent them + the table they refer to -----------------
// (c_api/api_data/gamma_table.h:20-27, common/common_query_data.h:9-21, table/table.h:100-120).  The engine fills
// these next to range_query_result for every request (search/gamma_engine.cc:355-357); the reference's GPU model
// and the HIP plugins' device filters (filter_bridge.h) read them.
enum class DataType : std::uint16_t { INT = 0, LONG, FLOAT, DOUBLE, STRING, VECTOR };

namespace tig_gamma {

struct TermFilter {
  std::string field;
  std::string value;   // items separated by \001
  int is_union;        // FilterOperator: 0 And, 1 Or, 2 Not (table/field_range_index.h:23)
};

struct RangeFilter {
  std::string field;
  std::string lower_value;   // raw bytes of the field's type
  std::string upper_value;
  bool include_lower;
  bool include_upper;
};

// Table, the three accessors a model may use (table/table.h:105-120); in-memory stand-in for the standalone build
class Table {
 public:
  int AddField(const std::string &name, DataType type) {
    names_.push_back(name);
    types_.push_back(type);
    values_.emplace_back();
    return (int)names_.size() - 1;
  }
  void AppendValue(int field_id, const std::string &raw) { values_[field_id].push_back(raw); }
  int GetFieldType(const std::string &field, DataType &type) {
    const int i = GetAttrIdx(field);
    if (i < 0) return -1;
    type = types_[i];
    return 0;
  }
  int GetAttrIdx(const std::string &field) const {
    for (size_t i = 0; i < names_.size(); i++)
      if (names_[i] == field) return (int)i;
    return -1;
  }
  int GetFieldRawValue(int docid, int field_id, std::string &value, const uint8_t *doc_v = nullptr) {
    (void)doc_v;
    if (field_id < 0 || (size_t)field_id >= values_.size() || docid < 0 || (size_t)docid >= values_[field_id].size()) {
      oob_reads_++;   // the engine's Table does not check its docid (table/table.cc): a caller must never get here
      return -1;
    }
    value = values_[field_id][docid];
    return 0;
  }
  // stand-alone build only: Table::Update's effect on one field, and the out-of-range reads the stub has seen
  void SetValue(int field_id, int docid, const std::string &raw) { values_[field_id][docid] = raw; }
  long oob_reads_ = 0;

 private:
  std::vector<std::string> names_;
  std::vector<DataType> types_;
  std::vector<std::vector<std::string>> values_;
};

// GammaSearchCondition (common/gamma_common_data.h:39-124), boundary members only
class GammaSearchCondition : public RetrievalContext {
 public:
  GammaSearchCondition(PerfTool *perf_tool) {
    range_query_result = nullptr;
    topn = 0;
    brute_force_search = false;
    has_rank = 1;
    min_score = std::numeric_limits<float>::min();
    max_score = std::numeric_limits<float>::max();
    perf_tool_ = perf_tool;
    docids_bitmap = nullptr;
    docids_bitmap_bits = 0;
    table = nullptr;
  }
  bool IsSimilarScoreValid(float score) const override { return (score <= max_score) && (score >= min_score); }
  bool IsValid(int id) const override {
    if ((range_query_result != nullptr && !range_query_result->Has(id)) ||
        (docids_bitmap && id >= 0 && id < docids_bitmap_bits && ((docids_bitmap[id >> 3] >> (id & 7)) & 1)))
      return false;
    return true;
  }
  MultiRangeQueryResults *range_query_result;
  std::vector<struct RangeFilter> range_filters;
  std::vector<struct TermFilter> term_filters;
  Table *table;
  int topn;
  bool brute_force_search;
  bool has_rank;
  float min_score, max_score;
  const uint8_t *docids_bitmap;   // BitmapManager view (util/bitmap_manager.h)
  int64_t docids_bitmap_bits;
};

}  // namespace tig_gamma
