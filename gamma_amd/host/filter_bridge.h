// filter_bridge.h -- GammaSearchCondition's range filter -> gamma_hip_range_filter[]
#pragma once
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gamma_hip.h"
#include "plugin_includes.h"

namespace tig_gamma {
// translate the engine's per-request filter (table/range_query_result.h) into the C ABI's POD
// descriptors; the bitmaps stay owned by the MultiRangeQueryResults
inline void FillRangeFilters(GammaSearchCondition *cond, gamma_hip_search_params &p,
                             std::vector<gamma_hip_range_filter> &rf) {
  if (!cond || !cond->range_query_result) return;
  p.has_range = 1;
  MultiRangeQueryResults *mr = cond->range_query_result;
  const size_t n = mr->Size();
  const RangeQueryResult *all = n ? mr->GetAllResult() : nullptr;
  for (size_t i = 0; i < n; i++) {
    RangeQueryResult &r = const_cast<RangeQueryResult &>(all[i]);   // accessors are non-const upstream
    gamma_hip_range_filter f;
    f.bitmap = reinterpret_cast<const uint8_t *>(r.Ref());
    f.bitmap_bytes = ((r.MaxAligned() - r.MinAligned() + 1) >> 3) + 1;
    f.min_doc = r.Min();
    f.max_doc = r.Max();
    f.min_aligned = r.MinAligned();
    f.b_not_in = r.NotIn() ? 1 : 0;
    rf.push_back(f);
  }
  p.n_range = (int)rf.size();
  p.range = rf.data();
}

// ---- scalar filters evaluated ON the device (retrieval_param "device_filters": 1) ------------------------------
// The engine hands every request its filters twice: flattened into docid bitmaps (range_query_result, what the
// CPU models test per scanned code) and as the client sent them -- range_filters, term_filters, table
// (search/gamma_engine.cc:355-357; the reference's own GPU model evaluates those per candidate against the
// Table, index/impl/gpu/gamma_index_ivfpq_gpu.cc:646-762).  DeviceColumns keeps a mirror of the filtered
// fields in HBM -- numeric columns as they are, STRING columns as dictionary-encoded item lists -- and turns a
// request's filters into the C ABI's clauses; the scan then evaluates them per code with the reference GPU
// model's rules, and nothing docs/8 bytes long is uploaded per request.
// The mirror is fed lazily: a field is read from the Table (GetFieldRawValue) up to the current doc count the
// first time a filter names it, and extended by the docs added since on every later request.  A doc update reaches
// the mirror through the model's Update (Refresh below: the engine hands an upserted doc's vid to every index of its
// vector fields, vector/vector_manager.cc:355-380 -- the same pass that brings the new vector); an update that
// rewrites ONLY scalar fields of a doc is not announced to retrieval models at all and is seen by the reference's
// GPU model (which reads the Table per candidate) but not by this mirror.
// Requests with more clauses than the C ABI carries (GAMMA_HIP_MAX_FIELD_FILTERS / _TERM_FILTERS / _TERM_ITEMS) take
// the request's docid bitmaps instead -- the same result, evaluated on the host by the engine -- and say so once.
// number of DOCUMENTS behind a vector store of `nvec` vectors: with several vectors per document (VIDMgr,
// vector/raw_vector_common.h:36-110) the table holds fewer docs than the store holds vectors, and Table::GetFieldRawValue
// does not check its docid (table/table.cc)
inline int64_t DocCountOf(RetrievalModel *model, int64_t nvec) {
  RawVector *rv = model ? dynamic_cast<RawVector *>(model->vector_) : nullptr;
  if (nvec > 0 && rv && rv->VidMgr() && rv->VidMgr()->MultiVids()) return (int64_t)rv->VidMgr()->VID2DocID((int)nvec - 1) + 1;
  return nvec;
}

class DeviceColumns {
 public:
  // false: this request cannot take the device path (no table, too many clauses, unknown field, ...): use the bitmaps
  bool Prepare(gamma_hip_index *h, GammaSearchCondition *cond, int64_t ndocs, gamma_hip_search_params &p,
               std::vector<gamma_hip_field_filter> &ff, std::vector<gamma_hip_term_filter> &tf) {
    if (!cond || !cond->table) return false;
    if (cond->range_filters.empty() && cond->term_filters.empty()) return false;
    if (cond->range_filters.size() > GAMMA_HIP_MAX_FIELD_FILTERS || cond->term_filters.size() > GAMMA_HIP_MAX_TERM_FILTERS)
      return TooMany();
    std::lock_guard<std::mutex> g(mu_);
    Table *t = cond->table;
    table_ = t;
    for (auto &r : cond->range_filters) {
      DataType type;
      if (t->GetFieldType(r.field, type) || type == DataType::STRING || type == DataType::VECTOR) return false;
      const int fid = t->GetAttrIdx(r.field);
      if (fid < 0 || Sync(h, t, fid, type, ndocs)) return false;
      gamma_hip_field_filter f;
      memset(&f, 0, sizeof(f));
      f.field_id = fid;
      f.include_lower = r.include_lower ? 1 : 0;
      f.include_upper = r.include_upper ? 1 : 0;
      if (type == DataType::INT) {
        int lo = 0, hi = 0;
        memcpy(&lo, r.lower_value.data(), std::min(sizeof(lo), r.lower_value.size()));
        memcpy(&hi, r.upper_value.data(), std::min(sizeof(hi), r.upper_value.size()));
        f.lower_i = lo;
        f.upper_i = hi;
      } else if (type == DataType::LONG) {
        memcpy(&f.lower_i, r.lower_value.data(), std::min(sizeof(f.lower_i), r.lower_value.size()));
        memcpy(&f.upper_i, r.upper_value.data(), std::min(sizeof(f.upper_i), r.upper_value.size()));
      } else if (type == DataType::FLOAT) {
        float lo = 0, hi = 0;
        memcpy(&lo, r.lower_value.data(), std::min(sizeof(lo), r.lower_value.size()));
        memcpy(&hi, r.upper_value.data(), std::min(sizeof(hi), r.upper_value.size()));
        f.lower_f = lo;
        f.upper_f = hi;
      } else {
        memcpy(&f.lower_f, r.lower_value.data(), std::min(sizeof(f.lower_f), r.lower_value.size()));
        memcpy(&f.upper_f, r.upper_value.data(), std::min(sizeof(f.upper_f), r.upper_value.size()));
      }
      ff.push_back(f);
    }
    for (auto &tm : cond->term_filters) {
      DataType type;
      if (t->GetFieldType(tm.field, type) || type != DataType::STRING) return false;
      const int fid = t->GetAttrIdx(tm.field);
      if (fid < 0 || Sync(h, t, fid, type, ndocs)) return false;
      std::vector<std::string> items = Split(tm.value);
      if (items.size() > GAMMA_HIP_MAX_TERM_ITEMS) return TooMany();
      gamma_hip_term_filter f;
      memset(&f, 0, sizeof(f));
      f.field_id = fid;
      f.op = tm.is_union;
      f.n_items = (int)items.size();
      auto &dict = fields_[fid].dict;
      for (size_t k = 0; k < items.size(); k++) {
        auto it = dict.find(items[k]);
        f.items[k] = it == dict.end() ? -1 : it->second;   // an item no doc carries
      }
      tf.push_back(f);
    }
    p.has_range = 0;   // the clauses replace the request's docid bitmaps
    p.n_range = 0;
    p.range = nullptr;
    p.n_field = (int)ff.size();
    p.field = ff.empty() ? nullptr : ff.data();
    p.n_term = (int)tf.size();
    p.term = tf.empty() ? nullptr : tf.data();
    return true;
  }

  // The docs whose vectors the engine has just handed to the model's Update: their mirrored fields are read again from
  // the Table and rewritten on the device (gamma_hip_field_update / gamma_hip_term_update).
  int Refresh(gamma_hip_index *h, const std::vector<int64_t> &docids) {
    std::lock_guard<std::mutex> g(mu_);
    if (!table_ || fields_.empty()) return 0;
    std::string raw;
    for (auto &kv : fields_) {
      const int fid = kv.first;
      Field &f = kv.second;
      for (int64_t doc : docids) {
        if (doc < 0 || doc >= f.synced) continue;   // not mirrored yet: Sync reads the current value when it gets there
        raw.clear();
        if (table_->GetFieldRawValue((int)doc, fid, raw)) continue;
        if (f.type == DataType::STRING) {
          std::vector<int32_t> items;
          for (auto &s : Split(raw)) {
            auto it = f.dict.find(s);
            if (it == f.dict.end()) it = f.dict.emplace(s, (int)f.dict.size()).first;
            items.push_back(it->second);
          }
          if (gamma_hip_term_update(h, fid, doc, (int32_t)items.size(), items.data())) return -1;
        } else {
          uint8_t v[8] = {0};
          memcpy(v, raw.data(), std::min(sizeof(v), raw.size()));
          if (gamma_hip_field_update(h, fid, doc, v)) return -1;
        }
      }
    }
    return 0;
  }

 private:
  struct Field {
    int64_t synced = 0;
    DataType type = DataType::INT;
    std::map<std::string, int> dict;   // STRING fields: item -> id
  };
  bool TooMany() {
    if (!warned_) {
      warned_ = true;
      fprintf(stderr, "[gamma_hip] a request carries more filter clauses than the device evaluates (%d range, %d term, %d "
              "items per term): such requests take the engine's docid bitmaps\n", GAMMA_HIP_MAX_FIELD_FILTERS,
              GAMMA_HIP_MAX_TERM_FILTERS, GAMMA_HIP_MAX_TERM_ITEMS);
    }
    return false;
  }
  static std::vector<std::string> Split(const std::string &v) {   // utils::split(v, "\001") of the reference
    std::vector<std::string> out;
    size_t a = 0;
    while (a <= v.size()) {
      size_t b = v.find('\001', a);
      if (b == std::string::npos) b = v.size();
      if (b > a) out.push_back(v.substr(a, b - a));
      a = b + 1;
    }
    return out;
  }
  int Sync(gamma_hip_index *h, Table *t, int fid, DataType type, int64_t ndocs) {
    Field &f = fields_[fid];
    f.type = type;
    if (f.synced >= ndocs) return 0;
    int64_t n = ndocs - f.synced;
    std::string raw;
    // (ndocs is the caller's count of DOCUMENTS; a docid the table does not know -- GetFieldRawValue fails -- ends the
    //  pass: the watermark stops there and the doc is read when it exists)
    if (type == DataType::STRING) {
      std::vector<int32_t> counts(n), items;
      for (int64_t i = 0; i < n; i++) {
        raw.clear();
        if (t->GetFieldRawValue((int)(f.synced + i), fid, raw)) {
          n = i;
          counts.resize(n);
          break;
        }
        std::vector<std::string> its = Split(raw);
        counts[i] = (int32_t)its.size();
        for (auto &s : its) {
          auto it = f.dict.find(s);
          if (it == f.dict.end()) it = f.dict.emplace(s, (int)f.dict.size()).first;
          items.push_back(it->second);
        }
      }
      if (n > 0 && gamma_hip_term_append(h, fid, n, counts.data(), items.data())) return -1;
    } else {
      const int dt = type == DataType::INT ? GAMMA_HIP_FIELD_INT : type == DataType::LONG ? GAMMA_HIP_FIELD_LONG
                   : type == DataType::FLOAT ? GAMMA_HIP_FIELD_FLOAT : GAMMA_HIP_FIELD_DOUBLE;
      const size_t es = (dt == GAMMA_HIP_FIELD_INT || dt == GAMMA_HIP_FIELD_FLOAT) ? 4 : 8;
      std::vector<uint8_t> buf((size_t)n * es, 0);
      for (int64_t i = 0; i < n; i++) {
        raw.clear();
        if (t->GetFieldRawValue((int)(f.synced + i), fid, raw)) {
          n = i;
          break;
        }
        memcpy(&buf[(size_t)i * es], raw.data(), std::min(es, raw.size()));
      }
      if (n > 0 && gamma_hip_field_append(h, fid, dt, n, buf.data())) return -1;
    }
    f.synced += n;
    return 0;
  }
  std::mutex mu_;
  std::map<int, Field> fields_;
  Table *table_ = nullptr;
  bool warned_ = false;
};

}  // namespace tig_gamma
