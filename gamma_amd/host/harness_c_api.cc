// harness_c_api.cc -- extern "C" test harness that drives the RetrievalModel plugins the way
// VectorManager does (vector/vector_manager.cc:161-192,305-349,433-491): create by name through
// the reflector, set vector_, Init, Add, Indexing, Parse + Search with a GammaSearchCondition.
// Used by the Python tests through ctypes; not part of the product surface.
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "gamma_index_ivfpq_hip.h"
#include "iwpq_io.h"
#include "plugin_api.h"

using namespace tig_gamma;

namespace {
// MemoryRawVector stand-in: the engine-owned raw vector store the model reads through VectorReader, with the
// engine's delete bitmap behind RawVector::Bitmap() (vector/raw_vector.h:171; GammaEngine owns the manager,
// search/gamma_engine.cc:255)
class MemVectorReader : public RawVector {
 public:
  MemVectorReader(int d) : RawVector(new VectorMetaInfo("vec", d, VectorValueType::FLOAT), &bitmap_), d_(d) {
    bitmap_.Init(1 << 22);
  }
  // documents with several vectors: docid of every vid appended from now on (tests)
  void SetMultiVids() { vid_mgr_ = &vids_; }
  VIDMgr vids_{true};
  int Gets(const std::vector<int64_t> &vids, ScopeVectors &vecs) const override {
    std::lock_guard<std::mutex> g(mu_);
    for (auto v : vids) {
      if (v < 0 || (size_t)v >= data_.size() / d_) return -1;
      // a copy the ScopeVectors owns: the store may grow (and move) while a model still reads the rows
      uint8_t *c = new uint8_t[sizeof(float) * d_];
      memcpy(c, &data_[(size_t)v * d_], sizeof(float) * d_);
      vecs.Add(c, true);
    }
    return 0;
  }
  void Append(int n, const float *x) {
    std::lock_guard<std::mutex> g(mu_);
    data_.insert(data_.end(), x, x + (size_t)n * d_);
    meta_info_->size_ += n;
  }
  int d_;
  std::vector<float> data_;
  bitmap::BitmapManager bitmap_;
  mutable std::mutex mu_;
};

struct Host {
  MemVectorReader *store;
  RetrievalModel *model;
  Table table;   // the scalar fields of the docs (device filters read it through GammaSearchCondition::table)
  std::string last_perf;   // PerfTool summary of the last search (what the engine logs at online_log_level=debug)
  std::mutex perf_mu;      // gh_host_search runs on any number of client threads
};
}  // namespace

extern "C" {

void *gh_host_new(const char *retrieval_type, int d) {
  RetrievalModel *m = reflector().GetNewModel(retrieval_type);
  if (!m) return nullptr;
  Host *h = new Host();
  h->store = new MemVectorReader(d);
  h->model = m;
  m->vector_ = h->store;
  return h;
}
void gh_host_free(void *hp) {
  Host *h = (Host *)hp;
  delete h->model;
  delete h->store;
  delete h;
}
int gh_host_init(void *hp, const char *retrieval_param, int indexing_size) {
  return ((Host *)hp)->model->Init(retrieval_param, indexing_size);
}
// AddToStore (raw vector append) without indexing: what happens before the index is trained
void gh_host_store(void *hp, int n, const float *x) { ((Host *)hp)->store->Append(n, x); }
// multi-vector documents: docid of vids [first, first + n)
void gh_host_set_vid2docid(void *hp, int first, int n, const int *docids) {
  Host *h = (Host *)hp;
  h->store->SetMultiVids();
  for (int i = 0; i < n; i++) h->store->vids_.Add(first + i, docids[i]);
}
int gh_host_indexing(void *hp) { return ((Host *)hp)->model->Indexing(); }
// model->Add for vectors already in the store (AddRTVecsToIndex)
int gh_host_add(void *hp, int n, const float *x) {
  Host *h = (Host *)hp;
  bool ok = h->model->Add(n, reinterpret_cast<const uint8_t *>(x));
  if (ok) h->model->indexed_count_ += n;
  return ok ? 1 : 0;
}
int gh_host_update(void *hp, int64_t vid, const float *x) {
  Host *h = (Host *)hp;
  memcpy(&h->store->data_[(size_t)vid * h->store->d_], x, sizeof(float) * h->store->d_);
  std::vector<int64_t> ids{vid};
  std::vector<const uint8_t *> vecs{reinterpret_cast<const uint8_t *>(x)};
  return h->model->Update(ids, vecs);
}
// the engine's update pass (vector/vector_manager.cc:355-380): a batch of updated vids handed to the model at once
int gh_host_update_batch(void *hp, int n, const int64_t *vids, const float *x) {
  Host *h = (Host *)hp;
  const int d = h->store->d_;
  std::vector<int64_t> ids(vids, vids + n);
  std::vector<const uint8_t *> vecs(n);
  for (int i = 0; i < n; i++) {
    memcpy(&h->store->data_[(size_t)vids[i] * d], x + (size_t)i * d, sizeof(float) * d);
    vecs[i] = reinterpret_cast<const uint8_t *>(x + (size_t)i * d);
  }
  return h->model->Update(ids, vecs);
}
// GammaEngine::Delete (search/gamma_engine.cc:802-824): the doc bit in the engine's bitmap, then the models
int gh_host_delete(void *hp, const int64_t *vids, int n) {
  Host *h = (Host *)hp;
  // GammaEngine::Delete sets the DOC bit, then hands the doc's vids to the model (search/gamma_engine.cc:802-824)
  for (int i = 0; i < n; i++)
    h->store->bitmap_.Set((uint32_t)(h->store->VidMgr() ? h->store->VidMgr()->VID2DocID((int)vids[i]) : vids[i]));
  std::vector<int64_t> ids(vids, vids + n);
  return h->model->Delete(ids);
}
// the engine's bitmap alone (what BitmapManager::Load restores after a restart, before the models' Load)
void gh_host_engine_bitmap_set(void *hp, const int64_t *vids, int n) {
  Host *h = (Host *)hp;
  for (int i = 0; i < n; i++) h->store->bitmap_.Set((uint32_t)vids[i]);
}
int gh_host_search(void *hp, const char *retrieval_params, int has_rank, int brute_force, float min_score,
                   float max_score, int n, const float *x, int k, float *distances, int64_t *ids);
// Brute-force Search calls from `nthreads` client threads WHILE the indexing thread adds `n` vectors in
// batches of `batch` (the model is untrained or brute_force_search is set: both paths mirror the vector
// store on demand).  Returns the number of failed calls; afterwards the mirror must hold exactly n rows.
int gh_host_search_during_add(void *hp, const char *retrieval_params, int nthreads, int n, int batch,
                              const float *x, int d, const float *q, int nq, int k) {
  Host *h = (Host *)hp;
  std::atomic<int> failed(0), stop(0);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t]() {
      std::vector<float> D((size_t)k);
      std::vector<int64_t> I((size_t)k);
      int i = t;
      while (!stop.load()) {
        if (gh_host_search(hp, retrieval_params, 1, 1, -1e30f, 1e30f, 1, q + (size_t)(i % nq) * d, k, D.data(), I.data()))
          failed++;
        i += nthreads;
      }
    });
  for (int i0 = 0; i0 < n; i0 += batch) {
    const int nb = std::min(batch, n - i0);
    h->store->Append(nb, x + (size_t)i0 * d);
    if (!h->model->Add(nb, reinterpret_cast<const uint8_t *>(x + (size_t)i0 * d))) failed++;
    else h->model->indexed_count_ += nb;
  }
  stop = 1;
  for (auto &t : th) t.join();
  return failed.load();
}
int gh_host_search(void *hp, const char *retrieval_params, int has_rank, int brute_force, float min_score,
                   float max_score, int n, const float *x, int k, float *distances, int64_t *ids) {
  Host *h = (Host *)hp;
  PerfTool perf;
  GammaSearchCondition cond(&perf);
  cond.topn = k;
  cond.has_rank = has_rank != 0;
  cond.brute_force_search = brute_force != 0;
  cond.min_score = min_score;
  cond.max_score = max_score;
  cond.retrieval_params_ = h->model->Parse(retrieval_params);   // owned by the context
  if (!cond.retrieval_params_) return -100;
  const int rc = h->model->Search(&cond, n, reinterpret_cast<const uint8_t *>(x), k, distances, ids);
  {
    std::lock_guard<std::mutex> g(h->perf_mu);
    h->last_perf = perf.OutputPerf().str();
  }
  return rc;
}
// the PerfTool summary of the last gh_host_search (single-threaded use)
int gh_host_last_perf(void *hp, char *out, int cap) {
  Host *h = (Host *)hp;
  const int n = std::min<int>(cap - 1, (int)h->last_perf.size());
  if (cap > 0) {
    memcpy(out, h->last_perf.data(), n);
    out[n] = 0;
  }
  return n;
}
// Table::Update of one field of one doc (table/table.cc:420-470), and the stub's count of out-of-range reads
void gh_host_table_set(void *hp, int field_id, int docid, const uint8_t *raw, int len) {
  ((Host *)hp)->table.SetValue(field_id, docid, std::string(reinterpret_cast<const char *>(raw), (size_t)len));
}
long gh_host_table_oob_reads(void *hp) { return ((Host *)hp)->table.oob_reads_; }
// Search with a scalar filter: range clause i matches docids[off_i .. off_i+counts[i]) (not_in[i]
// inverts it); the MultiRangeQueryResults is built the way field_range_index.cc fills one
// (SetRange over the matching ids, Resize, Set(doc - MinAligned)).
int gh_host_search_filtered(void *hp, const char *retrieval_params, int has_rank, int brute_force,
                            float min_score, float max_score, int n, const float *x, int k, float *distances,
                            int64_t *ids, int n_range, const int64_t *docids, const int *counts,
                            const int *not_in) {
  Host *h = (Host *)hp;
  PerfTool perf;
  GammaSearchCondition cond(&perf);
  MultiRangeQueryResults mr;
  size_t off = 0;
  for (int i = 0; i < n_range; i++) {
    RangeQueryResult r;
    if (counts[i] > 0) {
      for (int j = 0; j < counts[i]; j++) r.SetRange((int)docids[off + j], (int)docids[off + j]);
    } else {
      r.SetRange(0, 0);
    }
    r.Resize();
    for (int j = 0; j < counts[i]; j++) r.Set((int)docids[off + j] - r.MinAligned());
    r.SetNotIn(not_in[i] != 0);
    off += counts[i];
    mr.Add(std::move(r));
  }
  cond.range_query_result = &mr;
  cond.topn = k;
  cond.has_rank = has_rank != 0;
  cond.brute_force_search = brute_force != 0;
  cond.min_score = min_score;
  cond.max_score = max_score;
  cond.retrieval_params_ = h->model->Parse(retrieval_params);
  if (!cond.retrieval_params_) return -100;
  return h->model->Search(&cond, n, reinterpret_cast<const uint8_t *>(x), k, distances, ids);
}
// Closed-loop clients, the pattern of the reference's tools/perf.cc: `nthreads` threads, each issuing
// `calls` Search calls of `nq_call` queries taken round-robin from a pool of `npool` queries.  Returns the
// wall time in seconds (< 0 on error); lat_us[nthreads * calls] receives every call's latency.
double gh_host_concurrent_clients(void *hp, const char *retrieval_params, int has_rank, int nthreads, int calls,
                                  int nq_call, const float *pool, int npool, int d, int k, float *lat_us) {
  std::vector<std::thread> th;
  std::atomic<int> failed(0);
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < nthreads; t++) {
    th.emplace_back([=, &failed]() {
      std::vector<float> D((size_t)nq_call * k);
      std::vector<int64_t> I((size_t)nq_call * k);
      for (int i = 0; i < calls; i++) {
        const int at = (int)(((int64_t)(t * calls + i) * nq_call) % std::max(1, npool - nq_call + 1));
        auto a = std::chrono::steady_clock::now();
        int rc = gh_host_search(hp, retrieval_params, has_rank, 0, -1e30f, 1e30f, nq_call, pool + (size_t)at * d, k,
                                D.data(), I.data());
        auto b = std::chrono::steady_clock::now();
        if (rc) failed++;
        if (lat_us) lat_us[(size_t)t * calls + i] = std::chrono::duration<float, std::micro>(b - a).count();
      }
    });
  }
  for (auto &x : th) x.join();
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return failed.load() ? -1.0 : dt;
}
// Same clients, each with its OWN scalar filter (client t admits docids [t*stride, t*stride + span), every
// other client a NOT-IN clause), results checked: the calls are repeated one at a time afterwards and
// must be bit-identical.  Returns the number of calls whose concurrent result differed (< 0: a call
// failed); *seconds receives the wall time of the concurrent phase.
int gh_host_concurrent_filtered_check(void *hp, const char *retrieval_params, int has_rank, int nthreads, int calls,
                                      const float *pool, int npool, int d, int k, int stride, int span,
                                      double *seconds) {
  std::vector<std::vector<float>> Dc(nthreads);
  std::vector<std::vector<int64_t>> Ic(nthreads);
  std::atomic<int> failed(0);
  auto one = [&](int t, int i, float *D, int64_t *I) -> int {
    std::vector<int64_t> docs(span);
    for (int j = 0; j < span; j++) docs[j] = (int64_t)t * stride + j;
    int cnt = span, not_in = t & 1;
    const int at = (int)(((int64_t)t * calls + i) % npool);
    return gh_host_search_filtered(hp, retrieval_params, has_rank, 0, -1e30f, 1e30f, 1, pool + (size_t)at * d, k, D, I,
                                   1, docs.data(), &cnt, &not_in);
  };
  std::vector<std::thread> th;
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < nthreads; t++) {
    Dc[t].resize((size_t)calls * k);
    Ic[t].resize((size_t)calls * k);
    th.emplace_back([&, t]() {
      for (int i = 0; i < calls; i++)
        if (one(t, i, Dc[t].data() + (size_t)i * k, Ic[t].data() + (size_t)i * k)) failed++;
    });
  }
  for (auto &x : th) x.join();
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (failed.load()) return -1;
  int bad = 0;
  std::vector<float> D(k);
  std::vector<int64_t> I(k);
  for (int t = 0; t < nthreads; t++)
    for (int i = 0; i < calls; i++) {
      if (one(t, i, D.data(), I.data())) return -1;
      if (memcmp(D.data(), Dc[t].data() + (size_t)i * k, sizeof(float) * k) ||
          memcmp(I.data(), Ic[t].data() + (size_t)i * k, sizeof(int64_t) * k))
        bad++;
    }
  return bad;
}
// ---- scalar fields + requests that carry their filters as the client sent them ----
int gh_host_table_add_field(void *hp, const char *name, int data_type) {
  return ((Host *)hp)->table.AddField(name, (DataType)data_type);
}
// n more docs' raw values of one field: numeric = n * elem bytes; STRING = concatenated, lens[i] bytes each
void gh_host_table_append(void *hp, int field_id, int n, const uint8_t *raw, int elem, const int *lens) {
  Host *h = (Host *)hp;
  size_t off = 0;
  for (int i = 0; i < n; i++) {
    const int len = lens ? lens[i] : elem;
    h->table.AppendValue(field_id, std::string(reinterpret_cast<const char *>(raw) + off, (size_t)len));
    off += (size_t)len;
  }
}
struct HRange {
  const char *field;
  const uint8_t *lower, *upper;
  int nbytes, include_lower, include_upper;
};
struct HTerm {
  const char *field, *value;
  int value_len, is_union;
};
int gh_host_search_scalar(void *hp, const char *retrieval_params, int has_rank, int brute_force, int n, const float *x,
                          int k, float *distances, int64_t *ids, int n_range, const HRange *rg, int n_term,
                          const HTerm *tm) {
  Host *h = (Host *)hp;
  PerfTool perf;
  GammaSearchCondition cond(&perf);
  cond.topn = k;
  cond.has_rank = has_rank != 0;
  cond.brute_force_search = brute_force != 0;
  cond.min_score = -1e30f;
  cond.max_score = 1e30f;
  cond.table = &h->table;
  for (int i = 0; i < n_range; i++)
    cond.range_filters.push_back(RangeFilter{rg[i].field, std::string((const char *)rg[i].lower, rg[i].nbytes),
                                             std::string((const char *)rg[i].upper, rg[i].nbytes),
                                             rg[i].include_lower != 0, rg[i].include_upper != 0});
  for (int i = 0; i < n_term; i++)
    cond.term_filters.push_back(TermFilter{tm[i].field, std::string(tm[i].value, tm[i].value_len), tm[i].is_union});
  cond.retrieval_params_ = h->model->Parse(retrieval_params);
  if (!cond.retrieval_params_) return -100;
  return h->model->Search(&cond, n, reinterpret_cast<const uint8_t *>(x), k, distances, ids);
}
int gh_host_dump(void *hp, const char *dir) { return ((Host *)hp)->model->Dump(dir); }
int gh_host_load(void *hp, const char *dir) { return ((Host *)hp)->model->Load(dir); }
long gh_host_mem_bytes(void *hp) { return ((Host *)hp)->model->GetTotalMemBytes(); }
// trained state of a HIPIVFPQ model (for parity checks against the oracle)
int gh_host_ivfpq_state(void *hp, float *cc, float *pq) {
  GammaIVFPQHIPIndex *m = dynamic_cast<GammaIVFPQHIPIndex *>(((Host *)hp)->model);
  if (!m || !m->is_trained_) return -1;
  if (cc) memcpy(cc, m->coarse_centroids_.data(), sizeof(float) * m->coarse_centroids_.size());
  if (pq) memcpy(pq, m->pq_centroids_.data(), sizeof(float) * m->pq_centroids_.size());
  return 0;
}

int gh_host_ivfpq_set_trained(void *hp, const float *cc, const float *pq) {
  GammaIVFPQHIPIndex *m = dynamic_cast<GammaIVFPQHIPIndex *>(((Host *)hp)->model);
  if (GammaIVFFlatHIPIndex *fl = dynamic_cast<GammaIVFFlatHIPIndex *>(((Host *)hp)->model)) return fl->SetTrainedCoarse(cc);
  return m ? m->SetTrained(cc, pq) : -1;
}
// HIPIVFPQModelParams::Parse for host-logic tests: out = {rc, ncentroids, nsubvector, nbits_per_idx,
// nprobe, metric(0 IP / 1 L2), bucket_init_size, bucket_max_size, has_hnsw, has_opq}
void gh_parse_ivfpq_model_params(const char *str, int *out) {
  HIPIVFPQModelParams p;
  out[0] = p.Parse(str);
  out[1] = p.ncentroids;
  out[2] = p.nsubvector;
  out[3] = p.nbits_per_idx;
  out[4] = p.nprobe;
  out[5] = (int)p.metric_type;
  out[6] = p.bucket_init_size;
  out[7] = p.bucket_max_size;
  out[8] = p.has_hnsw;
  out[9] = p.has_opq;
}
// HIPIVFPQRetrievalParameters via Parse on an un-Init'ed model: out = {rc, metric, recall_num, nprobe}
void gh_parse_ivfpq_retrieval_params(const char *str, int *out) {
  GammaIVFPQHIPIndex m;
  RetrievalParameters *rp = m.Parse(str);
  HIPIVFPQRetrievalParameters *ip = dynamic_cast<HIPIVFPQRetrievalParameters *>(rp);
  out[0] = ip ? 0 : -1;
  if (ip) {
    out[1] = (int)ip->GetDistanceComputeType();
    out[2] = ip->RecallNum();
    out[3] = ip->Nprobe();
  }
  delete rp;
}
// ---- IwPQ file IO (host only) ----
int gh_iwpq_write(const char *path, int d, int64_t ntotal, int metric, int nlist, int nprobe, const float *cc,
                  int M, const float *pqc, const int64_t *sizes, const uint8_t *codes, const int64_t *ids) {
  IwPQFile f;
  f.d = d;
  f.ntotal = ntotal;
  f.metric = metric;
  f.nlist = nlist;
  f.nprobe = nprobe;
  f.coarse.assign(cc, cc + (size_t)nlist * d);
  f.code_size = f.M = M;
  f.pq.assign(pqc, pqc + (size_t)M * 256 * (d / M));
  f.sizes.resize(nlist);
  f.codes.resize(nlist);
  f.ids.resize(nlist);
  size_t off = 0;
  for (int l = 0; l < nlist; l++) {
    const size_t n = (size_t)sizes[l];
    f.sizes[l] = n;
    f.codes[l].assign(codes + off * M, codes + (off + n) * M);
    f.ids[l].assign(ids + off, ids + off + n);
    off += n;
  }
  return WriteIwPQ(path, f);
}
// pass 1 (cc == NULL): hdr = {d, ntotal, metric, nlist, nprobe, M, nbits, code_size, by_residual, sum sizes};
// pass 2: fills cc, pqc, sizes[nlist], codes and ids (concatenated in list order)
int gh_iwpq_read(const char *path, int64_t *hdr, float *cc, float *pqc, int64_t *sizes, uint8_t *codes,
                 int64_t *ids) {
  IwPQFile f;
  const int rc = ReadIwPQ(path, &f);
  if (rc) return rc;
  size_t tot = 0;
  for (size_t l = 0; l < f.nlist; l++) tot += f.sizes[l];
  const int64_t h[10] = {f.d, f.ntotal, f.metric, (int64_t)f.nlist, (int64_t)f.nprobe, (int64_t)f.M,
                         (int64_t)f.nbits, (int64_t)f.code_size, f.by_residual ? 1 : 0, (int64_t)tot};
  memcpy(hdr, h, sizeof(h));
  if (!cc) return 0;
  memcpy(cc, f.coarse.data(), sizeof(float) * f.coarse.size());
  memcpy(pqc, f.pq.data(), sizeof(float) * f.pq.size());
  size_t off = 0;
  for (size_t l = 0; l < f.nlist; l++) {
    sizes[l] = (int64_t)f.sizes[l];
    if (f.sizes[l]) {
      memcpy(codes + off * f.code_size, f.codes[l].data(), f.codes[l].size());
      memcpy(ids + off, f.ids[l].data(), sizeof(int64_t) * f.ids[l].size());
    }
    off += f.sizes[l];
  }
  return 0;
}

int gh_model_registered(const char *name) {
  RetrievalModel *m = reflector().GetNewModel(name);
  if (!m) return 0;
  delete m;
  return 1;
}

}  // extern "C"
