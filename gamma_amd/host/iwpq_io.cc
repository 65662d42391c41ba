// iwpq_io.cc -- see iwpq_io.h
#include "iwpq_io.h"

#include <stdio.h>
#include <string.h>

namespace tig_gamma {

namespace {
uint32_t fourcc(const char *s) {
  return (uint32_t)(uint8_t)s[0] | ((uint32_t)(uint8_t)s[1] << 8) | ((uint32_t)(uint8_t)s[2] << 16) |
         ((uint32_t)(uint8_t)s[3] << 24);
}

struct Out {
  explicit Out(FILE *fp) : f(fp), ok(true) {}
  FILE *f;
  bool ok;
  template <class T>
  void one(const T &v) { ok = ok && fwrite(&v, sizeof(T), 1, f) == 1; }
  template <class T>
  void raw(const T *p, size_t n) { ok = ok && (n == 0 || fwrite(p, sizeof(T), n, f) == n); }
  template <class T>
  void vec(const std::vector<T> &v) {
    one<size_t>(v.size());
    raw(v.data(), v.size());
  }
};

struct In {
  explicit In(FILE *fp) : f(fp), ok(true) {}
  FILE *f;
  bool ok;
  template <class T>
  void one(T &v) { ok = ok && fread(&v, sizeof(T), 1, f) == 1; }
  template <class T>
  void raw(T *p, size_t n) { ok = ok && (n == 0 || fread(p, sizeof(T), n, f) == n); }
  template <class T>
  void vec(std::vector<T> &v, size_t limit) {
    size_t n = 0;
    one(n);
    if (!ok || n > limit) {
      ok = false;
      return;
    }
    v.resize(n);
    raw(v.data(), n);
  }
};

void write_index_header(Out &o, int d, int64_t ntotal, int metric) {
  const int64_t dummy = 1 << 20;
  const uint8_t trained = 1;
  o.one<int>(d);
  o.one<int64_t>(ntotal);
  o.one<int64_t>(dummy);
  o.one<int64_t>(dummy);
  o.one<uint8_t>(trained);
  o.one<int>(metric);
}

bool read_index_header(In &in, int &d, int64_t &ntotal, int &metric) {
  int64_t dummy;
  uint8_t trained;
  in.one(d);
  in.one(ntotal);
  in.one(dummy);
  in.one(dummy);
  in.one(trained);
  in.one(metric);
  if (in.ok && metric > 1) {   // faiss writes metric_arg for the exotic metrics
    float arg;
    in.one(arg);
  }
  return in.ok;
}
}  // namespace

static int write_file(const std::string &path, const IwPQFile &x, bool flat, int indexed_count) {
  if (x.coarse.size() != x.nlist * (size_t)x.d || x.sizes.size() != x.nlist || x.codes.size() != x.nlist ||
      x.ids.size() != x.nlist)
    return -2;
  FILE *fp = fopen(path.c_str(), "wb");
  if (!fp) return -1;
  Out o(fp);
  o.one<uint32_t>(fourcc(flat ? "IvFl" : "IwPQ"));
  write_index_header(o, x.d, x.ntotal, x.metric);
  o.one<size_t>(x.nlist);
  o.one<size_t>(x.nprobe);
  o.one<uint32_t>(fourcc("IxF2"));   // the coarse quantizer is an IndexFlatL2 for both metrics
  write_index_header(o, x.d, (int64_t)x.nlist, 1);
  o.vec(x.coarse);
  o.one<uint8_t>(0);                 // DirectMap::NoMap
  o.one<size_t>(0);
  if (!flat) {
    o.one<uint8_t>(x.by_residual ? 1 : 0);
    o.one<size_t>(x.code_size);
    o.one<size_t>((size_t)x.d);
    o.one<size_t>(x.M);
    o.one<size_t>(x.nbits);
    o.vec(x.pq);
  }
  o.one<uint32_t>(fourcc("ilar"));
  o.one<size_t>(x.nlist);
  o.one<size_t>(x.code_size);
  o.one<uint32_t>(fourcc("full"));
  o.vec(x.sizes);
  for (size_t l = 0; l < x.nlist && o.ok; l++) {
    const size_t n = x.sizes[l];
    if (n == 0) continue;
    if (x.codes[l].size() != n * x.code_size || x.ids[l].size() != n) {
      o.ok = false;
      break;
    }
    o.raw(x.codes[l].data(), n * x.code_size);
    o.raw(x.ids[l].data(), n);
  }
  if (flat) o.one<int>(indexed_count);   // GammaIndexIVFFlat::Dump, gamma_index_ivfflat.cc:645
  const bool ok = o.ok;
  return (fclose(fp) == 0 && ok) ? 0 : -1;
}

int WriteIwPQ(const std::string &path, const IwPQFile &x) { return write_file(path, x, false, 0); }
int WriteIvFl(const std::string &path, const IwPQFile &x, int indexed_count) { return write_file(path, x, true, indexed_count); }

static int read_file(const std::string &path, IwPQFile *x, bool flat, int *indexed_count) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) return -1;
  In in(fp);
  int rc = 0;
  const size_t kMax = (size_t)1 << 36;
  do {
    uint32_t h = 0;
    in.one(h);
    if (!in.ok || h != fourcc(flat ? "IvFl" : "IwPQ")) { rc = -2; break; }
    if (!read_index_header(in, x->d, x->ntotal, x->metric)) break;
    in.one(x->nlist);
    in.one(x->nprobe);
    in.one(h);
    if (!in.ok || (h != fourcc("IxF2") && h != fourcc("IxFI") && h != fourcc("IxFl"))) { rc = -2; break; }   // hnsw quantizer: unsupported
    int qd, qm;
    int64_t qn;
    if (!read_index_header(in, qd, qn, qm)) break;
    in.vec(x->coarse, kMax);
    if (!in.ok || qd != x->d || x->coarse.size() != x->nlist * (size_t)x->d) { rc = -2; break; }
    uint8_t dm = 0;
    in.one(dm);
    std::vector<int64_t> dmap;
    in.vec(dmap, kMax);
    if (!in.ok || dm == 2) { rc = -2; break; }   // hashtable direct maps are not produced by Gamma
    if (!flat) {
      uint8_t br = 1;
      in.one(br);
      x->by_residual = br != 0;
      in.one(x->code_size);
      size_t pd = 0;
      in.one(pd);
      in.one(x->M);
      in.one(x->nbits);
      in.vec(x->pq, kMax);
      if (!in.ok || pd != (size_t)x->d) { rc = -2; break; }
    } else {
      x->code_size = sizeof(float) * (size_t)x->d;
    }
    in.one(h);
    if (!in.ok || h != fourcc("ilar")) { rc = -2; break; }   // an opq record would sit here: unsupported
    size_t nl = 0, cs = 0;
    in.one(nl);
    in.one(cs);
    uint32_t lt = 0;
    in.one(lt);
    if (!in.ok || nl != x->nlist || cs != x->code_size) { rc = -2; break; }
    std::vector<size_t> raw;
    in.vec(raw, kMax);
    if (!in.ok) break;
    x->sizes.assign(x->nlist, 0);
    if (lt == fourcc("full")) {
      if (raw.size() != x->nlist) { rc = -2; break; }
      x->sizes = raw;
    } else if (lt == fourcc("sprs")) {
      if (raw.size() & 1) { rc = -2; break; }
      for (size_t i = 0; i + 1 < raw.size(); i += 2) {
        if (raw[i] >= x->nlist) { rc = -2; break; }
        x->sizes[raw[i]] = raw[i + 1];
      }
      if (rc) break;
    } else { rc = -2; break; }
    x->codes.assign(x->nlist, std::vector<uint8_t>());
    x->ids.assign(x->nlist, std::vector<int64_t>());
    for (size_t l = 0; l < x->nlist && in.ok; l++) {
      const size_t n = x->sizes[l];
      if (n == 0) continue;
      if (n > kMax / (x->code_size + 8)) { in.ok = false; break; }
      x->codes[l].resize(n * x->code_size);
      x->ids[l].resize(n);
      in.raw(x->codes[l].data(), n * x->code_size);
      in.raw(x->ids[l].data(), n);
    }
    if (flat && indexed_count) in.one(*indexed_count);
  } while (0);
  if (rc == 0 && !in.ok) rc = -1;
  fclose(fp);
  return rc;
}

int ReadIwPQ(const std::string &path, IwPQFile *x) { return read_file(path, x, false, nullptr); }
int ReadIvFl(const std::string &path, IwPQFile *x, int *indexed_count) { return read_file(path, x, true, indexed_count); }

}  // namespace tig_gamma
