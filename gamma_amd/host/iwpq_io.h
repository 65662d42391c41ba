// iwpq_io.h -- reader / writer of the reference's IVFPQ dump file "ivfpq.index"
// (GammaIVFPQIndex::Dump/Load, index/impl/gamma_index_ivfpq.cc:958-1048; record layout in
// index/gamma_index_io.cc:16-192, which is faiss 1.7.1's "IwPQ" layout, faiss:impl/index_write.cpp):
//
//   u32 "IwPQ"
//   ivf header : int d | i64 ntotal | i64 1<<20 | i64 1<<20 | u8 is_trained | i32 metric
//                size_t nlist | size_t nprobe
//                quantizer  : u32 "IxF2" | index header (d, ntotal = nlist, ...) | vector<float> xb
//                direct map : u8 type (0) | vector<i64> (empty)
//   u8 by_residual | size_t code_size
//   product quantizer : size_t d | size_t M | size_t nbits | vector<float> centroids
//   inverted lists    : u32 "ilar" | size_t nlist | size_t code_size | u32 "full" | vector<size_t> sizes
//                       then for every non-empty list: codes[size*code_size] | i64 ids[size]
//                       (ids keep bit 63 = superseded; "sprs" = (list, size) pairs is also read)
// vector<T> = size_t count followed by the elements.  Host only, no device dependency.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace tig_gamma {

struct IwPQFile {
  int d = 0;
  int64_t ntotal = 0;         // the reference never advances faiss's ntotal: it dumps 0
  int metric = 1;             // faiss::MetricType: 0 inner product, 1 L2
  size_t nlist = 0, nprobe = 0;
  std::vector<float> coarse;  // nlist * d
  bool by_residual = true;
  size_t code_size = 0, M = 0, nbits = 8;
  std::vector<float> pq;      // M * 256 * (d / M)
  std::vector<size_t> sizes;  // nlist
  std::vector<std::vector<uint8_t>> codes;
  std::vector<std::vector<int64_t>> ids;
};

// 0 ok; -1 cannot open / short write or read; -2 not an IwPQ file or unsupported sub-record
int WriteIwPQ(const std::string &path, const IwPQFile &f);
int ReadIwPQ(const std::string &path, IwPQFile *f);
// The IVFFLAT model's "ivfflat.index" (GammaIndexIVFFlat::Dump/Load, index/impl/gamma_index_ivfflat.cc:620-690):
//   u32 "IvFl" | the same ivf header | the same inverted lists with code_size = 4 d (a list's codes are its vectors)
//   | int indexed_count.  by_residual / code_size / the product quantizer are absent; f.pq stays empty.
int WriteIvFl(const std::string &path, const IwPQFile &f, int indexed_count);
int ReadIvFl(const std::string &path, IwPQFile *f, int *indexed_count);

}  // namespace tig_gamma
