// filter_bridge.h -- GammaSearchCondition's range filter -> gamma_hip_range_filter[]
#pragma once
#include <vector>

#include "../../include/gamma_hip.h"
#include "plugin_includes.h"

namespace tig_gamma {
// translate the engine's per-request filter (table/range_query_result.h) into the C ABI's POD
// descriptors; the bitmaps stay owned by the MultiRangeQueryResults
static void FillRangeFilters(GammaSearchCondition *cond, gamma_hip_search_params &p,
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

}  // namespace tig_gamma
