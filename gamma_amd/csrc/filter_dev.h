// filter_dev.h -- device-side validity predicate (GammaSearchCondition::IsValid,
// common/gamma_common_data.h:99-108): range bitmaps (table/range_query_result.h:53-67,
// 169-179) AND NOT delete bitmap (util/bitmap_manager.cc:187-192).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gh {

__device__ __forceinline__ bool bm_test(const uint8_t* bm, int64_t id) {
    return (bm[id >> 3] >> (id & 7)) & 1;
}

__device__ __forceinline__ bool is_valid_doc(const FilterDesc& f, int64_t vid) {
    const int doc = (int)vid;
    if (f.has_range) {
        if (f.n_range == 0) return false;  // MultiRangeQueryResults::Has on empty set
        for (int i = 0; i < f.n_range; i++) {
            const RangeDesc& r = f.range[i];
            bool has;
            if (r.b_not_in) {
                has = (doc < r.min_doc || doc > r.max_doc) ? true
                                                           : !bm_test(r.bitmap, doc - r.min_aligned);
            } else {
                has = (doc < r.min_doc || doc > r.max_doc) ? false
                                                           : bm_test(r.bitmap, doc - r.min_aligned);
            }
            if (!has) return false;
        }
    }
    if (f.del_bitmap && doc >= 0 && (int64_t)doc < f.del_bits && bm_test(f.del_bitmap, doc))
        return false;
    return true;
}

}  // namespace gh
