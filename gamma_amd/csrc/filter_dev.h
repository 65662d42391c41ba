// filter_dev.h -- device-side validity predicate (GammaSearchCondition::IsValid,
// common/gamma_common_data.h:99-108): range bitmaps (table/range_query_result.h:53-67,
// 169-179) AND numeric column predicates AND NOT delete bitmap (util/bitmap_manager.cc:187-192).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gh {

__device__ __forceinline__ bool bm_test(const uint8_t* bm, int64_t id) {
    return (bm[id >> 3] >> (id & 7)) & 1;
}

__device__ __forceinline__ bool is_valid_doc(const FilterDesc& f, int64_t vid) {
    // GammaSearchCondition::IsValid: docid = raw_vec->VidMgr()->VID2DocID(id) (common/gamma_common_data.h:100)
    const int doc = (f.vid2doc && vid >= 0 && vid < f.n_vid2doc) ? f.vid2doc[vid] : (int)vid;
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
    for (int i = 0; i < f.n_field; i++) {
        const FieldDesc& c = f.field[i];
        if (doc < 0 || (int64_t)doc >= c.n) return false;
        bool in;
        if (c.dtype <= 1) {
            const int64_t v = c.dtype == 0 ? (int64_t) reinterpret_cast<const int32_t*>(c.col)[doc]
                                           : reinterpret_cast<const int64_t*>(c.col)[doc];
            in = ((c.incl & 1) ? v >= c.lo_i : v > c.lo_i) && ((c.incl & 2) ? v <= c.hi_i : v < c.hi_i);
        } else if (c.dtype == 2) {
            const float v = reinterpret_cast<const float*>(c.col)[doc];
            const float lo = (float)c.lo_f, hi = (float)c.hi_f;
            in = ((c.incl & 1) ? v >= lo : v > lo) && ((c.incl & 2) ? v <= hi : v < hi);
        } else {
            const double v = reinterpret_cast<const double*>(c.col)[doc];
            in = ((c.incl & 1) ? v >= c.lo_f : v > c.lo_f) && ((c.incl & 2) ? v <= c.hi_f : v < c.hi_f);
        }
        if (!in) return false;
    }
    for (int i = 0; i < f.n_term; i++) {
        const TermDesc& t = f.term[i];
        if (doc < 0 || (int64_t)doc >= t.n) return false;
        const int64_t row = t.off[doc];   // start << 16 | number of items
        const int64_t b = row >> 16, e = b + (row & 0xffff);
        // Or: any term item among the doc's items; And: all of them (as the reference's GPU model); Not: none
        bool any = false, all = true;
        for (int k = 0; k < t.n_items; k++) {
            bool in = false;
            for (int64_t j = b; j < e && !in; j++) in = t.tok[j] == t.items[k];
            any |= in;
            all &= in;
        }
        const bool pass = t.op == 1 ? any : (t.op == 2 ? !any : all);
        if (!pass) return false;
    }
    if (f.del_bitmap && doc >= 0 && (int64_t)doc < f.del_bits && bm_test(f.del_bitmap, doc))
        return false;
    return true;
}

}  // namespace gh
