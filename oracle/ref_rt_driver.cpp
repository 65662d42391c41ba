// extern "C" driver around the REAL realtime inverted index of the reference
// (realtime/realtime_invert_index.{h,cc}, realtime/realtime_mem_data.{h,cc}), compiled from
// the sources where they lie under /root/reference by oracle/Makefile.ref.  Our own code;
// TEST INFRASTRUCTURE: pins the list-writer semantics (AddKeys / Update / Delete /
// CompactIfNeed, growth law) of oracle/gamma_oracle.c and of libgamma_hip.so.
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "realtime/realtime_invert_index.h"
#include "util/bitmap_manager.h"
#include "vector/raw_vector_common.h"

INITIALIZE_EASYLOGGINGPP

using namespace tig_gamma;

namespace {
struct RefRT {
    bitmap::BitmapManager* bm;
    VIDMgr* vid_mgr;
    realtime::RTInvertIndex* rt;
    int code_size;
};
}  // namespace

extern "C" {

void* ref_rt_new(int nlist, int code_size, int bucket_init, int bucket_max, int bitmap_bits) {
    RefRT* r = new RefRT;
    r->bm = new bitmap::BitmapManager();
    r->bm->Init(bitmap_bits);
    r->vid_mgr = new VIDMgr(false);
    r->rt = new realtime::RTInvertIndex(nlist, code_size, r->vid_mgr, r->bm, bucket_init, bucket_max);
    r->code_size = code_size;
    if (!r->rt->Init()) return nullptr;
    return r;
}
void ref_rt_free(void* h) {
    RefRT* r = (RefRT*)h;
    delete r->rt;
    delete r->vid_mgr;
    delete r->bm;
    delete r;
}
int ref_rt_add_keys(void* h, int list_no, int n, const int64_t* keys, const uint8_t* codes) {
    RefRT* r = (RefRT*)h;
    std::map<int, std::vector<long>> nk;
    std::map<int, std::vector<uint8_t>> nc;
    nk[list_no].assign(keys, keys + n);
    nc[list_no].assign(codes, codes + (size_t)n * r->code_size);
    return r->rt->AddKeys(nk, nc) ? 1 : 0;
}
int ref_rt_update(void* h, int list_no, int vid, const uint8_t* code) {
    RefRT* r = (RefRT*)h;
    std::vector<uint8_t> c(code, code + r->code_size);
    return r->rt->Update(list_no, vid, c);
}
int ref_rt_delete(void* h, const int* vids, int n) {
    RefRT* r = (RefRT*)h;
    return r->rt->Delete(const_cast<int*>(vids), n);
}
int ref_rt_bitmap_set(void* h, int docid) { return ((RefRT*)h)->bm->Set(docid); }
int ref_rt_compact_if_need(void* h) { return ((RefRT*)h)->rt->CompactIfNeed(); }
int64_t ref_rt_list_size(void* h, int l) {
    RefRT* r = (RefRT*)h;
    long* ids = nullptr;
    size_t n = 0;
    uint8_t* codes = nullptr;
    r->rt->GetIvtList(l, ids, n, codes);
    return (int64_t)n;
}
int64_t ref_rt_list_capacity(void* h, int l) {
    RefRT* r = (RefRT*)h;
    return r->rt->cur_ptr_->cur_invert_ptr_->cur_bucket_keys_[l];
}
void ref_rt_get_list(void* h, int l, int64_t* ids_out, uint8_t* codes_out) {
    RefRT* r = (RefRT*)h;
    long* ids = nullptr;
    size_t n = 0;
    uint8_t* codes = nullptr;
    r->rt->GetIvtList(l, ids, n, codes);
    memcpy(ids_out, ids, n * sizeof(int64_t));
    memcpy(codes_out, codes, n * r->code_size);
}
int64_t ref_rt_vid_pos(void* h, int64_t vid) {
    RefRT* r = (RefRT*)h;
    auto* inv = r->rt->cur_ptr_->cur_invert_ptr_;
    if (vid < 0 || (size_t)vid >= inv->nids_) return -1;
    return inv->vid_bucket_no_pos_[vid];
}

}  // extern "C"
