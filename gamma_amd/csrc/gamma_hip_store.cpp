// gamma_hip_store.cpp -- the writers of libgamma_hip.so: the realtime inverted-list arena (growth law, repack,
// versioned (offset, length) tables), training state, Add / Update / Delete / compaction, device-side encoding, the raw
// vector store, scalar columns and the delete bitmap.  C ABI: include/gamma_hip.h.
#include "gamma_hip_internal.h"

namespace ghi {


// ---- mapped ranges ------------------------------------------------------------------------
}  // namespace ghi
bool VmRange::reserve(int device_, size_t chunk_min) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device_;
    size_t g = 0, free_b = 0, total_b = 0;
    void* va = nullptr;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0 ||
        hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) {
        (void)hipGetLastError();
        return false;
    }
    // whole chunks of at least 2 MB: hipMemSetAccess refuses chunk sizes that are only a multiple of the reported 4 KB
    // granularity (tools/exp/vmm_probe.cpp)
    const size_t chunk = std::max<size_t>(std::max<size_t>(g, chunk_min), (size_t)2 << 20) / g * g;
    const size_t want = (total_b + chunk - 1) / chunk * chunk;
    if (hipMemAddressReserve(&va, want, 0, nullptr, 0) != hipSuccess || !va) {
        (void)hipGetLastError();
        return false;
    }
    base = static_cast<char*>(va);
    va_bytes = want;
    mapped = 0;
    gran = chunk;
    device = device_;
    return true;
}
static bool vm_unmap_fence();
hipError_t VmRange::map_to(size_t bytes, const char** what) {
    *what = "";
    if (bytes <= mapped) return hipSuccess;
    size_t add = std::max(bytes - mapped, mapped / 8);   // fewer, larger chunks
    add = (add + gran - 1) / gran * gran;
    if (mapped + add > va_bytes) add = va_bytes - mapped;
    if (mapped + add < bytes) {
        *what = "beyond the reserved address range";
        return hipErrorOutOfMemory;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemGenericAllocationHandle_t hnd;
    hipError_t e = hipMemCreate(&hnd, add, &prop, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *what = "hipMemCreate";
        return e;
    }
    char* at = base + mapped;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    // beside the kernels of the searches in flight: page-table updates of a range they do not read yet
    // (tools/exp/vmm_probe.cpp: mapping beside kernels on the range and beside a thread allocating and launching)
    e = hipMemMap(at, add, 0, hnd, 0);
    *what = "hipMemMap";
    if (e == hipSuccess) {
        e = hipMemSetAccess(at, add, &acc, 1);
        *what = "hipMemSetAccess";
        if (e != hipSuccess) {   // seen for chunks that are not a multiple of 2 MB: the whole mapped range is accepted
            (void)hipGetLastError();
            e = hipMemSetAccess(base, mapped + add, &acc, 1);
        }
        if (e != hipSuccess) {
            (void)hipMemUnmap(at, add);
            if (!vm_unmap_fence()) tainted = true;   // (the next map_to would hand out the same address)
        }
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipMemRelease(hnd);
        return e;
    }
    chunks.push_back(hnd);
    chunk_bytes.push_back(add);
    mapped += add;
    *what = "";
    return hipSuccess;
}
// An address that was unmapped and is mapped again (to other physical memory) read back stale contents -- zeros -- now and then
// on this runtime when the GPU was shared with other processes (tests/test_gpu_fuzz.py::test_random_realtime_script under six
// processes: 10-18 % of the scripts, whether the range was freed and reserved again or merely unmapped and remapped;
// hipDeviceSynchronize does not help).  An ordinary allocation and release by the runtime in between does (0 of 400): it goes
// through the driver's page-table path and leaves no stale translation behind.  Every unmap ends with one.
static bool vm_unmap_fence() {
    // (the runtime serves allocations below 2 MB from blocks it keeps: 1 MB and 4 KB fences changed nothing, 2 / 4 / 16 / 64 MB
    //  all cured it -- 0 of 600 scripts each)
    void* t = nullptr;
    for (size_t bytes : {(size_t)16 << 20, (size_t)2 << 20}) {
        if (hipMalloc(&t, bytes) == hipSuccess) {
            (void)hipFree(t);
            return true;
        }
        (void)hipGetLastError();
    }
    return false;   // the device is out of memory: the caller must not map these addresses again
}
void VmRange::unmap_all() {
    if (!base) return;
    size_t off = 0;
    for (size_t i = 0; i < chunks.size(); i++) {
        (void)hipMemUnmap(base + off, chunk_bytes[i]);
        (void)hipMemRelease(chunks[i]);
        off += chunk_bytes[i];
    }
    if (!chunks.empty() && !vm_unmap_fence()) tainted = true;
    chunks.clear();
    chunk_bytes.clear();
    mapped = 0;
}
void VmRange::release() {
    if (!base) return;
    size_t off = 0;
    for (size_t i = 0; i < chunks.size(); i++) {
        (void)hipMemUnmap(base + off, chunk_bytes[i]);
        (void)hipMemRelease(chunks[i]);
        off += chunk_bytes[i];
    }
    // (no fence: the address range is NOT given back -- whoever reserved it next would map addresses with stale
    //  translations; it stays reserved for the life of the process, address space only)
    if (chunks.empty() || vm_unmap_fence()) (void)hipMemAddressFree(base, va_bytes);
    chunks.clear();
    chunk_bytes.clear();
    base = nullptr;
    va_bytes = mapped = 0;
}
namespace ghi {

static int vm_fail(H* h, const char* who, hipError_t e, const char* what) {
    if (e == hipErrorOutOfMemory) return fail(h, GAMMA_HIP_ENOMEM, (std::string(who) + ": out of device memory (" + what + ")").c_str());
    h->err = std::string(who) + ": " + what + " failed: " + hipGetErrorString(e);
    return GAMMA_HIP_EDEVICE;
}

// ---- arena ---------------------------------------------------------------------------
// entries the three mapped arrays hold
static int64_t arena_vm_cap(const H* h) {
    int64_t c = std::min<int64_t>((int64_t)(h->vm_codes.mapped / (size_t)h->code_size), (int64_t)(h->vm_ids.mapped / sizeof(int64_t)));
    if (h->keep_sums) c = std::min<int64_t>(c, (int64_t)(h->vm_sums.mapped / sizeof(float)));
    return c;
}
// the first reservation of an index: mapped ranges when the runtime offers them AND the first chunks map (a failure
// here leaves the reallocating arena, nothing behind)
static bool arena_vm_start(H* h, int64_t entries) {
    if (getenv("GAMMA_HIP_NO_ARENA_VMM")) return false;
    const size_t chunk = (size_t)8 << 20;
    const char* what = "";
    bool ok = h->vm_codes.reserve(h->device, chunk) && h->vm_ids.reserve(h->device, chunk) &&
              (!h->keep_sums || h->vm_sums.reserve(h->device, chunk));
    ok = ok && h->vm_codes.map_to((size_t)entries * h->code_size, &what) == hipSuccess &&
         h->vm_ids.map_to((size_t)entries * sizeof(int64_t), &what) == hipSuccess &&
         (!h->keep_sums || h->vm_sums.map_to((size_t)entries * sizeof(float), &what) == hipSuccess);
    if (!ok) {
        h->vm_codes.release();
        h->vm_ids.release();
        h->vm_sums.release();
        (void)hipGetLastError();
        return false;
    }
    return true;
}
static void arena_vm_adopt(H* h) {
    h->d_codes = reinterpret_cast<uint8_t*>(h->vm_codes.base);
    h->d_ids = reinterpret_cast<int64_t*>(h->vm_ids.base);
    h->d_sums = h->keep_sums ? reinterpret_cast<float*>(h->vm_sums.base) : nullptr;
    h->arena_cap = arena_vm_cap(h);
}

int arena_reserve(H* h, int64_t need_entries) {
    if (h->arena_used + need_entries <= h->arena_cap) return GAMMA_HIP_OK;
    if (!h->d_codes && h->arena_cap == 0 && !h->arena_vmm && arena_vm_start(h, h->arena_used + need_entries)) {
        h->arena_vmm = true;
        arena_vm_adopt(h);
        return GAMMA_HIP_OK;
    }
    if (h->arena_vmm) {
        // map more physical memory behind the three arrays in place: nothing moves, the searches in flight go on
        const int64_t need = h->arena_used + need_entries;
        const char* what = "";
        hipError_t e = h->vm_codes.map_to((size_t)need * h->code_size, &what);
        if (e == hipSuccess) e = h->vm_ids.map_to((size_t)need * sizeof(int64_t), &what);
        if (e == hipSuccess && h->keep_sums) e = h->vm_sums.map_to((size_t)need * sizeof(float), &what);
        h->arena_cap = arena_vm_cap(h);
        if (e != hipSuccess) return vm_fail(h, "arena growth", e, what);
        return GAMMA_HIP_OK;
    }
    h->arena_regrows++;
    int64_t ncap = std::max<int64_t>(h->arena_cap * 2, h->arena_used + need_entries);
    ncap += ncap / 8;
    uint8_t* nc = nullptr;
    int64_t* ni = nullptr;
    if (h->wl) GH_CHECK(h, h->wl->exclusive());   // the old arrays are freed below: no search may be reading them
    GH_CHECK(h, hipMalloc((void**)&nc, (size_t)ncap * h->code_size));
    if (hipMalloc((void**)&ni, (size_t)ncap * sizeof(int64_t)) != hipSuccess) {
        (void)hipFree(nc);
        return fail(h, GAMMA_HIP_ENOMEM, "arena growth: out of memory");
    }
    float* ns = nullptr;
    if (h->keep_sums && hipMalloc((void**)&ns, (size_t)ncap * sizeof(float)) != hipSuccess) {
        (void)hipFree(nc);
        (void)hipFree(ni);
        return fail(h, GAMMA_HIP_ENOMEM, "arena growth: out of memory");
    }
    if (ns && h->d_sums && h->arena_used > 0)
        GH_CHECK(h, hipMemcpyAsync(ns, h->d_sums, (size_t)h->arena_used * sizeof(float), hipMemcpyDeviceToDevice, h->wstream));
    if (h->arena_used > 0) {
        GH_CHECK(h, hipMemcpyAsync(nc, h->d_codes, (size_t)h->arena_used * h->code_size,
                                   hipMemcpyDeviceToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(ni, h->d_ids, (size_t)h->arena_used * sizeof(int64_t),
                                   hipMemcpyDeviceToDevice, h->wstream));
    }
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    if (h->d_codes) GH_CHECK(h, hipFree(h->d_codes));
    if (h->d_ids) GH_CHECK(h, hipFree(h->d_ids));
    if (h->d_sums) GH_CHECK(h, hipFree(h->d_sums));
    h->d_codes = nc;
    h->d_ids = ni;
    h->d_sums = ns;
    h->arena_cap = ncap;
    return GAMMA_HIP_OK;
}

// sums of arena entries [pos, pos + n) of list l, behind the copy of their codes on the writer stream
static int sums_range(H* h, int l, int64_t pos, int n) {
    if (!h->d_sums || !h->trained || n <= 0) return GAMMA_HIP_OK;
    gh::launch_code_sums_one(h->wstream, h->d_T2, h->d_codes, h->code_size, l, pos, n, h->d_sums);
    return GAMMA_HIP_OK;
}
// every list at its current extent (after a repack, a compaction, a new table): the device tables of the current
// version must be the host mirror's (publish_meta before)
static int sums_all_lists(H* h) {
    if (!h->d_sums || !h->trained) return GAMMA_HIP_OK;
    gh::launch_code_sums_lists(h->wstream, h->d_T2, h->d_codes, h->code_size, h->d_list_off, h->d_list_len, h->nlist,
                               std::max(1, h->max_list_len), h->d_sums);
    return GAMMA_HIP_OK;
}

// The reference frees a bucket's old memory after a grow / compact swap (delayed by 1 s,
// realtime_mem_data.cc:457-466).  Here grown and compacted extents are abandoned inside the arena
// (arena_waste); once they are more than half of what is in use -- and worth at least a megabyte of codes --
// every list moves into a fresh, tight arena: one kernel, offsets re-published in stream order.
// Safe by construction (round 5): the version of the list tables that points at the new extents is published only after
// the target has been READ BACK through its new mapping and found equal to the source -- per-list checksums of ids + codes
// (k_list_checksum) taken at the source before the move and at the target in a launch of its own, behind a translation
// fence (the stale-translation failure seen on this runtime lets a kernel read back what it wrote itself; a later launch
// over fresh translations is what a search would see).  A mismatch keeps the old version (nothing has been freed), retires
// the target's address ranges, counts (gamma_hip_ivfpq_repack_verify_stats, logged by the plugins) and repeats the move
// into ordinary hipMalloc'd arrays -- the handle's arena then stays outside virtual memory management.  The fence after
// every unmap stays as belt and braces.
static int repack_checksums(H* h, const uint8_t* codes, const int64_t* ids, const int64_t* d_off, std::vector<unsigned long long>* out) {
    GH_CHECK(h, h->we_chk.ensure((size_t)h->nlist * sizeof(unsigned long long)));
    gh::launch_list_checksum(h->wstream, codes, ids, d_off, h->d_list_len, h->nlist, h->code_size, std::max(1, h->max_list_len),
                             h->we_chk.as<unsigned long long>());
    out->resize(h->nlist);
    GH_CHECK(h, hipMemcpyAsync(out->data(), h->we_chk.p, (size_t)h->nlist * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

// one attempt: GAMMA_HIP_OK = moved, verified, committed; kRepackMismatch = the target did not read back as written (old
// version intact, target given back / retired); anything else = error (old version intact)
static const int kRepackMismatch = 1;
static int arena_repack_once(H* h, bool vmm, const std::vector<int64_t>& noff, int64_t total, int64_t ncap,
                             const std::vector<unsigned long long>& src_sum) {
    uint8_t* nc = nullptr;
    int64_t* ni = nullptr;
    float* ns = nullptr;
    // Mapped arena: the handle keeps TWO sets of address ranges and the repack moves the lists from the set in use into the
    // other one (physical chunks mapped there for the occasion), gives the old set's physical memory back and swaps the roles.
    struct AltGuard {   // an error below gives the target set's physical memory back
        H* h;
        bool armed = true;
        ~AltGuard() {
            if (armed) {
                h->alt_codes.unmap_all();
                h->alt_ids.unmap_all();
                h->alt_sums.unmap_all();
            }
        }
    } alt_guard{h, vmm};
    struct MallocGuard {
        void* p[3] = {nullptr, nullptr, nullptr};
        bool armed = true;
        ~MallocGuard() {
            if (armed)
                for (void* q : p)
                    if (q) (void)hipFree(q);
        }
    } mg;
    if (vmm) {
        for (VmRange* r : {&h->alt_codes, &h->alt_ids, &h->alt_sums})
            if (r->tainted) {   // unmapped without a fence, or failed a read-back: these addresses are not mapped again
                h->vm_retired.push_back(std::move(*r));
                *r = VmRange();
            }
        const size_t chunk = (size_t)8 << 20;
        const char* what = "";
        hipError_t e = hipSuccess;
        if ((!h->alt_codes.on() && !h->alt_codes.reserve(h->device, chunk)) || (!h->alt_ids.on() && !h->alt_ids.reserve(h->device, chunk)) ||
            (h->keep_sums && !h->alt_sums.on() && !h->alt_sums.reserve(h->device, chunk))) {
            e = hipErrorOutOfMemory;
            what = "address range";
        }
        if (e == hipSuccess) e = h->alt_codes.map_to((size_t)ncap * h->code_size, &what);
        if (e == hipSuccess) e = h->alt_ids.map_to((size_t)ncap * sizeof(int64_t), &what);
        if (e == hipSuccess && h->keep_sums) e = h->alt_sums.map_to((size_t)ncap * sizeof(float), &what);
        if (e != hipSuccess) return vm_fail(h, "arena repack", e, what);
        nc = reinterpret_cast<uint8_t*>(h->alt_codes.base);
        ni = reinterpret_cast<int64_t*>(h->alt_ids.base);
        ns = h->keep_sums ? reinterpret_cast<float*>(h->alt_sums.base) : nullptr;
    } else {
        if (hipMalloc((void**)&nc, (size_t)ncap * h->code_size) != hipSuccess) return fail(h, GAMMA_HIP_ENOMEM, "arena repack: out of memory");
        mg.p[0] = nc;
        if (hipMalloc((void**)&ni, (size_t)ncap * sizeof(int64_t)) != hipSuccess) return fail(h, GAMMA_HIP_ENOMEM, "arena repack: out of memory");
        mg.p[1] = ni;
        if (h->keep_sums && hipMalloc((void**)&ns, (size_t)ncap * sizeof(float)) != hipSuccess)
            return fail(h, GAMMA_HIP_ENOMEM, "arena repack: out of memory");
        mg.p[2] = ns;
    }
    GH_CHECK(h, h->we_stage.ensure((size_t)h->nlist * sizeof(int64_t)));
    GH_CHECK(h, hipMemcpyAsync(h->we_stage.p, noff.data(), (size_t)h->nlist * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
    gh::launch_repack_lists(h->wstream, h->d_codes, h->d_ids, nc, ni, h->d_list_off, h->we_stage.as<int64_t>(),
                            h->d_list_len, h->nlist, h->code_size, h->max_list_len);
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    static const bool no_verify = getenv("GAMMA_HIP_NO_REPACK_VERIFY") != nullptr;
    if (!no_verify) {
        if (vmm) (void)vm_unmap_fence();   // translations the move may have cached are gone before the read-back
        // fault injection for the tests (GAMMA_HIP_FAULT_REPACK=<n>: the first n read-backs see a zeroed entry)
        static std::atomic<int> fault_left{getenv("GAMMA_HIP_FAULT_REPACK") ? atoi(getenv("GAMMA_HIP_FAULT_REPACK")) : 0};
        if (total > 0 && fault_left.load() > 0 && fault_left.fetch_sub(1) > 0) {
            int l0 = 0;
            while (l0 < h->nlist - 1 && h->h_list_len[l0] == 0) l0++;
            GH_CHECK(h, hipMemsetAsync(ni + noff[l0], 0, sizeof(int64_t), h->wstream));
            GH_CHECK(h, hipMemsetAsync(nc + noff[l0] * h->code_size, 0xa5, (size_t)h->code_size, h->wstream));
        }
        std::vector<unsigned long long> dst_sum;
        GH_TRY(repack_checksums(h, nc, ni, h->we_stage.as<int64_t>(), &dst_sum));
        h->repack_verified++;
        if (dst_sum != src_sum) {
            h->repack_verify_failures++;
            if (vmm) {   // the guard unmaps the target (with its fence); its addresses are never mapped again
                h->alt_codes.tainted = h->alt_ids.tainted = true;
                if (h->keep_sums) h->alt_sums.tainted = true;
            }
            return kRepackMismatch;
        }
    }
    if (vmm) {
        alt_guard.armed = false;   // the target set is the arena from here
        h->vm_codes.unmap_all();
        h->vm_ids.unmap_all();
        h->vm_sums.unmap_all();
        std::swap(h->vm_codes, h->alt_codes);
        std::swap(h->vm_ids, h->alt_ids);
        std::swap(h->vm_sums, h->alt_sums);
        arena_vm_adopt(h);
    } else {
        mg.armed = false;
        if (h->arena_vmm) {   // leaving virtual memory management after a failed read-back: the mapped sets go
            h->vm_codes.unmap_all();
            h->vm_ids.unmap_all();
            h->vm_sums.unmap_all();
            h->arena_vmm = false;
        } else {
            GH_CHECK(h, hipFree(h->d_codes));
            GH_CHECK(h, hipFree(h->d_ids));
            if (h->d_sums) GH_CHECK(h, hipFree(h->d_sums));
        }
        h->d_codes = nc;
        h->d_ids = ni;
        h->d_sums = ns;
        h->arena_cap = ncap;
    }
    h->arena_used = total;
    h->arena_waste = 0;
    h->h_list_off = noff;
    h->n_repacks++;
    GH_TRY(publish_meta(h));
    return sums_all_lists(h);   // the sums follow their codes: recomputed at the new offsets (the caller drains the stream)
}

int arena_repack(H* h) {
    int64_t total = 0;
    std::vector<int64_t> noff(h->nlist);
    for (int l = 0; l < h->nlist; l++) {
        noff[l] = total;
        total += h->h_list_cap[l];
    }
    const int64_t ncap = total + total / 8 + 1024;
    if (h->wl) GH_CHECK(h, h->wl->exclusive());   // every list moves and the old arrays are freed
    GH_TRY(publish_meta(h));                      // the device tables the kernels below read = the host mirror
    std::vector<unsigned long long> src_sum;
    GH_TRY(repack_checksums(h, h->d_codes, h->d_ids, h->d_list_off, &src_sum));
    int rc = arena_repack_once(h, h->arena_vmm, noff, total, ncap, src_sum);
    if (rc == kRepackMismatch && h->arena_vmm) rc = arena_repack_once(h, false, noff, total, ncap, src_sum);
    if (rc == kRepackMismatch)
        return fail(h, GAMMA_HIP_EDEVICE, "arena repack: the moved lists did not read back as written (the previous arena stays in use)");
    return rc;
}
int arena_repack_if_need(H* h) {
    const int64_t min_waste = std::max<int64_t>(h->repack_min_entries, 1);
    if (h->arena_waste < min_waste || h->arena_waste * 2 < h->arena_used) return GAMMA_HIP_OK;
    return arena_repack(h);
}

// RealTimeMemData::ExtendBucketIfNeed + RTInvertBucketData::ExtendBucketMem
// (realtime_mem_data.cc:383-421,152-188): same growth law, region moved inside the arena.
int list_ensure(H* h, int l, int add) {
    const int len = h->h_list_len[l], cap = h->h_list_cap[l];
    if ((int64_t)len + add <= cap) return GAMMA_HIP_OK;
    if ((int64_t)cap * 2 >= h->bucket_max) return fail(h, GAMMA_HIP_EFULL, "exceed the max bucket keys");
    const int least = len + add;
    // the growth step is worked out on a copy of the list's extend counter and committed only once the arena holds
    // the new extent (a failed reservation leaves the list as it was)
    uint8_t et = h->h_extend_time[l];
    double coef = extend_coefficient(++et);
    int ext = (int)(cap * coef);
    while (ext < least) {
        coef = extend_coefficient(++et);
        ext = (int)(ext * coef);
    }
    GH_TRY(arena_reserve(h, ext));
    h->h_extend_time[l] = et;
    const int64_t noff = h->arena_used;
    if (len > 0) {
        GH_CHECK(h, hipMemcpyAsync(h->d_codes + noff * h->code_size,
                                   h->d_codes + h->h_list_off[l] * h->code_size,
                                   (size_t)len * h->code_size, hipMemcpyDeviceToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(h->d_ids + noff, h->d_ids + h->h_list_off[l],
                                   (size_t)len * sizeof(int64_t), hipMemcpyDeviceToDevice, h->wstream));
        if (h->d_sums)
            GH_CHECK(h, hipMemcpyAsync(h->d_sums + noff, h->d_sums + h->h_list_off[l], (size_t)len * sizeof(float),
                                       hipMemcpyDeviceToDevice, h->wstream));
    }
    h->arena_waste += cap;
    h->arena_used += ext;
    h->h_list_off[l] = noff;   // the old extent stays intact: searches in flight read it through their version
    h->h_list_cap[l] = ext;
    return GAMMA_HIP_OK;
}

int add_keys_locked(H* h, int l, int n, const int64_t* vids, const uint8_t* codes) {
    if (l < 0 || l >= h->nlist || n < 0) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
    if (n == 0) return GAMMA_HIP_OK;
    GH_TRY(list_ensure(h, l, n));
    const int64_t pos = h->h_list_off[l] + h->h_list_len[l];
    GH_CHECK(h, hipMemcpyAsync(h->d_ids + pos, vids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice,
                               h->wstream));
    GH_CHECK(h, hipMemcpyAsync(h->d_codes + pos * h->code_size, codes, (size_t)n * h->code_size,
                               hipMemcpyHostToDevice, h->wstream));
    GH_TRY(sums_range(h, l, pos, n));
    for (int i = 0; i < n; i++) {
        const int64_t v = vids[i];
        if (v < 0) {   // superseded slot restored from a dump (ReadInvertedLists, gamma_index_io.cc:186-189)
            h->h_deleted[l]++;
            h->n_moved++;
            continue;
        }
        if ((size_t)v >= h->vid_pos.size()) h->vid_pos.resize(std::max<size_t>(h->vid_pos.size() * 2, v + 1), -1);
        h->vid_pos[v] = ((int64_t)l << 32) | (int64_t)(h->h_list_len[l] + i);
        if (h->doc_deleted(v)) h->h_deleted[l]++;  // realtime_mem_data.cc:293-296
    }
    h->h_list_len[l] += n;  // publish after the copies (realtime_mem_data.cc:299-300)
    h->ntotal += n;
    GH_TRY(publish_meta(h));
    // the host buffers may be reused by the caller as soon as we return
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return arena_repack_if_need(h);
}
// Spatial order of the coarse centroids by recursive principal-axis bisection: split the set at
// the median of its projection on the dominant direction (a few power iterations), recurse.
// Neighbouring ranks = neighbouring centroids.  Used only to order queries for cache locality.
void centroid_order_rec(const float* cc, int d, std::vector<int>& idx, int lo, int hi, std::vector<float>& proj,
                        std::vector<double>& mean, std::vector<double>& dir, std::vector<double>& tmp) {
    const int n = hi - lo;
    if (n <= 2) return;
    for (int t = 0; t < d; t++) mean[t] = 0;
    for (int i = lo; i < hi; i++) {
        const float* c = cc + (size_t)idx[i] * d;
        for (int t = 0; t < d; t++) mean[t] += c[t];
    }
    for (int t = 0; t < d; t++) mean[t] /= n;
    // start from the direction to the point farthest from the mean (never orthogonal to the data)
    double best = -1;
    int far = lo;
    for (int i = lo; i < hi; i++) {
        const float* c = cc + (size_t)idx[i] * d;
        double s = 0;
        for (int t = 0; t < d; t++) s += (c[t] - mean[t]) * (c[t] - mean[t]);
        if (s > best) { best = s; far = i; }
    }
    for (int t = 0; t < d; t++) dir[t] = cc[(size_t)idx[far] * d + t] - mean[t];
    for (int it = 0; it < 6; it++) {
        for (int t = 0; t < d; t++) tmp[t] = 0;
        for (int i = lo; i < hi; i++) {
            const float* c = cc + (size_t)idx[i] * d;
            double pr = 0;
            for (int t = 0; t < d; t++) pr += (c[t] - mean[t]) * dir[t];
            for (int t = 0; t < d; t++) tmp[t] += pr * (c[t] - mean[t]);
        }
        double nrm = 0;
        for (int t = 0; t < d; t++) nrm += tmp[t] * tmp[t];
        if (nrm <= 0) break;
        nrm = std::sqrt(nrm);
        for (int t = 0; t < d; t++) dir[t] = tmp[t] / nrm;
    }
    for (int i = lo; i < hi; i++) {
        const float* c = cc + (size_t)idx[i] * d;
        double pr = 0;
        for (int t = 0; t < d; t++) pr += (c[t] - mean[t]) * dir[t];
        proj[idx[i]] = (float)pr;
    }
    const int mid = lo + n / 2;
    std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                     [&](int a, int b) { return proj[a] < proj[b] || (proj[a] == proj[b] && a < b); });
    centroid_order_rec(cc, d, idx, lo, mid, proj, mean, dir, tmp);
    centroid_order_rec(cc, d, idx, mid, hi, proj, mean, dir, tmp);
}

std::vector<int> centroid_rank(const float* cc, int nlist, int d) {
    std::vector<int> idx(nlist), rank(nlist);
    for (int i = 0; i < nlist; i++) idx[i] = i;
    std::vector<float> proj(nlist);
    std::vector<double> mean(d), dir(d), tmp(d);
    centroid_order_rec(cc, d, idx, 0, nlist, proj, mean, dir, tmp);
    for (int i = 0; i < nlist; i++) rank[idx[i]] = i;
    return rank;
}

}  // namespace ghi

using namespace ghi;

extern "C" {


int gamma_hip_field_append(gamma_hip_index* h, int field_id, int dtype, int64_t n, const void* values) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (dtype < GAMMA_HIP_FIELD_INT || dtype > GAMMA_HIP_FIELD_DOUBLE || n < 0 || (n > 0 && !values))
        return fail(h, GAMMA_HIP_EINVAL, "bad column append");
    GH_CHECK(h, hipSetDevice(h->device));
    auto& c = h->fields[field_id];
    if (c.n == 0 && c.cap == 0) c.dtype = dtype;
    if (c.dtype != dtype) return fail(h, GAMMA_HIP_EINVAL, "column dtype mismatch");
    const size_t es = field_elem_size(dtype);
    if (c.n + n > c.cap) {
        const int64_t ncap = std::max<int64_t>(c.n + n, std::max<int64_t>(1 << 16, c.cap * 2));
        uint8_t* nd = nullptr;
        GH_CHECK(h, lk.exclusive());   // the old column is freed below
        GH_CHECK(h, hipMalloc((void**)&nd, (size_t)ncap * es));
        if (c.n) GH_CHECK(h, hipMemcpyAsync(nd, c.d, (size_t)c.n * es, hipMemcpyDeviceToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        if (c.d) (void)hipFree(c.d);
        c.d = nd;
        c.cap = ncap;
    }
    if (n) {
        GH_CHECK(h, hipMemcpyAsync(c.d + (size_t)c.n * es, values, (size_t)n * es, hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
    }
    c.n += n;
    return GAMMA_HIP_OK;
}

int gamma_hip_field_update(gamma_hip_index* h, int field_id, int64_t docid, const void* value) {
    if (!h || !value) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    auto it = h->fields.find(field_id);
    if (it == h->fields.end() || docid < 0 || docid >= it->second.n) return fail(h, GAMMA_HIP_EINVAL, "bad column update");
    GH_CHECK(h, hipSetDevice(h->device));
    const size_t es = field_elem_size(it->second.dtype);
    GH_CHECK(h, hipMemcpyAsync(it->second.d + (size_t)docid * es, value, es, hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

int gamma_hip_vid2docid_append(gamma_hip_index* h, int64_t n, const int32_t* docids) {
    if (!h || n < 0 || (n > 0 && !docids)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    const int64_t have = (int64_t)h->h_v2d.size();
    if (have + n > h->v2d_cap) {
        const int64_t ncap = std::max<int64_t>(have + n, std::max<int64_t>(1 << 16, h->v2d_cap * 2));
        int32_t* nd = nullptr;
        GH_CHECK(h, lk.exclusive());   // the old array is freed below
        GH_CHECK(h, hipMalloc((void**)&nd, (size_t)ncap * sizeof(int32_t)));
        if (have) GH_CHECK(h, hipMemcpyAsync(nd, h->d_v2d, (size_t)have * sizeof(int32_t), hipMemcpyDeviceToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        if (h->d_v2d) (void)hipFree(h->d_v2d);
        h->d_v2d = nd;
        h->v2d_cap = ncap;
    }
    if (n) {
        GH_CHECK(h, hipMemcpyAsync(h->d_v2d + have, docids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        h->h_v2d.insert(h->h_v2d.end(), docids, docids + n);   // published last: searches enqueued before see the shorter map
    }
    return GAMMA_HIP_OK;
}

int64_t gamma_hip_vid2docid_count(gamma_hip_index* h) {
    if (!h) return -1;
    std::lock_guard<std::mutex> g(h->mu);
    return (int64_t)h->h_v2d.size();
}

int64_t gamma_hip_field_count(gamma_hip_index* h, int field_id) {
    if (!h) return -1;
    std::lock_guard<std::mutex> g(h->mu);
    auto it = h->fields.find(field_id);
    return it == h->fields.end() ? 0 : it->second.n;
}

// A STRING column on the device: doc i = items tok[start .. start + len), its row word = start << 16 | len.  One 8-byte
// word per doc (not CSR offsets) so that a doc's items can be REWRITTEN: the new items go in place when they fit, else
// to the end of the item array, and the row word -- a single aligned store -- switches the doc over.
static int term_reserve(gamma_hip_index* h, WriteLock& lk, gamma_hip_index::TermColumn& c, int64_t add_docs, int64_t add_tok) {
    if (c.ndocs + add_docs <= c.cap_docs && c.ntok + add_tok <= c.cap_tok && c.d_off) return GAMMA_HIP_OK;
    // growth frees the old arrays: no search may be reading them
    GH_CHECK(h, lk.exclusive());
    const int64_t nd = std::max<int64_t>(c.ndocs + add_docs, std::max<int64_t>(1 << 16, c.cap_docs * 2));
    const int64_t nt = std::max<int64_t>(c.ntok + add_tok, std::max<int64_t>(1 << 16, c.cap_tok * 2));
    int64_t* no = nullptr;
    int32_t* ntk = nullptr;
    GH_CHECK(h, hipMalloc((void**)&no, (size_t)nd * sizeof(int64_t)));
    if (hipMalloc((void**)&ntk, (size_t)nt * sizeof(int32_t)) != hipSuccess) {
        (void)hipFree(no);
        return fail(h, GAMMA_HIP_ENOMEM, "term column: out of memory");
    }
    if (c.d_off) {
        if (c.ndocs) GH_CHECK(h, hipMemcpyAsync(no, c.d_off, (size_t)c.ndocs * sizeof(int64_t), hipMemcpyDeviceToDevice, h->wstream));
        if (c.ntok) GH_CHECK(h, hipMemcpyAsync(ntk, c.d_tok, (size_t)c.ntok * sizeof(int32_t), hipMemcpyDeviceToDevice, h->wstream));
    }
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    if (c.d_off) (void)hipFree(c.d_off);
    if (c.d_tok) (void)hipFree(c.d_tok);
    c.d_off = no;
    c.d_tok = ntk;
    c.cap_docs = nd;
    c.cap_tok = nt;
    return GAMMA_HIP_OK;
}

int gamma_hip_term_append(gamma_hip_index* h, int field_id, int64_t n_docs, const int32_t* counts,
                          const int32_t* items) {
    if (!h || n_docs < 0 || (n_docs > 0 && !counts)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    auto& c = h->terms[field_id];
    int64_t add_tok = 0;
    for (int64_t i = 0; i < n_docs; i++) {
        if (counts[i] < 0 || counts[i] > 65535) return fail(h, GAMMA_HIP_EINVAL, "item count out of range");
        add_tok += counts[i];
    }
    if (add_tok > 0 && !items) return fail(h, GAMMA_HIP_EINVAL, "null items");
    GH_TRY(term_reserve(h, lk, c, n_docs, add_tok));
    if (n_docs > 0) {
        std::vector<int64_t> rows(n_docs);
        int64_t run = c.ntok;
        for (int64_t i = 0; i < n_docs; i++) {
            rows[i] = (run << 16) | (int64_t)counts[i];
            run += counts[i];
        }
        if (add_tok) GH_CHECK(h, hipMemcpyAsync(c.d_tok + c.ntok, items, (size_t)add_tok * sizeof(int32_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(c.d_off + c.ndocs, rows.data(), (size_t)n_docs * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        c.h_rows.insert(c.h_rows.end(), rows.begin(), rows.end());
        c.ntok += add_tok;
        c.ndocs += n_docs;   // published last: a search enqueued before sees the shorter column
    }
    return GAMMA_HIP_OK;
}

// a doc's items rewritten (the doc's STRING field was updated in the table)
int gamma_hip_term_update(gamma_hip_index* h, int field_id, int64_t docid, int32_t count, const int32_t* items) {
    if (!h || count < 0 || count > 65535 || (count > 0 && !items)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    auto it = h->terms.find(field_id);
    if (it == h->terms.end() || docid < 0 || docid >= it->second.ndocs) return fail(h, GAMMA_HIP_EINVAL, "term update: unknown field / doc");
    auto& c = it->second;
    const int64_t old = c.h_rows[docid];
    int64_t start = old >> 16;
    if (count > (int32_t)(old & 0xffff)) {   // does not fit in place: the new items go to the end of the item array
        GH_TRY(term_reserve(h, lk, c, 0, count));
        start = c.ntok;
        c.ntok += count;
    }
    const int64_t row = (start << 16) | (int64_t)count;
    if (count) GH_CHECK(h, hipMemcpyAsync(c.d_tok + start, items, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipMemcpyAsync(c.d_off + docid, &row, sizeof(row), hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    c.h_rows[docid] = row;
    return GAMMA_HIP_OK;
}

int64_t gamma_hip_term_count(gamma_hip_index* h, int field_id) {
    if (!h) return -1;
    std::lock_guard<std::mutex> g(h->mu);
    auto it = h->terms.find(field_id);
    return it == h->terms.end() ? 0 : it->second.ndocs;
}

int gamma_hip_raw_init(gamma_hip_index* h, int d) {
    if (!h || d <= 0) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->raw_d != 0 && h->raw_d != d) return fail(h, GAMMA_HIP_EINVAL, "raw store dimension mismatch");
    if (h->raw_d == 0 && !getenv("GAMMA_HIP_NO_RAW_VMM")) {
        // reserve the address range the store may ever need (the device's memory): physical chunks are mapped into it
        // as rows arrive (raw_reserve).  Any failure -- here or of the FIRST chunk -- leaves the reallocating store.
        GH_CHECK(h, hipSetDevice(h->device));
        if (h->raw_vm.reserve(h->device, (size_t)64 << 20)) {   // chunks of whole 64 MB
            h->raw_vmm = true;
            h->d_raw = reinterpret_cast<float*>(h->raw_vm.base);
        }
    }
    h->raw_d = d;
    return GAMMA_HIP_OK;
}

static int raw_reserve(H* h, int64_t need) {
    if (need <= h->raw_cap) return GAMMA_HIP_OK;
    if (h->raw_vmm) {
        // map more physical memory behind the rows in place: nothing moves
        const size_t row = (size_t)h->raw_d * sizeof(float);
        const char* what = "";
        hipError_t e = h->raw_vm.map_to((size_t)need * row, &what);
        if (e == hipSuccess) {
            h->raw_cap = (int64_t)(h->raw_vm.mapped / row);
            return GAMMA_HIP_OK;
        }
        if (h->raw_vm.mapped > 0 || h->nraw > 0) return vm_fail(h, "raw store", e, what);
        // nothing is mapped yet: the range is given back and the store reallocates from here on
        h->raw_vm.release();
        h->raw_vmm = false;
        h->d_raw = nullptr;
    }
    h->raw_regrows++;
    int64_t ncap = std::max<int64_t>(need, h->raw_cap + h->raw_cap / 2);
    ncap = std::max<int64_t>(ncap, 1024);
    float* np = nullptr;
    if (h->wl) GH_CHECK(h, h->wl->exclusive());   // the old store is freed below
    GH_CHECK(h, hipMalloc((void**)&np, (size_t)ncap * h->raw_d * sizeof(float)));
    if (h->nraw > 0)
        GH_CHECK(h, hipMemcpyAsync(np, h->d_raw, (size_t)h->nraw * h->raw_d * sizeof(float),
                                   hipMemcpyDeviceToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    if (h->d_raw) GH_CHECK(h, hipFree(h->d_raw));
    h->d_raw = np;
    h->raw_cap = ncap;
    return GAMMA_HIP_OK;
}

int gamma_hip_raw_append(gamma_hip_index* h, int64_t n, const float* vecs) {
    if (!h || n < 0 || (n > 0 && !vecs)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (h->raw_sparse) return fail(h, GAMMA_HIP_EUNSUPPORTED, "the raw store holds this shard's rows only (gamma_hip_raw_put)");
    if (n == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(raw_reserve(h, h->nraw + n));
    GH_CHECK(h, hipMemcpyAsync(h->d_raw + h->nraw * h->raw_d, vecs, (size_t)n * h->raw_d * sizeof(float),
                               hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->nraw += n;
    return GAMMA_HIP_OK;
}

int gamma_hip_raw_put(gamma_hip_index* h, int64_t n, const int64_t* vids, const float* vecs) {
    if (!h || n < 0 || (n > 0 && (!vids || !vecs))) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (!h->raw_sparse && h->nraw > 0) return fail(h, GAMMA_HIP_EINVAL, "raw_put on a store that holds rows by vector id");
    if (n == 0) return GAMMA_HIP_OK;
    int64_t lo = INT64_MAX, hi = -1;
    for (int64_t i = 0; i < n; i++) {
        if (vids[i] < 0 || vids[i] >= ((int64_t)1 << 31)) return fail(h, GAMMA_HIP_EINVAL, "raw_put: vector id out of range");
        lo = std::min(lo, vids[i]);
        hi = std::max(hi, vids[i]);
    }
    h->raw_sparse = true;
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(raw_reserve(h, h->nraw + n));
    GH_CHECK(h, hipMemcpyAsync(h->d_raw + h->nraw * h->raw_d, vecs, (size_t)n * h->raw_d * sizeof(float),
                               hipMemcpyHostToDevice, h->wstream));
    if ((int64_t)h->h_raw_slot.size() <= hi) h->h_raw_slot.resize((size_t)hi + 1, -1);
    for (int64_t i = 0; i < n; i++) h->h_raw_slot[vids[i]] = (int32_t)(h->nraw + i);
    if (hi >= h->raw_slot_cap) {   // the device map grows: a new array, the whole mirror (readers are drained first)
        const int64_t ncap = std::max<int64_t>(hi + 1 + (hi + 1) / 2, 1 << 16);
        int32_t* np = nullptr;
        if (h->wl) GH_CHECK(h, h->wl->exclusive());
        GH_CHECK(h, hipMalloc((void**)&np, (size_t)ncap * sizeof(int32_t)));
        GH_CHECK(h, hipMemsetAsync(np, 0xff, (size_t)ncap * sizeof(int32_t), h->wstream));
        GH_CHECK(h, hipMemcpyAsync(np, h->h_raw_slot.data(), h->h_raw_slot.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        if (h->d_raw_slot) GH_CHECK(h, hipFree(h->d_raw_slot));
        h->d_raw_slot = np;
        h->raw_slot_cap = ncap;
    } else {
        GH_CHECK(h, hipMemcpyAsync(h->d_raw_slot + lo, h->h_raw_slot.data() + lo, (size_t)(hi - lo + 1) * sizeof(int32_t),
                                   hipMemcpyHostToDevice, h->wstream));
    }
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->nraw += n;
    return GAMMA_HIP_OK;
}

int gamma_hip_raw_write(gamma_hip_index* h, int64_t first_vid, int64_t n, const float* vecs) {
    if (!h || n < 0 || first_vid < 0 || (n > 0 && !vecs)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->raw_sparse) return fail(h, GAMMA_HIP_EUNSUPPORTED, "the raw store holds this shard's rows only (gamma_hip_raw_put)");
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    if (first_vid > h->nraw) return fail(h, GAMMA_HIP_EINVAL, "raw write would leave a gap");
    if (n == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(raw_reserve(h, first_vid + n));
    GH_CHECK(h, hipMemcpyAsync(h->d_raw + first_vid * h->raw_d, vecs, (size_t)n * h->raw_d * sizeof(float),
                               hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->nraw = std::max(h->nraw, first_vid + n);
    return GAMMA_HIP_OK;
}

int gamma_hip_raw_update(gamma_hip_index* h, int64_t vid, const float* vec) {
    if (!h || !vec) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->raw_sparse) return fail(h, GAMMA_HIP_EUNSUPPORTED, "the raw store holds this shard's rows only (gamma_hip_raw_put)");
    if (vid < 0 || vid >= h->nraw) return fail(h, GAMMA_HIP_EINVAL, "vid out of range");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipMemcpyAsync(h->d_raw + vid * h->raw_d, vec, (size_t)h->raw_d * sizeof(float),
                               hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

// n rows rewritten (vids below the row count; others are skipped: the mirror has not reached them yet), one wait
int gamma_hip_raw_update_batch(gamma_hip_index* h, int64_t n, const int64_t* vids, const float* vecs) {
    if (!h || n < 0 || (n > 0 && (!vids || !vecs))) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    WriteLock lk(h);
    if (h->raw_sparse) return fail(h, GAMMA_HIP_EUNSUPPORTED, "the raw store holds this shard's rows only (gamma_hip_raw_put)");
    GH_CHECK(h, hipSetDevice(h->device));
    for (int64_t i = 0; i < n; i++) {
        if (vids[i] < 0 || vids[i] >= h->nraw) continue;
        GH_CHECK(h, hipMemcpyAsync(h->d_raw + vids[i] * h->raw_d, vecs + i * h->raw_d, (size_t)h->raw_d * sizeof(float),
                                   hipMemcpyHostToDevice, h->wstream));
    }
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

// VectorReader::Gets (vector/raw_vector.cc:99-109 -> MemoryRawVector::GetVector, vector/memory_raw_vector.cc:136-142): rows
// by vector id, read back from the device store (what compute_dis hands to the exact distance); one wait for n rows
int gamma_hip_raw_gets(gamma_hip_index* h, int64_t n, const int64_t* vids, float* out) {
    if (!h || n < 0 || (n > 0 && (!vids || !out))) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    WriteLock lk(h);   // no row is read half written
    if (h->raw_sparse) return fail(h, GAMMA_HIP_EUNSUPPORTED, "the raw store holds this shard's rows only (gamma_hip_raw_put)");
    if (h->raw_d <= 0) return fail(h, GAMMA_HIP_EINVAL, "raw store not initialised");
    for (int64_t i = 0; i < n; i++)
        if (vids[i] < 0 || vids[i] >= h->nraw) return fail(h, GAMMA_HIP_EINVAL, "vid out of range");
    GH_CHECK(h, hipSetDevice(h->device));
    for (int64_t i = 0; i < n; i++)
        GH_CHECK(h, hipMemcpyAsync(out + i * h->raw_d, h->d_raw + vids[i] * h->raw_d, (size_t)h->raw_d * sizeof(float),
                                   hipMemcpyDeviceToHost, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

int64_t gamma_hip_raw_count(gamma_hip_index* h) { return h ? h->nraw : -1; }

int gamma_hip_raw_stats(gamma_hip_index* h, int64_t* out4) {
    if (!h || !out4) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    out4[0] = h->nraw;
    out4[1] = h->raw_cap;
    out4[2] = h->raw_regrows;
    out4[3] = h->raw_vmm ? 1 : 0;
    return GAMMA_HIP_OK;
}

/* ---- delete bitmap ------------------------------------------------------------------- */
static int bitmap_reserve(H* h, int64_t nbits) {
    size_t bytes = (((size_t)nbits >> 3) + 1 + 3) & ~(size_t)3;
    if (bytes <= h->bitmap_cap_bytes) return GAMMA_HIP_OK;
    size_t ncap = std::max(bytes, h->bitmap_cap_bytes * 2);
    uint8_t* np = nullptr;
    if (h->wl) GH_CHECK(h, h->wl->exclusive());   // the old bitmap is freed below
    GH_CHECK(h, hipMalloc((void**)&np, ncap));
    GH_CHECK(h, hipMemsetAsync(np, 0, ncap, h->wstream));
    if (h->d_bitmap)
        GH_CHECK(h, hipMemcpyAsync(np, h->d_bitmap, h->bitmap_cap_bytes, hipMemcpyDeviceToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    if (h->d_bitmap) GH_CHECK(h, hipFree(h->d_bitmap));
    h->d_bitmap = np;
    h->bitmap_cap_bytes = ncap;
    h->h_bitmap.resize(ncap, 0);
    return GAMMA_HIP_OK;
}

int gamma_hip_bitmap_upload(gamma_hip_index* h, const uint8_t* bitmap, int64_t nbits) {
    if (!h || nbits < 0 || (nbits > 0 && !bitmap)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    GH_TRY(bitmap_reserve(h, nbits));
    size_t bytes = ((size_t)nbits >> 3) + 1;  // bitmap::create, util/bitmap.cc:15-23
    GH_CHECK(h, hipMemsetAsync(h->d_bitmap, 0, h->bitmap_cap_bytes, h->wstream));
    GH_CHECK(h, hipMemcpyAsync(h->d_bitmap, bitmap, bytes, hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    std::fill(h->h_bitmap.begin(), h->h_bitmap.end(), 0);
    memcpy(h->h_bitmap.data(), bitmap, bytes);
    h->bitmap_bits = nbits;
    h->bitmap_any = false;
    for (size_t i = 0; i < bytes && !h->bitmap_any; i++) h->bitmap_any = bitmap[i] != 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_bitmap_set(gamma_hip_index* h, const int64_t* docids, int64_t n, int value) {
    if (!h || n < 0 || (n > 0 && !docids)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (n == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    int64_t mx = 0;
    for (int64_t i = 0; i < n; i++) mx = std::max(mx, docids[i]);
    if (mx >= h->bitmap_bits) {
        GH_TRY(bitmap_reserve(h, mx + 1));
        h->bitmap_bits = std::max<int64_t>(h->bitmap_bits, mx + 1);
    }
    GH_CHECK(h, h->we_stage.ensure((size_t)n * sizeof(int64_t)));
    GH_CHECK(h, hipMemcpyAsync(h->we_stage.p, docids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
    gh::launch_bitmap_set(h->wstream, h->d_bitmap, h->we_stage.as<int64_t>(), n, h->bitmap_bits, value);
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    for (int64_t i = 0; i < n; i++) {
        int64_t id = docids[i];
        if (id < 0) continue;
        if (value) h->bitmap_any = true;
        if (value) h->h_bitmap[id >> 3] |= (uint8_t)(1u << (id & 7));
        else h->h_bitmap[id >> 3] &= (uint8_t)~(1u << (id & 7));
    }
    return GAMMA_HIP_OK;
}

/* ---- IVFPQ / IVFFLAT models ----------------------------------------------------------- */
static int ivf_init_locked(gamma_hip_index* h, int d, int nlist, int M, int metric, int bucket_init_size,
                           int bucket_max_size, bool flat);
int gamma_hip_ivfpq_init(gamma_hip_index* h, int d, int nlist, int M, int nbits, int metric,
                         int bucket_init_size, int bucket_max_size) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "already initialised");
    if (d <= 0 || nlist <= 0 || M <= 0) return fail(h, GAMMA_HIP_EINVAL, "bad d/nlist/M");
    if (nbits != 8) return fail(h, GAMMA_HIP_EINVAL, "only nbits_per_idx == 8 is supported on device");
    if (d % M != 0) return fail(h, GAMMA_HIP_EINVAL, "d must be divisible by nsubvector");
    if (d / M > 64) return fail(h, GAMMA_HIP_EINVAL, "dsub > 64 unsupported");
    if (M > 64) return fail(h, GAMMA_HIP_EINVAL, "nsubvector > 64 unsupported (LUT must fit 64 KiB LDS)");
    return ivf_init_locked(h, d, nlist, M, metric, bucket_init_size, bucket_max_size, false);
}

// faiss::precomputed_table_max_bytes (faiss:IndexIVFPQ.cpp:379): process-wide like the library's extern
static std::atomic<int64_t> g_table_max_bytes{(int64_t)1 << 31};
int gamma_hip_set_precomputed_table_max_bytes(int64_t bytes) {
    if (bytes < 0) return GAMMA_HIP_EINVAL;
    g_table_max_bytes.store(bytes);
    return GAMMA_HIP_OK;
}
int64_t gamma_hip_get_precomputed_table_max_bytes(void) { return g_table_max_bytes.load(); }
int gamma_hip_ivfpq_use_precomputed_table(gamma_hip_index* h) {
    if (!h || !h->ivf_init || h->ivfflat) return GAMMA_HIP_EINVAL;
    return h->table_mode;
}

static int ivf_init_locked(gamma_hip_index* h, int d, int nlist, int M, int metric, int bucket_init_size,
                           int bucket_max_size, bool flat) {
    if (metric != GAMMA_HIP_METRIC_IP && metric != GAMMA_HIP_METRIC_L2) return fail(h, GAMMA_HIP_EINVAL, "bad metric");
    GH_CHECK(h, hipSetDevice(h->device));
    h->ivfflat = flat;
    // initialize_IVFPQ_precomputed_table's rule (faiss:IndexIVFPQ.cpp:441-449): `table_size > max` keeps mode 0.  faiss
    // applies it at train / Load; the outcome depends on nlist, M and the process-wide limit only, so it is fixed here,
    // where the arena decides whether it keeps the per-code table sums (they are sums of T2 entries: none in mode 0).
    h->table_mode = (!flat && (int64_t)nlist * M * 256 * (int64_t)sizeof(float) > g_table_max_bytes.load()) ? 0 : 1;
    h->keep_sums = !flat && h->table_mode == 1 && getenv("GAMMA_HIP_NO_CODE_SUMS") == nullptr;
    h->d = d;
    h->nlist = nlist;
    h->M = M;
    h->dsub = d / M;
    h->code_size = M;   // IVFFLAT: M = 1, one dummy byte per entry (the arena code keeps its shape)
    h->metric = metric;
    h->bucket_init = bucket_init_size > 0 ? bucket_init_size : 1000;
    h->bucket_max = bucket_max_size > 0 ? bucket_max_size : 1280000;
    GH_CHECK(h, hipMalloc((void**)&h->d_cc, (size_t)nlist * d * sizeof(float)));
    GH_CHECK(h, hipMalloc((void**)&h->d_cc_norms, (size_t)nlist * sizeof(float)));
    GH_CHECK(h, hipMalloc((void**)&h->d_pqc, flat ? 256 : (size_t)M * 256 * h->dsub * sizeof(float)));
    if (flat || h->table_mode == 1) GH_CHECK(h, hipMalloc((void**)&h->d_T2, flat ? 256 : (size_t)nlist * M * 256 * sizeof(float)));
    for (int v = 0; v < H::NVER; v++) {
        GH_CHECK(h, hipMalloc((void**)&h->d_ver_off[v], (size_t)nlist * sizeof(int64_t)));
        GH_CHECK(h, hipMalloc((void**)&h->d_ver_len[v], (size_t)nlist * sizeof(int)));
        GH_CHECK(h, hipHostMalloc(&h->pin_ver[v], (size_t)nlist * (sizeof(int64_t) + sizeof(int)), hipHostMallocDefault));
    }
    h->d_list_off = h->d_ver_off[0];
    h->d_list_len = h->d_ver_len[0];
    // RTInvertBucketData::Init (realtime_mem_data.cc:57-96): bucket_init entries per list
    h->h_list_off.resize(nlist);
    h->h_list_len.assign(nlist, 0);
    h->h_list_cap.assign(nlist, h->bucket_init);
    h->h_deleted.assign(nlist, 0);
    h->h_extend_time.assign(nlist, 0);
    for (int l = 0; l < nlist; l++) h->h_list_off[l] = (int64_t)l * h->bucket_init;
    h->arena_used = 0;
    GH_TRY(arena_reserve(h, (int64_t)nlist * h->bucket_init));
    h->arena_used = (int64_t)nlist * h->bucket_init;
    GH_TRY(publish_meta(h));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->vid_pos.assign((size_t)nlist * h->bucket_init, -1);
    h->ivf_init = true;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfflat_init(gamma_hip_index* h, int d, int nlist, int metric, int bucket_init_size, int bucket_max_size) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "already initialised");
    if (d <= 0 || nlist <= 0) return fail(h, GAMMA_HIP_EINVAL, "bad d/nlist");
    return ivf_init_locked(h, d, nlist, 1, metric, bucket_init_size, bucket_max_size, true);
}

int gamma_hip_ivfflat_set_trained(gamma_hip_index* h, const float* cc) {
    if (!h || !cc) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init || !h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfflat not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipMemcpyAsync(h->d_cc, cc, (size_t)h->nlist * h->d * sizeof(float), hipMemcpyHostToDevice, h->wstream));
    gh::launch_row_norms(h->wstream, h->d_cc, h->nlist, h->d, h->d_cc_norms);
    GH_CHECK(h, hipGetLastError());
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->trained = true;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_set_trained(gamma_hip_index* h, const float* cc, const float* pqc, const float* table) {
    if (!h || !cc || !pqc) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init || h->ivfflat) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    const size_t ncc = (size_t)h->nlist * h->d, npq = (size_t)h->M * 256 * h->dsub;
    const size_t nt = (size_t)h->nlist * h->M * 256;
    GH_CHECK(h, hipMemcpyAsync(h->d_cc, cc, ncc * sizeof(float), hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipMemcpyAsync(h->d_pqc, pqc, npq * sizeof(float), hipMemcpyHostToDevice, h->wstream));
    gh::launch_row_norms(h->wstream, h->d_cc, h->nlist, h->d, h->d_cc_norms);
    if (h->table_mode == 0) {
        // no table (a supplied one is not taken either: the reference would not have built it)
    } else if (table)
        GH_CHECK(h, hipMemcpyAsync(h->d_T2, table, nt * sizeof(float), hipMemcpyHostToDevice, h->wstream));
    else
        gh::launch_precompute_table(h->wstream, h->d_cc, h->nlist, h->d, h->M, h->d_pqc, h->d_T2);
    {
        std::vector<int> rank = centroid_rank(cc, h->nlist, h->d);
        if (!h->d_list_rank) GH_CHECK(h, hipMalloc((void**)&h->d_list_rank, (size_t)h->nlist * sizeof(int)));
        GH_CHECK(h, hipMemcpyAsync(h->d_list_rank, rank.data(), (size_t)h->nlist * sizeof(int),
                                   hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));   // rank is a local
    }
    if (h->keep_sums) {
        if (!h->d_t2max) GH_CHECK(h, hipMalloc((void**)&h->d_t2max, (size_t)h->nlist * sizeof(float)));
        gh::launch_t2_rowmax(h->wstream, h->d_T2, h->nlist, h->M, h->d_t2max);
    }
    GH_CHECK(h, hipGetLastError());
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->t2max_all = 0.f;
    if (h->keep_sums && h->d_t2max) {   // the largest of the per-list bounds: the filter pass's margin without a look-up per list
        std::vector<float> tm((size_t)h->nlist);
        GH_CHECK(h, hipMemcpy(tm.data(), h->d_t2max, tm.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (float v : tm) h->t2max_all = std::max(h->t2max_all, v);
    }
    h->trained = true;
    if (h->ntotal > 0) {   // a new table under existing lists: their sums follow it
        GH_TRY(sums_all_lists(h));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_get_precomputed_table(gamma_hip_index* h, float* out) {
    if (!h || !out) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "not trained");
    if (h->table_mode == 0) return fail(h, GAMMA_HIP_EUNSUPPORTED, "table mode 0: no precomputed table (it would exceed precomputed_table_max_bytes)");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, hipMemcpyAsync(out, h->d_T2, (size_t)h->nlist * h->M * 256 * sizeof(float),
                               hipMemcpyDeviceToHost, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

/* ---- realtime lists ------------------------------------------------------------------- */
int gamma_hip_ivfpq_add_keys(gamma_hip_index* h, int list_no, int n, const int64_t* vids, const uint8_t* codes) {
    if (!h || (n > 0 && (!vids || !codes))) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    return add_keys_locked(h, list_no, n, vids, codes);
}

int gamma_hip_ivfpq_add_keys_batch(gamma_hip_index* h, int nlists, const int32_t* list_nos,
                                   const int32_t* counts, const int64_t* vids, const uint8_t* codes) {
    if (!h || nlists < 0 || (nlists > 0 && (!list_nos || !counts || !vids || !codes))) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    int64_t off = 0;
    // grow first so that no copy below races with an arena move; a list named twice reserves for the sum
    {
        std::map<int, int64_t> per_list;
        for (int i = 0; i < nlists; i++) {
            if (list_nos[i] < 0 || list_nos[i] >= h->nlist || counts[i] < 0) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
            per_list[list_nos[i]] += counts[i];
        }
        for (auto& kv : per_list) {
            if (kv.second > h->bucket_max) return fail(h, GAMMA_HIP_EFULL, "exceed the max bucket keys");
            GH_TRY(list_ensure(h, kv.first, (int)kv.second));
        }
    }
    std::vector<int> r_list, r_n;      // the appended ranges, for the code sums (alive until the wait below)
    std::vector<int64_t> r_pos;
    int r_max = 0;
    for (int i = 0; i < nlists; i++) {
        const int l = list_nos[i], n = counts[i];
        if (n == 0) continue;
        const int64_t pos = h->h_list_off[l] + h->h_list_len[l];
        r_list.push_back(l);
        r_n.push_back(n);
        r_pos.push_back(pos);
        r_max = std::max(r_max, n);
        GH_CHECK(h, hipMemcpyAsync(h->d_ids + pos, vids + off, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(h->d_codes + pos * h->code_size, codes + off * h->code_size,
                                   (size_t)n * h->code_size, hipMemcpyHostToDevice, h->wstream));
        for (int j = 0; j < n; j++) {
            const int64_t v = vids[off + j];
            if (v < 0) {   // superseded slot restored from a dump: same accounting as add_keys_locked, so that
                h->h_deleted[l]++;   // the scan reads the ids (n_moved) and never returns the slot
                h->n_moved++;
                continue;
            }
            if ((size_t)v >= h->vid_pos.size()) h->vid_pos.resize(std::max<size_t>(h->vid_pos.size() * 2, v + 1), -1);
            h->vid_pos[v] = ((int64_t)l << 32) | (int64_t)(h->h_list_len[l] + j);
            if (h->doc_deleted(v)) h->h_deleted[l]++;
        }
        h->h_list_len[l] += n;
        h->ntotal += n;
        if (h->h_list_len[l] > h->max_list_len) h->max_list_len = h->h_list_len[l];
        off += n;
    }
    if (h->d_sums && h->trained && !r_list.empty()) {
        const size_t nr = r_list.size(), o_n = nr * sizeof(int), o_pos = (2 * nr * sizeof(int) + 7) & ~(size_t)7;
        GH_CHECK(h, h->we_stage.ensure(o_pos + nr * sizeof(int64_t)));
        char* b = h->we_stage.as<char>();
        GH_CHECK(h, hipMemcpyAsync(b, r_list.data(), nr * sizeof(int), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(b + o_n, r_n.data(), nr * sizeof(int), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(b + o_pos, r_pos.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
        gh::launch_code_sums_ranges(h->wstream, h->d_T2, h->d_codes, h->code_size, reinterpret_cast<int*>(b),
                                    reinterpret_cast<int64_t*>(b + o_pos), reinterpret_cast<int*>(b + o_n), (int)nr, r_max,
                                    h->d_sums);
    }
    // publish the new lengths (and moved extents) after the copies, in stream order
    GH_TRY(publish_meta(h));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return arena_repack_if_need(h);
}

int gamma_hip_ivfpq_update(gamma_hip_index* h, int list_no, int64_t vid, const uint8_t* code) {
    if (!h || !code) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init || list_no < 0 || list_no >= h->nlist) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
    if (vid < 0 || (size_t)vid >= h->vid_pos.size()) return GAMMA_HIP_OK;  // realtime_mem_data.cc:307
    const int64_t bp = h->vid_pos[vid];
    if (bp == -1) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const int ob = (int)(bp >> 32), op = (int)(bp & 0xffffffff);
    if (ob == list_no) {
        GH_CHECK(h, hipMemcpyAsync(h->d_codes + (h->h_list_off[ob] + op) * h->code_size, code, h->code_size,
                                   hipMemcpyHostToDevice, h->wstream));
        GH_TRY(sums_range(h, ob, h->h_list_off[ob] + op, 1));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        return GAMMA_HIP_OK;
    }
    GH_TRY(list_ensure(h, list_no, 1));   // before the old entry is given up: a full list must not lose the vector
    gh::launch_mark_moved(h->wstream, h->d_ids, h->h_list_off[ob] + op);
    h->h_deleted[ob]++;
    h->n_moved++;
    h->ntotal -= 1;  // add_keys_locked re-counts it
    return add_keys_locked(h, list_no, 1, &vid, code);
}

// n list updates with the codes already computed: what RTInvertIndex::Update (realtime_mem_data.cc:305-327) does for
// each entry in order, with ONE publish of the lists' tables and ONE wait for the device at the end.
//   ops[i] (nullptr = all 0)   0: Update -- a vid this handle does not hold is ignored (:307-311)
//                              1: the vid is held by ANOTHER shard of a list-sharded index and joins list_nos[i] here: AddKeys
//                              2: the vid leaves this shard: the first half of Update alone (gamma_hip_ivfpq_remove)
int gamma_hip_ivfpq_apply_updates(gamma_hip_index* h, int n, const int32_t* list_nos, const int64_t* vids,
                                  const uint8_t* codes, const uint8_t* ops) {
    if (!h || n < 0 || (n > 0 && (!list_nos || !vids || !codes))) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    const int cs = h->code_size;
    // room first: an upper bound of the entries every list may receive (an entry that stays in its list needs none,
    // but a vid named twice may have moved in between -- counted as a move), so that no extent moves while the copies
    // below are in flight
    {
        std::map<int, int64_t> per_list;
        for (int i = 0; i < n; i++) {
            const int op = ops ? ops[i] : 0;
            if (op == 2) continue;
            if (list_nos[i] < 0 || list_nos[i] >= h->nlist) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
            if (op == 0) {
                const int64_t v = vids[i];
                if (v < 0 || (size_t)v >= h->vid_pos.size() || h->vid_pos[v] == -1) continue;
            }
            per_list[list_nos[i]]++;
        }
        for (auto& kv : per_list) {
            if (h->h_list_len[kv.first] + kv.second > h->bucket_max) return fail(h, GAMMA_HIP_EFULL, "exceed the max bucket keys");
            GH_TRY(list_ensure(h, kv.first, (int)kv.second));
        }
    }
    bool changed = false;
    for (int i = 0; i < n; i++) {
        const int op = ops ? ops[i] : 0;
        const int64_t vid = vids[i];
        const int l = list_nos[i];
        const uint8_t* code = codes + (size_t)i * cs;
        int64_t bp = -1;
        if (vid >= 0 && (size_t)vid < h->vid_pos.size()) bp = h->vid_pos[vid];
        if (op != 1) {
            if (bp == -1) continue;   // not held here
            const int ob = (int)(bp >> 32), opos = (int)(bp & 0xffffffff);
            if (op == 0 && ob == l) {   // same list: the code is rewritten in place
                GH_CHECK(h, hipMemcpyAsync(h->d_codes + (h->h_list_off[ob] + opos) * cs, code, cs, hipMemcpyHostToDevice, h->wstream));
                GH_TRY(sums_range(h, ob, h->h_list_off[ob] + opos, 1));
                continue;
            }
            gh::launch_mark_moved(h->wstream, h->d_ids, h->h_list_off[ob] + opos);
            h->h_deleted[ob]++;
            h->n_moved++;
            h->ntotal -= 1;
            h->vid_pos[vid] = -1;
            if (op == 2) continue;
        }
        // AddKeys of one entry (add_keys_locked without its publish)
        const int64_t pos = h->h_list_off[l] + h->h_list_len[l];
        GH_CHECK(h, hipMemcpyAsync(h->d_ids + pos, vids + i, sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(h->d_codes + pos * cs, code, cs, hipMemcpyHostToDevice, h->wstream));
        GH_TRY(sums_range(h, l, pos, 1));
        if ((size_t)vid >= h->vid_pos.size()) h->vid_pos.resize(std::max<size_t>(h->vid_pos.size() * 2, vid + 1), -1);
        h->vid_pos[vid] = ((int64_t)l << 32) | (int64_t)h->h_list_len[l];
        if (h->doc_deleted(vid)) h->h_deleted[l]++;
        h->h_list_len[l] += 1;
        h->ntotal += 1;
        if (h->h_list_len[l] > h->max_list_len) h->max_list_len = h->h_list_len[l];
        changed = true;
    }
    if (changed) GH_TRY(publish_meta(h));
    GH_CHECK(h, hipGetLastError());
    GH_CHECK(h, hipStreamSynchronize(h->wstream));   // the caller's buffers are free again
    return changed ? arena_repack_if_need(h) : GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_has_vid(gamma_hip_index* h, const int64_t* vids, int n, uint8_t* out) {
    if (!h || n < 0 || (n > 0 && (!vids || !out))) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    for (int i = 0; i < n; i++)
        out[i] = vids[i] >= 0 && (size_t)vids[i] < h->vid_pos.size() && h->vid_pos[vids[i]] != -1;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_remove(gamma_hip_index* h, int64_t vid) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    if (vid < 0 || (size_t)vid >= h->vid_pos.size()) return GAMMA_HIP_OK;
    const int64_t bp = h->vid_pos[vid];
    if (bp == -1) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    const int ob = (int)(bp >> 32), op = (int)(bp & 0xffffffff);
    gh::launch_mark_moved(h->wstream, h->d_ids, h->h_list_off[ob] + op);
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    h->h_deleted[ob]++;
    h->n_moved++;
    h->ntotal -= 1;
    h->vid_pos[vid] = -1;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_delete(gamma_hip_index* h, const int64_t* vids, int n) {
    if (!h || (n > 0 && !vids)) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    for (int i = 0; i < n; i++) {
        if (vids[i] < 0 || (size_t)vids[i] >= h->vid_pos.size()) continue;
        const int64_t bp = h->vid_pos[vids[i]];
        if (bp == -1) continue;
        h->h_deleted[bp >> 32]++;
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_compact_if_need(gamma_hip_index* h) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    std::vector<int64_t> ids;
    std::vector<uint8_t> codes;
    bool changed = false;
    for (int l = 0; l < h->nlist; l++) {
        const int len = h->h_list_len[l];
        if (!((float)h->h_deleted[l] / len >= 0.3f)) continue;  // Compactable, :373-377
        ids.resize(len);
        codes.resize((size_t)len * h->code_size);
        GH_CHECK(h, hipMemcpyAsync(ids.data(), h->d_ids + h->h_list_off[l], (size_t)len * sizeof(int64_t), hipMemcpyDeviceToHost, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(codes.data(), h->d_codes + h->h_list_off[l] * h->code_size, (size_t)len * h->code_size, hipMemcpyDeviceToHost, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        int pos = 0;
        for (int i = 0; i < len; i++) {  // CompactOne, :98-112
            const int64_t id = ids[i];
            const int64_t v = id & ~kDelMask;
            const bool deleted = h->doc_deleted(v);
            if (!(id & kDelMask) && !deleted) {
                ids[pos] = id;
                memmove(codes.data() + (size_t)pos * h->code_size, codes.data() + (size_t)i * h->code_size, h->code_size);
                h->vid_pos[id] = ((int64_t)l << 32) | pos;
                pos++;
            }
        }
        // new region of the same capacity (copy-on-write swap, :426-474)
        GH_TRY(arena_reserve(h, h->h_list_cap[l]));
        const int64_t noff = h->arena_used;
        h->arena_used += h->h_list_cap[l];
        h->arena_waste += h->h_list_cap[l];
        if (pos > 0) {
            GH_CHECK(h, hipMemcpyAsync(h->d_ids + noff, ids.data(), (size_t)pos * sizeof(int64_t), hipMemcpyHostToDevice, h->wstream));
            GH_CHECK(h, hipMemcpyAsync(h->d_codes + noff * h->code_size, codes.data(), (size_t)pos * h->code_size, hipMemcpyHostToDevice, h->wstream));
            GH_TRY(sums_range(h, l, noff, pos));
        }
        h->h_list_off[l] = noff;
        h->ntotal -= (len - pos);
        h->h_list_len[l] = pos;
        h->h_deleted[l] = 0;
        GH_CHECK(h, hipStreamSynchronize(h->wstream));   // ids / codes are locals of this loop
        changed = true;
    }
    if (changed) {
        GH_TRY(publish_meta(h));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        GH_TRY(arena_repack_if_need(h));
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_arena_stats(gamma_hip_index* h, int64_t* out4) {
    if (!h || !out4) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    out4[0] = h->arena_cap;
    out4[1] = h->arena_used;
    out4[2] = h->arena_waste;
    out4[3] = h->n_repacks;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_repack_verify_stats(gamma_hip_index* h, int64_t* out2) {
    if (!h || !out2) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    out2[0] = h->repack_verified;
    out2[1] = h->repack_verify_failures;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_arena_growth(gamma_hip_index* h, int64_t* out2) {
    if (!h || !out2) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    out2[0] = h->arena_regrows;
    out2[1] = h->arena_vmm ? 1 : 0;
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_set_repack_threshold(gamma_hip_index* h, int64_t min_waste_entries) {
    if (!h || min_waste_entries < 0) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    h->repack_min_entries = min_waste_entries;
    return arena_repack_if_need(h);
}

int64_t gamma_hip_ivfpq_list_size(gamma_hip_index* h, int l) {
    if (!h || !h->ivf_init || l < 0 || l >= h->nlist) return -1;
    return h->h_list_len[l];
}
int64_t gamma_hip_ivfpq_list_capacity(gamma_hip_index* h, int l) {
    if (!h || !h->ivf_init || l < 0 || l >= h->nlist) return -1;
    return h->h_list_cap[l];
}

int gamma_hip_ivfpq_get_list(gamma_hip_index* h, int l, int64_t* vids, uint8_t* codes) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init || l < 0 || l >= h->nlist) return fail(h, GAMMA_HIP_EINVAL, "bad list_no");
    const int len = h->h_list_len[l];
    if (len == 0) return GAMMA_HIP_OK;
    GH_CHECK(h, hipSetDevice(h->device));
    if (vids) GH_CHECK(h, hipMemcpyAsync(vids, h->d_ids + h->h_list_off[l], (size_t)len * sizeof(int64_t), hipMemcpyDeviceToHost, h->wstream));
    if (codes) GH_CHECK(h, hipMemcpyAsync(codes, h->d_codes + h->h_list_off[l] * h->code_size, (size_t)len * h->code_size, hipMemcpyDeviceToHost, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_set_list_mask(gamma_hip_index* h, const uint8_t* owned) {
    if (!h) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->ivf_init) return fail(h, GAMMA_HIP_EINVAL, "ivfpq not initialised");
    GH_CHECK(h, hipSetDevice(h->device));
    GH_CHECK(h, lk.exclusive());
    if (!owned) {
        if (h->d_list_mask) GH_CHECK(h, hipFree(h->d_list_mask));
        h->d_list_mask = nullptr;
        h->h_list_mask.clear();
        return GAMMA_HIP_OK;
    }
    if (!h->d_list_mask) GH_CHECK(h, hipMalloc((void**)&h->d_list_mask, (size_t)h->nlist));
    h->h_list_mask.assign(owned, owned + h->nlist);
    GH_CHECK(h, hipMemcpyAsync(h->d_list_mask, owned, (size_t)h->nlist, hipMemcpyHostToDevice, h->wstream));
    GH_CHECK(h, hipStreamSynchronize(h->wstream));
    return GAMMA_HIP_OK;
}

/* ---- device-side encode / add ----------------------------------------------------------- */
// exact: the arithmetic form faiss picks from the size of the WHOLE assign() call (n < 20), not of a chunk
static int encode_locked(H* h, int64_t n, const float* d_vecs, int* d_assign, uint8_t* d_codes_out, bool exact) {
    // quantizer->assign == search with k = 1 (faiss rule for the arithmetic form)
    hipStream_t s = h->wstream;   // own stream and own workspace: runs beside the searches
    const int d = h->d, nlist = h->nlist;
    GH_CHECK(h, h->we_mat.ensure((size_t)n * nlist * sizeof(float)));
    GH_CHECK(h, h->we_cdis.ensure((size_t)n * sizeof(float)));
    if (exact) {
        gh::launch_pairwise(s, true, d_vecs, (int)n, d, h->d_cc, nlist, h->we_mat.as<float>(), nlist);
    } else {
        gh::launch_l2_gemmform(s, d_vecs, (int)n, d, h->d_cc, nlist, nullptr, h->d_cc_norms,
                               h->we_mat.as<float>(), nlist, true);
    }
    gh::launch_select_topk(s, true, h->we_mat.as<float>(), nlist, nullptr, nlist, nlist, (int)n, 1,
                           h->we_cdis.as<float>(), d_assign);
    if (h->ivfflat) GH_CHECK(h, hipMemsetAsync(d_codes_out, 0, (size_t)n, s));   // the dummy byte of every entry
    else gh::launch_pq_encode(s, d_vecs, n, d, h->M, d_assign, h->d_cc, h->d_pqc, d_codes_out);
    GH_CHECK(h, hipGetLastError());
    return GAMMA_HIP_OK;
}

static int encode_host(gamma_hip_index* h, int64_t n, const float* vecs, int64_t* list_nos, uint8_t* codes, bool exact);

int gamma_hip_ivfpq_encode(gamma_hip_index* h, int64_t n, const float* vecs, int64_t* list_nos, uint8_t* codes) {
    return encode_host(h, n, vecs, list_nos, codes, n < 20);
}

// every vector assigned as a call of its own would assign it (quantizer->assign(1, ..): the exact form, what
// GammaIVFPQIndex::Update runs per vector, gamma_index_ivfpq.cc:398), in one device pass
int gamma_hip_ivfpq_encode_each(gamma_hip_index* h, int64_t n, const float* vecs, int64_t* list_nos, uint8_t* codes) {
    return encode_host(h, n, vecs, list_nos, codes, true);
}

static int encode_host(gamma_hip_index* h, int64_t n, const float* vecs, int64_t* list_nos, uint8_t* codes, bool exact) {
    if (!h || n < 0 || (n > 0 && (!vecs || !list_nos || !codes))) return GAMMA_HIP_EINVAL;
    WriteLock lk(h);
    if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "not trained");
    if (n == 0) return GAMMA_HIP_OK;
    if (!exact && blas_form_not_restated(n, h->nlist, h->d)) h->blas_unrestated++;
    GH_CHECK(h, hipSetDevice(h->device));
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n, 65536), (int64_t)(h->dist_budget_bytes / ((size_t)h->nlist * sizeof(float)))));
    std::vector<int> assign(chunk);
    for (int64_t i0 = 0; i0 < n; i0 += chunk) {
        const int64_t nc = std::min(chunk, n - i0);
        GH_CHECK(h, h->we_x.ensure((size_t)nc * h->d * sizeof(float)));
        GH_CHECK(h, h->we_assign.ensure((size_t)nc * sizeof(int)));
        GH_CHECK(h, h->we_codes.ensure((size_t)nc * h->code_size));
        GH_CHECK(h, hipMemcpyAsync(h->we_x.p, vecs + i0 * h->d, (size_t)nc * h->d * sizeof(float), hipMemcpyHostToDevice, h->wstream));
        GH_TRY(encode_locked(h, nc, h->we_x.as<float>(), h->we_assign.as<int>(), h->we_codes.as<uint8_t>(), exact));
        GH_CHECK(h, hipMemcpyAsync(assign.data(), h->we_assign.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, h->wstream));
        GH_CHECK(h, hipMemcpyAsync(codes + i0 * h->code_size, h->we_codes.p, (size_t)nc * h->code_size, hipMemcpyDeviceToHost, h->wstream));
        GH_CHECK(h, hipStreamSynchronize(h->wstream));
        for (int64_t i = 0; i < nc; i++) list_nos[i0 + i] = assign[i];
    }
    return GAMMA_HIP_OK;
}

// GammaIVFPQIndex::Update for a batch (gamma_index_ivfpq.cc:375-422; the engine drains up to 20 000 updated vids per
// pass, vector/vector_manager.cc:355-380): one encode, the list updates in order, one publish
int gamma_hip_ivfpq_update_batch(gamma_hip_index* h, int n, const int64_t* vids, const float* vecs) {
    if (!h || n < 0 || (n > 0 && (!vids || !vecs))) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    int cs = 0;
    {
        std::lock_guard<std::mutex> g(h->mu);
        if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "not trained");
        cs = h->code_size;
    }
    std::vector<int64_t> lno(n);
    std::vector<uint8_t> codes((size_t)n * cs);
    GH_TRY(encode_host(h, n, vecs, lno.data(), codes.data(), true));
    std::vector<int32_t> l32(n);
    for (int i = 0; i < n; i++) l32[i] = (int32_t)lno[i];
    return gamma_hip_ivfpq_apply_updates(h, n, l32.data(), vids, codes.data(), nullptr);
}

int gamma_hip_assign(gamma_hip_index* h, int d, int64_t n, const float* x, int k, const float* centroids,
                     int32_t* assign, float* dis) {
    if (!h || d <= 0 || n < 0 || k <= 0 || (n > 0 && (!x || !centroids || !assign))) return GAMMA_HIP_EINVAL;
    SearchLock lk(h);
    if (h->replay_pending) {
        GH_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_rdone, 0));
        h->replay_pending = false;
    }
    if (n == 0) return GAMMA_HIP_OK;
    if (blas_form_not_restated(n, k, d)) h->blas_unrestated++;
    GH_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // centroids + their norms live in the (otherwise unused here) partial-result buffers
    GH_CHECK(h, h->w_part_v.ensure((size_t)k * d * sizeof(float)));
    GH_CHECK(h, h->w_xn.ensure((size_t)k * sizeof(float)));
    GH_CHECK(h, hipMemcpyAsync(h->w_part_v.p, centroids, (size_t)k * d * sizeof(float), hipMemcpyHostToDevice, s));
    gh::launch_row_norms(s, h->w_part_v.as<float>(), k, d, h->w_xn.as<float>());
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(h->dist_budget_bytes / ((size_t)k * sizeof(float)))));
    for (int64_t i0 = 0; i0 < n; i0 += chunk) {
        const int64_t nc = std::min(chunk, n - i0);
        GH_CHECK(h, h->w_x.ensure((size_t)nc * d * sizeof(float)));
        GH_CHECK(h, h->w_mat.ensure((size_t)nc * k * sizeof(float)));
        GH_CHECK(h, h->w_assign.ensure((size_t)nc * sizeof(int)));
        GH_CHECK(h, h->w_coarse_dis.ensure((size_t)nc * sizeof(float)));
        GH_CHECK(h, hipMemcpyAsync(h->w_x.p, x + i0 * d, (size_t)nc * d * sizeof(float), hipMemcpyHostToDevice, s));
        gh::launch_l2_gemmform(s, h->w_x.as<float>(), (int)nc, d, h->w_part_v.as<float>(), k, nullptr,
                               h->w_xn.as<float>(), h->w_mat.as<float>(), k, true);
        gh::launch_select_topk(s, true, h->w_mat.as<float>(), k, nullptr, k, k, (int)nc, 1,
                               h->w_coarse_dis.as<float>(), h->w_assign.as<int>());
        GH_CHECK(h, hipGetLastError());
        GH_CHECK(h, hipMemcpyAsync(assign + i0, h->w_assign.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, s));
        if (dis) GH_CHECK(h, hipMemcpyAsync(dis + i0, h->w_coarse_dis.p, (size_t)nc * sizeof(float), hipMemcpyDeviceToHost, s));
        GH_CHECK(h, hipStreamSynchronize(s));
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_ivfpq_add(gamma_hip_index* h, int64_t n, const float* vecs, int64_t first_vid) {
    if (!h || n < 0 || (n > 0 && !vecs)) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    std::vector<int64_t> lno(n);
    std::vector<uint8_t> codes;
    {
        std::lock_guard<std::mutex> g(h->mu);
        if (!h->trained) return fail(h, GAMMA_HIP_ENOTTRAINED, "not trained");
        codes.resize((size_t)n * h->code_size);
    }
    GH_TRY(gamma_hip_ivfpq_encode(h, n, vecs, lno.data(), codes.data()));
    // group by list in ascending list order (std::map in gamma_index_ivfpq.cc:428-494)
    const int cs = h->code_size;
    std::vector<int64_t> order(n);
    for (int64_t i = 0; i < n; i++) {
        if (lno[i] < 0) lno[i] = (first_vid + i) % h->nlist;
        order[i] = i;
    }
    // list-sharded index: every shard is handed the same batch and keeps the vectors whose list it owns
    // (realtime inserts route to the owner of the assigned list) -- no exchange needed
    {
        std::lock_guard<std::mutex> g(h->mu);
        if (!h->h_list_mask.empty()) {
            int64_t m = 0;
            for (int64_t i = 0; i < n; i++)
                if (h->h_list_mask[lno[i]]) order[m++] = i;
            order.resize(m);
        }
    }
    const int64_t nkeep = (int64_t)order.size();
    if (nkeep == 0) return GAMMA_HIP_OK;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return lno[a] < lno[b]; });
    std::vector<int32_t> lists, counts;
    std::vector<int64_t> vids(nkeep);
    std::vector<uint8_t> gcodes((size_t)nkeep * cs);
    for (int64_t i = 0; i < nkeep; i++) {
        const int64_t src = order[i];
        vids[i] = first_vid + src;
        memcpy(gcodes.data() + (size_t)i * cs, codes.data() + (size_t)src * cs, cs);
        if (lists.empty() || lists.back() != (int32_t)lno[src]) {
            lists.push_back((int32_t)lno[src]);
            counts.push_back(0);
        }
        counts.back()++;
    }
    return gamma_hip_ivfpq_add_keys_batch(h, (int)lists.size(), lists.data(), counts.data(), vids.data(), gcodes.data());
}

}  // extern "C"
