// gamma_hip_group.cpp -- several GPUs behind ONE index object in one process: a group of handles, the index sharded by
// IVF list (include/gamma_hip.h, "several GPUs in ONE process").  The reference's GPU model does this with faiss's
// IndexShards (index/impl/gpu/gamma_gpu_cloner.cpp:200-269, index/impl/gpu/gamma_index_ivfpq_gpu.cc:356-436,
// faiss:IndexShards.cpp:283-345: a host thread per GPU, results merged on the host); here the split is by list (SURVEY
// 8e), the exchange is device-to-device and the merge runs on the GPU that owns the query.
//
// Built on the public C ABI of the members only (plus the HIP runtime for the peer copies and events): a member is an
// ordinary handle with a list mask.  One host thread per member enqueues that member's share of a call on the member's
// own stream; the threads meet at two barriers (assignment exchanged, candidates exchanged) and cross-device ordering
// is by events.  gamma_amd/dist.py is the same sequence of steps with one PROCESS per GPU and RCCL collectives.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/gamma_hip.h"

namespace {

struct GBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        else p = nullptr;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// all members' threads meet here; reusable
struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    int n = 0, waiting = 0;
    uint64_t gen = 0;
    bool acc_ok = true, phase_ok[2] = {true, true};
    // Every member passes its own status; all of them get the SAME answer for the phase: whether every member was fine
    // when it arrived.  The go / no-go decisions of a multi-phase call are taken from this snapshot only -- a member that
    // fails after the barrier cannot make the others disagree about which barriers are still to come.
    bool arrive(bool my_ok = true) {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = gen;
        acc_ok = acc_ok && my_ok;
        if (++waiting == n) {
            waiting = 0;
            phase_ok[g & 1] = acc_ok;   // (a member can be at most one generation ahead of the slowest reader)
            acc_ok = true;
            gen++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
        return phase_ok[g & 1];
    }
};

}  // namespace

// ---- optional transport: RCCL over xGMI (the north star's wording: an all-gather of the assignment, the per-shard
//      top-recall_num tables exchanged before the final merge).  Resolved at run time (dlopen: no link-time dependency on
//      the library, whose copy inside the process may be the one a host framework already loaded); one communicator per
//      member from ncclCommInitAll, every member's thread issues its own calls on its own stream.  Off by default
//      (GAMMA_HIP_GROUP_RCCL=1 or gamma_hip_group_set_transport): peer copies and the barriers below do the same job, and a
//      group whose members share a device (tests) cannot form a communicator. ----
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && AllGather && AllReduce && Send && Recv && GroupStart && GroupEnd; }
};
constexpr int kNcclChar = 0;   // ncclInt8: the exchanges are counted in bytes
constexpr int kNcclFloat = 7, kNcclMax = 2, kNcclMin = 3;   // ncclFloat32, ncclMax, ncclMin (nccl.h)
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* l = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // a copy already in the process first
        if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!l) l = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!l) return;
        api.lib = l;
        api.CommInitAll = reinterpret_cast<int (*)(void**, int, const int*)>(dlsym(l, "ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(l, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(l, "ncclAllGather"));
        api.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(l, "ncclAllReduce"));
        api.Send = reinterpret_cast<int (*)(const void*, size_t, int, int, void*, hipStream_t)>(dlsym(l, "ncclSend"));
        api.Recv = reinterpret_cast<int (*)(void*, size_t, int, int, void*, hipStream_t)>(dlsym(l, "ncclRecv"));
        api.GroupStart = reinterpret_cast<int (*)()>(dlsym(l, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<int (*)()>(dlsym(l, "ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(l, "ncclGetErrorString"));
    });
    return &api;
}
}  // namespace

struct gamma_hip_group {
    std::vector<gamma_hip_index*> m;
    std::vector<int> dev;
    std::vector<int> owner;   // list -> member; empty until gamma_hip_group_set_owners
    std::string err;
    std::mutex mu;            // one group-level call at a time
    int next_enc = 0;         // members take turns encoding Add / Update batches
    bool replicate = false;   // gamma_hip_group_set_placement: every member holds every list, queries are split
    // 0: peer copies; 1 (default since round 6; GAMMA_HIP_GROUP_RCCL=0 turns it off): RCCL wherever a communicator of the
    // members' DISTINCT devices can be formed -- members sharing a device, a missing librccl.so or a failing
    // ncclCommInitAll fall back to peer copies and say so (gamma_hip_group_transport)
    int transport = (getenv("GAMMA_HIP_GROUP_RCCL") && atoi(getenv("GAMMA_HIP_GROUP_RCCL")) == 0) ? 0 : 1;
    std::vector<void*> comm;  // one per member once formed
    bool comm_tried = false;
    std::string transport_note;
    int64_t rccl_calls = 0;   // searches whose exchanges went through RCCL

    struct Member {
        GBuf x, cdis, probe, rdis, rids, all_dis, all_ids, D, I;
        // exact ties across members: flagged queries' inputs (owner side / shard side), exports, the owner's copy of all exports
        GBuf fx, fcd, fpr, sx, scd, spr, ex_vals, ex_ids, ex_off, av, ai, ao, cutf, cutall;
        // two-phase shard search: the bounds this member's scan exports (what its peers read), its working copy, a peer's
        GBuf bound, bound_pub, bound_peer;
        hipEvent_t ev_coarse = nullptr, ev_scan = nullptr, ev_tie = nullptr, ev_bound = nullptr;
    };
    std::vector<Member> mb;

    // one persistent thread per member: run(job) executes job(i) on thread i and returns when all are done
    std::vector<std::thread> th;
    std::mutex pmu;
    std::condition_variable pcv, dcv;
    std::function<void(int)> job;
    uint64_t job_gen = 0;
    int job_left = 0;
    bool stop = false;
    Barrier bar;

    void worker(int i) {
        (void)hipSetDevice(dev[i]);
        uint64_t seen = 0;
        for (;;) {
            std::function<void(int)> f;
            {
                std::unique_lock<std::mutex> lk(pmu);
                pcv.wait(lk, [&] { return stop || job_gen != seen; });
                if (stop) return;
                seen = job_gen;
                f = job;
            }
            f(i);
            {
                std::lock_guard<std::mutex> lk(pmu);
                if (--job_left == 0) dcv.notify_all();
            }
        }
    }
    void run(const std::function<void(int)>& f) {
        std::unique_lock<std::mutex> lk(pmu);
        job = f;
        job_left = (int)m.size();
        job_gen++;
        pcv.notify_all();
        dcv.wait(lk, [&] { return job_left == 0; });
    }
};

namespace {

int gfail(gamma_hip_group* g, int code, const std::string& msg) {
    g->err = msg;
    return code;
}
int member_fail(gamma_hip_group* g, int i, int rc) {
    g->err = "member " + std::to_string(i) + ": " + gamma_hip_strerror(rc) + " (" + gamma_hip_last_error(g->m[i]) + ")";
    return rc;
}

// src on device sd -> dst on device dd, ordered on stream s (a stream of device dd)
hipError_t copy_between(void* dst, int dd, const void* src, int sd, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if (dd == sd) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
    return hipMemcpyPeerAsync(dst, dd, src, sd, bytes, s);
}

}  // namespace

extern "C" {

int gamma_hip_group_create(const int* devices, int n, gamma_hip_group** out) {
    if (!out || !devices || n <= 0 || n > 64) return GAMMA_HIP_EINVAL;
    *out = nullptr;
    gamma_hip_group* g = new (std::nothrow) gamma_hip_group();
    if (!g) return GAMMA_HIP_ENOMEM;
    g->dev.assign(devices, devices + n);
    g->mb.resize(n);
    for (int i = 0; i < n; i++) {
        gamma_hip_index* h = nullptr;
        const int rc = gamma_hip_create(devices[i], &h);
        if (rc != GAMMA_HIP_OK) {
            for (auto* mh : g->m) gamma_hip_destroy(mh);
            delete g;
            return rc;
        }
        g->m.push_back(h);
    }
    // direct copies between the members' devices where the fabric allows it (xGMI); failure is not an error, the
    // runtime then stages peer copies itself
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(devices[i]);
        for (int j = 0; j < n; j++) {
            if (devices[j] == devices[i]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
                if (e != hipSuccess) (void)hipGetLastError();   // already enabled, or refused: both fine
            }
        }
        if (hipEventCreateWithFlags(&g->mb[i].ev_coarse, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->mb[i].ev_tie, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->mb[i].ev_bound, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->mb[i].ev_scan, hipEventDisableTiming) != hipSuccess) {
            for (auto* mh : g->m) gamma_hip_destroy(mh);
            delete g;
            return GAMMA_HIP_EDEVICE;
        }
    }
    g->bar.n = n;
    for (int i = 0; i < n; i++) g->th.emplace_back([g, i] { g->worker(i); });
    *out = g;
    return GAMMA_HIP_OK;
}

int gamma_hip_group_destroy(gamma_hip_group* g) {
    if (!g) return GAMMA_HIP_OK;
    {
        std::lock_guard<std::mutex> lk(g->pmu);
        g->stop = true;
        g->pcv.notify_all();
    }
    for (auto& t : g->th)
        if (t.joinable()) t.join();
    for (size_t i = 0; i < g->m.size(); i++) {
        (void)hipSetDevice(g->dev[i]);
        (void)gamma_hip_synchronize(g->m[i]);
        gamma_hip_group::Member& b = g->mb[i];
        for (GBuf* p : {&b.x, &b.cdis, &b.probe, &b.rdis, &b.rids, &b.all_dis, &b.all_ids, &b.D, &b.I, &b.fx, &b.fcd, &b.fpr, &b.sx,
                        &b.scd, &b.spr, &b.ex_vals, &b.ex_ids, &b.ex_off, &b.av, &b.ai, &b.ao, &b.cutf, &b.cutall})
            p->release();
        if (b.ev_tie) (void)hipEventDestroy(b.ev_tie);
        if (b.ev_bound) (void)hipEventDestroy(b.ev_bound);
        if (b.ev_coarse) (void)hipEventDestroy(b.ev_coarse);
        if (b.ev_scan) (void)hipEventDestroy(b.ev_scan);
    }
    if (!g->comm.empty() && rccl_api()->CommDestroy)
        for (void* c : g->comm)
            if (c) (void)rccl_api()->CommDestroy(c);
    for (auto* h : g->m) gamma_hip_destroy(h);
    delete g;
    return GAMMA_HIP_OK;
}

int gamma_hip_group_set_transport(gamma_hip_group* g, int rccl) {
    if (!g || (rccl != 0 && rccl != 1)) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    g->transport = rccl;
    if (rccl == 1 && g->comm.empty()) g->comm_tried = false;   // (formed at the next sharded search)
    return GAMMA_HIP_OK;
}

int gamma_hip_group_transport(gamma_hip_group* g, int64_t* out2) {
    if (!g || !out2) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    out2[0] = (g->transport == 1 && !g->comm.empty()) ? 1 : 0;
    out2[1] = g->rccl_calls;
    return GAMMA_HIP_OK;
}

const char* gamma_hip_group_transport_note(gamma_hip_group* g) { return g ? g->transport_note.c_str() : ""; }

int gamma_hip_group_size(const gamma_hip_group* g) { return g ? (int)g->m.size() : 0; }
gamma_hip_index* gamma_hip_group_member(gamma_hip_group* g, int i) {
    return (g && i >= 0 && i < (int)g->m.size()) ? g->m[i] : nullptr;
}
const char* gamma_hip_group_last_error(gamma_hip_group* g) { return g ? g->err.c_str() : "null group"; }

int gamma_hip_group_set_owners(gamma_hip_group* g, const int64_t* weights) {
    if (!g) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    const int W = (int)g->m.size();
    const int nlist = gamma_hip_ivfpq_nlist(g->m[0]);
    if (nlist <= 0) return gfail(g, GAMMA_HIP_EINVAL, "set_owners: members not initialised");
    for (int i = 1; i < W; i++)
        if (gamma_hip_ivfpq_nlist(g->m[i]) != nlist) return gfail(g, GAMMA_HIP_EINVAL, "set_owners: members differ in nlist");
    g->owner.assign(nlist, 0);
    if (g->replicate) {
        // every member holds every list (member 0 answers the per-list getters); no mask
        for (int i = 0; i < W; i++) {
            const int rc = gamma_hip_ivfpq_set_list_mask(g->m[i], nullptr);
            if (rc) return member_fail(g, i, rc);
        }
        return GAMMA_HIP_OK;
    }
    if (!weights) {
        for (int l = 0; l < nlist; l++) g->owner[l] = l % W;
    } else {
        // greedy longest-processing-time on the weights (gamma_amd/dist.py balance_lists): heaviest list first onto the
        // lightest member; the +1 spreads empty lists as well
        std::vector<int> order(nlist);
        for (int l = 0; l < nlist; l++) order[l] = l;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weights[a] > weights[b]; });
        std::vector<int64_t> load(W, 0);
        for (int l : order) {
            int best = 0;
            for (int i = 1; i < W; i++)
                if (load[i] < load[best]) best = i;
            g->owner[l] = best;
            load[best] += std::max<int64_t>(0, weights[l]) + 1;
        }
    }
    std::vector<uint8_t> mask(nlist);
    for (int i = 0; i < W; i++) {
        for (int l = 0; l < nlist; l++) mask[l] = g->owner[l] == i ? 1 : 0;
        const int rc = gamma_hip_ivfpq_set_list_mask(g->m[i], mask.data());
        if (rc) return member_fail(g, i, rc);
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_group_set_placement(gamma_hip_group* g, int replicate) {
    if (!g) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    if (!g->owner.empty() && g->replicate != (replicate != 0))
        return gfail(g, GAMMA_HIP_EINVAL, "set_placement: before gamma_hip_group_set_owners");
    g->replicate = replicate != 0;
    return GAMMA_HIP_OK;
}
int gamma_hip_group_placement(const gamma_hip_group* g) { return g && g->replicate ? 1 : 0; }

int gamma_hip_group_owner(const gamma_hip_group* g, int l) {
    return (g && l >= 0 && l < (int)g->owner.size()) ? g->owner[l] : -1;
}

int gamma_hip_group_ivfpq_add(gamma_hip_group* g, int64_t n, const float* vecs, int64_t first_vid) {
    if (!g || n < 0 || (n > 0 && !vecs)) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->owner.empty()) return gfail(g, GAMMA_HIP_EINVAL, "add: gamma_hip_group_set_owners first");
    const int W = (int)g->m.size(), nlist = (int)g->owner.size();
    // one encode for the batch (quantizer->assign + residual + pq.compute_codes, gamma_index_ivfpq.cc:424-512)
    const int e = g->next_enc++ % W;
    const int cs = gamma_hip_ivfpq_code_size(g->m[e]);
    if (cs <= 0) return gfail(g, GAMMA_HIP_EINVAL, "add: members not initialised");
    std::vector<int64_t> lno(n);
    std::vector<uint8_t> codes((size_t)n * cs);
    int rc = gamma_hip_ivfpq_encode(g->m[e], n, vecs, lno.data(), codes.data());
    if (rc) return member_fail(g, e, rc);
    // AddKeys at the owner, lists in ascending order, entries in batch order (the std::map of gamma_index_ivfpq.cc:428-494)
    std::vector<int64_t> order(n);
    for (int64_t i = 0; i < n; i++) {
        if (lno[i] < 0 || lno[i] >= nlist) lno[i] = (first_vid + i) % nlist;
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return lno[a] < lno[b]; });
    for (int o = 0; o < W; o++) {
        std::vector<int32_t> lists, counts;
        std::vector<int64_t> vids;
        std::vector<uint8_t> gcodes;
        for (int64_t i = 0; i < n; i++) {
            const int64_t src = order[i];
            if (!g->replicate && g->owner[lno[src]] != o) continue;
            if (lists.empty() || lists.back() != (int32_t)lno[src]) {
                lists.push_back((int32_t)lno[src]);
                counts.push_back(0);
            }
            counts.back()++;
            vids.push_back(first_vid + src);
            gcodes.insert(gcodes.end(), codes.begin() + (size_t)src * cs, codes.begin() + (size_t)(src + 1) * cs);
        }
        if (lists.empty()) continue;
        if (g->replicate) {   // the same grouped batch to every member
            for (int i = 0; i < W; i++) {
                rc = gamma_hip_ivfpq_add_keys_batch(g->m[i], (int)lists.size(), lists.data(), counts.data(), vids.data(), gcodes.data());
                if (rc) return member_fail(g, i, rc);
            }
            break;
        }
        rc = gamma_hip_ivfpq_add_keys_batch(g->m[o], (int)lists.size(), lists.data(), counts.data(), vids.data(), gcodes.data());
        if (rc) return member_fail(g, o, rc);
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_group_ivfpq_add_keys(gamma_hip_group* g, int l, int n, const int64_t* vids, const uint8_t* codes) {
    if (!g) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    if (l < 0 || l >= (int)g->owner.size()) return gfail(g, GAMMA_HIP_EINVAL, "add_keys: bad list (or no owners yet)");
    const int o = g->owner[l];
    if (g->replicate) {
        for (size_t i = 0; i < g->m.size(); i++) {
            const int rc = gamma_hip_ivfpq_add_keys(g->m[i], l, n, vids, codes);
            if (rc) return member_fail(g, (int)i, rc);
        }
        return GAMMA_HIP_OK;
    }
    const int rc = gamma_hip_ivfpq_add_keys(g->m[o], l, n, vids, codes);
    return rc ? member_fail(g, o, rc) : GAMMA_HIP_OK;
}

int64_t gamma_hip_group_ivfpq_list_size(gamma_hip_group* g, int l) {
    if (!g) return -1;
    std::lock_guard<std::mutex> lk(g->mu);
    if (l < 0 || l >= (int)g->owner.size()) return -1;
    return gamma_hip_ivfpq_list_size(g->m[g->owner[l]], l);
}

int gamma_hip_group_ivfpq_get_list(gamma_hip_group* g, int l, int64_t* vids, uint8_t* codes) {
    if (!g) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    if (l < 0 || l >= (int)g->owner.size()) return gfail(g, GAMMA_HIP_EINVAL, "get_list: bad list (or no owners yet)");
    const int o = g->owner[l];
    const int rc = gamma_hip_ivfpq_get_list(g->m[o], l, vids, codes);
    return rc ? member_fail(g, o, rc) : GAMMA_HIP_OK;
}

int gamma_hip_group_ivfpq_update(gamma_hip_group* g, int n, const int64_t* vids, const float* vecs) {
    if (!g || n < 0 || (n > 0 && (!vids || !vecs))) return GAMMA_HIP_EINVAL;
    if (n == 0) return GAMMA_HIP_OK;
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->owner.empty()) return gfail(g, GAMMA_HIP_EINVAL, "update: gamma_hip_group_set_owners first");
    const int W = (int)g->m.size(), nlist = (int)g->owner.size();
    const int e = g->next_enc++ % W;
    const int cs = gamma_hip_ivfpq_code_size(g->m[e]);
    if (cs <= 0) return gfail(g, GAMMA_HIP_EINVAL, "update: bad code size");
    std::vector<int64_t> lno(n);
    std::vector<uint8_t> codes((size_t)n * cs);
    int rc = gamma_hip_ivfpq_encode_each(g->m[e], n, vecs, lno.data(), codes.data());
    if (rc) return member_fail(g, e, rc);
    if (g->replicate) {   // one encode, the same in-place Update at every member (a member ignores vids it never held)
        std::vector<int32_t> l32(n);
        for (int j = 0; j < n; j++) l32[j] = (int32_t)lno[j];
        for (int i = 0; i < W; i++) {
            rc = gamma_hip_ivfpq_apply_updates(g->m[i], n, l32.data(), vids, codes.data(), nullptr);
            if (rc) return member_fail(g, i, rc);
        }
        return GAMMA_HIP_OK;
    }
    // who holds each vid now (at most one member); kept current while the batch is routed, so that a vid named
    // twice is followed through its first move
    std::unordered_map<int64_t, int> holder;
    {
        std::vector<uint8_t> has(n);
        for (int i = 0; i < W; i++) {
            rc = gamma_hip_ivfpq_has_vid(g->m[i], vids, n, has.data());
            if (rc) return member_fail(g, i, rc);
            for (int j = 0; j < n; j++)
                if (has[j]) holder[vids[j]] = i;
        }
    }
    // RealTimeMemData::Update (realtime_mem_data.cc:305-327) when the list a vector leaves and the list it joins may
    // belong to different members: per member the entries that concern it, in batch order
    std::vector<std::vector<int>> idx(W);
    std::vector<std::vector<uint8_t>> ops(W);
    for (int j = 0; j < n; j++) {
        auto it = holder.find(vids[j]);
        if (it == holder.end()) continue;   // never added: Update ignores it (:307-311)
        if (lno[j] < 0 || lno[j] >= nlist) continue;
        const int from = it->second, to = g->owner[lno[j]];
        if (from == to) {
            idx[to].push_back(j);
            ops[to].push_back(0);
        } else {
            idx[from].push_back(j);
            ops[from].push_back(2);   // leaves: flag the old entry
            idx[to].push_back(j);
            ops[to].push_back(1);     // the owner of the new list appends
            it->second = to;
        }
    }
    for (int i = 0; i < W; i++) {
        const int ni = (int)idx[i].size();
        if (ni == 0) continue;
        std::vector<int32_t> l32(ni);
        std::vector<int64_t> v(ni);
        std::vector<uint8_t> c((size_t)ni * cs);
        for (int t = 0; t < ni; t++) {
            const int j = idx[i][t];
            l32[t] = (int32_t)lno[j];
            v[t] = vids[j];
            memcpy(c.data() + (size_t)t * cs, codes.data() + (size_t)j * cs, cs);
        }
        rc = gamma_hip_ivfpq_apply_updates(g->m[i], ni, l32.data(), v.data(), c.data(), ops[i].data());
        if (rc) return member_fail(g, i, rc);
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_group_ivfpq_delete(gamma_hip_group* g, const int64_t* vids, int n) {
    if (!g || (n > 0 && !vids)) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    for (size_t i = 0; i < g->m.size(); i++) {   // a member counts the vids it holds and ignores the others
        const int rc = gamma_hip_ivfpq_delete(g->m[i], vids, n);
        if (rc) return member_fail(g, (int)i, rc);
    }
    return GAMMA_HIP_OK;
}

int gamma_hip_group_ivfpq_compact_if_need(gamma_hip_group* g) {
    if (!g) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    for (size_t i = 0; i < g->m.size(); i++) {
        const int rc = gamma_hip_ivfpq_compact_if_need(g->m[i]);
        if (rc) return member_fail(g, (int)i, rc);
    }
    return GAMMA_HIP_OK;
}

int64_t gamma_hip_group_total_mem_bytes(gamma_hip_group* g) {
    if (!g) return 0;
    int64_t t = 0;
    for (auto* h : g->m) t += gamma_hip_total_mem_bytes(h);
    return t;
}

static int group_search(gamma_hip_group* g, const gamma_hip_search_params* p, int nq, const float* x, int k,
                        float* distances, int64_t* labels, bool on_device);

int gamma_hip_group_ivfpq_search(gamma_hip_group* g, const gamma_hip_search_params* p, int nq, const float* x, int k,
                                 float* distances, int64_t* labels) {
    return group_search(g, p, nq, x, k, distances, labels, false);
}

int gamma_hip_group_ivfpq_search_device(gamma_hip_group* g, const gamma_hip_search_params* p, int nq, const float* d_x, int k,
                                        float* d_distances, int64_t* d_labels) {
    return group_search(g, p, nq, d_x, k, d_distances, d_labels, true);
}

// on_device: x / distances / labels live on member 0's device (the other members pull the batch over the fabric and
// push their slice of the results back); the call still returns when the results are in place
static int group_search(gamma_hip_group* g, const gamma_hip_search_params* p, int nq, const float* x, int k,
                        float* distances, int64_t* labels, bool on_device) {
    if (!g || !p) return GAMMA_HIP_EINVAL;
    if (nq < 0) return GAMMA_HIP_EINVAL;
    if (k <= 0 || nq == 0) return GAMMA_HIP_OK;   // gamma_index_ivfpq.cc:753-756
    if (!x || !distances || !labels) return GAMMA_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->owner.empty()) return gfail(g, GAMMA_HIP_EINVAL, "search: gamma_hip_group_set_owners first");
    const int W = (int)g->m.size();
    const int d = gamma_hip_ivfpq_dim(g->m[0]);
    if (d <= 0) return gfail(g, GAMMA_HIP_EINVAL, "search: members not initialised");
    const int P = p->nprobe, R = std::max(p->recall_num, k);
    if (P <= 0) return gfail(g, GAMMA_HIP_EINVAL, "search: nprobe out of range");
    // faiss chooses the coarse path from the size of the WHOLE call (faiss:utils/distances.cpp:346): the slices must agree
    gamma_hip_search_params pp = *p;
    if (pp.coarse_mode < 0) pp.coarse_mode = nq < 20 ? 0 : 1;
    // (gamma_hip_blas_form_not_restated: every member counts the shape of ITS slice -- a remainder block of the whole call
    //  that a slice boundary hides is not counted here)
    const int per = (nq + W - 1) / W;
    // transport of the two exchanges of the path: RCCL when asked for and a communicator can be formed
    if (g->transport == 1 && !g->replicate && !g->comm_tried) {
        g->comm_tried = true;
        bool distinct = true;
        for (int i = 0; i < W; i++)
            for (int j = 0; j < i; j++) distinct = distinct && g->dev[i] != g->dev[j];
        RcclApi* api = rccl_api();
        if (!distinct) {
            g->transport_note = "RCCL asked for, but members share a device: peer copies";
        } else if (!api->ok()) {
            g->transport_note = "RCCL asked for, but librccl.so could not be loaded: peer copies";
        } else {
            g->comm.assign(W, nullptr);
            const int e = api->CommInitAll(g->comm.data(), W, g->dev.data());
            if (e != 0) {
                g->comm.clear();
                g->transport_note = std::string("ncclCommInitAll failed (") + (api->GetErrorString ? api->GetErrorString(e) : "?") + "): peer copies";
            } else {
                g->transport_note = "RCCL";
            }
        }
    }
    const bool use_rccl = g->transport == 1 && !g->replicate && (int)g->comm.size() == W;
    if (use_rccl) g->rccl_calls++;
    RcclApi* const nccl = use_rccl ? rccl_api() : nullptr;
    std::vector<int> rcs(W, GAMMA_HIP_OK);
    std::vector<std::string> errs(W);
    std::vector<int64_t> rowmax(W, 0);                 // tie phase: every member's longest export row of the round
    std::vector<int> nfl(W, 0);                        // flagged queries of every member's slice
    std::vector<const int32_t*> lists(W, nullptr);     // and the device lists of their slice-local indices
    auto slice = [&](int i, int* q0, int* q1) {
        *q0 = std::min(nq, i * per);
        *q1 = std::min(nq, *q0 + per);
    };
    g->run([&](int i) {
        gamma_hip_group::Member& b = g->mb[i];
        gamma_hip_index* h = g->m[i];
        hipStream_t s = (hipStream_t)gamma_hip_stream(h);
        int q0, q1;
        slice(i, &q0, &q1);
        const int nql = q1 - q0;
        int& rc = rcs[i];
        if (g->replicate) {
            // query-parallel over replicated lists: the member answers its slice with the ordinary single-handle search
            // (exact ties and all); nothing is exchanged but the slice and its k results
            auto hipr = [&](hipError_t e, const char* what) {
                if (e != hipSuccess && rc == GAMMA_HIP_OK) {
                    rc = e == hipErrorOutOfMemory ? GAMMA_HIP_ENOMEM : GAMMA_HIP_EDEVICE;
                    errs[i] = std::string(what) + ": " + hipGetErrorString(e);
                }
            };
            hipr(hipSetDevice(g->dev[i]), "hipSetDevice");
            if (nql <= 0 || rc != GAMMA_HIP_OK) return;
            const bool local = on_device && g->dev[i] == g->dev[0];
            const float* xs = x + (size_t)q0 * d;
            float* Ds = distances + (size_t)q0 * k;
            int64_t* Is = labels + (size_t)q0 * k;
            if (!local) {
                hipr(b.x.ensure((size_t)nql * d * sizeof(float)), "alloc");
                hipr(b.D.ensure((size_t)nql * k * sizeof(float)), "alloc");
                hipr(b.I.ensure((size_t)nql * k * sizeof(int64_t)), "alloc");
                if (rc != GAMMA_HIP_OK) return;
                if (on_device) hipr(copy_between(b.x.p, g->dev[i], xs, g->dev[0], (size_t)nql * d * sizeof(float), s), "queries to the member");
                else hipr(hipMemcpyAsync(b.x.p, xs, (size_t)nql * d * sizeof(float), hipMemcpyHostToDevice, s), "H2D queries");
            }
            if (rc == GAMMA_HIP_OK) {
                const int r = gamma_hip_ivfpq_search_device(h, &pp, nql, local ? xs : b.x.as<float>(), k, local ? Ds : b.D.as<float>(),
                                                           local ? Is : b.I.as<int64_t>());
                if (r != GAMMA_HIP_OK) {
                    rc = r;
                    errs[i] = std::string(gamma_hip_strerror(r)) + " (" + gamma_hip_last_error(h) + ")";
                }
            }
            if (rc == GAMMA_HIP_OK && !local) {
                if (on_device) {
                    hipr(copy_between(Ds, g->dev[0], b.D.p, g->dev[i], (size_t)nql * k * sizeof(float), s), "results");
                    hipr(copy_between(Is, g->dev[0], b.I.p, g->dev[i], (size_t)nql * k * sizeof(int64_t), s), "results");
                } else {
                    hipr(hipMemcpyAsync(Ds, b.D.p, (size_t)nql * k * sizeof(float), hipMemcpyDeviceToHost, s), "D2H");
                    hipr(hipMemcpyAsync(Is, b.I.p, (size_t)nql * k * sizeof(int64_t), hipMemcpyDeviceToHost, s), "D2H");
                }
            }
            hipr(hipStreamSynchronize(s), "sync");
            return;
        }
        auto hip = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && rc == GAMMA_HIP_OK) {
                rc = e == hipErrorOutOfMemory ? GAMMA_HIP_ENOMEM : GAMMA_HIP_EDEVICE;
                errs[i] = std::string(what) + ": " + hipGetErrorString(e);
            }
        };
        auto abi = [&](int r) {
            if (r != GAMMA_HIP_OK && rc == GAMMA_HIP_OK) {
                rc = r;
                errs[i] = std::string(gamma_hip_strerror(r)) + " (" + gamma_hip_last_error(h) + ")";
            }
        };
        hip(hipSetDevice(g->dev[i]), "hipSetDevice");
        hip(b.x.ensure((size_t)nq * d * sizeof(float)), "alloc");
        // (W * per rows: the in-place all-gather moves whole slices, the last one's tail rows are never read)
        hip(b.cdis.ensure((size_t)W * per * P * sizeof(float)), "alloc");
        hip(b.probe.ensure((size_t)W * per * P * sizeof(int32_t)), "alloc");
        hip(b.rdis.ensure((size_t)nq * R * sizeof(float)), "alloc");
        hip(b.rids.ensure((size_t)nq * R * sizeof(int64_t)), "alloc");
        hip(b.all_dis.ensure((size_t)W * per * R * sizeof(float)), "alloc");
        hip(b.all_ids.ensure((size_t)W * per * R * sizeof(int64_t)), "alloc");
        hip(b.D.ensure((size_t)per * k * sizeof(float)), "alloc");
        hip(b.I.ensure((size_t)per * k * sizeof(int64_t)), "alloc");
        // 0. the whole batch to every member (each scans its lists for all queries); coarse quantizer for the own slice
        if (rc == GAMMA_HIP_OK) {
            if (on_device) hip(copy_between(b.x.p, g->dev[i], x, g->dev[0], (size_t)nq * d * sizeof(float), s), "queries to the member");
            else hip(hipMemcpyAsync(b.x.p, x, (size_t)nq * d * sizeof(float), hipMemcpyHostToDevice, s), "H2D queries");
            if (nql > 0)
                abi(gamma_hip_ivfpq_coarse_device(h, &pp, nql, b.x.as<float>() + (size_t)q0 * d, b.cdis.as<float>() + (size_t)q0 * P,
                                                  b.probe.as<int32_t>() + (size_t)q0 * P));
            hip(hipEventRecord(b.ev_coarse, s), "record");
        }
        bool all_ok = g->bar.arrive(rc == GAMMA_HIP_OK);   // every member's assignment is on its stream
        auto ncc = [&](int e, const char* what) {
            if (e != 0 && rc == GAMMA_HIP_OK) {
                rc = GAMMA_HIP_EDEVICE;
                errs[i] = std::string(what) + ": " + (nccl && nccl->GetErrorString ? nccl->GetErrorString(e) : "RCCL error");
            }
        };
        // 1. pull the other slices of the assignment; scan of the owned probed lists, local top-R of every query
        if (all_ok && nccl) {
            // RCCL: ONE all-gather of the assignment over xGMI (in place: this member's slice sits at its offset already)
            ncc(nccl->GroupStart(), "ncclGroupStart");
            ncc(nccl->AllGather(b.cdis.as<float>() + (size_t)i * per * P, b.cdis.p, (size_t)per * P * sizeof(float), kNcclChar,
                                g->comm[i], s), "ncclAllGather");
            ncc(nccl->AllGather(b.probe.as<int32_t>() + (size_t)i * per * P, b.probe.p, (size_t)per * P * sizeof(int32_t), kNcclChar,
                                g->comm[i], s), "ncclAllGather");
            ncc(nccl->GroupEnd(), "ncclGroupEnd");
        }
        if (all_ok) {
            for (int j = 0; j < W && !nccl; j++) {
                if (j == i) continue;
                int a0, a1;
                slice(j, &a0, &a1);
                if (a1 <= a0) continue;
                hip(hipStreamWaitEvent(s, g->mb[j].ev_coarse, 0), "wait");
                hip(copy_between(b.cdis.as<float>() + (size_t)a0 * P, g->dev[i], g->mb[j].cdis.as<float>() + (size_t)a0 * P, g->dev[j],
                                 (size_t)(a1 - a0) * P * sizeof(float), s), "assignment exchange");
                hip(copy_between(b.probe.as<int32_t>() + (size_t)a0 * P, g->dev[i], g->mb[j].probe.as<int32_t>() + (size_t)a0 * P,
                                 g->dev[j], (size_t)(a1 - a0) * P * sizeof(int32_t), s), "assignment exchange");
            }
            // The scan in TWO PHASES around a reduction of one float per query (gamma_hip_ivfpq_search_shard_bounded): every
            // member bounds its own recall_num-th best from the query's nearest probes it owns, the minimum (L2) / maximum (IP)
            // over the members bounds the GLOBAL one, and the members' other probes only keep what is within it.  The
            // reduction: ncclAllReduce over xGMI, or -- peer copies -- every member publishes its bounds, the members meet at a
            // barrier, pull each other's and combine.  It is one more meeting point of the call: a member that does not get
            // as far as its scan arrives there all the same.
            struct Red {
                gamma_hip_group* g; int i, W; RcclApi* nccl; bool* ok_in; bool arrived; std::string* err;
            } red{g, i, W, nccl, nullptr, false, &errs[i]};
            bool my_ok = rc == GAMMA_HIP_OK;
            red.ok_in = &my_ok;
            auto reduce = [](void* user, float* d_bound, int n, int take_max, void* stream) -> int {
                Red& r = *static_cast<Red*>(user);
                gamma_hip_group* g = r.g;
                auto& b = g->mb[r.i];
                hipStream_t s = static_cast<hipStream_t>(stream);
                r.arrived = true;
                if (r.nccl) {
                    const bool all = g->bar.arrive(*r.ok_in);   // (every member is about to enter the collective, or none does)
                    if (!all) return 0;
                    const int e = r.nccl->AllReduce(d_bound, d_bound, (size_t)n, kNcclFloat, take_max ? kNcclMax : kNcclMin, g->comm[r.i], s);
                    if (e != 0) *r.err = std::string("ncclAllReduce: ") + (r.nccl->GetErrorString ? r.nccl->GetErrorString(e) : "RCCL error");
                    return e != 0;
                }
                bool ok = *r.ok_in && b.bound_pub.ensure((size_t)n * sizeof(float)) == hipSuccess &&
                          b.bound_peer.ensure((size_t)n * sizeof(float)) == hipSuccess;
                ok = ok && hipMemcpyAsync(b.bound_pub.p, d_bound, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess &&
                     hipEventRecord(b.ev_bound, s) == hipSuccess;
                const bool all = g->bar.arrive(ok);   // every member's bounds are on its stream
                if (!all) return ok ? 0 : 1;        // (the bounds stay this member's own: still valid)
                for (int j = 0; j < r.W && ok; j++) {
                    if (j == r.i) continue;
                    ok = hipStreamWaitEvent(s, g->mb[j].ev_bound, 0) == hipSuccess &&
                         copy_between(b.bound_peer.p, g->dev[r.i], g->mb[j].bound_pub.p, g->dev[j], (size_t)n * sizeof(float), s) == hipSuccess &&
                         gamma_hip_bound_combine(stream, d_bound, b.bound_peer.as<float>(), n, take_max) == GAMMA_HIP_OK;
                }
                if (!ok) *r.err = "bound reduction: peer copy failed";
                return ok ? 0 : 1;
            };
            hip(b.bound.ensure((size_t)nq * sizeof(float)), "alloc");
            my_ok = rc == GAMMA_HIP_OK;
            if (rc == GAMMA_HIP_OK)
                abi(gamma_hip_ivfpq_search_shard_bounded(h, &pp, nq, b.x.as<float>(), b.cdis.as<float>(), b.probe.as<int32_t>(), k,
                                                         b.rdis.as<float>(), b.rids.as<int64_t>(), b.bound.as<float>(), reduce, &red));
            if (!red.arrived) (void)g->bar.arrive(false);   // (this member never got to the reduction: its peers are waiting there)
            // did this member's own top-R cut of a query go through a tie?  (the merge at the query's owner asks)
            hip(b.cutf.ensure((size_t)nq), "alloc");
            if (rc == GAMMA_HIP_OK) abi(gamma_hip_ivfpq_shard_cut_flags(h, nq, b.cutf.as<uint8_t>()));
            // (what the exchange below receives into is allocated on THIS side of the barrier: a member that cannot get it says so
            //  in the go / no-go snapshot instead of leaving its peers inside a collective -- ADVICE r4.  An RCCL error AFTER the
            //  barrier is fatal for the group: nothing can call a member back out of ncclGroupEnd)
            hip(b.cutall.ensure((size_t)W * per), "alloc");
            hip(hipEventRecord(b.ev_scan, s), "record");
        }
        all_ok = g->bar.arrive(rc == GAMMA_HIP_OK);   // every member's candidate tables are on its stream
        // 2. the exchange of the path: the candidates of the own query slice from every member ([W][per][R]);
        // 3. merge to the global top-R, compute_dis, results to the caller
        if (all_ok && nccl) {
            // RCCL: the per-shard tables of every owner's slice in one grouped exchange (every member sends slice j of its
            // tables to member j and receives its own slice of theirs -- an all-to-all over xGMI); members with an empty
            // slice still take part
            ncc(nccl->GroupStart(), "ncclGroupStart");
            for (int j = 0; j < W; j++) {   // (never cut short: the other members' halves of the exchange are under way)
                int a0, a1;
                slice(j, &a0, &a1);
                if (a1 > a0) {
                    ncc(nccl->Send(b.rdis.as<float>() + (size_t)a0 * R, (size_t)(a1 - a0) * R * sizeof(float), kNcclChar, j, g->comm[i], s), "ncclSend");
                    ncc(nccl->Send(b.rids.as<int64_t>() + (size_t)a0 * R, (size_t)(a1 - a0) * R * sizeof(int64_t), kNcclChar, j, g->comm[i], s), "ncclSend");
                    ncc(nccl->Send(b.cutf.as<uint8_t>() + a0, (size_t)(a1 - a0), kNcclChar, j, g->comm[i], s), "ncclSend");
                }
                if (nql > 0) {
                    ncc(nccl->Recv(b.all_dis.as<float>() + (size_t)j * per * R, (size_t)nql * R * sizeof(float), kNcclChar, j, g->comm[i], s), "ncclRecv");
                    ncc(nccl->Recv(b.all_ids.as<int64_t>() + (size_t)j * per * R, (size_t)nql * R * sizeof(int64_t), kNcclChar, j, g->comm[i], s), "ncclRecv");
                    ncc(nccl->Recv(b.cutall.as<uint8_t>() + (size_t)j * per, (size_t)nql, kNcclChar, j, g->comm[i], s), "ncclRecv");
                }
            }
            ncc(nccl->GroupEnd(), "ncclGroupEnd");
        }
        if (all_ok && nql > 0) {
            for (int j = 0; j < W && !nccl; j++) {
                if (j != i) hip(hipStreamWaitEvent(s, g->mb[j].ev_scan, 0), "wait");
                hip(copy_between(b.all_dis.as<float>() + (size_t)j * per * R, g->dev[i], g->mb[j].rdis.as<float>() + (size_t)q0 * R,
                                 g->dev[j], (size_t)nql * R * sizeof(float), s), "candidate exchange");
                hip(copy_between(b.all_ids.as<int64_t>() + (size_t)j * per * R, g->dev[i], g->mb[j].rids.as<int64_t>() + (size_t)q0 * R,
                                 g->dev[j], (size_t)nql * R * sizeof(int64_t), s), "candidate exchange");
            }
            for (int j = 0; j < W && rc == GAMMA_HIP_OK && !nccl; j++)
                hip(copy_between(b.cutall.as<uint8_t>() + (size_t)j * per, g->dev[i], g->mb[j].cutf.as<uint8_t>() + q0, g->dev[j], (size_t)nql, s),
                    "cut flags");
            if (rc == GAMMA_HIP_OK) abi(gamma_hip_ivfpq_merge_set_shard_flags(h, b.cutall.as<uint8_t>()));
            if (rc == GAMMA_HIP_OK)
                abi(gamma_hip_ivfpq_merge_rerank(h, &pp, W, per, b.x.as<float>() + (size_t)q0 * d, k, b.all_dis.as<float>(),
                                                 b.all_ids.as<int64_t>(), 0, nql, b.D.as<float>(), b.I.as<int64_t>()));
        }
        // 4. exact ties across members (include/gamma_hip.h): the queries a tie can change, listed by their owner, get their
        //    candidate streams exported by every member and replayed at the owner
        static const bool gdbg = getenv("GAMMA_HIP_GROUP_DBG") != nullptr;
        const auto tp0 = std::chrono::steady_clock::now();
        auto since = [&](std::chrono::steady_clock::time_point t) {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
        };
        {
            int nf_i = 0;
            const int32_t* list_i = nullptr;
            if (all_ok && nql > 0 && rc == GAMMA_HIP_OK) abi(gamma_hip_ivfpq_merge_flagged(h, &nf_i, &list_i));
            nfl[i] = rc == GAMMA_HIP_OK ? nf_i : 0;
            lists[i] = list_i;
            if (gdbg && i == 0) fprintf(stderr, "group: member 0 waited %.3f ms for its merge, %d flagged\n", since(tp0), nf_i);
            all_ok = g->bar.arrive(rc == GAMMA_HIP_OK);
            int total = 0;
            for (int j = 0; j < W; j++) total += nfl[j];
            if (all_ok && total > 0) {
                const int fcap = 256;   // flagged queries per round
                for (int o = 0; o < W; o++) {
                    for (int f0 = 0; f0 < nfl[o]; f0 += fcap) {
                        const int nf = std::min(fcap, nfl[o] - f0);
                        if (i == o) {   // the flagged queries' vectors and assignment rows, compact
                            hip(b.fx.ensure((size_t)nf * d * sizeof(float)), "alloc");
                            hip(b.fcd.ensure((size_t)nf * P * sizeof(float)), "alloc");
                            hip(b.fpr.ensure((size_t)nf * P * sizeof(int32_t)), "alloc");
                            if (rc == GAMMA_HIP_OK) {
                                abi(gamma_hip_gather_rows(h, b.x.as<float>() + (size_t)q0 * d, d, lists[o] + f0, nf, b.fx.p));
                                abi(gamma_hip_gather_rows(h, b.cdis.as<float>() + (size_t)q0 * P, P, lists[o] + f0, nf, b.fcd.p));
                                abi(gamma_hip_gather_rows(h, b.probe.as<int32_t>() + (size_t)q0 * P, P, lists[o] + f0, nf, b.fpr.p));
                            }
                            hip(hipEventRecord(b.ev_tie, s), "record");
                        }
                        bool ok2 = g->bar.arrive(rc == GAMMA_HIP_OK);   // the owner's compact inputs are on its stream
                        gamma_hip_group::Member& ob = g->mb[o];
                        hip(b.sx.ensure((size_t)nf * d * sizeof(float)), "alloc");
                        hip(b.scd.ensure((size_t)nf * P * sizeof(float)), "alloc");
                        hip(b.spr.ensure((size_t)nf * P * sizeof(int32_t)), "alloc");
                        hip(b.ex_off.ensure((size_t)nf * (P + 1) * sizeof(int32_t)), "alloc");
                        rowmax[i] = 0;
                        if (ok2 && rc == GAMMA_HIP_OK) {
                            if (i != o) hip(hipStreamWaitEvent(s, ob.ev_tie, 0), "wait");
                            hip(copy_between(b.sx.p, g->dev[i], ob.fx.p, g->dev[o], (size_t)nf * d * sizeof(float), s), "tie inputs");
                            hip(copy_between(b.scd.p, g->dev[i], ob.fcd.p, g->dev[o], (size_t)nf * P * sizeof(float), s), "tie inputs");
                            hip(copy_between(b.spr.p, g->dev[i], ob.fpr.p, g->dev[o], (size_t)nf * P * sizeof(int32_t), s), "tie inputs");
                            // how long this member's export rows get: the exports of all members share one row stride
                            int64_t mine = 0;
                            if (rc == GAMMA_HIP_OK) abi(gamma_hip_ivfpq_shard_export_rows(h, &pp, nf, b.spr.as<int32_t>(), &mine));
                            rowmax[i] = mine;
                        }
                        ok2 = g->bar.arrive(rc == GAMMA_HIP_OK);   // every member's longest row is known
                        int64_t stride = 4;
                        for (int j = 0; j < W; j++) stride = std::max<int64_t>(stride, (rowmax[j] + 3) & ~(int64_t)3);
                        hip(b.ex_vals.ensure((size_t)nf * stride * sizeof(float)), "alloc");
                        hip(b.ex_ids.ensure((size_t)nf * stride * sizeof(int64_t)), "alloc");
                        if (ok2 && rc == GAMMA_HIP_OK)
                            abi(gamma_hip_ivfpq_shard_export(h, &pp, nf, b.sx.as<float>(), b.scd.as<float>(), b.spr.as<int32_t>(), stride,
                                                             b.ex_vals.as<float>(), b.ex_ids.as<int64_t>(), b.ex_off.as<int32_t>()));
                        hip(hipStreamSynchronize(s), "sync");
                        ok2 = g->bar.arrive(rc == GAMMA_HIP_OK);   // every member's export is complete
                        if (i == o && ok2) {
                            hip(b.av.ensure((size_t)W * nf * stride * sizeof(float)), "alloc");
                            hip(b.ai.ensure((size_t)W * nf * stride * sizeof(int64_t)), "alloc");
                            hip(b.ao.ensure((size_t)W * nf * (P + 1) * sizeof(int32_t)), "alloc");
                            for (int j = 0; j < W && rc == GAMMA_HIP_OK; j++) {
                                hip(copy_between(b.av.as<float>() + (size_t)j * nf * stride, g->dev[i], g->mb[j].ex_vals.p, g->dev[j],
                                                 (size_t)nf * stride * sizeof(float), s), "exports");
                                hip(copy_between(b.ai.as<int64_t>() + (size_t)j * nf * stride, g->dev[i], g->mb[j].ex_ids.p, g->dev[j],
                                                 (size_t)nf * stride * sizeof(int64_t), s), "exports");
                                hip(copy_between(b.ao.as<int32_t>() + (size_t)j * nf * (P + 1), g->dev[i], g->mb[j].ex_off.p, g->dev[j],
                                                 (size_t)nf * (P + 1) * sizeof(int32_t), s), "exports");
                            }
                            if (rc == GAMMA_HIP_OK)
                                abi(gamma_hip_ivfpq_merge_replay(h, &pp, W, nf, b.x.as<float>() + (size_t)q0 * d, stride, b.av.as<float>(),
                                                                 b.ai.as<int64_t>(), b.ao.as<int32_t>(), k, lists[o] + f0, b.D.as<float>(),
                                                                 b.I.as<int64_t>()));
                            hip(hipStreamSynchronize(s), "sync");
                        }
                        (void)g->bar.arrive(rc == GAMMA_HIP_OK);   // the exports may be overwritten
                    }
                }
            }
        }
        if (gdbg && i == 0) fprintf(stderr, "group: tie phase done %.3f ms after the merge was enqueued\n", since(tp0));
        // (all_ok: the snapshot of the last barrier every member passed -- the tie phase's own failures show in rc)
        if (all_ok && nql > 0) {   // the slice's rows to the caller
            if (rc == GAMMA_HIP_OK && on_device) {
                hip(copy_between(distances + (size_t)q0 * k, g->dev[0], b.D.p, g->dev[i], (size_t)nql * k * sizeof(float), s), "results");
                hip(copy_between(labels + (size_t)q0 * k, g->dev[0], b.I.p, g->dev[i], (size_t)nql * k * sizeof(int64_t), s), "results");
            } else if (rc == GAMMA_HIP_OK) {
                hip(hipMemcpyAsync(distances + (size_t)q0 * k, b.D.p, (size_t)nql * k * sizeof(float), hipMemcpyDeviceToHost, s), "D2H");
                hip(hipMemcpyAsync(labels + (size_t)q0 * k, b.I.p, (size_t)nql * k * sizeof(int64_t), hipMemcpyDeviceToHost, s), "D2H");
            }
        }
        hip(hipStreamSynchronize(s), "sync");
        (void)g->bar.arrive(rc == GAMMA_HIP_OK);   // nobody is reading this member's tables any more
    });
    for (int i = 0; i < W; i++)
        if (rcs[i] != GAMMA_HIP_OK) return gfail(g, rcs[i], "member " + std::to_string(i) + ": " + errs[i]);
    return GAMMA_HIP_OK;
}

}  // extern "C"
