// plugin_stress.cc -- TEST INFRASTRUCTURE: the engine's threading contract (Search from any number of client
// threads while ONE indexing thread adds / updates and API threads delete; tests/test.h:1033-1062, search/gamma_engine.cc
// 1012-1043, 802-824) driven against the HIPIVFPQ and HIPFLAT plugins through the harness, with the C ABI stubbed on the
// CPU oracle (stub_abi.cpp).  Built with -fsanitize=thread and with -fsanitize=address,undefined by tests/sanitize/Makefile;
// tests/test_sanitizers.py runs both and fails on any report.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <thread>
#include <vector>

extern "C" {
void* gh_host_new(const char* retrieval_type, int d);
void gh_host_free(void* hp);
int gh_host_init(void* hp, const char* retrieval_param, int indexing_size);
void gh_host_store(void* hp, int n, const float* x);
int gh_host_indexing(void* hp);
int gh_host_add(void* hp, int n, const float* x);
int gh_host_update(void* hp, int64_t vid, const float* x);
int gh_host_delete(void* hp, const int64_t* vids, int n);
int gh_host_search(void* hp, const char* retrieval_params, int has_rank, int brute_force, float min_score, float max_score, int n,
                   const float* x, int k, float* distances, int64_t* ids);
long gh_host_mem_bytes(void* hp);
}

static uint32_t rng_state = 12345;
static float frand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((rng_state >> 8) & 0xffff) / 256.f;
}

static int run(const char* type, const char* params, bool ivf) {
    const int d = 32, n0 = 6000, nadd = 6000, batch = 500, nq = 64, k = 10;
    std::vector<float> base((size_t)(n0 + nadd) * d), q((size_t)nq * d);
    for (auto& v : base) v = frand();
    for (auto& v : q) v = frand();
    void* h = gh_host_new(type, d);
    if (!h) return 10;
    if (gh_host_init(h, params, 4000)) return 11;
    gh_host_store(h, n0, base.data());
    if (ivf && gh_host_indexing(h)) return 12;
    for (int i = 0; i < n0; i += batch)
        if (!gh_host_add(h, batch, base.data() + (size_t)i * d)) return 13;
    std::atomic<int> failed(0), stop(0), searches(0);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++)   // client threads: one query per call, as tests/test.h issues them
        th.emplace_back([&, t]() {
            std::vector<float> D(k);
            std::vector<int64_t> I(k);
            for (int i = t; !stop.load(); i += 4) {
                const int rc = gh_host_search(h, "", (i & 1), !ivf || (i % 7 == 0), -1e30f, 1e30f, 1, q.data() + (size_t)(i % nq) * d, k,
                                              D.data(), I.data());
                if (rc) failed++;
                for (int j = 0; j < k; j++)
                    if (I[j] < -1 || I[j] >= n0 + nadd) failed++;
                searches++;
            }
        });
    std::thread deleter([&]() {   // API thread: deletes of vectors added long ago
        for (int i = 0; i < 300 && !stop.load(); i++) {
            int64_t v = (int64_t)(i * 17 % n0);
            if (gh_host_delete(h, &v, 1)) failed++;
            std::this_thread::yield();
        }
    });
    // the indexing thread: the store grows, then Add, then an update pass
    for (int i = 0; i < nadd; i += batch) {
        gh_host_store(h, batch, base.data() + (size_t)(n0 + i) * d);
        if (!gh_host_add(h, batch, base.data() + (size_t)(n0 + i) * d)) failed++;
        if (ivf)
            for (int u = 0; u < 5; u++) {
                const int64_t vid = (int64_t)((i + u * 131) % (n0 + i));
                if (gh_host_update(h, vid, base.data() + (size_t)((vid * 7 + 3) % (n0 + nadd)) * d)) failed++;
            }
        (void)gh_host_mem_bytes(h);
    }
    while (searches.load() < 200) std::this_thread::yield();
    stop = 1;
    for (auto& t : th) t.join();
    deleter.join();
    gh_host_free(h);
    fprintf(stderr, "%s: %d searches, %d failures\n", type, searches.load(), failed.load());
    return failed.load() ? 1 : 0;
}

int main() {
#ifdef GAMMA_STRESS_GROUP
    // the REAL gamma_hip_group.cpp (member threads, barriers, go / no-go snapshots) behind the plugin's "devices" key, on three
    // stub handles: replicate placement (every member holds every list, queries split)
    return run("HIPIVFPQ", "{\"ncentroids\": 32, \"nsubvector\": 8, \"nprobe\": 8, \"metric_type\": \"L2\", \"devices\": \"0,1,2\", "
                           "\"placement\": \"replicate\"}", true);
#else
    int rc = run("HIPIVFPQ", "{\"ncentroids\": 32, \"nsubvector\": 8, \"nprobe\": 8, \"metric_type\": \"L2\"}", true);
    if (rc) return rc;
    return run("HIPFLAT", "{\"metric_type\": \"L2\"}", false);
#endif
}
