// gamma_hip_train.cpp -- training on the device: faiss::Clustering::train (faiss:Clustering.cpp:255-560) the way
// GammaIVFPQIndex::Indexing runs it (index/impl/gamma_index_ivfpq.cc:272-354 -> IndexIVFPQ::train).
// The training set stays in HBM; per iteration the device assigns every point to its nearest centroid (the coarse
// quantizer's kernels: GEMM form from 20 points on, as faiss's IndexFlatL2::search) and sums every cluster's points --
// in ascending point order, one float accumulator per (cluster, dimension), which is compute_centroids' arithmetic --
// and the host does what is sequential and tiny: the random permutations (std::mt19937, faiss:utils/random.cpp), the
// stable grouping of the points by cluster, and split_clusters for clusters left empty.  Same subsampling, same
// initialisation, same sums, same splits as the library: oracle/gamma_oracle.c go_kmeans is the CPU statement of
// exactly this function (bit-identical results, tests/test_gpu_training.py) and is itself pinned against the
// compiled faiss (tests/test_training_cpu.py).
#include <random>

#include "gamma_hip_internal.h"

using namespace ghi;

namespace {

// rand_perm, faiss:utils/random.cpp:136-146
void rand_perm(std::vector<int>& perm, size_t n, int64_t seed) {
    perm.resize(n);
    for (size_t i = 0; i < n; i++) perm[i] = (int)i;
    std::mt19937 mt((unsigned int)seed);
    for (size_t i = 0; i + 1 < n; i++) {
        const int i2 = (int)(i + mt() % (unsigned long)(int)(n - i));
        std::swap(perm[i], perm[i2]);
    }
}

// split_clusters, faiss:Clustering.cpp:220-268
int split_clusters(int d, int k, int64_t n, float* hassign, float* centroids) {
    int nsplit = 0;
    std::mt19937 mt(1234u);
    for (int ci = 0; ci < k; ci++) {
        if (hassign[ci] != 0) continue;
        int cj;
        for (cj = 0; 1; cj = (cj + 1) % k) {
            const float p = (float)(((double)hassign[cj] - 1.0) / (double)(float)(n - k));
            const float r = mt() / float(mt.max());
            if (r < p) break;
        }
        memcpy(centroids + (size_t)ci * d, centroids + (size_t)cj * d, sizeof(float) * d);
        for (int j = 0; j < d; j++) {
            float& a = centroids[(size_t)ci * d + j];
            float& b = centroids[(size_t)cj * d + j];
            if (j % 2 == 0) {
                a = (float)((double)a * (1 + 1 / 1024.));
                b = (float)((double)b * (1 - 1 / 1024.));
            } else {
                a = (float)((double)a * (1 - 1 / 1024.));
                b = (float)((double)b * (1 + 1 / 1024.));
            }
        }
        hassign[ci] = hassign[cj] / 2;
        hassign[cj] -= hassign[ci];
        nsplit++;
    }
    return nsplit;
}

}  // namespace

extern "C" {

void gamma_hip_rand_perm(int32_t* perm, int64_t n, int64_t seed) {
    std::vector<int> p;
    rand_perm(p, (size_t)std::max<int64_t>(n, 0), seed);
    for (int64_t i = 0; i < n; i++) perm[i] = p[i];
}

int gamma_hip_kmeans(gamma_hip_index* h, int d, int64_t n, const float* x_in, int k, int niter, int64_t seed,
                     int max_points_per_centroid, float* centroids, float* objective) {
    if (!h || d <= 0 || k <= 0 || niter < 0 || max_points_per_centroid <= 0 || !x_in || !centroids) return GAMMA_HIP_EINVAL;
    if (n < k) return fail(h, GAMMA_HIP_EINVAL, "k-means: fewer training points than clusters");
    SearchLock lk(h);
    GH_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (objective) *objective = 0.f;
    // subsample_training_set (:92-120): the first k * max_points of a permutation
    std::vector<float> xsub;
    const float* x = x_in;
    if (n > (int64_t)k * max_points_per_centroid) {
        std::vector<int> perm;
        rand_perm(perm, (size_t)n, seed);
        const int64_t n2 = (int64_t)k * max_points_per_centroid;
        xsub.resize((size_t)n2 * d);
        for (int64_t i = 0; i < n2; i++) memcpy(&xsub[(size_t)i * d], x_in + (size_t)perm[i] * d, sizeof(float) * d);
        n = n2;
        x = xsub.data();
    }
    if (n == k) {   // :334-355
        memcpy(centroids, x, sizeof(float) * (size_t)d * k);
        return GAMMA_HIP_OK;
    }
    {   // initial centroids: k points of a second permutation (:412-420)
        std::vector<int> perm;
        rand_perm(perm, (size_t)n, seed + 1);
        for (int i = 0; i < k; i++) memcpy(centroids + (size_t)i * d, x + (size_t)perm[i] * d, sizeof(float) * d);
    }
    if (niter == 0) return GAMMA_HIP_OK;
    // device state: the training set (resident for the whole run), the centroids, their norms
    DevBuf d_x, d_cen, d_cn, d_assign, d_dis, d_order, d_seg, d_has;
    auto cleanup = [&]() {
        for (DevBuf* b : {&d_x, &d_cen, &d_cn, &d_assign, &d_dis, &d_order, &d_seg, &d_has}) b->release();
    };
    struct Guard {
        std::function<void()> f;
        ~Guard() { f(); }
    } guard{cleanup};
    GH_CHECK(h, d_x.ensure((size_t)n * d * sizeof(float)));
    GH_CHECK(h, d_cen.ensure((size_t)k * d * sizeof(float)));
    GH_CHECK(h, d_cn.ensure((size_t)k * sizeof(float)));
    GH_CHECK(h, d_assign.ensure((size_t)n * sizeof(int)));
    GH_CHECK(h, d_dis.ensure((size_t)n * sizeof(float)));
    GH_CHECK(h, d_order.ensure((size_t)n * sizeof(int)));
    GH_CHECK(h, d_seg.ensure((size_t)(k + 1) * sizeof(int)));
    GH_CHECK(h, d_has.ensure((size_t)k * sizeof(float)));
    GH_CHECK(h, hipMemcpyAsync(d_x.p, x, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice, s));
    GH_CHECK(h, hipMemcpyAsync(d_cen.p, centroids, (size_t)k * d * sizeof(float), hipMemcpyHostToDevice, s));
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(h->dist_budget_bytes / ((size_t)k * sizeof(float)))));
    GH_CHECK(h, h->w_mat.ensure((size_t)chunk * k * sizeof(float)));
    std::vector<int> assign(n), order(n), seg(k + 1);
    std::vector<float> hassign(k), dis;
    const bool exact = n < 20;   // IndexFlatL2::search: the exact form below 20 queries (faiss:utils/distances.cpp:346)
    if (!exact && blas_form_not_restated(n, k, d)) h->blas_unrestated++;
    for (int it = 0; it < niter; it++) {
        // index.search(nx, x, 1, dis, assign)
        if (!exact) gh::launch_row_norms(s, d_cen.as<float>(), k, d, d_cn.as<float>());
        for (int64_t i0 = 0; i0 < n; i0 += chunk) {
            const int64_t nc = std::min(chunk, n - i0);
            if (exact) gh::launch_pairwise(s, true, d_x.as<float>() + i0 * d, (int)nc, d, d_cen.as<float>(), k, h->w_mat.as<float>(), k);
            else gh::launch_l2_gemmform(s, d_x.as<float>() + i0 * d, (int)nc, d, d_cen.as<float>(), k, nullptr, d_cn.as<float>(),
                                        h->w_mat.as<float>(), k, true);
            gh::launch_select_topk(s, true, h->w_mat.as<float>(), k, nullptr, k, k, (int)nc, 1, d_dis.as<float>() + i0,
                                   d_assign.as<int>() + i0);
        }
        GH_CHECK(h, hipGetLastError());
        GH_CHECK(h, hipMemcpyAsync(assign.data(), d_assign.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
        if (it == niter - 1 && objective) {
            dis.resize(n);
            GH_CHECK(h, hipMemcpyAsync(dis.data(), d_dis.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
        }
        GH_CHECK(h, hipStreamSynchronize(s));
        // the points of every cluster in ascending point order (a stable counting sort): compute_centroids adds them
        // to the cluster's accumulators in exactly that order (:160-189)
        std::fill(seg.begin(), seg.end(), 0);
        for (int64_t i = 0; i < n; i++) {
            if (assign[i] < 0 || assign[i] >= k) return fail(h, GAMMA_HIP_EDEVICE, "k-means: bad assignment");
            seg[assign[i] + 1]++;
        }
        for (int c = 0; c < k; c++) seg[c + 1] += seg[c];
        {
            std::vector<int> at(seg.begin(), seg.end() - 1);
            for (int64_t i = 0; i < n; i++) order[at[assign[i]]++] = (int)i;
        }
        GH_CHECK(h, hipMemcpyAsync(d_order.p, order.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
        GH_CHECK(h, hipMemcpyAsync(d_seg.p, seg.data(), (size_t)(k + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        gh::launch_centroid_update(s, d_x.as<float>(), d, d_order.as<int>(), d_seg.as<int>(), k, d_cen.as<float>(), d_has.as<float>());
        GH_CHECK(h, hipGetLastError());
        // clusters left empty are re-seeded from large ones on the host (rare: a download of the centroids only then)
        bool any_empty = false;
        for (int c = 0; c < k; c++) {
            hassign[c] = (float)(seg[c + 1] - seg[c]);
            any_empty |= seg[c + 1] == seg[c];
        }
        if (any_empty) {
            GH_CHECK(h, hipMemcpyAsync(centroids, d_cen.p, (size_t)k * d * sizeof(float), hipMemcpyDeviceToHost, s));
            GH_CHECK(h, hipStreamSynchronize(s));
            split_clusters(d, k, n, hassign.data(), centroids);
            GH_CHECK(h, hipMemcpyAsync(d_cen.p, centroids, (size_t)k * d * sizeof(float), hipMemcpyHostToDevice, s));
        }
    }
    GH_CHECK(h, hipMemcpyAsync(centroids, d_cen.p, (size_t)k * d * sizeof(float), hipMemcpyDeviceToHost, s));
    GH_CHECK(h, hipStreamSynchronize(s));
    if (objective && !dis.empty()) {
        float obj = 0;   // :478-481, float accumulation in point order
        for (int64_t j = 0; j < n; j++) obj += dis[j];
        *objective = obj;
    }
    return GAMMA_HIP_OK;
}

// IndexIVFPQ::train as GammaIVFPQIndex::Indexing configures it (index/impl/gamma_index_ivfpq.cc:172-185,272-354):
// train_q1 -- Clustering(d, nlist) with cp.niter = 10, seed 1234, max_points_per_centroid 256 -- then
// train_residual_o (faiss:IndexIVFPQ.cpp:67-106): at most 256 * 256 points (fvecs_maybe_subsample with pq.cp.seed =
// 1234), their residuals to the nearest coarse centroid (by_residual), and ProductQuantizer::train: one
// Clustering(dsub, 256, niter 25) per sub-quantizer.  coarse: nlist*d, pq: M*256*(d/M) fp32 host out.
int gamma_hip_ivfpq_train(gamma_hip_index* h, int d, int64_t n, const float* x, int nlist, int M, float* coarse, float* pq) {
    if (!h || d <= 0 || nlist <= 0 || M <= 0 || d % M != 0 || !x || !coarse || !pq) return GAMMA_HIP_EINVAL;
    int rc = gamma_hip_kmeans(h, d, n, x, nlist, 10, 1234, 256, coarse, nullptr);
    if (rc) return rc;
    const int64_t nmax = 256 * 256;
    std::vector<float> subset;
    const float* xs = x;
    int64_t ns = n;
    if (n > nmax) {
        std::vector<int> perm;
        rand_perm(perm, (size_t)n, 1234);
        subset.resize((size_t)nmax * d);
        for (int64_t i = 0; i < nmax; i++) memcpy(&subset[(size_t)i * d], x + (size_t)perm[i] * d, sizeof(float) * d);
        xs = subset.data();
        ns = nmax;
    }
    std::vector<int32_t> assign((size_t)ns);
    rc = gamma_hip_assign(h, d, ns, xs, nlist, coarse, assign.data(), nullptr);
    if (rc) return rc;
    const int dsub = d / M;
    std::vector<float> slice((size_t)ns * dsub);
    for (int m = 0; m < M; m++) {
        for (int64_t i = 0; i < ns; i++) {
            if (assign[i] < 0 || assign[i] >= nlist) return fail(h, GAMMA_HIP_EDEVICE, "training: bad assignment");
            const float* xi = xs + (size_t)i * d + m * dsub;
            const float* c = coarse + (size_t)assign[i] * d + m * dsub;
            for (int t = 0; t < dsub; t++) slice[(size_t)i * dsub + t] = xi[t] - c[t];
        }
        rc = gamma_hip_kmeans(h, dsub, ns, slice.data(), 256, 25, 1234, 256, pq + (size_t)m * 256 * dsub, nullptr);
        if (rc) return rc;
    }
    return GAMMA_HIP_OK;
}

}  // extern "C"
