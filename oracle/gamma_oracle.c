/*
 * gamma_oracle.c -- CPU restatement of the vearch/gamma retrieval hot path.
 * TEST INFRASTRUCTURE ONLY (see gamma_oracle.h).  Build: oracle/Makefile
 * (-O3 -mavx2 -mfma -ffp-contract=off: every fused multiply-add below is an explicit
 * fmaf() placed where the reference's -mfma build contracts one; nothing else fuses).
 *
 * Citations: paths relative to /root/reference; "faiss:" = faiss-1.7.1/faiss/ inside
 * third_party/faiss-1.7.1.tar.gz.
 */
#include "gamma_oracle.h"

#include <float.h>
#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ===================================================================================
 * fvec primitives.  The AVX2 build keeps 8 (or 4) independent lane accumulators, fuses
 * acc += a*b into one vfmadd per lane, then reduces with extractf128+add and two haddps:
 *   s[l] = acc[l+4] + acc[l];  result = (s0+s1) + (s2+s3)
 * faiss:utils/distances_simd.cpp:366-437 (AVX), :159-176 (norm, SSE), :207-345 (_ny).
 * =================================================================================== */
static inline float hsum4(const float s[4]) { return (s[0] + s[1]) + (s[2] + s[3]); }

float go_fvec_L2sqr(const float* x, const float* y, size_t d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= d; i += 8)
        for (int l = 0; l < 8; l++) {
            float t = x[i + l] - y[i + l];
            acc[l] = fmaf(t, t, acc[l]);
        }
    float s[4];
    for (int l = 0; l < 4; l++) s[l] = acc[l + 4] + acc[l];
    size_t rem = d - i;
    if (rem >= 4) {
        for (int l = 0; l < 4; l++) {
            float t = x[i + l] - y[i + l];
            s[l] = fmaf(t, t, s[l]);
        }
        i += 4;
        rem -= 4;
    }
    if (rem > 0) /* masked_read: lanes >= rem contribute fma(0,0,s)=s */
        for (size_t l = 0; l < rem; l++) {
            float t = x[i + l] - y[i + l];
            s[l] = fmaf(t, t, s[l]);
        }
    return hsum4(s);
}

float go_fvec_inner_product(const float* x, const float* y, size_t d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= d; i += 8)
        for (int l = 0; l < 8; l++) acc[l] = fmaf(x[i + l], y[i + l], acc[l]);
    float s[4];
    for (int l = 0; l < 4; l++) s[l] = acc[l + 4] + acc[l];
    size_t rem = d - i;
    if (rem >= 4) {
        for (int l = 0; l < 4; l++) s[l] = fmaf(x[i + l], y[i + l], s[l]);
        i += 4;
        rem -= 4;
    }
    if (rem > 0)
        for (size_t l = 0; l < rem; l++) s[l] = fmaf(x[i + l], y[i + l], s[l]);
    return hsum4(s);
}

float go_fvec_norm_L2sqr(const float* x, size_t d) {
    float acc[4] = {0, 0, 0, 0};
    size_t i = 0;
    for (; i + 4 <= d; i += 4)
        for (int l = 0; l < 4; l++) acc[l] = fmaf(x[i + l], x[i + l], acc[l]);
    /* masked tail: gcc 11 emits vmulps + vaddps here (the product is shared with the
     * zero-fill path), i.e. NOT fused -- restated as built */
    for (size_t l = 0; l < d - i; l++) acc[l] = acc[l] + x[i + l] * x[i + l];
    return hsum4(acc);
}

/* fvec_op_ny_D{1,2,4,8,12}: SSE, 4 lanes, then two haddps.  For "accu = op0; accu +=
 * op1; accu += op2" the reference build (gcc 11.4 -O3 -mfma) rounds the SECOND product
 * and fuses the first and third: a = p1; a = fma(x0,y0,a); a = fma(x2,y2,a)
 * (objdump of distances_simd.o: vmulps 0x10(%rdx) ; vfmadd231ps -0x20(%rdx)). */
void go_fvec_inner_products_ny(float* dis, const float* x, const float* y, size_t d, size_t ny) {
    size_t i;
    switch (d) {
        case 1:
            for (i = 0; i < ny; i++) dis[i] = x[0] * y[i];
            return;
        case 2:
            for (i = 0; i + 1 < ny; i += 2) { /* vmulps + haddps: both products rounded */
                dis[i] = x[0] * y[2 * i] + x[1] * y[2 * i + 1];
                dis[i + 1] = x[0] * y[2 * i + 2] + x[1] * y[2 * i + 3];
            }
            if (i < ny) dis[i] = fmaf(x[0], y[2 * i], x[1] * y[2 * i + 1]); /* scalar odd tail */
            return;
        case 4:
            for (i = 0; i < ny; i++, y += 4) {
                float a[4];
                for (int l = 0; l < 4; l++) a[l] = x[l] * y[l];
                dis[i] = hsum4(a);
            }
            return;
        case 8:
            for (i = 0; i < ny; i++, y += 8) {
                float a[4];
                for (int l = 0; l < 4; l++) a[l] = x[4 + l] * y[4 + l];
                for (int l = 0; l < 4; l++) a[l] = fmaf(x[l], y[l], a[l]);
                dis[i] = hsum4(a);
            }
            return;
        case 12:
            for (i = 0; i < ny; i++, y += 12) {
                float a[4];
                for (int l = 0; l < 4; l++) a[l] = x[4 + l] * y[4 + l];
                for (int l = 0; l < 4; l++) a[l] = fmaf(x[l], y[l], a[l]);
                for (int l = 0; l < 4; l++) a[l] = fmaf(x[8 + l], y[8 + l], a[l]);
                dis[i] = hsum4(a);
            }
            return;
        default: /* fvec_inner_products_ny_ref -> fvec_inner_product per row */
            for (i = 0; i < ny; i++, y += d) dis[i] = go_fvec_inner_product(x, y, d);
            return;
    }
}

void go_fvec_L2sqr_ny(float* dis, const float* x, const float* y, size_t d, size_t ny) {
    size_t i;
    switch (d) {
        case 1:
            for (i = 0; i < ny; i++) {
                float t = x[0] - y[i];
                dis[i] = t * t;
            }
            return;
        case 2:
            for (i = 0; i + 1 < ny; i += 2) {
                float t0 = x[0] - y[2 * i], t1 = x[1] - y[2 * i + 1];
                float t2 = x[0] - y[2 * i + 2], t3 = x[1] - y[2 * i + 3];
                dis[i] = t0 * t0 + t1 * t1;
                dis[i + 1] = t2 * t2 + t3 * t3;
            }
            if (i < ny) {
                float t0 = x[0] - y[2 * i], t1 = x[1] - y[2 * i + 1];
                dis[i] = fmaf(t0, t0, t1 * t1);
            }
            return;
        case 4:
            for (i = 0; i < ny; i++, y += 4) {
                float a[4];
                for (int l = 0; l < 4; l++) {
                    float t = x[l] - y[l];
                    a[l] = t * t;
                }
                dis[i] = hsum4(a);
            }
            return;
        case 8:
        case 12:
            for (i = 0; i < ny; i++, y += d) {
                float a[4];
                for (int l = 0; l < 4; l++) { /* second block: plain square */
                    float t = x[4 + l] - y[4 + l];
                    a[l] = t * t;
                }
                for (int l = 0; l < 4; l++) { /* first block fused on top */
                    float t = x[l] - y[l];
                    a[l] = fmaf(t, t, a[l]);
                }
                if (d == 12)
                    for (int l = 0; l < 4; l++) {
                        float t = x[8 + l] - y[8 + l];
                        a[l] = fmaf(t, t, a[l]);
                    }
                dis[i] = hsum4(a);
            }
            return;
        default: /* fvec_L2sqr_ny_ref -> fvec_L2sqr per row */
            for (i = 0; i < ny; i++, y += d) dis[i] = go_fvec_L2sqr(x, y, d);
            return;
    }
}

/* c = a + bf*b, fused (faiss:utils/distances_simd.cpp:713-750, both variants contract) */
void go_fvec_madd(size_t n, const float* a, float bf, const float* b, float* c) {
    for (size_t i = 0; i < n; i++) c[i] = fmaf(bf, b[i], a[i]);
}

/* ===================================================================================
 * Binary heap, 1-based sift exactly as faiss:utils/Heap.h.  keep_smallest=1 is CMax
 * (cmp(a,b) = a > b, neutral FLT_MAX), 0 is CMin (a < b, neutral -FLT_MAX).
 * =================================================================================== */
#define GO_MIN_K_RESERVOIR 100 /* faiss:utils/distances.cpp:305 distance_compute_min_k_reservoir */
static inline int hcmp(int ks, float a, float b) { return ks ? (a > b) : (a < b); }
static inline float hneutral(int ks) { return ks ? FLT_MAX : -FLT_MAX; }

void go_heap_heapify(int ks, size_t k, float* val, int64_t* ids) {
    for (size_t i = 0; i < k; i++) {
        val[i] = hneutral(ks);
        ids[i] = -1;
    }
}

static inline void sift_down(int ks, size_t k, float* bh_val, int64_t* bh_ids, float val,
                             int64_t id) {
    /* shared body of heap_pop (val = last element) and heap_replace_top */
    bh_val--;
    bh_ids--;
    size_t i = 1, i1, i2;
    for (;;) {
        i1 = i << 1;
        i2 = i1 + 1;
        if (i1 > k) break;
        if (i2 == k + 1 || hcmp(ks, bh_val[i1], bh_val[i2])) {
            if (hcmp(ks, val, bh_val[i1])) break;
            bh_val[i] = bh_val[i1];
            bh_ids[i] = bh_ids[i1];
            i = i1;
        } else {
            if (hcmp(ks, val, bh_val[i2])) break;
            bh_val[i] = bh_val[i2];
            bh_ids[i] = bh_ids[i2];
            i = i2;
        }
    }
    bh_val[i] = val;
    bh_ids[i] = id;
}

void go_heap_replace_top(int ks, size_t k, float* val, int64_t* ids, float v, int64_t id) {
    sift_down(ks, k, val, ids, v, id);
}

void go_heap_pop(int ks, size_t k, float* val, int64_t* ids) {
    /* faiss:utils/Heap.h:46-72: sift the last element down from the root; element k-1 is
     * then undefined (left in place). */
    sift_down(ks, k, val, ids, val[k - 1], ids[k - 1]);
}

void go_heap_push(int ks, size_t k, float* bh_val, int64_t* bh_ids, float val, int64_t id) {
    bh_val--;
    bh_ids--;
    size_t i = k, i_father;
    while (i > 1) {
        i_father = i >> 1;
        if (!hcmp(ks, val, bh_val[i_father])) break;
        bh_val[i] = bh_val[i_father];
        bh_ids[i] = bh_ids[i_father];
        i = i_father;
    }
    bh_val[i] = val;
    bh_ids[i] = id;
}

size_t go_heap_reorder(int ks, size_t k, float* bh_val, int64_t* bh_ids) {
    size_t i, ii;
    for (i = 0, ii = 0; i < k; i++) {
        float val = bh_val[0];
        int64_t id = bh_ids[0];
        go_heap_pop(ks, k - i, bh_val, bh_ids);
        bh_val[k - ii - 1] = val;
        bh_ids[k - ii - 1] = id;
        if (id != -1) ii++;
    }
    size_t nel = ii;
    memmove(bh_val, bh_val + k - ii, ii * sizeof(*bh_val));
    memmove(bh_ids, bh_ids + k - ii, ii * sizeof(*bh_ids));
    for (; ii < k; ii++) {
        bh_val[ii] = hneutral(ks);
        bh_ids[ii] = -1;
    }
    return nel;
}

void go_heap_stream(int ks, size_t k, size_t n, const float* vals, const int64_t* ids,
                    float* heap_vals, int64_t* heap_ids, float* sorted_vals,
                    int64_t* sorted_ids) {
    go_heap_heapify(ks, k, heap_vals, heap_ids);
    for (size_t i = 0; i < n; i++)
        if (hcmp(ks, heap_vals[0], vals[i]))
            go_heap_replace_top(ks, k, heap_vals, heap_ids, vals[i], ids[i]);
    memcpy(sorted_vals, heap_vals, k * sizeof(float));
    memcpy(sorted_ids, heap_ids, k * sizeof(int64_t));
    go_heap_reorder(ks, k, sorted_vals, sorted_ids);
}

void go_heap_pop_push_stream(int ks, size_t k, size_t n, const float* vals, const int64_t* ids,
                             float* sorted_vals, int64_t* sorted_ids) {
    go_heap_heapify(ks, k, sorted_vals, sorted_ids);
    for (size_t i = 0; i < n; i++)
        if (hcmp(ks, sorted_vals[0], vals[i])) {
            go_heap_pop(ks, k, sorted_vals, sorted_ids);
            go_heap_push(ks, k, sorted_vals, sorted_ids, vals[i], ids[i]);
        }
    go_heap_reorder(ks, k, sorted_vals, sorted_ids);
}

/* ===================================================================================
 * ReservoirTopN (faiss:impl/ResultHandler.h:131-187) and the partition it shrinks with
 * (partition_fuzzy_median3, faiss:utils/partitioning.cpp:29-215; float keys never take the
 * SIMD branch, :746-764).  knn_L2sqr / knn_inner_product collect through it from k = 100 on
 * (distance_compute_min_k_reservoir, faiss:utils/distances.cpp:341-358): same k best as the heap
 * wherever the keys are distinct; inside exact ties WHICH entries stay and in what order they come
 * out is this structure's doing.  ks = 1: keep the smallest (CMax), 0: the largest (CMin).
 * =================================================================================== */
typedef struct {
    float* vals;
    int64_t* ids;
    size_t i, n, capacity;
    float threshold;
} go_reservoir;

size_t go_reservoir_capacity(size_t k) { return (2 * k + 15) & ~(size_t)15; } /* :211 */

static float res_median3(float a, float b, float c) { /* partitioning.cpp:29-40: plain >, whatever the order kept */
    if (a > b) {
        float t = a;
        a = b;
        b = t;
    }
    if (c > b) return b;
    if (c > a) return c;
    return a;
}

static float res_sample_threshold_median3(int ks, const float* vals, size_t n, float thresh_inf, float thresh_sup) {
    const size_t big_prime = 6700417; /* :49 */
    float val3[3];
    int vi = 0;
    for (size_t i = 0; i < n; i++) {
        float v = vals[(i * big_prime) % n];
        if (hcmp(ks, v, thresh_inf) && hcmp(ks, thresh_sup, v)) {
            val3[vi++] = v;
            if (vi == 3) break;
        }
    }
    if (vi == 3) return res_median3(val3[0], val3[1], val3[2]);
    if (vi != 0) return val3[0];
    return thresh_inf;
}

static float res_partition_fuzzy(int ks, float* vals, int64_t* ids, size_t n, size_t q_min, size_t q_max, size_t* q_out) {
    /* (q_min == 0 and q_max >= n cannot happen for a reservoir: k >= 100, q_max = (capacity + k) / 2 < capacity) */
    float thresh_inf = hneutral(!ks), thresh_sup = hneutral(ks);
    float thresh = res_median3(vals[0], vals[n / 2], vals[n - 1]);
    size_t n_eq = 0, n_lt = 0, q = 0;
    for (int it = 0; it < 200; it++) {
        n_lt = n_eq = 0; /* count_lt_and_eq :75-92 */
        for (size_t i = 0; i < n; i++) {
            if (hcmp(ks, thresh, vals[i])) n_lt++;
            else if (vals[i] == thresh) n_eq++;
        }
        if (n_lt <= q_min) {
            if (n_lt + n_eq >= q_min) {
                q = q_min;
                break;
            }
            thresh_inf = thresh;
        } else if (n_lt <= q_max) {
            q = n_lt;
            break;
        } else {
            thresh_sup = thresh;
        }
        float new_thresh = res_sample_threshold_median3(ks, vals, n, thresh_inf, thresh_sup);
        if (new_thresh == thresh_inf) break; /* nothing between the bounds */
        thresh = new_thresh;
    }
    int64_t n_eq_1 = (int64_t)q - (int64_t)n_lt;
    if (n_eq_1 < 0) { /* more than q entries at the lower bound (:196-200) */
        q = q_min;
        thresh = nextafterf(thresh, ks ? -HUGE_VALF : HUGE_VALF); /* C::Crev::nextafter */
        n_eq_1 = (int64_t)q;
    }
    size_t wp = 0; /* compress_array :94-116: stable, the first n_eq_1 entries equal to thresh stay */
    size_t left = (size_t)n_eq_1;
    for (size_t i = 0; i < n; i++) {
        if (hcmp(ks, thresh, vals[i])) {
            vals[wp] = vals[i];
            ids[wp] = ids[i];
            wp++;
        } else if (left > 0 && vals[i] == thresh) {
            vals[wp] = vals[i];
            ids[wp] = ids[i];
            wp++;
            left--;
        }
    }
    *q_out = q;
    (void)wp;
    return thresh;
}

static void res_begin(go_reservoir* r, int ks, size_t n, size_t capacity, float* vals, int64_t* ids) {
    r->vals = vals;
    r->ids = ids;
    r->i = 0;
    r->n = n;
    r->capacity = capacity;
    r->threshold = hneutral(ks);
}
static inline void res_add(go_reservoir* r, int ks, float val, int64_t id) { /* :152-161 */
    if (hcmp(ks, r->threshold, val)) {
        if (r->i == r->capacity) /* shrink_fuzzy :165-170 */
            r->threshold = res_partition_fuzzy(ks, r->vals, r->ids, r->capacity, r->n, (r->capacity + r->n) / 2, &r->i);
        r->vals[r->i] = val;
        r->ids[r->i] = id;
        r->i++;
    }
}
static void res_to_result(const go_reservoir* r, int ks, float* heap_dis, int64_t* heap_ids) { /* :172-186 */
    const size_t m = r->i < r->n ? r->i : r->n;
    for (size_t j = 0; j < m; j++) go_heap_push(ks, j + 1, heap_dis, heap_ids, r->vals[j], r->ids[j]);
    if (r->i < r->n) {
        go_heap_reorder(ks, r->i, heap_dis, heap_ids);
        go_heap_heapify(ks, r->n - r->i, heap_dis + r->i, heap_ids + r->i);
    } else {
        for (size_t j = r->n; j < r->i; j++) /* heap_addn, faiss:utils/Heap.h:247-260 */
            if (hcmp(ks, heap_dis[0], r->vals[j])) go_heap_replace_top(ks, r->n, heap_dis, heap_ids, r->vals[j], r->ids[j]);
        go_heap_reorder(ks, r->n, heap_dis, heap_ids);
    }
}

/* a stream of (value, id) through the reservoir: what the device's replay of a tied row is checked against */
void go_reservoir_stream(int ks, size_t k, size_t n, const float* vals, const int64_t* ids, float* sorted_vals,
                         int64_t* sorted_ids) {
    const size_t cap = go_reservoir_capacity(k);
    float* rv = (float*)malloc(sizeof(float) * cap);
    int64_t* ri = (int64_t*)malloc(sizeof(int64_t) * cap);
    go_reservoir r;
    res_begin(&r, ks, k, cap, rv, ri);
    for (size_t i = 0; i < n; i++) res_add(&r, ks, vals[i], ids ? ids[i] : (int64_t)i);
    res_to_result(&r, ks, sorted_vals, sorted_ids);
    free(rv);
    free(ri);
}

/* ===================================================================================
 * Brute-force kNN: IndexFlatL2::search -> knn_L2sqr (faiss:IndexFlat.cpp:35-55,
 * faiss:utils/distances.cpp:334-360).  HeapResultHandler below 100 results, ReservoirResultHandler
 * from there on (:341-358; the restatement above).
 * mode 0 = exhaustive_L2sqr_seq (:130-155); mode 1 = exhaustive_L2sqr_blas (:215-296)
 * with sgemm_ restated as a k-sequential single-accumulator fmaf chain.
 * =================================================================================== */
/* How the compiled reference's sgemm_ (MKL, /opt/conda/lib/libmkl_rt.so: the BLAS oracle/Makefile.ref links, as the
 * survey's build does) sums the K dimension of exhaustive_L2sqr_blas's ip_block = x . y^T, measured in this container
 * (tests/test_oracle_vs_ref.py::test_gemm_form_is_the_compiled_sgemm pins it):
 *   K <= 384:                   ONE k-ascending fma chain per output element;
 *   384 < K <= 768, K % 8 == 0: TWO chains, [0, K/2) and [K/2, K), each from zero, added once (C = P1; C += P2);
 *   otherwise (K > 768, or an odd split): MKL's blocking is not restated -- one chain, 1-ulp-level differences.
 * Shape-dependent corners that are NOT restated either: K = 384 exactly with a database block (ny mod 1024) of 9..512
 * rows is already split in two; remainder blocks of a few rows (nx mod 4096 or ny mod 1024 below 8) take a
 * different kernel.
 * Returns the split point, 0 = none.  The device's GEMM-form kernels use the same rule (kernels.hip gemm_k_split). */
size_t go_gemm_k_split(size_t d) { return (d > 384 && d <= 768 && d % 8 == 0) ? d / 2 : 0; }

__attribute__((unused)) static float dot_seq(const float* x, const float* y, size_t d) {
    float ip = 0.f;
    for (size_t t = 0; t < d; t++) ip = fmaf(x[t], y[t], ip);
    return ip;
}

void go_knn_L2sqr(int mode, const float* x, const float* y, size_t d, size_t nx, size_t ny,
                  size_t k, float* D, int64_t* I) {
    float* yn = NULL;
    float* yT = NULL; /* mode 1: centroids in blocks of 8, transposed: yT[b][t][8] */
    const size_t nb = (ny + 7) / 8;
    if (mode == 1) {
        yn = (float*)malloc(sizeof(float) * (ny ? ny : 1));
#pragma omp parallel for
        for (int64_t j = 0; j < (int64_t)ny; j++) yn[j] = go_fvec_norm_L2sqr(y + j * d, d);
        /* The GEMM form's inner product is ONE k-ascending fmaf chain per (query, centroid) -- what an
         * fp32 MFMA computes and the stand-in for sgemm_.  Eight centroids run side by side in
         * the eight AVX lanes, each lane its own chain: same bits as dot_seq, 8x fewer dependent steps. */
        yT = (float*)aligned_alloc(32, sizeof(float) * 8 * d * (nb ? nb : 1));
#pragma omp parallel for
        for (int64_t b = 0; b < (int64_t)nb; b++)
            for (size_t t = 0; t < d; t++)
                for (int l = 0; l < 8; l++) {
                    size_t j = (size_t)b * 8 + l;
                    yT[((size_t)b * d + t) * 8 + l] = j < ny ? y[j * d + t] : 0.f;
                }
    }
#pragma omp parallel for schedule(dynamic)
    for (int64_t i = 0; i < (int64_t)nx; i++) {
        const float* xi = x + i * d;
        float* hd = D + i * k;
        int64_t* hi = I + i * k;
        const int use_res = k >= GO_MIN_K_RESERVOIR;
        go_reservoir rs;
        float* rv = NULL;
        int64_t* ri = NULL;
        if (use_res) {
            const size_t cap = go_reservoir_capacity(k);
            rv = (float*)malloc(sizeof(float) * cap);
            ri = (int64_t*)malloc(sizeof(int64_t) * cap);
            res_begin(&rs, 1, k, cap, rv, ri);
        } else {
            go_heap_heapify(1, k, hd, hi);
        }
#define GO_KNN_ADD(dis, j)                                                         \
    do {                                                                           \
        if (use_res) res_add(&rs, 1, (dis), (int64_t)(j));                         \
        else if (hd[0] > (dis)) go_heap_replace_top(1, k, hd, hi, (dis), (int64_t)(j)); \
    } while (0)
        if (mode == 0) {
            for (size_t j = 0; j < ny; j++) {
                float dis = go_fvec_L2sqr(xi, y + j * d, d);
                GO_KNN_ADD(dis, j);
            }
        } else {
            float xn = go_fvec_norm_L2sqr(xi, d);
            for (size_t b = 0; b < nb; b++) {
                const float* yb = yT + b * d * 8;
                __m256 acc = _mm256_setzero_ps();
                const size_t ksp = go_gemm_k_split(d);
                for (size_t t = 0; t < (ksp ? ksp : d); t++)
                    acc = _mm256_fmadd_ps(_mm256_broadcast_ss(xi + t), _mm256_load_ps(yb + t * 8), acc);
                if (ksp) { /* second K block: its own chain from zero, added to the first (sgemm_'s C += A2 B2) */
                    __m256 acc2 = _mm256_setzero_ps();
                    for (size_t t = ksp; t < d; t++)
                        acc2 = _mm256_fmadd_ps(_mm256_broadcast_ss(xi + t), _mm256_load_ps(yb + t * 8), acc2);
                    acc = _mm256_add_ps(acc, acc2);
                }
                float ip[8];
                _mm256_storeu_ps(ip, acc);
                for (int l = 0; l < 8 && b * 8 + l < ny; l++) {
                    size_t j = b * 8 + l;
                    float dis = (xn + yn[j]) - 2 * ip[l];
                    if (dis < 0) dis = 0;
                    GO_KNN_ADD(dis, j);
                }
            }
        }
#undef GO_KNN_ADD
        if (use_res) {
            res_to_result(&rs, 1, hd, hi);
            free(rv);
            free(ri);
        } else {
            go_heap_reorder(1, k, hd, hi);
        }
    }
    free(yn);
    free(yT);
}

void go_knn_inner_product(const float* x, const float* y, size_t d, size_t nx, size_t ny,
                          size_t k, float* D, int64_t* I) {
#pragma omp parallel for schedule(dynamic)
    for (int64_t i = 0; i < (int64_t)nx; i++) {
        const float* xi = x + i * d;
        float* hd = D + i * k;
        int64_t* hi = I + i * k;
        if (k >= GO_MIN_K_RESERVOIR) { /* knn_inner_product, faiss:utils/distances.cpp:307-332 */
            const size_t cap = go_reservoir_capacity(k);
            float* rv = (float*)malloc(sizeof(float) * cap);
            int64_t* ri = (int64_t*)malloc(sizeof(int64_t) * cap);
            go_reservoir rs;
            res_begin(&rs, 0, k, cap, rv, ri);
            for (size_t j = 0; j < ny; j++) res_add(&rs, 0, go_fvec_inner_product(xi, y + j * d, d), (int64_t)j);
            res_to_result(&rs, 0, hd, hi);
            free(rv);
            free(ri);
            continue;
        }
        go_heap_heapify(0, k, hd, hi);
        for (size_t j = 0; j < ny; j++) {
            float ip = go_fvec_inner_product(xi, y + j * d, d);
            if (hd[0] < ip) go_heap_replace_top(0, k, hd, hi, ip, (int64_t)j);
        }
        go_heap_reorder(0, k, hd, hi);
    }
}

/* ===================================================================================
 * Product quantizer pieces
 * =================================================================================== */
/* ProductQuantizer::compute_inner_prod_table, faiss:impl/ProductQuantizer.cpp:518-531.
 * centroid layout centroids[m][j][dsub]; table layout tab[m*ksub + j]. */
void go_pq_inner_prod_table(const float* pqc, int M, int ksub, int dsub, const float* x,
                            float* table) {
    for (int m = 0; m < M; m++)
        go_fvec_inner_products_ny(table + (size_t)m * ksub, x + (size_t)m * dsub,
                                  pqc + (size_t)m * ksub * dsub, dsub, ksub);
}

/* compute_code, faiss:impl/ProductQuantizer.cpp:321-348 (argmin, strict <, init 1e20).
 * The reference switches to BLAS distance tables for dsub >= 16 (:484-500); this
 * restatement keeps the exact form for every dsub (documented in DESIGN.md). */
void go_pq_compute_codes(const float* pqc, int M, int ksub, int dsub, const float* x,
                         uint8_t* codes, size_t n) {
    int d = M * dsub;
#pragma omp parallel
    {
        float* dist = (float*)malloc(sizeof(float) * ksub);
#pragma omp for
        for (int64_t i = 0; i < (int64_t)n; i++) {
            for (int m = 0; m < M; m++) {
                go_fvec_L2sqr_ny(dist, x + i * d + (size_t)m * dsub,
                                 pqc + (size_t)m * ksub * dsub, dsub, ksub);
                float mindis = 1e20f;
                int best = 0;
                for (int j = 0; j < ksub; j++)
                    if (dist[j] < mindis) {
                        mindis = dist[j];
                        best = j;
                    }
                codes[i * M + m] = (uint8_t)best;
            }
        }
        free(dist);
    }
}

/* initialize_IVFPQ_precomputed_table, use_precomputed_table == 1 branch
 * (faiss:IndexIVFPQ.cpp:461-479): T2[l][m][j] = ||c_mj||^2 + 2 <centroid_l,m , c_mj> */
void go_ivfpq_precompute_table(const float* cc, int nlist, int d, const float* pqc, int M,
                               int ksub, float* table) {
    int dsub = d / M;
    size_t tsz = (size_t)M * ksub;
    float* r_norms = (float*)malloc(sizeof(float) * tsz);
    for (int m = 0; m < M; m++)
        for (int j = 0; j < ksub; j++)
            r_norms[(size_t)m * ksub + j] =
                    go_fvec_norm_L2sqr(pqc + ((size_t)m * ksub + j) * dsub, dsub);
#pragma omp parallel for
    for (int64_t l = 0; l < nlist; l++) {
        float* tab = table + l * tsz;
        go_pq_inner_prod_table(pqc, M, ksub, dsub, cc + l * d, tab);
        go_fvec_madd(tsz, r_norms, 2.0f, tab, tab);
    }
    free(r_norms);
}

/* ===================================================================================
 * Validity predicates (common/gamma_common_data.h:95-108, table/range_query_result.h)
 * =================================================================================== */
static inline int bm_test(const uint8_t* bm, int64_t id) {
    return (bm[id >> 3] >> (id & 7)) & 1; /* util/bitmap.cc:25-27 */
}

static inline int range_has(const go_range_filter* r, int doc) {
    /* RangeQueryResult::Has, table/range_query_result.h:53-67 */
    if (r->b_not_in) {
        if (doc < r->min_doc || doc > r->max_doc) return 1;
        return !bm_test(r->bitmap, doc - r->min_aligned);
    }
    if (doc < r->min_doc || doc > r->max_doc) return 0;
    return bm_test(r->bitmap, doc - r->min_aligned);
}

static inline int ctx_is_valid(const go_search_ctx* c, int64_t vid) {
    if (!c) return 1;
    int docid = (c->vid2docid && vid >= 0 && vid < c->n_vid2docid) ? c->vid2docid[vid] : (int)vid; /* VIDMgr::VID2DocID */
    if (c->has_range) {
        /* MultiRangeQueryResults::Has, :169-179: empty => false */
        if (c->n_range == 0) return 0;
        for (int i = 0; i < c->n_range; i++)
            if (!range_has(&c->range[i], docid)) return 0;
    }
    if (c->docids_bitmap && docid >= 0 && (int64_t)docid < c->docids_bitmap_bits &&
        bm_test(c->docids_bitmap, docid)) /* BitmapManager::Test, util/bitmap_manager.cc:187-192 */
        return 0;
    return 1;
}

static inline int ctx_score_valid(const go_search_ctx* c, float s) {
    if (!c) return 1;
    return (s <= c->max_score) && (s >= c->min_score);
}

/* ===================================================================================
 * Realtime inverted lists (realtime/realtime_mem_data.cc)
 * =================================================================================== */
#define GO_DEL_MASK ((int64_t)(1ULL << 63)) /* kDelIdxMask, realtime_mem_data.h:26 */
#define GO_RECOVER_MASK (~GO_DEL_MASK)
#define GO_PI 3.14159265                     /* realtime_mem_data.h:24 */

typedef struct {
    int64_t* ids;
    uint8_t* codes;
    int64_t size;      /* retrieve_idx_pos_ */
    int32_t capacity;  /* cur_bucket_keys_ */
    uint8_t extend_time;
    int32_t deleted;
} go_bucket;

struct go_ivfpq {
    int d, nlist, M, nbits, ksub, dsub, code_size, metric;
    int use_precomputed_table;
    float* cc;    /* coarse centroids nlist*d */
    float* pqc;   /* PQ centroids M*ksub*dsub */
    float* table; /* T2 nlist*M*ksub */
    go_bucket* b;
    int bucket_init, bucket_max;
    int64_t* vid_pos; /* vid_bucket_no_pos_: (bucket<<32)|pos or -1 */
    int64_t nids;
    int64_t compacted_num;
    const float* raw;
    int64_t nraw;
    int64_t indexed_vec_count;
    const uint8_t* docids_bitmap; /* borrowed; RTInvertBucketData::docids_bitmap_ */
    int64_t docids_bits;
    const int32_t* vid2docid; /* borrowed; vid_mgr_->VID2DocID, NULL = identity */
    int64_t n_vid2docid;
};

go_ivfpq* go_ivfpq_new(int d, int nlist, int M, int nbits, int metric, int bucket_init_size,
                       int bucket_max_size) {
    if (nbits != 8 || d % M != 0) return NULL;
    go_ivfpq* ix = (go_ivfpq*)calloc(1, sizeof(go_ivfpq));
    ix->d = d;
    ix->nlist = nlist;
    ix->M = M;
    ix->nbits = nbits;
    ix->ksub = 1 << nbits;
    ix->dsub = d / M;
    ix->code_size = M; /* nbits == 8 */
    ix->metric = metric;
    ix->use_precomputed_table = 0; /* gamma_index_ivfpq.cc:180; becomes 1 after train */
    ix->bucket_init = bucket_init_size > 0 ? bucket_init_size : 1000;
    ix->bucket_max = bucket_max_size > 0 ? bucket_max_size : 1280000;
    ix->b = (go_bucket*)calloc(nlist, sizeof(go_bucket));
    for (int i = 0; i < nlist; i++) { /* RTInvertBucketData::Init, :57-96 */
        ix->b[i].ids = (int64_t*)malloc(sizeof(int64_t) * ix->bucket_init);
        ix->b[i].codes = (uint8_t*)malloc((size_t)ix->bucket_init * ix->code_size);
        ix->b[i].capacity = ix->bucket_init;
    }
    ix->nids = (int64_t)nlist * ix->bucket_init;
    ix->vid_pos = (int64_t*)malloc(sizeof(int64_t) * ix->nids);
    for (int64_t i = 0; i < ix->nids; i++) ix->vid_pos[i] = -1;
    return ix;
}

void go_ivfpq_free(go_ivfpq* ix) {
    if (!ix) return;
    for (int i = 0; i < ix->nlist; i++) {
        free(ix->b[i].ids);
        free(ix->b[i].codes);
    }
    free(ix->b);
    free(ix->vid_pos);
    free(ix->cc);
    free(ix->pqc);
    free(ix->table);
    free(ix);
}

/* faiss's process-wide limit `precomputed_table_max_bytes` (faiss:IndexIVFPQ.cpp:379, 2 GiB): a table
 * nlist * M * ksub * 4 bytes above it is NOT built and the index stays in table mode 0
 * (faiss:IndexIVFPQ.cpp:441-449).  Settable like the library's extern so a test can reach the branch at a small shape. */
static size_t g_precomputed_table_max_bytes = ((size_t)1) << 31;
void go_set_precomputed_table_max_bytes(size_t bytes) { g_precomputed_table_max_bytes = bytes; }
size_t go_get_precomputed_table_max_bytes(void) { return g_precomputed_table_max_bytes; }

void go_ivfpq_set_trained(go_ivfpq* ix, const float* cc, const float* pqc, const float* table) {
    size_t ncc = (size_t)ix->nlist * ix->d, npq = (size_t)ix->M * ix->ksub * ix->dsub;
    size_t nt = (size_t)ix->nlist * ix->M * ix->ksub;
    free(ix->cc);
    free(ix->pqc);
    free(ix->table);
    ix->table = NULL;
    ix->cc = (float*)malloc(ncc * sizeof(float));
    ix->pqc = (float*)malloc(npq * sizeof(float));
    memcpy(ix->cc, cc, ncc * sizeof(float));
    memcpy(ix->pqc, pqc, npq * sizeof(float));
    /* train_residual_o -> precompute_table (faiss:IndexIVFPQ.cpp:132-135) and Load
     * (gamma_index_ivfpq.cc:1033-1034) start from use_precomputed_table == 0 and let
     * initialize_IVFPQ_precomputed_table choose (faiss:IndexIVFPQ.cpp:426-452; Gamma's quantizer is always an
     * IndexFlatL2, gamma_index_ivfpq.cc:147, so the inner-product exit at :427-434 is never taken):
     * table_size > precomputed_table_max_bytes -> no table, the mode stays 0; otherwise mode 1. */
    if (nt * sizeof(float) > g_precomputed_table_max_bytes) {
        ix->use_precomputed_table = 0;
        return;
    }
    ix->table = (float*)malloc(nt * sizeof(float));
    if (table)
        memcpy(ix->table, table, nt * sizeof(float));
    else
        go_ivfpq_precompute_table(ix->cc, ix->nlist, ix->d, ix->pqc, ix->M, ix->ksub, ix->table);
    ix->use_precomputed_table = 1;
}

int go_ivfpq_use_precomputed_table(const go_ivfpq* ix) { return ix->use_precomputed_table; }

const float* go_ivfpq_table(go_ivfpq* ix) { return ix->table; }

void go_ivfpq_set_docids_bitmap(go_ivfpq* ix, const uint8_t* bm, int64_t nbits) {
    ix->docids_bitmap = bm;
    ix->docids_bits = nbits;
}

void go_ivfpq_set_vid2docid(go_ivfpq* ix, const int32_t* map, int64_t n) {
    ix->vid2docid = map;
    ix->n_vid2docid = n;
}
static inline int64_t ix_docid(const go_ivfpq* ix, int64_t vid) {
    return (ix->vid2docid && vid >= 0 && vid < ix->n_vid2docid) ? ix->vid2docid[vid] : vid;
}

void go_ivfpq_set_raw(go_ivfpq* ix, const float* raw, int64_t n) {
    ix->raw = raw;
    ix->nraw = n;
}

static double extend_coefficient(uint8_t t) { return 1.1 + GO_PI / 2 - atan((double)t); }

/* RTInvertBucketData::ExtendBucketMem, :152-188 */
static int bucket_extend(go_ivfpq* ix, int bno, int increment) {
    go_bucket* b = &ix->b[bno];
    int least = (int)b->size + increment;
    double coefficient = extend_coefficient(++b->extend_time);
    int extend_size = (int)(b->capacity * coefficient);
    while (extend_size < least) {
        coefficient = extend_coefficient(++b->extend_time);
        extend_size = (int)(extend_size * coefficient);
    }
    uint8_t* nc = (uint8_t*)malloc((size_t)extend_size * ix->code_size);
    int64_t* ni = (int64_t*)malloc(sizeof(int64_t) * extend_size);
    if (!nc || !ni) return 0;
    memcpy(nc, b->codes, (size_t)b->size * ix->code_size);
    memcpy(ni, b->ids, sizeof(int64_t) * b->size);
    free(b->codes);
    free(b->ids);
    b->codes = nc;
    b->ids = ni;
    b->capacity = extend_size;
    return 1;
}

/* RealTimeMemData::ExtendBucketIfNeed, :383-421 */
static int bucket_extend_if_need(go_ivfpq* ix, int bno, size_t keys_size) {
    go_bucket* b = &ix->b[bno];
    if ((size_t)b->size + (int)keys_size <= (size_t)b->capacity) return 0;
    if (b->capacity * 2 >= ix->bucket_max) return -1;
    if (!bucket_extend(ix, bno, (int)keys_size)) return -2;
    return 0;
}

static void ids_extend(go_ivfpq* ix) { /* ExtendIDs, :203-218 */
    int64_t n2 = ix->nids * 2;
    int64_t* na = (int64_t*)malloc(sizeof(int64_t) * n2);
    memcpy(na, ix->vid_pos, sizeof(int64_t) * ix->nids);
    for (int64_t i = ix->nids; i < n2; i++) na[i] = -1;
    free(ix->vid_pos);
    ix->vid_pos = na;
    ix->nids = n2;
}

/* RealTimeMemData::AddKeys, :264-303 (delete-bitmap bump of deleted_nums_ needs the
 * bitmap; callers that track deletes pass it via go_ivfpq_delete) */
int go_ivfpq_add_keys(go_ivfpq* ix, int list_no, int n, const int64_t* keys, const uint8_t* codes) {
    if (bucket_extend_if_need(ix, list_no, (size_t)n)) return 0;
    go_bucket* b = &ix->b[list_no];
    int64_t pos = b->size;
    memcpy(b->ids + pos, keys, sizeof(int64_t) * n);
    memcpy(b->codes + pos * ix->code_size, codes, (size_t)n * ix->code_size);
    for (int i = 0; i < n; i++) {
        while (keys[i] >= ix->nids) ids_extend(ix);
        ix->vid_pos[keys[i]] = ((int64_t)list_no << 32) | pos;
        pos++;
        /* :293-296: a key whose doc is already deleted counts as deleted at once */
        if (ix->docids_bitmap && keys[i] >= 0 && ix_docid(ix, keys[i]) < ix->docids_bits &&
            bm_test(ix->docids_bitmap, ix_docid(ix, keys[i])))
            b->deleted++;
    }
    b->size = pos; /* publish after the copies */
    return 1;
}

int64_t go_ivfpq_list_size(go_ivfpq* ix, int l) { return ix->b[l].size; }
int64_t go_ivfpq_list_capacity(go_ivfpq* ix, int l) { return ix->b[l].capacity; }
void go_ivfpq_get_list(go_ivfpq* ix, int l, int64_t* ids, uint8_t* codes) {
    memcpy(ids, ix->b[l].ids, sizeof(int64_t) * ix->b[l].size);
    memcpy(codes, ix->b[l].codes, (size_t)ix->b[l].size * ix->code_size);
}
int64_t go_ivfpq_vid_pos(go_ivfpq* ix, int64_t vid) {
    return (vid >= 0 && vid < ix->nids) ? ix->vid_pos[vid] : -1;
}

/* quantizer->assign + compute_residuals + pq.compute_codes
 * (gamma_index_ivfpq.cc:455-472).  assign == search with k=1; the nx>=20 branch of the
 * reference goes through sgemm_, restated as mode 1 of go_knn_L2sqr. */
static void encode_impl(go_ivfpq* ix, int mode, int64_t n, const float* x, int64_t* list_nos,
                        uint8_t* codes) {
    int d = ix->d;
    float* dis = (float*)malloc(sizeof(float) * (n ? n : 1));
    go_knn_L2sqr(mode, x, ix->cc, d, n, ix->nlist, 1, dis, list_nos);
    free(dis);
    float* res = (float*)malloc(sizeof(float) * (size_t)(n ? n : 1) * d);
    for (int64_t i = 0; i < n; i++) {
        if (list_nos[i] < 0)
            memset(res + i * d, 0, sizeof(float) * d);
        else
            for (int t = 0; t < d; t++) res[i * d + t] = x[i * d + t] - ix->cc[list_nos[i] * d + t];
    }
    go_pq_compute_codes(ix->pqc, ix->M, ix->ksub, ix->dsub, res, codes, n);
    free(res);
}

static int g_assign_mode = -1; /* -1: faiss rule (n<20 seq else BLAS form) */
void go_set_assign_mode(int mode) { g_assign_mode = mode; }

void go_ivfpq_encode(go_ivfpq* ix, int64_t n, const float* x, int64_t* list_nos, uint8_t* codes) {
    int mode = g_assign_mode < 0 ? (n < 20 ? 0 : 1) : g_assign_mode;
    encode_impl(ix, mode, n, x, list_nos, codes);
}

/* GammaIVFPQIndex::Add, gamma_index_ivfpq.cc:424-512: group by list in ascending list
 * order (std::map), vids consecutive from indexed_vec_count_. */
int go_ivfpq_add(go_ivfpq* ix, int64_t n, const float* x) {
    int64_t* lno = (int64_t*)malloc(sizeof(int64_t) * (n ? n : 1));
    uint8_t* codes = (uint8_t*)malloc((size_t)(n ? n : 1) * ix->code_size);
    go_ivfpq_encode(ix, n, x, lno, codes);
    int64_t vid0 = ix->indexed_vec_count;
    int* cnt = (int*)calloc(ix->nlist, sizeof(int));
    for (int64_t i = 0; i < n; i++) {
        if (lno[i] < 0) lno[i] = (vid0 + i) % ix->nlist;
        cnt[lno[i]]++;
    }
    int ok = 1;
    int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (n ? n : 1));
    uint8_t* kc = (uint8_t*)malloc((size_t)(n ? n : 1) * ix->code_size);
    for (int l = 0; l < ix->nlist && ok; l++) {
        if (!cnt[l]) continue;
        int c = 0;
        for (int64_t i = 0; i < n; i++)
            if (lno[i] == l) {
                keys[c] = vid0 + i;
                memcpy(kc + (size_t)c * ix->code_size, codes + i * ix->code_size, ix->code_size);
                c++;
            }
        if (!go_ivfpq_add_keys(ix, l, c, keys, kc)) ok = 0;
    }
    if (ok) ix->indexed_vec_count = vid0 + n;
    free(keys);
    free(kc);
    free(cnt);
    free(lno);
    free(codes);
    return ok;
}

/* GammaIVFPQIndex::Update -> RealTimeMemData::Update (gamma_index_ivfpq.cc:375-422,
 * realtime_mem_data.cc:305-327) */
/* RealTimeMemData::Update(bucket_no, vid, codes), realtime_mem_data.cc:305-327 */
int go_ivfpq_update_code(go_ivfpq* ix, int list_no, int64_t vid, const uint8_t* code) {
    if (vid < 0 || vid >= ix->nids) return 0;
    int64_t bp = ix->vid_pos[vid];
    if (bp == -1) return 0;
    int old_b = (int)(bp >> 32), old_pos = (int)(bp & 0xffffffff);
    if (old_b == list_no) {
        memcpy(ix->b[old_b].codes + (size_t)old_pos * ix->code_size, code, ix->code_size);
        return 0;
    }
    ix->b[old_b].ids[old_pos] |= GO_DEL_MASK;
    ix->b[old_b].deleted++;
    return go_ivfpq_add_keys(ix, list_no, 1, &vid, code);
}

/* Helpers of the list-sharded Update test (gamma_amd/dist.py sharded_update): whether this index holds a live
 * entry of vid, and the first half of RealTimeMemData::Update alone (:318-321: flag the old entry, count it). */
int go_ivfpq_has_vid(go_ivfpq* ix, int64_t vid) {
    return vid >= 0 && vid < ix->nids && ix->vid_pos[vid] != -1;
}

int go_ivfpq_remove(go_ivfpq* ix, int64_t vid) {
    if (vid < 0 || vid >= ix->nids) return 0;
    int64_t bp = ix->vid_pos[vid];
    if (bp == -1) return 0;
    int old_b = (int)(bp >> 32), old_pos = (int)(bp & 0xffffffff);
    ix->b[old_b].ids[old_pos] |= GO_DEL_MASK;
    ix->b[old_b].deleted++;
    ix->vid_pos[vid] = -1;
    return 0;
}

int go_ivfpq_update(go_ivfpq* ix, int64_t vid, const float* x) {
    int64_t lno;
    uint8_t* code = (uint8_t*)malloc(ix->code_size);
    encode_impl(ix, 0, 1, x, &lno, code);
    int ret = go_ivfpq_update_code(ix, (int)lno, vid, code);
    free(code);
    return ret;
}

/* RealTimeMemData::Delete, :329-335,190-199: counter only */
int go_ivfpq_delete(go_ivfpq* ix, const int64_t* vids, int n, const uint8_t* docids_bitmap) {
    (void)docids_bitmap;
    for (int i = 0; i < n; i++) {
        if (vids[i] >= ix->nids) continue;
        int64_t bp = ix->vid_pos[vids[i]];
        if (bp == -1) continue;
        ix->b[bp >> 32].deleted++;
    }
    return 0;
}

/* CompactIfNeed / CompactBucket, :354-381,119-150 */
int go_ivfpq_compact_if_need(go_ivfpq* ix, const uint8_t* docids_bitmap) {
    if (!docids_bitmap) docids_bitmap = ix->docids_bitmap;
    for (int l = 0; l < ix->nlist; l++) {
        go_bucket* b = &ix->b[l];
        if (!((float)b->deleted / b->size >= 0.3f)) continue;
        int64_t* ni = (int64_t*)malloc(sizeof(int64_t) * b->capacity);
        uint8_t* nc = (uint8_t*)malloc((size_t)b->capacity * ix->code_size);
        int pos = 0;
        for (int64_t i = 0; i < b->size; i++) {
            int64_t id = b->ids[i];
            if (!(id & GO_DEL_MASK) &&
                !(docids_bitmap && bm_test(docids_bitmap, ix_docid(ix, id & GO_RECOVER_MASK)))) {
                ni[pos] = id;
                memcpy(nc + (size_t)pos * ix->code_size, b->codes + (size_t)i * ix->code_size,
                       ix->code_size);
                ix->vid_pos[id] = ((int64_t)l << 32) | pos;
                pos++;
            }
        }
        free(b->ids);
        free(b->codes);
        b->ids = ni;
        b->codes = nc;
        ix->compacted_num += b->size - pos;
        b->size = pos;
        b->deleted = 0;
    }
    return 0;
}

/* ===================================================================================
 * GammaIVFPQIndex::Search (gamma_index_ivfpq.cc:514-566) + search_preassigned (:701-890)
 * =================================================================================== */
int go_ivfpq_search(go_ivfpq* ix, const go_search_ctx* ctx, int metric, int nprobe,
                    int recall_num, int has_rank, int coarse_mode, int nq, const float* x, int k,
                    float* distances, int64_t* labels, float* coarse_dis_out,
                    int64_t* coarse_idx_out, float* recall_dis_out, int64_t* recall_ids_out) {
    if (k <= 0) return 0; /* :753-756 */
    if (nprobe <= 0 || nprobe > ix->nlist) return -1;
    int d = ix->d, M = ix->M, ksub = ix->ksub;
    size_t tsz = (size_t)M * ksub;
    int R = recall_num < k ? k : recall_num; /* :762-765 */
    int ks = metric == GO_METRIC_IP ? 0 : 1;

    float* coarse_dis = (float*)malloc(sizeof(float) * (size_t)nq * nprobe);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)nq * nprobe);
    int mode = coarse_mode < 0 ? (nq < 20 ? 0 : 1) : coarse_mode; /* distances.cpp:303,346 */
    if (mode == 2) {
        /* search_preassigned with an assignment computed elsewhere (:563,701): coarse_dis_out / coarse_idx_out are INPUTS
         * -- e.g. the compiled library's own quantizer->search with its BLAS path (tests/gen_golden.py, blas leg) */
        if (!coarse_dis_out || !coarse_idx_out) return -1;
        memcpy(coarse_dis, coarse_dis_out, sizeof(float) * (size_t)nq * nprobe);
        memcpy(idx, coarse_idx_out, sizeof(int64_t) * (size_t)nq * nprobe);
    } else {
        go_knn_L2sqr(mode, x, ix->cc, d, nq, ix->nlist, nprobe, coarse_dis, idx);
        if (coarse_dis_out) memcpy(coarse_dis_out, coarse_dis, sizeof(float) * (size_t)nq * nprobe);
        if (coarse_idx_out) memcpy(coarse_idx_out, idx, sizeof(int64_t) * (size_t)nq * nprobe);
    }

#pragma omp parallel
    {
        float* sim_table = (float*)malloc(sizeof(float) * tsz);
        float* sim_table_2 = (float*)malloc(sizeof(float) * tsz);
        float* residual = (float*)malloc(sizeof(float) * d);
        float* rdis = (float*)malloc(sizeof(float) * R);
        int64_t* rids = (int64_t*)malloc(sizeof(int64_t) * R);
#pragma omp for schedule(dynamic)
        for (int i = 0; i < nq; i++) {
            const float* xi = x + (size_t)i * d;
            float* simi = distances + (size_t)i * k;
            int64_t* idxi = labels + (size_t)i * k;
            /* set_query -> init_query_IP / init_query_L2 (.h:148-168) */
            if (metric == GO_METRIC_IP)
                go_pq_inner_prod_table(ix->pqc, M, ksub, ix->dsub, xi, sim_table);
            else if (ix->use_precomputed_table == 1)
                go_pq_inner_prod_table(ix->pqc, M, ksub, ix->dsub, xi, sim_table_2);
            go_heap_heapify(ks, k, simi, idxi);
            go_heap_heapify(ks, R, rdis, rids);

            for (int ik = 0; ik < nprobe; ik++) {
                int64_t key = idx[(size_t)i * nprobe + ik];
                if (key < 0 || key >= ix->nlist) continue; /* scan_one_list :602-609 */
                const go_bucket* b = &ix->b[key];
                int64_t list_size = b->size;
                if (list_size == 0) continue;
                float dis0 = 0;
                /* set_list -> precompute_list_tables (.h:184-257), by_residual = true */
                if (metric == GO_METRIC_IP) {
                    dis0 = go_fvec_inner_product(xi, ix->cc + key * d, d);
                } else if (ix->use_precomputed_table == 1) {
                    dis0 = coarse_dis[(size_t)i * nprobe + ik];
                    go_fvec_madd(tsz, ix->table + key * tsz, -2.0f, sim_table_2, sim_table);
                } else {
                    for (int t = 0; t < d; t++) residual[t] = xi[t] - ix->cc[key * d + t];
                    for (int m = 0; m < M; m++) /* compute_distance_table */
                        go_fvec_L2sqr_ny(sim_table + (size_t)m * ksub, residual + m * ix->dsub,
                                         ix->pqc + (size_t)m * ksub * ix->dsub, ix->dsub, ksub);
                }
                /* scan_list_with_table (.h:575-601) + KnnSearchResults::add (.h:363-369) */
                const uint8_t* codes = b->codes;
                for (int64_t j = 0; j < list_size; j++, codes += ix->code_size) {
                    int64_t id = b->ids[j];
                    if (id & GO_DEL_MASK) continue;
                    if (!ctx_is_valid(ctx, id & GO_RECOVER_MASK)) continue;
                    float dis = dis0;
                    const float* tab = sim_table;
                    for (int m = 0; m < M; m++) {
                        dis += tab[codes[m]];
                        tab += ksub;
                    }
                    if (hcmp(ks, rdis[0], dis)) go_heap_replace_top(ks, R, rdis, rids, dis, id);
                }
            }

            /* compute_dis (:642-697) */
            if (has_rank) {
                for (int j = 0; j < R; j++) {
                    if (rids[j] == -1) continue;
                    const float* v = ix->raw + rids[j] * d;
                    float dis = metric == GO_METRIC_IP ? go_fvec_inner_product(xi, v, d)
                                                       : go_fvec_L2sqr(xi, v, d);
                    if (ctx_score_valid(ctx, dis) && hcmp(ks, simi[0], dis)) {
                        go_heap_pop(ks, k, simi, idxi);
                        go_heap_push(ks, k, simi, idxi, dis, rids[j]);
                    }
                }
                go_heap_reorder(ks, k, simi, idxi);
                if (recall_dis_out) go_heap_reorder(ks, R, rdis, rids);
            } else {
                int c = 0;
                go_heap_reorder(ks, R, rdis, rids);
                for (int j = 0; j < R; j++) {
                    if (rids[j] == -1) continue;
                    float dis = rdis[j];
                    if (ctx_score_valid(ctx, dis)) {
                        simi[c] = dis;
                        idxi[c] = rids[j];
                        ++c;
                    }
                    if (c >= k) break;
                }
            }
            if (recall_dis_out) {
                memcpy(recall_dis_out + (size_t)i * R, rdis, sizeof(float) * R);
                memcpy(recall_ids_out + (size_t)i * R, rids, sizeof(int64_t) * R);
            }
        }
        free(sim_table);
        free(sim_table_2);
        free(residual);
        free(rdis);
        free(rids);
    }
    free(coarse_dis);
    free(idx);
    return 0;
}

/* ===================================================================================
 * GammaFLATIndex::Search (index/impl/gamma_index_flat.cc:118-300).  The reference's
 * "parallel over queries" branch runs serially (orphaned omp for, :237); results do not
 * depend on that, so this restatement parallelises over queries.
 * =================================================================================== */
int go_flat_search(const float* raw, int64_t n, int d, const go_search_ctx* ctx, int metric,
                   int nq, const float* x, int k, float* distances, int64_t* labels) {
    if (!x) return -1;
    int ks = metric == GO_METRIC_IP ? 0 : 1;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < nq; i++) {
        const float* xi = x + (size_t)i * d;
        float* simi = distances + (size_t)i * k;
        int64_t* idxi = labels + (size_t)i * k;
        go_heap_heapify(ks, k, simi, idxi);
        for (int64_t vid = 0; vid < n; vid++) {
            if (!ctx_is_valid(ctx, vid)) continue;
            const float* yi = raw + vid * d;
            float dis = metric == GO_METRIC_IP ? go_fvec_inner_product(xi, yi, d)
                                               : go_fvec_L2sqr(xi, yi, d);
            if (!ctx_score_valid(ctx, dis)) continue;
            if (hcmp(ks, simi[0], dis)) {
                go_heap_pop(ks, k, simi, idxi);
                go_heap_push(ks, k, simi, idxi, dis, vid);
            }
        }
        go_heap_reorder(ks, k, simi, idxi);
    }
    return 0;
}

/* ===================================================================================
 * IVFFLAT (index/impl/gamma_index_ivfflat.{h,cc}): the lists hold the vectors themselves
 * (code_size = 4 d, gamma_index_ivfflat.cc:155); a search scans the probed lists in probe order with
 * GammaIVFFlatScanner1::scan_codes (gamma_index_ivfflat.h:52-75): bit 63, IsValid, exact
 * fvec_L2sqr / fvec_inner_product, score window, heap_pop + heap_push into the k-heap; then heap_reorder
 * (search_preassigned pmode 0, gamma_index_ivfflat.cc:539-567).  Here the lists of a go_ivfpq (ids only are
 * used) stand for the RTInvertIndex and the raw store for the list payload: the same floats.
 * =================================================================================== */
int go_ivfflat_search(go_ivfpq* ix, const go_search_ctx* ctx, int metric, int nprobe, int coarse_mode,
                      int nq, const float* x, int k, float* distances, int64_t* labels,
                      float* coarse_dis_out, int64_t* coarse_idx_out) {
    if (k <= 0) return 0;
    if (nprobe <= 0 || nprobe > ix->nlist || !ix->raw) return -1;
    const int d = ix->d;
    const int ks = metric == GO_METRIC_IP ? 0 : 1;
    float* coarse_dis = (float*)malloc(sizeof(float) * (size_t)nq * nprobe);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)nq * nprobe);
    int mode = coarse_mode < 0 ? (nq < 20 ? 0 : 1) : coarse_mode; /* quantizer->search, :413 */
    go_knn_L2sqr(mode, x, ix->cc, d, nq, ix->nlist, nprobe, coarse_dis, idx);
    if (coarse_dis_out) memcpy(coarse_dis_out, coarse_dis, sizeof(float) * (size_t)nq * nprobe);
    if (coarse_idx_out) memcpy(coarse_idx_out, idx, sizeof(int64_t) * (size_t)nq * nprobe);
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < nq; i++) {
        const float* xi = x + (size_t)i * d;
        float* simi = distances + (size_t)i * k;
        int64_t* idxi = labels + (size_t)i * k;
        go_heap_heapify(ks, k, simi, idxi);
        for (int ik = 0; ik < nprobe; ik++) {
            int64_t key = idx[(size_t)i * nprobe + ik];
            if (key < 0 || key >= ix->nlist) continue; /* scan_one_list, :490-503 */
            const go_bucket* b = &ix->b[key];
            for (int64_t j = 0; j < b->size; j++) {
                int64_t id = b->ids[j];
                if (id & GO_DEL_MASK) continue;
                int64_t vid = id & GO_RECOVER_MASK;
                if (!ctx_is_valid(ctx, vid)) continue;
                if (vid >= ix->nraw) continue; /* not in the store: cannot happen through Add */
                const float* yj = ix->raw + vid * d;
                float dis = metric == GO_METRIC_IP ? go_fvec_inner_product(xi, yj, d) : go_fvec_L2sqr(xi, yj, d);
                if (ctx_score_valid(ctx, dis) && hcmp(ks, simi[0], dis)) {
                    go_heap_pop(ks, k, simi, idxi);
                    go_heap_push(ks, k, simi, idxi, dis, vid);
                }
            }
        }
        go_heap_reorder(ks, k, simi, idxi);
    }
    free(coarse_dis);
    free(idx);
    return 0;
}

/* GammaIndexIVFFlat::Add / Update (gamma_index_ivfflat.cc:305-374): quantizer->assign, AddKeys / Update with the
 * vector as the code.  list_nos out: the assignment (n >= 20 takes the BLAS form unless go_set_assign_mode). */
void go_ivfflat_assign(go_ivfpq* ix, int64_t n, const float* x, int64_t* list_nos) {
    int mode = g_assign_mode < 0 ? (n < 20 ? 0 : 1) : g_assign_mode;
    float* dis = (float*)malloc(sizeof(float) * (size_t)n);
    go_knn_L2sqr(mode, x, ix->cc, ix->d, n, ix->nlist, 1, dis, list_nos);
    free(dis);
}

int go_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ===================================================================================
 * Training: faiss::Clustering::train (faiss:Clustering.cpp:255-560) as GammaIVFPQIndex::Indexing drives it
 * (index/impl/gamma_index_ivfpq.cc:272-354 -> IndexIVFPQ::train: train_q1 with cp.niter = 10, then
 * train_residual_o, faiss:IndexIVFPQ.cpp:67-131, and ProductQuantizer::train, faiss:impl/ProductQuantizer.cpp:249-302).
 * nredo 1, no weights, no codec, no frozen centroids, L2.  The random numbers are std::mt19937's
 * (faiss:utils/random.cpp:18-38,136-146); the assignment step is go_knn_L2sqr (mode 1 from 20 points on: the
 * stand-in for sgemm_), so the result is a DETERMINISTIC function of the input -- the device trainer
 * is compared with it bit for bit; against the compiled faiss (MKL sgemm_ order, its own thread count) the
 * comparison is on the quantisation error (tests/test_training_cpu.py).
 * =================================================================================== */
typedef struct {
    uint32_t mt[624];
    int idx;
} go_mt19937;

static void mt_seed(go_mt19937* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(go_mt19937* g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* rand_perm, faiss:utils/random.cpp:136-146 */
void go_rand_perm(int* perm, size_t n, int64_t seed) {
    for (size_t i = 0; i < n; i++) perm[i] = (int)i;
    go_mt19937 g;
    mt_seed(&g, (uint32_t)seed);
    for (size_t i = 0; i + 1 < n; i++) {
        const int i2 = (int)(i + (size_t)(mt_next(&g) % (uint64_t)(int)(n - i)));
        const int t = perm[i];
        perm[i] = perm[i2];
        perm[i2] = t;
    }
}

/* compute_centroids + the normalisation, faiss:Clustering.cpp:138-208: per centroid a float sum over its points in
 * ascending point order, then c[j] *= 1 / hassign */
void go_kmeans_update(int d, int k, int64_t n, const float* x, const int64_t* assign, float* hassign, float* centroids) {
    memset(centroids, 0, sizeof(float) * (size_t)d * k);
    memset(hassign, 0, sizeof(float) * (size_t)k);
    for (int64_t i = 0; i < n; i++) {
        const int64_t ci = assign[i];
        float* c = centroids + ci * d;
        const float* xi = x + i * d;
        hassign[ci] += 1.0f;
        for (int j = 0; j < d; j++) c[j] += xi[j];
    }
    for (int ci = 0; ci < k; ci++) {
        if (hassign[ci] == 0) continue;
        const float norm = 1 / hassign[ci];
        float* c = centroids + (size_t)ci * d;
        for (int j = 0; j < d; j++) c[j] *= norm;
    }
}

/* split_clusters, faiss:Clustering.cpp:220-268 */
int go_kmeans_split(int d, int k, int64_t n, float* hassign, float* centroids) {
    int nsplit = 0;
    go_mt19937 g;
    mt_seed(&g, 1234u);
    for (int ci = 0; ci < k; ci++) {
        if (hassign[ci] != 0) continue;
        int cj;
        for (cj = 0; 1; cj = (cj + 1) % k) {
            const float p = (float)(((double)hassign[cj] - 1.0) / (double)(float)(n - k));
            const float r = (float)mt_next(&g) / 4294967296.0f;   /* mt() / float(mt.max()) */
            if (r < p) break;
        }
        memcpy(centroids + (size_t)ci * d, centroids + (size_t)cj * d, sizeof(float) * d);
        for (int j = 0; j < d; j++) {
            float* a = &centroids[(size_t)ci * d + j];
            float* b = &centroids[(size_t)cj * d + j];
            if (j % 2 == 0) {
                *a = (float)((double)*a * (1 + 1 / 1024.));
                *b = (float)((double)*b * (1 - 1 / 1024.));
            } else {
                *a = (float)((double)*a * (1 - 1 / 1024.));
                *b = (float)((double)*b * (1 + 1 / 1024.));
            }
        }
        hassign[ci] = hassign[cj] / 2;
        hassign[cj] -= hassign[ci];
        nsplit++;
    }
    return nsplit;
}

/* Clustering::train.  centroids: k*d out.  Returns the objective of the last assignment (sum of the distances,
 * accumulated in float in point order, :478-481). */
static int g_kmeans_assign_mode = -1; /* -1: faiss's rule (exact below 20 points); 0 / 1 force a form (tests) */
void go_set_kmeans_assign_mode(int mode) { g_kmeans_assign_mode = mode; }

float go_kmeans(int d, int64_t n, const float* x_in, int k, int niter, int64_t seed, int max_points_per_centroid,
                float* centroids) {
    const float* x = x_in;
    float* x_new = NULL;
    if (n > (int64_t)k * max_points_per_centroid) { /* subsample_training_set, :92-120 */
        int* perm = (int*)malloc(sizeof(int) * (size_t)n);
        go_rand_perm(perm, (size_t)n, seed);
        const int64_t n2 = (int64_t)k * max_points_per_centroid;
        x_new = (float*)malloc(sizeof(float) * (size_t)n2 * d);
        for (int64_t i = 0; i < n2; i++) memcpy(x_new + i * d, x_in + (size_t)perm[i] * d, sizeof(float) * d);
        free(perm);
        n = n2;
        x = x_new;
    }
    if (n == k) { /* :334-355 */
        memcpy(centroids, x, sizeof(float) * (size_t)d * k);
        free(x_new);
        return 0.f;
    }
    {
        int* perm = (int*)malloc(sizeof(int) * (size_t)n);
        go_rand_perm(perm, (size_t)n, seed + 1); /* :412 */
        for (int i = 0; i < k; i++) memcpy(centroids + (size_t)i * d, x + (size_t)perm[i] * d, sizeof(float) * d);
        free(perm);
    }
    int64_t* assign = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    float* dis = (float*)malloc(sizeof(float) * (size_t)n);
    float* hassign = (float*)malloc(sizeof(float) * (size_t)k);
    float obj = 0;
    for (int it = 0; it < niter; it++) {
        go_knn_L2sqr(g_kmeans_assign_mode >= 0 ? g_kmeans_assign_mode : (n < 20 ? 0 : 1), x, centroids, (size_t)d, (size_t)n,
                     (size_t)k, 1, dis, assign);
        obj = 0;
        for (int64_t j = 0; j < n; j++) obj += dis[j];
        go_kmeans_update(d, k, n, x, assign, hassign, centroids);
        go_kmeans_split(d, k, n, hassign, centroids);
    }
    free(assign);
    free(dis);
    free(hassign);
    free(x_new);
    return obj;
}

/* IndexIVFPQ::train as Gamma configures it: coarse k-means (niter 10, seed 1234), then the product quantizer on the
 * residuals of at most 256 * 256 points (fvecs_maybe_subsample with pq.cp.seed = 1234; niter 25 per sub-quantizer) */
void go_ivfpq_train(int d, int nlist, int M, int64_t n, const float* x, float* cc, float* pq) {
    go_kmeans(d, n, x, nlist, 10, 1234, 256, cc);
    const int64_t nmax = 256 * 256;
    const float* xs = x;
    float* sub = NULL;
    int64_t ns = n;
    if (n > nmax) {
        int* perm = (int*)malloc(sizeof(int) * (size_t)n);
        go_rand_perm(perm, (size_t)n, 1234);
        sub = (float*)malloc(sizeof(float) * (size_t)nmax * d);
        for (int64_t i = 0; i < nmax; i++) memcpy(sub + i * d, x + (size_t)perm[i] * d, sizeof(float) * d);
        free(perm);
        xs = sub;
        ns = nmax;
    }
    int64_t* assign = (int64_t*)malloc(sizeof(int64_t) * (size_t)ns);
    float* dis = (float*)malloc(sizeof(float) * (size_t)ns);
    go_knn_L2sqr(ns < 20 ? 0 : 1, xs, cc, (size_t)d, (size_t)ns, (size_t)nlist, 1, dis, assign);
    const int dsub = d / M;
    float* slice = (float*)malloc(sizeof(float) * (size_t)ns * dsub);
    for (int m = 0; m < M; m++) {
        for (int64_t i = 0; i < ns; i++)
            for (int t = 0; t < dsub; t++) /* compute_residual: x - centroid */
                slice[i * dsub + t] = xs[i * d + m * dsub + t] - cc[(size_t)assign[i] * d + m * dsub + t];
        go_kmeans(dsub, ns, slice, 256, 25, 1234, 256, pq + (size_t)m * 256 * dsub);
    }
    free(slice);
    free(assign);
    free(dis);
    free(sub);
}
