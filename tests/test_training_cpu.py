"""CPU: the training restatement (oracle/gamma_oracle.c "Training": faiss::Clustering::train, IndexIVFPQ::train as
GammaIVFPQIndex::Indexing configures them) against the compiled faiss of oracle/_ref.
  * the random numbers: go_rand_perm == faiss::rand_perm (std::mt19937), exactly;
  * k-means with the exact assignment form on both sides (blas threshold raised): centroids bit for bit -- subsampling,
    initialisation, centroid sums, empty-cluster splits are faiss's;
  * with the production assignment form (the library's default BLAS threshold: MKL sgemm_ there, the restated GEMM form
    here -- the same sums bit for bit, go_gemm_k_split) k-means and the whole IndexIVFPQ::train are bit-identical too."""
import numpy as np
import pytest

from gamma_amd import synth
from oracle import binding as B

pytestmark = pytest.mark.skipif(not B.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def test_rand_perm_is_the_librarys():
    import ctypes as C
    R = B.ref()
    for n, seed in ((1, 5), (2, 5), (1000, 1234), (4099, 1235), (70000, 99)):
        a = np.empty(n, np.int32)
        b = np.empty(n, np.int32)
        B.lib().go_rand_perm(a.ctypes.data_as(C.POINTER(C.c_int)), n, seed)
        R.ref_rand_perm(b.ctypes.data_as(C.POINTER(C.c_int)), n, seed)
        assert np.array_equal(a, b), (n, seed)


@pytest.mark.parametrize("n,d,k,niter", [(3000, 16, 32, 10), (70000, 8, 256, 6), (520, 4, 256, 25), (64, 8, 64, 3)])
def test_kmeans_with_exact_assignment_is_bit_identical(n, d, k, niter):
    """520 points for 256 centroids leaves clusters empty: split_clusters and its RandomGenerator(1234) run;
    70000 > 256 * 256: subsample_training_set runs; n == k: the copy corner case."""
    R = B.ref()
    x = synth.sift_like(n, d=d, seed=7)
    old = R.ref_get_blas_threshold()
    R.ref_set_blas_threshold(1 << 30)
    try:
        ref_c = np.empty((k, d), np.float32)
        ref_obj = R.ref_kmeans(d, n, B._fp(x), k, niter, 1234, B._fp(ref_c))
    finally:
        R.ref_set_blas_threshold(old)
    cen, obj = B.kmeans(x, k, niter, seed=1234, assign_mode=0)
    assert cen.tobytes() == ref_c.tobytes()
    assert obj == pytest.approx(ref_obj, rel=0, abs=0)


@pytest.mark.parametrize("n,d,k,niter,kind", [(3000, 16, 32, 10, "sift"), (70000, 8, 256, 6, "sift"), (520, 4, 256, 25, "sift"),
                                              (20000, 128, 64, 10, "sift"), (5000, 24, 40, 10, "gauss"), (20000, 96, 128, 10, "gauss")])
def test_kmeans_with_the_production_assignment_is_bit_identical(n, d, k, niter, kind):
    """The library's DEFAULT blas threshold (20): every assignment step of Clustering::train goes through
    exhaustive_L2sqr_blas -- MKL sgemm_.  The restated GEMM form is that computation bit for bit (go_gemm_k_split,
    tests/test_oracle_vs_ref.py::test_gemm_form_is_the_compiled_sgemm), so the trained centroids are the library's."""
    R = B.ref()
    x = synth.sift_like(n, d=d, seed=7) if kind == "sift" else \
        (np.random.default_rng(3).standard_normal((n, d)) * 1.5).astype(np.float32)
    old = R.ref_get_blas_threshold()
    R.ref_set_blas_threshold(20)
    try:
        ref_c = np.empty((k, d), np.float32)
        ref_obj = R.ref_kmeans(d, n, B._fp(x), k, niter, 1234, B._fp(ref_c))
    finally:
        R.ref_set_blas_threshold(old)
    cen, obj = B.kmeans(x, k, niter, seed=1234, assign_mode=-1)
    assert cen.tobytes() == ref_c.tobytes()
    assert obj == pytest.approx(ref_obj, rel=0, abs=0)


@pytest.mark.parametrize("d,nlist,M,n,kind", [(32, 64, 8, 12000, "sift"), (128, 256, 16, 256 * 64, "sift"), (64, 64, 8, 12000, "gauss")])
def test_ivfpq_training_is_the_librarys_bit_for_bit(d, nlist, M, n, kind):
    """What GammaIVFPQIndex::Indexing produces (index/impl/gamma_index_ivfpq.cc:272-354 -> IndexIVFPQ::train: coarse
    k-means with cp.niter = 10, residuals, ProductQuantizer::train) with the library's default BLAS threshold: coarse
    centroids and PQ codebooks of the restatement == the compiled library's, bit for bit."""
    R = B.ref()
    x = synth.sift_like(n, d=d, seed=21) if kind == "sift" else np.random.default_rng(5).standard_normal((n, d)).astype(np.float32)
    old = R.ref_get_blas_threshold()
    R.ref_set_blas_threshold(20)
    try:
        r = B.RefIVFPQ(d, nlist, M, 8, B.METRIC_L2)
        r.train(x)
        ref_cc, ref_pq = r.coarse_centroids(), r.pq_centroids()
    finally:
        R.ref_set_blas_threshold(old)
    cc, pq = B.ivfpq_train(x, nlist, M)
    assert cc.tobytes() == ref_cc.tobytes()
    assert pq.tobytes() == ref_pq.tobytes()
