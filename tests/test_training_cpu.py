"""CPU: the training restatement (oracle/gamma_oracle.c "Training": faiss::Clustering::train, IndexIVFPQ::train as
GammaIVFPQIndex::Indexing configures them) against the compiled faiss of oracle/_ref.
  * the random numbers: go_rand_perm == faiss::rand_perm (std::mt19937), exactly;
  * k-means with the exact assignment form on both sides (blas threshold raised): centroids bit for bit -- subsampling,
    initialisation, centroid sums, empty-cluster splits are faiss's;
  * with the production assignment form (GEMM: MKL sgemm_ there, the k-ascending chain here) the two runs part ways at
    the first point whose two nearest centroids differ by an ulp, so they are compared on what training is for: the
    quantisation error of the coarse quantizer and the reconstruction error of the product quantizer, within 1 %."""
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


def _quant_err(x, cen):
    D, _ = B.knn_L2sqr(x, cen, 1, mode=0)
    return float(D.sum())


def test_ivfpq_training_matches_the_library_within_one_percent():
    """What GammaIVFPQIndex::Indexing produces on the same training set: coarse quantisation error and PQ
    reconstruction error of the restatement (production assignment form) against faiss's IndexIVFPQ::train."""
    d, nlist, M = 32, 64, 8
    x = synth.sift_like(12000, d=d, seed=21)
    r = B.RefIVFPQ(d, nlist, M, 8, B.METRIC_L2)
    r.train(x)
    ref_cc, ref_pq = r.coarse_centroids(), r.pq_centroids()
    cc, pq = B.ivfpq_train(x, nlist, M)
    e_ref, e_own = _quant_err(x, ref_cc), _quant_err(x, cc)
    assert abs(e_own - e_ref) <= 0.01 * e_ref, (e_own, e_ref)

    def recon_err(cc_, pq_):
        o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2)
        o.set_trained(cc_, pq_, None)
        lno, codes = o.encode(x[:4000])
        dsub = d // M
        rec = cc_[lno].copy()
        for m in range(M):
            rec[:, m * dsub:(m + 1) * dsub] += pq_[m][codes[:, m]]
        return float(((x[:4000] - rec) ** 2).sum())

    r_ref, r_own = recon_err(ref_cc, ref_pq), recon_err(cc, pq)
    assert abs(r_own - r_ref) <= 0.01 * r_ref, (r_own, r_ref)
