"""CPU, only where oracle/_ref exists (the container with /root/reference): live bit-exact
comparison of the oracle with the real faiss 1.7.1 on fresh random inputs -- the wider
version of what tests/golden pins."""
import numpy as np
import pytest

from gamma_amd import synth
from oracle import binding as B

pytestmark = pytest.mark.skipif(not B.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def same(a, b):
    return np.float32(a).tobytes() == np.float32(b).tobytes()


def test_scalar_primitives_random():
    L, R = B.lib(), B.ref()
    rng = np.random.default_rng(123)
    for d in [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17, 24, 31, 32, 33, 63, 64, 65, 96, 127, 128,
              129, 200, 256, 512, 768, 1000]:
        for _ in range(20):
            x = (rng.standard_normal(d) * rng.uniform(0.01, 1000)).astype(np.float32)
            y = (rng.standard_normal(d) * rng.uniform(0.01, 1000)).astype(np.float32)
            assert same(L.go_fvec_L2sqr(B._fp(x), B._fp(y), d), R.ref_fvec_L2sqr(B._fp(x), B._fp(y), d))
            assert same(L.go_fvec_inner_product(B._fp(x), B._fp(y), d),
                        R.ref_fvec_inner_product(B._fp(x), B._fp(y), d))
            assert same(L.go_fvec_norm_L2sqr(B._fp(x), d), R.ref_fvec_norm_L2sqr(B._fp(x), d))


def test_ny_and_madd_random():
    L, R = B.lib(), B.ref()
    rng = np.random.default_rng(5)
    for d in [1, 2, 3, 4, 6, 8, 12, 16, 20, 32]:
        for ny in [256, 255, 1, 9]:
            x = rng.standard_normal(d).astype(np.float32)
            y = rng.standard_normal((ny, d)).astype(np.float32)
            for fn in ("fvec_inner_products_ny", "fvec_L2sqr_ny"):
                a, b = np.empty(ny, np.float32), np.empty(ny, np.float32)
                getattr(L, "go_" + fn)(B._fp(a), B._fp(x), B._fp(y), d, ny)
                getattr(R, "ref_" + fn)(B._fp(b), B._fp(x), B._fp(y), d, ny)
                assert a.tobytes() == b.tobytes(), (fn, d, ny)
    a = rng.standard_normal(4097).astype(np.float32)
    b = rng.standard_normal(4097).astype(np.float32)
    for bf in (-2.0, 2.0, 0.3):
        for off in (0, 1):  # aligned SSE path and unaligned scalar path
            n = 4096
            c1, c2 = np.empty(n + 1, np.float32), np.empty(n + 1, np.float32)
            L.go_fvec_madd(n, B._fp(a[off:]), bf, B._fp(b[off:]), B._fp(c1[off:]))
            R.ref_fvec_madd(n, B._fp(a[off:]), bf, B._fp(b[off:]), B._fp(c2[off:]))
            assert c1[off:off + n].tobytes() == c2[off:off + n].tobytes()


@pytest.mark.parametrize("cfg", [(32, 64, 8, 12000, 50, B.METRIC_L2, 8, 100),
                                 (128, 64, 16, 12000, 40, B.METRIC_L2, 16, 200),
                                 (96, 32, 8, 8000, 30, B.METRIC_IP, 8, 60),
                                 (64, 32, 4, 8000, 30, B.METRIC_L2, 4, 50)])
def test_ivfpq_end_to_end(cfg):
    d, nlist, M, N, nq, metric, nprobe, Rk = cfg
    R = B.ref()
    R.ref_set_blas_threshold(1 << 30)
    B.lib().go_set_assign_mode(0)
    base = synth.sift_like(N, d=d, seed=77)
    q = synth.sift_like(nq, d=d, seed=78)
    r = B.RefIVFPQ(d, nlist, M, 8, metric)
    r.train(base[:nlist * 64])
    r.add(base)
    o = B.OracleIVFPQ(d, nlist, M, 8, metric)
    o.set_trained(r.coarse_centroids(), r.pq_centroids(), None)
    assert o.table().tobytes() == r.precomputed_table().tobytes()
    assert o.add(base)
    for l in range(nlist):
        io, co = o.get_list(l)
        ir, cr = r.get_list(l)
        assert np.array_equal(io, ir) and np.array_equal(co, cr)
    o.set_raw(base)
    ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
    for m in (B.METRIC_L2, B.METRIC_IP):
        r.set_metric(m)
        Dr, Ir = r.search(q, Rk, nprobe)
        _, _, st = o.search(q, 10, nprobe, recall_num=Rk, has_rank=False, metric=m, ctx=ctx,
                            coarse_mode=0, want_stages=True)
        assert st["recall_dis"].tobytes() == Dr.tobytes()
        assert np.array_equal(st["recall_ids"], Ir)
    r.set_metric(metric)
    R.ref_set_blas_threshold(20)


def test_default_blas_coarse_path_close_to_gemm_form():
    """With faiss's default threshold (nq >= 20 -> MKL sgemm) the reference's own coarse
    distances depend on the BLAS; our restated GEMM form must agree to rounding."""
    R = B.ref()
    R.ref_set_blas_threshold(20)
    rng = np.random.default_rng(9)
    y = synth.sift_like(512, d=64, seed=3)
    x = synth.sift_like(64, d=64, seed=4)
    D = np.empty((64, 8), np.float32)
    I = np.empty((64, 8), np.int64)
    R.ref_flat_l2_search(64, 512, B._fp(y), 64, B._fp(x), 8, B._fp(D), B._ip(I))
    D1, I1 = B.knn_L2sqr(x, y, 8, mode=1)
    assert np.allclose(D, D1, rtol=1e-5, atol=1e-2)
    assert (I == I1).mean() > 0.99
