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


@pytest.mark.parametrize("cfg", [(32, 48, 8, 9000, 40, 8, 100), (128, 32, 16, 8000, 32, 16, 200), (96, 24, 8, 6000, 24, 6, 80),
                                 (64, 32, 32, 8000, 30, 8, 60), (48, 16, 2, 4000, 20, 4, 50)])
def test_ivfpq_table_mode_0_end_to_end(cfg):
    """Live: faiss::precomputed_table_max_bytes lowered below the index's table -> the compiled library trains into table
    mode 0 (faiss:IndexIVFPQ.cpp:441-449) and scores with residual tables; the oracle under the same limit agrees bit for
    bit (dsub 4, 8, 12, 2 and the generic AVX row at dsub 24)."""
    d, nlist, M, N, nq, nprobe, Rk = cfg
    R = B.ref()
    L = B.lib()
    R.ref_set_blas_threshold(1 << 30)
    L.go_set_assign_mode(0)
    saved_r, saved_o = R.ref_get_precomputed_table_max_bytes(), L.go_get_precomputed_table_max_bytes()
    limit = nlist * M * 1024 - 1
    R.ref_set_precomputed_table_max_bytes(limit)
    L.go_set_precomputed_table_max_bytes(limit)
    try:
        base = synth.sift_like(N, d=d, seed=177)
        q = synth.sift_like(nq, d=d, seed=178)
        r = B.RefIVFPQ(d, nlist, M, 8, B.METRIC_L2)
        r.train(base[:nlist * 64])
        r.add(base)
        assert r.use_precomputed_table() == 0 and r.precomputed_table().size == 0
        o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2)
        o.set_trained(r.coarse_centroids(), r.pq_centroids(), None)
        assert o.use_precomputed_table() == 0 and o.table() is None
        assert o.add(base)
        o.set_raw(base)
        ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
        Dr, Ir = r.search(q, Rk, nprobe)
        _, _, st = o.search(q, 10, nprobe, recall_num=Rk, has_rank=False, metric=B.METRIC_L2, ctx=ctx,
                            coarse_mode=0, want_stages=True)
        assert st["recall_dis"].tobytes() == Dr.tobytes()
        assert np.array_equal(st["recall_ids"], Ir)
    finally:
        R.ref_set_precomputed_table_max_bytes(saved_r)
        L.go_set_precomputed_table_max_bytes(saved_o)
        R.ref_set_blas_threshold(20)


@pytest.mark.parametrize("kind", ["sift", "gauss"])
def test_gemm_form_is_the_compiled_sgemm(kind):
    """With faiss's default threshold (nq >= 20) the coarse distances come from exhaustive_L2sqr_blas: norms + MKL's
    sgemm_ (faiss:utils/distances.cpp:215-296,303-305).  The restated GEMM form (oracle mode 1 = the device's default
    path) is that computation BIT FOR BIT: one k-ascending fma chain per element up to K = 384, two half chains added
    once for 384 < K <= 768 with K % 8 == 0 (go_gemm_k_split) -- every BASELINE shape (d = 128, 768).  Beyond that MKL's
    blocking is not restated (K = 1024 below: ulp-level differences, documented).  Not covered either: remainder blocks
    of a few rows (nq mod 4096 or nlist mod 1024 below 8), where MKL switches kernels, and K = 384
    exactly with a database block of 9..512 rows (nlist mod 1024 in that range), which MKL already splits in two."""
    R = B.ref()
    R.ref_set_blas_threshold(20)
    try:
        for d, nx, ny in ((32, 64, 512), (64, 100, 1000), (96, 40, 300), (128, 512, 2048), (128, 20, 4160), (200, 33, 700),
                          (256, 64, 1024), (376, 64, 1500), (384, 64, 2048), (392, 64, 512), (512, 50, 1024), (640, 64, 512), (768, 256, 2048),
                          (768, 21, 333), (128, 4160, 1024)):
            rng = np.random.default_rng(d + nx)
            if kind == "sift":
                y = (synth.sift_like(ny, d=d, seed=3) + rng.random((ny, d))).astype(np.float32)   # centroid-like
                x = synth.sift_like(nx, d=d, seed=4)
            else:
                y = (rng.standard_normal((ny, d)) * 3).astype(np.float32)
                x = rng.standard_normal((nx, d)).astype(np.float32)
            k = 32
            D = np.empty((nx, k), np.float32)
            I = np.empty((nx, k), np.int64)
            R.ref_flat_l2_search(d, ny, B._fp(y), nx, B._fp(x), k, B._fp(D), B._ip(I))
            D1, I1 = B.knn_L2sqr(x, y, k, mode=1)
            assert D.tobytes() == D1.tobytes() and np.array_equal(I, I1), (d, nx, ny)
            D0, _ = B.knn_L2sqr(x, y, k, mode=0)
            assert D.tobytes() != D0.tobytes()          # (the library really took its BLAS path)
        # beyond the restated range: close, not identical
        d, nx, ny = 1024, 64, 512
        rng = np.random.default_rng(1)
        y = (rng.standard_normal((ny, d)) * 3).astype(np.float32)
        x = rng.standard_normal((nx, d)).astype(np.float32)
        D = np.empty((nx, 8), np.float32)
        I = np.empty((nx, 8), np.int64)
        R.ref_flat_l2_search(d, ny, B._fp(y), nx, B._fp(x), 8, B._fp(D), B._ip(I))
        D1, I1 = B.knn_L2sqr(x, y, 8, mode=1)
        assert np.allclose(D, D1, rtol=1e-5) and (I == I1).mean() > 0.99
    finally:
        R.ref_set_blas_threshold(20)


def test_unrestated_blas_corners_differ_by_ulps():
    """What the shapes counted by gamma_hip_blas_form_not_restated cost: exhaustive_L2sqr_blas against the
    restated GEMM form (oracle mode 1 = the device's default path) at a query remainder block of 1 / 3 rows (nq = 4097,
    4099), a 3-row database remainder (nlist = 1027) and K = 384 with a 76-row database remainder.  Outside the remainder
    rows / columns every distance is the library's bit for bit; inside, some entries differ -- by a few ulps of the
    distance -- and a probe list can change only where two centroids are that close.  A shape right next to the corner
    (nq = 4103: an 7-row remainder here happens to agree) is not required to differ."""
    R = B.ref()
    R.ref_set_blas_threshold(20)
    rng = np.random.default_rng(0)
    seen_diff = 0
    for d, nx, ny, rows, cols in ((128, 4097, 256, slice(4096, None), slice(None)), (128, 4099, 1027, slice(4096, None), slice(None)),
                                  (384, 40, 1100, slice(None), slice(1024, None))):
        y = (rng.standard_normal((ny, d)) * 3).astype(np.float32)
        x = rng.standard_normal((nx, d)).astype(np.float32)
        D, I = _ref_flat(R.ref_flat_l2_search, x, y, ny)
        D1, I1 = B.knn_L2sqr(x, y, ny, mode=1)
        M = np.empty((nx, ny), np.float32)
        M1 = np.empty((nx, ny), np.float32)
        np.put_along_axis(M, I, D, axis=1)
        np.put_along_axis(M1, I1, D1, axis=1)
        diff = M.view(np.uint32) != M1.view(np.uint32)
        inside = np.zeros_like(diff)
        inside[rows, cols] = True
        assert not (diff & ~inside).any(), (d, nx, ny)          # everything outside the remainder block: the library's bits
        nd = int(diff.sum())
        seen_diff += nd
        if nd:
            rel = np.abs(M[diff].astype(np.float64) - M1[diff]) / np.maximum(np.abs(M[diff]), 1e-30)
            assert rel.max() < 4e-6, (d, nx, ny, rel.max())       # a few ulps of fp32
            assert nd < 0.5 * inside.sum()
    assert seen_diff > 0      # (if MKL on this machine ever agrees everywhere the corners can be dropped from the count)


def _ref_flat(fn, x, y, k):
    D = np.empty((len(x), k), np.float32)
    I = np.empty((len(x), k), np.int64)
    fn(x.shape[1], len(y), B._fp(y), len(x), B._fp(x), k, B._fp(D), B._ip(I))
    return D, I


def test_reservoir_selection_is_the_compiled_librarys_inside_ties():
    """From k = 100 on knn_L2sqr / knn_inner_product collect through ReservoirTopN (faiss:utils/distances.cpp:307-358):
    the restatement (go_reservoir_*, partition_fuzzy_median3) returns the compiled library's labels at every rank on data
    where most keys tie -- both the sequential and the BLAS form -- and k = 99 still takes the heap."""
    R = B.ref()
    try:
        for seed in range(24):
            rng = np.random.default_rng(seed)
            d = int(rng.choice([4, 8, 16, 32]))
            ny = int(rng.choice([150, 600, 1024, 3000, 4096]))
            hi = int(rng.choice([2, 3, 5, 16]))     # few distinct values: many exact ties
            y = rng.integers(0, hi, size=(ny, d)).astype(np.float32)
            nx = int(rng.choice([5, 19, 25]))
            x = rng.integers(0, hi, size=(nx, d)).astype(np.float32)
            for k in (99, 100, 128, 200, 256):
                if k > ny:
                    continue
                for mode, thr in ((0, 1000000), (1, 1)):
                    if mode == 1 and (ny % 1024) and (ny % 1024) < 8:   # sgemm_'s remainder kernel is not restated
                        continue
                    R.ref_set_blas_threshold(thr)
                    Dr, Ir = _ref_flat(R.ref_flat_l2_search, x, y, k)
                    Do, Io = B.knn_L2sqr(x, y, k, mode=mode)
                    assert Dr.tobytes() == Do.tobytes() and np.array_equal(Ir, Io), (seed, d, ny, k, mode)
                R.ref_set_blas_threshold(1000000)
                Dr, Ir = _ref_flat(R.ref_flat_ip_search, x, y, k)
                Do = np.empty((nx, k), np.float32)
                Io = np.empty((nx, k), np.int64)
                B.lib().go_knn_inner_product(B._fp(x), B._fp(y), d, nx, ny, k, B._fp(Do), B._ip(Io))
                assert Dr.tobytes() == Do.tobytes() and np.array_equal(Ir, Io), (seed, d, ny, k, "ip")
    finally:
        R.ref_set_blas_threshold(20)
